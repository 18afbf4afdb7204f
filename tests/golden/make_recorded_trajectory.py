#!/usr/bin/env python3
"""Turn the reference's RECORDED Gowalla run (3 layers, d = 64, 1000 epochs, a test every 10) into a data
fixture: tests/golden/gowalla/recorded_trajectory.json.

Source (data files the reference holds, read as bytes; nothing of the reference is imported or executed):
  /root/reference/LightGCN_work/code/runs/07-10-17h52m32s--lgn/Test/{Recall,NDCG,Precision}@[20]/20/events.out.tfevents.*
written by `Procedure.Test` through `w.add_scalars` (Procedure.py:196-204 of the reference) and
  /root/reference/LightGCN_work/README.md:93  (the published end state of the same configuration).

No TensorFlow / tensorboard here, so the two container formats are decoded by hand:
  TFRecord framing : u64 length | u32 masked-crc(length) | payload | u32 masked-crc(payload)
  Event (protobuf) : 1 = wall_time (double), 2 = step (varint), 5 = summary { 1 = value { 1 = tag, 2 = simple_value (float) } }
The CRCs are not verified (no crc32c in the image); a truncated record ends the scan.

Run in the build container only (the reference does not exist on the GPU box); the JSON is committed.
"""
import glob
import json
import os
import struct
import sys

RUN = "/root/reference/LightGCN_work/code/runs/07-10-17h52m32s--lgn/Test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gowalla", "recorded_trajectory.json")


def records(blob):
    off = 0
    while off + 12 <= len(blob):
        (n,) = struct.unpack_from("<Q", blob, off)
        if off + 12 + n + 4 > len(blob):
            return
        yield blob[off + 12: off + 12 + n]
        off += 12 + n + 4


def varint(b, i):
    v = s = 0
    while True:
        c = b[i]; i += 1
        v |= (c & 0x7F) << s; s += 7
        if c < 0x80:
            return v, i


def fields(b):
    """(field number, wire type, value) of one protobuf message; nested messages stay bytes."""
    i = 0
    while i < len(b):
        key, i = varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]; i += 8
        elif wt == 5:
            v = b[i:i + 4]; i += 4
        elif wt == 2:
            n, i = varint(b, i); v = b[i:i + n]; i += n
        else:
            raise ValueError(f"wire type {wt}")
        yield f, wt, v


def scalars(path):
    out = []
    for rec in records(open(path, "rb").read()):
        wall = step = None
        vals = []
        for f, wt, v in fields(rec):
            if f == 1 and wt == 1:
                wall = struct.unpack("<d", v)[0]
            elif f == 2 and wt == 0:
                step = v
            elif f == 5 and wt == 2:
                for f2, wt2, v2 in fields(v):
                    if f2 == 1 and wt2 == 2:
                        tag = val = None
                        for f3, wt3, v3 in fields(v2):
                            if f3 == 1 and wt3 == 2:
                                tag = v3.decode()
                            elif f3 == 2 and wt3 == 5:
                                val = struct.unpack("<f", v3)[0]
                        if val is not None:
                            vals.append((tag, val))
        for tag, val in vals:
            out.append({"epoch": int(step or 0), "wall_time": wall, "tag": tag, "value": float(val)})
    return out


def main():
    if not os.path.isdir(RUN):
        sys.exit("reference run directory not present (build container only)")
    series = {}
    for key, sub in (("recall", "Recall@[20]"), ("ndcg", "NDCG@[20]"), ("precision", "Precision@[20]")):
        (path,) = glob.glob(os.path.join(glob.escape(os.path.join(RUN, sub, "20")), "events.out.tfevents.*"))
        pts = scalars(path)
        series[key] = {"file": os.path.relpath(path, "/root/reference"), "tag": pts[0]["tag"],
                       "epochs": [p["epoch"] for p in pts], "values": [p["value"] for p in pts],
                       "wall_time": [p["wall_time"] for p in pts]}
    ep = series["recall"]["epochs"]
    assert all(series[k]["epochs"] == ep for k in series), "the three series do not share their epochs"
    doc = {
        "what": "Recorded 1000-epoch Gowalla run of the reference (lgn, layer 3, recdim 64, bpr_batch 2048, seed 2020), "
                "tensorboard scalars written by Procedure.Test every 10 epochs; values are float32 as stored",
        "made_by": "tests/golden/make_recorded_trajectory.py",
        "readme_published": {"source": "LightGCN_work/README.md:93", "layer": 3, "recall": 0.1824, "ndcg": 0.1547, "precision": 0.05589},
        "points": len(ep), "epochs": ep,
        "recall": series["recall"]["values"], "ndcg": series["ndcg"]["values"], "precision": series["precision"]["values"],
        "wall_time": series["recall"]["wall_time"],
        "files": {k: series[k]["file"] for k in series}, "tags": {k: series[k]["tag"] for k in series},
    }
    r = doc["recall"]
    doc["summary"] = {"recall_epoch0": r[0], "recall_epoch10": r[ep.index(10)] if 10 in ep else None,
                      "recall_last": r[-1], "last_epoch": ep[-1], "recall_max": max(r), "recall_argmax_epoch": ep[r.index(max(r))],
                      "first_epoch_recall_ge_0.18": next((e for e, v in zip(ep, r) if v >= 0.18), None),
                      "seconds_per_epoch_recorded": (doc["wall_time"][-1] - doc["wall_time"][0]) / max(1, ep[-1] - ep[0])}
    json.dump(doc, open(OUT, "w"), indent=1)
    print(json.dumps(doc["summary"], indent=1))


if __name__ == "__main__":
    main()
