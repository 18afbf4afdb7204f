#!/usr/bin/env python3
"""End-to-end Gowalla epochs (Procedure.BPR_train_original: sample -> upload -> shuffle -> 394 fused steps) when the triplets come from the
HOST -- the paths on which something crosses PCIe every epoch:
  * --sampler python: utils.py:84-110 on the numpy-legacy stream (the reference's default without cppimport), a host loop, then the
    [T,3] int32 triplets (9.7 MB) go up;
  * --gpu_sampler 0: sampling.cpp's stream drawn on the host (the native plugin), same upload;
next to the default (device sampler, nothing uploaded).  Each with the epoch prefetch on and off, fp32 and bf16 storage."""
import argparse, importlib, io, contextlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import materialize_gowalla, GOWALLA_NPZ

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=8)
ap.add_argument("--out", default=None)
a = ap.parse_args()
sys.argv = [sys.argv[0]]
import numpy as np
import torch
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
w = pkg.world
d = materialize_gowalla(GOWALLA_NPZ, "/tmp/lgcn_hs_gowalla")
rows = []
for act in ("fp32", "bf16"):
    for name, flags in (("device sampler (default)", []), ("host python-mode sampler", ["--sampler", "python"]),
                        ("host cpp-mode sampler", ["--gpu_sampler", "0"])):
        for prefetch in (1, 0):
            w.configure(["--dataset", "gowalla", "--tensorboard", "0", "--act_dtype", act, "--prefetch_epoch", str(prefetch),
                         "--checkpoint_dir", "/tmp/lgcn_hs_ckpt"] + flags)
            with contextlib.redirect_stdout(io.StringIO()):
                ds = pkg.dataloader.Loader(w.config, path=d)
                pkg.sampling.seed(w.seed); pkg.utils.set_seed(w.seed)
                model = pkg.model.LightGCN(w.config, ds).to(w.device)
                bpr = pkg.utils.BPRLoss(model, w.config)
                secs = []
                for e in range(1, a.epochs + 1):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    pkg.Procedure.BPR_train_original(ds, model, bpr, e)
                    torch.cuda.synchronize(); secs.append(time.perf_counter() - t0)
            rest = secs[1:]
            steps = (ds.trainDataSize + w.config['bpr_batch_size'] - 1) // w.config['bpr_batch_size']
            r = {"act_dtype": act, "triplets": name, "prefetch": prefetch, "ms_per_epoch": 1e3 * sum(rest) / len(rest),
                 "steps_per_sec": steps * len(rest) / sum(rest), "first_epoch_ms": 1e3 * secs[0]}
            rows.append(r); print(json.dumps(r), flush=True)
            ds._lgcn_next_epoch = None
            del model, bpr, ds
# the upload alone: [T,3] int32 from pageable host memory, as sample_epoch_to_device does it
S = np.zeros((810128, 3), np.int32)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); x = torch.from_numpy(S).to("cuda"); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
up = {"upload_ms_9.7MB_pageable": 1e3 * min(ts)}
print(json.dumps(up))
if a.out:
    json.dump({"rows": rows, **up}, open(a.out, "w"), indent=1)
