#!/usr/bin/env python3
"""Train Gowalla (BASELINE configs[1]: 3 layers, dim 64, batch 2048, seed 2020, cpp-mode
sampler) for N epochs on the GPU through the product's own Procedure.BPR_train_original /
Procedure.Test and print the trajectory next to the reference's:

* epochs 1..10: tests/golden/gowalla/golden_long.json (captured by IMPORTING the reference on CPU — same code,
  same seed: the 1e-4 bar of north_star applies);
* every 10th epoch up to 990: tests/golden/gowalla/recorded_trajectory.json (the reference's own recorded
  1000-epoch tensorboard run, main.py:185-225: a test BEFORE training epoch e whenever e % 10 == 0, i.e. after e
  trained epochs) and README.md:93's published end state. That run is a NEIGHBOUR (another machine, upstream's
  sampler build): it is compared inside a band, not to 1e-4.
"""
import argparse, importlib, io, contextlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import materialize_gowalla, GOWALLA_NPZ

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=10)
ap.add_argument("--act_dtype", default="fp32")
ap.add_argument("--out", default=None)
ap.add_argument("--prefetch_epoch", type=int, default=0)
ap.add_argument("--test_every", type=int, default=10)
ap.add_argument("--quiet", type=int, default=0)
ap.add_argument("--reg_rows", default="propagated", choices=["propagated", "ego"],
                help="propagated = this reference's loss (model.py:173); ego = upstream LightGCN's (the recorded run's)")
a = ap.parse_args()
sys.argv = [sys.argv[0]]
import torch
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
w = pkg.world
w.configure(["--dataset", "gowalla", "--tensorboard", "0", "--act_dtype", a.act_dtype, "--prefetch_epoch", str(a.prefetch_epoch),
             "--reg_rows", a.reg_rows, "--checkpoint_dir", "/tmp/lgcn_traj_ckpt"])
d = materialize_gowalla(GOWALLA_NPZ, "/tmp/lgcn_traj_gowalla")
gdir = os.path.join(REPO, "tests", "golden", "gowalla")
ref = json.load(open(os.path.join(gdir, "golden_long.json")))
rec = json.load(open(os.path.join(gdir, "recorded_trajectory.json")))
rec_at = {e: {k: rec[k][i] for k in ("recall", "ndcg", "precision")} for i, e in enumerate(rec["epochs"])}
t_wall0 = time.time()
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.sampling.seed(w.seed); pkg.utils.set_seed(w.seed)
    model = pkg.model.LightGCN(w.config, ds).to(w.device)
    bpr = pkg.utils.BPRLoss(model, w.config)
    r0 = pkg.Procedure.Test(ds, model, 0)
rows = [{"epoch": 0, "test": {k: float(v[0]) for k, v in r0.items()}, "ref_test": {k: v[0] for k, v in ref["test_epoch0"].items()},
         "recorded": rec_at.get(0)}]
t_train = t_test = 0.0
t_to_018 = None
torch.cuda.synchronize(); t_loop0 = time.time()
for e in range(1, a.epochs + 1):
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        info = pkg.Procedure.BPR_train_original(ds, model, bpr, e)
    torch.cuda.synchronize(); dt = time.time() - t0; t_train += dt
    rt = ref["trajectory"][e - 1] if e <= len(ref["trajectory"]) else {}
    row = {"epoch": e, "info": info, "seconds": dt}
    if rt:
        row["ref_avg_loss"] = rt.get("avg_loss"); row["ref_steps"] = rt.get("steps")
    if "test" in rt or e == a.epochs or e % a.test_every == 0:
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            r = pkg.Procedure.Test(ds, model, e)
        row["test_seconds"] = time.time() - t0; t_test += row["test_seconds"]
        row["test"] = {k: float(v[0]) for k, v in r.items()}
        row["elapsed_since_loop_start"] = time.time() - t_loop0
        if t_to_018 is None and row["test"]["recall"] >= 0.18:
            t_to_018 = {"epoch": e, "seconds": row["elapsed_since_loop_start"]}
        if "test" in rt:
            row["ref_test"] = {k: v[0] for k, v in rt["test"].items()}
            row["abs_diff"] = {k: abs(row["test"][k] - row["ref_test"][k]) for k in row["test"]}
        if e in rec_at:
            row["recorded"] = rec_at[e]
            row["recorded_diff"] = {k: row["test"][k] - rec_at[e][k] for k in row["test"]}
    rows.append(row)
    if "test" in row and not a.quiet or e == a.epochs:
        print(json.dumps(row), flush=True)
tests = [r for r in rows if "test" in r]
best = max(tests, key=lambda r: r["test"]["recall"])
summary = {"epochs": a.epochs, "act_dtype": a.act_dtype, "reg_rows": a.reg_rows, "prefetch_epoch": a.prefetch_epoch,
           "train_seconds_total": t_train, "test_seconds_total": t_test, "loop_seconds": time.time() - t_loop0,
           "wall_seconds_incl_setup": time.time() - t_wall0, "seconds_per_epoch": t_train / max(1, a.epochs),
           "final": tests[-1]["test"], "final_epoch": tests[-1]["epoch"],
           "max_recall": best["test"]["recall"], "max_recall_epoch": best["epoch"],
           "time_to_recall_0.18": t_to_018,
           "at_epoch_990": next((r["test"] for r in tests if r["epoch"] == 990), None),
           "recorded_epoch_990": rec_at.get(990), "recorded_max_recall": max(rec["recall"]),
           "recorded_first_epoch_recall_ge_0.18": rec["summary"]["first_epoch_recall_ge_0.18"],
           "recorded_seconds_per_epoch": rec["summary"]["seconds_per_epoch_recorded"],
           "readme_published": rec["readme_published"],
           "max_abs_recorded_diff_recall_after_epoch_100": max((abs(r["recorded_diff"]["recall"]) for r in tests
                                                                if "recorded_diff" in r and r["epoch"] >= 100), default=None)}
print(json.dumps(summary))
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump({"summary": summary, "trajectory": [r for r in rows if "test" in r],
               "epoch_seconds": [r.get("seconds") for r in rows[1:]]}, open(a.out, "w"), indent=1)
