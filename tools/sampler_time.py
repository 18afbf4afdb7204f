#!/usr/bin/env python3
"""GPU: time one epoch of the device sampler (lgcn_sample_negative_device: stream expansion + triplets) on Gowalla and on the
Yelp2018-shaped synthetic graph; the host sampler beside it.  Prints one JSON line."""
import importlib, io, contextlib, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bench
BIG = "--c5" in sys.argv
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
w = pkg.world; w.configure(["--tensorboard", "0", "--checkpoint_dir", "/tmp/lgcn_samp_ckpt"])
dev = torch.device("cuda", 0)
out = {}
d = bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0")
with contextlib.redirect_stdout(io.StringIO()):
    sets = {"gowalla": pkg.dataloader.Loader(w.config, path=d), "yelp2018-shaped": bench.synthetic_dataset(pkg, "yelp2018-shaped", w.config, dev)}
    if BIG:
        sets = {"synthetic-10m": bench.synthetic_dataset(pkg, "synthetic-10m", w.config, dev)}
for name, ds in sets.items():
    S = pkg.sampling
    S.seed(2020)
    csr = ds.pos_csr()
    S.sample_negative_device(ds.n_users, ds.m_items, ds.trainDataSize, csr, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 1 if BIG else 5
    for _ in range(reps):
        S.sample_negative_device(ds.n_users, ds.m_items, ds.trainDataSize, csr, dev)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    if not BIG:
        S.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, csr, 1)
    t_host = time.perf_counter() - t0
    out[name] = {"device_ms": t_dev * 1e3, "host_ms": t_host * 1e3, "triplets": int(ds.n_users * (ds.trainDataSize // ds.n_users))}
print(json.dumps(out))
