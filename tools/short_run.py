#!/usr/bin/env python3
"""GPU: what a 20-step timed region (the driver's `--steps 20 --warmup 5`) loses against steady state, and to what:
cold clocks / caches, or the fixed cost of the bracket (first launch after an idle queue + wake-up of the synchronise)."""
import contextlib, importlib, io, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
dev = torch.device("cuda", 0)
d = bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0")
w = pkg.world; w.configure(["--tensorboard", "0"])
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.utils.set_seed(2020)
    m = pkg.model.LightGCN(w.config, ds).to(dev)
B = 2048
rng = np.random.Generator(np.random.PCG64(1))
T = 500 * B
U = torch.from_numpy(rng.integers(0, ds.n_users, T).astype(np.int32)).to(dev)
P = torch.from_numpy(rng.integers(0, ds.m_items, T).astype(np.int32)).to(dev)
N = torch.from_numpy(rng.integers(0, ds.m_items, T).astype(np.int32)).to(dev)

def run(lo, n):
    return m.fused_epoch(U[lo * B:(lo + n) * B], P[lo * B:(lo + n) * B], N[lo * B:(lo + n) * B], B)

def timed(lo, n, spin=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(lo, n)
    if spin:
        ev = torch.cuda.Event(); ev.record()
        while not ev.query():
            pass
    else:
        torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)

def split(lo, n):
    """host wall time of the bracket vs the GPU's own time between an event before the first launch and one after the last"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    run(lo, n)
    e1.record()
    t_enq = time.perf_counter() - t0
    while not e1.query():
        pass
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return {"host_us": round(t_host * 1e6, 1), "enqueue_us": round(t_enq * 1e6, 1), "gpu_us": round(e0.elapsed_time(e1) * 1e3, 1)}

out = {}
run(0, 5)
out["cold_5_warm_20"] = timed(5, 20)
out["again_20"] = timed(25, 20)
out["again_20_spin"] = timed(45, 20, spin=True)
out["400"] = timed(65, 400)
out["after_400_20"] = timed(465, 20)
out["after_400_20_spin"] = timed(5, 20, spin=True)
time.sleep(1.0)
out["after_sleep_20"] = timed(25, 20)
out["after_sleep_20b"] = timed(45, 20)
print(json.dumps({k: round(v, 1) for k, v in out.items()}))
run(0, 400)
print(json.dumps({"split_20_a": split(5, 20), "split_20_b": split(25, 20), "split_400": split(45, 400), "split_20_c": split(5, 20)}))
