#!/usr/bin/env python3
"""GPU (one device): the COMPUTE of one rank's data-parallel step at world sizes 1..8, without the collective.
Rank 0's part 1 (forward + its 2048-triplet shard) runs for real; the all-gather is replaced by a device copy of
rank 0's block into every slot of the gathered buffer (same sizes, same kernels downstream: k_scatter over
world x 2048 triplets, k_g32, a first backward layer whose input gets denser with the world size, two more layers,
Adam).  Reports steps/s per rank and the weak-scaling efficiency the compute alone allows."""
import contextlib, importlib, io, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
import bench
sys.argv = [sys.argv[0]]
pkg = importlib.import_module(bench.PKG)
L = pkg._lib
dev = torch.device("cuda", 0)
d = bench.materialize_gowalla(bench.GOWALLA_NPZ, "/tmp/lgcn_bench_data/gowalla_r0")
out = {}
B = 2048
steps = 200
for world in (1, 2, 4, 8):
    Bg = B * world
    w = pkg.world; w.configure(["--tensorboard", "0", "--bpr_batch", str(B)])      # per-rank batch: what bench.py --gpus N configures
    with contextlib.redirect_stdout(io.StringIO()):
        ds = pkg.dataloader.Loader(w.config, path=d)
        pkg.utils.set_seed(2020)
        m = pkg.model.LightGCN(w.config, ds).to(dev)
        u, p, n = pkg.Procedure.sample_epoch_to_device(ds, dev)
    st = m._state(max_batch=Bg, need_ctx=True, dp_world=world)
    lib = L.load()
    L.check(lib.lgcn_ctx_set_dp_local(st['ctx'], 1), "set_dp_local")      # as lgcn_train_epoch_dp runs it: own rows added in part 1
    blk = pkg.parallel.block_numel(Bg, world, m.latent_dim)
    gathered = torch.zeros(world * blk, dtype=torch.float32, device=dev)
    loss = torch.empty(3, dtype=torch.float32, device=dev)
    stream = L.current_stream()

    def step(t):
        lo = t * Bg
        uu, pp, nn = u[lo:lo + Bg], p[lo:lo + Bg], n[lo:lo + Bg]
        L.check(lib.lgcn_train_step_dp_part1(st['ctx'], L.tp(uu), L.tp(pp), L.tp(nn), Bg, world, 0, stream), "part1")
        gathered.view(world, blk).copy_(st['contrib'][:blk].unsqueeze(0).expand(world, blk))      # stand-in for the all-gather
        L.check(lib.lgcn_train_step_dp_part2(st['ctx'], L.tp(uu), L.tp(pp), L.tp(nn), Bg, world, L.tp(gathered), L.tp(loss), stream), "part2")
    nsteps = min(steps, len(u) // Bg - 6)
    for t in range(5):
        step(t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(5, 5 + nsteps):
        step(t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[f"world{world}"] = {"steps_per_sec_per_rank": nsteps / dt, "us_per_step": dt / nsteps * 1e6, "steps": nsteps}
    del m, st
base = out["world1"]["steps_per_sec_per_rank"]
for k in out:
    out[k]["compute_only_weak_scaling_efficiency"] = out[k]["steps_per_sec_per_rank"] / base
print(json.dumps(out))
