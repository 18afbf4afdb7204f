#!/usr/bin/env python3
"""bench.py -- BPR training steps/sec of the MI355X-native LightGCN hot path.

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one BPRLoss.stageOne (utils.py:53-64 of the reference): K-layer
propagation + BPR loss + backward + Adam on one batch of B triplets, with the
epoch's triplets already resident in HBM.  Workload at N=1 = BASELINE.json
configs[1]: Gowalla (real graph, reconstructed; 29 858 x 40 981, 810 128 train
interactions), 3 layers, dim 64, BPR batch 2048.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
PKG = "graph-and-sequential-recommendation-systems_amd"
GOWALLA_NPZ = os.path.join(REPO, "tests", "golden", "gowalla", "gowalla.npz")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable


def materialize_gowalla(npz, dst):
    """tests/golden/gowalla/gowalla.npz (interaction lists, file order) -> train.txt/test.txt."""
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(dst, "train.txt")) and os.path.exists(os.path.join(dst, "test.txt")):
        return dst
    z = np.load(npz)
    for name in ("train", "test"):
        users, ptr, items = z[name + "_users"], z[name + "_ptr"], z[name + "_items"]
        with open(os.path.join(dst, name + ".txt.tmp"), "w") as f:
            for k, u in enumerate(users):
                f.write(str(int(u)) + " " + " ".join(map(str, items[ptr[k]:ptr[k + 1]].tolist())) + "\n")
        os.replace(os.path.join(dst, name + ".txt.tmp"), os.path.join(dst, name + ".txt"))
    return dst


def synth_bipartite(n_users, m_items, E, seed, dst):
    """Power-law bipartite graph of a given shape: user degrees ~ Zipf clipped to
    [1, m_items/4] rescaled to E, items by popularity, no duplicates, PCG64(seed)."""
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(dst, "train.txt")):
        return dst
    rng = np.random.Generator(np.random.PCG64(seed))
    raw = rng.zipf(1.8, n_users).astype(np.float64)
    raw = np.clip(raw, 1, m_items // 4)
    deg = np.maximum(1, np.floor(raw * (E / raw.sum()))).astype(np.int64)
    deg = np.minimum(deg, m_items // 4)
    diff = E - int(deg.sum())
    idx = rng.permutation(n_users)
    i = 0
    while diff != 0 and i < 50 * n_users:
        u = idx[i % n_users]
        if diff > 0 and deg[u] < m_items // 4:
            deg[u] += 1; diff -= 1
        elif diff < 0 and deg[u] > 1:
            deg[u] -= 1; diff += 1
        i += 1
    pop = 1.0 / np.arange(1, m_items + 1) ** 0.9
    pop = pop[rng.permutation(m_items)]
    pop /= pop.sum()
    cdf = np.cumsum(pop)
    with open(os.path.join(dst, "train.txt.tmp"), "w") as f, open(os.path.join(dst, "test.txt"), "w") as ft:
        for u in range(n_users):
            k = int(deg[u])
            got = np.unique(np.searchsorted(cdf, rng.random(int(k * 1.3) + 4)))
            while len(got) < k:
                got = np.unique(np.concatenate([got, np.searchsorted(cdf, rng.random(k))]))
            got = np.minimum(rng.permutation(got)[:k], m_items - 1)
            f.write(str(u) + " " + " ".join(map(str, np.unique(got).tolist())) + "\n")
            ft.write(f"{u} {int(rng.integers(0, m_items))}\n")
    os.replace(os.path.join(dst, "train.txt.tmp"), os.path.join(dst, "train.txt"))
    return dst


WORKLOADS = {
    # name: (n_users, m_items, E, K, d, B)
    "gowalla": (29858, 40981, 810128, 3, 64, 2048),          # BASELINE configs[1] (real graph)
    "yelp2018-shaped": (31668, 38048, 1237259, 3, 64, 8192),  # configs[2], synthetic graph of that shape
    "amazon-book-shaped": (52643, 91599, 2380730, 4, 128, 2048),  # configs[3]
}


def spmm_bytes(N, nnz, d, s):
    """algorithmic (compulsory) bytes of one SpMM launch."""
    return nnz * 8 + (N + 1) * 4 + 2 * N * d * s


def step_bytes(N, nnz, d, s, K, B):
    return 2 * K * spmm_bytes(N, nnz, d, s) + 2 * (K + 2) * N * d * s + (12 * B + 3 * B * d * s + 24 * B * d) + 28 * N * d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="gowalla", choices=list(WORKLOADS))
    ap.add_argument("--act_dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--xcd_remap", type=int, default=1)
    ap.add_argument("--row_order", default="cocluster", choices=["natural", "rcm", "cocluster"])
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_secondary", action="store_true", help="skip the extra run with the other activation dtype")
    ap.add_argument("--dp_reduce", default="rows", choices=["rows", "dense"],
                    help="data-parallel gradient exchange: all-gather of gradient rows (default) or dense all-reduce")
    ap.add_argument("--force_dp", action="store_true", help="use the data-parallel step (RCCL all-gather) even at world size 1")
    ap.add_argument("--cpu_seconds", type=float, default=12.0)
    ap.add_argument("--data_dir", default=os.path.join(tempfile.gettempdir(), "lgcn_bench_data"))
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dp = world > 1 or a.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    sys.argv = [sys.argv[0]]
    pkg = importlib.import_module(PKG)
    w = pkg.world
    n_users, m_items, E, K, d, B = WORKLOADS[a.workload]
    w.configure(["--layer", str(K), "--recdim", str(d), "--bpr_batch", str(B), "--act_dtype", a.act_dtype,
                 "--xcd_remap", str(a.xcd_remap), "--row_order", a.row_order, "--tensorboard", "0",
                 "--dataset", a.workload])
    data_dir = os.path.join(a.data_dir, f"{a.workload}_r{rank}")
    if a.workload == "gowalla" and os.path.exists(GOWALLA_NPZ):
        materialize_gowalla(GOWALLA_NPZ, data_dir)
        data_kind = "gowalla interaction graph (real, reconstructed from the reference's cache); random-init weights, seed 2020"
    else:
        synth_bipartite(n_users, m_items, E, 2020, data_dir)
        data_kind = "synthetic"
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        ds = pkg.dataloader.Loader(w.config, path=data_dir)
        pkg.sampling.seed(2020)
        pkg.utils.set_seed(2020)
        model = pkg.model.LightGCN(w.config, ds).to(dev)
    N = ds.n_users + ds.m_items
    nnz = int(ds.getSparseGraphCSR().nnz)

    # triplets for warmup + timed steps, resident in HBM before the timed region
    Bg = B * world                                   # weak scaling: per-GPU batch fixed
    need = (a.warmup + a.steps) * Bg
    us, ps, ns = [], [], []
    have = 0
    while have < need:
        u, p, n = pkg.Procedure.sample_epoch_to_device(ds, dev)
        us.append(u); ps.append(p); ns.append(n); have += len(u)
    users, pos, neg = torch.cat(us)[:need], torch.cat(ps)[:need], torch.cat(ns)[:need]

    if not use_dp:
        def run(lo, steps):
            return model.fused_epoch(users[lo:lo + steps * B], pos[lo:lo + steps * B], neg[lo:lo + steps * B], B)
    else:
        dp = pkg.parallel.DataParallelBPR(model, w.config, reduce=a.dp_reduce)
        dp.lazy = True

        def run(lo, steps):
            out = []
            for s in range(steps):
                sl = slice(lo + s * Bg, lo + (s + 1) * Bg)
                out.append(dp.stageOne(users[sl], pos[sl], neg[sl]))
            return torch.stack(out)

    def barrier():
        if use_dp:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, a.warmup)
    barrier()
    t0 = time.perf_counter()
    losses = run(a.warmup * Bg, a.steps)
    barrier()
    dt = time.perf_counter() - t0
    if use_dp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    model.check_device_errors()
    first_loss = float(losses.reshape(-1, 3)[0, 0]) if losses.dim() == 2 else float(losses[0])
    last_loss = float(losses.reshape(-1, 3)[-1, 0]) if losses.dim() == 2 else float(losses[-1])

    out = None
    if rank == 0:
        s = 4 if a.act_dtype == "fp32" else 2
        steps_per_sec = a.steps / dt
        out = {
            "metric": f"BPR training steps/sec (one step = K-layer LightGCN propagation + BPR loss + backward + Adam on a batch of {B} triplets per GPU)",
            "value": steps_per_sec * world, "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.act_dtype == "fp32" else "f32 accumulate, bf16 activation storage",
            "data": data_kind,
            "config": {"workload": f"{a.workload}: {ds.n_users} users x {ds.m_items} items, {ds.trainDataSize} train "
                                   f"interactions, nnz(A_hat)={nnz}, layers={K}, dim={d}, bpr_batch={B} per GPU",
                       "global_batch": Bg, "per_gpu_batch": B, "global_steps_per_sec": steps_per_sec,
                       "triplets_per_sec": steps_per_sec * Bg, "parallelism": f"dp{world} (replicated tables, "
                       "batch-sharded, gradient-row all-gather over RCCL)" if use_dp else "single GPU",
                       "act_dtype": a.act_dtype, "xcd_remap": a.xcd_remap, "row_order": a.row_order,
                       "first_loss": first_loss, "last_loss": last_loss},
            "step_algorithmic_bytes": step_bytes(N, nnz, d, s, K, B),
            "step_roofline_frac": step_bytes(N, nnz, d, s, K, B) * steps_per_sec / (HBM_PEAK_GBS * 1e9),
        }

    # ---- the same workload with the other activation storage type (single GPU; reported, not the headline)
    if rank == 0 and not use_dp and not a.no_secondary:
        other = "bf16" if a.act_dtype == "fp32" else "fp32"
        cfg2 = dict(w.config); cfg2['act_dtype'] = other
        with contextlib.redirect_stdout(io.StringIO()):
            pkg.utils.set_seed(2020)
            model2 = pkg.model.LightGCN(cfg2, ds).to(dev)
        model2.fused_epoch(users[:a.warmup * B], pos[:a.warmup * B], neg[:a.warmup * B], B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model2.fused_epoch(users[a.warmup * B:(a.warmup + a.steps) * B], pos[a.warmup * B:(a.warmup + a.steps) * B],
                           neg[a.warmup * B:(a.warmup + a.steps) * B], B)
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
        out["config"][f"{other}_activation_storage_steps_per_sec"] = a.steps / dt2
        del model2

    # ---- dominant kernel (dense CSR-SpMM layer) timed live with HIP events on the launch stream
    if rank == 0:
        L = pkg._lib
        st = model._state(max_batch=B, need_ctx=True, dp_world=world)
        adt = 0 if a.act_dtype == "fp32" else 1
        tdt = torch.float32 if adt == 0 else torch.bfloat16
        x = (torch.randn(N, d, device=dev) * 0.1).to(tdt)     # same graph object (same row order) as the step
        y = torch.empty_like(x)
        lib = L.load()
        stream = L.current_stream()

        def spmm_once():
            L.check(lib.lgcn_spmm_csr(st['graph'].handle, L.tp(x), adt, L.tp(y), adt, d, stream), "spmm")
        for _ in range(10):
            spmm_once()
        reps = 100
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            spmm_once()
        e1.record()
        torch.cuda.synchronize()
        t_spmm = e0.elapsed_time(e1) / reps * 1e-3
        bytes_spmm = spmm_bytes(N, nnz, d, s)
        achieved = bytes_spmm / t_spmm / 1e9
        traffic = None
        tf = os.path.join(REPO, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f"{a.workload}:{a.act_dtype}:k_spmm", None)
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": f"k_spmm<{d},{'float' if adt == 0 else 'bf16'}> (dense CSR-SpMM layer)",
                           "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": traffic, "algorithmic_bytes_per_launch": bytes_spmm,
                           "avg_launch_us": t_spmm * 1e6,
                           "gather_bytes_upper_bound": nnz * d * s}

    # ---- CPU baseline: the oracle (C/OpenMP port of the reference path) on the host cores
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as orc
        adj = ds.getSparseGraphCSR()
        pkg.utils.set_seed(2020)
        e0 = model._table.detach().cpu().numpy().copy()
        tr = orc.Trainer(ds.n_users, adj.indptr, adj.indices, adj.data, e0, K, w.config['decay'], w.config['lr'])
        hu, hp, hn = (t[:64 * B].cpu().numpy().astype(np.int64) for t in (users, pos, neg))
        tr.stageOne(hu[:B], hp[:B], hn[:B])                 # warm-up step
        t0 = time.perf_counter()
        n_cpu = 0
        while n_cpu < 63 and (time.perf_counter() - t0) < a.cpu_seconds:
            lo = (n_cpu + 1) * B
            tr.stageOne(hu[lo:lo + B], hp[lo:lo + B], hn[lo:lo + B])
            n_cpu += 1
        dt_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n_cpu / dt_cpu, "unit": "steps/s", "cores": orc.num_threads(), "kind": "port",
                               "sample": f"{n_cpu} consecutive stageOne steps of the same workload (same graph, same "
                                         f"triplets, B={B}) through oracle/lgcn_oracle.c (C + OpenMP, fp32), "
                                         f"{dt_cpu:.1f} s; the reference's own torch-CPU path measured 1.45-2.22 "
                                         "steps/s on 8 cores (BASELINE.md)"}
    if rank == 0:
        print(json.dumps(out))
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
