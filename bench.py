#!/usr/bin/env python3
"""bench.py -- BPR training steps/sec of the MI355X-native LightGCN hot path.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one BPRLoss.stageOne (utils.py:53-64 of the reference): K-layer
propagation + BPR loss + backward + Adam on one batch of B triplets, with the
epoch's triplets already resident in HBM.  Workload at N=1 = BASELINE.json
configs[1]: Gowalla (real graph, reconstructed; 29 858 x 40 981, 810 128 train
interactions), 3 layers, dim 64, BPR batch 2048.  Rank 0 prints ONE JSON line.

N > 1: one process per GPU over RCCL.  Started under torch.distributed.run the
script reads RANK/LOCAL_RANK/WORLD_SIZE; started plainly (`python bench.py
--gpus 8`) it launches `python -m torch.distributed.run ...` itself as a CHILD
process before anything touches the GPU and relays rank 0's line.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
PKG = "graph-and-sequential-recommendation-systems_amd"
GOWALLA_NPZ = os.path.join(REPO, "tests", "golden", "gowalla", "gowalla.npz")
TRAFFIC_JSON = os.path.join(REPO, "profiles", "hbm_traffic.json")
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 achievable
FP32_MFMA_PEAK_TF = 157.3      # v_mfma_f32_32x32x2_f32 (same guide)
BF16_MFMA_PEAK_TF = 157.3 * 16  # dense bf16 MFMA = 16 x the fp32 rate (~2.5 PFLOP/s)
GOLDEN_LONG = os.path.join(REPO, "tests", "golden", "gowalla", "golden_long.json")


def materialize_gowalla(npz, dst):
    """tests/golden/gowalla/gowalla.npz (interaction lists, file order) -> train.txt/test.txt."""
    os.makedirs(dst, exist_ok=True)
    if os.path.exists(os.path.join(dst, "train.txt")) and os.path.exists(os.path.join(dst, "test.txt")):
        return dst
    z = np.load(npz)
    for name in ("train", "test"):
        users, ptr, items = z[name + "_users"], z[name + "_ptr"], z[name + "_items"]
        tmp = os.path.join(dst, f"{name}.txt.tmp.{os.getpid()}")
        with open(tmp, "w") as f:
            for k, u in enumerate(users):
                f.write(str(int(u)) + " " + " ".join(map(str, items[ptr[k]:ptr[k + 1]].tolist())) + "\n")
        os.replace(tmp, os.path.join(dst, name + ".txt"))
    return dst


WORKLOADS = {
    # name: (n_users, m_items, E, K, d, B, default steps, default warmup)
    "gowalla": (29858, 40981, 810128, 3, 64, 2048, 400, 20),                 # BASELINE configs[1] (real graph)
    "yelp2018-shaped": (31668, 38048, 1237259, 3, 64, 8192, 200, 20),        # configs[2], synthetic graph of that shape
    "amazon-book-shaped": (52643, 91599, 2380730, 4, 128, 2048, 200, 20),    # configs[3]
    "synthetic-10m": (10_000_000, 1_000_000, 200_000_000, 3, 256, 2048, 20, 3),   # configs[4]
}


def spmm_bytes(N, nnz, d, s):
    """algorithmic (compulsory) bytes of one SpMM launch."""
    return nnz * 8 + (N + 1) * 4 + 2 * N * d * s


def step_bytes(N, nnz, d, s, K, B):
    return 2 * K * spmm_bytes(N, nnz, d, s) + 2 * (K + 2) * N * d * s + (12 * B + 3 * B * d * s + 24 * B * d) + 28 * N * d


def synthetic_dataset(pkg, name, config, device):
    """Seeded synthetic graph of a BASELINE shape, CSR arrays straight into the dataset object."""
    n_users, m_items, E = WORKLOADS[name][:3]
    indptr, indices = pkg.synthetic.power_law_bipartite(n_users, m_items, E, seed=2020, device=device)
    rng = np.random.Generator(np.random.PCG64(7))
    test = {int(u): [int(rng.integers(0, m_items))] for u in range(min(n_users, 1000))}
    return pkg.dataloader.CsrLoader(indptr, indices, m_items, test_dict=test, config=config)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child job.  The parent has
    not touched the GPU (and never will); it relays the child's output and exit code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def read_traffic(key, lib_hash):
    """PMC-measured memory-side bytes per launch of the dominant kernel, valid only for the
    library build it was measured on (profiles/pmc_traffic.py writes it)."""
    try:
        rec = json.load(open(TRAFFIC_JSON)).get(key)
    except Exception:
        return None, "no profiles/hbm_traffic.json"
    if not isinstance(rec, dict):
        return None, f"no PMC record for {key}"
    if rec.get("lib_hash") != lib_hash:
        return None, f"PMC record for {key} was taken on another build of the kernels (stale); re-run profiles/pmc_traffic.py"
    return rec.get("bytes"), rec.get("source")


_T0 = time.perf_counter()


def progress(msg):
    """stderr progress line (a multi-minute setup must not look hung to a watchdog)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="gowalla", choices=list(WORKLOADS))
    ap.add_argument("--act_dtype", default=None, choices=["fp32", "bf16", "fp8"],
                    help="activation storage (accumulation, parameters, Adam stay fp32).  Default: bf16 for gowalla -- BASELINE.json configs[1] "
                         "names it ('Gowalla 3-layer dim=64 bf16') and it holds the metric (Recall@20 within 3e-6 of the reference at 10 epochs, "
                         "inside the recorded run's band at 1000: DESIGN 2a) -- with the fp32 run of the same region reported beside it; fp32 elsewhere")
    ap.add_argument("--xcd_remap", type=int, default=1)
    ap.add_argument("--dense_last", default="auto", choices=["auto", "0", "1"], help="last forward layer: on the batch rows only (0) or densely (1)")
    ap.add_argument("--row_order", default=None, choices=["natural", "rcm", "cocluster", "xcd"])
    ap.add_argument("--hub_nnz", type=int, default=0, help="hub plan threshold of the batch-row kernel (0: library default, < 0: off)")
    ap.add_argument("--hub_chunk", type=int, default=0, help="non-zeros per chunk of the hub plan's rows (0: library default)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_secondary", action="store_true", help="skip the extra run with the other activation dtype")
    ap.add_argument("--spmm_only", action="store_true", help="only the dominant-kernel loop (profiling helper)")
    ap.add_argument("--spmm_reps", type=int, default=2000, help="launches of the dominant kernel timed live for the roofline object (2000 x ~27 us: a 50 ms average)")
    ap.add_argument("--dp_reduce", default="rows", choices=["rows", "dense"],
                    help="data-parallel gradient exchange: all-gather of gradient rows (default) or dense all-reduce")
    ap.add_argument("--dp_shard", default="batch", choices=["batch", "rows", "cols"],
                    help="batch: replicated propagation, sharded batch; rows: row-sharded propagation too; cols: every rank holds "
                         "d / N columns of the tables and sees the whole batch (one all-reduce of 3*B floats per step; d / N must be 32..256)")
    ap.add_argument("--force_dp", action="store_true", help="use the data-parallel step (RCCL) even at world size 1")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong (default) = the reference's step, ONE global batch of B triplets sharded over the ranks "
                         "(SURVEY 8d C4; north_star: 'shards BPR batches'); weak = B triplets PER rank, global batch N*B.  Either "
                         "way `value` is global optimizer steps/s; triplets/s is reported beside it")
    ap.add_argument("--cpu_seconds", type=float, default=12.0)
    ap.add_argument("--no_steady", action="store_true", help="skip the 400-step steady-state region that follows the headline region")
    ap.add_argument("--no_epochs", action="store_true", help="skip the end-to-end epochs (sampler + shuffle + train) and the 10-epoch quality check")
    ap.add_argument("--no_eval", action="store_true", help="skip the evaluation-kernel measurement (eval_topk in the JSON line)")
    ap.add_argument("--data_dir", default=os.path.join(tempfile.gettempdir(), "lgcn_bench_data"))
    a = ap.parse_args()
    n_users, m_items, E, K, d, B, dsteps, dwarm = WORKLOADS[a.workload]
    d_full = d
    if a.steps is None:
        a.steps = dsteps
    if a.warmup is None:
        a.warmup = dwarm
    if a.row_order is None:
        a.row_order = "natural" if a.workload == "synthetic-10m" else "xcd"
    if a.act_dtype is None:
        a.act_dtype = "bf16" if a.workload == "gowalla" else "fp32"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a.gpus))          # before any GPU call in this process
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist
    dry = os.environ.get("LGCN_BENCH_DRYRUN") == "1"     # launch plumbing only (CPU test): gloo, no GPU
    if dry:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ones = torch.ones(1)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)          # what rccl_ranks_observed measures on the GPU path
        dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_rank_plus_1": float(t.item()),
                              "workload": a.workload, "steps": a.steps, "warmup": a.warmup, "scaling": a.scaling,
                              "act_dtype": a.act_dtype,
                              "global_batch": B * world if a.scaling == "weak" else B,
                              "rccl_ranks_observed": int(ones.item()),
                              "steady_state_steps": 0 if (a.workload == "synthetic-10m" or a.no_steady) else 400,
                              "extra_objects": (["roofline", "cpu_baseline", "eval_topk", "end_to_end_epoch", "quality"]
                                                if world == 1 and a.workload == "gowalla" else ["roofline"])}))
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # REHEARSAL HOOK (LGCN_BENCH_ONE_GPU=1): every rank on GPU 0, torch.distributed over gloo (it moves GPU tensors between the
    # processes through the host; RCCL refuses two ranks on one device) and the per-step loop of parallel.py -- the N-rank code
    # path of this script end to end on a one-GPU box.  Its numbers say nothing about scaling and are labelled as such.
    one_gpu = os.environ.get("LGCN_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        os.environ["LGCN_DP_PYTHON_LOOP"] = "1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dp = world > 1 or a.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    sys.argv = [sys.argv[0]]
    pkg = importlib.import_module(PKG)
    w = pkg.world
    w.configure(["--layer", str(K), "--recdim", str(d), "--bpr_batch", str(B), "--act_dtype", a.act_dtype,
                 "--xcd_remap", str(a.xcd_remap), "--row_order", a.row_order, "--tensorboard", "0",
                 "--dataset", a.workload, "--dense_last", a.dense_last, "--hub_nnz", str(a.hub_nnz)])
    w.config['hub_chunk'] = a.hub_chunk
    import io
    import contextlib
    t_setup = time.perf_counter()
    progress(f"workload {a.workload}: building the dataset")
    with contextlib.redirect_stdout(io.StringIO()):
        if a.workload == "gowalla" and os.path.exists(GOWALLA_NPZ):
            data_dir = materialize_gowalla(GOWALLA_NPZ, os.path.join(a.data_dir, f"gowalla_r{rank}"))
            ds = pkg.dataloader.Loader(w.config, path=data_dir)
            data_kind = "gowalla interaction graph (real, reconstructed from the reference's cache); random-init weights, seed 2020"
        else:
            ds = synthetic_dataset(pkg, a.workload, w.config, dev)
            data_kind = ("synthetic power-law bipartite graph of the configured shape (synthetic.py, seed 2020); "
                         "random-init weights, seed 2020")
        progress(f"interactions ready ({ds.trainDataSize} edges); normalised adjacency + model init")
        ds.getSparseGraphCSR()
        progress("adjacency built")
        pkg.sampling.seed(2020)
        pkg.utils.set_seed(2020)
        if use_dp and a.dp_shard == "cols":
            model = pkg.parallel.column_shard(pkg.model.LightGCN, w.config, ds, world, rank, dev)
            d_full, d = d, model.latent_dim        # this rank's width: what its kernels run at
        else:
            model = pkg.model.LightGCN(w.config, ds).to(dev)
    progress("model on the device")
    N = ds.n_users + ds.m_items
    nnz = int(ds.getSparseGraphCSR().nnz)
    ACT = {"fp32": (0, 4.0), "bf16": (1, 2.0), "fp8": (2, 1.0 + 4.0 / d)}     # code, bytes per stored element (fp8: + the row scale)
    adt, s = ACT[a.act_dtype]
    ADT_NAME = {0: "fp32", 1: "bf16", 2: "fp8"}
    L = pkg._lib
    lib_hash = pkg.build.kernel_hash()

    def spmm_kernel_time(reps, adt=adt):
        """dominant kernel (dense CSR-SpMM layer) timed live with HIP events on the launch stream"""
        st = model._state()
        x = torch.randn(N, d, device=dev) * 0.1               # same graph object (same row order) as the step
        if adt == 2:
            x = st['graph'].to_fp8(x)
        elif adt == 1:
            x = x.to(torch.bfloat16)
        y = torch.empty_like(x)
        lib = L.load()
        stream = L.current_stream()

        def spmm_once():
            L.check(lib.lgcn_spmm_csr(st['graph'].handle, L.tp(x), adt, L.tp(y), adt, d, stream), "spmm")
        for _ in range(max(2, reps // 10)):
            spmm_once()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            spmm_once()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    def roofline(t_spmm, adt=adt):
        s = ACT[ADT_NAME[adt]][1]
        bytes_spmm = spmm_bytes(N, nnz, d, s)
        achieved = bytes_spmm / t_spmm / 1e9
        traffic, note = read_traffic(f"{a.workload}:{ADT_NAME[adt]}:k_spmm", lib_hash)
        return {"bound": "hbm", "kernel": f"k_spmm<{d},{ {0: 'float', 1: 'bf16', 2: 'fp8'}[adt] }> (dense CSR-SpMM layer)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": note, "algorithmic_bytes_per_launch": bytes_spmm,
                "avg_launch_us": t_spmm * 1e6, "gather_bytes_upper_bound": nnz * d * s,
                # the row gathers themselves (nnz rows of d*s bytes, whatever level serves them) against the chip-wide rate the
                # guide measures for rows gathered out of the XCDs' L2 (MI355X_MICROARCH.md "Indexed rows": 66-73 GB/s per CU =
                # 16.8-18.8 TB/s; = 16 channels x 64 B/clk x 8 XCDs): where the table fits L2 + Infinity Cache this, not HBM, is
                # the roof the kernel sits under; tables beyond the caches (synthetic-10m) are bounded by HBM through `traffic`
                "gather": {"bound": "l2-gather", "achieved": nnz * d * s / t_spmm / 1e9, "peak": 17800.0, "unit": "GB/s",
                           "frac": nnz * d * s / t_spmm / 1e9 / 17800.0}}

    def stream_copy_gbs():
        """the box's measured stream-copy bandwidth (report it beside the 8 TB/s spec): a device-to-device copy of 1 GiB,
        bytes read + bytes written per second"""
        n = 1 << 28
        src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        for _ in range(3):
            dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        return 2.0 * 4 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    if a.spmm_only:
        reps = a.spmm_reps if a.workload != "synthetic-10m" else min(a.spmm_reps, 5)
        print(json.dumps({"spmm_only": True, "workload": a.workload, "act_dtype": a.act_dtype,
                          "row_order": a.row_order, "roofline": roofline(spmm_kernel_time(reps))}))
        return

    # triplets for warmup + timed steps, resident in HBM before the timed region
    Bg = B * world if a.scaling == "weak" else B     # weak: per-GPU batch fixed; strong: the global batch is the reference's B
    steady_steps = 0 if (a.workload == "synthetic-10m" or a.no_steady) else 400
    need = (a.warmup + a.steps + steady_steps) * Bg
    if ds.trainDataSize <= 4 * need:
        us, ps, ns = [], [], []
        have = 0
        while have < need:
            u, p, n = pkg.Procedure.sample_epoch_to_device(ds, dev)
            us.append(u); ps.append(p); ns.append(n); have += len(u)
        users, pos, neg = torch.cat(us)[:need], torch.cat(ps)[:need], torch.cat(ns)[:need]
    else:       # a full epoch of the big graph is 200 M triplets: draw only what the run consumes
        rng = np.random.Generator(np.random.PCG64(11))
        S = pkg.sampling.sample_negative_ByUser(rng.integers(0, ds.n_users, need).astype(np.int32), ds.m_items,
                                                ds.pos_csr(), 1)
        users, pos, neg = (torch.from_numpy(np.ascontiguousarray(S[:, c])).to(dev) for c in range(3))
    setup_s = time.perf_counter() - t_setup
    progress(f"{need} triplets resident; warm-up ({a.warmup} steps)")

    if not use_dp:
        def run(lo, steps):
            return model.fused_epoch(users[lo:lo + steps * B], pos[lo:lo + steps * B], neg[lo:lo + steps * B], B)
    else:
        dp = pkg.parallel.DataParallelBPR(model, w.config, reduce=a.dp_reduce, shard=a.dp_shard)

        def run(lo, steps):
            return dp.train_epoch(users[lo:lo + steps * Bg], pos[lo:lo + steps * Bg], neg[lo:lo + steps * Bg], Bg)

    def barrier():
        if use_dp:
            dist.barrier()
        torch.cuda.synchronize()

    # The measurements that are NOT the headline come first -- the dominant kernel's live timing (every rank) and the
    # same workload with the other activation storage type -- so that the headline's K steps run on a GPU that is
    # already at its working clocks (a 20-step region right after set-up measured 4-5 % under steady state).
    reps = a.spmm_reps if a.workload != "synthetic-10m" else 5
    t_spmm = spmm_kernel_time(reps)      # ~50 ms of back-to-back launches: also what brings the GPU to its working clocks for the legs below
    secondary = None
    if rank == 0 and not use_dp and not a.no_secondary and a.workload != "synthetic-10m":
        other = "bf16" if a.act_dtype != "bf16" else "fp32"
        cfg2 = dict(w.config); cfg2['act_dtype'] = other
        with contextlib.redirect_stdout(io.StringIO()):
            pkg.utils.set_seed(2020)
            model2 = pkg.model.LightGCN(cfg2, ds).to(dev)
        # (the same protocol as the headline region below: context built and warm before, W warm-up steps, K timed steps closed by a polled event)
        model2.fused_epoch(users[:a.warmup * B], pos[:a.warmup * B], neg[:a.warmup * B], B)
        spmm_kernel_time(min(200, reps), ACT[other][0])
        model2.fused_epoch(users[:a.warmup * B], pos[:a.warmup * B], neg[:a.warmup * B], B)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        model2.fused_epoch(users[a.warmup * B:(a.warmup + a.steps) * B], pos[a.warmup * B:(a.warmup + a.steps) * B],
                           neg[a.warmup * B:(a.warmup + a.steps) * B], B)
        done2 = torch.cuda.Event(); done2.record()
        while not done2.query():
            pass
        torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
        secondary = (other, a.steps / dt2, spmm_kernel_time(max(200, reps // 4), ACT[other][0]))
        del model2
    spmm_kernel_time(min(200, reps))     # (a short burst directly ahead of the warm-up steps)

    run(0, a.warmup)
    progress(f"timed region ({a.steps} steps)")
    barrier()
    t0 = time.perf_counter()
    losses = run(a.warmup * Bg, a.steps)
    done = torch.cuda.Event(); done.record()
    while not done.query():          # poll instead of sleeping in the driver: the wake-up of a blocking wait is ~1 % of 20 steps
        pass
    barrier()
    dt = time.perf_counter() - t0
    if use_dp:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    progress(f"timed region done: {dt:.3f} s")
    model.check_device_errors()
    # ---- the same loop for 400 more steps: the K-step region above is ~3 ms at the driver's K = 20 and reads ~2 % under this
    steady = None
    if steady_steps:
        barrier()
        t0s = time.perf_counter()
        run((a.warmup + a.steps) * Bg, steady_steps)
        barrier()
        dts = time.perf_counter() - t0s
        if use_dp:
            t = torch.tensor([dts], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dts = float(t.item())
        steady = steady_steps / dts
    # ---- how many ranks the data path's collectives really span (not WORLD_SIZE: an all-reduce of 1 per rank)
    ranks_observed = None
    if use_dp:
        ranks_observed = dp.ranks_observed()
    first_loss = float(losses.reshape(-1, 3)[0, 0])
    last_loss = float(losses.reshape(-1, 3)[-1, 0])

    out = None
    if rank == 0:
        steps_per_sec = a.steps / dt
        par = "single GPU"
        if use_dp:
            par = (f"dp{world} (replicated tables, batch-sharded, "
                   + ("gradient-row all-gather" if a.dp_reduce == "rows" else "dense gradient all-reduce")
                   + (" + row-sharded propagation with one all-gather per layer" if a.dp_shard == "rows" else "")
                   + " over RCCL)")
            if a.dp_shard == "cols":
                par = f"dp{world} (column-sharded tables: {d} of {d * world} columns per rank, whole batch on every rank, one all-reduce of 3*B floats per step over RCCL)"
        out = {
            # `value` = GLOBAL optimizer steps per second in both modes (never multiplied by the world size: with replicated
            # propagation every rank repeats the 2K-1 SpMMs, so this number is ~flat in N by construction of the algorithm;
            # what grows under weak scaling is config.triplets_per_sec)
            "metric": f"BPR training steps/sec (one step = K-layer LightGCN propagation + BPR loss + backward + Adam on a global batch of {Bg} triplets)",
            "value": steps_per_sec, "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f32" if a.act_dtype == "fp32" else f"{a.act_dtype} (activation storage; f32 accumulation, parameters, gradients and Adam)",
            "data": data_kind,
            "config": {"workload": f"{a.workload}: {ds.n_users} users x {ds.m_items} items, {ds.trainDataSize} train "
                                   f"interactions, nnz(A_hat)={nnz}, layers={K}, dim={d_full}, bpr_batch={B}",
                       "global_batch": Bg, "per_gpu_batch": (Bg + world - 1) // world, "global_steps_per_sec": steps_per_sec,
                       "triplets_per_sec": steps_per_sec * Bg, "batches_of_B_per_sec": steps_per_sec * Bg / B,
                       "scaling_mode": a.scaling, "multi_gpu_status": ("REHEARSAL: all ranks share GPU 0, gloo collectives, per-step Python loop -- not a scaling measurement" if one_gpu else
                                            "unmeasured on multi-GPU hardware by the builder (1-GPU boxes only)") if world > 1 else "n/a",
                       "parallelism": par,
                       "act_dtype": a.act_dtype, "xcd_remap": a.xcd_remap, "row_order": a.row_order,
                       "first_loss": first_loss, "last_loss": last_loss, "setup_seconds": setup_s},
            "steady_state_steps_per_sec": steady, "steady_state_steps": steady_steps,
            "rccl_ranks_observed": ranks_observed[0] if ranks_observed else (1 if world == 1 else None),
            "rccl_ranks_source": ranks_observed[1] if ranks_observed else "single GPU: no collective on the path",
            "step_algorithmic_bytes": step_bytes(N, nnz, d, s, K, B),
            "step_roofline_frac": step_bytes(N, nnz, d, s, K, B) * steps_per_sec / (HBM_PEAK_GBS * 1e9),
        }

    if rank == 0:
        if secondary is not None:       # the same workload with the other activation storage type (reported, not the headline)
            out["config"][f"{secondary[0]}_activation_storage_steps_per_sec"] = secondary[1]
            out[f"value_{secondary[0]}"] = secondary[1]        # same K-step region, same triplets, the other storage type
            # the dominant kernel of that other mode (BASELINE configs[1] names bf16 activation storage), same definition
            out[f"roofline_{secondary[0]}"] = roofline(secondary[2], ACT[secondary[0]][0])
        out["roofline"] = roofline(t_spmm)
        try:
            sc = stream_copy_gbs()
            out["roofline"]["stream_copy_measured"] = {"GB/s": sc, "what": "device-to-device copy of 1 GiB, read + write bytes per second (torch copy kernel)",
                                                       "frac_of_measured": out["roofline"]["achieved"] / sc}
            out["step_roofline_frac_of_measured_stream"] = out["step_algorithmic_bytes"] * out["value"] / (sc * 1e9)
        except Exception as e:      # noqa: BLE001
            out["roofline"]["stream_copy_measured"] = {"error": repr(e)}

    # ---- the evaluation kernel on the same model (reported, not the headline): Procedure.Test's scoring +
    #      masking + top-20 in one launch, priced against the fp32 matrix-core peak (the reference ranks in fp32)
    if rank == 0 and world == 1 and not a.no_eval and a.workload == "gowalla" and getattr(ds, "testDict", None):   # (the synthetic shapes hold 1 000 test users)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                model.eval()
                w.config['eval_fused'] = 1
                res = pkg.Procedure.Test(ds, model, 0)               # builds the evaluation index, warms the kernels
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(3):
                    res = pkg.Procedure.Test(ds, model, 0)
                torch.cuda.synchronize(); t_test = (time.perf_counter() - t0) / 3
                model.train()
            ev = ds._lgcn_eval_index
            E = model.propagated_table()
            topk = torch.empty(len(ev.users), 20, dtype=torch.int32, device=dev)
            L, lib = pkg._lib, pkg._lib.load()
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0_.record()
            for _ in range(5):       # the call Procedure.Test makes: train-positive masks precomputed once per dataset (ev.masks)
                L.check(lib.lgcn_eval_topk_masked(L.tp(E), model.n_users, model.m_items, model.latent_dim, L.tp(ev.users32), len(ev.users),
                                                  L.tp(ev.train_ptr), L.tp(ev.train_idx32), 20, L.tp(topk), None,
                                                  L.tp(ev.masks) if ev.masks is not None else None, L.current_stream()), "lgcn_eval_topk_masked")
            e1_.record(); torch.cuda.synchronize()
            t_ev = e0_.elapsed_time(e1_) / 5 * 1e-3
            e0_.record()
            for _ in range(5):       # the same sweep walking the train CSR with a cursor (no masks): what round 3 measured
                L.check(lib.lgcn_eval_topk(L.tp(E), model.n_users, model.m_items, model.latent_dim, L.tp(ev.users32), len(ev.users),
                                           L.tp(ev.train_ptr), L.tp(ev.train_idx32), 20, L.tp(topk), None, L.current_stream()), "lgcn_eval_topk")
            e1_.record(); torch.cuda.synchronize()
            t_ev_cursor = e0_.elapsed_time(e1_) / 5 * 1e-3
            flop = 2.0 * len(ev.users) * model.m_items * model.latent_dim
            # The kernel runs on the BF16 matrix cores: every fp32 product is six bf16 product planes (hh, hm, mh, hl, lh, mm),
            # so the work ISSUED is 6 x 2*U*M*d bf16 FLOP, priced against the dense bf16 MFMA peak (157.3 x 16 TFLOP/s,
            # MI355X_MICROARCH.md) -- that is `frac`, a fraction of a roof the kernel runs under.  The fp32-equivalent rate
            # (2*U*M*d per second, what the reference's sgemm would have to deliver) is reported beside it, with its ratio to
            # the fp32 MFMA peak as a COMPARISON only: it is not bounded by 1.
            planes = 6 if model.latent_dim <= 64 else 1
            peak = BF16_MFMA_PEAK_TF if planes == 6 else FP32_MFMA_PEAK_TF
            out["eval_topk"] = {"kernel": "k_eval_topk (scores + train mask + top-20 of every test user, one launch)", "ms": t_ev * 1e3,
                                "bound": "mfma", "achieved": planes * flop / t_ev / 1e12, "peak": peak,
                                "unit": "TFLOP/s of bf16 MFMA work issued (six product planes per fp32 product)" if planes == 6 else "TFLOP/s (fp32 MFMA)",
                                "frac": planes * flop / t_ev / 1e12 / peak,
                                "fp32_equivalent_tflops": flop / t_ev / 1e12, "fp32_equivalent_vs_fp32_mfma_peak": flop / t_ev / 1e12 / FP32_MFMA_PEAK_TF,
                                "users": len(ev.users), "items": model.m_items, "train_positive_masks": ev.masks is not None,
                                "ms_without_masks": t_ev_cursor * 1e3, "Procedure_Test_ms": t_test * 1e3}
        except Exception as e:      # noqa: BLE001 -- a secondary measurement never takes the headline down
            out["eval_topk"] = {"error": repr(e)}

    # ---- whole epochs as the reference runs them (main.py:215-225): sample (sampling.cpp stream, on the device) + shuffle
    #      (numpy's, on the host) + permutation apply + 394 steps, from seed 2020 -- timed end to end, and the model they
    #      produce after 10 epochs evaluated against the Recall@20 the imported reference reached on CPU with that seed
    #      (tests/golden/gowalla/golden_long.json: 0.1203589; north_star bar 1e-4).  SURVEY 8d: "sampler time folded into an
    #      end-to-end epoch steps/sec".
    if rank == 0 and world == 1 and not a.no_epochs and a.workload == "gowalla" and os.path.exists(GOLDEN_LONG):
        try:
            gold = json.load(open(GOLDEN_LONG))["trajectory"][9]["test"]
            steps_per_epoch = (ds.n_users * (ds.trainDataSize // ds.n_users) + B - 1) // B

            def epochs(act, prefetch, n_epochs=10):
                cfg = dict(w.config); cfg['act_dtype'] = act
                old = w.config.get('prefetch_epoch', 1)
                w.config['prefetch_epoch'] = prefetch
                ds._lgcn_next_epoch = None
                try:
                    with contextlib.redirect_stdout(io.StringIO()):
                        pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
                        mq = pkg.model.LightGCN(cfg, ds).to(dev)
                        bq = pkg.utils.BPRLoss(mq, cfg)
                        secs = []
                        for e in range(1, n_epochs + 1):
                            torch.cuda.synchronize(); t0 = time.perf_counter()
                            pkg.Procedure.BPR_train_original(ds, mq, bq, e)
                            torch.cuda.synchronize(); secs.append(time.perf_counter() - t0)
                        mq.eval()
                        res = pkg.Procedure.Test(ds, mq, n_epochs)
                finally:
                    w.config['prefetch_epoch'] = old
                    ds._lgcn_next_epoch = None
                del mq, bq
                return secs, {k: float(v[0]) for k, v in res.items()}
            e2e, quality = {}, {}
            for act, prefetch in (("fp32", 1), ("fp32", 0), ("bf16", 1)):
                secs, res = epochs(act, prefetch)
                rest = secs[1:]                                   # the first epoch builds the context (and samples in line)
                if act == "bf16":
                    e2e["bf16_prefetch_on"] = {"steps_per_sec": steps_per_epoch * len(rest) / sum(rest), "ms_per_epoch": 1e3 * sum(rest) / len(rest)}
                if act == "fp32":
                    e2e["prefetch_on" if prefetch else "prefetch_off"] = {
                        "steps_per_sec": steps_per_epoch * len(rest) / sum(rest), "ms_per_epoch": 1e3 * sum(rest) / len(rest),
                        "first_epoch_ms": 1e3 * secs[0]}
                if prefetch:
                    quality[act] = {"recall@20": res["recall"], "ndcg@20": res["ndcg"], "precision@20": res["precision"],
                                    "abs_diff_recall": abs(res["recall"] - gold["recall"][0]),
                                    "abs_diff_ndcg": abs(res["ndcg"] - gold["ndcg"][0]),
                                    "within_1e-4": bool(abs(res["recall"] - gold["recall"][0]) <= 1e-4 and abs(res["ndcg"] - gold["ndcg"][0]) <= 1e-4)}
            e2e.update({"steps_per_epoch": steps_per_epoch, "epochs_timed": 9,
                        "what": "(prefetch_on / prefetch_off: fp32 tables) Procedure.BPR_train_original: device sampler (bit-exact sampling.cpp stream) + the numpy-legacy shuffle "
                                "(MT19937 + Fisher-Yates, same permutation; on the device since round 4) + permutation apply + the fused "
                                "steps; prefetch_on samples epoch e+1 on a side stream under epoch e (same triplets), prefetch_off is the "
                                "reference's strict order"})
            out["end_to_end_epoch"] = e2e
            out["quality"] = {"epochs": 10, "seed": 2020,
                              "reference": {"recall@20": gold["recall"][0], "ndcg@20": gold["ndcg"][0], "precision@20": gold["precision"][0],
                                            "source": "tests/golden/gowalla/golden_long.json (the reference imported and run on CPU, make_golden.py)"},
                              **quality}
        except Exception as e:      # noqa: BLE001
            out["end_to_end_epoch"] = {"error": repr(e)}

    # ---- CPU baseline on the host cores: the oracle (C/OpenMP port of the reference path) and an
    #      op-for-op torch-CPU eager restatement of the reference's stageOne, same workload
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.workload != "synthetic-10m":
        from oracle import oracle as orc
        from oracle import torch_eager
        adj = ds.getSparseGraphCSR()
        e0 = model._table.detach().cpu().numpy().copy()
        hu, hp, hn = (t[:64 * B].cpu().numpy().astype(np.int64) for t in (users, pos, neg))
        tr = orc.Trainer(ds.n_users, adj.indptr, adj.indices, adj.data, e0, K, w.config['decay'], w.config['lr'])

        def timed(stepper, budget, cap=63):
            stepper(hu[:B], hp[:B], hn[:B])                  # warm-up step
            t0 = time.perf_counter(); n = 0
            while n < cap and (time.perf_counter() - t0) < budget:
                lo = ((n + 1) % 63) * B
                stepper(hu[lo:lo + B], hp[lo:lo + B], hn[lo:lo + B]); n += 1
            return n, time.perf_counter() - t0
        n_cpu, dt_cpu = timed(tr.stageOne, a.cpu_seconds)
        phys = os.cpu_count()
        try:
            import psutil
            phys = psutil.cpu_count(logical=False) or phys
        except Exception:
            pass
        te = torch_eager.EagerTrainer(ds.n_users, adj.indptr, adj.indices, adj.data, e0, K, w.config['decay'],
                                      w.config['lr'], threads=phys)
        n_te, dt_te = timed(te.stageOne, a.cpu_seconds, cap=30)
        out["cpu_baseline"] = {
            "value": n_cpu / dt_cpu, "unit": "steps/s", "cores": orc.num_threads(), "kind": "port",
            "physical_cores": phys, "logical_cpus": os.cpu_count(),
            "sample": f"{n_cpu} consecutive stageOne steps of the same workload (same graph, same "
                      f"triplets, B={B}) through oracle/lgcn_oracle.c (C + OpenMP, fp32), {dt_cpu:.1f} s",
            "torch_eager": {"value": n_te / dt_te, "unit": "steps/s", "cores": torch.get_num_threads(),
                            "sample": f"{n_te} stageOne steps, op-for-op torch-CPU restatement of the reference's "
                                      f"model.py:201-231 + utils.py:53-64 (oracle/torch_eager.py), {dt_te:.1f} s; the "
                                      "reference itself measured 1.45-2.22 steps/s on 8 cores (BASELINE.md)"}}
    if rank == 0:
        sys.stdout.flush()
        print("\n" + json.dumps(out), flush=True)        # on a line of its own whatever a library wrote to stdout before
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
