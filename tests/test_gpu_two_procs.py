"""Two REAL processes on one GPU (-m gpu): torch.distributed over gloo carries the collectives (it moves GPU tensors between
processes through the host; RCCL refuses two ranks on one device), and parallel.DataParallelBPR runs its per-step loop
(LGCN_DP_PYTHON_LOOP=1) -- the code path every rank falls back to when the library's own RCCL communicator is not available -- in
the gradient-row, dense all-reduce and column-sharded modes.  What the in-process loopback tests (W threads, the C loop) do not
cover: separate address spaces, torch.distributed's collectives on the library's buffers (the column-sharded step all-reduces a
library-owned device array wrapped through __cuda_array_interface__), parallel.column_shard, gather_table and the evaluation of a
column-sharded model (the ranks' propagated columns all-gathered).  Row-sharded propagation has no per-step loop and is not here."""
import importlib
import os
import shutil
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, PKG_NAME, REPO

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, work):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["LGCN_DP_PYTHON_LOOP"] = "1"
    sys.argv = [sys.argv[0]]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    pkg = importlib.import_module(PKG_NAME)
    w = pkg.world
    d = os.path.join(work, f"tiny_r{rank}")
    os.makedirs(d, exist_ok=True)
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(GOLDEN, "tiny", f), os.path.join(d, f))
    B, dim = 48, 64

    def configure():
        w.configure(["--dataset", "tiny", "--tensorboard", "0", "--recdim", str(dim), "--bpr_batch", str(B), "--row_order", "natural",
                     "--checkpoint_dir", os.path.join(work, f"ckpt{rank}")])

    def fresh(cfg=None):
        pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
        return pkg.model.LightGCN(cfg or w.config, ds).to(dev)

    configure()
    ds = pkg.dataloader.Loader(w.config, path=d)
    rng = np.random.Generator(np.random.PCG64(5))
    T = 3 * B + 7
    U, P, N = (torch.from_numpy(rng.integers(0, hi, T).astype(np.int32)).to(dev) for hi in (ds.n_users, ds.m_items, ds.m_items))
    ref = fresh()
    E0 = ref._table.detach().clone()
    want_loss = ref.fused_epoch(U, P, N, B).cpu().numpy()
    want = ref._table.detach().cpu().numpy()
    ok = True
    for mode in ("rows", "dense"):
        m = fresh()
        assert torch.equal(m._table.detach(), E0)
        dp = pkg.parallel.DataParallelBPR(m, w.config, reduce=mode)
        got_loss = dp.train_epoch(U, P, N, B).cpu().numpy()
        ok = ok and np.array_equal(got_loss, want_loss) and np.array_equal(m._table.detach().cpu().numpy().view(np.uint32), want.view(np.uint32))
        m.check_device_errors()
    # column-sharded: this rank holds 32 of the 64 columns
    pkg.utils.set_seed(2020)
    mc = pkg.parallel.column_shard(pkg.model.LightGCN, w.config, ds, world, rank, dev)
    lo, hi = pkg.parallel.column_range(dim, world, rank)
    ok = ok and mc.latent_dim == dim // world and torch.equal(mc._table.detach(), E0[:, lo:hi])
    dpc = pkg.parallel.DataParallelBPR(mc, w.config, shard="cols")
    got_loss = dpc.train_epoch(U, P, N, B).cpu().numpy()
    full = dpc.gather_table().cpu().numpy()
    ok = ok and np.abs(got_loss - want_loss).max() < 2e-6 and full.shape == want.shape and np.abs(full - want).max() < 2e-6
    # evaluation of the column shard: the propagated columns of both ranks side by side = the propagated table of the full model
    mc.eval(); ref.eval()
    with torch.no_grad():
        pt = mc.propagated_table().cpu().numpy()
        pr = ref.propagated_table().cpu().numpy()
    ok = ok and pt.shape == pr.shape and np.abs(pt - pr).max() < 1e-5
    r = pkg.Procedure.Test(ds, mc, 0)
    box = [None] * world
    dist.all_gather_object(box, {k: float(v[0]) for k, v in r.items()})
    ok = ok and box[0] == box[1]
    mc.check_device_errors()
    # the fork's optional branches (popularity gate + item-item smoothing) through the per-step loop, both batch-sharded forms:
    # the dense one all-reduces G64, three blocks of terms and the library-owned int64 array of MLP gradient sums
    import json
    meta = json.load(open(os.path.join(GOLDEN, "tiny", "golden_gate_i2i.json")))
    w.configure([])
    w.dataset = "tiny"
    w.config.update({'lightGCN_n_layers': meta["K"], 'latent_dim_rec': meta["d"], 'bpr_batch_size': B, 'decay': meta["decay"], 'lr': meta["lr"],
                     'use_pop_gate': meta["use_pop_gate"], 'use_item_item': meta["use_item_item"],
                     'i2i_path': os.path.join(GOLDEN, "tiny", "i2i_tiny.npz"), 'i2i_alpha': meta["i2i_alpha"], 'fused_variants': 1,
                     'checkpoint_dir': os.path.join(work, f"ckptv{rank}")})
    dsv = pkg.dataloader.Loader(w.config, path=d)

    def fresh_v():
        pkg.utils.set_seed(meta["seed"])
        mv = pkg.model.LightGCN(w.config, dsv).to(dev)
        mv.train()
        return mv
    refv = fresh_v()
    want_loss_v = refv.fused_epoch(U, P, N, B).cpu().numpy()
    want_v = {k: v.detach().cpu().numpy().copy() for k, v in refv.state_dict().items()}
    for mode in ("rows", "dense"):
        mv = fresh_v()
        dpv = pkg.parallel.DataParallelBPR(mv, w.config, reduce=mode)
        got_loss = dpv.train_epoch(U, P, N, B).cpu().numpy()
        ok = ok and np.array_equal(got_loss, want_loss_v)
        for k, v in mv.state_dict().items():
            ok = ok and np.array_equal(v.detach().cpu().numpy().view(np.uint32), want_v[k].view(np.uint32))
        mv.check_device_errors()
    ok = ok and len(want_v) > 2          # (the gate's MLP parameters are in the state_dict)
    t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(os.path.join(work, "ok.txt"), "w").write(str(int(t.item())))
    dist.destroy_process_group()


def test_two_processes_one_gpu_python_loop(tmp_path):
    work = str(tmp_path)
    mp.spawn(_worker, args=(2, _free_port(), work), nprocs=2, join=True)
    assert open(os.path.join(work, "ok.txt")).read() == "1"


def _refusal_worker(rank, world, port, work):
    """No LGCN_DP_PYTHON_LOOP here: DataParallelBPR really tries the library's own RCCL communicator -- and RCCL really refuses
    (two ranks on one device: ncclCommInitRank returns "invalid usage" on both)."""
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("LGCN_DP_PYTHON_LOOP", None)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.argv = [sys.argv[0]]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    pkg = importlib.import_module(PKG_NAME)
    w = pkg.world
    d = os.path.join(work, f"tiny_r{rank}")
    os.makedirs(d, exist_ok=True)
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(GOLDEN, "tiny", f), os.path.join(d, f))
    B = 48
    w.configure(["--dataset", "tiny", "--tensorboard", "0", "--bpr_batch", str(B), "--row_order", "natural",
                 "--checkpoint_dir", os.path.join(work, f"ckpt{rank}")])
    ds = pkg.dataloader.Loader(w.config, path=d)
    rng = np.random.Generator(np.random.PCG64(6))
    T = 2 * B + 5
    U, P, N = (torch.from_numpy(rng.integers(0, hi, T).astype(np.int32)).to(dev) for hi in (ds.n_users, ds.m_items, ds.m_items))
    pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
    ref = pkg.model.LightGCN(w.config, ds).to(dev)
    want_loss = ref.fused_epoch(U, P, N, B).cpu().numpy()
    want = ref._table.detach().cpu().numpy()
    pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
    m = pkg.model.LightGCN(w.config, ds).to(dev)
    dp = pkg.parallel.DataParallelBPR(m, w.config)
    own = dp._own_communicator_ok()                       # collective decision: must come back False on BOTH ranks, and come back
    got_loss = dp.train_epoch(U, P, N, B).cpu().numpy()   # ... and the epoch runs over torch.distributed's collectives instead
    seen = dp.ranks_observed()
    ok = (own is False and seen == (2, "torch.distributed") and np.array_equal(got_loss, want_loss)
          and np.array_equal(m._table.detach().cpu().numpy().view(np.uint32), want.view(np.uint32)))
    t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(os.path.join(work, "ok_refusal.txt"), "w").write(str(int(t.item())))
    dist.destroy_process_group()


def test_real_rccl_refusal_falls_back_on_every_rank(tmp_path):
    """The fallback agreement of DataParallelBPR against a REAL failure of the library's communicator: two processes on one GPU, RCCL's
    ncclCommInitRank refuses (invalid usage) on both ranks, every rank drops its communicator, nobody hangs, and the epoch runs over
    torch.distributed -- bit for bit the single-GPU epoch (the CPU tests only simulate the failure)."""
    work = str(tmp_path)
    mp.spawn(_refusal_worker, args=(2, _free_port(), work), nprocs=2, join=True)
    assert open(os.path.join(work, "ok_refusal.txt")).read() == "1"

