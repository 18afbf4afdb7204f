"""N>1 path on CPU: two gloo ranks exercise the data-parallel exchange of parallel.py (shard
bounds, per-rank block layout [3*S*d | S | S], the all-gather) with gradient rows computed by the
oracle, and check that the gathered blocks reduce to the oracle's single-process gradient G and
loss for the global batch.  The HIP kernels that produce / consume these blocks are covered on the
GPU by test_gpu_parity.py::test_bitwise_reproducible_and_dp_split."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO, PKG_NAME, GoldenSet


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rows_and_terms(orc, E, g, users, pos, neg, B_global, decay):
    """Per-triplet gradient rows (slot-major [3,b,d]) and loss terms, by finite assembly from the oracle:
    the oracle's bpr() on ONE triplet with inv_B and lambda of the GLOBAL batch."""
    d = E.shape[1]
    rows = np.zeros((3, len(users), d), np.float32)
    lt = np.zeros(len(users), np.float32); rt = np.zeros(len(users), np.float32)
    for i, (u, p, n) in enumerate(zip(users, pos, neg)):
        eu, ep, en = E[u], E[g.n_users + p], E[g.n_users + n]
        x = np.float32(eu @ ep) - np.float32(eu @ en)
        z = np.exp(-abs(x)); sig_neg = (1 / (1 + z)) if x < 0 else z / (1 + z)
        gb = np.float32(-(1.0 / B_global) * sig_neg); lam = np.float32(decay / B_global)
        rows[0, i] = gb * (ep - en) + lam * eu
        rows[1, i] = gb * eu + lam * ep
        rows[2, i] = -gb * eu + lam * en
        lt[i] = min(x, 0) - np.log1p(z)
        rt[i] = eu @ eu + ep @ ep + en @ en
    return rows, lt, rt


def _worker(rank, world, port, out):
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG_NAME)
    from oracle import oracle as orc
    g = GoldenSet("tiny")
    E = orc.propagate(g.z["adj_indptr"], g.z["adj_indices"], g.z["adj_data"], g.e0(), g.K)
    rng = np.random.Generator(np.random.PCG64(0))
    B = 37                                                   # ragged: shards of 19 and 18
    users = rng.integers(0, g.n_users, B); pos = rng.integers(0, g.m_items, B); neg = rng.integers(0, g.m_items, B)
    par = pkg.parallel
    S = par.shard_size(B, world)
    lo, hi = par.shard_bounds(B, world, rank)
    rows, lt, rt = _rows_and_terms(orc, E, g, users[lo:hi], pos[lo:hi], neg[lo:hi], B, g.meta["decay"])
    block = np.zeros(par.block_numel(B, world, g.d), np.float32)
    blk_rows = block[:3 * S * g.d].reshape(3, S, g.d)
    blk_rows[:, :hi - lo] = rows
    block[3 * S * g.d:3 * S * g.d + (hi - lo)] = lt
    block[3 * S * g.d + S:3 * S * g.d + S + (hi - lo)] = rt
    gathered = par.exchange(torch.from_numpy(block)).numpy()
    # every rank reduces all blocks exactly as k_scatter / k_finish index them
    blk = par.block_numel(B, world, g.d)
    G = np.zeros_like(E, dtype=np.float64); fl = fr = 0.0
    for b in range(B):
        r, i = b // S, b % S
        base = r * blk
        for c, row in ((0, users[b]), (1, g.n_users + pos[b]), (2, g.n_users + neg[b])):
            G[row] += gathered[base + (c * S + i) * g.d: base + (c * S + i + 1) * g.d]
        fl += gathered[base + 3 * S * g.d + i]; fr += gathered[base + 3 * S * g.d + S + i]
    bpr, reg, G_ref = orc.bpr(E, g.n_users, users, pos, neg, g.meta["decay"])
    ok = (np.allclose(G, G_ref, rtol=1e-5, atol=1e-9) and abs(-fl / B - bpr) < 1e-6 and abs(0.5 * fr / B - reg) < 1e-6
          and gathered.shape[0] == world * blk)
    t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out, "w").write(str(int(t.item())))
    dist.destroy_process_group()


def test_dp_exchange_two_gloo_ranks(tmp_path):
    out = os.path.join(str(tmp_path), "ok.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "1"


def test_shard_bounds(pkg):
    par = pkg.parallel
    for B in (1, 7, 37, 2048, 2049):
        for world in (1, 2, 3, 8):
            S = par.shard_size(B, world)
            cover = []
            for r in range(world):
                lo, hi = par.shard_bounds(B, world, r)
                assert 0 <= lo <= hi <= B and hi - lo <= S
                cover.extend(range(lo, hi))
            assert cover == list(range(B))
            assert par.block_numel(B, world, 64) == 3 * S * 64 + 2 * S


def _rs_worker(rank, world, port, out):
    """Row-sharded propagation over two gloo ranks: each rank computes ONLY the rows it owns of every
    layer (oracle SpMM restricted to its row ranges), the owners' ranges are exchanged with
    parallel.exchange_rows (the torch.distributed form of the library's grouped RCCL broadcasts), and
    the K-layer mean must equal the oracle's single-process propagate() bit for bit on every rank."""
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG_NAME)
    from oracle import oracle as orc
    g = GoldenSet("lastfm")
    ip, ix, vv = g.z["adj_indptr"], g.z["adj_indices"], g.z["adj_data"]
    par = pkg.parallel
    ranges = par.row_ranges(ip, g.n_users, world)
    mine = par.owned_rows(ranges, rank)
    N = len(ip) - 1
    sub_ip = np.concatenate([[0], np.cumsum(np.diff(ip)[mine])]).astype(np.int32)
    sel = np.concatenate([np.arange(ip[r], ip[r + 1]) for r in mine]) if len(mine) else np.zeros(0, np.int64)
    x = g.e0().copy()
    acc = x.astype(np.float32).copy()
    for _ in range(g.K):
        y = torch.full((N, g.d), float("nan"))                  # rows of other owners: unknown until exchanged
        y[torch.from_numpy(mine).long()] = torch.from_numpy(orc.spmm(sub_ip, ix[sel], vv[sel], x))
        par.exchange_rows(y, ranges)
        x = y.numpy()
        acc = acc + x
    out_mean = acc / np.float32(g.K + 1)
    ref = orc.propagate(ip, ix, vv, g.e0(), g.K)
    ok = bool(np.array_equal(out_mean.view(np.uint32), ref.view(np.uint32)))
    t = torch.tensor([1 if ok else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out, "w").write(str(int(t.item())))
    dist.destroy_process_group()


def test_row_sharded_exchange_two_gloo_ranks(tmp_path):
    out = os.path.join(str(tmp_path), "ok_rs.txt")
    mp.spawn(_rs_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "1"


def test_row_ranges_partition_and_balance(pkg, lastfm):
    ip = lastfm.z["adj_indptr"].astype(np.int64)
    N = len(ip) - 1
    for world in (1, 2, 3, 8):
        rr = pkg.parallel.row_ranges(ip, lastfm.n_users, world)
        assert rr.shape == (world, 4) and rr[0, 0] == 0 and rr[-1, 1] == lastfm.n_users and rr[0, 2] == lastfm.n_users and rr[-1, 3] == N
        assert np.array_equal(rr[1:, 0], rr[:-1, 1]) and np.array_equal(rr[1:, 2], rr[:-1, 3])
        own = np.concatenate([pkg.parallel.owned_rows(rr, r) for r in range(world)])
        assert np.array_equal(np.sort(own), np.arange(N))
        nnz = np.array([(ip[rr[r, 1]] - ip[rr[r, 0]]) + (ip[rr[r, 3]] - ip[rr[r, 2]]) for r in range(world)], np.float64)
        assert nnz.max() <= 1.3 * nnz.mean() + 2 * np.diff(ip).max()


def _comm_fail_worker(rank, world, port, out, case):
    """The library's communicator cannot be had (rank 0 used to raise before the broadcast the other
    ranks were already waiting in -> mismatched collectives).  Every rank must come out of _own_communicator_ok()
    with False, having run the same collective sequence."""
    import shutil, tempfile
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    pkg = importlib.import_module(PKG_NAME)
    g = GoldenSet("tiny")
    work = tempfile.mkdtemp(prefix=f"lgcn_dpfail{rank}")
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(g.dir, f), os.path.join(work, f))
    w = pkg.world
    w.configure(["--dataset", "tiny", "--tensorboard", "0", "--layer", str(g.K), "--recdim", str(g.d), "--bpr_batch", str(g.B)])
    ds = pkg.dataloader.Loader(w.config, path=work)
    model = pkg.model.LightGCN(w.config, ds)                      # CPU tensors: only the agreement protocol runs
    lib = pkg._lib.load()
    calls = {"id": 0, "init": 0}
    if case == "unique_id" and rank == 0:
        def bad_id(buf):
            calls["id"] += 1
            return 12
        lib.lgcn_dp_unique_id = bad_id
        lib.lgcn_dp_available = lambda: 1
    elif case == "unique_id":
        lib.lgcn_dp_available = lambda: 1
    if case == "unavailable":
        lib.lgcn_dp_available = (lambda: 0) if rank == 1 else (lambda: 1)
    def no_init(*a):
        calls["init"] += 1
        return 12
    lib.lgcn_dp_init = no_init                                    # must never be entered: it is a collective
    dp = pkg.parallel.DataParallelBPR(model, w.config, reduce='rows', shard='batch')
    ok = dp._own_communicator_ok()
    good = (ok is False) and calls["init"] == 0 and dp._own_communicator_ok() is False
    if case == "unique_id" and rank == 0:
        good = good and calls["id"] == 1
    # shard='rows' without the communicator: a clear error, not a recursion between _step and train_epoch
    dp2 = pkg.parallel.DataParallelBPR(model, w.config, reduce='rows', shard='rows')
    u = torch.zeros(4, dtype=torch.int64)
    try:
        dp2.stageOne(u, u, u)
        good = False
    except RecursionError:
        good = False
    except RuntimeError as e:
        good = good and "row-sharded" in str(e)
    t = torch.tensor([1 if good else 0]); dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out, "w").write(str(int(t.item())))
    shutil.rmtree(work, ignore_errors=True)
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["unique_id", "unavailable"])
def test_communicator_failure_falls_back_on_every_rank(tmp_path, case):
    out = os.path.join(str(tmp_path), "ok.txt")
    mp.spawn(_comm_fail_worker, args=(2, _free_port(), out, case), nprocs=2, join=True)
    assert open(out).read() == "1"


def _cols_worker(rank, world, port, out):
    """Column-sharded step (parallel.py shard='cols') over gloo, on the oracle's numbers: every rank propagates ITS columns of the
    table (propagation is per column), forms the partial scores / reg terms of the whole batch over those columns, all-reduces
    the 3*B floats, and builds the gradient rows of its columns from the COMPLETE scores -- the concatenation over the ranks
    must be the oracle's single-process G and loss."""
    sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module(PKG_NAME)
    from oracle import oracle as orc
    g = GoldenSet("tiny")
    A = (g.z["adj_indptr"], g.z["adj_indices"], g.z["adj_data"])
    lo, hi = pkg.parallel.column_range(g.d, world, rank)
    E_loc = orc.propagate(*A, np.ascontiguousarray(g.e0()[:, lo:hi]), g.K)              # this rank's columns only
    E = orc.propagate(*A, g.e0(), g.K)
    rng = np.random.Generator(np.random.PCG64(3))
    B = 41
    users = rng.integers(0, g.n_users, B); pos = rng.integers(0, g.m_items, B); neg = rng.integers(0, g.m_items, B)
    users[:3] = users[0]; pos[5] = neg[6]
    eu, ep, en = E_loc[users], E_loc[g.n_users + pos], E_loc[g.n_users + neg]
    part = np.concatenate([(eu * ep).sum(1), (eu * en).sum(1), (eu * eu + ep * ep + en * en).sum(1)]).astype(np.float32)
    t = torch.from_numpy(part.copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    ps, ns, rr = t.numpy()[:B], t.numpy()[B:2 * B], t.numpy()[2 * B:]
    x = ps - ns
    z = np.exp(-np.abs(x)); sig_neg = np.where(x < 0, 1 / (1 + z), z / (1 + z))
    gb = (-(1.0 / B) * sig_neg).astype(np.float32)[:, None]; lam = np.float32(g.meta["decay"] / B)
    G_loc = np.zeros_like(E_loc, dtype=np.float64)
    np.add.at(G_loc, users, gb * (ep - en) + lam * eu)
    np.add.at(G_loc, g.n_users + pos, gb * eu + lam * ep)
    np.add.at(G_loc, g.n_users + neg, -gb * eu + lam * en)
    bpr, reg, G_ref = orc.bpr(E, g.n_users, users, pos, neg, g.meta["decay"])
    loss = -(np.minimum(x, 0) - np.log1p(z)).sum() / B
    ok = (np.allclose(E_loc, E[:, lo:hi], rtol=0, atol=0)                                # propagation IS per column (bit for bit)
          and np.allclose(G_loc, G_ref[:, lo:hi], rtol=2e-5, atol=1e-9) and abs(loss - bpr) < 1e-6 and abs(0.5 * rr.sum() / B - reg) < 1e-6)
    tt = torch.tensor([1 if ok else 0]); dist.all_reduce(tt, op=dist.ReduceOp.MIN)
    if rank == 0:
        open(out, "w").write(str(int(tt.item())))
    dist.destroy_process_group()


def test_column_sharded_step_two_gloo_ranks(tmp_path):
    out = os.path.join(str(tmp_path), "ok_cols.txt")
    mp.spawn(_cols_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "1"


def test_column_range(pkg):
    par = pkg.parallel
    assert [par.column_range(64, 2, r) for r in range(2)] == [(0, 32), (32, 64)]
    assert [par.column_range(256, 8, r) for r in (0, 7)] == [(0, 32), (224, 256)]
    for d, w in ((64, 4), (64, 3), (128, 8)):
        with pytest.raises(ValueError):
            par.column_range(d, w, 0)
