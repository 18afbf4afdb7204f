"""Batch-sharded data parallelism for BPRLoss.stageOne (one process per GPU,
torch.distributed backend "nccl" = RCCL over xGMI).  The reference has no
distributed code at all; the contract is BASELINE.json's north_star:
replicated embedding tables, sharded BPR batches, gradient reduction, identical
Adam step on every replica.

What is exchanged.  The gradient of the loss w.r.t. the PROPAGATED table is a
sum of <= 3B rows, and backward propagation is linear, so the ranks
exchange those rows BEFORE backward propagation instead of all-reducing a dense
[N,d] gradient after it:
    rank r   : forward propagation (replicated) + gradient rows and loss terms of
               its shard  [3*S*d | S | S] floats, S = ceil(B/world)
    all ranks: ONE all-gather of that block (B=2048,d=64: 1.6 MB total vs 18 MB
               dense for Gowalla; d=128 Amazon-Book: 3.2 MB vs 73.9 MB)
    all ranks: order-independent fixed-point reduction of all 3B rows, backward
               propagation, Adam -- bitwise identical on every rank and to the
               single-GPU step, so replicas never drift and need no parameter sync.
`shard='cols'` shards the COLUMNS of the tables instead of the batch (column_shard below): propagation, gradient
scatter, backward and Adam are independent per column, so a rank's SpMM work falls with the world size -- the one mode
in which it does at Gowalla / Amazon-Book size -- and the step needs ONE all-reduce of 3*B floats (the triplets' partial
scores and reg terms).  Not bitwise equal to one GPU: the dot products are summed as W partial sums.
`reduce='dense'` is the literal north_star wording -- RCCL all-reduce (SUM) of the
gradient table, here the fixed-point G64 [N,d] int64, plus SUM of the loss terms (every
rank flags the rows of the whole batch itself) -- kept as an alternative: also bitwise exact, but N*d*8 bytes
per step (Gowalla 36 MB) instead of 1.6 MB.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def shard_size(B, world):
    return (B + world - 1) // world


def shard_bounds(B, world, rank):
    """[lo, hi) of the global batch handled by `rank` (contiguous slices)."""
    S = shard_size(B, world)
    lo = min(rank * S, B)
    return lo, min(lo + S, B)


def block_numel(B, world, d):
    S = shard_size(B, world)
    return 3 * S * d + 2 * S


def row_ranges(adj_indptr, n_users, world):
    """Row-sharded propagation: rank r owns a contiguous range of user rows and a contiguous range of
    item rows of the [N,d] tables, each range holding ~1/world of that block's non-zeros (the two
    blocks of A_hat hold E non-zeros each, but a user row and an item row are differently long: one
    range per block balances both).  -> int64 [world, 4] = (u_lo, u_hi, i_lo, i_hi), row ids."""
    ip = np.asarray(adj_indptr, np.int64)
    N = len(ip) - 1
    out = np.zeros((world, 4), np.int64)
    for col, lo, hi in ((0, 0, n_users), (2, n_users, N)):
        base, total = ip[lo], ip[hi] - ip[lo]
        cuts = np.searchsorted(ip[lo:hi + 1] - base, np.arange(world + 1) * (total / world), side='left') + lo
        cuts[0], cuts[-1] = lo, hi
        cuts = np.maximum.accumulate(cuts)
        out[:, col], out[:, col + 1] = cuts[:-1], cuts[1:]
    return out


def owned_rows(ranges, rank):
    u_lo, u_hi, i_lo, i_hi = (int(v) for v in ranges[rank])
    return np.concatenate([np.arange(u_lo, u_hi, dtype=np.int32), np.arange(i_lo, i_hi, dtype=np.int32)])


def column_range(d, world, rank):
    """[lo, hi) of the embedding columns rank `rank` holds under shard='cols' (d / world each; the kernels take 32/64/128/256)."""
    if d % world or (d // world) not in (32, 64, 128, 256):
        raise ValueError(f"column sharding needs d / world in (32, 64, 128, 256); got d={d}, world={world}")
    w = d // world
    return rank * w, (rank + 1) * w


def column_shard(model_cls, config, dataset, world, rank, device):
    """This rank's column slice of the model the single-GPU run would create: the FULL table is initialised exactly as
    LightGCN.__init__ does (same seed, same draw: model.py:57-63), then columns [lo, hi) become a LightGCN of width d / world.
    -> the sharded model (its config is a copy with latent_dim_rec = d / world)."""
    d = int(config['latent_dim_rec'])
    lo, hi = column_range(d, world, rank)
    full = model_cls(config, dataset).to(device)
    cfg = dict(config)
    cfg['latent_dim_rec'] = hi - lo
    part = model_cls(cfg, dataset).to(device)
    with torch.no_grad():
        part._table.copy_(full._table[:, lo:hi])
    full._drop_device_state()
    del full
    part.col_range = (lo, hi, d)
    return part


class _DevArray:
    """a device buffer the library owns (float32, or int64 with typestr '<i8'), as something torch.as_tensor can wrap (no copy)"""

    def __init__(self, ptr, n, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def exchange_rows(buf, ranges, group=None):
    """The exchange after a row-sharded layer: every owner broadcasts its two row ranges of `buf`
    [N,d] in place (torch.distributed form, used on CPU tensors over gloo in the tests; on the GPU the
    library issues the same broadcasts itself as one RCCL group)."""
    for q in range(len(ranges)):
        src = dist.get_global_rank(group, q) if group is not None else q
        for lo, hi in ((int(ranges[q][0]), int(ranges[q][1])), (int(ranges[q][2]), int(ranges[q][3]))):
            if hi > lo:
                dist.broadcast(buf[lo:hi], src=src, group=group)
    return buf


def exchange(local_block, group=None):
    """all-gather of one rank's [3*S*d | S | S] block -> [world * block] on every rank.
    Works on CUDA tensors over RCCL and on CPU tensors over gloo (tests)."""
    world = dist.get_world_size(group)
    out = torch.empty(world * local_block.numel(), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(out, local_block.contiguous(), group=group)
    return out


class DataParallelBPR:
    """Drop-in for utils.BPRLoss when torch.distributed is initialised: same
    `stageOne(users, pos, neg)` on the GLOBAL batch (identical on every rank)."""

    def __init__(self, recmodel, config, group=None, reduce='rows', shard='batch'):
        from .utils import _AdamView
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        # (row-sharded and column-sharded propagation refuse the optional branches: the first was never wired to carry the smoothing's
        #  extra exchange, the second cannot -- the gate's MLPs mix the columns of a row)
        if getattr(recmodel, 'has_variants', False) and not (getattr(recmodel, 'fused_variants', False) and shard == 'batch'):
            raise NotImplementedError("with the popularity gate / item-item smoothing, data-parallel training needs the fused step "
                                      "(--fused_variants 1) and batch sharding: shard='batch' (reduce='rows' or 'dense')")
        self.model = recmodel
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.weight_decay = config['decay']
        self.lr = config['lr']
        self.opt = _AdamView(recmodel.parameters(), lr=self.lr).bind(recmodel)
        recmodel.config['decay'] = self.weight_decay
        self.lazy = False
        if reduce not in ('rows', 'dense'):
            raise ValueError("reduce must be 'rows' (all-gather of gradient rows) or 'dense' (all-reduce of the table)")
        self.reduce = reduce
        if shard not in ('batch', 'rows', 'cols'):
            raise ValueError("shard must be 'batch' (replicated propagation), 'rows' (row-sharded propagation) or 'cols' (column-sharded tables)")
        self.shard = shard
        self.ranges = None
        if shard == 'rows':
            if reduce != 'rows':
                raise ValueError("row-sharded propagation exchanges gradient rows (reduce='rows')")
            self.ranges = row_ranges(recmodel._adj.indptr, recmodel.n_users, self.world)
        self._comm = None          # lgcn_dp handle: the library's own RCCL communicator (lazy)
        self._comm_ok = None       # decided collectively at the first epoch
        self._gathered = None

    # -- the library's RCCL communicator: collectives are issued from C on the kernels' stream --------
    def _communicator(self):
        """Rank 0 draws the ncclUniqueId, torch.distributed carries the 128 bytes to the other
        ranks, every rank joins.  Only on GPU tensors (the gloo/CPU tests use `exchange`)."""
        if self._comm is None:
            lib = _lib.load()
            buf = (C.c_char * 128)()
            err0 = None
            if self.rank == 0:
                # a failure here must not keep rank 0 out of the broadcast the other ranks are already waiting in:
                # it sends an empty id instead, and every rank treats that as "no communicator" (same collective
                # sequence on every rank: broadcast, then the all-reduce of _own_communicator_ok)
                try:
                    _lib.check(lib.lgcn_dp_unique_id(buf), "lgcn_dp_unique_id")
                except Exception as e:      # noqa: BLE001
                    err0 = e
            box = [b"" if err0 is not None else bytes(buf)]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                                       group=self.group)
            if len(box[0]) != 128:
                raise err0 if err0 is not None else _lib.LgcnError("rank 0 could not create the RCCL unique id")
            h = C.c_void_p()
            _lib.check(lib.lgcn_dp_init(box[0], self.world, self.rank, C.byref(h)), "lgcn_dp_init")
            self._comm = h
        return self._comm

    def _own_communicator_ok(self):
        """True when every rank joined the library's communicator.  If any rank could not (librccl not
        loadable, ncclCommInitRank refused), ALL ranks drop theirs and the epoch runs as the per-step loop
        over torch.distributed's RCCL collectives instead -- same kernels, same results, one host call per
        step instead of per epoch.  (Row-sharded propagation has no such loop and raises.)"""
        if self._comm_ok is None:
            err = None
            dev = self.model._table.device
            # stage 1: can every rank resolve RCCL at all?  ncclCommInitRank is itself a collective: a rank that cannot
            # even load the library must keep the others from entering it.
            have = torch.tensor([1 if _lib.load().lgcn_dp_available() else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(have, op=dist.ReduceOp.MIN, group=self.group)
            if int(have.item()) == 0:
                err = _lib.LgcnError("librccl could not be resolved on every rank")
            else:
                try:
                    self._communicator()
                except Exception as e:      # noqa: BLE001 -- reported below, decided collectively
                    err = e
            flag = torch.tensor([0 if err is None else 1], dtype=torch.int32, device=self.model._table.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            self._comm_ok = int(flag.item()) == 0
            if not self._comm_ok:
                self.close()
                if self.shard == 'rows':
                    raise RuntimeError(f"row-sharded propagation needs the library's RCCL communicator: {err or 'another rank failed'}")
                if self.rank == 0:
                    import sys
                    print(f"[lgcn] own RCCL communicator unavailable ({err or 'another rank failed'}); "
                          "using torch.distributed collectives per step", file=sys.stderr)
        return self._comm_ok

    def ranks_observed(self):
        """How many ranks the collectives of the data path actually span: a SUM all-reduce of 1.0 per rank through the
        library's own RCCL communicator (the one lgcn_train_epoch_dp issues its collectives on), or -- when the ranks agreed
        to fall back -- through torch.distributed.  -> (count, 'library RCCL communicator' | 'torch.distributed')."""
        dev = self.model._table.device
        one = torch.ones(1, dtype=torch.float32, device=dev)
        if dev.type == 'cuda' and os.environ.get("LGCN_DP_PYTHON_LOOP") != "1" and self._own_communicator_ok():
            _lib.check(_lib.load().lgcn_dp_allreduce_sum_f32(self._communicator(), _lib.tp(one), 1, _lib.current_stream()),
                       "lgcn_dp_allreduce_sum_f32")
            return int(round(float(one.item()))), "library RCCL communicator"
        dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.group)
        return int(round(float(one.item()))), "torch.distributed"

    def close(self):
        if self._comm is not None:
            _lib.load().lgcn_dp_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stageOne(self, users, pos, neg):
        loss = self._step(users, pos, neg)
        return loss[0] if self.lazy else loss[0].cpu().item()

    def train_epoch(self, users, pos, neg, global_batch):
        """The loop of main.py:223-225 over already-shuffled device id arrays (identical on every
        rank): consecutive global batches, last one short.  -> device tensor [steps, 3]."""
        m = self.model
        dev = m._table.device
        T = int(users.numel())
        if self.shard == 'cols' and dev.type == 'cuda' and os.environ.get("LGCN_DP_PYTHON_LOOP") != "1" and self._own_communicator_ok():
            users, pos, neg = m._ids(users, dev), m._ids(pos, dev), m._ids(neg, dev)
            st = m._state(max_batch=max(int(global_batch), int(m.config.get('bpr_batch_size', global_batch))), need_ctx=True, dp_world=1)
            lib = _lib.load()
            lib.lgcn_ctx_set_lr(st['ctx'], float(self.opt.param_groups[0]['lr']))
            steps = (T + global_batch - 1) // global_batch
            losses = torch.empty(steps, 3, dtype=torch.float32, device=dev)
            _lib.check(lib.lgcn_train_epoch_dp(st['ctx'], self._communicator(), _lib.tp(users), _lib.tp(pos), _lib.tp(neg), T,
                                               int(global_batch), 3, None, None, _lib.tp(losses), _lib.current_stream()), "lgcn_train_epoch_dp")
            m._cache = None
            return losses
        if dev.type != 'cuda' or os.environ.get("LGCN_DP_PYTHON_LOOP") == "1" or not self._own_communicator_ok():
            if self.shard == 'rows':        # (no per-step form exists: _step would come straight back here)
                raise RuntimeError("row-sharded propagation (shard='rows') runs only through the library's RCCL communicator "
                                   "on CUDA tensors; it has no torch.distributed per-step loop (LGCN_DP_PYTHON_LOOP / CPU)")
            out = [self._step(users[t:t + global_batch], pos[t:t + global_batch], neg[t:t + global_batch])
                   for t in range(0, T, global_batch)]
            return torch.stack(out)
        # one host call per epoch: the C loop issues kernels and RCCL collectives on one stream
        users, pos, neg = m._ids(users, dev), m._ids(pos, dev), m._ids(neg, dev)
        st = m._state(max_batch=max(int(global_batch), int(m.config.get('bpr_batch_size', global_batch))), need_ctx=True,
                      dp_world=self.world,
                      row_subset=owned_rows(self.ranges, self.rank) if self.shard == 'rows' else None)
        lib = _lib.load()
        lib.lgcn_ctx_set_lr(st['ctx'], float(self.opt.param_groups[0]['lr']))
        steps = (T + global_batch - 1) // global_batch
        losses = torch.empty(steps, 3, dtype=torch.float32, device=dev)
        n = self.world * int(lib.lgcn_dp_block_floats(st['ctx'], int(global_batch), self.world))
        if self.reduce == 'rows' and (self._gathered is None or self._gathered.numel() < n):
            self._gathered = torch.empty(n, dtype=torch.float32, device=dev)
        mode = 2 if self.shard == 'rows' else (0 if self.reduce == 'rows' else 1)
        rr = np.ascontiguousarray(self.ranges, np.int64) if self.shard == 'rows' else None
        _lib.check(lib.lgcn_train_epoch_dp(st['ctx'], self._communicator(), _lib.tp(users), _lib.tp(pos), _lib.tp(neg), T,
                                           int(global_batch), mode, _lib.npp(rr) if rr is not None else None,
                                           _lib.tp(self._gathered) if self.reduce == 'rows' else None,
                                           _lib.tp(losses), _lib.current_stream()), "lgcn_train_epoch_dp")
        m._cache = None
        return losses

    def gather_table(self):
        """shard='cols': the full [N, d] table (every rank's columns, all-gathered) -- what evaluation scores with."""
        t = self.model._table.detach().contiguous()
        parts = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(parts, t, group=self.group)
        return torch.cat(parts, dim=1)

    def _step_cols(self, users, pos, neg):
        """one column-sharded step with the all-reduce through torch.distributed (the fallback of the C loop)"""
        m = self.model
        dev = m._table.device
        users, pos, neg = m._ids(users, dev), m._ids(pos, dev), m._ids(neg, dev)
        B = int(users.numel())
        st = m._state(max_batch=max(B, int(m.config.get('bpr_batch_size', B))), need_ctx=True, dp_world=1)
        lib = _lib.load()
        lib.lgcn_ctx_set_lr(st['ctx'], float(self.opt.param_groups[0]['lr']))
        stream = _lib.current_stream()
        ptr = C.c_void_p()
        _lib.check(lib.lgcn_train_step_cols_part1(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B, C.byref(ptr), stream),
                   "lgcn_train_step_cols_part1")
        partial = torch.as_tensor(_DevArray(ptr.value, 3 * B), device=dev)
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=self.group)
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        _lib.check(lib.lgcn_train_step_cols_part2(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B, _lib.tp(loss), stream),
                   "lgcn_train_step_cols_part2")
        m._cache = None
        return loss

    def _step(self, users, pos, neg):
        if self.shard == 'rows':
            return self.train_epoch(users, pos, neg, int(len(users)))[0]
        if self.shard == 'cols':
            return self._step_cols(users, pos, neg)
        m = self.model
        dev = m._table.device
        users, pos, neg = m._ids(users, dev), m._ids(pos, dev), m._ids(neg, dev)
        B = int(users.numel())
        st = m._state(max_batch=max(B, int(m.config.get('bpr_batch_size', B))), need_ctx=True,
                      dp_world=self.world)
        lib = _lib.load()
        lib.lgcn_ctx_set_lr(st['ctx'], float(self.opt.param_groups[0]['lr']))
        stream = _lib.current_stream()
        loss = torch.empty(3, dtype=torch.float32, device=dev)
        if self.reduce == 'rows':
            _lib.check(lib.lgcn_train_step_dp_part1(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B,
                                                    self.world, self.rank, stream), "lgcn_train_step_dp_part1")
            n = int(lib.lgcn_dp_block_floats(st['ctx'], B, self.world))
            gathered = exchange(st['contrib'][:n], self.group)
            gptr = _lib.tp(gathered)
        else:       # literal form: all-reduce of the (fixed-point) gradient table -- N*d*8 bytes per step
            _lib.check(lib.lgcn_train_step_dp_dense_part1(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B,
                                                          self.world, self.rank, stream),
                       "lgcn_train_step_dp_dense_part1")
            dist.all_reduce(st['G64'], op=dist.ReduceOp.SUM, group=self.group)
            gp, gn = C.c_void_p(), C.c_int32()
            _lib.check(lib.lgcn_ctx_gate_total(st['ctx'], C.byref(gp), C.byref(gn)), "lgcn_ctx_gate_total")
            dist.all_reduce(st['terms'][:(3 if gn.value else 2) * B], op=dist.ReduceOp.SUM, group=self.group)
            if gn.value:        # popularity gate: the ranks' fixed-point sums of the MLP parameter gradients
                dist.all_reduce(torch.as_tensor(_DevArray(gp.value, gn.value, "<i8"), device=dev), op=dist.ReduceOp.SUM, group=self.group)
            gptr = None
        _lib.check(lib.lgcn_train_step_dp_part2(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B,
                                                self.world, gptr, _lib.tp(loss), stream),
                   "lgcn_train_step_dp_part2")
        m._cache = None
        return loss
