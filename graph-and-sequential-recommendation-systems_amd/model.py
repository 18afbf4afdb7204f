"""LightGCN, mirror of the reference's model.py (model.py:37-231) for the hot path.

Same constructor `(config, dataset)`, same parameters (`embedding_user.weight`
[n_users,d], `embedding_item.weight` [m_items,d] -> same state_dict keys), same
methods (`computer`, `getEmbedding`, `bpr_loss`, `getUsersRating`, `forward`) plus
the optional `invalidate_cache()` hook main.py:190-191 probes for.

All propagation / loss / optimiser arithmetic runs in the hand-written gfx950
kernels behind include/lgcn_hip.h.  There is no torch.sparse.mm path and no CPU
fallback: on a machine without a HIP device every compute method raises.

The fork's optional branches (8f-4) are supported on the autograd path: the
popularity gate (model.py:66-96,139-157,176-181: two small dense MLPs, torch ops) and the item-item
smoothing (model.py:99-109,228-229: one more CSR SpMM, the same HIP kernel on the item-item graph, with
its transpose in backward).  With either switched on, BPRLoss.stageOne is the reference's own sequence
(bpr_loss -> backward -> torch.optim.Adam over ALL parameters) with every propagation in the HIP
kernels; the fused single-launch-chain step covers the default model only.
"""
import ctypes as C

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import world
from . import _lib


def _act_code(name):
    return {'fp32': _lib.F32, 'bf16': _lib.BF16, 'fp8': _lib.FP8}[name]


def _act_tables(n_tables, N, d, act_dtype, dev):
    """workspace of n_tables [N,d] tables of a storage type, as one tensor whose leading index is the table (fp8 tables
    are byte blobs: rows + row scales, lgcn_table_bytes each)"""
    if act_dtype == _lib.FP8:
        return torch.zeros(n_tables, int(_lib.load().lgcn_table_bytes(N, d, _lib.FP8)), dtype=torch.uint8, device=dev)
    return torch.zeros(n_tables, N, d, dtype=torch.float32 if act_dtype == _lib.F32 else torch.bfloat16, device=dev)


class _Propagate(torch.autograd.Function):
    """computer() with autograd: forward = lgcn_propagate_mean, backward = the Horner
    chain (1/(K+1)) * sum_k A^k g through lgcn_spmm_csr (A_hat is symmetric)."""

    @staticmethod
    def forward(ctx, user_w, item_w, model):
        ctx.model = model
        return model._propagate_dense()

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        g = (grad_out.contiguous().float() / float(m.n_layers + 1))
        h = g
        for _ in range(m.n_layers):
            h = g + m._spmm(h)
        return h[:m.n_users], h[m.n_users:], None


class _I2ISmooth(torch.autograd.Function):
    """model.py:228-229: items + alpha * (I2I @ items); backward g + alpha * (I2I^T @ g)."""

    @staticmethod
    def forward(ctx, items, model):
        ctx.model = model
        return model._i2i_apply(items, transpose=False)

    @staticmethod
    def backward(ctx, grad):
        return ctx.model._i2i_apply(grad, transpose=True), None


class LightGCN(nn.Module):
    def __init__(self, config, dataset):
        super().__init__()
        self.config = config
        self.dataset = dataset
        self.device = world.device
        self.n_users = dataset.n_users
        self.m_items = dataset.m_items
        self.latent_dim = config['latent_dim_rec']
        self.n_layers = config['lightGCN_n_layers']
        self.keep_prob = config.get('keep_prob', 0.6)
        if self.latent_dim not in (32, 64, 128, 256):
            raise ValueError("latent_dim_rec must be 32, 64, 128 or 256 for the HIP kernels")
        if not (1 <= self.n_layers <= _lib.MAX_LAYERS):
            raise ValueError(f"lightGCN_n_layers must be in 1..{_lib.MAX_LAYERS}")

        # Same RNG consumption as model.py:57-60 (two nn.Embedding ctors, then two normal_),
        # then both tables are moved into ONE contiguous [N,d] storage: rows [0,n_users) are
        # embedding_user.weight, rows [n_users,N) embedding_item.weight (the cat of model.py:209
        # becomes a no-op).
        self.embedding_user = nn.Embedding(self.n_users, self.latent_dim)
        self.embedding_item = nn.Embedding(self.m_items, self.latent_dim)
        nn.init.normal_(self.embedding_user.weight, std=0.1)
        nn.init.normal_(self.embedding_item.weight, std=0.1)
        with torch.no_grad():
            table = torch.cat([self.embedding_user.weight, self.embedding_item.weight], dim=0).contiguous()
        self._table = table
        self._rebind()

        # popularity gate (model.py:66-96): same modules, built in the same order -> same initial weights
        self.use_pop_gate = bool(config.get('use_pop_gate', False))
        self.pop_hidden = int(config.get('pop_hidden', 32))
        self.gate_hidden = int(config.get('gate_hidden', 64))
        self.gate_entropy_coeff = float(config.get('gate_entropy_coeff', 1e-4))
        self.pop_gate_temp = float(config.get('pop_gate_temp', 1.0))
        if self.use_pop_gate:
            counts = torch.clamp(torch.from_numpy(np.asarray(dataset.items_D)).float(), min=0.0)
            pop = torch.log1p(counts)
            self.item_pop_scalar = (pop - pop.mean()) / (pop.std() + 1e-8)
            self.pop_mlp = nn.Sequential(nn.Linear(1, self.pop_hidden), nn.ReLU(), nn.Linear(self.pop_hidden, self.latent_dim))
            self.gate_mlp = nn.Sequential(nn.Linear(self.latent_dim * 2, self.gate_hidden), nn.ReLU(), nn.Linear(self.gate_hidden, 1))
        else:
            self.item_pop_scalar, self.pop_mlp, self.gate_mlp = None, None, None

        # item-item graph (model.py:99-109): a CSR [m_items, m_items] from an .npz; unreadable -> warning, off
        self.use_item_item = bool(config.get('use_item_item', False))
        self.i2i_alpha = float(config.get('i2i_alpha', 0.0))
        self._i2i = None
        if self.use_item_item and config.get('i2i_path', None):
            try:
                import scipy.sparse as sp
                m = sp.load_npz(config['i2i_path']).tocsr().astype(np.float32)
                if m.shape != (self.m_items, self.m_items):
                    raise ValueError(f"shape {m.shape} is not ({self.m_items}, {self.m_items})")
                m.sort_indices()
                self._i2i = m
                world.cprint(f"[I2I] loaded {config['i2i_path']}, nnz={m.nnz}")
            except Exception as e:      # noqa: BLE001 -- the reference warns and goes on without it
                world.cprint(f"[I2I] WARNING: cannot load {config['i2i_path']}: {e}")

        self._adj = dataset.getSparseGraphCSR() if hasattr(dataset, 'getSparseGraphCSR') else None
        if self._adj is None:                       # generic BasicDataset: COO -> CSR
            g = dataset.getSparseGraph().coalesce().cpu()
            import scipy.sparse as sp
            idx = g.indices().numpy()
            self._adj = sp.csr_matrix((g.values().numpy().astype(np.float32), (idx[0], idx[1])),
                                      shape=tuple(g.shape))
            self._adj.sort_indices()
        N = self.n_users + self.m_items
        if self._adj.shape != (N, N):
            raise ValueError("adjacency shape does not match n_users + m_items")
        if len(self._adj.indices) and (self._adj.indices.min() < 0 or self._adj.indices.max() >= N):
            raise ValueError("adjacency column index out of range")
        self._gate_flat = None      # the eight MLP tensors of the gate in ONE buffer (what the fused step reads / updates)
        self._Graph = None
        self._dev = None            # device-side state (graph, workspace, context)
        self._cache = None          # propagated embeddings memoised between invalidate_cache() calls
        self._rating_cache = None
        self.f = nn.Sigmoid()

    # -- parameters live in one table ------------------------------------------------
    def _rebind(self):
        self.embedding_user.weight.data = self._table[:self.n_users]
        self.embedding_item.weight.data = self._table[self.n_users:]

    def _apply(self, fn, recurse=True):
        # keep the single-storage invariant through .to()/.cuda()/.float()
        new = fn(self._table)
        if new is not self._table:
            self._table = new
            self._drop_device_state()
        self._rebind()
        for mod in (self.pop_mlp, self.gate_mlp):
            if mod is not None:
                mod._apply(fn)
        if self.item_pop_scalar is not None:
            self.item_pop_scalar = fn(self.item_pop_scalar)
        for p in (self.embedding_user.weight, self.embedding_item.weight):
            if p.grad is not None:
                p.grad = fn(p.grad)
        return self

    def _check_table(self):
        """state_dict loads and optimisers write through the parameter views; make sure
        nobody replaced them by independent tensors."""
        u, i = self.embedding_user.weight, self.embedding_item.weight
        if (u.data_ptr() != self._table.data_ptr()
                or i.data_ptr() != self._table.data_ptr() + self.n_users * self.latent_dim * 4):
            with torch.no_grad():
                self._table[:self.n_users].copy_(u.data)
                self._table[self.n_users:].copy_(i.data)
            self._rebind()
        if self.use_pop_gate:
            self._pack_gate()

    def gate_parameters(self):
        """The gate's eight tensors in torch's named_parameters order (pop_mlp.0.weight ... gate_mlp.2.bias)."""
        return list(self.pop_mlp.parameters()) + list(self.gate_mlp.parameters())

    def _pack_gate(self):
        """The fused step reads and updates the gate's MLPs as ONE fp32 buffer (include/lgcn_hip.h: gate_params): keep the
        nn.Linear parameters views of it.  Returns True when the buffer was (re)made -- a context bound to the old one is stale."""
        ps = self.gate_parameters()
        flat, off, ok = self._gate_flat, 0, self._gate_flat is not None
        for prm in ps:
            ok = ok and prm.device == flat.device and prm.data_ptr() == flat.data_ptr() + 4 * off and prm.is_contiguous()
            off += prm.numel()
        if ok:
            return False
        flat = torch.cat([prm.data.reshape(-1).float() for prm in ps]).contiguous()
        off = 0
        for prm in ps:
            prm.data = flat[off:off + prm.numel()].view(prm.shape)
            off += prm.numel()
        self._gate_flat = flat
        if self._dev is not None and self._dev.get('ctx'):
            _lib.load().lgcn_ctx_destroy(self._dev['ctx'])
            self._dev['ctx'] = None
        return True

    @property
    def fused_variants(self):
        """The optional branches run inside the fused HIP step (default) unless --fused_variants 0 asks for the autograd
        path or the gate's shape is outside what k_triplet_gate holds in LDS (hidden sizes <= 64, d <= 128)."""
        if not self.has_variants:
            return False
        if not int(self.config.get('fused_variants', 1)):
            return False
        if self.use_pop_gate and (self.latent_dim > 128 or self.pop_hidden > 64 or self.gate_hidden > 64 or self.pop_gate_temp <= 0):
            return False
        return True

    @property
    def Graph(self):
        """torch sparse COO of A_hat as the reference keeps it (model.py:63); built lazily."""
        if self._Graph is None:
            self._Graph = self.dataset.getSparseGraph()
        return self._Graph

    # -- device state -----------------------------------------------------------------
    def _drop_device_state(self):
        if self._dev is not None and self._dev.get('ctx'):
            _lib.load().lgcn_ctx_destroy(self._dev['ctx'])
        if self._dev is not None and self._dev.get('graph_rs') is not None:
            self._dev['graph_rs'].close()
        if self._dev is not None and self._dev.get('graph') is not None:
            self._dev['graph'].close()
        for key in ('i2i', 'i2i_t', 'i2i_s', 'i2i_ts'):
            if self._dev is not None and self._dev.get(key) is not None:
                self._dev[key].close()
        self._dev = None
        self._cache = None
        self._rating_cache = None

    def __del__(self):
        try:
            self._drop_device_state()
        except Exception:
            pass

    def _state(self, max_batch=None, need_ctx=False, dp_world=1, row_subset=None):
        _lib.require_gpu()
        if not self._table.is_cuda:
            raise _lib.LgcnError("model parameters are not on the GPU: call .to(world.device) first")
        self._check_table()
        dev = self._table.device
        if self._dev is None:
            a = self._adj
            from . import reorder
            order, xcd_start = reorder.row_order(self.config.get('row_order', 'natural'), self.dataset, a,
                                                 cache_dir=getattr(self.dataset, 'path', None))
            self._dev = {
                'graph': _lib.Graph(torch.from_numpy(np.ascontiguousarray(a.indptr, np.int32)).to(dev),
                                    torch.from_numpy(np.ascontiguousarray(a.indices, np.int32)).to(dev),
                                    torch.from_numpy(np.ascontiguousarray(a.data, np.float32)).to(dev),
                                    d_max=self.latent_dim, row_order=order, xcd_start=xcd_start),
                'ctx': None, 'max_batch': 0,
            }
        st = self._dev
        if row_subset is not None and st.get('graph_rs') is None:
            # row-sharded propagation: the training context works on a plan of the owned rows only
            # (same device CSR arrays); evaluation keeps the full plan
            g = st['graph']
            st['graph_rs'] = _lib.Graph(g.indptr, g.indices, g.vals, d_max=self.latent_dim,
                                        row_order=np.ascontiguousarray(row_subset, np.int32))
        if need_ctx:
            max_batch = int(max_batch or self.config.get('bpr_batch_size', 2048))
            # a context is bound to ONE plan: the owned rows (row-sharded epochs, which exchange the other rows) or all
            # rows (every other caller: fused_step / fused_epoch / stageOne / batch-sharded data parallel).  A context
            # made for one must not serve the other -- it would propagate over the owned rows only, with no exchange.
            rows = 'owned' if row_subset is not None else 'all'
            if (st['ctx'] is None or st['max_batch'] < max_batch or st.get('dp_world', 1) != dp_world
                    or st.get('ctx_rows', 'all') != rows):
                self._make_ctx(max_batch, dp_world, rows)
        return st

    def _make_ctx(self, max_batch, dp_world, rows=None):
        st, dev = self._dev, self._table.device
        lib = _lib.load()
        old_step = 0
        if st['ctx'] is not None:
            old_step = lib.lgcn_ctx_get_step(st['ctx'])
            lib.lgcn_ctx_destroy(st['ctx'])
            st['ctx'] = None
        N, d, K = self.n_users + self.m_items, self.latent_dim, self.n_layers
        act_dtype = _act_code(self.config.get('act_dtype', 'fp32'))
        if 'adam_m' not in st:
            st['adam_m'] = torch.zeros(N, d, dtype=torch.float32, device=dev)
            st['adam_v'] = torch.zeros(N, d, dtype=torch.float32, device=dev)
        dense_last = self._dense_last(int(self.config.get('bpr_batch_size', max_batch)))     # the configured batch, not this context's capacity
        variants = self.fused_variants
        if variants:
            dense_last = True       # the optional branches score on the layer mean of EVERY row (model.py:221-229)
        st['act'] = _act_tables(max(1, K if dense_last else K - 1), N, d, act_dtype, dev)
        st['G64'] = torch.zeros(N, d, dtype=torch.int64, device=dev)
        st['bitmap'] = torch.zeros(2 * ((N + 31) // 32), dtype=torch.int32, device=dev)
        st['terms'] = torch.zeros(3 * max_batch, dtype=torch.float32, device=dev)
        shard = (max_batch + dp_world - 1) // dp_world
        gate_p = sum(prm.numel() for prm in self.gate_parameters()) if (variants and self.use_pop_gate) else 0
        # exchange block of a rank (lgcn_dp_block_floats): gradient rows, loss / reg (/ entropy) terms, the gate's fixed-point MLP sums
        st['contrib'] = torch.zeros(3 * shard * d + 3 * shard + 2 + 2 * gate_p, dtype=torch.float32, device=dev)
        st['err'] = torch.zeros(1, dtype=torch.int32, device=dev)
        cfg = _lib.TrainConfig()
        rows = rows or st.get('ctx_rows', 'all')
        cfg.graph = (st['graph_rs'] if rows == 'owned' else st['graph']).handle
        cfg.n_users, cfg.d, cfg.K = self.n_users, d, K
        cfg.act_dtype = act_dtype
        cfg.E0, cfg.adam_m, cfg.adam_v = self._table.data_ptr(), st['adam_m'].data_ptr(), st['adam_v'].data_ptr()
        cfg.act, cfg.G64 = st['act'].data_ptr(), st['G64'].data_ptr()
        cfg.bitmap, cfg.terms, cfg.contrib = st['bitmap'].data_ptr(), st['terms'].data_ptr(), st['contrib'].data_ptr()
        cfg.err, cfg.max_batch = st['err'].data_ptr(), max_batch
        cfg.decay = float(self.config.get('decay', 1e-4))
        cfg.lr = float(self.config.get('lr', 1e-3))
        cfg.beta1, cfg.beta2, cfg.eps = 0.9, 0.999, 1e-8
        cfg.reg_ego = 1 if str(self.config.get('reg_rows', 'propagated')) == 'ego' else 0
        if cfg.reg_ego and variants:
            raise _lib.LgcnError("--reg_rows ego (upstream LightGCN's L2 term) is defined for the default model, not with "
                                 "--use_pop_gate / --use_item_item")
        cfg.xcd_remap = int(self.config.get('xcd_remap', 1))
        cfg.dense_last = int(dense_last)
        cfg.hub_nnz = int(self.config.get('hub_nnz', 0))          # 0: library default; < 0: off
        cfg.hub_chunk = int(self.config.get('hub_chunk', 0))
        if variants and self.i2i_active:
            # the fused step's graphs hold alpha * I2I and its transpose (model.py:228-229 scales after the product)
            for key, mat in (('i2i_s', self._i2i), ('i2i_ts', self._i2i.T.tocsr())):
                if key not in st:
                    mat = mat.tocsr().astype(np.float32)
                    mat.sort_indices()
                    st[key] = _lib.Graph(torch.from_numpy(mat.indptr.astype(np.int32)).to(dev), torch.from_numpy(mat.indices.astype(np.int32)).to(dev),
                                         torch.from_numpy((np.float32(self.i2i_alpha) * mat.data).astype(np.float32)).to(dev), d_max=d)
            cfg.i2i, cfg.i2i_t = st['i2i_s'].handle, st['i2i_ts'].handle
        if variants and self.use_pop_gate:
            self._pack_gate()
            P = self._gate_flat.numel()
            for key in ('gate_m', 'gate_v', 'gate_grad'):
                if key not in st or st[key].numel() != P:
                    st[key] = torch.zeros(P, dtype=torch.float32, device=dev)
            st['item_pop'] = self.item_pop_scalar.to(device=dev, dtype=torch.float32).contiguous()
            cfg.item_pop, cfg.gate_params = st['item_pop'].data_ptr(), self._gate_flat.data_ptr()
            cfg.gate_adam_m, cfg.gate_adam_v, cfg.gate_grad = st['gate_m'].data_ptr(), st['gate_v'].data_ptr(), st['gate_grad'].data_ptr()
            cfg.pop_hidden, cfg.gate_hidden = self.pop_hidden, self.gate_hidden
            cfg.gate_entropy_coeff, cfg.pop_gate_temp = self.gate_entropy_coeff, self.pop_gate_temp
        st['dense_last'] = dense_last
        h = C.c_void_p()
        _lib.check(lib.lgcn_ctx_create(C.byref(cfg), C.byref(h)), "lgcn_ctx_create")
        lib.lgcn_ctx_set_step(h, old_step)
        st['ctx'], st['max_batch'], st['dp_world'], st['table_ptr'] = h, max_batch, dp_world, self._table.data_ptr()
        st['ctx_rows'] = rows

    def _dense_last(self, batch):
        """Last forward layer: on the 3B batch rows only (default) or densely?  The batch rows hold an
        expected  B * (E[deg(user)] + E_pop[deg(item)] + E[deg(item)])  non-zeros -- the positive is drawn
        proportionally to item popularity (sum d^2 / sum d), the negative uniformly.  Per non-zero the
        per-slot gathers cost 2.5-5x what the dense kernel does (measured: Gowalla 17 us at 0.15 nnz(A_hat),
        Yelp-shaped 236 us at 2.1, Amazon-shaped 364 us at 0.34 against dense layers of 29 / 46 / 210 us: a
        slot is one workgroup, and a hub row's tiles are walked serially), so 'auto' goes dense above 0.3
        ('--dense_last 0/1' forces it; the result is the same up to fp32 summation order)."""
        mode = str(self.config.get('dense_last', 'auto'))
        if mode in ('0', '1'):
            return mode == '1'
        deg = np.diff(self._adj.indptr).astype(np.float64)
        du, di = deg[:self.n_users], deg[self.n_users:]
        per_triplet = du.mean() + (di * di).sum() / max(di.sum(), 1.0) + di.mean()
        return bool(batch * per_triplet > 0.3 * self._adj.nnz)

    # -- kernels ------------------------------------------------------------------------
    def _spmm(self, x):
        return self._state()['graph'].spmm(x.float(), _lib.F32)

    @property
    def i2i_active(self):
        return self.use_item_item and self._i2i is not None and self.i2i_alpha > 0.0

    @property
    def has_variants(self):
        """True when a branch outside the fused step is on (BPRLoss then takes the autograd path)."""
        return self.use_pop_gate or self.i2i_active

    def _i2i_apply(self, x, transpose):
        st = self._state()
        key = 'i2i_t' if transpose else 'i2i'
        if key not in st:
            m = self._i2i.T.tocsr() if transpose else self._i2i
            m.sort_indices()
            dev = self._table.device
            st[key] = _lib.Graph(torch.from_numpy(m.indptr.astype(np.int32)).to(dev), torch.from_numpy(m.indices.astype(np.int32)).to(dev),
                                 torch.from_numpy(m.data.astype(np.float32)).to(dev), d_max=self.latent_dim)
        x = x.contiguous().float()
        return x + self.i2i_alpha * st[key].spmm(x, _lib.F32)

    def _fuse_item_embeddings(self, items_emb):
        """model.py:139-157: gate * items + (1 - gate) * pop_vec, gate = sigmoid(gate_mlp([items, pop_vec]) / T)."""
        pop_vec = self.pop_mlp(self.item_pop_scalar.unsqueeze(1))
        gate_logit = self.gate_mlp(torch.cat([items_emb, pop_vec], dim=1))
        if self.pop_gate_temp != 1.0:
            gate_logit = gate_logit / self.pop_gate_temp
        gate = torch.sigmoid(gate_logit)
        self._last_item_gate = gate
        return gate * items_emb + (1.0 - gate) * pop_vec

    def _propagate_dense(self):
        st = self._state()
        N, d, K = self.n_users + self.m_items, self.latent_dim, self.n_layers
        act_dtype = _act_code(self.config.get('act_dtype', 'fp32'))
        work = st.get('eval_work')
        if K > 1 and (work is None or st.get('eval_work_dtype') != act_dtype):
            work = st['eval_work'] = _act_tables(K - 1, N, d, act_dtype, self._table.device)
            st['eval_work_dtype'] = act_dtype
        out = torch.empty(N, d, dtype=torch.float32, device=self._table.device)
        _lib.check(_lib.load().lgcn_propagate_mean(
            st['graph'].handle, _lib.tp(self._table), K, d, act_dtype, _lib.tp(work) if K > 1 else None,
            _lib.tp(out), _lib.current_stream()), "lgcn_propagate_mean")
        return out

    # -- reference API --------------------------------------------------------------------
    def invalidate_cache(self):
        """Hook probed by main.py:190-191 before each Test."""
        self._cache = None
        self._rating_cache = None

    def train(self, mode=True):
        self.invalidate_cache()
        return super().train(mode)

    def propagated_table(self):
        """The [N,d] fp32 table computer() splits into users / items (item-item smoothing included), memoised
        between invalidate_cache() / train() calls."""
        if self._cache is None:
            with torch.no_grad():
                out = self._propagate_dense()
                cr = getattr(self, 'col_range', None)
                if cr is not None and cr[1] - cr[0] < cr[2]:
                    # a column shard of a wider model (parallel.column_shard): propagation is per column, so the ranks' propagated
                    # columns side by side ARE the full propagated table -- what evaluation has to score with
                    import torch.distributed as dist
                    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() * (cr[1] - cr[0]) == cr[2]):
                        raise _lib.LgcnError("this model holds columns %d..%d of %d: evaluation needs the process group that holds the others" % cr)
                    parts = [torch.empty_like(out) for _ in range(dist.get_world_size())]
                    dist.all_gather(parts, out.contiguous())
                    out = torch.cat(parts, dim=1)
                if self.i2i_active:
                    out = torch.cat([out[:self.n_users], self._i2i_apply(out[self.n_users:], transpose=False)], dim=0)
                self._cache = out
        return self._cache

    def rating_table(self):
        """What getUsersRating scores with (model.py:114-123): the propagated users and the FINAL item
        embeddings (popularity-gated when the gate is on).  This is what the fused evaluation kernels read."""
        if not self.use_pop_gate:
            return self.propagated_table()
        if getattr(self, '_rating_cache', None) is None:
            with torch.no_grad():
                out = self.propagated_table()
                self._rating_cache = torch.cat([out[:self.n_users], self._fuse_item_embeddings(out[self.n_users:])], dim=0).contiguous()
        return self._rating_cache

    def computer(self):
        """model.py:201-231 -> (all_users [n_users,d], all_items [m_items,d])."""
        if not self.training and not torch.is_grad_enabled():
            out = self.propagated_table()       # eval: propagate once instead of once per user batch
            return out[:self.n_users, :], out[self.n_users:, :]
        if torch.is_grad_enabled() and (self.embedding_user.weight.requires_grad
                                        or self.embedding_item.weight.requires_grad):
            out = _Propagate.apply(self.embedding_user.weight, self.embedding_item.weight, self)
        else:
            out = self._propagate_dense()
        all_users, all_items = out[:self.n_users, :], out[self.n_users:, :]
        if self.i2i_active:                      # model.py:228-229
            all_items = _I2ISmooth.apply(all_items, self) if all_items.requires_grad else self._i2i_apply(all_items, False)
        return all_users, all_items

    def getUsersRating(self, users):
        all_users, all_items = self.computer()
        i_emb = self._fuse_item_embeddings(all_items) if self.use_pop_gate else all_items
        return torch.matmul(all_users[users], i_emb.t())

    def getEmbedding(self, users, pos_items, neg_items):
        all_users, all_items = self.computer()
        i_emb = self._fuse_item_embeddings(all_items) if self.use_pop_gate else all_items
        return all_users[users], i_emb[pos_items], i_emb[neg_items], all_users, all_items

    def bpr_loss(self, users, pos, neg):
        """model.py:162-183 (unfused, autograd-capable): (bpr [- entropy term of the gate], reg_loss)."""
        u, pos_e, neg_e, _, _ = self.getEmbedding(users, pos, neg)
        pos_scores = torch.sum(u * pos_e, dim=1)
        neg_scores = torch.sum(u * neg_e, dim=1)
        bpr = -torch.mean(F.logsigmoid(pos_scores - neg_scores))
        if str(self.config.get('reg_rows', 'propagated')) == 'ego':      # upstream LightGCN: userEmb0 / posEmb0 / negEmb0
            u0, p0, n0 = self.embedding_user.weight[users.long()], self.embedding_item.weight[pos.long()], self.embedding_item.weight[neg.long()]
            reg_loss = (0.5 * (u0.norm(2).pow(2) + p0.norm(2).pow(2) + n0.norm(2).pow(2))) / float(u.shape[0])
        else:
            reg_loss = (0.5 * (u.norm(2).pow(2) + pos_e.norm(2).pow(2) + neg_e.norm(2).pow(2))) / float(u.shape[0])
        loss = bpr
        if self.use_pop_gate and hasattr(self, "_last_item_gate"):
            gates = torch.cat([self._last_item_gate[pos], self._last_item_gate[neg]], dim=0)
            gates = torch.clamp(gates, 1e-6, 1.0 - 1e-6)
            entropy = -(gates * torch.log(gates) + (1 - gates) * torch.log(1 - gates)).mean()
            loss = loss - self.gate_entropy_coeff * entropy
        return loss, reg_loss

    def forward(self, users, items):
        all_users, all_items = self.computer()
        i_emb = self._fuse_item_embeddings(all_items) if self.use_pop_gate else all_items
        return (all_users[users] * i_emb[items]).sum(dim=1)

    # -- fused path (what BPRLoss.stageOne calls) ------------------------------------------
    @staticmethod
    def _ids(t, dev):
        if not torch.is_tensor(t):
            t = torch.as_tensor(np.asarray(t))
        return t.to(device=dev, dtype=torch.int32).contiguous()

    def fused_step(self, users, pos, neg, loss_out=None, lr=None):
        """One BPRLoss.stageOne (utils.py:53-64) in the HIP kernels.  Returns a device
        tensor [3] = (bpr + decay*reg, bpr, reg); no host synchronisation."""
        if self.has_variants and not self.fused_variants:
            raise RuntimeError("this configuration of the popularity gate / item-item smoothing is outside the fused step "
                               "(--fused_variants 0, or gate hidden sizes > 64 / d > 128): use BPRLoss.stageOne (autograd path)")
        dev = self._table.device
        # the reference's call shape: three torch.long device tensors (main.py:217-225) -- narrowed inside the step's own launch
        # sequence (lgcn_train_step_i64) instead of three conversion kernels and allocations here
        as_i64 = all(torch.is_tensor(t) and t.dtype == torch.int64 and t.device == dev and t.is_contiguous() for t in (users, pos, neg))
        if not as_i64:
            users, pos, neg = self._ids(users, dev), self._ids(pos, dev), self._ids(neg, dev)
        B = int(users.numel())
        st = self._state(max_batch=max(B, int(self.config.get('bpr_batch_size', B))), need_ctx=True,
                         dp_world=(self._dev or {}).get('dp_world', 1))
        lib = _lib.load()
        if lr is not None:
            lib.lgcn_ctx_set_lr(st['ctx'], float(lr))
        if loss_out is None:
            loss_out = torch.empty(3, dtype=torch.float32, device=dev)
        if as_i64:
            if st.get('ids32') is None or st['ids32'].numel() < 3 * B:
                st['ids32'] = torch.empty(3 * max(B, st['max_batch']), dtype=torch.int32, device=dev)
            _lib.check(lib.lgcn_train_step_i64(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B, _lib.tp(st['ids32']),
                                               _lib.tp(loss_out), _lib.current_stream()), "lgcn_train_step")
        else:
            _lib.check(lib.lgcn_train_step(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), B,
                                           _lib.tp(loss_out), _lib.current_stream()), "lgcn_train_step")
        self.invalidate_cache()
        return loss_out

    def fused_epoch(self, users, pos, neg, batch_size, lr=None):
        """The loop of main.py:223-225 over already-shuffled device id arrays, one C call.
        Returns a device tensor [steps,3] of per-step (loss, bpr, reg)."""
        if self.has_variants and not self.fused_variants:
            raise RuntimeError("this configuration of the optional branches is outside the fused step: loop BPRLoss.stageOne")
        dev = self._table.device
        users, pos, neg = self._ids(users, dev), self._ids(pos, dev), self._ids(neg, dev)
        T = int(users.numel())
        steps = (T + batch_size - 1) // batch_size
        st = self._state(max_batch=batch_size, need_ctx=True, dp_world=(self._dev or {}).get('dp_world', 1))
        lib = _lib.load()
        if lr is not None:
            lib.lgcn_ctx_set_lr(st['ctx'], float(lr))
        losses = torch.empty(steps, 3, dtype=torch.float32, device=dev)
        _lib.check(lib.lgcn_train_epoch(st['ctx'], _lib.tp(users), _lib.tp(pos), _lib.tp(neg), T, int(batch_size),
                                        _lib.tp(losses), _lib.current_stream()), "lgcn_train_epoch")
        self.invalidate_cache()
        return losses

    def check_device_errors(self):
        if self._dev and self._dev.get('ctx'):
            _lib.check(_lib.load().lgcn_ctx_check(self._dev['ctx'], _lib.current_stream()), "device id check")

    @property
    def adam_step(self):
        return _lib.load().lgcn_ctx_get_step(self._dev['ctx']) if self._dev and self._dev.get('ctx') else 0
