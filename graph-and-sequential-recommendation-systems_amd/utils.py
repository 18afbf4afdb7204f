"""Training utilities, mirror of the reference's utils.py (utils.py:25-217).

BPRLoss.stageOne runs the fused HIP step; the samplers and the epoch shuffle run
the native bit-exact restatements (glibc rand() / numpy legacy MT19937 streams)."""
import os
from time import time

import numpy as np
import torch
from torch import nn, optim

from . import world
from . import _lib
from .dataloader import BasicDataset

# ==================== Negative Sampling Extension ====================
# utils.py:25-34 binds the cppimport-built module `sampling` and seeds it with world.seed;
# here the same four functions come from liblgcn_hip.so.
try:
    from . import sampling
    sampling.seed(world.seed)
    sample_ext = True
except Exception as _e:           # library missing: say so (the reference falls back silently)
    world.cprint(f"Cpp extension not loaded: {_e}")
    sample_ext = False


def compiled_sampling_module():
    """The same plugin as a COMPILED extension module named `sampling` (csrc/sampling_module.cpp -> sources/sampling<ext>.so,
    pybind11), loaded by path exactly as utils.py:25-34 loads the reference's cppimport build.  It binds the same C ABI
    as `sampling` above -- one library instance, ONE rand() stream shared by both.  Raises ImportError if it cannot be
    built here (no pybind11 / Python headers / g++); the ctypes binding above does not need it."""
    from . import build as _build
    return _build.load_sampling_module()


class _AdamView(optim.Adam):
    """torch.optim.Adam whose state tensors are views of the fused kernel's m/v tables, so
    `bpr.opt` keeps the reference's surface (param_groups[0]['lr'], state_dict(),
    load_state_dict(), LR schedulers -- main.py:36,61,197) while the update itself happens
    in the epilogue of the last backward SpMM."""

    def bind(self, model):
        self._model = model
        return self

    def _sync_from_kernel(self):
        m = self._model
        st = m._dev
        if not st or 'adam_m' not in st:
            return
        nu = m.n_users
        step = float(m.adam_step)
        for p, sl in ((m.embedding_user.weight, slice(0, nu)), (m.embedding_item.weight, slice(nu, None))):
            s = self.state[p]
            s['step'] = torch.tensor(step, dtype=torch.float32)
            s['exp_avg'], s['exp_avg_sq'] = st['adam_m'][sl], st['adam_v'][sl]
        if getattr(m, 'use_pop_gate', False) and 'gate_m' in st:       # the gate's MLPs: views of the fused step's flat Adam state
            off = 0
            for p in m.gate_parameters():
                s = self.state[p]
                s['step'] = torch.tensor(step, dtype=torch.float32)
                s['exp_avg'] = st['gate_m'][off:off + p.numel()].view(p.shape)
                s['exp_avg_sq'] = st['gate_v'][off:off + p.numel()].view(p.shape)
                off += p.numel()

    def state_dict(self):
        self._sync_from_kernel()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        m = self._model
        st = m._state(need_ctx=True)
        nu = m.n_users
        step = 0
        with torch.no_grad():
            for p, sl in ((m.embedding_user.weight, slice(0, nu)), (m.embedding_item.weight, slice(nu, None))):
                s = self.state.get(p, {})
                if 'exp_avg' in s:
                    st['adam_m'][sl].copy_(s['exp_avg'])
                    st['adam_v'][sl].copy_(s['exp_avg_sq'])
                    step = int(float(s['step']))
            if getattr(m, 'use_pop_gate', False) and 'gate_m' in st:
                off = 0
                for p in m.gate_parameters():
                    s = self.state.get(p, {})
                    if 'exp_avg' in s:
                        st['gate_m'][off:off + p.numel()].copy_(s['exp_avg'].reshape(-1))
                        st['gate_v'][off:off + p.numel()].copy_(s['exp_avg_sq'].reshape(-1))
                    off += p.numel()
        _lib.load().lgcn_ctx_set_step(st['ctx'], step)
        self._sync_from_kernel()

    def step(self, closure=None):
        raise RuntimeError("the Adam update is fused into BPRLoss.stageOne (HIP kernel epilogue); "
                           "call stageOne, or build a plain torch.optim.Adam for an unfused loop")


class DeferredLoss:
    """What BPRLoss.stageOne returns with --lazy_loss 1: the step's loss as a number that is read from the device only when somebody
    looks at it.  The reference returns `loss.cpu().item()` (utils.py:64) -- a host round trip per step during which the GPU idles
    (Gowalla: 5 650 instead of 6 600 steps/s) -- and then only ever adds the values up and formats the mean (Procedure.py:61-68,
    main.py:223-236).  This object supports exactly that arithmetic (+, -, *, / with numbers and with each other, float(), format(),
    comparisons, numpy conversion) and keeps sums as lists of 0-dim device tensors until a value is needed."""
    __slots__ = ("parts", "scale", "offset")

    def __init__(self, parts, scale=1.0, offset=0.0):
        self.parts, self.scale, self.offset = parts, scale, offset

    def __float__(self):
        if len(self.parts) == 1:
            v = float(self.parts[0].item())
        else:
            v = float(torch.stack([p.reshape(()) for p in self.parts]).double().sum().item()) if self.parts else 0.0
        return v * self.scale + self.offset

    def _lin(self, scale, offset):
        return DeferredLoss(self.parts, self.scale * scale, self.offset * scale + offset)

    def __add__(self, o):
        if isinstance(o, DeferredLoss):
            if o.scale == self.scale:
                return DeferredLoss(self.parts + o.parts, self.scale, self.offset + o.offset)
            return self._lin(1.0, float(o))
        return self._lin(1.0, float(o))
    __radd__ = __add__

    def __sub__(self, o):
        return self + (-o if not isinstance(o, DeferredLoss) else o._lin(-1.0, 0.0))

    def __rsub__(self, o):
        return self._lin(-1.0, float(o))

    def __neg__(self):
        return self._lin(-1.0, 0.0)

    def __mul__(self, o):
        return self._lin(float(o), 0.0)
    __rmul__ = __mul__

    def __truediv__(self, o):
        return self._lin(1.0 / float(o), 0.0)

    def __rtruediv__(self, o):
        return float(o) / float(self)

    def __format__(self, spec):
        return format(float(self), spec)

    def __repr__(self):
        return repr(float(self))
    __str__ = __repr__

    def __array__(self, dtype=None, copy=None):
        return np.asarray(float(self), dtype=dtype or np.float64)

    def item(self):
        return float(self)

    def __lt__(self, o): return float(self) < float(o)
    def __le__(self, o): return float(self) <= float(o)
    def __gt__(self, o): return float(self) > float(o)
    def __ge__(self, o): return float(self) >= float(o)
    def __eq__(self, o): return float(self) == float(o)
    def __hash__(self): return hash(float(self))


class BPRLoss:
    """utils.py:38-64."""

    def __init__(self, recmodel: nn.Module, config: dict):
        self.model = recmodel
        self.weight_decay = config['decay']
        self.lr = config['lr']
        # the optional branches (popularity gate, item-item smoothing) run inside the fused step too (model.fused_variants);
        # --fused_variants 0 (or a gate shape outside the kernel's limits) trains them through autograd and a plain torch
        # Adam over ALL parameters -- the reference's own sequence, utils.py:53-64
        self.fused = (not getattr(recmodel, 'has_variants', False)) or bool(getattr(recmodel, 'fused_variants', False))
        if self.fused:
            self.opt = _AdamView(recmodel.parameters(), lr=self.lr).bind(recmodel)
        else:
            self.opt = optim.Adam(recmodel.parameters(), lr=self.lr)
        recmodel.config['decay'] = self.weight_decay
        self.lazy = False          # True: stageOne returns a 0-dim device tensor (no host sync)
        self.deferred = bool(int(config.get('lazy_loss', 0)))     # stageOne returns a DeferredLoss (float-like, read on demand)

    def stageOne(self, users, pos, neg):
        if not self.fused:
            m = self.model
            dev = m._table.device
            users, pos, neg = (torch.as_tensor(np.asarray(t) if not torch.is_tensor(t) else t).to(dev).long() for t in (users, pos, neg))
            loss, reg_loss = m.bpr_loss(users, pos, neg)
            loss = loss + reg_loss * self.weight_decay
            self.opt.zero_grad()
            loss.backward()
            self.opt.step()
            m.invalidate_cache()
            return loss.detach() if self.lazy else loss.cpu().item()
        lr = self.opt.param_groups[0]['lr']
        out = self.model.fused_step(users, pos, neg, lr=lr)
        if self.lazy:
            return out[0]
        if self.deferred:                          # --lazy_loss 1: a number that is read from the device when it is looked at
            return DeferredLoss([out[0]])
        return out[0].cpu().item()                 # utils.py:64 (host sync per step)


# ==================== Sampling ====================
def _pos_csr(dataset):
    if hasattr(dataset, 'pos_csr'):
        return dataset.pos_csr()
    return sampling._csr_of(dataset.allPos)


def sampler_mode(dataset):
    """'cpp' / 'python' per --sampler; 'auto' picks cpp unless a user has no positives
    (where the reference's native sampler dies with SIGFPE)."""
    mode = world.config.get('sampler', 'auto')
    if mode == 'auto':
        if not sample_ext:
            return 'python'
        indptr, _ = _pos_csr(dataset)
        return 'python' if np.any(np.diff(indptr[:dataset.n_users + 1]) == 0) else 'cpp'
    return mode


def UniformSample_original(dataset, neg_ratio=1):
    """utils.py:68-81."""
    dataset: BasicDataset
    if sample_ext and sampler_mode(dataset) == 'cpp':
        S = sampling.sample_negative(dataset.n_users, dataset.m_items, dataset.trainDataSize,
                                     _pos_csr(dataset), neg_ratio)
    else:
        S = UniformSample_original_python(dataset)
    return S


def UniformSample_original_python(dataset):
    """utils.py:84-110 on the numpy-legacy stream (native, bit-exact): int64 [<=trainDataSize,3]."""
    indptr, indices = _pos_csr(dataset)
    S = np.empty((dataset.trainDataSize, 3), np.int64)
    rows = _lib.load().lgcn_sample_python(dataset.n_users, dataset.m_items, dataset.trainDataSize,
                                          _lib.npp(indptr), _lib.npp(indices), _lib.npp(S))
    if rows < 0:
        _lib.check(1, "lgcn_sample_python")
    return S[:rows]


# ==================== Utility helpers ====================
def set_seed(seed):
    """utils.py:114-120; the numpy legacy stream lives in the native library."""
    np.random.seed(seed)
    _lib.load().lgcn_np_seed(int(seed) & 0xFFFFFFFF)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)
    torch.manual_seed(seed)


def getFileName():
    """Legacy weight file name lgn-<dataset>-<layers>-<dim>.pth.tar / mf-<dataset>-<dim>.pth.tar (utils.py:123-132)."""
    dim = world.config['latent_dim_rec']
    stem = {'mf': f"mf-{world.dataset}-{dim}",
            'lgn': f"lgn-{world.dataset}-{world.config['lightGCN_n_layers']}-{dim}"}[world.model_name]
    return os.path.join(world.PATH, stem + ".pth.tar")


def minibatch(*tensors, **kwargs):
    """Consecutive slices of `batch_size` rows of every input, last one short; always a tuple, like the
    fork (utils.py:135-139)."""
    size = kwargs.get('batch_size', world.config['bpr_batch_size'])
    total = len(tensors[0])
    start = 0
    while start < total:
        yield tuple(t[start:start + size] for t in tensors)
        start += size


def shuffle_indices(n):
    """idx = np.arange(n); np.random.shuffle(idx) on the native numpy-legacy stream."""
    perm = np.empty(n, np.int64)
    _lib.check(_lib.load().lgcn_np_shuffle_perm(n, _lib.npp(perm)), "lgcn_np_shuffle_perm")
    return perm


_SHUFFLE_WS = {}      # (device, n) -> workspace tensor of the device shuffle


def shuffle_indices_device(n, device):
    """The same permutation from the same stream, produced on the GPU (csrc/lgcn_shuffle.hip) -> device int64 [n]."""
    lib = _lib.load()
    key = (str(device), int(n))
    ws = _SHUFFLE_WS.get(key)
    if ws is None:
        _SHUFFLE_WS.clear()
        ws = _SHUFFLE_WS[key] = torch.empty(int(lib.lgcn_np_shuffle_perm_device_workspace(int(n))), dtype=torch.uint8, device=device)
    perm = torch.empty(int(n), dtype=torch.int64, device=device)
    _lib.check(lib.lgcn_np_shuffle_perm_device(int(n), _lib.tp(perm), _lib.tp(ws), int(ws.numel()), _lib.current_stream()),
               "lgcn_np_shuffle_perm_device")
    return perm


def shuffle(*arrays, **kwargs):
    """utils.py:142-151."""
    require_indices = kwargs.get('indices', False)
    if len(set(len(x) for x in arrays)) != 1:
        raise ValueError("All inputs must have same length.")
    idx = shuffle_indices(len(arrays[0]))
    result = []
    for x in arrays:
        if torch.is_tensor(x):
            result.append(x[torch.from_numpy(idx).to(x.device)])
        else:
            result.append(x[idx])
    result = tuple(result)
    return (result, idx) if require_indices else result


class timer:
    """Context-manager timer with named accumulators -- the API Procedure.py:50,81-82 expects
    (the fork's own utils.timer lost it)."""
    TAPE = [-1]
    NAMED_TAPE = {}

    @staticmethod
    def get():
        return timer.TAPE.pop() if len(timer.TAPE) > 1 else -1

    @staticmethod
    def dict(select_keys=None):
        hint = "|"
        keys = timer.NAMED_TAPE.keys() if select_keys is None else select_keys
        for key in keys:
            hint = hint + f"{key}:{timer.NAMED_TAPE[key]:.2f}|"
        return hint

    @staticmethod
    def zero(select_keys=None):
        for key in (timer.NAMED_TAPE.keys() if select_keys is None else select_keys):
            timer.NAMED_TAPE[key] = 0

    def __init__(self, tape=None, **kwargs):
        if kwargs.get('name'):
            timer.NAMED_TAPE[kwargs['name']] = timer.NAMED_TAPE.get(kwargs['name'], 0.)
            self.named = kwargs['name']
        else:
            self.named = False
            self.tape = tape or timer.TAPE

    def __enter__(self):
        self.start = time()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        if self.named:
            timer.NAMED_TAPE[self.named] += time() - self.start
        else:
            self.tape.append(time() - self.start)


# ==================== Evaluation Metrics (utils.py:173-217) ====================
def _discounts(k):
    return 1.0 / np.log2(np.arange(2, k + 2))


def RecallPrecision_ATk(test_data, r, k):
    """Sums over the batch of recall@k (hits / |ground truth|) and precision@k (hits / k)."""
    hits = r[:, :k].sum(1)
    gt_sizes = np.fromiter((len(t) for t in test_data), dtype=np.int64, count=len(test_data))
    return {'recall': np.sum(hits / gt_sizes), 'precision': np.sum(hits) / k}


def NDCGatK_r(test_data, r, k):
    """Sum over the batch of DCG@k / IDCG@k, IDCG from min(k, |ground truth|) leading ones."""
    assert len(r) == len(test_data)
    disc = _discounts(k)
    ideal = np.zeros((len(test_data), k))
    for row, items in zip(ideal, test_data):
        row[:min(k, len(items))] = 1
    idcg = (ideal * disc).sum(axis=1)
    idcg[idcg == 0.] = 1.
    return np.sum((r[:, :k] * disc).sum(axis=1) / idcg)


def getLabel(groundTruth, predictTopK):
    """0/1 float32 vector: is the j-th predicted item in the ground truth?"""
    truth = groundTruth if isinstance(groundTruth, (list, set, tuple, np.ndarray)) else [groundTruth]
    return np.fromiter((1.0 if x in truth else 0.0 for x in predictTopK), dtype=np.float32, count=len(predictTopK))
