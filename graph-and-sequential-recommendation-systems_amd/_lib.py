"""ctypes binding of liblgcn_hip.so (include/lgcn_hip.h).

There is NO fallback: if the library cannot be loaded the import of any compute
entry point raises, loudly.  The CPU oracle under oracle/ is test infrastructure
and is never reached from here."""
import ctypes as C
import os

from . import build as _build

F32, BF16, FP8 = 0, 1, 2
ABI_VERSION = 11
MAX_LAYERS = 8

_c_i32p = C.POINTER(C.c_int32)
_c_i64p = C.POINTER(C.c_int64)
_c_f32p = C.POINTER(C.c_float)
_vp = C.c_void_p


def fp8_col_of_byte(d):
    """Column held by byte b of an fp8 table row (include/lgcn_hip.h, LGCN_FP8): with L = d/16 lanes per row, the 16 bytes at l*16
    hold the four 4-column chunks j*L + l, j = 0..3  ->  int64 [d]."""
    import numpy as np
    b = np.arange(d)
    l, j, e = b // 16, (b % 16) // 4, b % 4
    return ((j * (d // 16) + l) * 4 + e).astype(np.int64)


class TrainConfig(C.Structure):
    """Mirror of lgcn_train_config (include/lgcn_hip.h)."""
    _fields_ = [
        ("graph", _vp),
        ("n_users", C.c_int32), ("d", C.c_int32), ("K", C.c_int32), ("act_dtype", C.c_int32),
        ("E0", _vp), ("adam_m", _vp), ("adam_v", _vp),
        ("act", _vp), ("G64", _vp), ("bitmap", _vp), ("terms", _vp), ("contrib", _vp),
        ("err", _vp), ("max_batch", C.c_int32),
        ("decay", C.c_float),
        ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("xcd_remap", C.c_int32), ("dense_last", C.c_int32),
        ("hub_nnz", C.c_int32), ("hub_chunk", C.c_int32),
        ("i2i", _vp), ("i2i_t", _vp),
        ("item_pop", _vp), ("gate_params", _vp), ("gate_adam_m", _vp), ("gate_adam_v", _vp), ("gate_grad", _vp),
        ("pop_hidden", C.c_int32), ("gate_hidden", C.c_int32),
        ("gate_entropy_coeff", C.c_float), ("pop_gate_temp", C.c_float),
        ("reg_ego", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/lgcn_hip.h declares
SIGNATURES = {
    "lgcn_abi_version": (C.c_int, []),
    "lgcn_last_error": (C.c_char_p, []),
    "lgcn_device_available": (C.c_int, []),
    "lgcn_np_shuffle_perm_device_workspace": (C.c_int64, [C.c_int64]),
    "lgcn_np_shuffle_perm_device": (C.c_int, [C.c_int64, _vp, _vp, C.c_int64, _vp]),
    "lgcn_table_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "lgcn_to_fp8": (C.c_int, [_vp, _vp, C.c_int64, C.c_int32, _vp]),
    "lgcn_sampling_seed": (None, [C.c_uint]),
    "lgcn_sampling_randint": (C.c_int, [C.c_int]),
    "lgcn_sample_negative": (C.c_int, [C.c_int, C.c_int, C.c_int64, _vp, _vp, C.c_int, _vp]),
    "lgcn_sample_negative_by_user": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.c_int, _vp]),
    "lgcn_sample_negative_device_workspace": (C.c_int64, [C.c_int, C.c_int64]),
    "lgcn_sampler_test_margin": (None, [C.c_int64, C.c_int64]),
    "lgcn_sample_negative_device": (C.c_int, [C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "lgcn_np_seed": (None, [C.c_uint32]),
    "lgcn_sample_python": (C.c_int64, [C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp]),
    "lgcn_np_shuffle_perm": (C.c_int, [C.c_int64, _vp]),
    "lgcn_build_user_item_csr": (C.c_int, [C.c_int, C.c_int, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lgcn_adj_rowsum": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "lgcn_build_norm_adj": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "lgcn_graph_create": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int32, _vp, C.c_int64, _vp, C.POINTER(_vp)]),
    "lgcn_graph_destroy": (None, [_vp]),
    "lgcn_spmm_csr": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "lgcn_propagate_mean": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "lgcn_apply_perm": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, _vp, _vp, _vp, _vp]),
    "lgcn_ctx_create": (C.c_int, [C.POINTER(TrainConfig), C.POINTER(_vp)]),
    "lgcn_ctx_destroy": (None, [_vp]),
    "lgcn_ctx_get_step": (C.c_int64, [_vp]),
    "lgcn_ctx_hub_rows": (C.c_int64, [_vp]),
    "lgcn_ctx_set_step": (None, [_vp, C.c_int64]),
    "lgcn_ctx_set_lr": (None, [_vp, C.c_double]),
    "lgcn_ctx_set_dp_local": (C.c_int, [_vp, C.c_int]),
    "lgcn_train_step": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "lgcn_train_epoch": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_int32, _vp, _vp]),
    "lgcn_dp_block_floats": (C.c_int64, [_vp, C.c_int32, C.c_int32]),
    "lgcn_train_step_dp_part1": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "lgcn_train_step_dp_dense_part1": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    "lgcn_ctx_gate_total": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]),
    "lgcn_train_step_dp_part2": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "lgcn_ctx_check": (C.c_int, [_vp, _vp]),
    "lgcn_eval_topk": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "lgcn_eval_topk_masked": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp]),
    "lgcn_eval_mask_words": (C.c_int64, [C.c_int32, C.c_int32]),
    "lgcn_eval_build_masks": (C.c_int, [_vp, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp]),
    "lgcn_eval_topk_fp32": (C.c_int, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "lgcn_eval_metrics": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "lgcn_dp_available": (C.c_int, []),
    "lgcn_dp_unique_id": (C.c_int, [_vp]),
    "lgcn_dp_init": (C.c_int, [_vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "lgcn_train_step_i64": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp, _vp, _vp]),
    "lgcn_train_step_cols_part1": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, C.POINTER(_vp), _vp]),
    "lgcn_train_step_cols_part2": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int32, _vp, _vp]),
    "lgcn_dp_allreduce_sum_f32": (C.c_int, [_vp, _vp, C.c_int64, _vp]),
    "lgcn_dp_init_loopback": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "lgcn_dp_destroy": (None, [_vp]),
    "lgcn_dp_world": (C.c_int, [_vp]),
    "lgcn_dp_rank": (C.c_int, [_vp]),
    "lgcn_train_epoch_dp": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "lgcn_rs_phase": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "lgcn_rs_buffer": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(_vp), C.POINTER(C.c_int32)]),
}

_LIB = None


class LgcnError(RuntimeError):
    pass


def load():
    """Load liblgcn_hip.so.  A stale or missing library is rebuilt first -- except inside a rank of
    a distributed job or under a profiler, where the library must already exist (build it once
    with `python -m graph-and-sequential-recommendation-systems_amd.build` or __graft_entry__.build()).
    Raises if impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = _build.LIB_PATH
    if os.environ.get("LGCN_LIB_PATH"):
        pass                                   # explicit tuning variant: use as is
    elif _build.is_stale():
        if _build.must_not_build():
            if not os.path.exists(path):
                raise LgcnError(
                    f"liblgcn_hip.so is missing at {path}; this process is a distributed rank or runs under a "
                    "profiler and will not compile it. Build it first (__graft_entry__.build()).")
            # present but not provably current (e.g. built by hand with other flags): use it
        elif _build.find_hipcc() is not None:
            _build.build()
        elif not os.path.exists(path):
            raise LgcnError(
                f"liblgcn_hip.so is missing at {path} and hipcc is not available to build it. "
                "The MI355X HIP library is the product path; there is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise LgcnError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LgcnError(f"{path} does not export {name} (stale build?)") from e
        fn.restype, fn.argtypes = res, args
    if lib.lgcn_abi_version() != ABI_VERSION:
        raise LgcnError("liblgcn_hip.so ABI version mismatch")
    _LIB = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().lgcn_last_error()
        raise LgcnError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def npp(a):
    """numpy array -> void* (array must stay alive for the call)."""
    return a.ctypes.data_as(_vp)


def tp(t):
    """torch tensor -> void* device/host pointer."""
    return _vp(t.data_ptr()) if t is not None else _vp(0)


def require_gpu():
    import torch
    if not torch.cuda.is_available() or not load().lgcn_device_available():
        raise LgcnError("no MI355X/HIP device visible: the LightGCN HIP kernels cannot run "
                        "(there is deliberately no CPU fallback)")


def current_stream():
    import torch
    return _vp(torch.cuda.current_stream().cuda_stream)


class Graph:
    """Owner of an lgcn_graph handle over device CSR tensors (kept alive here)."""

    def __init__(self, indptr, indices, vals, d_max=256, row_order=None, xcd_start=None):
        import numpy as np
        import torch
        require_gpu()
        self.indptr = indptr.to(torch.int32).contiguous()
        self.indices = indices.to(torch.int32).contiguous()
        self.vals = vals.to(torch.float32).contiguous()
        if self.indices.numel() == 0:        # keep pointers non-null for an empty matrix
            self.indices = torch.zeros(1, dtype=torch.int32, device=self.indptr.device)
            self.vals = torch.zeros(1, dtype=torch.float32, device=self.indptr.device)
            nnz = 0
        else:
            nnz = int(indices.numel())
        self.n_rows = int(self.indptr.numel()) - 1
        self.nnz = nnz
        self.d_max = int(d_max)
        self.row_order = None
        if row_order is not None:
            self.row_order = torch.as_tensor(row_order).to(device=self.indptr.device, dtype=torch.int32).contiguous()
        xs = None
        if xcd_start is not None:
            xs = np.ascontiguousarray(xcd_start, np.int64)
            if xs.shape != (9,):
                raise LgcnError("Graph: xcd_start must hold 9 positions")
        h = _vp()
        check(load().lgcn_graph_create(tp(self.indptr), tp(self.indices), tp(self.vals), self.n_rows, nnz,
                                       int(d_max), tp(self.row_order),
                                       int(self.row_order.numel()) if self.row_order is not None else self.n_rows,
                                       npp(xs) if xs is not None else None, C.byref(h)), "lgcn_graph_create")
        self.handle = h

    def to_fp8(self, x):
        """fp32 [n_rows, d] -> the library's fp8 table (uint8 tensor: n_rows*d E4M3 bytes, then n_rows fp32 row scales, padded)."""
        import torch
        x = x.contiguous().float()
        n, d = int(x.shape[0]), int(x.shape[1])
        out = torch.zeros(int(load().lgcn_table_bytes(n, d, FP8)), dtype=torch.uint8, device=x.device)
        check(load().lgcn_to_fp8(tp(x), tp(out), n, d, current_stream()), "lgcn_to_fp8")
        return out

    @staticmethod
    def from_fp8(tab, n, d):
        """decode an fp8 table (as to_fp8 / an fp8 SpMM output lays it out) to fp32 [n, d] with torch (test helper)"""
        import torch
        q = tab[:n * d].view(torch.float8_e4m3fn).view(n, d).float()
        sc = tab[n * d:n * d + 4 * n].view(torch.float32)
        out = torch.empty_like(q)
        out[:, torch.from_numpy(fp8_col_of_byte(d)).to(q.device)] = q           # the bytes of a row are chunk-interleaved
        return out * sc[:, None]

    def spmm_fp8(self, xq, d, y_dtype=FP8):
        """A @ X for an fp8 table xq (see to_fp8): -> fp8 table (y_dtype FP8) or fp32 [n_rows, d] (F32)."""
        import torch
        if y_dtype == FP8:
            y = torch.zeros(int(load().lgcn_table_bytes(self.n_rows, d, FP8)), dtype=torch.uint8, device=xq.device)
        else:
            y = torch.empty(self.n_rows, d, dtype=torch.float32, device=xq.device)
        check(load().lgcn_spmm_csr(self.handle, tp(xq), FP8, tp(y), y_dtype, int(d), current_stream()), "lgcn_spmm_csr")
        return y

    def spmm(self, x, y_dtype=None):
        import torch
        if x.dim() != 2 or x.shape[0] != self.n_rows or x.shape[1] not in (32, 64, 128, 256) or x.shape[1] > self.d_max:
            raise LgcnError(f"Graph.spmm: x must be [{self.n_rows}, d] with d in (32,64,128,256) and d <= d_max={self.d_max}; "
                            f"got {tuple(x.shape)}")
        if x.device != self.indptr.device:
            raise LgcnError("Graph.spmm: x is not on the graph's device")
        x = x.contiguous()
        xd = BF16 if x.dtype == torch.bfloat16 else F32
        if xd == F32:
            x = x.float()
        yd = xd if y_dtype is None else y_dtype
        if yd == FP8:
            y = torch.zeros(int(load().lgcn_table_bytes(self.n_rows, int(x.shape[1]), FP8)), dtype=torch.uint8, device=x.device)
        else:
            y = torch.empty(x.shape, dtype=torch.bfloat16 if yd == BF16 else torch.float32, device=x.device)
        check(load().lgcn_spmm_csr(self.handle, tp(x), xd, tp(y), yd, int(x.shape[1]), current_stream()), "lgcn_spmm_csr")
        return y

    def close(self):
        if getattr(self, "handle", None):
            load().lgcn_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
