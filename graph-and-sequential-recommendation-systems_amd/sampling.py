"""The reference's native plugin `sampling` (sources/sampling.cpp:95-106), same four
functions, same dtypes/shapes/order and the same glibc rand() stream, served by
liblgcn_hip.so.  `utils.py` binds it exactly where the reference binds the
cppimport module (utils.py:25-31)."""
import numpy as np

from . import _lib


def _csr_of(allPos):
    """list[list[int]] / list[np.ndarray] -> (indptr int64, indices int32)."""
    if isinstance(allPos, tuple) and len(allPos) == 2:          # NEW: zero-copy CSR overload
        return np.ascontiguousarray(allPos[0], np.int64), np.ascontiguousarray(allPos[1], np.int32)
    lens = np.fromiter((len(a) for a in allPos), dtype=np.int64, count=len(allPos))
    indptr = np.zeros(len(allPos) + 1, np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = (np.concatenate([np.asarray(a, np.int32) for a in allPos])
               if len(allPos) and indptr[-1] else np.zeros(0, np.int32))
    return indptr, np.ascontiguousarray(indices, np.int32)


def randint(end):
    """generate int between [0 end]  (sampling.cpp:22-25)"""
    return _lib.load().lgcn_sampling_randint(int(end))


def seed(seed):
    """set random seed  (sampling.cpp:88-91)"""
    _lib.load().lgcn_sampling_seed(int(seed) & 0xFFFFFFFF)


def sample_negative(user_num, item_num, train_num, allPos, neg_num):
    """sampling negatives for all  (sampling.cpp:27-56) -> int32 [user_num*(train_num//user_num), 2+neg_num]"""
    indptr, indices = _csr_of(allPos)
    if len(indptr) - 1 < user_num:
        raise ValueError("allPos has fewer rows than user_num")
    S = np.empty((int(user_num) * (int(train_num) // int(user_num)), 2 + int(neg_num)), np.int32)
    _lib.check(_lib.load().lgcn_sample_negative(int(user_num), int(item_num), int(train_num),
                                                _lib.npp(indptr), _lib.npp(indices), int(neg_num),
                                                _lib.npp(S)), "sampling.sample_negative")
    return S


def sample_negative_ByUser(users, item_num, allPos, neg_num):
    """sampling negatives for given users  (sampling.cpp:58-86) -> int32 [len(users), 2+neg_num]"""
    indptr, indices = _csr_of(allPos)
    users = np.ascontiguousarray(users, np.int32)
    if len(users) and (users.min() < 0 or users.max() >= len(indptr) - 1):
        raise ValueError("user id out of range")
    S = np.empty((len(users), 2 + int(neg_num)), np.int32)
    _lib.check(_lib.load().lgcn_sample_negative_by_user(_lib.npp(users), len(users), int(item_num),
                                                        _lib.npp(indptr), _lib.npp(indices), int(neg_num),
                                                        _lib.npp(S)), "sampling.sample_negative_ByUser")
    return S


_DEVICE_CSR = {}          # id(indptr array) -> (device indptr, device indices, workspace)


def sample_negative_device(user_num, item_num, train_num, allPos, device):
    """NEW: sample_negative (neg_num = 1) on the GPU -- the same int32 rows from the same rand()
    stream, returned as a DEVICE tensor [user_num*(train_num//user_num), 3] (no host loop, no upload)."""
    import torch
    indptr, indices = _csr_of(allPos)
    key = (indptr.ctypes.data, indices.ctypes.data, str(device))
    ent = _DEVICE_CSR.get(key)
    lib = _lib.load()
    if ent is None:
        ws = int(lib.lgcn_sample_negative_device_workspace(int(user_num), int(train_num)))
        ent = (torch.from_numpy(indptr).to(device), torch.from_numpy(indices).to(device),
               torch.empty(max(ws, 256), dtype=torch.uint8, device=device), indptr, indices)
        _DEVICE_CSR.clear()
        _DEVICE_CSR[key] = ent
    d_ip, d_ix, work = ent[:3]
    T = int(user_num) * (int(train_num) // int(user_num))
    S = torch.empty(T, 3, dtype=torch.int32, device=device)
    _lib.check(lib.lgcn_sample_negative_device(int(user_num), int(item_num), int(train_num), _lib.npp(indptr),
                                               _lib.tp(d_ip), _lib.tp(d_ix), _lib.tp(S), _lib.tp(work), int(work.numel()),
                                               _lib.current_stream()), "sampling.sample_negative_device")
    return S
