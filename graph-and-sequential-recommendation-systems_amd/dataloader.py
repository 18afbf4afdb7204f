"""Dataset + graph, mirror of the reference's dataloader.py (dataloader.py:26-255).

Same files (`<dir>/train.txt`, `test.txt`: one `uid item item ...` line per user;
`s_pre_adj_mat.npz`: scipy CSR cache of A_hat), same attributes and methods.
The normalised adjacency is assembled by the native builder of include/lgcn_hip.h
(direct COO -> CSR + transpose, bit-identical values) instead of the reference's
dok -> lil block assignment (71 s on Gowalla, dataloader.py:223-227)."""
from os.path import join
from time import time
import warnings

import numpy as np
import scipy.sparse as sp
import torch
from torch.utils.data import Dataset

from . import world
from . import _lib
from .world import cprint


class BasicDataset(Dataset):
    """The dataset protocol of dataloader.py:26-48: five read-only attributes and three
    methods that every concrete dataset provides."""

    def _missing(self, *_a, **_k):
        raise NotImplementedError

    n_users = property(_missing)
    m_items = property(_missing)
    trainDataSize = property(_missing)
    testDict = property(_missing)
    allPos = property(_missing)
    getUserItemFeedback = _missing
    getUserPosItems = _missing
    getSparseGraph = _missing


def _read_interactions(path):
    """`uid item item ...` lines -> (unique users in file order, users[E], items[E])."""
    uniq, users, items = [], [], []
    with open(path, 'r') as f:
        for l in f:
            cols = l.split()
            if len(cols) < 2:                       # blank line or user without items
                continue
            uid = int(cols[0])
            its = np.array(cols[1:], dtype=np.int64)
            uniq.append(uid)
            users.append(np.full(len(its), uid, dtype=np.int64))
            items.append(its)
    if not users:
        return (np.zeros(0, np.int64),) * 3
    return np.array(uniq, dtype=np.int64), np.concatenate(users), np.concatenate(items)


class Loader(BasicDataset):
    """dataloader.py:51-255."""

    def __init__(self, config=world.config, path=None):
        if path is None:
            path = join(world.DATA_PATH, world.dataset)
        self.path = path
        cprint(f'loading [{self.path}]')
        self.split = config.get('A_split', False)
        self.folds = config.get('A_n_fold', 100)

        self.trainUniqueUsers, self.trainUser, self.trainItem = _read_interactions(join(path, 'train.txt'))
        self.testUniqueUsers, self.testUser, self.testItem = _read_interactions(join(path, 'test.txt'))
        self.traindataSize, self.testDataSize = len(self.trainUser), len(self.testUser)
        self.n_user = int(max(self.trainUser.max(initial=0), self.testUser.max(initial=0))) + 1
        self.m_item = int(max(self.trainItem.max(initial=0), self.testItem.max(initial=0))) + 1

        print(f"{self.trainDataSize} interactions for training")
        print(f"{self.testDataSize} interactions for testing")
        print(f"{world.dataset} Sparsity : {(self.trainDataSize + self.testDataSize) / self.n_users / self.m_items:.12f}")

        # UserItemNet: canonical CSR (sorted columns, duplicates summed), fp32 -- dataloader.py:133-136
        lib = _lib.load()
        E = self.traindataSize
        self._r_indptr = np.zeros(self.n_user + 1, np.int64)
        nnz = np.zeros(1, np.int64)
        indices = np.empty(max(E, 1), np.int32)
        vals = np.empty(max(E, 1), np.float32)
        _lib.check(lib.lgcn_build_user_item_csr(self.n_user, self.m_item, E, _lib.npp(self.trainUser),
                                                _lib.npp(self.trainItem), _lib.npp(self._r_indptr),
                                                _lib.npp(indices), _lib.npp(vals), _lib.npp(nnz)),
                   "lgcn_build_user_item_csr")
        self._r_indices, self._r_vals = indices[:int(nnz[0])], vals[:int(nnz[0])]
        self._finish_init()

    def _finish_init(self):
        """Everything dataloader.py:133-150 derives from the user-item CSR."""
        self.UserItemNet = sp.csr_matrix((self._r_vals, self._r_indices, self._r_indptr),
                                         shape=(self.n_user, self.m_item))
        self.users_D = np.array(self.UserItemNet.sum(axis=1)).squeeze()
        self.users_D[self.users_D == 0.] = 1.
        self.items_D = np.array(self.UserItemNet.sum(axis=0)).squeeze()
        self.items_D[self.items_D == 0.] = 1.

        self._allPos = None
        self.__testDict = self.__build_test()
        print(f"{world.dataset} is ready to go")
        self.Graph = None
        self._norm_adj = None

    n_users = property(lambda self: self.n_user)
    m_items = property(lambda self: self.m_item)
    trainDataSize = property(lambda self: self.traindataSize)
    testDict = property(lambda self: self.__testDict)

    @property
    def allPos(self):
        """list of int32 index arrays, sorted ascending (dataloader.py:178-180)."""
        if self._allPos is None:
            self._allPos = np.split(self._r_indices, self._r_indptr[1:-1])
        return self._allPos

    def pos_csr(self):
        """NEW: allPos as zero-copy CSR (int64 indptr, int32 indices) for the native sampler."""
        return self._r_indptr, self._r_indices

    def __build_test(self):
        """{user: [test items]} in file order (dataloader.py:165-171)."""
        grouped = {}
        for u, i in zip(self.testUser.tolist(), self.testItem.tolist()):
            grouped.setdefault(u, []).append(i)
        return grouped

    def getUserItemFeedback(self, users, items):
        return np.array(self.UserItemNet[users, items]).astype('uint8').reshape((-1,))

    def getUserPosItems(self, users):
        ap = self.allPos
        return [ap[u] for u in users]

    # ---------- adjacency ----------
    def _convert_sp_mat_to_sp_tensor(self, X):
        """scipy sparse -> torch sparse COO, fp32 values / int64 indices (dataloader.py:183-190)."""
        m = X.tocoo()
        ij = torch.from_numpy(np.vstack([m.row, m.col]).astype(np.int64))
        return torch.sparse_coo_tensor(ij, torch.from_numpy(m.data.astype(np.float32)), torch.Size(m.shape))

    def _split_A_hat(self, A):
        """`A_n_fold` row blocks of A_hat, the last one taking the remainder (dataloader.py:192-201)."""
        n_all = self.n_users + self.m_items
        step = n_all // self.folds
        bounds = [(k * step, n_all if k == self.folds - 1 else (k + 1) * step) for k in range(self.folds)]
        return [self._convert_sp_mat_to_sp_tensor(A[lo:hi]).coalesce().to(world.device) for lo, hi in bounds]

    def _build_norm_adj(self):
        """dataloader.py:218-234 through the native builder."""
        lib = _lib.load()
        N, E = self.n_user + self.m_item, len(self._r_indices)
        rowsum = np.empty(N, np.float32)
        _lib.check(lib.lgcn_adj_rowsum(self.n_user, self.m_item, _lib.npp(self._r_indptr),
                                       _lib.npp(self._r_indices), _lib.npp(self._r_vals), _lib.npp(rowsum)),
                   "lgcn_adj_rowsum")
        # d_inv exactly as the reference spells it (numpy's float32 power is not
        # correctly rounded, so it must be the same call: dataloader.py:231-232)
        with np.errstate(divide='ignore'):
            d_inv = np.power(rowsum, -0.5, where=rowsum != 0, out=np.zeros_like(rowsum))
        d_inv[np.isinf(d_inv)] = 0.
        indptr = np.empty(N + 1, np.int32)
        indices = np.empty(2 * E, np.int32)
        data = np.empty(2 * E, np.float32)
        _lib.check(lib.lgcn_build_norm_adj(self.n_user, self.m_item, _lib.npp(self._r_indptr),
                                           _lib.npp(self._r_indices), _lib.npp(self._r_vals), _lib.npp(d_inv),
                                           _lib.npp(indptr), _lib.npp(indices), _lib.npp(data)),
                   "lgcn_build_norm_adj")
        return sp.csr_matrix((data, indices, indptr), shape=(N, N))

    def getSparseGraphCSR(self):
        """NEW: A_hat as scipy CSR (int32 indices/indptr, fp32 data, sorted rows) -- what the
        HIP kernels consume; loaded from / saved to the reference's s_pre_adj_mat.npz."""
        if self._norm_adj is not None:
            return self._norm_adj
        pre_adj_path = join(self.path, 's_pre_adj_mat.npz') if self.path else None
        try:
            if pre_adj_path is None:
                raise FileNotFoundError
            norm_adj = sp.load_npz(pre_adj_path).tocsr()
            norm_adj.sort_indices()
            print("successfully loaded...")
        except Exception:
            print("generating adjacency matrix")
            s = time()
            norm_adj = self._build_norm_adj()
            print(f"costing {time() - s:.2f}s, saved norm_mat...")
            try:
                if pre_adj_path is not None:
                    sp.save_npz(pre_adj_path, norm_adj)
            except OSError as e:                     # read-only dataset dir
                warnings.warn(f"could not cache {pre_adj_path}: {e}")
        if norm_adj.indices.dtype != np.int32 or norm_adj.indptr.dtype != np.int32:
            norm_adj = sp.csr_matrix((norm_adj.data.astype(np.float32), norm_adj.indices.astype(np.int32),
                                      norm_adj.indptr.astype(np.int32)), shape=norm_adj.shape)
        if norm_adj.data.dtype != np.float32:
            norm_adj = norm_adj.astype(np.float32)
        self._norm_adj = norm_adj
        return norm_adj

    def getSparseGraph(self):
        """dataloader.py:203-246: coalesced torch sparse COO (or list of folds)."""
        print("loading adjacency matrix")
        if self.Graph is not None:
            return self.Graph
        norm_adj = self.getSparseGraphCSR()
        if self.split:
            self.Graph = self._split_A_hat(norm_adj)
            print("done split matrix")
        else:
            self.Graph = self._convert_sp_mat_to_sp_tensor(norm_adj).coalesce().to(world.device)
            print("don't split the matrix")
        return self.Graph

    def __getitem__(self, idx):
        return self.trainUniqueUsers[idx]

    def __len__(self):
        return len(self.trainUniqueUsers)


class CsrLoader(Loader):
    """NEW: the same dataset object built straight from the user-item CSR (int64 indptr
    [n_users+1], int32 indices sorted + unique per row) instead of train.txt/test.txt -- what the
    synthetic workloads (synthetic.py) and a caller that already holds the interactions in memory
    use.  Nothing is read from or written to disk unless `path` is given."""

    def __init__(self, indptr, indices, m_items, test_dict=None, config=world.config, path=None):
        self.path = path
        self.split = config.get('A_split', False)
        self.folds = config.get('A_n_fold', 100)
        self._r_indptr = np.ascontiguousarray(indptr, np.int64)
        self._r_indices = np.ascontiguousarray(indices, np.int32)
        if self._r_indptr[0] != 0 or self._r_indptr[-1] != len(self._r_indices) or np.any(np.diff(self._r_indptr) < 0):
            raise ValueError("CsrLoader: indptr does not describe indices")
        if len(self._r_indices) and (self._r_indices.min() < 0 or self._r_indices.max() >= m_items):
            raise ValueError("CsrLoader: item id out of range")
        self._r_vals = np.ones(len(self._r_indices), np.float32)
        self.n_user, self.m_item = len(self._r_indptr) - 1, int(m_items)
        self.traindataSize = len(self._r_indices)
        self.trainUniqueUsers = np.flatnonzero(np.diff(self._r_indptr) > 0).astype(np.int64)
        self.trainUser = self.trainItem = None            # the COO lists are not materialised
        td = dict(test_dict or {})
        self.testUser = np.fromiter((u for u, its in td.items() for _ in its), dtype=np.int64)
        self.testItem = np.fromiter((i for its in td.values() for i in its), dtype=np.int64)
        self.testUniqueUsers = np.fromiter(td.keys(), dtype=np.int64, count=len(td))
        self.testDataSize = len(self.testUser)
        self._finish_init()
