"""Processing order of the graph rows for the SpMM kernels (NEW; no reference counterpart).

The order never changes results or memory layout -- it only decides which rows the
hardware works on together.  Each XCD of an MI355X owns a 4 MiB L2 and gets a contiguous
slice of the order (csrc: tile_of_block), so an order in which neighbouring rows share
neighbours turns most of the 256-byte row gathers into L2 hits (measured on Gowalla:
L2 hit rate 0.56 -> ~0.8 of gather requests, the rest go to the Infinity Cache / HBM).

    xcd        recursive spectral bisection balanced by non-zeros: the first three levels are the
               8 XCD slices, deeper levels the order inside a slice (a few seconds on Gowalla)
    natural    rows 0..N-1 (users, then items)
    rcm        reverse Cuthill-McKee of A (scipy), cheap (20 ms on Gowalla)
    cocluster  spectral co-clustering of the user-item matrix (scikit-learn), user cluster k
               followed by item cluster k, k = 32; a few seconds on Gowalla; cached on disk
"""
import os
import warnings

import numpy as np


def natural(n_users, m_items):
    return np.arange(n_users + m_items, dtype=np.int32)


def rcm(adj):
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    return np.ascontiguousarray(reverse_cuthill_mckee(adj.tocsr(), symmetric_mode=True), dtype=np.int32)


def cocluster(R, n_users, m_items, k=32, seed=0):
    """R: scipy CSR user-item matrix."""
    from sklearn.cluster import SpectralCoclustering
    deg_u = np.asarray(R.sum(axis=1)).ravel()
    deg_i = np.asarray(R.sum(axis=0)).ravel()
    ku, ki = np.flatnonzero(deg_u > 0), np.flatnonzero(deg_i > 0)        # isolated nodes break the SVD scaling
    k = int(max(2, min(k, len(ku) // 16, len(ki) // 16)))
    m = SpectralCoclustering(n_clusters=k, random_state=seed, svd_method='arpack').fit(R[ku][:, ki])
    lab_u = np.full(n_users, k, np.int64); lab_u[ku] = m.row_labels_
    lab_i = np.full(m_items, k, np.int64); lab_i[ki] = m.column_labels_
    parts = []
    for c in range(k + 1):
        parts.append(np.flatnonzero(lab_u == c))
        parts.append(n_users + np.flatnonzero(lab_i == c))
    return np.concatenate(parts).astype(np.int32)


def _fiedler(sub, seed=0):
    """Second singular pair of D_u^-1/2 sub D_i^-1/2 -> coordinates (users, items) on the line that
    the spectral relaxation of the normalised cut puts the nodes on.  None if it cannot be had."""
    from scipy.sparse.linalg import svds
    import scipy.sparse as sp
    du = np.asarray(sub.sum(axis=1)).ravel()
    di = np.asarray(sub.sum(axis=0)).ravel()
    ku, ki = np.flatnonzero(du > 0), np.flatnonzero(di > 0)
    if len(ku) < 8 or len(ki) < 8:
        return None
    core = sub[ku][:, ki]
    su, si = 1.0 / np.sqrt(du[ku]), 1.0 / np.sqrt(di[ki])
    M = sp.diags(su) @ core @ sp.diags(si)
    rng = np.random.RandomState(seed)
    try:
        u, s_, vt = svds(M.astype(np.float64), k=2, v0=rng.rand(min(M.shape)), tol=1e-3, maxiter=300)
    except Exception:
        return None
    j = int(np.argmin(s_))                        # the smaller of the two largest = the second one
    fu = np.zeros(sub.shape[0]); fi = np.zeros(sub.shape[1])
    fu[ku] = u[:, j] * su
    fi[ki] = vt[j] * si
    return fu, fi


def _row_cost():
    """Work of a row in units of one non-zero: cost = nnz + ROW_COST.  Measured on Gowalla (slices balanced
    by non-zeros alone differ 1.9x in their number of short rows): 0 / 8 / 16 give the same launch time,
    32 and 64 are slower, and on the heavy-tailed synthetic shapes 16 costs 15-20 %: the default is 0."""
    return float(os.environ.get("LGCN_ROW_COST", "0"))


def xcd_order(R, n_users, m_items, parts=8, leaf_nnz=4096, seed=0):
    """Recursive spectral bisection of the bipartite graph, balanced by work (row nnz of A_hat).
    The first log2(parts) levels give the XCD partition (each XCD's rows mostly gather rows of its
    own part: the part of the table an XCD touches fits its 4 MiB L2); deeper levels give the
    order inside a part (rows close in the order share neighbours).
    -> (order int32 [N], xcd_start int64 [parts+1] positions in the order)."""
    R = R.tocsr()
    Rt = R.T.tocsr()
    wu = np.diff(R.indptr).astype(np.float64) + _row_cost()          # work of a user row / an item row
    wi = np.diff(Rt.indptr).astype(np.float64) + _row_cost()
    top_levels = int(np.log2(parts))
    out, bounds = [], []

    def emit(U, I, cu=None, ci=None):
        if cu is not None:                              # interleave along the last coordinate
            ids = np.concatenate([U, n_users + I]); co = np.concatenate([cu, ci])
            out.append(ids[np.argsort(co, kind="stable")])
        else:
            out.append(np.concatenate([U, n_users + I]))

    def rec(U, I, depth, cu=None, ci=None):
        w = wu[U].sum() + wi[I].sum()
        if depth >= top_levels and (w <= leaf_nnz or len(U) + len(I) <= 32 or depth > 24):
            emit(U, I, cu, ci)
            return
        f = _fiedler(R[U][:, I], seed + depth) if (len(U) and len(I)) else None
        if f is None:
            if depth >= top_levels:
                emit(U, I, cu, ci)
                return
            fu, fi = np.arange(len(U), dtype=np.float64), np.arange(len(I), dtype=np.float64)   # arbitrary split
            if len(I):
                fi = fi * (max(len(U), 1) / len(I))
        else:
            fu, fi = f
        co = np.concatenate([fu, fi]); ww = np.concatenate([wu[U], wi[I]])
        o = np.argsort(co, kind="stable")
        cw = np.cumsum(ww[o])
        cut = int(np.searchsorted(cw, cw[-1] / 2.0))
        left = np.zeros(len(co), bool); left[o[:cut + 1]] = True
        if left.all() or not left.any():
            left[:] = False; left[o[:len(o) // 2]] = True
        lu, li = left[:len(U)], left[len(U):]
        rec(U[lu], I[li], depth + 1, fu[lu], fi[li])
        if depth + 1 == top_levels:
            bounds.append(sum(len(x) for x in out))
        rec(U[~lu], I[~li], depth + 1, fu[~lu], fi[~li])
        if depth + 1 == top_levels:
            bounds.append(sum(len(x) for x in out))

    rec(np.arange(n_users), np.arange(m_items), 0)
    order = np.concatenate(out).astype(np.int32)
    xs = np.array([0] + bounds, np.int64)
    assert len(xs) == parts + 1 and xs[-1] == n_users + m_items
    return order, xs


def row_order(method, dataset, adj, cache_dir=None):
    """-> (order, xcd_start): int32 permutation of 0..N-1 (None = natural order) and the int64[9]
    cut of that order into the 8 XCD slices (None = let the library balance the slices by work)."""
    n_users, m_items = dataset.n_users, dataset.m_items
    N = n_users + m_items
    if method in (None, 'natural', 'none'):
        return None, None
    tag = method if method != 'xcd' else f"xcd{_row_cost():g}"
    cache = os.path.join(cache_dir, f"s_row_order_{tag}.npz") if cache_dir else None
    if cache and os.path.exists(cache):
        try:
            z = np.load(cache)
            o = z["order"]
            xs = z["xcd_start"] if "xcd_start" in z.files else None
            if o.shape == (N,) and np.array_equal(np.sort(o), np.arange(N)) and (xs is None or xs.shape == (9,)):
                return np.ascontiguousarray(o, dtype=np.int32), (None if xs is None else np.ascontiguousarray(xs, np.int64))
        except Exception:
            pass
    xs = None
    try:
        if method == 'rcm':
            o = rcm(adj)
        elif method == 'cocluster':
            o = cocluster(dataset.UserItemNet, n_users, m_items)
        elif method == 'xcd':
            o, xs = xcd_order(dataset.UserItemNet, n_users, m_items)
        else:
            raise ValueError(f"unknown row order '{method}'")
        if o.shape != (N,) or not np.array_equal(np.sort(o), np.arange(N)):
            raise RuntimeError("ordering is not a permutation")
    except ValueError:
        raise
    except Exception as e:                 # scikit-learn missing, ARPACK not converging, ...
        warnings.warn(f"row order '{method}' unavailable ({type(e).__name__}: {e}); falling back to rcm")
        o, xs = rcm(adj), None
    assert o.shape == (N,) and np.array_equal(np.sort(o), np.arange(N))
    if cache:
        try:
            if xs is None:
                np.savez(cache, order=o)
            else:
                np.savez(cache, order=o, xcd_start=xs)
        except OSError:
            pass
    return o, xs
