"""Processing order of the graph rows for the SpMM kernels (NEW; no reference counterpart).

The order never changes results or memory layout -- it only decides which rows the
hardware works on together.  Each XCD of an MI355X owns a 4 MiB L2 and gets a contiguous
slice of the order (csrc: tile_of_block), so an order in which neighbouring rows share
neighbours turns most of the 256-byte row gathers into L2 hits (measured on Gowalla:
L2 hit rate 0.56 -> ~0.8 of gather requests, the rest go to the Infinity Cache / HBM).

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


def row_order(method, dataset, adj, cache_dir=None):
    """-> int32 permutation of 0..N-1 (or None for the natural order)."""
    n_users, m_items = dataset.n_users, dataset.m_items
    N = n_users + m_items
    if method in (None, 'natural', 'none'):
        return None
    cache = os.path.join(cache_dir, f"s_row_order_{method}.npy") if cache_dir else None
    if cache and os.path.exists(cache):
        try:
            o = np.load(cache)
            if o.shape == (N,) and np.array_equal(np.sort(o), np.arange(N)):
                return np.ascontiguousarray(o, dtype=np.int32)
        except Exception:
            pass
    try:
        if method == 'rcm':
            o = rcm(adj)
        elif method == 'cocluster':
            o = cocluster(dataset.UserItemNet, n_users, m_items)
        else:
            raise ValueError(f"unknown row order '{method}'")
        if o.shape != (N,) or not np.array_equal(np.sort(o), np.arange(N)):
            raise RuntimeError("ordering is not a permutation")
    except ValueError:
        raise
    except Exception as e:                 # scikit-learn missing, ARPACK not converging, ...
        warnings.warn(f"row order '{method}' unavailable ({type(e).__name__}: {e}); falling back to rcm")
        o = rcm(adj)
    assert o.shape == (N,) and np.array_equal(np.sort(o), np.arange(N))
    if cache:
        try:
            np.save(cache, o)
        except OSError:
            pass
    return o
