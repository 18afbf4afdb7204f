#!/usr/bin/env python3
"""CPU prototype (numpy): numpy-legacy shuffle = MT19937 + rejection-sampled Fisher-Yates, restated in the parallel form a device version would take
(DESIGN.md 8.5): sort the steps by target (j_i, i); a step parent = the next step that writes its position; pointer jumping to the chain roots.
Checks the result against np.random.RandomState.shuffle for several n and seeds."""
import numpy as np

def raw_stream(rs, count):
    st = rs.get_state()
    bg = np.random.MT19937()
    bg.state = {'bit_generator': 'MT19937', 'state': {'key': st[1], 'pos': st[2]}}
    return bg.random_raw(count).astype(np.uint64)

def mask_of(i):
    m = i
    for s in (1, 2, 4, 8, 16, 32): m |= m >> s
    return m

def align(R, n):
    """sequential reference of the acceptance alignment: j[i] for i = n-1..1, draws consumed"""
    j = np.zeros(n, np.int64); q = 0
    for i in range(n - 1, 0, -1):
        m = mask_of(i)
        while True:
            v = int(R[q]) & m; q += 1
            if v <= i: break
        j[i] = v
    return j, q

def perm_from_j(j, n):
    """parallel formulation: sort (j_i, i), parent = next occurrence, pointer jumping"""
    idx = np.arange(1, n)                       # steps
    key = j[1:]
    order = np.lexsort((idx, key))             # by key, then by i ascending
    ks, vs = key[order], idx[order]
    start = np.searchsorted(ks, np.arange(n + 1), side='left')   # group q = [start[q], start[q+1])
    def nxt(q, t):
        """smallest step i > t with j_i == q, or -1 (vectorised)"""
        lo, hi = start[q], start[q + 1]
        # upper_bound of t inside vs[lo:hi]: use a global searchsorted on composite key
        comp = ks.astype(np.int64) * (n + 1) + vs
        pos = np.searchsorted(comp, q.astype(np.int64) * (n + 1) + t, side='right')
        ok = pos < hi
        return np.where(ok, vs[np.minimum(pos, len(vs) - 1)], -1)
    nodes = np.arange(n)
    parent = nxt(nodes, nodes)                  # parent(i) = next(i, i)
    root = nodes.copy()
    p = parent.copy()
    # pointer jumping: root[i] = root of chain
    cur = np.where(p >= 0, p, nodes)
    for _ in range(64):
        nx = np.where(parent[cur] >= 0, parent[cur], cur)
        if np.array_equal(nx, cur): break
        cur = nx
    root = cur                                   # root(i) for chain starting AT node i (i itself if no parent)
    out = np.empty(n, np.int64)
    jj = j.copy(); jj[0] = 0
    first = nxt(jj, nodes)                       # first hop from (q = j_p, t = p); p = 0: q = 0, t = 0
    out = np.where(first >= 0, root[np.maximum(first, 0)], jj)
    return out

for n in (1, 2, 3, 7, 64, 1000, 5000, 70000):
    for seed in (0, 2020, 7):
        rs = np.random.RandomState(seed)
        R = raw_stream(rs, 3 * n + 100)
        want = np.arange(n); rs.shuffle(want)
        j, used = align(R, n)
        got = perm_from_j(j, n) if n > 1 else np.arange(n)
        assert np.array_equal(got, want), (n, seed, got[:10], want[:10])
        # stream position afterwards
        rs2 = np.random.RandomState(seed); _ = raw_stream(rs2, 1)
    print("n", n, "ok; draws per element", used / max(1, n - 1))
