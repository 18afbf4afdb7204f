"""Seeded synthetic bipartite interaction graphs of a given shape (NEW).

The reference's Yelp2018 / Amazon-Book train splits are absent from its checkout and the
10M x 1M graph of BASELINE.json configs[4] never existed as a file, so those workloads run
on generated graphs: user degrees ~ Zipf(1.8) clipped to [1, m_items/4] and rescaled to hit
E exactly, items drawn by power-law popularity (rank^-0.9, ranks permuted) WITHOUT
replacement per user, every user >= 1 item (so the native sampler is defined).

Everything is vectorised (torch ops; runs on the GPU when one is given, else on the CPU)
and hands back CSR arrays directly -- no text file, no per-user Python loop -- so the
200 M-edge graph is built in seconds on an MI355X.

Reproducibility: the degree sequence and the popularity ranks come from numpy PCG64(seed) and are
the same everywhere; the item draws use torch's generator OF THE GIVEN DEVICE, so a seed names one
graph per device type (CPU vs GPU), not one graph overall.  Every measurement and parity test in
this repository builds and checks its graph in the same process, on the same device.
"""
import numpy as np
import torch


def _degrees(n_users, m_items, E, rng):
    cap = max(1, m_items // 4)
    raw = np.clip(rng.zipf(1.8, n_users).astype(np.float64), 1, cap)
    deg = np.minimum(np.maximum(1, np.floor(raw * (E / raw.sum()))).astype(np.int64), cap)
    order = rng.permutation(n_users)
    diff = int(E - deg.sum())
    while diff != 0:                                  # a few vectorised rounds
        if diff > 0:
            ok = order[deg[order] < cap]
            if len(ok) == 0:
                raise ValueError("E is larger than n_users * (m_items // 4)")
            take = ok[:diff]
            deg[take] += 1
            diff -= len(take)
        else:
            ok = order[deg[order] > 1]
            if len(ok) == 0:
                raise ValueError("E is smaller than n_users")
            take = ok[:-diff]
            deg[take] -= 1
            diff += len(take)
    return deg


def power_law_bipartite(n_users, m_items, E, seed=2020, device=None):
    """-> (indptr int64 [n_users+1], indices int32 [E]) numpy arrays: CSR of the user-item
    matrix, columns sorted ascending and unique per row."""
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    device = torch.device(device)
    rng = np.random.Generator(np.random.PCG64(seed))
    deg_np = _degrees(n_users, m_items, E, rng)
    pop = 1.0 / np.arange(1, m_items + 1, dtype=np.float64) ** 0.9
    pop = pop[rng.permutation(m_items)]
    cdf = torch.from_numpy(np.cumsum(pop / pop.sum())).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))

    need = torch.from_numpy(deg_np).to(device)
    have = torch.empty(0, dtype=torch.int64, device=device)       # sorted keys user * m_items + item
    over = 1.3
    for _ in range(200):
        act = torch.nonzero(need > 0).flatten()
        if act.numel() == 0:
            break
        cnt = (need[act].double() * over).ceil().long() + 4
        total = int(cnt.sum())
        users = torch.repeat_interleave(act, cnt)
        items = torch.searchsorted(cdf, torch.rand(total, generator=gen, device=device, dtype=torch.float64))
        key = users * m_items + items.clamp_(max=m_items - 1)
        del items
        if have.numel():                                           # drop what the user already has
            pos = torch.searchsorted(have, key).clamp_(max=have.numel() - 1)
            keep = have[pos] != key
            del pos
            key, users = key[keep], users[keep]
            del keep
        # first occurrence of every key in draw order (stable sort keeps draw order among equals) ...
        skey, perm = torch.sort(key, stable=True)
        first = torch.ones_like(skey, dtype=torch.bool)
        first[1:] = skey[1:] != skey[:-1]
        draw = torch.sort(perm[first]).values                      # ... back in draw order (grouped by user)
        del skey, perm, first
        key, users = key[draw], users[draw]
        del draw
        # rank inside the user's run; accept until the user's quota is full
        start = torch.ones_like(users, dtype=torch.bool)
        start[1:] = users[1:] != users[:-1]
        idx = torch.arange(users.numel(), device=device)
        run_start = torch.cummax(torch.where(start, idx, torch.zeros_like(idx)), 0).values
        ok = (idx - run_start) < need[users]
        del start, idx, run_start
        key, users = key[ok], users[ok]
        need = need - torch.bincount(users, minlength=n_users)
        have = torch.sort(torch.cat([have, key])).values
        del key, users, ok
        over = min(over * 1.6, 16.0)
    else:
        raise RuntimeError("synthetic generator did not converge")
    assert have.numel() == E
    indices = (have % m_items).to(torch.int32).cpu().numpy()
    indptr = np.zeros(n_users + 1, np.int64)
    np.cumsum(deg_np, out=indptr[1:])
    return indptr, indices
