"""Scale test (-m gpu; ~60 GB of HBM, well under a minute): a 9 M-row graph at dim 256 puts element
offsets beyond 2^31 (row*d = 2.3e9), i.e. the regime of BASELINE configs[4] (10 M users x 1 M items,
dim 256).  Checked against the CPU oracle on a sample of rows (every long / split row, the first and
the last row, 4 096 random ones -- the oracle runs on the sub-problem those rows touch), against
torch's own sparse kernels on the whole matrix (an independent implementation on the same device)
and through size-independent properties."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_rows(oracle, A, X, rows):
    """oracle.spmm restricted to `rows`: the sub-matrix of those rows with its columns relabelled
    compactly, times the gathered slice of X (same arithmetic, same order within a row)."""
    sub = A[rows]
    cols, inv = np.unique(sub.indices, return_inverse=True)
    xs = X[torch.from_numpy(cols).to(X.device).long()].cpu().numpy()
    return oracle.spmm(sub.indptr.astype(np.int32), inv.astype(np.int32), sub.data, xs)


@pytest.mark.skipif(os.environ.get("LGCN_SKIP_LARGE") == "1", reason="LGCN_SKIP_LARGE=1")
def test_nine_million_rows_dim256(pkg, oracle):
    import ctypes as C
    import scipy.sparse as sp
    L = pkg._lib
    n_users, m_items, d, K, B = 8_000_000, 1_000_000, 256, 2, 2048
    N = n_users + m_items
    rng = np.random.Generator(np.random.PCG64(0))
    deg = 1 + rng.poisson(4.0, n_users)
    deg[:3] = [3000, 700, 65]                                     # long / split rows
    tu = np.repeat(np.arange(n_users, dtype=np.int64), deg)
    ti = rng.integers(0, m_items, len(tu))
    ti[:50] = m_items - 1                                          # make the very last row long-ish
    R = sp.csr_matrix((np.ones(len(tu), np.float32), (tu, ti)), shape=(n_users, m_items))
    R.sum_duplicates()
    A = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)
    A.sort_indices()
    A.data = rng.uniform(0.01, 0.3, A.nnz).astype(np.float32)
    assert A.shape == (N, N) and N * d > 2**31
    ip, ix, vv = (torch.from_numpy(x).to(DEV) for x in (A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data))
    g = L.Graph(ip, ix, vv, d_max=d)
    gen = torch.Generator(device=DEV); gen.manual_seed(1)
    X = torch.randn(N, d, device=DEV, generator=gen) * 0.1
    Y = g.spmm(X)
    sample = np.unique(np.concatenate([np.flatnonzero(np.diff(A.indptr) > 64)[:64], [0, 1, 2, N - 1],
                                       rng.integers(0, N, 4096)]))
    ref_rows = _oracle_rows(oracle, A, X, sample)
    got_rows = Y[torch.from_numpy(sample).to(DEV)].cpu().numpy()
    # (the sample holds the 3000- and 700-term rows: partial sums reach ~1, so 5e-6 absolute)
    np.testing.assert_allclose(got_rows, ref_rows, rtol=2e-5, atol=5e-6)
    At = torch.sparse_csr_tensor(ip.long(), ix.long(), vv, size=(N, N), device=DEV)
    Yref = torch.sparse.mm(At, X)
    err = float((Y - Yref).abs().max()); scale = float(Yref.abs().max())
    assert err < 2e-5 * max(1.0, scale), (err, scale)
    assert torch.equal(Y[-1] != 0, Yref[-1] != 0)                  # last row: offset (N-1)*d*4 B = 9.2 GB
    del Yref
    # propagate_mean (K = 2) = (X + AX + A(AX)) / 3
    work = torch.empty(K - 1, N, d, device=DEV)
    out = torch.empty(N, d, device=DEV)
    L.check(L.load().lgcn_propagate_mean(g.handle, L.tp(X), K, d, 0, L.tp(work), L.tp(out), L.current_stream()), "mean")
    ref = (X + Y + torch.sparse.mm(At, Y)) / 3.0
    assert float((out - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    del At, Y, work
    # one fused training step on the 9.2 GB table
    E0 = X.clone()
    st = {k: torch.zeros(N, d, device=DEV) for k in ("m", "v")}
    act = torch.zeros(1, N, d, device=DEV)
    G64 = torch.zeros(N, d, dtype=torch.int64, device=DEV)
    bitmap = torch.zeros(2 * ((N + 31) // 32), dtype=torch.int32, device=DEV)
    terms = torch.zeros(2 * B, device=DEV); ebuf = torch.zeros(3 * B * d, device=DEV); errf = torch.zeros(1, dtype=torch.int32, device=DEV)
    cfg = L.TrainConfig()
    cfg.graph = g.handle; cfg.n_users, cfg.d, cfg.K, cfg.act_dtype = n_users, d, K, 0
    cfg.E0, cfg.adam_m, cfg.adam_v = E0.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr()
    cfg.act, cfg.G64, cfg.bitmap, cfg.terms, cfg.ebuf = act.data_ptr(), G64.data_ptr(), bitmap.data_ptr(), terms.data_ptr(), ebuf.data_ptr()
    cfg.contrib, cfg.err, cfg.max_batch, cfg.decay = None, errf.data_ptr(), B, 1e-4
    cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.xcd_remap = 1e-3, 0.9, 0.999, 1e-8, 1
    h = C.c_void_p()
    L.check(L.load().lgcn_ctx_create(C.byref(cfg), C.byref(h)), "ctx")
    u = torch.randint(0, n_users, (B,), device=DEV, dtype=torch.int32); u[0] = n_users - 1
    p = torch.randint(0, m_items, (B,), device=DEV, dtype=torch.int32); p[0] = m_items - 1
    n = torch.randint(0, m_items, (B,), device=DEV, dtype=torch.int32)
    loss = torch.empty(3, device=DEV)
    L.check(L.load().lgcn_train_step(h, L.tp(u), L.tp(p), L.tp(n), B, L.tp(loss), L.current_stream()), "step")
    torch.cuda.synchronize()
    assert L.load().lgcn_ctx_check(h, L.current_stream()) == 0
    # loss from the (pre-step) propagated rows, computed by torch
    eu, ep, en = ref[u.long()], ref[n_users + p.long()], ref[n_users + n.long()]
    x = (eu * ep).sum(1) - (eu * en).sum(1)
    bpr = -torch.nn.functional.logsigmoid(x).mean()
    reg = 0.5 * (eu.pow(2).sum() + ep.pow(2).sum() + en.pow(2).sum()) / B
    assert abs(float(loss[1]) - float(bpr)) < 2e-5 and abs(float(loss[2]) - float(reg)) < 2e-5 * max(1.0, float(reg))
    delta = (E0 - X).abs()
    assert torch.isfinite(E0).all() and float(delta.max()) <= 1.001e-3          # Adam's first step moves <= lr
    assert float(delta[-1].max()) > 0 and int((delta.sum(1) > 0).sum()) > 3 * B  # last row (positive of triplet 0) moved
    assert int(G64.abs().sum()) == 0
    L.load().lgcn_ctx_destroy(h)
    g.close()
