"""fp8 activation storage (--act_dtype fp8, LGCN_FP8; -m gpu): OCP E4M3 rows with one power-of-two fp32 scale per row, fp32
accumulation; parameters, Adam state and the gradient scatter stay fp32 / fixed point.  NEW (no counterpart in the reference,
which computes in fp32): a separately measured storage mode like bf16, so it is checked the way bf16 is --
  * the quantiser against a restatement with torch's own float8_e4m3fn cast, BIT FOR BIT (bytes and scales);
  * the SpMM kernels on an fp8 table against the oracle run on the DECODED table (same numbers in, fp32 arithmetic: tight), and
    their fp8 output against the fp32 result within half an fp8 ulp;
  * the propagation and the fused step of the model against that same layer-by-layer restatement / the fp32 oracle (loose);
  * run-to-run and data-parallel invariance, which are bitwise whatever the storage type.
The Recall@20 delta it costs is measured by tools/gowalla_trajectory.py --act_dtype fp8 (profiles/r04)."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import assert_rows_close, spmm_sum_bound
from test_gpu_parity import _dev, _make_model, _random_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def quantise_rows(x):
    """The library's fp8 table of an fp32 [n, d] array, restated: scale = 2^(e - 6) for max|row| = 1.f * 2^e (so max / scale lies in
    [64, 128)), rows below 2^-100 -> zeros with scale 1; element = RNE(x / scale) in OCP E4M3 (torch.float8_e4m3fn).
    -> (bytes uint8 [n, d], scales float32 [n])"""
    x = np.ascontiguousarray(x, np.float32)
    amax = np.abs(x).max(axis=1)
    E = (amax.view(np.uint32) >> 23).astype(np.int64)
    ok = (E >= 27) & (E < 255)
    scale = np.where(ok, np.ldexp(1.0, (E - 6 - 127).clip(-126, 127)), 1.0).astype(np.float32)
    inv = np.where(ok, np.ldexp(1.0, (127 + 6 - E).clip(-126, 127)), 0.0).astype(np.float32)
    q = torch.from_numpy(x * inv[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).numpy(), scale


def decode(bytes_, scale):
    return torch.from_numpy(bytes_).view(torch.float8_e4m3fn).float().numpy() * scale[:, None]


def table_parts(pkg, tab, n, d):
    """(bytes in COLUMN order [n, d], scales [n]) of a library fp8 table: the bytes of a row are stored chunk-interleaved
    (byte l*16 + 4j + e = column (j*d/16 + l)*4 + e: what makes the fp32 side of the kernels' epilogues coalesced)."""
    t = tab.cpu()
    raw = t[:n * d].view(n, d).numpy()
    nat = np.empty_like(raw)
    nat[:, pkg._lib.fp8_col_of_byte(d)] = raw
    return nat, t[n * d:n * d + 4 * n].view(torch.float32).numpy()


@pytest.mark.parametrize("d", [64, 128, 256])
def test_fp8_quantiser_bit_exact_vs_torch(pkg, d):
    L = pkg._lib
    rng = np.random.Generator(np.random.PCG64(d))
    n = 3001
    X = rng.normal(0, 0.1, (n, d)).astype(np.float32)
    X *= np.exp(rng.uniform(-30, 10, n)).astype(np.float32)[:, None]         # row magnitudes over 17 decades
    X[5] = 0.0                                                                # a zero row
    X[6] = 1e-35                                                              # below 2^-100: stored as zeros
    X[7, 3] = 7.0; X[7, 4:] *= 1e-6                                           # one outlier: the rest of the row is subnormal in fp8
    X[8] = np.float32(2.0) ** rng.integers(-20, 20, d)                        # exact powers of two
    X[9, :] = 0.0; X[9, 0] = -3.0e38                                          # near the top of fp32
    X[n - 1] = np.linspace(-1, 1, d, dtype=np.float32)                        # the last row
    tab = L.Graph.to_fp8(None, _dev(X))
    assert tab.numel() == int(L.load().lgcn_table_bytes(n, d, L.FP8)) and tab.numel() % 256 == 0
    got_b, got_s = table_parts(pkg, tab, n, d)
    want_b, want_s = quantise_rows(X)
    assert np.array_equal(got_s, want_s)
    assert np.array_equal(got_b, want_b), np.argwhere(got_b != want_b)[:5]
    assert not (got_b & 0x7f == 0x7f).any()                                   # no NaN code is ever produced
    # what the storage costs: relative error of an element <= 2^-4 (3 mantissa bits, RNE) above the subnormal range
    dec = decode(got_b, got_s)
    big = np.abs(X) >= np.abs(X).max(axis=1, keepdims=True) * 2.0 ** -12
    big[6] = False
    assert (np.abs(dec - X)[big] <= 2.0 ** -4 * np.abs(X)[big]).all()
    assert np.array_equal(pkg._lib.Graph.from_fp8(tab, n, d).cpu().numpy(), dec)
    order = pkg._lib.fp8_col_of_byte(d)
    assert sorted(order.tolist()) == list(range(d)) and order[0] == 0 and order[4] == (d // 16) * 4 and order[16] == 4


@pytest.mark.parametrize("d", [64, 128, 256])
def test_spmm_fp8_tables_vs_oracle(pkg, oracle, d):
    """A X with X an fp8 table: fp32 output against the oracle on the decoded table (identical inputs, fp32 arithmetic both
    sides: the usual 2e-5), fp8 output within half an fp8 ulp of that result (+ the subnormal step), scales as the restated
    quantiser gives them; an fp32 input with fp8 output (the first backward layer's shape) likewise.  Long rows take the split path."""
    L = pkg._lib
    rng = np.random.Generator(np.random.PCG64(100 + d))
    n = 3001
    indptr, indices, vals = _random_graph(rng, n, 9, heavy=3)
    X = (rng.normal(0, 0.1, (n, d)) * np.exp(rng.uniform(-8, 2, n))[:, None]).astype(np.float32)
    g = L.Graph(_dev(indptr.astype(np.int32)), _dev(indices.astype(np.int32)), _dev(vals.astype(np.float32)), d_max=d)
    xq = g.to_fp8(_dev(X))
    Xd = decode(*quantise_rows(X))
    ref = oracle.spmm(indptr, indices, vals, Xd)
    got32 = g.spmm_fp8(xq, d, L.F32).cpu().numpy()
    # two fp32 summation orders of the same terms: the computed per-element bound (2 n + 2) * 2^-24 * sum |v x|
    assert_rows_close(got32, ref, spmm_sum_bound(indptr, vals, indices, Xd), "fp8 table -> fp32")
    rowmax = np.abs(ref).max(axis=1, keepdims=True)
    tol = 2.0 ** -4 * np.abs(ref) + rowmax * 2.0 ** -15 + 1e-30

    def check_fp8_out(tab, what):
        b, s = table_parts(pkg, tab, n, d)
        wb, ws = quantise_rows(ref)
        assert (np.abs(decode(b, s) - ref) <= tol).all(), what
        same_scale = s == ws                       # (a row maximum within rounding of a power of two may land in the other binade)
        assert same_scale.mean() > 0.995, what
        assert (b[same_scale] == wb[same_scale]).mean() > 0.999, what
    check_fp8_out(g.spmm_fp8(xq, d, L.FP8), "fp8 -> fp8")
    ref = oracle.spmm(indptr, indices, vals, X)                       # fp32 table in, fp8 out
    rowmax = np.abs(ref).max(axis=1, keepdims=True)
    tol = 2.0 ** -4 * np.abs(ref) + rowmax * 2.0 ** -15 + 1e-30
    check_fp8_out(g.spmm(_dev(X), L.FP8), "fp32 -> fp8")
    g.close()


def test_fp8_refusals(pkg, tiny, tmp_path):
    L, lib = pkg._lib, pkg._lib.load()
    x = torch.zeros(8, 32, device=DEV)
    assert lib.lgcn_to_fp8(L.tp(x), L.tp(torch.zeros(2048, dtype=torch.uint8, device=DEV)), 8, 32, None) == 3      # d = 32
    assert b"64, 128 or 256" in lib.lgcn_last_error()
    g = L.Graph(torch.zeros(9, dtype=torch.int32, device=DEV), torch.zeros(0, dtype=torch.int32, device=DEV), torch.zeros(0, device=DEV), d_max=64)
    xb = torch.zeros(8, 64, device=DEV, dtype=torch.bfloat16)
    y = torch.zeros(int(lib.lgcn_table_bytes(8, 64, L.FP8)), dtype=torch.uint8, device=DEV)
    assert lib.lgcn_spmm_csr(g.handle, L.tp(xb), L.BF16, L.tp(y), L.FP8, 64, None) == 3                            # bf16 -> fp8 is not a launch
    g.close()


@pytest.mark.parametrize("which", ["tiny", "lastfm"])
def test_fp8_activation_mode_propagation_and_step(pkg, oracle, tiny, lastfm, tmp_path, which):
    """--act_dtype fp8 through the model: computer() against the layer-by-layer restatement (quantise E0, propagate with the oracle,
    quantise, ...; the last layer and the mean in fp32) and one fused stageOne against the fp32 oracle."""
    g = tiny if which == "tiny" else lastfm
    ds, m = _make_model(pkg, g, tmp_path, act_dtype="fp8")
    A = (g.z["adj_indptr"], g.z["adj_indices"], g.z["adj_data"])
    with torch.no_grad():
        au, ai = m.computer()
    got = torch.cat([au, ai]).cpu().numpy()
    layers = [g.e0()]
    x = decode(*quantise_rows(g.e0())) if g.K >= 2 else g.e0()
    for k in range(1, g.K + 1):
        y = oracle.spmm(*A, x)
        layers.append(y if k == g.K else decode(*quantise_rows(y)))
        x = layers[-1]
    want = sum(layers) / np.float32(g.K + 1)
    err = np.abs(got - want)
    scale = np.abs(want).max()
    # a sum that differs in its last bit between the two summation orders can land on the other side of an fp8 rounding
    # boundary (one element in ~1e5): everything within 2 % of the table's scale, all but such elements tight
    assert err.max() <= 2e-2 * scale and (err <= 2e-5 * np.abs(want) + 1e-7 * scale).mean() > 0.999, (err.max(), scale)
    ref32 = oracle.propagate(*A, g.e0(), g.K)
    assert np.abs(got - ref32).max() <= 0.1 * np.abs(ref32).max()            # and it still is the propagation (storage error only)
    # one fused step: loss against the fp32 oracle's (the step reads rows quantised to 3 mantissa bits: 1e-2), Adam's first
    # step bounded by lr, workspace clean, and the same step on a fresh model bit for bit
    tr = oracle.Trainer(g.n_users, *A, g.e0(), g.K, g.meta["decay"], g.meta["lr"])
    u, p, n = g.z["b_users"], g.z["b_pos"], g.z["b_neg"]
    l_ref = tr.stageOne(u, p, n)
    bpr = pkg.utils.BPRLoss(m, pkg.world.config)
    l_got = bpr.stageOne(_dev(u), _dev(p), _dev(n))
    assert abs(l_got - l_ref) < 1e-2 * max(1.0, abs(l_ref)), (l_got, l_ref)
    P = m._table.detach().cpu().numpy()
    assert np.isfinite(P).all() and np.abs(P - g.e0()).max() <= 1.001 * g.meta["lr"]
    moved = (np.abs(P - g.e0()).max(axis=1) > 0).mean()
    assert moved > 0.5                                                        # the gradient reached most rows through the fp8 layers
    # direction of the step against the fp32 oracle's: Adam's first step is lr * sign(g) where |g| >> eps
    big = np.abs(tr.e0 - g.e0()) > 0.9 * g.meta["lr"]
    assert (np.sign(P - g.e0())[big] == np.sign(tr.e0 - g.e0())[big]).mean() > 0.97
    assert int(m._dev['G64'].abs().sum()) == 0
    m.check_device_errors()
    ds2, m2 = _make_model(pkg, g, tmp_path, act_dtype="fp8")
    l2 = pkg.utils.BPRLoss(m2, pkg.world.config).stageOne(_dev(u), _dev(p), _dev(n))
    assert l2 == l_got and torch.equal(m2._table, m._table)


@pytest.mark.parametrize("mode", ["rows", "dense", "row_sharded"])
@pytest.mark.parametrize("world", [2, 3])
def test_fp8_dp_epoch_loopback_bitwise(pkg, tiny, tmp_path, world, mode):
    """fp8 activation storage under data parallelism (the C loop through the loopback communicator): replicas quantise the same
    numbers the same way, the gradient exchange is fixed point -- every rank ends bit for bit where the single-GPU epoch ends.
    Row-sharded propagation: every owner quantises its own rows (a row's scale depends on that row alone) and broadcasts the row
    bytes AND the row scales; layer 1 gathers fp8(E0), requantised after every exchange of the owners' Adam rows.
    Then: one rank fails before its first collective (its context was made for a smaller batch) while the other is already waiting
    in the all-gather -- the failing rank's exit must release it (loopback abort) instead of leaving it there."""
    import threading
    g = tiny
    rng = np.random.Generator(np.random.PCG64(29 * world + len(mode)))
    B = 48
    T = 3 * B + 1
    u = rng.integers(0, g.n_users, T); p = rng.integers(0, g.m_items, T); n = rng.integers(0, g.m_items, T)
    U, P, Nn = (_dev(x, torch.int32) for x in (u, p, n))
    ds, ref = _make_model(pkg, g, tmp_path, act_dtype="fp8", B=B)
    want_loss = ref.fused_epoch(U, P, Nn, B).cpu().numpy()
    want = ref._table.cpu().numpy().view(np.uint32)
    L, lib = pkg._lib, pkg._lib.load()
    models = [_make_model(pkg, g, tmp_path, act_dtype="fp8", B=B)[1] for _ in range(world)]
    par = pkg.parallel
    ranges = par.row_ranges(models[0]._adj.indptr, models[0].n_users, world) if mode == "row_sharded" else None
    rr = np.ascontiguousarray(ranges, np.int64) if ranges is not None else None
    states = [mm._state(max_batch=B, need_ctx=True, dp_world=world,
                        row_subset=par.owned_rows(ranges, r) if ranges is not None else None) for r, mm in enumerate(models)]
    comms = (C.c_void_p * world)()
    L.check(lib.lgcn_dp_init_loopback(world, comms), "loopback")
    code = {"rows": 0, "dense": 1, "row_sharded": 2}[mode]
    steps = (T + B - 1) // B
    streams = [torch.cuda.Stream() for _ in range(world)]
    gathered = [torch.empty(world * par.block_numel(B, world, g.d), device=DEV) for _ in range(world)]
    losses = [torch.empty(steps, 3, device=DEV) for _ in range(world)]
    torch.cuda.synchronize()
    rcs, errs = [None] * world, [None] * world

    def rank_main(r):
        rcs[r] = lib.lgcn_train_epoch_dp(states[r]['ctx'], comms[r], L.tp(U), L.tp(P), L.tp(Nn), T, B, code, L.npp(rr) if rr is not None else None,
                                         L.tp(gathered[r]), L.tp(losses[r]), C.c_void_p(streams[r].cuda_stream))
        if rcs[r]:
            errs[r] = lib.lgcn_last_error()
    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads)
    torch.cuda.synchronize()
    assert rcs == [0] * world, (rcs, errs)
    for r, mm in enumerate(models):
        assert np.array_equal(losses[r].cpu().numpy(), want_loss), (mode, world, r)
        assert np.array_equal(mm._table.cpu().numpy().view(np.uint32), want), (mode, world, r)
        mm.check_device_errors()
    if mode == "rows" and world == 2:          # rank 1 leaves before its first collective; rank 0 is released, not left waiting
        small = _make_model(pkg, g, tmp_path, act_dtype="fp8", B=B // 2)[1]
        st_small = small._state(max_batch=B // 2, need_ctx=True, dp_world=world)
        ctxs = [states[0]['ctx'], st_small['ctx']]
        rcs = [None] * world

        def ab_main(r):
            rcs[r] = lib.lgcn_train_epoch_dp(ctxs[r], comms[r], L.tp(U), L.tp(P), L.tp(Nn), T, B, 0, None,
                                             L.tp(gathered[r]), L.tp(losses[r]), C.c_void_p(streams[r].cuda_stream))
        threads = [threading.Thread(target=ab_main, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=60)
        torch.cuda.synchronize()
        assert not any(t.is_alive() for t in threads) and rcs[1] == 3 and rcs[0] not in (0, None), rcs
    for r in range(world):
        lib.lgcn_dp_destroy(comms[r])
