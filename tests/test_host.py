"""CPU-only tests of the product's host side: the C-ABI library loads and exports every
symbol include/lgcn_hip.h declares, the native samplers / shuffle / graph builder are
bit-exact against the golden fixtures, the oracle and oracle/_ref, and the config surface
matches the reference's flags.  No device compute is launched here."""
import hashlib
import importlib
import json
import os
import re
import shutil
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, PKG_NAME


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_library_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(REPO, "include", "lgcn_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lgcn_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = pkg._lib.load()
    assert set(pkg._lib.SIGNATURES) == declared
    for name in declared:
        assert hasattr(lib, name), name
    want = int(re.search(r"#define\s+LGCN_ABI_VERSION\s+(\d+)", open(os.path.join(REPO, "include", "lgcn_hip.h")).read()).group(1))
    assert lib.lgcn_abi_version() == want == pkg._lib.ABI_VERSION
    assert lib.lgcn_device_available() in (0, 1)


def test_graft_entry_build_checks(pkg):
    """__graft_entry__.build() = compile + check_build(); the checks (ABI version of header, binding and library
    agree; every declared symbol exported and bound) run here on the library the session already has, and the
    module's own constants cannot drift from them (round 2 shipped a build() that asserted a stale literal)."""
    import __graft_entry__ as ge
    lib = ge.check_build(pkg)
    assert lib is pkg._lib.load()
    src = open(ge.__file__).read()
    assert "lgcn_abi_version() ==" in src and not re.search(r"lgcn_abi_version\(\) == \d", src)


def test_glibc_stream_and_randint(pkg):
    s = pkg.sampling
    s.seed(2020)
    assert [s.randint(1000) for _ in range(5)] == [917, 315, 409, 907, 967]
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2020, 0xFFFFFFFF, 0):
        libc.srand(ctypes.c_uint(seed)); s.seed(seed)
        assert [libc.rand() % 1000003 for _ in range(3000)] == [s.randint(1000003) for _ in range(3000)]


def _load(pkg, g, tmp_path):
    d = os.path.join(str(tmp_path), g.name)
    os.makedirs(d, exist_ok=True)
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(g.dir, f), os.path.join(d, f))
    pkg.world.dataset = g.name
    return pkg.dataloader.Loader(pkg.world.config, path=d)


def test_loader_and_graph_bit_exact(pkg, tiny, lastfm, tmp_path):
    for g in (tiny, lastfm):
        ds = _load(pkg, g, tmp_path)
        assert (ds.n_users, ds.m_items, ds.trainDataSize) == (g.n_users, g.m_items, g.meta["trainDataSize"])
        adj = ds.getSparseGraphCSR()
        assert adj.indptr.dtype == np.int32 and adj.indices.dtype == np.int32 and adj.data.dtype == np.float32
        assert np.array_equal(adj.indptr, g.z["adj_indptr"])
        assert np.array_equal(adj.indices, g.z["adj_indices"])
        assert np.array_equal(adj.data.view(np.uint32), g.z["adj_data"].view(np.uint32))
        assert os.path.exists(os.path.join(ds.path, "s_pre_adj_mat.npz"))      # same cache file as the reference
        # reload goes through the cache and gives the same matrix
        ds2 = pkg.dataloader.Loader(pkg.world.config, path=ds.path)
        adj2 = ds2.getSparseGraphCSR()
        assert np.array_equal(adj2.data.view(np.uint32), adj.data.view(np.uint32))
        coo = ds2.getSparseGraph()
        assert coo.is_sparse and coo.shape == (g.n_users + g.m_items,) * 2 and coo._nnz() == len(adj.data)
        assert all(np.all(np.diff(p) > 0) for p in ds.allPos if len(p) > 1)
        td = ds.testDict
        assert sum(len(v) for v in td.values()) == len(g.test_user)


def test_compiled_sampling_module_matches_reference_plugin(tmp_path):
    """The pybind11 extension module `sampling` (csrc/sampling_module.cpp, loaded BY PATH as utils.py:25-34 loads the
    reference's cppimport build): the reference's four names and docstrings, int32 outputs of the reference's shapes,
    rows identical to the ones the reference's own compiled plugin produced (tiny fixture), one rand() stream shared
    with the ctypes binding, list-of-arrays and CSR-tuple forms, ValueError where the reference dies with SIGFPE.
    Runs in a process of its own: two extension modules called `sampling` (this one and the reference's build under
    oracle/_ref that test_sampler_vs_compiled_reference_and_oracle imports) cannot live in one interpreter."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(REPO, "tests", "sampling_module_check.py"), str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), p.stdout[-2000:] + p.stderr[-2000:]


def test_sampler_cpp_matches_golden(pkg, tiny, tmp_path):
    ds = _load(pkg, tiny, tmp_path)
    pkg.sampling.seed(2020)
    for e in (1, 2):
        S = pkg.sampling.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.allPos, 1)
        assert S.dtype == np.int32 and np.array_equal(S, tiny.z[f"S_epoch{e}"])
    # zero-copy CSR overload gives the same stream
    pkg.sampling.seed(2020)
    S = pkg.sampling.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.pos_csr(), 1)
    assert np.array_equal(S, tiny.z["S_epoch1"])


def test_sampler_python_and_shuffle_match_golden(pkg, tiny, lastfm, tmp_path):
    z = np.load(os.path.join(tiny.dir, "golden_python_sampler.npz"))
    ds = _load(pkg, tiny, tmp_path)
    pkg.utils.set_seed(2020)
    S1 = pkg.utils.UniformSample_original_python(ds)
    perm = pkg.utils.shuffle_indices(len(S1))
    S2 = pkg.utils.UniformSample_original_python(ds)
    assert S1.dtype == np.int64 and np.array_equal(S1, z["S_python_epoch1"])
    assert np.array_equal(perm, z["perm_after_epoch1"]) and np.array_equal(S2, z["S_python_epoch2"])
    ds = _load(pkg, lastfm, tmp_path)
    assert pkg.utils.sampler_mode(ds) == "python"          # 14 users without positives
    pkg.utils.set_seed(2020)
    S = pkg.utils.UniformSample_original(ds)
    assert sha(S) == lastfm.meta["S_epoch1_sha256"]
    (su,), idx = pkg.utils.shuffle(S[:, 0], indices=True)
    assert np.array_equal(su, lastfm.z["shuf_users_epoch1"])
    # native stream == numpy's own legacy stream
    np.random.seed(77); pkg._lib.load().lgcn_np_seed(77)
    ref = np.arange(12345); np.random.shuffle(ref)
    assert np.array_equal(ref, pkg.utils.shuffle_indices(12345))


def test_sampler_rejects_user_without_positives(pkg, lastfm, tmp_path):
    ds = _load(pkg, lastfm, tmp_path)
    with pytest.raises(pkg._lib.LgcnError, match="no training positives"):
        pkg.sampling.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.pos_csr(), 1)


def test_sampler_vs_compiled_reference_and_oracle(pkg, oracle):
    ref_dir = os.path.join(REPO, "oracle", "_ref")
    have_ref = os.path.isdir(ref_dir) and any(f.startswith("sampling") and f.endswith(".so") for f in os.listdir(ref_dir))
    if have_ref:
        sys.path.insert(0, ref_dir)
        ref = importlib.import_module("sampling")
    rng = np.random.Generator(np.random.PCG64(11))
    n_users, m_items = 257, 403
    rows = [np.sort(rng.choice(m_items, size=int(rng.integers(1, 90)), replace=False)).astype(np.int32)
            for _ in range(n_users)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows)
    train_num = int(indptr[-1])
    for seed in (2020, 3):
        pkg.sampling.seed(seed); oracle.srand(seed)
        if have_ref:
            ref.seed(seed)
        for neg in (1, 2):
            a = pkg.sampling.sample_negative(n_users, m_items, train_num, rows, neg)
            b = oracle.sample_negative(n_users, m_items, train_num, indptr, indices, neg)
            assert np.array_equal(a, b)
            if have_ref:
                c = ref.sample_negative(n_users, m_items, train_num, rows, neg)
                assert c.dtype == a.dtype and c.shape == a.shape and np.array_equal(a, c)
        us = rng.integers(0, n_users, 50).astype(np.int32)
        a = pkg.sampling.sample_negative_ByUser(us, m_items, rows, 2)
        assert np.array_equal(a, oracle.sample_negative_by_user(us, m_items, indptr, indices, 2))
        if have_ref:
            assert np.array_equal(a, ref.sample_negative_ByUser(us.tolist(), m_items, rows, 2))


def test_gowalla_sampler_hashes(pkg):
    """Full-size integer parity: 806 166 triplets per epoch, two epochs of the glibc stream,
    the python-mode epoch and the 806 166-element shuffle, against hashes captured from the
    reference on the reconstructed Gowalla graph."""
    npz = os.path.join(GOLDEN, "gowalla", "gowalla.npz")
    if not os.path.exists(npz):
        pytest.skip("gowalla.npz fixture not generated")
    meta = json.load(open(os.path.join(GOLDEN, "gowalla", "golden_samplers.json")))
    z = np.load(npz)
    lib = pkg._lib.load()
    n_users, m_items = 29858, 40981
    tu = np.repeat(z["train_users"].astype(np.int64), np.diff(z["train_ptr"]))
    ti = z["train_items"].astype(np.int64)
    indptr = np.zeros(n_users + 1, np.int64); nnz = np.zeros(1, np.int64)
    indices = np.empty(len(ti), np.int32); vals = np.empty(len(ti), np.float32)
    assert lib.lgcn_build_user_item_csr(n_users, m_items, len(ti), pkg._lib.npp(tu), pkg._lib.npp(ti),
                                        pkg._lib.npp(indptr), pkg._lib.npp(indices), pkg._lib.npp(vals),
                                        pkg._lib.npp(nnz)) == 0
    assert int(nnz[0]) == 810128
    pkg.sampling.seed(2020)
    for e in (1, 2):
        S = pkg.sampling.sample_negative(n_users, m_items, 810128, (indptr, indices), 1)
        assert S.shape == (806166, 3) and sha(S) == meta[f"cpp_epoch{e}"]["sha256"]
    pkg._lib.load().lgcn_np_seed(2020)
    perm = pkg.utils.shuffle_indices(806166)
    assert perm[:5].tolist() == meta["shuffle_806166_head"] and sha(perm) == meta["shuffle_806166_sha256"]
    pkg._lib.load().lgcn_np_seed(2020)
    S = np.empty((810128, 3), np.int64)
    rows = lib.lgcn_sample_python(n_users, m_items, 810128, pkg._lib.npp(indptr), pkg._lib.npp(indices), pkg._lib.npp(S))
    assert rows == 810128 and sha(S) == meta["python_epoch1"]["sha256"]


def test_config_surface_matches_reference_flags(pkg):
    w = pkg.world
    w.configure([])
    expect = {'lr': 0.001, 'decay': 1e-4, 'lightGCN_n_layers': 3, 'latent_dim_rec': 64, 'bpr_batch_size': 2048,
              'test_u_batch_size': 100, 'dropout': 0, 'keep_prob': 0.6, 'A_split': False, 'A_n_fold': 100,
              'epochs': 1000, 'multicore': 0, 'pretrain': 0, 'seed': 2020, 'model': 'lgn', 'dataset': 'gowalla',
              'exp_smooth_beta': 0.5, 'use_ppr_weights': False, 'ppr_weights_path': None, 'use_scheduler': False,
              'sched_gamma': 0.5, 'sched_milestones': [120, 240, 360, 480], 'use_pop_gate': False,
              'pop_hidden': 32, 'gate_hidden': 64, 'gate_entropy_coeff': 1e-4, 'pop_gate_temp': 1.0,
              'use_item_item': False, 'i2i_path': None, 'i2i_alpha': 0.0, 'checkpoint_dir': './checkpoints'}
    for k, v in expect.items():
        assert w.config[k] == v, k
    assert (w.seed, w.dataset, w.comment, w.tensorboard, w.LOAD, w.model_name, w.TRAIN_epochs, w.topks) == \
        (2020, 'gowalla', 'lgn', 1, 0, 'lgn', 1000, [20])
    w.configure(['--layer', '4', '--recdim', '128', '--bpr_batch', '8192', '--topks', '[20, 40]', '--A_split'])
    assert w.config['lightGCN_n_layers'] == 4 and w.config['latent_dim_rec'] == 128 and w.topks == [20, 40]
    assert w.config['A_split'] is True
    w.configure([])


def test_additive_flags_default_to_the_reference_behaviour(pkg):
    """The NEW flags of rounds 1-4: their defaults reproduce the reference (its loss, its arithmetic); the alternatives parse."""
    w = pkg.world
    w.configure([])
    assert w.config['reg_rows'] == 'propagated'            # model.py:173 of THIS reference (upstream LightGCN: 'ego')
    assert w.config['act_dtype'] == 'fp32' and w.config['gpu_shuffle'] == 1 and w.config['gpu_sampler'] == 1
    assert w.config['prefetch_epoch'] == 1
    w.configure(['--reg_rows', 'ego', '--act_dtype', 'fp8', '--gpu_shuffle', '0'])
    assert (w.config['reg_rows'], w.config['act_dtype'], w.config['gpu_shuffle']) == ('ego', 'fp8', 0)
    with pytest.raises(SystemExit):
        w.configure(['--reg_rows', 'both'])
    w.configure([])


def test_table_and_mask_sizes(pkg):
    """Pure size arithmetic of the ABI (no GPU): fp8 tables = rows + fp32 row scales padded to 256 bytes; evaluation masks =
    one 32-bit word per (32-item tile, evaluation slot), the slots padded to 128."""
    lib = pkg._lib.load()
    L = pkg._lib
    assert lib.lgcn_table_bytes(1000, 64, L.F32) == 1000 * 64 * 4 and lib.lgcn_table_bytes(1000, 64, L.BF16) == 1000 * 64 * 2
    assert lib.lgcn_table_bytes(1000, 64, L.FP8) == (1000 * 64 + 1000 * 4 + 255) // 256 * 256
    assert lib.lgcn_table_bytes(11_000_000, 256, L.FP8) == (11_000_000 * 260 + 255) // 256 * 256          # > 2^31: 64-bit
    assert lib.lgcn_table_bytes(0, 64, L.FP8) == 0
    assert lib.lgcn_eval_mask_words(40981, 29858) == ((40981 + 31) // 32) * ((29858 + 127) // 128 * 128)
    assert lib.lgcn_eval_mask_words(0, 5) == 0 and lib.lgcn_eval_mask_words(33, 1) == 2 * 128


def test_deferred_loss_is_a_number(pkg):
    """--lazy_loss 1: what BPRLoss.stageOne returns then.  The arithmetic the reference does with the per-step loss (Procedure.py:61-68,
    main.py:223-236: += into a float, / total_batch, :.3f) and the usual conversions, without a device."""
    import torch
    D = pkg.utils.DeferredLoss
    vals = [0.5, 0.25, 1.0, 0.125]
    tot = 0.0
    for v in vals:
        tot += D([torch.tensor(v)])
    assert isinstance(tot, D) and float(tot) == sum(vals) and f"{tot / len(vals):.3f}" == f"{sum(vals) / len(vals):.3f}"
    a, b = D([torch.tensor(0.5)]), D([torch.tensor(0.25)])
    assert float(2 * a - 1) == 0.0 and float(1 - a) == 0.5 and float(-(a + b) * 2 + 1) == -0.5 and 1.0 / a == 2.0
    assert b < a and a > 0.3 and a == 0.5 and max(a, b) is a and float(sum([a, b])) == 0.75
    assert np.float64(a) == 0.5 and float(np.asarray(b)) == 0.25 and a.item() == 0.5 and repr(a) == "0.5"
    w = pkg.world
    w.configure(['--lazy_loss', '1']); assert w.config['lazy_loss'] == 1
    w.configure([]); assert w.config['lazy_loss'] == 0


def test_minibatch_and_timer(pkg):
    u = pkg.utils
    a = np.arange(10)
    out = list(u.minibatch(a, a * 2, batch_size=4))
    assert [len(x[0]) for x in out] == [4, 4, 2] and np.array_equal(out[2][1], [16, 18])
    with u.timer(name="Sample"):
        pass
    assert u.timer.dict().startswith("|Sample:")
    u.timer.zero()
    assert u.timer.NAMED_TAPE["Sample"] == 0


def test_compute_fails_loudly_without_gpu(pkg, tiny, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ds = _load(pkg, tiny, tmp_path)
    pkg.world.config['lightGCN_n_layers'], pkg.world.config['latent_dim_rec'] = 3, 64
    m = pkg.model.LightGCN(pkg.world.config, ds)
    assert list(m.state_dict().keys()) == ['embedding_user.weight', 'embedding_item.weight']
    with pytest.raises(pkg._lib.LgcnError):
        m.computer()
    bpr = pkg.utils.BPRLoss(m, pkg.world.config)
    with pytest.raises(pkg._lib.LgcnError):
        bpr.stageOne(np.array([0]), np.array([1]), np.array([2]))


def test_model_init_matches_reference_rng(pkg, tiny, tmp_path):
    ds = _load(pkg, tiny, tmp_path)
    pkg.world.config['lightGCN_n_layers'], pkg.world.config['latent_dim_rec'] = tiny.K, tiny.d
    pkg.utils.set_seed(2020)
    m = pkg.model.LightGCN(pkg.world.config, ds)
    assert np.array_equal(m.embedding_user.weight.detach().numpy(), tiny.z["E0_user"])
    assert np.array_equal(m.embedding_item.weight.detach().numpy(), tiny.z["E0_item"])
    # one storage
    assert m.embedding_item.weight.data_ptr() == m._table.data_ptr() + tiny.n_users * tiny.d * 4
    sd = {k: v.clone() + 1 for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert m.embedding_user.weight.data_ptr() == m._table.data_ptr()
    assert np.allclose(m._table[:tiny.n_users].numpy(), tiny.z["E0_user"] + 1)


def test_row_orders_are_permutations(pkg, tiny, lastfm, tmp_path):
    for g in (tiny, lastfm):
        ds = _load(pkg, g, tmp_path)
        adj = ds.getSparseGraphCSR()
        N = g.n_users + g.m_items
        assert pkg.reorder.row_order('natural', ds, adj) == (None, None)
        for method in ('rcm', 'cocluster', 'xcd'):
            o, xs = pkg.reorder.row_order(method, ds, adj, cache_dir=ds.path)
            assert o.dtype == np.int32 and np.array_equal(np.sort(o), np.arange(N)), method
            assert any(f.startswith(f"s_row_order_{method}") for f in os.listdir(ds.path))
            o2, xs2 = pkg.reorder.row_order(method, ds, adj, cache_dir=ds.path)                    # cached
            assert np.array_equal(o, o2)
            if method == 'xcd':      # the 8 XCD slices: a monotone cut of the order, balanced by non-zeros
                assert xs.dtype == np.int64 and xs.shape == (9,) and xs[0] == 0 and xs[-1] == N
                assert np.all(np.diff(xs) >= 0) and np.array_equal(xs, xs2)
                if g is lastfm:
                    w = np.diff(adj.indptr)[o]
                    per = np.array([w[xs[x]:xs[x + 1]].sum() for x in range(8)], np.float64)
                    assert per.max() < 1.25 * per.mean(), per
            else:
                assert xs is None and xs2 is None
    with pytest.raises(ValueError):
        pkg.reorder.row_order('bogus', ds, adj)


def test_synthetic_generator_and_csr_loader(pkg):
    """The vectorised generator behind the BASELINE configs[2..4] shapes: exact E, every user >= 1 item,
    sorted unique columns, seeded; CsrLoader builds the same dataset object the text Loader does."""
    ip, ix = pkg.synthetic.power_law_bipartite(3000, 2000, 40000, seed=5, device="cpu")
    ip2, ix2 = pkg.synthetic.power_law_bipartite(3000, 2000, 40000, seed=5, device="cpu")
    assert np.array_equal(ip, ip2) and np.array_equal(ix, ix2)
    deg = np.diff(ip)
    assert ip[-1] == 40000 and len(ix) == 40000 and deg.min() >= 1 and deg.max() <= 2000 // 4
    key = np.repeat(np.arange(3000, dtype=np.int64), deg) * 2000 + ix
    assert np.all(np.diff(key) > 0)                                   # sorted, no duplicates
    assert np.bincount(ix, minlength=2000).max() > 20 * 40000 / 2000   # heavy-tailed item popularity
    pkg.world.configure([])
    ds = pkg.dataloader.CsrLoader(ip, ix, 2000, test_dict={0: [1, 2], 5: [3]}, config=pkg.world.config)
    assert (ds.n_users, ds.m_items, ds.trainDataSize) == (3000, 2000, 40000)
    assert np.array_equal(ds.allPos[7], ix[ip[7]:ip[8]]) and ds.testDict == {0: [1, 2], 5: [3]}
    adj = ds.getSparseGraphCSR()
    from oracle import oracle as orc
    oip, oix, odata, _ = orc.build_norm_adj(3000, 2000, ip, ix, np.ones(40000, np.float32))
    assert np.array_equal(adj.indptr, oip) and np.array_equal(adj.indices, oix) and np.array_equal(adj.data, odata)
    with pytest.raises(ValueError):
        pkg.dataloader.CsrLoader(ip, ix, 1000, config=pkg.world.config)          # item id out of range


def test_user_item_csr_matches_scipy_on_random_inputs(pkg):
    """Native COO -> canonical CSR (sorted columns, duplicates summed) vs scipy, incl. duplicates,
    empty users and the id range check."""
    import scipy.sparse as sp
    lib = pkg._lib.load()
    rng = np.random.Generator(np.random.PCG64(42))
    for n_users, m_items, E in ((1, 1, 1), (7, 5, 40), (300, 211, 5000), (50, 1000, 0)):
        tu = rng.integers(0, n_users, E).astype(np.int64)
        ti = rng.integers(0, m_items, E).astype(np.int64)
        indptr = np.zeros(n_users + 1, np.int64); nnz = np.zeros(1, np.int64)
        indices = np.empty(max(E, 1), np.int32); vals = np.empty(max(E, 1), np.float32)
        rc = lib.lgcn_build_user_item_csr(n_users, m_items, E, pkg._lib.npp(tu), pkg._lib.npp(ti),
                                          pkg._lib.npp(indptr), pkg._lib.npp(indices), pkg._lib.npp(vals), pkg._lib.npp(nnz))
        assert rc == 0
        R = sp.csr_matrix((np.ones(E, np.float32), (tu, ti)), shape=(n_users, m_items))
        R.sum_duplicates(); R.sort_indices()
        k = int(nnz[0])
        assert k == R.nnz and np.array_equal(indptr, R.indptr)
        assert np.array_equal(indices[:k], R.indices) and np.array_equal(vals[:k], R.data)
    bad = np.array([0, 9], np.int64)
    assert lib.lgcn_build_user_item_csr(5, 5, 2, pkg._lib.npp(bad), pkg._lib.npp(bad), pkg._lib.npp(np.zeros(6, np.int64)),
                                        pkg._lib.npp(np.zeros(2, np.int32)), pkg._lib.npp(np.zeros(2, np.float32)),
                                        pkg._lib.npp(np.zeros(1, np.int64))) == 3


def test_sampler_invariants(pkg, tiny, tmp_path):
    """Every triplet: positive is one of the user's train items, negative is not; cpp mode draws
    trainDataSize//n_users triplets per user grouped by user."""
    ds = _load(pkg, tiny, tmp_path)
    pos_sets = [set(p.tolist()) for p in ds.allPos]
    pkg.sampling.seed(99)
    for neg_num in (1, 3):
        S = pkg.sampling.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.pos_csr(), neg_num)
        per = ds.trainDataSize // ds.n_users
        assert S.shape == (ds.n_users * per, 2 + neg_num)
        assert np.array_equal(S[:, 0], np.repeat(np.arange(ds.n_users), per))
        for row in S:
            assert row[1] in pos_sets[row[0]] and all(x not in pos_sets[row[0]] for x in row[2:])
            assert all(0 <= x < ds.m_items for x in row[1:])
    pkg.utils.set_seed(5)
    S = pkg.utils.UniformSample_original_python(ds)
    assert len(S) == ds.trainDataSize
    for u, p, n in S:
        assert p in pos_sets[u] and n not in pos_sets[u]


@pytest.mark.parametrize("tag", ["gate", "i2i", "gate_i2i"])
def test_optional_branches_construct_like_the_reference(pkg, tmp_path, tag):
    """Popularity gate / item-item smoothing: same state_dict keys and bit-identical initial
    parameters as the reference (fixtures captured by tests/golden/make_golden.py tiny_*), on the CPU;
    the fused step refuses the model, compute needs the GPU."""
    import json, shutil
    from conftest import GOLDEN
    tiny = os.path.join(GOLDEN, "tiny")
    gz = np.load(os.path.join(tiny, f"golden_{tag}.npz"))
    meta = json.load(open(os.path.join(tiny, f"golden_{tag}.json")))
    d = str(tmp_path)
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(tiny, f), os.path.join(d, f))
    w = pkg.world
    w.configure([])
    w.dataset = "tiny"
    w.config.update({'lightGCN_n_layers': meta["K"], 'latent_dim_rec': meta["d"], 'bpr_batch_size': meta["B"],
                     'use_pop_gate': meta["use_pop_gate"], 'use_item_item': meta["use_item_item"],
                     'i2i_path': os.path.join(tiny, "i2i_tiny.npz") if meta["use_item_item"] else None, 'i2i_alpha': meta["i2i_alpha"]})
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.utils.set_seed(meta["seed"])
    m = pkg.model.LightGCN(w.config, ds)
    sd = m.state_dict()
    assert sorted(sd) == sorted(k[3:] for k in gz.files if k.startswith("P0."))
    for k, v in sd.items():
        assert np.array_equal(v.numpy(), gz["P0." + k]), k
    assert m.has_variants and m.i2i_active == meta["use_item_item"] and m.use_pop_gate == meta["use_pop_gate"]
    assert m.fused_variants and pkg.utils.BPRLoss(m, w.config).fused          # the branches run inside the fused step ...
    w.config['fused_variants'] = 0
    assert not m.fused_variants and not pkg.utils.BPRLoss(m, w.config).fused  # ... unless the autograd path is asked for
    w.config['fused_variants'] = 1
    if meta["use_pop_gate"]:                  # the gate's eight tensors are views of one buffer, in named_parameters order
        assert m._pack_gate() and not m._pack_gate()
        flat = m._gate_flat
        off = 0
        for k, prm in m.named_parameters():
            if k.startswith("embedding"):
                continue
            assert prm.data_ptr() == flat.data_ptr() + 4 * off and np.array_equal(prm.detach().numpy(), gz["P0." + k]), k
            off += prm.numel()
        assert off == flat.numel()
    w.config.update({'use_item_item': True, 'i2i_path': os.path.join(d, "missing.npz"), 'i2i_alpha': 0.5, 'use_pop_gate': False})
    m2 = pkg.model.LightGCN(w.config, ds)            # unreadable file: the reference warns and trains without it
    assert not m2.i2i_active and not m2.has_variants
    w.configure([])
