#!/usr/bin/env python3
"""GPU debug: row-sharded emulation vs unsharded, step by step, reporting the rows that differ."""
import ctypes as C, importlib, os, shutil, sys, tempfile
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
sys.argv = [sys.argv[0]]
from conftest import GoldenSet
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
DEV = "cuda:0"
which, world = os.environ.get("RS_CASE", "lastfm,4").split(",")
world = int(world)
g = GoldenSet(which)
tmp = tempfile.mkdtemp()

def make():
    d = os.path.join(tmp, "d"); os.makedirs(d, exist_ok=True)
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(g.dir, f), os.path.join(d, f))
    w = pkg.world; w.configure([]); w.dataset = g.name
    w.config.update({'lightGCN_n_layers': g.K, 'latent_dim_rec': g.d, 'bpr_batch_size': g.B, 'decay': g.meta["decay"], 'lr': g.meta["lr"]})
    ds = pkg.dataloader.Loader(w.config, path=d)
    pkg.sampling.seed(w.seed); pkg.utils.set_seed(w.seed)
    return ds, pkg.model.LightGCN(w.config, ds).to(DEV)

L, lib = pkg._lib, pkg._lib.load()
rng = np.random.Generator(np.random.PCG64(world))
B = 64
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(torch.int32).to(DEV)
batches = [tuple(dev(rng.integers(0, hi, b)) for hi in (g.n_users, g.m_items, g.m_items)) for b in (B, B, 37)]
ds, ref = make()
ds, m = make()
adj = ds.getSparseGraphCSR()
ranges = pkg.parallel.row_ranges(adj.indptr, ds.n_users, world)
own = [pkg.parallel.owned_rows(ranges, r) for r in range(world)]
owner = np.empty(adj.shape[0], np.int32)
for r, o in enumerate(own): owner[o] = r
st = m._state(max_batch=B, need_ctx=True, dp_world=world)
print("dense_last", st['dense_last'], "K", g.K, "ranges", ranges.tolist())
full = st['graph']
graphs = [L.Graph(full.indptr, full.indices, full.vals, d_max=g.d, row_order=o) for o in own]
if os.environ.get("RS_FULLPLAN") == "1":
    graphs = [full]
ctxs = []
for r in range(world):
    cfg = L.TrainConfig(); cfg.graph = graphs[r].handle
    cfg.n_users, cfg.d, cfg.K, cfg.act_dtype = ds.n_users, g.d, g.K, 0
    cfg.E0, cfg.adam_m, cfg.adam_v = m._table.data_ptr(), st['adam_m'].data_ptr(), st['adam_v'].data_ptr()
    cfg.act, cfg.G64, cfg.bitmap = st['act'].data_ptr(), st['G64'].data_ptr(), st['bitmap'].data_ptr()
    cfg.terms, cfg.ebuf, cfg.contrib = st['terms'].data_ptr(), st['ebuf'].data_ptr(), st['contrib'].data_ptr()
    cfg.err, cfg.max_batch, cfg.decay = st['err'].data_ptr(), B, float(g.meta["decay"])
    cfg.lr, cfg.beta1, cfg.beta2, cfg.eps, cfg.xcd_remap = float(g.meta["lr"]), 0.9, 0.999, 1e-8, 1
    cfg.dense_last = int(st['dense_last'])
    h = C.c_void_p(); L.check(lib.lgcn_ctx_create(C.byref(cfg), C.byref(h)), "ctx"); ctxs.append(h)
K = g.K; fl = K if st['dense_last'] else K - 1
stream = L.current_stream()
deg = np.diff(adj.indptr)
for i, (u, p, n) in enumerate(batches):
    b = len(u)
    lref = ref.fused_step(u, p, n).cpu().numpy()
    def phase(r, ph, k, gathered=None, loss=None):
        L.check(lib.lgcn_rs_phase(ctxs[r], ph, k, L.tp(u), L.tp(p), L.tp(n), b, world, r, L.tp(gathered) if gathered is not None else None,
                                  L.tp(loss) if loss is not None else None, stream), f"phase {ph} {k}")
    for k in range(1, fl + 1):
        for r in range(world): phase(r, 0, k)
        torch.cuda.synchronize()
        refact = ref._dev['act'][k - 1].cpu().numpy(); myact = st['act'][k - 1].cpu().numpy()
        bad = np.flatnonzero((refact.view(np.uint32) != myact.view(np.uint32)).any(1))
        print(f"step {i} FWD {k}: differing rows {len(bad)}", bad[:8], owner[bad[:8]], deg[bad[:8]])
    nblk = pkg.parallel.block_numel(b, world, g.d); blocks = []
    for r in range(world):
        phase(r, 1, 0); blocks.append(st['contrib'][:nblk].clone())
    gathered = torch.cat(blocks)
    phase(0, 2, 0, gathered)
    for k in range(K, 0, -1):
        for r in range(world): phase(r, 3, k, gathered)
    loss = torch.empty(3, device=DEV); phase(0, 4, 0, gathered, loss)
    torch.cuda.synchronize()
    a_, b_ = m._table.cpu().numpy(), ref._table.cpu().numpy()
    bad = np.flatnonzero((a_.view(np.uint32) != b_.view(np.uint32)).any(1))
    print(f"step {i}: loss equal {np.array_equal(loss.cpu().numpy().view(np.uint32), lref.view(np.uint32))}; table rows differing {len(bad)}", bad[:10], "owners", owner[bad[:10]], "deg", deg[bad[:10]],
          "max abs diff", float(np.abs(a_ - b_).max()))
    for key in ('adam_m', 'adam_v'):
        x, y = st[key].cpu().numpy(), ref._dev[key].cpu().numpy()
        print("   ", key, "rows differing", int((x.view(np.uint32) != y.view(np.uint32)).any(1).sum()))

# ---- isolate: dense SpMM through the subset plans vs the full plan (same X)
torch.manual_seed(0)
X = torch.randn(adj.shape[0], g.d, device=DEV) * 0.1
Yf = full.spmm(X)
Ys = torch.zeros_like(Yf)
for gr in graphs:
    L.check(lib.lgcn_spmm_csr(gr.handle, L.tp(X), 0, L.tp(Ys), 0, g.d, stream), "spmm")
torch.cuda.synchronize()
bad = np.flatnonzero((Yf.cpu().numpy().view(np.uint32) != Ys.cpu().numpy().view(np.uint32)).any(1))
print("dense SpMM full vs subsets: differing rows", len(bad), bad[:10], "deg", deg[bad[:10]])
nat = L.Graph(full.indptr, full.indices, full.vals, d_max=g.d)
Yn = nat.spmm(X)
bad = np.flatnonzero((Yf.cpu().numpy().view(np.uint32) != Yn.cpu().numpy().view(np.uint32)).any(1))
print("dense SpMM xcd order vs natural order: differing rows", len(bad), bad[:10], "deg", deg[bad[:10]])
