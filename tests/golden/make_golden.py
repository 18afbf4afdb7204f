#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ by IMPORTING the
reference (read-only, /root/reference) in this container.

TEST INFRASTRUCTURE ONLY.  Nothing here is shipped or imported by the product
path, and nothing of the reference's source text is written to the repo: the
outputs are data only (inputs + expected outputs as .npz / .json / .txt
interaction lists).

Usage:  python tests/golden/make_golden.py [tiny] [lastfm] [gowalla] [gowalla_long] [tiny_gate] [tiny_i2i] [tiny_gate_i2i] [tiny_gate_i2i_k1|_k2|_k4]

What the reference pieces are (all paths relative to
/root/reference/LightGCN_work/code):
  dataloader.Loader / getSparseGraph     dataloader.py:51-246
  model.LightGCN (computer, bpr_loss)    model.py:37-231 (incl. the optional popularity gate :66-96,139-181 and
                                         item-item smoothing :99-109,228-229 -- targets tiny_gate / tiny_i2i)
  utils.BPRLoss.stageOne                 utils.py:38-64
  utils.UniformSample_original(_python)  utils.py:68-110
  utils.shuffle / minibatch / set_seed   utils.py:114-151
  Procedure.test_one_batch + metrics     Procedure.py:89-121, utils.py:173-217
  sources/sampling.cpp                   compiled by oracle/Makefile into oracle/_ref
Order of seeding follows SURVEY 3.2: sampling.seed -> utils.set_seed -> model.
The shipped epoch loop (main.py:215-225) and Test (Procedure.py:162-192) do
not run as shipped, so their semantics are restated here around the
reference's own building blocks.
"""
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF_CODE = "/root/reference/LightGCN_work/code"
REF_DATA = "/root/reference/LightGCN_work/data"

sys.dont_write_bytecode = True


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# --------------------------------------------------------------------------
# dataset materialisation (our own code; inputs are the reference's data files)
# --------------------------------------------------------------------------
def write_interactions(path, lists):
    with open(path, "w") as f:
        for u, items in lists:
            f.write(str(u) + " " + " ".join(str(int(i)) for i in items) + "\n")


def make_tiny(dst):
    """50 users x 80 items, every user >= 2 train items, seeded PCG64(2020)."""
    rng = np.random.Generator(np.random.PCG64(2020))
    n_users, m_items = 50, 80
    pop = 1.0 / np.arange(1, m_items + 1) ** 0.8
    pop /= pop.sum()
    train, test = [], []
    for u in range(n_users):
        deg = int(rng.integers(3, 18))
        items = rng.choice(m_items, size=deg, replace=False, p=pop)
        n_test = max(1, deg // 5)
        test.append((u, items[:n_test]))
        train.append((u, items[n_test:]))
    # make sure the largest ids appear so that n_users/m_items are as stated
    train[-1] = (train[-1][0], np.unique(np.append(train[-1][1], m_items - 1)))
    test[-1] = (test[-1][0], np.setdiff1d(test[-1][1], train[-1][1]))
    if len(test[-1][1]) == 0:
        test[-1] = (test[-1][0], np.setdiff1d(np.arange(m_items), train[-1][1])[:1])
    os.makedirs(dst, exist_ok=True)
    write_interactions(os.path.join(dst, "train.txt"), train)
    write_interactions(os.path.join(dst, "test.txt"), test)


def make_lastfm(dst):
    """data/lastfm/data1.txt,test1.txt: 1-indexed 'user\\titem\\tweight' triplets ->
    0-indexed 'uid item item ...' grouped by user in order of first appearance."""
    os.makedirs(dst, exist_ok=True)
    for src, name in (("data1.txt", "train.txt"), ("test1.txt", "test.txt")):
        a = np.loadtxt(os.path.join(REF_DATA, "lastfm", src), dtype=np.int64)
        users, items = a[:, 0] - 1, a[:, 1] - 1
        groups, order = {}, []
        for u, i in zip(users, items):
            if u not in groups:
                groups[u] = []
                order.append(u)
            groups[u].append(i)
        write_interactions(os.path.join(dst, name), [(u, groups[u]) for u in order])


def make_gowalla(dst):
    """Rebuild train.txt from code/cache/gowalla_sp_graph.npz (raw COO of
    [[0,R],[R^T,0]] in train.txt order); test.txt is shipped."""
    os.makedirs(dst, exist_ok=True)
    z = np.load(os.path.join(REF_CODE, "cache", "gowalla_sp_graph.npz"))
    keys = list(z.keys())
    if "row" in keys:
        row, col = z["row"], z["col"]
        n_users = int(z["n_users"]) if "n_users" in keys else 29858
    else:  # scipy save_npz layout
        import scipy.sparse as sp
        m = sp.load_npz(os.path.join(REF_CODE, "cache", "gowalla_sp_graph.npz")).tocoo()
        row, col, n_users = m.row, m.col, 29858
    sel = row < n_users
    u, i = row[sel].astype(np.int64), (col[sel] - n_users).astype(np.int64)
    cuts = np.flatnonzero(np.diff(u)) + 1
    starts = np.concatenate([[0], cuts])
    ends = np.concatenate([cuts, [len(u)]])
    write_interactions(os.path.join(dst, "train.txt"),
                       [(int(u[s]), i[s:e]) for s, e in zip(starts, ends)])
    shutil.copyfile(os.path.join(REF_DATA, "gowalla", "test.txt"), os.path.join(dst, "test.txt"))


def pack_dataset(src_dir, out_npz):
    """train/test interaction lists -> compact npz (file order preserved)."""
    out = {}
    for name in ("train", "test"):
        users, ptr, items = [], [0], []
        with open(os.path.join(src_dir, name + ".txt")) as f:
            for l in f:
                c = l.split()
                if len(c) < 2:
                    continue
                users.append(int(c[0]))
                items.extend(int(x) for x in c[1:])
                ptr.append(len(items))
        out[name + "_users"] = np.asarray(users, np.int32)
        out[name + "_ptr"] = np.asarray(ptr, np.int32)
        out[name + "_items"] = np.asarray(items, np.int32)
    np.savez_compressed(out_npz, **out)


# --------------------------------------------------------------------------
# reference import
# --------------------------------------------------------------------------
def import_reference(dataset_name):
    sys.argv = ["make_golden", "--dataset", dataset_name, "--tensorboard", "0"]
    if REF_CODE not in sys.path:
        sys.path.insert(0, REF_CODE)
    import world, dataloader, model, utils, Procedure  # noqa: E401
    world.dataset = dataset_name
    return world, dataloader, model, utils, Procedure


def load_ref_sampler(utils, world):
    """What utils.py:25-31 does when cppimport exists."""
    sys.path.insert(0, os.path.join(REPO, "oracle", "_ref"))
    import sampling
    utils.sampling = sampling
    utils.sample_ext = True
    sampling.seed(world.seed)
    return sampling


def ref_test(world, utils, Procedure, dataset, Recmodel, u_batch_size):
    """Semantics of Procedure.py:162-192 with the minibatch 1-tuple unwrapped."""
    import torch
    testDict = dataset.testDict
    Recmodel = Recmodel.eval()
    max_K = max(world.topks)
    batch_result = []
    with torch.no_grad():
        users = list(testDict.keys())
        for (batch_users,) in utils.minibatch(users, batch_size=u_batch_size):
            allPos = dataset.getUserPosItems(batch_users)
            groundTrue = [testDict[u] for u in batch_users]
            batch_gpu = torch.Tensor(batch_users).long().to(world.device)
            rating_K = Recmodel.getUsersRating(batch_gpu)
            exclude_idx, exclude_items = [], []
            for i, items in enumerate(allPos):
                exclude_idx.extend([i] * len(items))
                exclude_items.extend(items)
            rating_K[exclude_idx, exclude_items] = -(1 << 10)
            _, topk = torch.topk(rating_K, k=max_K)
            for i, u in enumerate(batch_users):
                batch_result.append(Procedure.test_one_batch((topk[i], groundTrue[i])))
    res = {m: np.mean([r[m] for r in batch_result], axis=0) for m in ("precision", "recall", "ndcg")}
    Recmodel.train()
    return {k: [float(x) for x in v] for k, v in res.items()}


def ref_epoch(world, utils, dataset, bpr, B, max_steps=None):
    """Semantics of main.py:215-225."""
    import torch
    S = utils.UniformSample_original(dataset)
    users = torch.tensor(S[:, 0], dtype=torch.long, device=world.device)
    pos = torch.tensor(S[:, 1], dtype=torch.long, device=world.device)
    neg = torch.tensor(S[:, 2], dtype=torch.long, device=world.device)
    users, pos, neg = utils.shuffle(users, pos, neg)
    losses = []
    for (bu, bp, bn) in utils.minibatch(users, pos, neg, batch_size=B):
        losses.append(bpr.stageOne(bu, bp, bn))
        if max_steps is not None and len(losses) >= max_steps:
            break
    return S, (users.numpy(), pos.numpy(), neg.numpy()), losses


def graph_csr(dataset):
    import scipy.sparse as sp
    g = sp.load_npz(os.path.join(dataset.path, "s_pre_adj_mat.npz")).tocsr()
    g.sort_indices()
    return g.indptr.astype(np.int32), g.indices.astype(np.int32), g.data.astype(np.float32)


def setup(name, K, d, B, workdir, cpp):
    import torch
    world, dataloader, model, utils, Procedure = import_reference(name)
    world.config["lightGCN_n_layers"] = K
    world.config["latent_dim_rec"] = d
    world.config["bpr_batch_size"] = B
    dataset = dataloader.Loader(world.config, path=workdir)
    if cpp:
        load_ref_sampler(utils, world)
    else:
        utils.sample_ext = False
    utils.set_seed(world.seed)
    Recmodel = model.LightGCN(world.config, dataset).to(world.device)
    bpr = utils.BPRLoss(Recmodel, world.config)
    return world, dataloader, model, utils, Procedure, dataset, Recmodel, bpr, torch


def params(Recmodel):
    return (Recmodel.embedding_user.weight.detach().numpy().copy(),
            Recmodel.embedding_item.weight.detach().numpy().copy())


# --------------------------------------------------------------------------
def gen_small(name, K, d, B, cpp, out_dir, full):
    """tiny / lastfm: everything dumped (full arrays)."""
    os.makedirs(out_dir, exist_ok=True)
    work = tempfile.mkdtemp(prefix="golden_" + name)
    (make_tiny if name == "tiny" else make_lastfm)(work)
    for f in ("train.txt", "test.txt"):
        shutil.copyfile(os.path.join(work, f), os.path.join(out_dir, f))
    (world, dataloader, model, utils, Procedure, dataset, Recmodel, bpr, torch) = \
        setup(name, K, d, B, work, cpp)
    out, meta = {}, {"dataset": name, "K": K, "d": d, "B": B, "sampler": "cpp" if cpp else "python",
                     "seed": world.seed, "lr": world.config["lr"], "decay": world.config["decay"],
                     "n_users": dataset.n_users, "m_items": dataset.m_items,
                     "trainDataSize": dataset.trainDataSize, "torch": torch.__version__,
                     "numpy": np.__version__}
    out["adj_indptr"], out["adj_indices"], out["adj_data"] = graph_csr(dataset)
    out["E0_user"], out["E0_item"] = params(Recmodel)
    with torch.no_grad():
        au, ai = Recmodel.computer()
    out["computer_users"], out["computer_items"] = au.numpy().copy(), ai.numpy().copy()
    meta["test_epoch0"] = ref_test(world, utils, Procedure, dataset, Recmodel, 100)

    # --- one explicit batch: loss, reg, grads (no optimiser step) ------------
    rng = np.random.Generator(np.random.PCG64(7))
    nb = min(B, 256)
    bu = rng.integers(0, dataset.n_users, nb)
    bu = np.array([u if len(dataset.allPos[u]) else dataset.trainUniqueUsers[0] for u in bu])
    bp = np.array([dataset.allPos[u][rng.integers(0, len(dataset.allPos[u]))] for u in bu])
    bn = rng.integers(0, dataset.m_items, nb)
    out["b_users"], out["b_pos"], out["b_neg"] = bu.astype(np.int64), bp.astype(np.int64), bn.astype(np.int64)
    loss, reg = Recmodel.bpr_loss(torch.tensor(bu).long(), torch.tensor(bp).long(), torch.tensor(bn).long())
    total = loss + reg * world.config["decay"]
    Recmodel.zero_grad()
    total.backward()
    meta["b_loss"], meta["b_reg"], meta["b_total"] = float(loss), float(reg), float(total)
    out["b_grad_user"] = Recmodel.embedding_user.weight.grad.numpy().copy()
    out["b_grad_item"] = Recmodel.embedding_item.weight.grad.numpy().copy()
    Recmodel.zero_grad()

    # --- epochs through the reference's own sampler/shuffle/stageOne ----------
    n_epochs = 2
    for e in range(1, n_epochs + 1):
        S, (su, sp_, sn), losses = ref_epoch(world, utils, dataset, bpr, B)
        out[f"S_epoch{e}"] = S
        out[f"shuf_users_epoch{e}"] = su.astype(np.int64)
        out[f"shuf_pos_epoch{e}"] = sp_.astype(np.int64)
        out[f"shuf_neg_epoch{e}"] = sn.astype(np.int64)
        out[f"losses_epoch{e}"] = np.asarray(losses, np.float64)
        meta[f"S_epoch{e}_sha256"], meta[f"S_epoch{e}_dtype"] = sha(S), str(S.dtype)
        out[f"P_user_epoch{e}"], out[f"P_item_epoch{e}"] = params(Recmodel)
        if e == 1:
            st = bpr.opt.state_dict()["state"]
            out["adam_m_user_epoch1"] = st[0]["exp_avg"].numpy().copy()
            out["adam_v_user_epoch1"] = st[0]["exp_avg_sq"].numpy().copy()
            out["adam_m_item_epoch1"] = st[1]["exp_avg"].numpy().copy()
            out["adam_v_item_epoch1"] = st[1]["exp_avg_sq"].numpy().copy()
            meta["adam_step_epoch1"] = int(st[0]["step"])
        meta[f"test_epoch{e}"] = ref_test(world, utils, Procedure, dataset, Recmodel, 100)
    if not full:
        # keep the fixture small: E0 stays whole (it is an input), every other
        # float table keeps every 8th row; integer arrays keep epoch 1 only
        meta["row_stride"] = 8
        for k in list(out.keys()):
            if out[k].dtype == np.float32 and out[k].ndim == 2 and not k.startswith("E0_"):
                out[k] = out[k][::8].copy()
            if k.endswith("epoch2") and k != "losses_epoch2":
                del out[k]
            elif k.startswith("adam_") or k in ("shuf_pos_epoch1", "shuf_neg_epoch1"):
                del out[k]
    np.savez_compressed(os.path.join(out_dir, "golden.npz"), **out)
    with open(os.path.join(out_dir, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, json.dumps(meta, sort_keys=True)[:600])
    shutil.rmtree(work, ignore_errors=True)


def gen_tiny_both_samplers(out_dir):
    """Extra: the python-mode sampler on tiny (fresh interpreter state needed for
    the numpy stream, so it runs the reference functions directly)."""
    work = tempfile.mkdtemp(prefix="golden_tinypy")
    make_tiny(work)
    world, dataloader, model, utils, Procedure = import_reference("tiny")
    dataset = dataloader.Loader(world.config, path=work)
    out = {}
    np.random.seed(2020)
    S1 = utils.UniformSample_original_python(dataset)
    idx = np.arange(len(S1)); np.random.shuffle(idx)
    S2 = utils.UniformSample_original_python(dataset)
    out["S_python_epoch1"], out["perm_after_epoch1"], out["S_python_epoch2"] = S1, idx, S2
    np.savez_compressed(os.path.join(out_dir, "golden_python_sampler.npz"), **out)
    shutil.rmtree(work, ignore_errors=True)


def gen_gowalla(out_dir, n_epochs, tag):
    os.makedirs(out_dir, exist_ok=True)
    work = tempfile.mkdtemp(prefix="golden_gowalla")
    make_gowalla(work)
    if not os.path.exists(os.path.join(out_dir, "gowalla.npz")):
        pack_dataset(work, os.path.join(out_dir, "gowalla.npz"))
    K, d, B = 3, 64, 2048
    (world, dataloader, model, utils, Procedure, dataset, Recmodel, bpr, torch) = \
        setup("gowalla", K, d, B, work, True)
    meta = {"dataset": "gowalla", "K": K, "d": d, "B": B, "sampler": "cpp", "seed": world.seed,
            "n_users": dataset.n_users, "m_items": dataset.m_items,
            "trainDataSize": dataset.trainDataSize, "torch": torch.__version__}
    out = {}
    ip, ix, dv = graph_csr(dataset)
    meta["adj_sha256"] = {"indptr": sha(ip), "indices": sha(ix), "data": sha(dv)}
    meta["adj_nnz"] = int(len(ix))
    out["adj_data_head"], out["adj_indices_head"] = dv[:256], ix[:256]
    pu, pi = params(Recmodel)
    meta["E0_sha256"] = {"user": sha(pu), "item": sha(pi)}
    out["E0_user_head"] = pu[:4]
    with torch.no_grad():
        au, ai = Recmodel.computer()
    out["computer_users_head"], out["computer_items_head"] = au[:8].numpy().copy(), ai[:8].numpy().copy()
    meta["computer_sum"] = [float(au.double().sum()), float(ai.double().sum())]
    meta["computer_abs_sum"] = [float(au.double().abs().sum()), float(ai.double().abs().sum())]
    t0 = time.time()
    meta["test_epoch0"] = ref_test(world, utils, Procedure, dataset, Recmodel, 2000)
    print("epoch0", meta["test_epoch0"], f"{time.time()-t0:.1f}s", flush=True)
    traj = []
    for e in range(1, n_epochs + 1):
        t0 = time.time()
        S, (su, sp_, sn), losses = ref_epoch(world, utils, dataset, bpr, B)
        dt = time.time() - t0
        rec = {"epoch": e, "steps": len(losses), "avg_loss": float(np.mean(losses)),
               "S_sha256": sha(S), "S_head": S[:3].tolist(), "S_tail": S[-1].tolist(),
               "shuf_users_head": su[:5].tolist(), "train_seconds": dt}
        if e <= 2:
            out[f"losses_epoch{e}"] = np.asarray(losses, np.float64)
        if e in (1, 2, 5, 10):
            rec["test"] = ref_test(world, utils, Procedure, dataset, Recmodel, 2000)
        traj.append(rec)
        print(json.dumps(rec), flush=True)
        meta["trajectory"] = traj
        with open(os.path.join(out_dir, f"golden_{tag}.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        np.savez_compressed(os.path.join(out_dir, f"golden_{tag}.npz"), **out)
    shutil.rmtree(work, ignore_errors=True)


def gen_gowalla_samplers(out_dir):
    """Sampler/shuffle hashes only (fast): cpp epochs 1-2, python epoch 1, shuffle."""
    work = tempfile.mkdtemp(prefix="golden_gowalla_s")
    make_gowalla(work)
    world, dataloader, model, utils, Procedure = import_reference("gowalla")
    dataset = dataloader.Loader(world.config, path=work)
    sampling = load_ref_sampler(utils, world)
    meta = {}
    os.makedirs(out_dir, exist_ok=True)
    for e in (1, 2):
        S = utils.UniformSample_original(dataset)
        meta[f"cpp_epoch{e}"] = {"sha256": sha(S), "shape": list(S.shape), "dtype": str(S.dtype),
                                 "head": S[:3].tolist(), "tail": S[-1].tolist()}
    sampling.seed(2020)
    meta["rand_mod_1000"] = [sampling.randint(1000) for _ in range(5)]
    np.random.seed(2020)
    idx = np.arange(806166); np.random.shuffle(idx)
    meta["shuffle_806166_head"] = idx[:5].tolist()
    meta["shuffle_806166_sha256"] = sha(idx)
    np.random.seed(2020)
    t0 = time.time()
    S = utils.UniformSample_original_python(dataset)
    meta["python_epoch1"] = {"sha256": sha(S), "shape": list(S.shape), "dtype": str(S.dtype),
                             "head": S[:3].tolist(), "tail": S[-1].tolist(), "seconds": time.time() - t0}
    with open(os.path.join(out_dir, "golden_samplers.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(meta)[:800])
    shutil.rmtree(work, ignore_errors=True)


def make_tiny_i2i(train_dir, out_npz):
    """Item-item matrix for the tiny dataset (our own construction, an INPUT fixture): co-occurrence
    R^T R without the diagonal, every row scaled to sum 1 (fp32 CSR)."""
    import scipy.sparse as sp
    rows, cols = [], []
    with open(os.path.join(train_dir, "train.txt")) as f:
        for l in f:
            c = l.split()
            rows += [int(c[0])] * (len(c) - 1)
            cols += [int(x) for x in c[1:]]
    n_users, m_items = max(rows) + 1, max(cols) + 1
    R = sp.csr_matrix((np.ones(len(rows), np.float32), (rows, cols)), shape=(n_users, m_items))
    C = (R.T @ R).tolil()
    C.setdiag(0)
    C = C.tocsr()
    C.eliminate_zeros()
    rs = np.asarray(C.sum(axis=1)).ravel()
    rs[rs == 0] = 1.0
    C = sp.diags((1.0 / rs).astype(np.float32)) @ C
    C = C.tocsr().astype(np.float32)
    C.sort_indices()
    sp.save_npz(out_npz, C)
    return C


def gen_variant(tag, gate, i2i, out_dir, K=3):
    """The fork's optional branches on the tiny dataset: popularity gate (model.py:66-96,139-157,176-181)
    and item-item smoothing (model.py:99-109,228-229), through the reference's own model / BPRLoss.
    K: propagation depth (targets tiny_gate_i2i_k1 / _k2 / _k4: the depths at which the product's fused step takes other
    code paths -- one row bitmap and k_finish at K = 1, alternating bitmaps at 2 / 4 -- pinned by the reference too)."""
    os.makedirs(out_dir, exist_ok=True)
    work = tempfile.mkdtemp(prefix="golden_" + tag)
    make_tiny(work)
    d, B = 64, 64
    import torch
    world, dataloader, model, utils, Procedure = import_reference("tiny")
    world.config["lightGCN_n_layers"], world.config["latent_dim_rec"], world.config["bpr_batch_size"] = K, d, B
    world.config["use_pop_gate"] = bool(gate)
    alpha = 0.3
    if i2i:
        i2i_path = os.path.join(out_dir, "i2i_tiny.npz")
        if not os.path.exists(i2i_path):
            make_tiny_i2i(work, i2i_path)
        world.config["use_item_item"], world.config["i2i_path"], world.config["i2i_alpha"] = True, i2i_path, alpha
    dataset = dataloader.Loader(world.config, path=work)
    utils.sample_ext = False
    utils.set_seed(world.seed)
    Recmodel = model.LightGCN(world.config, dataset).to(world.device)
    assert (Recmodel.i2i_adj is not None) == bool(i2i)
    bpr = utils.BPRLoss(Recmodel, world.config)
    meta = {"tag": tag, "K": K, "d": d, "B": B, "use_pop_gate": bool(gate), "use_item_item": bool(i2i),
            "i2i_alpha": alpha if i2i else 0.0, "seed": world.seed, "lr": world.config["lr"], "decay": world.config["decay"],
            "pop_hidden": world.config["pop_hidden"], "gate_hidden": world.config["gate_hidden"],
            "gate_entropy_coeff": world.config["gate_entropy_coeff"], "pop_gate_temp": world.config["pop_gate_temp"],
            "torch": torch.__version__}
    out = {}

    def dump(prefix):
        for k, v in Recmodel.state_dict().items():
            out[prefix + k] = v.detach().numpy().copy()
    dump("P0.")
    with torch.no_grad():
        au, ai = Recmodel.computer()
        out["computer_users"], out["computer_items"] = au.numpy().copy(), ai.numpy().copy()
        out["rating_users_0_9"] = Recmodel.getUsersRating(torch.arange(10)).numpy().copy()
    meta["test_epoch0"] = ref_test(world, utils, Procedure, dataset, Recmodel, 100)
    rng = np.random.Generator(np.random.PCG64(7))
    batches = []
    for _ in range(4):
        bu = rng.integers(0, dataset.n_users, B)
        bp = np.array([dataset.allPos[u][rng.integers(0, len(dataset.allPos[u]))] for u in bu])
        bn = rng.integers(0, dataset.m_items, B)
        batches.append((bu.astype(np.int64), bp.astype(np.int64), bn.astype(np.int64)))
    out["batches"] = np.asarray(batches)                               # [4, 3, B]
    bu, bp, bn = (torch.tensor(x).long() for x in batches[0])
    loss, reg = Recmodel.bpr_loss(bu, bp, bn)
    total = loss + reg * world.config["decay"]
    Recmodel.zero_grad()
    total.backward()
    meta["b_loss"], meta["b_reg"], meta["b_total"] = float(loss), float(reg), float(total)
    for k, prm in Recmodel.named_parameters():
        out["G0." + k] = prm.grad.numpy().copy()
    Recmodel.zero_grad()
    losses = []
    for (bu, bp, bn) in batches[1:]:
        losses.append(bpr.stageOne(torch.tensor(bu).long(), torch.tensor(bp).long(), torch.tensor(bn).long()))
    meta["step_losses"] = [float(x) for x in losses]
    dump("P3.")
    meta["test_after3"] = ref_test(world, utils, Procedure, dataset, Recmodel, 100)
    np.savez_compressed(os.path.join(out_dir, f"golden_{tag}.npz"), **out)
    with open(os.path.join(out_dir, f"golden_{tag}.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(tag, json.dumps(meta, sort_keys=True)[:500])
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["tiny", "lastfm", "gowalla"]
    # each target runs in a fresh interpreter: `world` parses argv at import and the
    # RNG streams must start clean
    if len(what) > 1:
        import subprocess
        for w in what:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), w])
        sys.exit(0)
    w = what[0]
    if w == "tiny":
        gen_small("tiny", 3, 64, 64, True, os.path.join(HERE, "tiny"), True)
    elif w == "tiny_py":
        gen_tiny_both_samplers(os.path.join(HERE, "tiny"))
    elif w == "lastfm":
        gen_small("lastfm", 2, 64, 2048, False, os.path.join(HERE, "lastfm"), False)
    elif w == "gowalla_samplers":
        gen_gowalla_samplers(os.path.join(HERE, "gowalla"))
    elif w == "gowalla":
        gen_gowalla(os.path.join(HERE, "gowalla"), 1, "short")
    elif w in ("tiny_gate", "tiny_i2i", "tiny_gate_i2i"):
        gen_variant(w[5:], "gate" in w, "i2i" in w, os.path.join(HERE, "tiny"))
    elif w in ("tiny_gate_i2i_k1", "tiny_gate_i2i_k2", "tiny_gate_i2i_k4"):
        gen_variant(w[5:], True, True, os.path.join(HERE, "tiny"), K=int(w[-1]))
    elif w == "gowalla_long":
        gen_gowalla(os.path.join(HERE, "gowalla"), 10, "long")
    else:
        raise SystemExit("unknown target " + w)
