"""Training / evaluation procedures, mirror of the reference's Procedure.py.

BPR_train_original keeps the reference's signature, return string and CSV side
effect (Procedure.py:30-83) -- its body is the epoch of main.py:215-225: native
sampler -> device upload -> native shuffle -> fused HIP epoch, with ONE host
read of the per-step losses at the end instead of a `.item()` per step.

Test keeps the semantics of Procedure.py:127-206 with the fork's broken
`minibatch` 1-tuple unwrapped: propagate once, score, mask train
positives with -(1<<10), top-k, per-user precision / recall / NDCG, mean.
"""
import csv
import multiprocessing
import os

import numpy as np
import torch

from . import world
from . import utils
from . import _lib
from .utils import timer

CORES = multiprocessing.cpu_count() // 2


def sample_epoch_to_device(dataset, device):
    """Sample (utils.UniformSample_original), upload, shuffle (utils.shuffle) -- the
    semantics of main.py:216-220.  Returns int32 device tensors users, pos, neg [T]."""
    S = utils.UniformSample_original(dataset)
    T = len(S)
    perm = utils.shuffle_indices(T)
    S32 = torch.from_numpy(np.ascontiguousarray(S[:, :3], dtype=np.int32)).to(device)
    permd = torch.from_numpy(perm).to(device)
    users = torch.empty(T, dtype=torch.int32, device=device)
    pos = torch.empty_like(users)
    neg = torch.empty_like(users)
    _lib.check(_lib.load().lgcn_apply_perm(_lib.tp(S32), 3, _lib.tp(permd), T, _lib.tp(users), _lib.tp(pos),
                                           _lib.tp(neg), _lib.current_stream()), "lgcn_apply_perm")
    return users, pos, neg


def BPR_train_original(dataset, recommend_model, loss_class, epoch, neg_k=1, w=None):
    """Procedure.py:30-83."""
    Recmodel = recommend_model
    Recmodel.train()
    bpr = loss_class
    B = world.config['bpr_batch_size']

    with timer(name="Sample"):
        users, posItems, negItems = sample_epoch_to_device(dataset, world.device)

    total_batch = len(users) // B + 1
    lr = bpr.opt.param_groups[0]['lr']
    losses = Recmodel.fused_epoch(users, posItems, negItems, B, lr=lr)     # [steps,3] on device
    step_losses = losses[:, 0].double().cpu().numpy()                      # the epoch's only sync
    Recmodel.check_device_errors()
    aver_loss = float(step_losses.sum())
    if world.tensorboard and w is not None:
        for batch_i, cri in enumerate(step_losses):
            w.add_scalar('BPRLoss/BPR', float(cri), epoch * total_batch + batch_i)
    aver_loss /= total_batch

    save_path = world.config.get('path', world.config.get('checkpoint_dir', './checkpoints'))
    os.makedirs(save_path, exist_ok=True)
    train_csv = os.path.join(save_path, 'train_epoch_metrics.csv')
    if not os.path.exists(train_csv):
        with open(train_csv, 'w', newline='') as f:
            csv.writer(f).writerow(['epoch', 'loss'])
    with open(train_csv, 'a', newline='') as f:
        csv.writer(f).writerow([epoch, aver_loss])

    time_info = timer.dict()
    timer.zero()
    return f"loss{aver_loss:.3f}-{time_info}"


def test_one_batch(X):
    """Procedure.py:89-121: metrics of ONE user from (top-k item ids, ground truth)."""
    sorted_items = X[0].cpu().numpy() if torch.is_tensor(X[0]) else np.asarray(X[0])
    groundTrue = X[1]
    if not isinstance(groundTrue, (list, set, tuple, np.ndarray)):
        groundTrue = [groundTrue]
    test_data = [groundTrue]
    r = np.expand_dims(utils.getLabel(groundTrue, sorted_items), axis=0)
    pre, recall, ndcg = [], [], []
    for k in world.topks:
        ret = utils.RecallPrecision_ATk(test_data, r, k)
        pre.append(ret['precision'])
        recall.append(ret['recall'])
        ndcg.append(utils.NDCGatK_r(test_data, r, k))
    return {'precision': np.array(pre), 'recall': np.array(recall), 'ndcg': np.array(ndcg)}


def _batch_metrics(hits, gt_len, topks):
    """Vectorised test_one_batch over a user batch: hits [Bu,maxK] 0/1, gt_len [Bu]."""
    out = {m: np.zeros((hits.shape[0], len(topks))) for m in ('precision', 'recall', 'ndcg')}
    for j, k in enumerate(topks):
        r = hits[:, :k].astype(np.float32)
        right = r.sum(1)
        out['precision'][:, j] = right / k
        out['recall'][:, j] = right / gt_len
        disc = 1. / np.log2(np.arange(2, k + 2))
        ideal = (np.arange(k)[None, :] < np.minimum(k, gt_len)[:, None]).astype(np.float64)
        idcg = (ideal * disc).sum(1)
        idcg[idcg == 0.] = 1.
        out['ndcg'][:, j] = (r * disc).sum(1) / idcg
    return out


def Test(dataset, Recmodel, epoch, w=None, multicore=0):
    """Procedure.py:127-206 (multicore is accepted and ignored: the reference creates a Pool
    and never uses it)."""
    u_batch_size = world.config['test_u_batch_size']
    testDict = dataset.testDict
    Recmodel = Recmodel.eval()
    max_K = max(world.topks)
    if hasattr(Recmodel, "invalidate_cache"):
        Recmodel.invalidate_cache()
    dev = world.device
    per_user = {m: [] for m in ('precision', 'recall', 'ndcg')}
    with torch.no_grad():
        users = list(testDict.keys())
        indptr, indices = dataset.pos_csr() if hasattr(dataset, 'pos_csr') else utils._pos_csr(dataset)
        for (batch_users,) in utils.minibatch(users, batch_size=u_batch_size):
            bu = np.asarray(batch_users, np.int64)
            batch_gpu = torch.from_numpy(bu).to(dev)
            rating_K = Recmodel.getUsersRating(batch_gpu)
            # mask seen interactions (Procedure.py:177-181)
            lens = (indptr[bu + 1] - indptr[bu]).astype(np.int64)
            ex_idx = np.repeat(np.arange(len(bu)), lens)
            ex_items = np.concatenate([indices[indptr[u]:indptr[u + 1]] for u in bu]) if lens.sum() else np.zeros(0, np.int64)
            rating_K[torch.from_numpy(ex_idx).to(dev), torch.from_numpy(ex_items.astype(np.int64)).to(dev)] = -(1 << 10)
            _, topk = torch.topk(rating_K, k=max_K)
            # labels: is top-k item in the user's test list?
            gt_len = np.array([len(testDict[u]) for u in batch_users], np.int64)
            gt_idx = np.repeat(np.arange(len(bu)), gt_len)
            gt_items = np.concatenate([np.asarray(testDict[u], np.int64) for u in batch_users])
            is_gt = torch.zeros(rating_K.shape, dtype=torch.bool, device=dev)
            is_gt[torch.from_numpy(gt_idx).to(dev), torch.from_numpy(gt_items).to(dev)] = True
            hits = is_gt.gather(1, topk).cpu().numpy()
            bm = _batch_metrics(hits, gt_len, world.topks)
            for m in per_user:
                per_user[m].append(bm[m])
    results = {m: np.mean(np.concatenate(per_user[m], 0), axis=0) for m in per_user}

    save_path = world.config.get('path', world.config.get('checkpoint_dir', './checkpoints'))
    os.makedirs(save_path, exist_ok=True)
    valid_csv = os.path.join(save_path, 'valid_epoch_metrics.csv')
    if not os.path.exists(valid_csv):
        with open(valid_csv, 'w', newline='') as f:
            csv.writer(f).writerow(['epoch', 'precision', 'recall', 'ndcg'])
    prec, rec, nd = float(results['precision'][0]), float(results['recall'][0]), float(results['ndcg'][0])
    with open(valid_csv, 'a', newline='') as f:
        csv.writer(f).writerow([epoch, prec, rec, nd])
    if world.tensorboard and w is not None:
        for m, tag in (('recall', 'Recall'), ('precision', 'Precision'), ('ndcg', 'NDCG')):
            w.add_scalars(f'Test/{tag}@{world.topks}',
                          {str(world.topks[i]): results[m][i] for i in range(len(world.topks))}, epoch)
    print(results)
    return results
