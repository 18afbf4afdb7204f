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
    S32 = None
    if (torch.device(device).type == 'cuda' and int(world.config.get('gpu_sampler', 1)) and utils.sample_ext
            and utils.sampler_mode(dataset) == 'cpp'):
        # the native sampler's stream and rows, produced on the device (sampling.cpp:27-56, bit-exact)
        try:
            S32 = utils.sampling.sample_negative_device(dataset.n_users, dataset.m_items, dataset.trainDataSize,
                                                        utils._pos_csr(dataset), device)
        except _lib.LgcnError as e:
            # rc 5: a dense dataset rejected more negatives than the device stream's margin (T/50 + 65536 draws) holds.
            # The host generator has not been advanced yet, so the bit-exact host sampler takes this epoch over.
            if "(rc=5)" not in str(e):
                raise
            world.cprint("[sampler] GPU sampler margin exceeded (dense dataset): host sampler for this epoch")
    if S32 is not None:
        T = int(S32.shape[0])
    else:
        S = utils.UniformSample_original(dataset)
        T = len(S)
        S32 = torch.from_numpy(np.ascontiguousarray(S[:, :3], dtype=np.int32)).to(device)
    permd = None
    if torch.device(device).type == 'cuda' and int(world.config.get('gpu_shuffle', 1)) and 2 <= T < (1 << 31) - 16:
        # utils.shuffle's permutation (np.random.shuffle of arange: utils.py:148-149) from the same MT19937 stream, on the device
        permd = utils.shuffle_indices_device(T, device)
    if permd is None:
        permd = torch.from_numpy(utils.shuffle_indices(T)).to(device)
    users = torch.empty(T, dtype=torch.int32, device=device)
    pos = torch.empty_like(users)
    neg = torch.empty_like(users)
    _lib.check(_lib.load().lgcn_apply_perm(_lib.tp(S32), 3, _lib.tp(permd), T, _lib.tp(users), _lib.tp(pos),
                                           _lib.tp(neg), _lib.current_stream()), "lgcn_apply_perm")
    return users, pos, neg



def _prefetch_next_epoch(dataset):
    """Sample + shuffle + upload the NEXT epoch on a side stream while the GPU still runs this one.
    The sampler / shuffle streams do not depend on training, so the triplets are the same ones the
    reference would draw at the start of the next epoch -- provided nothing else consumes the two
    RNG streams in between (--prefetch_epoch 0 restores the strict order)."""
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        u, p, n = sample_epoch_to_device(dataset, world.device)
        ev = torch.cuda.Event()
        ev.record(side)
    dataset._lgcn_next_epoch = (u, p, n, ev)       # kept on the dataset itself (an id()-keyed table could hand a recycled id another dataset's epoch)


def BPR_train_original(dataset, recommend_model, loss_class, epoch, neg_k=1, w=None):
    """Procedure.py:30-83."""
    Recmodel = recommend_model
    Recmodel.train()
    bpr = loss_class
    B = world.config['bpr_batch_size']
    prefetch = bool(world.config.get('prefetch_epoch', 1)) and world.device.type == 'cuda'

    with timer(name="Sample"):
        if prefetch and getattr(dataset, '_lgcn_next_epoch', None) is not None:
            users, posItems, negItems, ev = dataset._lgcn_next_epoch
            dataset._lgcn_next_epoch = None
            torch.cuda.current_stream().wait_event(ev)
        else:
            users, posItems, negItems = sample_epoch_to_device(dataset, world.device)

    total_batch = len(users) // B + 1
    lr = bpr.opt.param_groups[0]['lr']
    if getattr(Recmodel, 'has_variants', False) and not getattr(Recmodel, 'fused_variants', False):
        # popularity gate / item-item smoothing on the autograd path: the reference's own loop (Procedure.py:56-66) over stageOne
        was_lazy, bpr.lazy = bpr.lazy, True
        per = [bpr.stageOne(users[t:t + B], posItems[t:t + B], negItems[t:t + B]) for t in range(0, len(users), B)]
        bpr.lazy = was_lazy
        losses = torch.stack(per).reshape(-1, 1)
    else:
        losses = Recmodel.fused_epoch(users, posItems, negItems, B, lr=lr)     # [steps,3] on device
    if prefetch:
        with timer(name="Sample"):
            _prefetch_next_epoch(dataset)                                   # host work under the GPU's epoch
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


class _EvalIndex:
    """Device-resident CSR of the train positives and of the test lists, built once per dataset:
    the per-batch Python loops of Procedure.py:170-181 become two index_put calls."""

    def __init__(self, dataset, dev):
        indptr, indices = dataset.pos_csr() if hasattr(dataset, 'pos_csr') else utils._pos_csr(dataset)
        self.users = np.fromiter(dataset.testDict.keys(), dtype=np.int64, count=len(dataset.testDict))
        self.train_ptr = torch.from_numpy(np.ascontiguousarray(indptr, np.int64)).to(dev)
        self.train_idx = torch.from_numpy(np.ascontiguousarray(indices, np.int64)).to(dev)
        lens = np.fromiter((len(dataset.testDict[u]) for u in self.users.tolist()), dtype=np.int64, count=len(self.users))
        ptr = np.zeros(len(self.users) + 1, np.int64); np.cumsum(lens, out=ptr[1:])
        items = np.concatenate([np.asarray(dataset.testDict[u], np.int64) for u in self.users.tolist()]) if len(self.users) else np.zeros(0, np.int64)
        self.test_len = lens
        self.test_ptr = torch.from_numpy(ptr).to(dev)
        self.test_idx = torch.from_numpy(items).to(dev)
        # what the fused kernels take: int32 ids, the test lists ascending per user
        self.users32 = torch.from_numpy(self.users.astype(np.int32)).to(dev)
        self.train_idx32 = self.train_idx.to(torch.int32)
        order = np.lexsort((items, np.repeat(np.arange(len(self.users)), lens)))
        self.test_sorted32 = torch.from_numpy(items[order].astype(np.int32)).to(dev)
        # the train-positive masks of the evaluated users, tile-major (lgcn_eval_build_masks): built once per dataset, so
        # the item sweep of every Test holds no data-dependent load.  Only while they stay small (<= 1 GiB); else the sweep
        # walks the train CSR with a cursor as before.
        self.masks = None
        if torch.device(dev).type == 'cuda' and len(self.users):
            lib = _lib.load()
            words = int(lib.lgcn_eval_mask_words(int(dataset.m_items), len(self.users)))
            if 0 < words <= (1 << 28) and int(world.config.get('eval_masks', 1)):
                self.masks = torch.empty(words, dtype=torch.int32, device=dev)
                _lib.check(lib.lgcn_eval_build_masks(_lib.tp(self.users32), len(self.users), _lib.tp(self.train_ptr), _lib.tp(self.train_idx32),
                                                     int(dataset.m_items), _lib.tp(self.masks), _lib.current_stream()), "lgcn_eval_build_masks")

    @staticmethod
    def _expand(ptr, rows):
        """(local row id, position) pairs of the CSR rows `rows`."""
        start, end = ptr[rows], ptr[rows + 1]
        lens = end - start
        local = torch.repeat_interleave(torch.arange(len(rows), device=ptr.device), lens)
        offs = torch.arange(int(lens.sum()), device=ptr.device) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens)
        return local, torch.repeat_interleave(start, lens) + offs


def _test_fused(Recmodel, ev, max_K):
    """Procedure.py:162-192 in two launches: lgcn_eval_topk (scores on the matrix cores + train mask +
    top-K, no score matrix in memory) and lgcn_eval_metrics (hits, precision / recall / NDCG, sums)."""
    lib = _lib.load()
    E = Recmodel.rating_table()
    n = len(ev.users)
    dev = E.device
    topk = torch.empty(n, max_K, dtype=torch.int32, device=dev)
    _lib.check(lib.lgcn_eval_topk_masked(_lib.tp(E), Recmodel.n_users, Recmodel.m_items, int(E.shape[1]),      # (a column shard scores with the gathered table)
                                         _lib.tp(ev.users32), n, _lib.tp(ev.train_ptr), _lib.tp(ev.train_idx32),
                                         max_K, _lib.tp(topk), None, _lib.tp(ev.masks) if ev.masks is not None else None,
                                         _lib.current_stream()), "lgcn_eval_topk")
    ks = torch.tensor(list(world.topks), dtype=torch.int32)
    per_user = torch.empty(n, 3 * len(ks), dtype=torch.float64, device=dev)
    sums = torch.empty(3 * len(ks), dtype=torch.float64, device=dev)
    _lib.check(lib.lgcn_eval_metrics(_lib.tp(topk), n, max_K, _lib.tp(ev.test_ptr), _lib.tp(ev.test_sorted32),
                                     _lib.tp(ks), len(ks), _lib.tp(per_user), _lib.tp(sums), _lib.current_stream()),
               "lgcn_eval_metrics")
    m = (sums.cpu().numpy() / max(n, 1)).reshape(3, len(ks))
    return {'precision': m[0], 'recall': m[1], 'ndcg': m[2]}, topk


def Test(dataset, Recmodel, epoch, w=None, multicore=0):
    """Procedure.py:127-206 (multicore is accepted and ignored: the reference creates a Pool
    and never uses it).  Per-user results do not depend on the user batch size, so
    several `test_u_batch_size` batches are scored per launch (bounded by ~1 GiB of scores)."""
    u_batch_size = world.config['test_u_batch_size']
    Recmodel = Recmodel.eval()
    max_K = max(world.topks)
    if hasattr(Recmodel, "invalidate_cache"):
        Recmodel.invalidate_cache()
    dev = world.device
    ev = getattr(dataset, '_lgcn_eval_index', None)
    if ev is None or ev.train_ptr.device != torch.empty(0, device=dev).device:
        ev = _EvalIndex(dataset, dev)
        try:
            dataset._lgcn_eval_index = ev
        except Exception:
            pass
    fused = max_K <= 64 and int(world.config.get('eval_fused', 1)) and hasattr(Recmodel, 'propagated_table')
    if fused:
        with torch.no_grad():
            results, _ = _test_fused(Recmodel, ev, max_K)
        return _finish_test(results, epoch, w)
    per_user = {m: [] for m in ('precision', 'recall', 'ndcg')}
    chunk = max(u_batch_size, min(8192, (1 << 28) // max(1, dataset.m_items)) // u_batch_size * u_batch_size)
    with torch.no_grad():
        users_dev = torch.from_numpy(ev.users).to(dev)
        for s in range(0, len(ev.users), chunk):
            batch_gpu = users_dev[s:s + chunk]
            rating_K = Recmodel.getUsersRating(batch_gpu)
            ex_row, ex_pos = ev._expand(ev.train_ptr, batch_gpu)           # mask seen interactions
            rating_K[ex_row, ev.train_idx[ex_pos]] = -(1 << 10)             # (Procedure.py:177-181)
            _, topk = torch.topk(rating_K, k=max_K)
            rows = torch.arange(s, min(s + chunk, len(ev.users)), device=dev)
            gt_row, gt_pos = ev._expand(ev.test_ptr, rows)
            is_gt = torch.zeros(rating_K.shape, dtype=torch.bool, device=dev)
            is_gt[gt_row, ev.test_idx[gt_pos]] = True
            hits = is_gt.gather(1, topk).cpu().numpy()
            bm = _batch_metrics(hits, ev.test_len[s:s + chunk], world.topks)
            for m in per_user:
                per_user[m].append(bm[m])
    results = {m: np.mean(np.concatenate(per_user[m], 0), axis=0) for m in per_user}
    return _finish_test(results, epoch, w)


def _finish_test(results, epoch, w):
    """CSV / tensorboard side effects and the return value of Procedure.py:195-206."""
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
