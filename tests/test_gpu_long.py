"""The Recall@20 half of BASELINE's metric at the reference's own horizon (-m gpu; ~2 minutes of GPU: LGCN_SKIP_LONG=1 skips it).

What the reference holds for this configuration (Gowalla, 3 layers, d = 64, batch 2048, seed 2020) is a RECORDED 1000-epoch run
(code/runs/07-10-17h52m32s--lgn, parsed into tests/golden/gowalla/recorded_trajectory.json by make_recorded_trajectory.py)
and the table of LightGCN_work/README.md:88-95 (layer 3: 0.1824 / 0.1547 / 0.05589).  Both were produced by UPSTREAM LightGCN's
loss (L2 term on the embedding tables' own rows).  The fork under /root/reference moved that term to the PROPAGATED rows
(model.py:173) -- the loss the fixtures of make_golden.py pin and the default here (--reg_rows propagated): it matches the
imported reference to 1e-17 on Recall@20 after 10 epochs, and run to the horizon it OVERFITS (peak 0.1713 at epoch 230, 0.158 at
epoch 1000: profiles/r04/gowalla_1000ep_{fp32,bf16}.json) -- that is the reference's code, not a defect of this build.  With
--reg_rows ego the same kernels train upstream's loss, and THAT is what the recorded run can be compared with: a neighbour
(another machine and torch, unknown sampler build), so inside a band, not to 1e-4.
"""
import contextlib
import io
import json
import os
import sys

import pytest
import torch

from conftest import GOLDEN, REPO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _train(pkg, data_dir, act, reg_rows, epochs, test_every=10):
    w = pkg.world
    w.configure(["--dataset", "gowalla", "--tensorboard", "0", "--act_dtype", act, "--prefetch_epoch", "1", "--reg_rows", reg_rows,
                 "--checkpoint_dir", os.path.join(data_dir, "ckpt")])
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        ds = pkg.dataloader.Loader(w.config, path=data_dir)
        pkg.sampling.seed(w.seed); pkg.utils.set_seed(w.seed)
        m = pkg.model.LightGCN(w.config, ds).to(DEV)
        bpr = pkg.utils.BPRLoss(m, w.config)
        for e in range(1, epochs + 1):
            pkg.Procedure.BPR_train_original(ds, m, bpr, e)
            if e % test_every == 0:
                r = pkg.Procedure.Test(ds, m, e)
                out[e] = {k: float(v[0]) for k, v in r.items()}
    ds._lgcn_next_epoch = None
    m.check_device_errors()
    m._drop_device_state()
    return out


@pytest.mark.skipif(os.environ.get("LGCN_SKIP_LONG") == "1", reason="LGCN_SKIP_LONG=1")
def test_gowalla_1000_epochs_upstream_loss_vs_recorded_run(pkg, tmp_path):
    sys.path.insert(0, REPO)
    from bench import materialize_gowalla, GOWALLA_NPZ
    rec = json.load(open(os.path.join(GOLDEN, "gowalla", "recorded_trajectory.json")))
    rec_at = dict(zip(rec["epochs"], rec["recall"]))
    assert rec["points"] == 100 and abs(rec_at[990] - 0.182116) < 1e-6 and abs(max(rec["recall"]) - 0.182504) < 1e-6
    d = materialize_gowalla(GOWALLA_NPZ, os.path.join(str(tmp_path), "gowalla"))
    runs = {act: _train(pkg, d, act, "ego", 1000) for act in ("fp32", "bf16")}
    for act, tr in runs.items():
        recall = {e: v["recall"] for e, v in tr.items()}
        worst = max(abs(recall[e] - rec_at[e]) for e in rec_at if e >= 100)
        print(f"[1000 epochs, upstream loss, {act}] Recall@20 epoch 990 {recall[990]:.6f} (recorded {rec_at[990]:.6f}), epoch 1000 {recall[1000]:.6f}, "
              f"max {max(recall.values()):.6f} (recorded {max(rec['recall']):.6f}; README 0.1824), max |diff| from epoch 100 on {worst:.2e}, "
              f"first epoch >= 0.18: {min(e for e, v in recall.items() if v >= 0.18)} (recorded {rec['summary']['first_epoch_recall_ge_0.18']})")
        # measured (profiles/r04/gowalla_1000ep_ego_*.json): -9.3e-5 at epoch 990, max 0.18242, worst point 1.3e-3
        assert abs(recall[990] - rec_at[990]) <= 1e-3, (act, recall[990])
        assert max(recall.values()) >= 0.1815 and abs(max(recall.values()) - 0.1824) <= 1e-3, (act, max(recall.values()))
        assert worst <= 2.5e-3, (act, worst)
        assert abs(recall[10] - rec_at[10]) <= 5e-4                         # (recorded 0.120140)
        assert tr[1000]["ndcg"] > 0.153 and tr[1000]["precision"] > 0.0555   # README: 0.1547 / 0.05589
    # bf16 activation storage (what BASELINE configs[1] names) against fp32 at the horizon: inside the recorded run's own
    # epoch-to-epoch spread of ~3e-4 (its last ten points span 0.18142 .. 0.18250); measured 4.7e-5
    assert abs(runs["fp32"][1000]["recall"] - runs["bf16"][1000]["recall"]) <= 3e-4
    assert abs(runs["fp32"][1000]["ndcg"] - runs["bf16"][1000]["ndcg"]) <= 3e-4
