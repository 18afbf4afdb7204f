#!/usr/bin/env python3
"""What ONE rank of a column-sharded run computes (shard='cols'): the Gowalla step at the width a rank holds (d / W), through the
column-sharded C loop on a one-rank loopback communicator (its all-reduce is a local copy: the collective's cost is NOT in here).
W = 1 (d = 64) next to W = 2 (d = 32): the compute-side ceiling of that mode's strong scaling."""
import contextlib, ctypes as C, importlib, io, json, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from bench import materialize_gowalla, GOWALLA_NPZ
sys.argv = [sys.argv[0]]
import torch
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
L, lib = pkg._lib, pkg._lib.load()
dev = torch.device("cuda", 0)
d_dir = materialize_gowalla(GOWALLA_NPZ, "/tmp/lgcn_cols_gowalla")
out = {}
for act in ("fp32", "bf16"):
    for W in (1, 2):
        w = pkg.world
        w.configure(["--dataset", "gowalla", "--tensorboard", "0", "--act_dtype", act, "--recdim", "64"])
        with contextlib.redirect_stdout(io.StringIO()):
            ds = pkg.dataloader.Loader(w.config, path=d_dir)
            pkg.sampling.seed(2020); pkg.utils.set_seed(2020)
            m = pkg.parallel.column_shard(pkg.model.LightGCN, w.config, ds, W, 0, dev)
            users, pos, neg = pkg.Procedure.sample_epoch_to_device(ds, dev)
        B, steps = 2048, 394
        T = steps * B
        st = m._state(max_batch=B, need_ctx=True, dp_world=1)
        comms = (C.c_void_p * 1)()
        L.check(lib.lgcn_dp_init_loopback(1, comms), "loopback")
        losses = torch.empty(steps, 3, device=dev)
        def epoch():
            L.check(lib.lgcn_train_epoch_dp(st['ctx'], comms[0], L.tp(users), L.tp(pos), L.tp(neg), T, B, 3, None, None, L.tp(losses), L.current_stream()), "epoch")
        epoch(); torch.cuda.synchronize()
        t0 = time.perf_counter(); epoch(); epoch(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        plain = m.fused_epoch(users[:T], pos[:T], neg[:T], B); torch.cuda.synchronize()
        t0 = time.perf_counter(); m.fused_epoch(users[:T], pos[:T], neg[:T], B); torch.cuda.synchronize(); dtp = time.perf_counter() - t0
        out[f"{act}_W{W}_d{m.latent_dim}"] = {"cols_loop_steps_per_sec": steps / dt, "us_per_step": 1e6 * dt / steps, "plain_epoch_steps_per_sec_same_width": steps / dtp}
        lib.lgcn_dp_destroy(comms[0]); m._drop_device_state(); del m
print(json.dumps(out, indent=1))
