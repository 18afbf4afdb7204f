"""Experiment driver: gather ceiling on the Gowalla graph (see gather_ceiling.hip)."""
import ctypes, importlib, os, subprocess, sys, io, contextlib
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from bench import materialize_gowalla, GOWALLA_NPZ
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "libexp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
                       os.path.join(here, "gather_ceiling.hip"), "-o", so])
lib = ctypes.CDLL(so)
sys.argv = [sys.argv[0]]
pkg = importlib.import_module("graph-and-sequential-recommendation-systems_amd")
d = materialize_gowalla(GOWALLA_NPZ, "/tmp/lgcn_exp_gowalla")
pkg.world.configure(["--dataset", "gowalla"])
with contextlib.redirect_stdout(io.StringIO()):
    ds = pkg.dataloader.Loader(pkg.world.config, path=d)
adj = ds.getSparseGraphCSR()
dev = "cuda:0"
idx = torch.from_numpy(adj.indices).to(dev); val = torch.from_numpy(adj.data).to(dev)
N, nnz = adj.shape[0], adj.nnz
vp = ctypes.c_void_p
for order in ("csr", "random"):
    ix = idx if order == "csr" else idx[torch.randperm(nnz, device=dev)]
    for bf in (0, 1):
        X = (torch.randn(N, 64, device=dev) * 0.1).to(torch.bfloat16 if bf else torch.float32)
        out = torch.empty(((nnz + 63) // 64 + 4) * 64, device=dev)
        for U in (4, 8, 16):
            st = vp(torch.cuda.current_stream().cuda_stream)
            f = lambda: lib.exp_tile_gather(vp(ix.data_ptr()), vp(val.data_ptr()), ctypes.c_int64(nnz), vp(X.data_ptr()), bf, U, vp(out.data_ptr()), st)
            for _ in range(5): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 50 * 1e3
            gb = nnz * 64 * (2 if bf else 4) / us / 1e3
            print(f"order={order:6s} dtype={'bf16' if bf else 'fp32'} U={U:2d}: {us:7.2f} us  gather {gb:7.1f} GB/s")
