"""Build liblgcn_hip.so (the C-ABI library of include/lgcn_hip.h) in-tree with hipcc
for gfx950.  hipcc cross-compiles without a GPU; the .so travels with the tree."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
SOURCES = [os.path.join(PKG_DIR, "csrc", "lgcn_device.hip"), os.path.join(PKG_DIR, "csrc", "lgcn_host.cpp")]
HEADER = os.path.join(REPO_DIR, "include", "lgcn_hip.h")
LIB_PATH = os.environ.get("LGCN_LIB_PATH") or os.path.join(PKG_DIR, "liblgcn_hip.so")   # env: pick a tuning variant


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in SOURCES + [HEADER])


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source for gfx950 into liblgcn_hip.so.  Returns the path."""
    global LIB_PATH
    if out is not None:
        LIB_PATH = out
    if not force and not is_stale():
        return LIB_PATH
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build liblgcn_hip.so (set HIPCC or install ROCm)")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
           "-I" + os.path.join(REPO_DIR, "include"), *extra_flags, *SOURCES, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
