"""Build liblgcn_hip.so (the C-ABI library of include/lgcn_hip.h) in-tree with hipcc
for gfx950.  hipcc cross-compiles without a GPU; the .so travels with the tree.

Staleness is decided by a hash of the sources stored next to the library
(`liblgcn_hip.so.srchash`), not by mtimes: a copied tree keeps no useful mtimes.
The build itself runs under an exclusive file lock with a per-process temporary
name, so concurrent builders (ranks of one job on a fresh checkout) cannot
clobber each other's output."""
import fcntl
import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
SOURCES = [os.path.join(PKG_DIR, "csrc", f) for f in
           ("lgcn_device.hip", "lgcn_eval.hip", "lgcn_sampler.hip", "lgcn_dp.cpp", "lgcn_host.cpp")]
HEADER = os.path.join(REPO_DIR, "include", "lgcn_hip.h")
LIB_PATH = os.environ.get("LGCN_LIB_PATH") or os.path.join(PKG_DIR, "liblgcn_hip.so")   # env: pick a tuning variant
BASE_FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17"]


def sources():
    return [s for s in SOURCES if os.path.exists(s)]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def source_hash(extra_flags=()):
    h = hashlib.sha256()
    for f in sources() + [HEADER]:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(list(BASE_FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()


def kernel_hash():
    """Hash of the source of the training kernels only (csrc/lgcn_device.hip): what a PMC measurement of
    k_spmm stays valid for."""
    with open(os.path.join(PKG_DIR, "csrc", "lgcn_device.hip"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def _stamp_path(lib=None):
    return (lib or LIB_PATH) + ".srchash"


def is_stale(lib=None):
    lib = lib or LIB_PATH
    if not os.path.exists(lib):
        return True
    try:
        with open(_stamp_path(lib)) as f:
            return f.read().strip() != source_hash()
    except OSError:
        return True


def must_not_build():
    """True inside a rank of a distributed job or under a profiler's preload: there the library
    has to exist already (N ranks compiling at once, or hipcc started as a child of a profiled
    process, are both wrong)."""
    if os.environ.get("LGCN_ALLOW_BUILD") == "1":
        return False
    if any(os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK")):
        return True
    pre = os.environ.get("LD_PRELOAD", "")
    return any(k in pre for k in ("rocprof", "roctracer", "rocprofiler")) or bool(os.environ.get("ROCPROFILER_LIBRARY_CTOR"))


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source for gfx950 into liblgcn_hip.so.  Returns the path."""
    global LIB_PATH
    if out is not None:
        LIB_PATH = out
    lib = LIB_PATH
    if not force and not is_stale(lib):
        return lib
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build liblgcn_hip.so (set HIPCC or install ROCm)")
    want = source_hash(extra_flags)
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not is_stale(lib):          # someone else built it while we waited
                return lib
            tmp = f"{lib}.tmp.{os.getpid()}"
            cmd = [hipcc, *BASE_FLAGS, "-I" + os.path.join(REPO_DIR, "include"), *extra_flags, *sources(),
                   "-ldl", "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, lib)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
            with open(_stamp_path(lib) + f".{os.getpid()}", "w") as f:
                f.write(want if not extra_flags else "variant:" + want)
            os.replace(_stamp_path(lib) + f".{os.getpid()}", _stamp_path(lib))
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
