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
           ("lgcn_device.hip", "lgcn_eval.hip", "lgcn_sampler.hip", "lgcn_shuffle.hip", "lgcn_ids.hip", "lgcn_dp_loopback.hip", "lgcn_dp.cpp", "lgcn_host.cpp")]
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


SAMPLING_SRC = os.path.join(PKG_DIR, "csrc", "sampling_module.cpp")


def sampling_module_path():
    import sysconfig
    return os.path.join(PKG_DIR, "sources", "sampling" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_sampling_module(force=False, verbose=False):
    """Compile csrc/sampling_module.cpp (pybind11) into sources/sampling<ext>.so: the reference's native plugin as
    a compiled extension module named `sampling` (sources/sampling.cpp:95-106), a binding over liblgcn_hip.so's C ABI
    (linked with an $ORIGIN rpath, so the pair moves together).  g++ only -- no hipcc, no GPU.  Returns the path, or
    None when pybind11 / Python headers are not available (the ctypes binding sampling.py serves the same names)."""
    import sysconfig
    try:
        import pybind11
    except ImportError:
        return None
    out = sampling_module_path()
    lib = build()
    stamp = out + ".srchash"
    h = hashlib.sha256()
    for f in (SAMPLING_SRC, HEADER):
        with open(f, "rb") as fh:
            h.update(fh.read())
    want = h.hexdigest()
    if not force and os.path.exists(out):
        try:
            if open(stamp).read().strip() == want:
                return out
        except OSError:
            pass
    if must_not_build():
        return out if os.path.exists(out) else None
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    inc_py = sysconfig.get_paths().get("include")
    if cxx is None or not inc_py or not os.path.exists(os.path.join(inc_py, "Python.h")):
        return None
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            tmp = f"{out}.tmp.{os.getpid()}"
            cmd = [cxx, "-O2", "-shared", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-I" + inc_py, "-I" + pybind11.get_include(),
                   "-I" + os.path.join(REPO_DIR, "include"), SAMPLING_SRC, "-o", tmp,
                   "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib), "-Wl,-rpath,$ORIGIN/.."]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, out)
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
            with open(stamp, "w") as f:
                f.write(want)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return out


def load_sampling_module():
    """Import the compiled `sampling` extension by path -- what utils.py:25-34 does through cppimport."""
    import importlib.util
    path = build_sampling_module()
    if path is None or not os.path.exists(path):
        raise ImportError("compiled sampling module not available (pybind11 / Python headers / g++ missing)")
    from . import _lib
    _lib.load()                       # map liblgcn_hip.so first: ONE library instance, one rand() stream for both bindings
    spec = importlib.util.spec_from_file_location("sampling", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_sampling_module(force=True, verbose=True))
