"""Body of tests/test_host.py::test_compiled_sampling_module_matches_reference_plugin (own process, see there)."""
import importlib
import os
import shutil
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
work = sys.argv[1]
sys.argv = [sys.argv[0]]
from conftest import GoldenSet, PKG_NAME      # noqa: E402

pkg = importlib.import_module(PKG_NAME)
tiny = GoldenSet("tiny")
for f in ("train.txt", "test.txt"):
    shutil.copyfile(os.path.join(tiny.dir, f), os.path.join(work, f))
pkg.world.configure(["--dataset", "tiny", "--tensorboard", "0"])
ds = pkg.dataloader.Loader(pkg.world.config, path=work)

mod = pkg.utils.compiled_sampling_module()
assert mod.__name__ == "sampling" and mod.__doc__ == "example plugin" and mod.abi_version == pkg._lib.ABI_VERSION
assert {n for n in dir(mod) if not n.startswith("_")} == {"randint", "seed", "sample_negative", "sample_negative_ByUser", "abi_version"}
mod.seed(pkg.world.seed)
for e in (1, 2):
    S = mod.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.allPos, 1)      # the reference's call (utils.py:77)
    assert S.dtype == np.int32 and S.flags["C_CONTIGUOUS"] and np.array_equal(S, tiny.z[f"S_epoch{e}"])
# both bindings drive the same generator: continue the stream through the other one
pkg.sampling.seed(11); a1 = pkg.sampling.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.allPos, 1)
a2 = mod.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.pos_csr(), 1); r3 = pkg.sampling.randint(10**6)
mod.seed(11); b1 = mod.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.pos_csr(), 1)
b2 = pkg.sampling.sample_negative(ds.n_users, ds.m_items, ds.trainDataSize, ds.allPos, 1); q3 = mod.randint(10**6)
assert np.array_equal(a1, b1) and np.array_equal(a2, b2) and r3 == q3
users = [3, 0, 7, 7, 1]
mod.seed(5); u1 = mod.sample_negative_ByUser(users, ds.m_items, ds.allPos, 2)
pkg.sampling.seed(5); u2 = pkg.sampling.sample_negative_ByUser(users, ds.m_items, ds.allPos, 2)
assert u1.shape == (5, 4) and u1.dtype == np.int32 and np.array_equal(u1, u2)
for bad in (lambda: mod.sample_negative(2, 5, 4, [np.array([1], np.int32), np.array([], np.int32)], 1),
            lambda: mod.sample_negative_ByUser([9], 5, [np.array([1], np.int32)], 1)):
    try:
        bad()
    except ValueError:
        pass
    else:
        raise AssertionError("ValueError expected")
print("OK")
