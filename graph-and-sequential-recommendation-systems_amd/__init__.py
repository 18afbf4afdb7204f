"""MI355X-native LightGCN / BPR training hot path.

Drop-in for the hot path of saamiya225/Graph-and-sequential-recommendation-systems
(LightGCN_work/code): the modules below mirror the reference's own module names
and call signatures, and route all device arithmetic through the C ABI of
include/lgcn_hip.h (hand-written gfx950 kernels in csrc/).

    world / parse / register   config surface (world.py, parse.py, register.py)
    dataloader                 BasicDataset / Loader, same files + graph format
    model                      LightGCN (computer, bpr_loss, getUsersRating, ...)
    utils                      BPRLoss.stageOne, samplers, shuffle, minibatch, metrics
    Procedure                  BPR_train_original, Test
    sampling                   the pybind11 `sampling` plugin's four functions
    parallel                   batch-sharded data parallel step over RCCL
    reorder                    locality ordering of graph rows for the SpMM kernels

The directory name contains '-', so import it with
    importlib.import_module("graph-and-sequential-recommendation-systems_amd")
Sub-modules are imported lazily: `world` parses sys.argv when first imported,
exactly like the reference's world.py.
"""
import importlib as _importlib

__all__ = ["build", "_lib", "world", "parse", "register", "dataloader", "model", "utils",
           "Procedure", "sampling", "parallel", "reorder", "synthetic"]


def __getattr__(name):
    if name in __all__:
        return _importlib.import_module("." + name, __name__)
    raise AttributeError(name)
