"""bench.py launch plumbing without a GPU: `python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must
start its own two ranks as a CHILD job before anything touches the GPU, rendezvous on 127.0.0.1, run
its barrier / max-over-ranks reduction and have rank 0 print ONE JSON line, rc 0.  With
LGCN_BENCH_DRYRUN=1 the ranks stop right before the first GPU call (gloo stands in for RCCL)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = dict(os.environ, LGCN_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


def test_bench_self_launches_two_ranks():
    p = _run(["--gpus", "2", "--steps", "7", "--warmup", "2", "--workload", "amazon-book-shaped"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] and j["n_gpus"] == 2 and j["max_rank_plus_1"] == 2.0
    assert j["steps"] == 7 and j["warmup"] == 2 and j["workload"] == "amazon-book-shaped" and j["act_dtype"] == "fp32"
    # default = strong scaling: the reference's ONE global batch of 2048 is sharded
    assert j["scaling"] == "strong" and j["global_batch"] == 2048
    # the ranks the collectives really span (an all-reduce of 1 per rank), not WORLD_SIZE; the 400-step steady-state region
    assert j["rccl_ranks_observed"] == 2 and j["steady_state_steps"] == 400


def test_bench_weak_scaling_flag():
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", "weak"])
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["scaling"] == "weak" and j["global_batch"] == 2 * 2048 and j["n_gpus"] == 2


def test_bench_single_rank_and_mismatch():
    p = _run([])                                           # defaults: N = 1, workload defaults
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["workload"] == "gowalla" and j["steps"] == 400 and j["warmup"] == 20
    # BASELINE configs[1] ("Gowalla 3-layer dim=64 bf16") names the storage type: the default there, fp32 on the other workloads
    assert j["act_dtype"] == "bf16"
    j32 = json.loads([l for l in _run(["--act_dtype", "fp32"]).stdout.splitlines() if l.startswith("{")][0])
    assert j32["act_dtype"] == "fp32"
    # N = 1: "scaling" is the configured mode (config.scaling_mode says the same), one rank observed, and the line carries
    # the end-to-end epoch rate and the 10-epoch quality object next to roofline / cpu_baseline
    assert j["scaling"] == "strong" and j["rccl_ranks_observed"] == 1
    assert {"end_to_end_epoch", "quality", "roofline", "cpu_baseline"} <= set(j["extra_objects"])
    # started by an external launcher with a different world size: refuse, do not guess
    p = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
