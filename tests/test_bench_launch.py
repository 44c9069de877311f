"""bench.py's N-rank launch path: `python bench.py --gpus N` must become N ranks or fail loudly.

CPU tests drive the self-spawn with the dry-run transport (gloo, no GPU work); the GPU tests check that a 1-GPU box refuses
`--gpus 2` instead of printing a 1-rank record, and that the forced data-parallel path (one-rank RCCL communicator owned by the
library, all-reduce inside the VM's `nn.sgd`) produces a valid line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, BENCH] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def _gpus():
    try:
        import torch
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def test_dry_run_spawns_two_ranks_that_rendezvous():
    rc, rec, err = _run(["--gpus", "2", "--dry-run"])
    assert rc == 0, err
    assert rec["n_gpus"] == 2 and rec["dry_run"] is True and rec["value"] is None and rec["config"]["parallelism"] == "dp2"


def test_a_rank_that_dies_behind_the_negotiation_fails_the_job_without_a_line():
    """VERDICT r5 #9c: rank 1 of a 3-rank job exits right after the ranks have counted each other.  The survivors' next collective fails within the
    process-group timeout (T4_BENCH_PG_TIMEOUT_S), they exit non-zero and NO rank prints a JSON line - a job that lost a rank never leaves a record."""
    import time
    t0 = time.time()
    rc, rec, err = _run(["--gpus", "3", "--dry-run"], {"T4_BENCH_TEST_DIE_RANK": "1", "T4_BENCH_PG_TIMEOUT_S": "10"}, timeout=300)
    assert rc != 0 and rec is None, (rc, rec, err[-600:])
    assert time.time() - t0 < 200


def test_world_size_mismatch_is_an_error_not_a_one_rank_record():
    rc, rec, err = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert rc == 2 and rec is None and "WORLD_SIZE=1" in err


@pytest.mark.skipif(_gpus() >= 2, reason="box really has 2 GPUs: the run would be a real benchmark")
def test_more_ranks_than_devices_fails_loudly():
    rc, rec, err = _run(["--gpus", "2"])
    assert rc == 2 and rec is None and "refusing to run fewer ranks" in err


def test_launch_helpers():
    from tensorforth_amd import launch
    assert launch.need_spawn(8, {}) and not launch.need_spawn(1, {}) and not launch.need_spawn(8, {"WORLD_SIZE": "8"})
    assert launch.check_world(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}) == (1, 2, 1)
    with pytest.raises(SystemExit):
        launch.check_world(8, {"WORLD_SIZE": "2", "RANK": "0"})


@pytest.mark.gpu
def test_forced_data_parallel_path_on_one_gpu():
    rc, rec, err = _run(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--gemm-iters", "20"], {"T4_BENCH_FORCE_DP": "1"})
    assert rc == 0, err
    assert rec["n_gpus"] == 1 and rec["config"]["allreduce"] == "rccl-native-in-vm" and rec["value"] > 0
    assert rec["roofline"]["frac"] > 0.3 and rec["roofline"]["word_level_us"] > rec["roofline"]["avg_launch_us"] * 0.9
