"""bench.py with N > 1 on the 1-GPU box: two ranks sharing GPU 0, gloo as the control plane (CSPM_BENCH_BACKEND=gloo -- the driver's
8-GPU runs use one rank per GPU and RCCL).  Covers the launcher bench.py becomes without torchrun, the barrier / MAX-over-ranks
timing and the whole-job `value`, for the sharded default (every rank times its own pairs) and for C4 (rank 0 holds the batch,
batch.run_batch dispatches it)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, backend="gloo", **extra_env):
    env = dict(os.environ, CSPM_BENCH_BACKEND=backend, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], check=True, env=env, timeout=900, capture_output=True, text=True)
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout  # rank 0 prints ONE line and nothing else reaches stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("config,pixels", [("C1", 450 * 375), ("C4", 1242 * 375)])
def test_two_ranks_one_line(config, pixels):
    steps = 2
    one = _bench("--gpus", "1", "--config", config, "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline")
    two = _bench("--gpus", "2", "--config", config, "--steps", str(steps), "--warmup", "1", "--no-cpu-baseline")
    for line, n in ((one, 1), (two, 2)):
        assert line["n_gpus"] == n and line["steps"] == steps and line["warmup"] == 1
        assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None
        assert config in line["config"]["workload"]
        # whole-job throughput: every rank's pairs over the slowest rank's time
        want = pixels * steps * n / (line["ms_per_step"] * steps / 1e3) / 1e6
        assert abs(line["value"] - want) <= 1e-6 * want
    assert "cpu_baseline" not in two or two["cpu_baseline"] is None  # the CPU leg runs at N = 1 only


def test_rccl_process_group_and_a_clean_stdout():
    """CSPM_BENCH_FORCE_DIST=1: bench.py creates an RCCL process group (backend nccl) even at one GPU, the barrier / MAX reduction run
    as RCCL collectives and C4 goes through run_batch's broadcast / scatter / gather -- the code path of the driver's 8-GPU runs, on
    the one GPU a test box has.  RCCL writes a version banner to file descriptor 1: the result line must still be the ONLY thing
    on stdout."""
    line = _bench("--gpus", "1", "--config", "C4", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", backend="nccl", CSPM_BENCH_FORCE_DIST="1")
    assert line["n_gpus"] == 1 and "batch.run_batch" in line["config"]["dispatch"]
    want = 1242 * 375 * 3 / (line["ms_per_step"] * 3 / 1e3) / 1e6
    assert abs(line["value"] - want) <= 1e-6 * want
