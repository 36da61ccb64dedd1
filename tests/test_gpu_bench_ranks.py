"""`python bench.py --gpus N` with N > 1 and no launcher around it: bench.py starts its own N ranks as fresh child
processes (before anything in the parent touches the GPU), rank 0 prints the one JSON line, the return code is relayed.
Rehearsed here on ONE GPU with --backend gloo (the ranks share the device, the all-gather goes through host memory);
RCCL needs one device per rank and is the driver's multi-GPU run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_two_ranks_started_by_bench_itself_prediction_tables_sharded_by_draw():
    one = run_bench("--gpus", "1", "--workload", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    two = run_bench("--gpus", "2", "--workload", "cfg5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--backend", "gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["draws_per_gpu"] == 500 and two["config"]["failed_draws"] == 0 and two["config"]["all_finite"]
    assert two["config"]["gathered_bytes"] == one["config"]["gathered_bytes"] == 2 * 2230 * 1000 * 8
    assert two["metric"] == one["metric"] and two["scaling"] == "strong"


@pytest.mark.timeout(900)
def test_three_ranks_grid_sharded_by_row_ragged():
    r = run_bench("--gpus", "3", "--workload", "cfg3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--backend", "gloo")
    assert r["n_gpus"] == 3 and r["config"]["evals_total"] == 60 * 1728 and r["config"]["evals_per_gpu"] == 20 * 1728
    assert r["config"]["failed_evals"] == 0 and r["roofline"]["bound"] == "valu-issue"
