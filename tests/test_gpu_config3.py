"""Config 3 of the benchmark (a fixed total of 4K diagrams split by image index over the ranks) as the
driver starts it: plain ``python bench.py --gpus N``, which launches its own ranks as child processes.
Rehearsed on the one-GPU box with two ranks on one device and gloo for the barrier / reductions; the data
path has no collective, so the device code that runs is the same as on N GPUs.

Replaces the serial loop over uploaded files of the reference (receiveFile.py:22)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra, timeout=420):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)  # the launch under test is the one without a launcher
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, lines


@pytest.mark.gpu
@pytest.mark.timeout(480)
def test_bench_two_ranks_started_by_bench_itself():
    p, lines = _run_bench(["--gpus", "2", "--same-device", "--dist-backend", "gloo", "--total", "64",
                           "--steps", "1", "--warmup", "1", "--no-cpu"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE json line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["images_total"] == 64 and out["config"]["images_per_gpu"] == 32
    assert out["parity_ok"] is True
    assert "all equal" in out["shard_gather"]
    assert out["roofline"]["traffic"] is not None  # per-image traffic scaled to the shard, not keyed by batch
    assert out["value"] > 0 and out["unit"] == "Mpix/s"


def test_bench_launch_command_is_a_child_process_not_an_exec():
    # host logic only: the launcher helper builds a torch.distributed.run command for this very file
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "subprocess.run(cmd" in src and "torch.distributed.run" in src
    assert "os.exec" not in src and "execv" not in src
