"""`python bench.py --gpus N` (the driver's command form) must start the N ranks by itself.

No GPU here: the ranks get as far as "needs a HIP device" — which is enough to see that the
launcher built the right command, that N processes came up with WORLD_SIZE=N and that their
failure status reaches the caller.  The GPU-side rehearsal is tests/test_gpu_bench.py.
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_command_is_the_drivers_form():
    cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "7", "--warmup", "2"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29999"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_gpus_2_starts_two_ranks_and_relays_their_status():
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side plumbing test")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode != 0                                    # the ranks failed (no device) and the launcher says so
    # the ranks ran main() with WORLD_SIZE=2 (torchrun may stop the second one before it gets to say so)
    assert "bench.py needs a HIP device" in p.stderr and "of 2)" in p.stderr
    assert "launch with torch.distributed.run" not in p.stderr  # the old refusal is gone
    assert p.stdout.strip() == ""                               # no JSON line from a failed run


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
