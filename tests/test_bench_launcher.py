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


def test_pmc_traffic_is_only_quoted_for_the_code_it_was_measured_on():
    """roofline.traffic comes from profiles/pmc_traffic.json, which records lol_gpu_kernel_key() of the kernel the counters were
    collected on: another key, workload or launch size gets null and a reason, never a stale figure."""
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]
    for kernel, r in rec.items():
        assert r["kernel_key"] and len(r["kernel_key"]) == 16 and min(r["dispatches_per_counter"].values()) >= 100
        got, why = bench.pmc_traffic(kernel, r["workload"], r["pixels_per_launch"], r["kernel_key"])
        assert got == r["traffic_bytes"] and abs(got / r["algorithmic_bytes"] - 1) < 0.02
        assert bench.pmc_traffic(kernel, r["workload"], r["pixels_per_launch"], "0" * 16)[0] is None
        assert "another" in bench.pmc_traffic(kernel, "c2", r["pixels_per_launch"], r["kernel_key"])[1]
        assert bench.pmc_traffic(kernel, r["workload"], 1234, r["kernel_key"])[0] is None
    assert bench.pmc_traffic("no_such_kernel", "c3", 1, "x")[0] is None
    # the two kernels' passes are different files (round 2's interpreter CSVs were copies of the specialised kernel's)
    spec, interp = rec["lol_render_spec"], rec["render_interp"]
    assert spec["counters_avg_per_dispatch"]["SQ_INSTS_SALU"] != interp["counters_avg_per_dispatch"]["SQ_INSTS_SALU"]


def test_pmc_summary_reports_the_common_value_and_lists_stray_dispatches(tmp_path):
    """A deterministic kernel's instruction counters are the same for every dispatch; a dispatch that comes back with another
    kernel's work counted in must not shift the summary (tools/pmc_summary.py), while cycle counters stay averages."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp"]
    for d in range(100):
        rows.append(f"{d},lol_render_spec,SQ_INSTS_SALU,{371313520 if d != 7 else 373803700},{d * 10},{d * 10 + 5}")
        rows.append(f"{d},lol_render_spec,GRBM_GUI_ACTIVE,{20000000 + d},{d * 10},{d * 10 + 5}")
        rows.append(f"{d},other_kernel,SQ_INSTS_SALU,1,{d * 10},{d * 10 + 5}")
    f = tmp_path / "pass.csv"
    f.write_text("\n".join(rows) + "\n")
    c, n, _ = pmc_summary.summarise([str(f)], "lol_render_spec")
    assert c["SQ_INSTS_SALU"] == 371313520.0 and n["_dispatches_off_the_common_value"] == {"SQ_INSTS_SALU": 1}
    assert abs(c["GRBM_GUI_ACTIVE"] - 20000049.5) < 1e-6 and n["SQ_INSTS_SALU"] == 100


def test_root_share_candidates_are_one_launch_splits():
    from loltracer_amd import multi
    for world in (2, 4, 8):
        cands = bench.root_share_candidates(world, 4320)
        assert cands[0][1] == 0 and len(cands) >= 3                      # the equal split first, then lighter roots
        for band, root in cands:
            P = multi.Partition(4320, world, band, root)
            assert sum(P.rank_rows) == 4320 and P.rank_rows[0] <= min(P.rank_rows[1:])
            assert max(P.rank_rows[1:]) - min(P.rank_rows[1:]) <= band
    assert bench.root_share_candidates(1, 2160) == [(2160, 0)] or bench.root_share_candidates(1, 2160)[0][1] == 0
