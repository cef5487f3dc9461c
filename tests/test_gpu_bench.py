"""bench.py end to end on the GPU box, in the driver's command form.

`python bench.py --gpus 2` must start two ranks by itself and print ONE JSON line.  A 1-GPU box has one
device, so the ranks share it and talk gloo (LOL_BENCH_REHEARSE=1) — the partition, the pipelined gather,
the assembly and the timing protocol are the real ones; only the transport differs from the 8-GPU run.
Rank 0 compares the assembled frame with a single-launch render by default (LOL_BENCH_CHECK=0 skips it) and the record
carries every rank's rows, kernel times, wall time and tile order.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOL_GPU_TUNING")}      # (a plain process, like the driver's)
    e.update(env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout      # stdout is the record and nothing else
    return json.loads(lines[0]), p.stderr


def test_gpus_2_self_launches_and_assembles_the_frame():
    out, err = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], LOL_BENCH_REHEARSE="1")       # the check is on by default
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["backend"] == "gloo"
    assert out["config"]["width"] == 7680 and out["config"]["height"] == 4320
    assert "[check] assembled 2-rank frame == single-launch frame: True" in err
    assert out["frame_equal_to_single_launch"] is True
    assert out["gather_ms"] is not None and out["gather_ms"] > 0
    assert out["value"] > 0 and out["scaling"] == "strong"
    # what every rank did, from the one all_gather after the timed loop
    pr = out["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1] and sum(r["rows"] for r in pr) == 4320 and all(r["frames"] == 3 for r in pr)
    assert all(0 < r["kernel_ms_min"] <= r["kernel_ms_avg"] <= r["kernel_ms_max"] for r in pr)
    assert all(r["wall_ms_per_frame"] > 0 and r["tile_order"] == "lpt" and not r["tile_order_deciding"] for r in pr)
    assert out["kernel_ms"]["rank0"] == pr[0]["kernel_ms_avg"] and out["kernel_ms"]["max"] >= out["kernel_ms"]["min"] > 0
    # one kernel stream per slot of the gather pipeline (round 5): a kernel's elapsed time is shared with its neighbour's
    n = out["kernel_ms"]["kernels_sharing_the_device"]
    assert n == out["config"]["kernel_streams"] == 2
    assert abs(out["gather_exposed_ms_estimated"] - (pr[0]["wall_ms_per_frame"] - pr[0]["kernel_ms_avg"] / n)) < 1e-3


def test_orbit_over_two_ranks_checks_frames_against_rank_0():
    out, err = _bench(["--gpus", "2", "--workload", "orbit", "--warmup", "1"], LOL_BENCH_REHEARSE="1")
    assert out["scaling"] == "weak" and out["steps"] == 256 and out["n_ranks_seen"] == 2
    assert out["frames_equal_to_rank0_render"] is True and "[check] 4 orbit frames of 2 ranks" in err
    assert [r["frames"] for r in out["per_rank"]] == [128, 128] and all(r["rows"] == 2160 for r in out["per_rank"])
    # the orbit's frames are independent: two in flight per rank, every one a new camera
    assert out["config"]["frames_in_flight"] == 2 and out["value_new_view"] == out["value"] and out["config"]["camera"].startswith("moving")


def test_default_line_carries_roofline_cpu_baseline_and_both_kernels():
    out, _ = _bench(["--steps", "5", "--warmup", "2"])
    assert out["n_gpus"] == 1 and out["unit"] == "Mpixels/s" and out["dtype"] == "f32"
    assert out["config"]["workload"].startswith("c3:")
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert cb["parity_vs_gpu"]["pixels_differing"] == 0
    ks = out["kernels"]
    assert ks["lol_render_spec"]["mpixels_per_s"] > 0 and ks["render_interp"]["mpixels_per_s"] > 0
    assert ks["render_interp"]["frame_equal_to_spec"] is True
    # the order of the tiles is the LIBRARY's business: its default, longest tiles first, with the sorts it has done so far
    assert out["tile_order"] == "lpt" and out["tile_order_mode"] == "lpt" and out["tile_order_decided_by"].startswith("liblol_gpu")
    assert out["tile_order_sorts_or_decisions"] >= 2
    # what the repeated view buys is on the record: the same frame in the two fixed orders (= any frame with a new camera)
    sch = out["scheduling"]
    assert sch["fixed_rows_frame_equal"] is True and sch["fixed_cols_frame_equal"] is True
    assert sch["new_view_mpixels_per_s"] == max(sch["fixed_rows_mpixels_per_s"], sch["fixed_cols_mpixels_per_s"])
    assert sch["repeated_view_mpixels_per_s"] > sch["new_view_mpixels_per_s"] > 0
    # the one number a reader takes away is labelled: `value` is the repeated view, the rate of a frame with a new camera stands
    # beside it at the top level, and the workload says "still camera"
    assert out["value_new_view"] == sch["new_view_mpixels_per_s"] and "STILL CAMERA" in out["config"]["workload"]
    assert out["config"]["camera"].startswith("still") and out["config"]["frames_in_flight"] == 1
    # the environment's part in the line: no tuning switch of the library took effect in a plain run
    assert out["config"]["env"]["lol_gpu_tuning_switches"] is None
    # frames in flight (never part of `value`): the same workload with 1 / 2 / 3 frames in flight, frames compared
    fl = out["frames_in_flight"]
    for cam in ("still_camera", "moving_camera"):
        assert fl[cam]["2_in_flight_last_frame_equal"] is True and fl[cam]["3_in_flight_last_frame_equal"] is True
        assert fl[cam]["1_in_flight_mpixels_per_s"] > 0
    assert out["value_moving_camera_2_in_flight"] == fl["moving_camera"]["2_in_flight_mpixels_per_s"] > 0
    assert out["host_surface"]["host_surface_mpixels_per_s"]["pipelined_3"] > 0
    # `valu` describes the machine; nothing in it called a fraction exceeds 1
    v = out["valu"]
    assert "frac" not in v and v["reference_equivalent_tops"] > 0 and 0 < v["lane_efficiency_modelled"] <= 1
    assert v["issue_frac"] is None or 0 < v["issue_frac"] <= 1


def test_the_real_backends_chatter_stays_off_stdout():
    """With the nccl (RCCL) process group up — one rank, LOL_BENCH_FORCE_PIPE — the library prints its version banner
    from native code; stdout must still be the one JSON line (the driver parses it), the banner goes to stderr."""
    out, err = _bench(["--steps", "3", "--warmup", "1", "--workload", "c4", "--no-cpu-baseline"], LOL_BENCH_FORCE_PIPE="1")
    assert out["backend"] == "nccl" and out["gather_ms"] is not None


def test_in_process_transport_and_root_emulation():
    """--transport cabi: the frame through lol_gpu_multi_* in ONE process (one device here: RCCL self-exchange + assembly) equals
    the single launch; --emulate-root-of 8: one GPU plays rank 0 of an 8-rank run (its band of every cycle, 1-rank RCCL gather,
    whole-frame assembly) and reports the root's cadence."""
    out, err = _bench(["--transport", "cabi", "--gpus", "1", "--workload", "c4", "--steps", "3", "--warmup", "1"])
    assert out["config"]["transport"] == "cabi" and out["frame_equal_to_single_launch"] is True and out["value"] > 0
    assert out["per_device"][0]["tile_order"] == "lpt" and not out["per_device"][0]["tile_order_deciding"]
    out, err = _bench(["--emulate-root-of", "8", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"], LOL_BENCH_ROOT_SHARE="16,15")
    assert out["metric"].startswith("EMULATION") and out["unit"] == "ms/frame" and out["emulated_world"] == 8
    assert out["partition"]["rows_per_rank"][0] == 512 and out["backend"] == "nccl"
    assert out["assembly"].startswith("lol_gpu_assemble_parts_at") and 0 < out["root_kernel_ms_estimated"] < out["root_kernel_elapsed_ms"] and out["root_kernel_ms_estimated"] < out["value"]
