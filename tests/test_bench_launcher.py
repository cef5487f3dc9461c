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
        # 4 B per pixel + what the scheduling tables of a repeated view move (LABNOTES.md §3.9): a 4-byte pixel-table entry read per
        # lane (33 MB for C3) and every lane storing its own 4 bytes (sectors written 1.7 times on average): 2.8 x the algorithmic
        # bytes, about 1 % of the HBM peak at this frame rate — the trade the round-3 review asked for
        assert got == r["traffic_bytes"] and 1.0 <= got / r["algorithmic_bytes"] < 3.2
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


def test_rank_processes_set_the_rccl_ipc_mode_themselves():
    """RCCL needs dmabuf IPC on this pool (HSA_ENABLE_IPC_MODE_LEGACY=0).  The driver starts the ranks itself with
    torch.distributed.run, so the launcher's environment is not enough: importing bench.py — what every rank does before
    its first HIP call — must set it, without overriding a value the caller chose."""
    code = "import os, sys; sys.path.insert(0, %r); os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); import bench; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0 and p.stdout.strip() == "0", p.stderr[-1500:]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.index('os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")') < src.index("import torch")
    code = "import os, sys; sys.path.insert(0, %r); os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '1'; import bench; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])" % ROOT
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.stdout.strip() == "1"


def test_multi_rank_record_schema():
    """What an N>1 line must carry so that the one run the driver makes explains itself: every rank's rows, kernel times, wall
    time, exposed (non-overlapped) time and tile order; min / max / rank-0 kernel time; the root's exposed gather time."""
    stats = []
    for rank in range(8):
        row = dict.fromkeys(bench.RANK_STATS, 0.0)
        row.update(rows=512 if rank == 0 else 544, frames=20, kernel_ms_avg=0.55 + 0.01 * rank, kernel_ms_min=0.54, kernel_ms_max=0.7,
                   wall_ms_per_frame=0.68 if rank == 0 else 0.66, host_issue_us_per_frame=85.0, tile_order_code=3 if rank % 2 else 0,
                   tile_rows_ms=0.6, tile_cols_ms=0.58, tile_deciding=0)
        stats.append(row)
    f = bench.per_rank_fields(stats, 0.68)
    assert set(f) >= {"per_rank", "kernel_ms", "gather_exposed_ms", "ms_per_step_over_slowest_kernel"}
    assert len(f["per_rank"]) == 8 and [r["rank"] for r in f["per_rank"]] == list(range(8))
    for r in f["per_rank"]:
        assert set(r) >= {"rank", "rows", "frames", "kernel_ms_avg", "kernel_ms_min", "kernel_ms_max", "wall_ms_per_frame",
                          "exposed_ms_per_frame", "host_issue_us_per_frame", "tile_order", "tile_trial_ms"}
    assert f["per_rank"][0]["rows"] == 512 and f["per_rank"][3]["tile_order"] == "lpt" and f["per_rank"][2]["tile_order"] == "rows"
    assert f["kernel_ms"] == {"min": 0.55, "max": 0.62, "rank0": 0.55, "slowest_rank": 7, "kernels_sharing_the_device": 1, "note": None}
    # two kernel streams per rank: a kernel's elapsed time is shared with its neighbour's, the derived figures say so
    g = bench.per_rank_fields([dict(row, kernel_ms_avg=2 * row["kernel_ms_avg"]) for row in stats], 0.68, 2)
    assert g["kernel_ms"]["kernels_sharing_the_device"] == 2 and "elapsed / 2" in g["kernel_ms"]["note"]
    # ... and under names of their own: an even split of the shared time is assumed, not measured (round-5 advisor)
    assert "gather_exposed_ms" not in g and "exposed_ms_per_frame" not in g["per_rank"][0]
    assert abs(g["gather_exposed_ms_estimated"] - f["gather_exposed_ms"]) < 1e-9
    assert abs(g["ms_per_step_over_slowest_kernel_estimated"] - f["ms_per_step_over_slowest_kernel"]) < 1e-3
    assert abs(f["gather_exposed_ms"] - 0.13) < 1e-9 and abs(f["per_rank"][0]["exposed_ms_per_frame"] - 0.13) < 1e-9
    # the names the record uses for the checks (bench.py main / run_cabi)
    src = open(os.path.join(ROOT, "bench.py")).read()
    for name in ('"frame_equal_to_single_launch"', '"frames_equal_to_rank0_render"', 'os.environ.get("LOL_BENCH_CHECK", "1") != "0"'):
        assert name in src


def _stats_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = dict.fromkeys(bench.RANK_STATS, 0.0)
        mine.update(rows=100 + rank, frames=5, kernel_ms_avg=1.0 + rank, kernel_ms_min=0.9 + rank, kernel_ms_max=1.2 + rank,
                    wall_ms_per_frame=1.5 + rank, tile_order_code=1 if rank == 1 else 3)
        stats = bench.gather_rank_stats(mine, torch.device("cpu"))
        sums = bench.gather_checksums([rank * 7 + 1, -(2 ** 62) - rank], torch.device("cpu"))
        q.put((rank, stats, sums))
    finally:
        dist.destroy_process_group()


def test_rank_stats_and_checksums_reach_every_rank_over_gloo():
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stats_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, stats, sums in got:
        assert [s["rows"] for s in stats] == [100.0, 101.0, 102.0] and [s["kernel_ms_avg"] for s in stats] == [1.0, 2.0, 3.0]
        assert [int(s["tile_order_code"]) for s in stats] == [3, 1, 3]
        assert sums == [[1, -(2 ** 62)], [8, -(2 ** 62) - 1], [15, -(2 ** 62) - 2]]
    fields = bench.per_rank_fields(got[0][1], 3.6)
    assert fields["kernel_ms"]["slowest_rank"] == 2 and fields["per_rank"][1]["tile_order"] == "cols"


def test_frame_checksum_sees_single_pixels_and_their_position():
    a = torch.zeros((5, 7), dtype=torch.int32)
    b = a.clone(); b[2, 3] = 1
    c = a.clone(); c[2, 4] = 1
    assert len({bench.frame_checksum(a), bench.frame_checksum(b), bench.frame_checksum(c)}) == 3
    neg = torch.full((3, 3), -1, dtype=torch.int32)              # XRGB with alpha set reads as a negative int32: still well defined
    assert bench.frame_checksum(neg) == bench.frame_checksum(neg.clone())


def test_valu_record_describes_the_machine_and_has_no_fraction_above_one():
    import json
    rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"]["lol_render_spec"]

    class Ctr:
        pixels, sdf_evals, march_steps, shadow_steps = 1000, 97000, 21900, 71500
    v = bench.valu_fields("lol_render_spec", rec["workload"], rec["pixels_per_launch"], rec["kernel_key"], 7800.0, Ctr, 107.0)
    assert abs(v["issue_frac"] - 2.0 / rec["cycles_per_valu_instruction_per_simd"]) < 1e-3 and 0 < v["issue_frac"] <= 1
    assert abs(v["valu_instructions_per_pixel"] - rec["valu_instructions_per_pixel"]) < 0.1
    assert 0 < v["lane_efficiency_modelled"] <= 1
    assert "frac" not in v and v["reference_equivalent_tops"] > 0
    for k, x in v.items():
        if k.endswith("frac") and x is not None:
            assert x <= 1.0
    # another kernel code: the issue figures are withheld, never quoted stale
    w = bench.valu_fields("lol_render_spec", rec["workload"], rec["pixels_per_launch"], "0" * 16, 7800.0, Ctr, 107.0)
    assert w["issue_frac"] is None and w["valu_instructions_per_pixel"] is None and "not quoted" in w["issue_source"]


# ------------------------------------------------------------------ a run on N > 1 ranks cannot hang, or fail without a word

RANK_PROGRAM = os.path.join(ROOT, "tests", "rank_programs", "two_ranks.py")


def _launch(mode, deadline_s, extra_env=None):
    """bench.launch_ranks in a process of its own (it must not have touched a GPU, and its JSON error line goes to stdout)."""
    import time
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "raise SystemExit(bench.launch_ranks(2, ['--mode', %r], script=%r, deadline_s=%r))" % (ROOT, mode, RANK_PROGRAM, deadline_s))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    for attempt in range(3):
        t0 = time.monotonic()
        p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        dt = time.monotonic() - t0
        # (the launcher takes a free port by asking the kernel for one and closing it again; now and then the rendezvous then finds it
        # taken — by the sockets of the test before this one — and the run ends before any rank program has started: once more)
        if p.returncode != 0 and "address already in use" in p.stderr.lower() and attempt < 2:
            continue
        break
    return p, dt


def _json_lines(text):
    import json
    out = []
    for ln in text.splitlines():
        try:
            d = json.loads(ln)
        except ValueError:
            continue
        if isinstance(d, dict):
            out.append(d)
    return out


def test_two_ranks_that_behave_finish_with_one_line():
    p, dt = _launch("ok", 120)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _json_lines(p.stdout) == [{"value": 4.0}]


def test_a_rank_that_dies_mid_run_ends_the_run_non_zero_within_the_deadline():
    """Rank 1 exits (status 17) while rank 0 waits for it in a collective: torch.distributed.run ends rank 0, the launcher hands
    the failure on — long before its own deadline or the group's timeout."""
    p, dt = _launch("die", 120)
    assert p.returncode not in (0, 124) and dt < 90, (p.returncode, dt, p.stderr[-2000:])
    assert not any("value" in d for d in _json_lines(p.stdout))          # no record from a failed run


def test_a_rank_that_never_joins_a_collective_trips_the_watchdog():
    """Rank 1 sleeps instead of joining; rank 0's stage has a 3 s limit: its watchdog writes the JSON error line (stage, rank) to
    stdout and ends the process with status 3, which ends the run — no ten-minute wait for the backend's own timeout."""
    p, dt = _launch("hang", 120)
    assert p.returncode not in (0, 124) and dt < 90, (p.returncode, dt, p.stderr[-2000:])
    errs = [d for d in _json_lines(p.stdout) if "error" in d]
    assert len(errs) == 1 and errs[0]["stage"] == "timed loop" and errs[0]["rank"] == 0 and "3 s" in errs[0]["error"]


def test_ranks_that_do_nothing_for_ever_are_ended_by_the_launchers_own_deadline():
    """No rank has a watchdog and none ever exits: after its deadline the launcher kills the ranks' whole process group (its own
    session: torch.distributed.run AND the ranks), prints the JSON error line and returns 124."""
    p, dt = _launch("sleep", 8)
    assert p.returncode == 124 and dt < 60, (p.returncode, dt, p.stderr[-2000:])
    errs = [d for d in _json_lines(p.stdout) if "error" in d]
    assert len(errs) == 1 and errs[0]["stage"] == "launcher deadline" and errs[0]["rank"] is None
    # nothing of the group is left behind
    left = subprocess.run(["ps", "-eo", "pid,args"], stdout=subprocess.PIPE, text=True).stdout
    assert "two_ranks.py --mode sleep" not in left


def test_every_process_group_of_bench_has_a_bounded_timeout():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert src.count("dist.init_process_group(") == 1 and "timeout=timedelta(seconds=COLLECTIVE_TIMEOUT_S)" in src
    assert bench.COLLECTIVE_TIMEOUT_S <= 120 and bench.LAUNCHER_DEADLINE_S <= 570      # the driver's limit for a run is ten minutes
