#!/usr/bin/env python3
"""bench.py — Mpixels/s of the gfx950 sphere-tracer on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|orbit]
                    [--transport dist|cabi] [--emulate-root-of N]

A step = one frame of the hot path (camera ray → march → normal → shadows/Phong → gamma →
XRGB8888) rendered from the scene already resident on the device, into a device framebuffer.

  N = 1   workload "c3": tests/golden/scenes/scene4.lol, 3840x2160, 256 march steps — the config
          BASELINE.json's metric is quoted on.
  N > 1   workload "c4": scene4.lol, 7680x4320, rows band-interleaved over the N ranks (one process
          per GPU), each rank renders its bands, then ONE RCCL gather to rank 0 which un-interleaves
          them into the final framebuffer.  Total work is fixed as N grows → "scaling": "strong".
          The root also receives and un-interleaves the whole frame, so its bands may be less tall than the
          others' (loltracer_amd.multi.Partition; LOL_BENCH_ROOT_SHARE = auto | equal | BAND,ROOT_BAND: `auto`
          times a few frames of each candidate split during set-up and keeps the fastest).
  orbit   256-frame camera orbit of scene4 at 3840x2160, frames striped over ranks, no collective.  A rank's frames are
          independent, so it keeps LOL_BENCH_FRAMES_IN_FLIGHT of them (default 2: measured best) in flight on the library's own streams
          (lol_gpu_set_frames_in_flight): the next frame's first waves fill the tail of the last one's launch.
  --transport cabi       ONE process drives all N devices through lol_gpu_multi_* (the in-process path the
                         reference's C host calls: band partition + RCCL send/recv group + assembly kernel).
  --emulate-root-of N    one GPU plays rank 0 of an N-rank run: renders the root's share of C4, gathers through a
                         1-rank RCCL group, assembles the whole frame — the root's cadence and host cost per frame.

The JSON line carries `roofline` (HBM-write roofline named by BASELINE.json: 4 B per pixel over the
render kernel's average launch duration measured with HIP events on the launch stream) and
`cpu_baseline` (the CPU oracle — a bit-faithful port of naive_renderer.c — on the host cores, on a
bounded row sample of the same frame).  This path is VALU-bound, not HBM-bound; `valu` describes the
MACHINE: the share of the SIMDs' raw issue rate the kernel reaches (2 cycles per wave64 VALU instruction
÷ the measured cycles per instruction), VALU instructions per pixel, and the lane efficiency of the wave
footprint — plus, without a fraction, what the measured rate is worth in the REFERENCE's unfused flops.

N > 1 (either transport): after the timed region the assembled frame is compared with a single-launch
render of the same frame (`frame_equal_to_single_launch`; LOL_BENCH_CHECK=0 skips it), and one all-gather
brings every rank's rows, kernel times, wall time per frame and tile order into `per_rank`, so that a poor
scaling curve can be read from the one run the driver makes.  The orbit compares every rank's first and
last frame with rank 0's render of the same cameras (`frames_equal_to_rank0_render`).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# RCCL (and any sharing of device memory between processes) needs dmabuf IPC on this pool: the host driver refuses the legacy
# mode with `hipIpcGetMemHandle: invalid argument`.  Set HERE, in every process that runs this file — the launcher, and the
# ranks whoever started them (the driver starts them itself with torch.distributed.run) — before torch is imported, i.e.
# before the first HIP call.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

from loltracer_amd import gpu, multi, scene as S

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s spec
BYTES_PER_PIXEL = 4             # one XRGB8888 store per pixel

WORKLOADS = {
    "c2": dict(scene="scene", w=1920, h=1080, max_steps=128),
    "c3": dict(scene="scene4", w=3840, h=2160, max_steps=256),
    "c4": dict(scene="scene4", w=7680, h=4320, max_steps=256),
    "orbit": dict(scene="scene4", w=3840, h=2160, max_steps=256, frames=256),
}


def orbit_camera(i: int, n: int = 256) -> S.Camera:
    """Frame i of the scene4 orbit: doubles on the host, rounded to float."""
    cx, cy, cz = 0.0, 1.0, -6.0
    R = math.sqrt(85.0)
    th = math.atan2(-2.0, 9.0) + 2.0 * math.pi * i / n
    px, py, pz = cx + R * math.sin(th), cy + 5.0, cz + R * math.cos(th)
    dx, dy, dz = cx - px, cy - py, cz - pz
    inv = 1.0 / math.sqrt(dx * dx + dy * dy + dz * dz)
    cam = S.Camera()
    cam.point = S.V3(px, py, pz)
    cam.direction = S.V3(dx * inv, dy * inv, dz * inv)
    cam.fov = float(np.float32(np.float32(150.0) / np.float32(180.0) * np.pi))
    return cam


def flops_per_sdf(prog: S.Program) -> float:
    """Unfused FP32 ops of one sdf() evaluation (sphere 10, smin 13, rbox ~22, plane 1, top 2)."""
    cost = {S.OP_SPHERE: 10, S.OP_RBOX: 22, S.OP_PLANE: 1, S.OP_SMIN: 13, S.OP_SMIN_R: 13, S.OP_TOP: 2}
    return float(sum(cost[prog.ops[i].op] for i in range(prog.n_ops)))


def pmc_traffic(kernel: str, workload: str, pixels_per_launch: int, kernel_key: str):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json:
    WRITE_SIZE + 2 x FETCH_SIZE, one counter group per pass, the gfx950 FETCH correction applied).  PMC
    counters cannot be read from inside this process, so the figure comes from the profile of the same
    command; it is reported only for the kernel / workload / launch size AND CODE it was measured on: the
    profile records lol_gpu_kernel_key() of the kernel it saw, and a kernel whose key differs gets null."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None, "profiles/pmc_traffic.json is missing or has no record of this kernel"
    if not rec or rec.get("workload") != workload or rec.get("pixels_per_launch") != pixels_per_launch:
        return None, "profiles/pmc_traffic.json was measured on another workload / launch size"
    if rec.get("kernel_key") != kernel_key:
        return None, (f"profiles/pmc_traffic.json was measured on kernel code {rec.get('kernel_key')}, this run is "
                      f"{kernel_key}: not quoted (re-run tools/final_profile.sh)")
    return rec["traffic_bytes"], ("profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command on this kernel code, "
                                  "tools/pmc_summary.py; not re-measured in this run)")


def pmc_issue(kernel: str, workload: str, pixels_per_launch: int, kernel_key: str):
    """The issue view of the same committed PMC passes, under the same rule as pmc_traffic (only for the code it was
    measured on): cycles per wave64 VALU instruction per SIMD, VALU instructions per pixel, transcendental share."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["kernels"].get(kernel)
    except (OSError, ValueError, KeyError):
        return None
    if (not rec or rec.get("workload") != workload or rec.get("pixels_per_launch") != pixels_per_launch
            or rec.get("kernel_key") != kernel_key or "cycles_per_valu_instruction_per_simd" not in rec):
        return None
    return {"cycles_per_valu_instruction": rec["cycles_per_valu_instruction_per_simd"],
            "valu_instructions_per_pixel": rec["valu_instructions_per_pixel"],
            "transcendental_share_of_valu": rec.get("transcendental_share_of_valu")}


def lane_efficiency_model(workload: str, dealt: bool = False):
    """Lane efficiency on this workload: needed SDF evaluations / evaluations the waves execute (a wave runs the maximum over
    its 64 lanes), from per-pixel step counts (the PMC cannot see it: finished lanes are predicated, not masked).  `dealt`: the
    waves of a view that repeats — pixels dealt by cost inside 64x16 regions (tests/tools/sorted_region_model.py on the oracle's
    counts, newest profiles/r*_sorted_region_model.json); else the 16x4 rectangles (tools/divergence.py, r*_divergence.json)."""
    import glob
    size = "%dx%d" % (WORKLOADS[workload]["w"], WORKLOADS[workload]["h"])
    if dealt:
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sorted_region_model*.json")), reverse=True):
            try:
                d = json.load(open(path))
            except (OSError, ValueError):
                continue
            if d.get("scene") == WORKLOADS[workload]["scene"] + ".lol" and d.get("size") == size and "64x16" in d:
                return {"lane_efficiency": round(d["64x16"]["by_total"]["lane_efficiency"], 4),
                        "rectangles_16x4": round(d["64x16"]["now"]["lane_efficiency"], 4),
                        "source": "profiles/" + os.path.basename(path) + " (pixels dealt to waves by cost inside 64x16 regions; oracle step counts of sampled strips)"}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_divergence*.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("scene") == WORKLOADS[workload]["scene"] + ".lol" and d.get("size") == size and "16x4" in d:
            return {"lane_efficiency": round(d["16x4"]["lane_efficiency"], 4), "march_only": round(d["16x4"]["march_only"], 4),
                    "shadow_only": round(d["16x4"]["shadow_only"], 4), "source": "profiles/" + os.path.basename(path)}
    return None


def valu_fields(kernel: str, workload: str, px_per_launch: int, kernel_key: str, kernel_mpix: float, ctr, flops_sdf: float, dealt: bool = False) -> dict:
    """`valu`: the machine first — issue_frac = 2.0 cycles (a SIMD issues one wave64 VALU instruction per 2 cycles at best:
    32 lanes per cycle) / the cycles per VALU instruction the PMC passes measured on THIS kernel code; VALU instructions
    per pixel; lane efficiency of the wave footprint — then, without any fraction, what the measured rate is worth in the
    reference's unfused flops (the kernel skips about a third of that work exactly, so a ratio to a peak would say nothing
    about the machine)."""
    out = {"bound": "fp32 VALU issue", "issue_peak_cycles_per_instruction": 2.0}
    issue = pmc_issue(kernel, workload, px_per_launch, kernel_key)
    if issue:
        out.update(issue_frac=round(2.0 / issue["cycles_per_valu_instruction"], 4),
                   cycles_per_valu_instruction=round(issue["cycles_per_valu_instruction"], 4),
                   valu_instructions_per_pixel=round(issue["valu_instructions_per_pixel"], 1),
                   transcendental_share_of_valu=issue["transcendental_share_of_valu"],
                   issue_source="profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command on this kernel code: "
                                "SQ_INSTS_VALU over GRBM_GUI_ACTIVE / 8 XCDs / 1024 SIMDs)")
    else:
        out.update(issue_frac=None, cycles_per_valu_instruction=None, valu_instructions_per_pixel=None,
                   issue_source="profiles/pmc_traffic.json holds no PMC passes of this kernel code / workload: not quoted "
                                "(re-run tools/final_profile.sh)")
    lane = lane_efficiency_model(workload, dealt) or (lane_efficiency_model(workload) if dealt else None)
    # a MODEL from the oracle's per-pixel step counts (the PMC cannot see predicated lanes), not a measurement of the dealt waves
    out["lane_efficiency_modelled"] = lane["lane_efficiency"] if lane else None
    out["lane_efficiency_detail"] = lane
    if ctr is not None and ctr.pixels:
        fpp = (ctr.sdf_evals * flops_sdf + ctr.march_steps * 9 + ctr.shadow_steps * 12) / ctr.pixels
        out["reference_flops_per_pixel"] = round(fpp, 1)
        out["reference_sdf_evals_per_pixel"] = round(ctr.sdf_evals / ctr.pixels, 2)
        out["reference_equivalent_tops"] = round(kernel_mpix * 1e6 * fpp / 1e12, 3)
        out["reference_equivalent_note"] = ("the oracle's unfused FP32 operations per pixel x the measured Mpixels/s: what a renderer "
                                            "that did ALL of naive_renderer.c's arithmetic would have to issue at this rate; the kernel "
                                            "skips part of it exactly (LABNOTES.md §3.6-3.7), so this is not a utilisation")
    return out


# ---- N > 1: what every rank did, on rank 0's line -----------------------------------------------------------------
RANK_STATS = ("rows", "frames", "kernel_ms_avg", "kernel_ms_min", "kernel_ms_max", "wall_ms_per_frame",
              "host_issue_us_per_frame", "tile_order_code", "tile_rows_ms", "tile_cols_ms", "tile_deciding")


def gather_rank_stats(local: dict, device) -> list:
    """One all_gather (a float64 tensor of RANK_STATS per rank; gloo on CPU tensors, RCCL on device tensors) after the
    timed loop: every rank's row in rank order, on every rank."""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"                                # (gloo gathers host tensors; RCCL wants them on the device)
    mine = torch.tensor([float(local[k]) for k in RANK_STATS], dtype=torch.float64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        rows = [mine]
    else:
        rows = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
    return [dict(zip(RANK_STATS, (float(v) for v in r.cpu()))) for r in rows]


def gather_checksums(mine: list, device) -> list:
    """every rank's list of 64-bit frame checksums (same length on every rank), in rank order"""
    if dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor(mine, dtype=torch.int64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [[int(v) for v in t.cpu()]]
    rows = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, t)
    return [[int(v) for v in r.cpu()] for r in rows]


def frame_checksum(frame) -> int:
    """position-weighted sum of the pixels modulo 2^64 (two frames that differ in one pixel differ in it)"""
    v = frame.reshape(-1).to(torch.int64)
    return int((v * (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1000003 + 1)).sum().item())


def per_rank_fields(stats: list, ms_per_step: float, overlap: int = 1) -> dict:
    """The N>1 part of the record from the gathered rows.  exposed_ms_per_frame = a rank's wall time per frame minus its
    own kernel time per frame: what was NOT hidden under its rendering — for rank 0 the part of gather + assembly that
    did not overlap the next frame's kernel, for the others the time they waited for the root.  `overlap` = kernel streams
    per rank (round 5): with two, a frame's kernel shares the device with its neighbour's for all of its run, so its ELAPSED
    time (kernel_ms_*, as measured) is about twice the time the frame costs the rank; the derived figures use elapsed / overlap."""
    per_rank = []
    for i, s in enumerate(stats):
        frames = max(s["frames"], 1.0)
        per_rank.append({"rank": i, "rows": int(s["rows"]), "frames": int(s["frames"]),
                         "kernel_ms_avg": round(s["kernel_ms_avg"], 4), "kernel_ms_min": round(s["kernel_ms_min"], 4),
                         "kernel_ms_max": round(s["kernel_ms_max"], 4), "wall_ms_per_frame": round(s["wall_ms_per_frame"], 4),
                         ("exposed_ms_per_frame" if overlap == 1 else "exposed_ms_per_frame_estimated"):
                             round(s["wall_ms_per_frame"] - s["kernel_ms_avg"] / overlap, 4),
                         "host_issue_us_per_frame": round(s["host_issue_us_per_frame"], 1),
                         "tile_order": ("rows", "cols", "auto", "lpt")[int(s["tile_order_code"]) & 3], "tile_order_deciding": bool(s["tile_deciding"]),
                         "tile_trial_ms": {"rows": round(s["tile_rows_ms"], 4), "cols": round(s["tile_cols_ms"], 4)}})
    k = [r["kernel_ms_avg"] for r in per_rank]
    return {"per_rank": per_rank,
            "kernel_ms": {"min": min(k), "max": max(k), "rank0": k[0], "slowest_rank": k.index(max(k)),
                          "kernels_sharing_the_device": overlap,
                          "note": None if overlap == 1 else "elapsed times of kernels that run side by side with the next frame's kernel "
                                                            "(one stream per slot of the gather pipeline): a frame costs its rank about elapsed / %d" % overlap},
            # measured quantities keep their names; with kernels side by side (overlap > 1) "what a frame costs its rank" is an ESTIMATE
            # (elapsed / overlap: an even split is assumed, not measured) and is named so — round 4's keys were measurements
            ("gather_exposed_ms" if overlap == 1 else "gather_exposed_ms_estimated"):
                per_rank[0]["exposed_ms_per_frame" if overlap == 1 else "exposed_ms_per_frame_estimated"],
            ("ms_per_step_over_slowest_kernel" if overlap == 1 else "ms_per_step_over_slowest_kernel_estimated"):
                round(ms_per_step / max(max(k) / overlap, 1e-9), 4)}


def host_cores() -> int:
    """The host cores this process may really use: its affinity mask, cut to the container's CPU quota when it has one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(sc: S.Scene, cfg: dict, target_s: float = 15.0, gpu_frame=None):
    """Time the CPU oracle on all host cores over a bounded, evenly spread row sample of the frame."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
    cores = host_cores()
    lib = O.lib()
    buf = np.zeros((h, w), dtype=np.uint32)

    def run(stride):
        ctr = O.Counters()
        t0 = time.perf_counter()
        lib.lol_oracle_render_sample(sc.ptr, C.byref(sc.c.camera), w, h, ms, cores, 0, h, stride,
                                     buf.ctypes.data, w * 4, C.byref(ctr))
        return time.perf_counter() - t0, ctr

    probe_stride = max(1, h // 32)
    t_probe, c_probe = run(probe_stride)
    rate = c_probe.pixels / max(t_probe, 1e-9)
    want_px = rate * target_s
    stride = max(1, int(math.ceil(w * h / max(want_px, 1.0))))
    t, ctr = run(stride)
    # one thread too, on a thinner sample of the same frame
    cores_all, cores = cores, 1
    t1, c1 = run(max(stride, h // 24))
    cores = cores_all
    base = dict(value=ctr.pixels / t / 1e6, unit="Mpixels/s", cores=cores, kind="port",
                value_1_thread=round(c1.pixels / t1 / 1e6, 4),
                # (rounds 1 - 5 carried the survey's one-off ratio of this port's time to naive_renderer.c's; the file cannot be
                # built here — SDL2 — so the ratio could never be re-measured by this repository, and the round-5 review found it
                # stale for scene.lol: dropped.  BASELINE.md §2 holds the survey's timings of the reference itself.)
                sample=f"every {stride}th row of the {w}x{h} frame ({ctr.pixels} px, {t:.1f} s, "
                       f"{cores} threads claiming rows from an atomic counter)")
    if gpu_frame is not None:
        # the oracle just rendered these rows of the same frame: use them as the checker for the GPU frame
        rows = np.arange(0, h, stride)
        g, o = gpu_frame[rows], buf[rows]
        sh = np.array([16, 8, 0], dtype=np.uint32)
        d = np.abs(((g[..., None] >> sh) & 0xFF).astype(np.int32) - ((o[..., None] >> sh) & 0xFF).astype(np.int32))
        base["parity_vs_gpu"] = {"pixels_compared": int(g.size), "pixels_differing": int((g != o).sum()),
                                 "max_channel_delta_lsb": int(d.max()),
                                 "note": "XRGB8888 of the timed GPU frame vs the oracle on the sampled rows; float colour "
                                         "parity (<=1e-4) and step-count equality are asserted by tests/test_gpu_parity.py"}
    return base, ctr


def orbit_parity(r, sc, cfg, timed, stream) -> dict:
    """Config 5 at its own size against the oracle, every pixel: the frames the timed pass left in its ring (rendered with
    frames in flight: the timed frames themselves) and one more camera of the orbit rendered now, sequentially.  Also says
    whether the in-flight frames equal a sequential render of the same cameras (what `frames_equal_to_rank0_render` says for
    N > 1)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
    cores = host_cores()
    frames, seq_equal = [], True
    probe = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    extra = cfg["frames"] // 2 + 1                                  # a camera on the far side of the orbit
    for k, gpu_frame in list(timed) + [(extra, None)]:
        cam = orbit_camera(k, cfg["frames"])
        r.render_into(probe.data_ptr(), w, h, ms, stream=stream, frame_camera=sc.frame_camera(w, h, cam))
        torch.cuda.synchronize()
        if gpu_frame is not None:
            seq_equal = seq_equal and bool(torch.equal(gpu_frame, probe))
        g = (gpu_frame if gpu_frame is not None else probe).cpu().numpy().view(np.uint32)
        o, _, _ = O.render(sc, w, h, ms, threads=cores, camera=cam)
        frames.append({"orbit_frame": int(k), "rendered": "timed pass, frames in flight" if gpu_frame is not None else "after it, sequentially",
                       "pixels_compared": int(g.size), "pixels_differing": int((g != o).sum())})
    return {"frames": frames, "timed_in_flight_frames_equal_sequential_renders": seq_equal,
            "checker": "oracle/lol_oracle.c on %d host threads, whole frames" % cores}


def both_kernels(r, sc, spec_frame, w, h, max_steps, fc, stream, spec_ms, px, frames: int = 5):
    """After the timed region: the same frame on the OTHER kernel of the library — the ahead-of-time macro-op
    interpreter `render_interp` (scene as data, fetched with scalar loads; what runs when hipRTC is unavailable) —
    timed with HIP events, and compared with the frame the default kernel just rendered."""
    out = {r.kernel_name(): {"mpixels_per_s": round(px / (spec_ms * 1e-3) / 1e6, 1), "kernel_ms_avg": round(spec_ms, 4)}}
    ri = gpu.Renderer(torch.cuda.current_device(), specialize=4 if r.kernel_name() == "lol_render_spec" else 1)
    try:
        ri.prepare(sc)
        if ri.kernel_name() == r.kernel_name():
            # LOL_GPU_SPECIALIZE=0 (or a failed hipRTC) forces both renderers onto the same kernel: nothing to compare
            out["note"] = f"the second renderer runs {ri.kernel_name()} too (LOL_GPU_SPECIALIZE / hipRTC): no second kernel timed"
            return out
        buf = torch.zeros_like(spec_frame)
        for _ in range(8):                                # warm-up: the view repeats, so the library's tables settle (three frames)
            ri.render_into(buf.data_ptr(), w, h, max_steps, stream=stream, frame_camera=fc)
        ev = []
        for _ in range(frames):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ri.render_into(buf.data_ptr(), w, h, max_steps, stream=stream, frame_camera=fc)
            e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        out[ri.kernel_name()] = {"mpixels_per_s": round(px / (ms * 1e-3) / 1e6, 1), "kernel_ms_avg": round(ms, 4),
                                 "frames": frames, "frame_equal_to_spec": bool(torch.equal(buf, spec_frame))}
    finally:
        ri.close()
    return out


def scheduling_rates(r, spec_frame, w, h, max_steps, fc, stream, scheduled_ms, px, frames: int = 10, warm: int = 16) -> dict:
    """What the timed frames owe to the repeated view (LABNOTES.md §3.9).  `value` is quoted on the workload BASELINE.json names —
    one camera, frame after frame — and while the camera stands still the library schedules a frame by what the frame before
    cost (every pixel still computed from scratch).  A frame with a NEW camera has no such tables and runs in a fixed tile
    order: the same frame is timed here in both fixed orders on the same context, and compared."""
    out = {"repeated_view_mpixels_per_s": round(px / (scheduled_ms * 1e-3) / 1e6, 1)}
    buf = torch.zeros_like(spec_frame)
    best = None
    for order in ("rows", "cols", "rows"):                # (rows twice: the first series also absorbs what is left of the ramp)
        r.set_tile_order(order)
        ev = []
        for i in range(warm + frames):                    # (the device has idled while the other context was torn down: let it ramp)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r.render_into(buf.data_ptr(), w, h, max_steps, stream=stream, frame_camera=fc)
            e1.record()
            if i >= warm:
                ev.append((e0, e1))
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        out[f"fixed_{order}_mpixels_per_s"] = round(px / (ms * 1e-3) / 1e6, 1)
        out[f"fixed_{order}_frame_equal"] = bool(torch.equal(buf, spec_frame))
        best = ms if best is None else min(best, ms)
    r.set_tile_order("lpt")
    out["new_view_mpixels_per_s"] = round(px / (best * 1e-3) / 1e6, 1)
    out["note"] = ("repeated view: pixels dealt to waves by the previous frame's per-pixel evaluation counts, waves handed out longest "
                   "first (schedule only; no pixel value is reused); new view: the better fixed tile order, which is what every frame "
                   "of a moving camera gets (the orbit workload measures that case)")
    return out


def startup_times(sc: S.Scene, w: int, h: int, max_steps: int, device: int) -> dict:
    """Tiered start-up (lol_gpu_upload_program; LABNOTES.md §3.8), on contexts of their own BEFORE anything else has compiled
    this scene in this process: how long render_prepare's part takes with the scene compiler really running (disk cache
    switched off for it), when the first frame is there (it renders on the interpreter), when the scene's own kernel takes
    over — and the same for a second context of the same scene (code object in the process's cache)."""
    buf = torch.zeros((h, w), dtype=torch.int32, device=f"cuda:{device}")
    torch.cuda.synchronize()
    out = {}
    saved = os.environ.get("LOL_GPU_CACHE_DIR")
    os.environ["LOL_GPU_CACHE_DIR"] = ""
    try:
        for tag in ("cold", "cached"):
            t0 = time.perf_counter()
            r0 = gpu.Renderer(device)
            t1 = time.perf_counter()
            r0.prepare(sc, wait=False)                    # = render_prepare: flatten + upload; returns with the compiler at work
            t2 = time.perf_counter()
            r0.render_into(buf.data_ptr(), w, h, max_steps)
            r0.sync()
            t3 = time.perf_counter()
            first_kernel = r0.kernel_name()
            while r0.specialize_state()[0] == 1:              # the scene compiler's (first) run
                time.sleep(0.0005)
            r0.render_into(buf.data_ptr(), w, h, max_steps)   # the frame boundary at which its kernel takes over
            r0.sync()
            t4 = time.perf_counter()
            compiler_ms = r0.specialize_state()[1]
            two_tiers = r0.specialize_state()[0] in (5, 6)    # scenes of 257 ... 1024 ops: the inlined form follows (LABNOTES.md §3.2)
            r0.specialize_wait()
            t5 = time.perf_counter()
            out[f"create_ms_{tag}"] = round((t1 - t0) * 1e3, 2)
            out[f"prepare_ms_{tag}"] = round((t2 - t1) * 1e3, 2)
            out[f"first_frame_ms_{tag}"] = round((t3 - t1) * 1e3, 2)          # from the start of prepare to the first frame finished
            out[f"first_frame_kernel_{tag}"] = first_kernel
            out[f"scene_kernel_ready_ms_{tag}"] = round((t4 - t1) * 1e3, 2)
            out[f"scene_compiler_ms_{tag}"] = round(compiler_ms, 2)
            if two_tiers:
                out[f"second_kernel_ready_ms_{tag}"] = round((t5 - t1) * 1e3, 2)
                out[f"second_compiler_ms_{tag}"] = round(r0.specialize_state()[1], 2)
            r0.close()
    finally:
        if saved is None:
            os.environ.pop("LOL_GPU_CACHE_DIR", None)
        else:
            os.environ["LOL_GPU_CACHE_DIR"] = saved
    out["note"] = ("prepare = lol_scene_flatten + lol_gpu_upload_program (tables, interpreter lists, on-device proofs of the fast paths; "
                   "hipRTC runs on a host thread); cold = first context of the process, disk cache off; cached = a second context, "
                   "code object in the process's cache")
    return out


def env_record() -> dict:
    """What of the environment took part in this line: the library's A/B switches that were honoured (lol_gpu_tuning_switches:
    only beside LOL_GPU_TUNING=1) and this file's own LOL_BENCH_* variables."""
    return {"lol_gpu_tuning_switches": gpu.tuning_switches() or None,
            "lol_bench": {k: v for k, v in sorted(os.environ.items()) if k.startswith("LOL_BENCH_")} or None}


def frames_in_flight_rates(r, sc, cfg, frames: int = 48) -> dict:
    """Frames in flight (lol_gpu_set_frames_in_flight; LABNOTES.md §3.11).  `value` keeps the reference's frame loop: frame i+1 is
    launched when frame i is done with the stream (main.c:189-194).  A host whose frames are independent may keep several
    in flight; this leg renders the SAME workload with 1, 2 and 3 frames in flight on the library's own streams, for a camera
    that stands still (scheduled per stream) and for one that moves every frame (a short arc of the orbit; fixed tile order),
    wall clock over `frames` frames after a ramp, frames compared with the one-stream frames.  Never part of `value`."""
    w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
    moving = [sc.frame_camera(w, h, orbit_camera(i, 256)) for i in range(0, 2 * frames, 2)] if cfg["scene"] == "scene4" else None
    still = [sc.frame_camera(w, h)]
    out = {"frames": frames}
    ring = [torch.zeros((h, w), dtype=torch.int32, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()                          # (the fills run on torch's stream, the frames on the library's own)
    ref = {}
    for label, cams in (("still_camera", still), ("moving_camera", moving)):
        if cams is None:
            continue
        res = {}
        for n in (1, 2, 3):
            r.set_frames_in_flight(n)

            def run(k):
                for i in range(k):
                    r.render_into(ring[i % n].data_ptr(), w, h, ms, frame_camera=cams[i % len(cams)])
                r.sync()
            run(40 if label == "still_camera" else 16)
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                run(frames)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[f"{n}_in_flight_mpixels_per_s"] = round(frames * w * h / best / 1e6, 1)
            last = ring[(frames - 1) % n]
            if n == 1:
                ref[label] = last.clone()
            else:
                res[f"{n}_in_flight_last_frame_equal"] = bool(torch.equal(last, ref[label]))
        out[label] = res
    r.set_frames_in_flight(1)
    out["note"] = ("the library's own streams in turn (stream == NULL), a ring of n destinations; still camera: every stream keeps its own "
                   "scheduling tables; moving camera: 2 orbit frames per step (2.8 degrees), fixed tile order")
    return out


def launcher_command(n: int, argv: list, port: int) -> list:
    """The command `python bench.py --gpus N` runs for N > 1: one rank per GPU under torch.distributed.run
    (the same form the driver uses), rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


# ---------------------------------------------------------------------------------------------------------------------------
# A run on N > 1 devices must not be able to hang, or to fail without a word (round-5 review: the first 8-GPU run this code ever
# gets is the driver's, and a stuck collective would have sat out RCCL's default of ten minutes — the driver's whole limit —
# and left nothing).  Three layers, each of which ends the run with ONE JSON line {"error", "stage", "rank"} on stdout and a
# non-zero status:
#   - every process group is created with a timeout (init_group: COLLECTIVE_TIMEOUT_S, 90 s), so a collective a rank never
#     joins raises in the ranks that did;
#   - every rank runs a host-side watchdog (Deadline): main() names the stage it is in, and a stage that outlives its limit —
#     a kernel that never ends, a gather that never completes, an in-process lol_gpu_multi_sync that never returns (the
#     watchdog is a thread of its own and ends the process with os._exit: no Python frame of the stuck thread has to run) —
#     gets the line and exit status 3; torch.distributed.run then ends the other ranks;
#   - the launcher of `python bench.py --gpus N` has a deadline of its own (launch_ranks: LAUNCHER_DEADLINE_S) after which it
#     kills the ranks' whole process group.
# Nothing here re-executes anything: children are started, or the process exits.
COLLECTIVE_TIMEOUT_S = float(os.environ.get("LOL_BENCH_COLLECTIVE_TIMEOUT_S", "90"))
LAUNCHER_DEADLINE_S = float(os.environ.get("LOL_BENCH_LAUNCHER_DEADLINE_S", "540"))
STAGE_LIMITS_S = {"process group": 120, "render_prepare": 180, "set-up": 150, "warm-up": 90, "timed loop": 120,
                  "checks": 120, "rank stats": 60, "record": 420}


def error_line(error: str, stage: str, rank) -> str:
    return json.dumps({"error": error, "stage": stage, "rank": rank})


class Deadline:
    """The host-side watchdog of one rank.  stage(name) starts the clock of a named stage (limit: STAGE_LIMITS_S, scaled by
    LOL_BENCH_STAGE_DEADLINE_SCALE, or the `seconds` given); a stage still running at its limit ends the process: one JSON line
    on `fd` (the saved stdout), the same on stderr, os._exit(3).  done() stops the watchdog."""

    def __init__(self, rank: int, fd: int, poll_s: float = 0.25):
        import threading
        self.rank, self.fd, self.poll_s = rank, fd, poll_s
        self.scale = float(os.environ.get("LOL_BENCH_STAGE_DEADLINE_SCALE", "1"))
        self._lock = threading.Lock()
        self._stage, self._until = None, None
        self._thread = threading.Thread(target=self._watch, name="bench-deadline", daemon=True)
        self._thread.start()

    def stage(self, name: str, seconds: float | None = None):
        limit = (seconds if seconds is not None else STAGE_LIMITS_S.get(name, 120)) * self.scale
        with self._lock:
            self._stage, self._until = name, time.monotonic() + limit
            self._limit = limit

    def done(self):
        with self._lock:
            self._stage, self._until = None, None

    def _watch(self):
        while True:
            time.sleep(self.poll_s)
            with self._lock:
                stage, until, limit = self._stage, self._until, getattr(self, "_limit", 0)
            if stage is not None and time.monotonic() > until:
                line = error_line(f"stage not finished after its limit of {limit:.0f} s: the process ends itself", stage, self.rank)
                try:
                    os.write(self.fd, (line + "\n").encode())
                    os.write(2, ("[bench] " + line + "\n").encode())
                finally:
                    os._exit(3)


def init_group(backend: str, rank: int, world: int, device=None):
    """dist.init_process_group with the bounded timeout every group of this file gets (a collective that a rank never joins
    raises after COLLECTIVE_TIMEOUT_S in the ranks that did, instead of waiting out the backend's default of ten minutes or more)."""
    from datetime import timedelta
    kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=timedelta(seconds=COLLECTIVE_TIMEOUT_S), **kw)


def launch_ranks(n: int, argv: list, script: str | None = None, deadline_s: float | None = None) -> int:
    """Start the N ranks as a child process group and wait for them — at most deadline_s (LAUNCHER_DEADLINE_S): after that
    the whole group is killed, ONE JSON error line goes to stdout and the status is 124.  Returns the ranks' exit status.
    Called before this process has made any HIP call (importing torch does not initialise the GPU).  `script`: what the ranks
    run (this file; the tests of the launcher pass small rank programs of their own)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL needs it on this pool
    cmd = launcher_command(n, argv, port)
    if script is not None:
        cmd[cmd.index(os.path.abspath(__file__))] = os.path.abspath(script)
    limit = LAUNCHER_DEADLINE_S if deadline_s is None else deadline_s
    # a session of its own: the deadline can end torch.distributed.run AND every rank it started with one killpg of exactly that group
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)   # stdout / stderr inherited: rank 0's JSON line
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=10)
                break
            except subprocess.TimeoutExpired:
                continue
        print(error_line(f"the {n} ranks had not finished after {limit:.0f} s: their process group was killed", "launcher deadline", None), flush=True)
        return 124
    except KeyboardInterrupt:
        try:
            os.killpg(proc.pid, signal.SIGTERM)
        except ProcessLookupError:
            pass
        return proc.wait()


def host_surface_rates(r, sc, cfg, cams, frames: int = 40):
    """Through the boundary the reference's host uses (naive_renderer.c:233-235: surf->pixels is HOST memory): whole
    frames into a pitched host surface, synchronously per frame (what render_thread does) in each host mode, and
    with two frames in flight (lol_gpu_render_host_begin / _end).  Wall-clock, never part of `value`."""
    w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
    pitch = (w + 16) * 4
    surf = np.zeros(h * pitch, dtype=np.uint8)
    cam_list = [sc.c.camera] if cams is None else cams
    out = {"pitch_bytes": pitch, "frames": frames}

    def rate(dt):
        return round(frames * w * h / dt / 1e6, 1)

    for i in range(3):
        r.render_host(surf.ctypes.data, w, h, ms, camera=cam_list[i % len(cam_list)], pitch_bytes=pitch)
    t0 = time.perf_counter()
    for i in range(frames):
        r.render_host(surf.ctypes.data, w, h, ms, camera=cam_list[i % len(cam_list)], pitch_bytes=pitch)
    out["sync_mpixels_per_s"] = rate(time.perf_counter() - t0)
    r.render_host_begin(w, h, ms, camera=cam_list[0])
    r.render_host_begin(w, h, ms, camera=cam_list[1 % len(cam_list)])
    r.render_host_end(surf.ctypes.data, pitch, w, h)
    t0 = time.perf_counter()
    for i in range(frames):
        r.render_host_begin(w, h, ms, camera=cam_list[(i + 2) % len(cam_list)])
        r.render_host_end(surf.ctypes.data, pitch, w, h)
    out["pipelined_mpixels_per_s"] = rate(time.perf_counter() - t0)
    r.render_host_end(surf.ctypes.data, pitch, w, h)
    # ... and with three (lol_gpu_set_frames_in_flight(3): the surface lags two frames, three kernels may overlap)
    try:
        r.set_frames_in_flight(3)
        for i in range(3):
            r.render_host_begin(w, h, ms, camera=cam_list[i % len(cam_list)])
        r.render_host_end(surf.ctypes.data, pitch, w, h)
        t0 = time.perf_counter()
        for i in range(frames):
            r.render_host_begin(w, h, ms, camera=cam_list[(i + 3) % len(cam_list)])
            r.render_host_end(surf.ctypes.data, pitch, w, h)
        out["pipelined_3_mpixels_per_s"] = rate(time.perf_counter() - t0)
        r.render_host_end(surf.ctypes.data, pitch, w, h)
        r.render_host_end(surf.ctypes.data, pitch, w, h)
    finally:
        r.set_frames_in_flight(1)
    # synchronous per frame (kernel, then the copy), with two frames in flight (the copy under the next frame's kernel, whose first
    # waves fill the tail of this one's launch: consecutive kernels on different streams) and with three
    out["host_surface_mpixels_per_s"] = {"sync": out["sync_mpixels_per_s"], "pipelined": out["pipelined_mpixels_per_s"],
                                         "pipelined_3": out.get("pipelined_3_mpixels_per_s")}
    return out


def root_share_candidates(world: int, h: int):
    """(band_rows, root_band_rows) splits to try, by LOL_BENCH_ROOT_SHARE: `equal` = the root like everyone else;
    `B,R` = bands of B rows, the root's R rows; `auto` (default) = the equal split and a few with a lighter root, timed
    during set-up (main).  One launch per rank in every case: only the heights of the bands differ."""
    mode = os.environ.get("LOL_BENCH_ROOT_SHARE", "auto")
    equal = multi.choose_band_rows(h, world) or 4
    if world == 1 or mode == "equal":
        return [(equal, 0)]
    if mode != "auto":
        try:
            b, r = (int(v) for v in mode.split(","))
        except ValueError:
            raise SystemExit(f"LOL_BENCH_ROOT_SHARE={mode}: want auto, equal or BAND,ROOT_BAND")
        return [(b, r)]
    # the root's share relative to an equal one: 1, then ~15/16 … 3/4; for every ratio the band height (16 / 12 / 8 rows)
    # whose last, partial cycle leaves the busiest of the OTHER ranks the fewest rows
    cands = [(equal, 0)]
    for ratio in (15 / 16, 7 / 8, 13 / 16, 3 / 4):
        best = None
        for band in (16, 12, 8):
            root = max(1, int(ratio * band + 0.5))
            if root >= band:
                continue
            P = multi.Partition(h, world, band, root)
            key = (max(P.rank_rows[1:]), -band)
            if best is None or key < best[0]:
                best = (key, (band, root))
        if best and best[1] not in cands:
            cands.append(best[1])
    return cands


def run_cabi(args, record_fd):
    """--transport cabi: ONE process, N devices behind lol_gpu_multi_* (lol_multi.hip) — what the reference's C host
    reaches through hip_renderer --devices: every device renders its bands, one grouped ncclSend/ncclRecv brings the
    parts to devices[0], the library's assembly kernel un-interleaves them.  Same workload, timing and record as the
    one-process-per-GPU transport."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: there is no CPU rendering path")
    n = args.gpus
    if torch.cuda.device_count() < n:
        raise SystemExit(f"--transport cabi --gpus {n}: only {torch.cuda.device_count()} device(s) visible")
    name = args.workload or ("c3" if n == 1 else "c4")
    if name == "orbit":
        raise SystemExit("--transport cabi renders whole frames over all devices; the orbit stripes frames over ranks")
    cfg = WORKLOADS[name]
    w, h, max_steps = cfg["w"], cfg["h"], cfg["max_steps"]
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", cfg["scene"] + ".lol"))
    # the watchdog (Deadline): lol_gpu_multi_sync waits for the devices without a bound of its own — a kernel or an RCCL group that
    # never completes would hold this process for ever; the watchdog's thread ends it with the JSON error line instead
    watchdog = Deadline(0, record_fd)
    watchdog.stage("render_prepare")
    m = gpu.MultiRenderer(list(range(n)))
    m.prepare(sc)
    watchdog.stage("set-up")
    torch.cuda.set_device(0)
    frames = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    fc = sc.frame_camera(w, h)

    def run(k):
        for i in range(k):
            m.render_into(frames[i % 2].data_ptr(), w, h, max_steps, frame_camera=fc)
        m.sync()

    per_dev = int(os.environ.get("LOL_BENCH_PARTS_PER_DEVICE", "1"))
    m.set_parts_per_device(per_dev)
    want_order = os.environ.get("LOL_BENCH_TILE_ORDER", "lpt")       # the library's default unless pinned (A/B runs)
    m.set_tile_order(want_order)
    trials = []
    cands = root_share_candidates(n, h)
    best = cands[0]
    if len(cands) > 1:
        for band, root_band in cands:
            m.set_band_rows(band); m.set_root_band_rows(root_band)
            run(4)
            t0 = time.perf_counter(); run(12); dt = time.perf_counter() - t0
            trials.append({"band_rows": band, "root_band_rows": root_band or band, "ms_per_frame": round(dt / 12 * 1e3, 4)})
        best = min(zip(cands, trials), key=lambda ct: ct[1]["ms_per_frame"])[0]
    if n > 1:
        m.set_band_rows(best[0]); m.set_root_band_rows(best[1])
    watchdog.stage("warm-up")
    run(max({"c2": 400, "c3": 100, "c4": 32}[name], gpu.TILE_TRIAL_FRAMES + 10))
    run(args.warmup)
    watchdog.stage("timed loop")
    t0 = time.perf_counter()
    run(args.steps)
    dt = time.perf_counter() - t0
    watchdog.stage("checks")
    # outside the timed region, by default (LOL_BENCH_CHECK=0 skips it): the frame assembled from every device's bands must
    # equal ONE launch of the whole frame on device 0
    check = None
    if os.environ.get("LOL_BENCH_CHECK", "1") != "0":
        r1 = gpu.Renderer(0)
        r1.prepare(sc)
        ref = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
        r1.render_into(ref.data_ptr(), w, h, max_steps, frame_camera=fc)
        r1.sync()
        check = bool(torch.equal(ref, frames[(args.steps - 1) % 2]))
        print(f"[check] assembled {n}-device frame == single-launch frame: {check}", file=sys.stderr, flush=True)
        r1.close()
    tiles = [m.tile_order(i) for i in range(n)]
    out = {
        "metric": f"Mpixels/s at {w}x{h}, <={max_steps} march steps; max |pixel delta| vs naive_renderer.c",
        "value": round(args.steps * w * h / dt / 1e6, 2), "unit": "Mpixels/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{name}: tests/golden/scenes/{cfg['scene']}.lol {w}x{h}, {max_steps} march steps, rows in bands "
                               f"over {n} device(s) of ONE process (lol_gpu_multi_*: RCCL send/recv group to device 0 + assembly kernel)",
                   "width": w, "height": h, "max_steps": max_steps, "transport": "cabi",
                   "parts_per_device": per_dev, "band_rows": best[0] if n > 1 else h, "root_band_rows": (best[1] or best[0]) if n > 1 else h,
                   "kernel": m.kernel_name()},
        "root_share_trials": trials or None, "frame_equal_to_single_launch": check,
        # lol_gpu_set_tile_order(AUTO) decides per device, on that device's own launches
        "per_device": [{"device": i, "tile_order": t["order"], "tile_order_deciding": t["deciding"], "tile_trial_ms": t["trial_ms"]}
                       for i, t in enumerate(tiles)],
    }
    os.write(record_fd, (json.dumps(out) + "\n").encode())
    m.close()
    watchdog.done()
    if check is False:
        raise SystemExit("bench.py: the assembled frame differs from the single-launch render (record printed above)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=list(WORKLOADS))
    ap.add_argument("--band-rows", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--transport", default="dist", choices=["dist", "cabi"])
    ap.add_argument("--emulate-root-of", type=int, default=0, metavar="N")
    args = ap.parse_args()

    if args.transport == "cabi":
        sys.stdout.flush()
        record_fd = os.dup(1)
        os.dup2(2, 1)                                 # RCCL's banner must not reach stdout (see below)
        return run_cabi(args, record_fd)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this process becomes the launcher.  It starts the N ranks as CHILD
        # processes (never an exec) before anything here has touched the GPU, relays their output and
        # exits with their status.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries ONE line, the JSON record: native libraries write there too (RCCL prints a version banner
    # on communicator creation), so file descriptor 1 points at stderr for the rest of the run and the record
    # goes to the saved descriptor at the end.
    sys.stdout.flush()
    record_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the watchdog (see Deadline above): for N > 1 every stage up to the record has a limit; a single rank keeps the limits of
    # the stages that can hang on a device (there is no collective to wait for, and its legs after the timed region take minutes)
    watchdog = Deadline(rank, record_fd)
    watchdog.stage("process group")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with "
                         f"`python bench.py --gpus {args.gpus}` or torch.distributed.run --nproc-per-node {args.gpus}")

    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs a HIP device: there is no CPU rendering path (rank {rank} of {world})")
    # Rehearsal switch for a 1-GPU box: LOL_BENCH_REHEARSE=1 puts every rank on device 0 and uses gloo, so
    # the N>1 code path (partition, pipelined gather, assembly, timing) can be exercised without N GPUs.
    rehearse = os.environ.get("LOL_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        init_group("gloo" if rehearse else "nccl", rank, world, dev)

    emulate = args.emulate_root_of
    if emulate and (world != 1 or emulate < 2):
        raise SystemExit("--emulate-root-of N wants N >= 2 and ONE process (it plays rank 0 of an N-rank run on one GPU)")
    name = args.workload or ("c3" if world == 1 and not emulate else "c4")
    cfg = WORKLOADS[name]
    w, h, max_steps = cfg["w"], cfg["h"], cfg["max_steps"]
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", cfg["scene"] + ".lol"))
    startup = None
    if world == 1 and not emulate and os.environ.get("LOL_BENCH_STARTUP", "1") != "0":
        try:
            startup = startup_times(sc, w, h, max_steps, local_rank)  # before this process has compiled the scene
        except Exception as e:                                    # noqa: BLE001 — an extra of the record must not cost the line
            startup = {"error": f"{type(e).__name__}: {e}"}
    watchdog.stage("render_prepare", 300 if startup is not None else None)      # (the start-up leg above compiled the scene cold, twice)
    r = gpu.Renderer(local_rank)
    r.prepare(sc)                                     # render_prepare: flatten + upload once; waits for the scene's own kernel
    watchdog.stage("set-up")
    # A side stream: its handle is non-NULL (NULL means "the context's own stream" in lol_gpu.h), and
    # torch.cuda.Event / the nccl gather below are ordered on whatever stream is current.
    side = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream, "expected a non-NULL HIP stream handle"

    if world > 1:
        # set-up, like render_prepare: bring the communicator up before any timed or warm-up step
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)
        dist.barrier()
        torch.cuda.synchronize()

    orbit = name == "orbit"
    if orbit and emulate:
        raise SystemExit("--emulate-root-of renders whole frames")
    # N>1: two frames in flight per rank — frame i's gather overlaps frame i+1's kernel (multi.GatherPipeline).
    depth = max(1, int(os.environ.get("LOL_BENCH_PIPELINE_DEPTH", "2")))     # 1 = gather each frame before the next renders
    # LOL_BENCH_FORCE_PIPE=1 under a 1-rank torchrun: exercise the real backend's gather path on one device
    force_pipe = (os.environ.get("LOL_BENCH_FORCE_PIPE") == "1" or bool(emulate)) and not orbit
    if force_pipe and world == 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        init_group("nccl", 0, 1, dev)
    piped = (world > 1 or force_pipe) and not orbit

    def assembler(staging, frame, P, stream_handle):
        """the root's un-interleave: the library's uint4 kernel on the assembly stream (lol_gpu_assemble_parts_at)"""
        gpu.assemble_parts_at(r, staging.data_ptr(), [gpu.Rows(*g) for g in P.geometry], P.part_row0, w, h,
                              frame.data_ptr(), w * 4, stream_handle)

    if os.environ.get("LOL_BENCH_ASSEMBLE") == "torch":     # A/B: the round-2 torch index copy instead
        assembler = None

    kernel_ms = []
    state = {"P": None, "pipe": None}

    # N>1: the kernels of consecutive frames on different streams (one per slot of the gather pipeline): a rank's launch of its
    # bands is a small launch whose ramp and tail the next frame's kernel fills (LABNOTES.md §3.11, §4).  LOL_BENCH_KERNEL_STREAMS=1:
    # every kernel on the one side stream, as before round 5.
    n_kstreams = depth if os.environ.get("LOL_BENCH_KERNEL_STREAMS", "slots") != "1" else 1
    kstreams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if (piped and n_kstreams > 1) else None

    def make_pipeline(band_rows, root_band_rows):
        P = multi.Partition(h, emulate or world, args.band_rows or band_rows, root_band_rows)
        state["P"] = P
        state["pipe"] = multi.GatherPipeline(w, h, P.band, dev, depth=depth, force_collective=force_pipe, partition=P,
                                             assembler=assembler, kernel_streams=kstreams)

    if orbit:
        frames_total = cfg["frames"]
        my_frames = list(range(rank, frames_total, world))
        cams = [sc.frame_camera(w, h, orbit_camera(i, frames_total)) for i in my_frames]
    else:
        cams = [sc.frame_camera(w, h)]
    # The orbit's frames are independent (BASELINE.json config 5: "frames striped", no per-frame collective): a rank keeps
    # several in flight on the library's own streams (lol_gpu_set_frames_in_flight), each into a destination of its own.  Every
    # other workload keeps the reference's loop — one frame after the other on one stream (main.c:189-194).
    fif = max(1, min(4, int(os.environ.get("LOL_BENCH_FRAMES_IN_FLIGHT", "2")))) if orbit else 1
    if fif > 1:
        r.set_frames_in_flight(fif)
    ring = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(fif)] if not piped else None
    local = ring[0] if ring else None
    ext_streams = {}

    def launch(dst_tensor, fc, timed):
        P = state["P"]
        # the frame's stream: the side stream — or, with frames in flight, whichever of the library's own streams is next
        # (events are recorded there: torch.cuda.Event.record() alone sees only torch's current stream)
        if fif > 1:
            handle = r.next_stream()
            ev_stream = ext_streams.get(handle) or ext_streams.setdefault(handle, torch.cuda.ExternalStream(handle, device=dev))
            s_arg = None
        else:
            # (the stream that is current: the side stream — or the slot's own kernel stream inside the gather pipeline)
            ev_stream = torch.cuda.current_stream(dev)
            s_arg = ev_stream.cuda_stream
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ev_stream)
        if P is None:
            r.render_into(dst_tensor.data_ptr(), w, h, max_steps, stream=s_arg, frame_camera=fc)
        elif P.rank_rows[rank]:                      # ONE launch: this rank's band of every cycle, compactly
            r.render_into(dst_tensor.data_ptr(), w, h, max_steps, rows=gpu.Rows(*P.geometry[rank]), stream=s_arg,
                          frame_camera=fc)
        if timed:
            e1.record(ev_stream)
            kernel_ms.append((e0, e1))

    def step(i, timed):
        fc = cams[i % len(cams)]
        if not piped:
            launch(ring[i % fif], fc, timed)
        else:
            state["pipe"].submit(lambda part: launch(part, fc, timed))

    def fence():
        if state["pipe"] is not None:
            state["pipe"].drain()
        torch.cuda.synchronize()                      # (the whole device: the library's own streams too)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Set-up: the split.  The root's extra work (receiving 7/8 of the frame, un-interleaving all of it) depends on the
    # machine, so `auto` measures instead of guessing: a few frames of each candidate, slowest rank's time, keep the best.
    trials = []
    if piped:
        cands = root_share_candidates(emulate or world, h)
        if emulate and len(cands) > 1:
            cands = cands[:1]                         # one GPU cannot rank the splits: the equal one unless told otherwise
        best = cands[0]
        if len(cands) > 1:
            for band_c, root_c in cands:
                make_pipeline(band_c, root_c)
                for i in range(4):
                    step(i, False)
                fence()
                t0 = time.perf_counter()
                for i in range(16):
                    step(i, False)
                fence()
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                trials.append({"band_rows": band_c, "root_band_rows": root_c or band_c, "ms_per_frame": round(float(t.item()) / 16 * 1e3, 4)})
                state["pipe"] = None
            # every rank holds the same all-reduced times, so every rank picks the same split
            best = min(zip(cands, trials), key=lambda ct: ct[1]["ms_per_frame"])[0]
        make_pipeline(*best)
    pipe, P = state["pipe"], state["P"]

    steps = args.steps
    if orbit:
        steps = len(cams)                             # one pass over this rank's stripe of the orbit
    # Set-up, not measurement: keep the device busy for about a quarter of a second so that the clocks have ramped before
    # the W warm-up steps — the timed region of the default run is only ~40 ms, and a cold start moved it by several %.
    # A fixed frame count per workload (every rank issues the same number of gathers), none for the 256-frame orbit.
    # The order in which a launch hands out its tiles is the LIBRARY's business (lol_gpu_set_tile_order; its default: longest
    # tiles first, from the costs the tiles of the frames before reported — sorted on the device, on the frame's own stream, a
    # few small kernels every sixteenth frame (LPT_RESORT) that are part of what is timed), so the rate reported here is the rate a host gets
    # through the boundary with no tuning of its own.  LOL_BENCH_TILE_ORDER = rows | cols | auto pins another mode (A/B runs).
    want_order = os.environ.get("LOL_BENCH_TILE_ORDER", "lpt")      # the library's default: longest tiles first
    if want_order not in ("rows", "cols", "auto", "lpt"):
        raise SystemExit(f"LOL_BENCH_TILE_ORDER={want_order}: want lpt, auto, rows or cols")
    r.set_tile_order(want_order)
    prewarm = {"c2": 400, "c3": 100, "c4": 32 * max(world, emulate), "orbit": gpu.TILE_TRIAL_FRAMES + 10}[name]
    prewarm = max(prewarm, gpu.TILE_TRIAL_FRAMES + 10)
    watchdog.stage("warm-up")
    for i in range(prewarm):
        step(i, False)
    fence()
    for i in range(args.warmup):
        step(i, False)
    fence()
    watchdog.stage("timed loop")
    tile = r.tile_order()                             # everything issued so far has finished: the trials are in
    host_s = 0.0
    t0 = time.perf_counter()
    for i in range(steps):
        th = time.perf_counter()
        step(i, True)
        host_s += time.perf_counter() - th            # host time spent issuing the frame (launches, gather, assembly)
    fence()
    dt_local = dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if force_pipe and world == 1 and rank == 0:
        print("[force-pipe] 1-rank nccl gather path completed", file=sys.stderr, flush=True)
    # ---- everything below is outside the timed region ----
    watchdog.stage("checks")
    if fif > 1:
        r.set_frames_in_flight(1)
    # Correctness of what was just timed, by default (LOL_BENCH_CHECK=0 skips it; the driver's plain `bench.py --gpus N` gets it):
    # partitioned frames: the frame assembled on rank 0 from every rank's bands == ONE launch of the whole frame on rank 0;
    # the orbit: every rank's first and last frame == rank 0's own render of the same cameras (compared by checksum).
    check_on = os.environ.get("LOL_BENCH_CHECK", "1") != "0"
    frame_equal = frames_equal = None
    # the orbit's timed frames were rendered with `fif` frames in flight into a ring: the last `fif` of this rank's stripe are still
    # there (frame i in ring[i % fif]) — the frames that were TIMED, not renders made again afterwards (round-5 advisor)
    orbit_timed = None
    if orbit and ring is not None and steps >= fif:
        orbit_timed = [(my_frames[i], ring[i % fif].clone()) for i in range(steps - fif, steps)]
    if check_on and pipe is not None and not pipe.single and not emulate:
        final = pipe.drain()
        if rank == 0:
            ref = torch.zeros((h, w), dtype=torch.int32, device=dev)
            r.render_into(ref.data_ptr(), w, h, max_steps, stream=stream, frame_camera=cams[0])
            torch.cuda.synchronize()
            frame_equal = bool(torch.equal(final, ref))
            print(f"[check] assembled {world}-rank frame == single-launch frame: {frame_equal}", file=sys.stderr, flush=True)
    if check_on and orbit and world > 1:
        probe = torch.zeros((h, w), dtype=torch.int32, device=dev)
        mine = []
        for k in (0, len(cams) - 1):
            r.render_into(probe.data_ptr(), w, h, max_steps, stream=stream, frame_camera=cams[k])
            torch.cuda.synchronize()
            mine.append(frame_checksum(probe))
        sums = gather_checksums(mine, dev)
        if rank == 0:
            frames_equal, compared = True, 0
            for rk in range(world):
                theirs = list(range(rk, cfg["frames"], world))
                for k, got in zip((theirs[0], theirs[-1]), sums[rk]):
                    r.render_into(probe.data_ptr(), w, h, max_steps, stream=stream,
                                  frame_camera=sc.frame_camera(w, h, orbit_camera(k, cfg["frames"])))
                    torch.cuda.synchronize()
                    frames_equal = frames_equal and frame_checksum(probe) == got
                    compared += 1
            print(f"[check] {compared} orbit frames of {world} ranks == rank 0's render of the same cameras: {frames_equal}",
                  file=sys.stderr, flush=True)

    # the exchange step on its own (outside the timed region): barrier, gather this rank's last part to rank 0,
    # un-interleave there, wait — best of 3.  In the timed loop it overlaps the next frame's kernel.
    gather_ms = None
    if pipe is not None and not pipe.single:
        ts = []
        for _ in range(3):
            fence()
            t0 = time.perf_counter()
            pipe.regather()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        gather_ms = round(min(ts) * 1e3, 4)

    k_ms = [a.elapsed_time(b) for a, b in kernel_ms]
    k_avg = sum(k_ms) / max(len(k_ms), 1)
    px_per_launch = (P.rank_rows[rank] if P is not None else h) * w      # pixels this rank renders per frame
    # what every rank did: one all_gather after the timed loop (a rank's wall time is its own, before the max over ranks)
    watchdog.stage("rank stats")
    rank_stats = gather_rank_stats({
        "rows": P.rank_rows[rank] if P is not None else h, "frames": steps,
        "kernel_ms_avg": k_avg, "kernel_ms_min": min(k_ms) if k_ms else 0.0, "kernel_ms_max": max(k_ms) if k_ms else 0.0,
        "wall_ms_per_frame": dt_local / max(steps, 1) * 1e3, "host_issue_us_per_frame": host_s / max(steps, 1) * 1e6,
        "tile_order_code": {"rows": 0, "cols": 1, "auto": 2, "lpt": 3}[tile["order"]], "tile_rows_ms": tile["trial_ms"]["rows"],
        "tile_cols_ms": tile["trial_ms"]["cols"], "tile_deciding": tile["deciding"]}, dev)
    if orbit:
        total_px = cfg["frames"] * w * h              # all ranks together render each frame once
        steps_reported = cfg["frames"]
    else:
        total_px = steps * w * h
        steps_reported = steps
    value = total_px / dt / 1e6
    watchdog.stage("record", None if world > 1 else 1500)      # (one rank: the oracle leg and the other extras of the record run here)

    if rank == 0:
        overlap = len(kstreams) if kstreams else (fif if orbit else 1)      # kernels that share the device: the elapsed time of one is that many frames' worth
        achieved = px_per_launch * BYTES_PER_PIXEL / (k_avg * 1e-3) / 1e9     # (the contract's figure: bytes over the launch's own duration)
        traffic, traffic_source = pmc_traffic(r.kernel_name(), name, px_per_launch, r.kernel_key())
        band = P.band if P is not None else h
        out = {
            "metric": f"Mpixels/s at {w}x{h}, <={max_steps} march steps; max |pixel delta| vs naive_renderer.c",
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": steps_reported,
            "warmup": args.warmup, "ms_per_step": round(dt / steps_reported * 1e3, 4), "higher_is_better": True,
            "scaling": "weak" if orbit else "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{name}: tests/golden/scenes/{cfg['scene']}.lol {w}x{h}, {max_steps} march steps"
                                   + (f", {cfg['frames']}-frame orbit striped over ranks, a new camera every frame, {fif} frame(s) in flight per rank" if orbit else
                                      (f", rows in bands of {band} ({P.root_band or band} for rank 0) over {P.world} ranks + RCCL gather to rank 0"
                                       if P is not None else ", one kernel launch per frame")
                                      + ", STILL CAMERA: the same view frame after frame (every pixel computed again in every frame; the "
                                        "library schedules a frame by what the frame before it cost — `value_new_view` is the rate of a frame "
                                        "whose camera has just moved; frames stay on the device — `value_through_render_thread_as_main_c_calls_it` "
                                        "is what a host gets through the reference's unmodified protocol, one frame at a time into a host surface)"),
                       "width": w, "height": h, "max_steps": max_steps, "band_rows": band, "frames_in_flight": fif,
                       "kernel_streams": len(kstreams) if kstreams else 1,
                       "camera": "moving (orbit)" if orbit else "still (repeated view)",
                       "kernel": r.kernel_name(), "kernel_key": r.kernel_key(), "transport": "dist",
                       "env": env_record()},
            # what the process group really is: ranks seen by torch.distributed and its backend ("nccl" = RCCL)
            "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "backend": dist.get_backend() if dist.is_initialized() else None,
            "gather_ms": gather_ms,
            "partition": P.describe() if P is not None else None,
            "root_share_trials": trials or None,
            # lol_gpu_set_tile_order: the library's mode and what it is doing (lpt = longest tiles first: sorts so far; auto: its trials)
            "tile_order": tile["order"], "tile_order_mode": tile["mode"], "tile_order_sorts_or_decisions": tile["decisions"],
            "tile_order_trials_ms": tile["trial_ms"] if tile["mode"] == "auto" else None,
            "tile_order_decided_by": "liblol_gpu (its default)" if "LOL_BENCH_TILE_ORDER" not in os.environ else "LOL_BENCH_TILE_ORDER",
            "assembly": None if pipe is None else ("lol_gpu_assemble_parts_at (library kernel, own stream)" if assembler else "torch index_select"),
            "host_issue_us_per_frame": round(host_s / max(steps, 1) * 1e6, 1),
            "prewarm_frames": prewarm,            # untimed set-up frames before the W warm-up steps (clock ramp)
            "parity_checker": "oracle/lol_oracle.c — the CPU restatement of naive_renderer.c, pinned to values composed from the reference's "
                              "own compiled primitives; its loop structure is restated (naive_renderer.c needs SDL2 to build): LABNOTES.md §5",
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6),
                         "traffic": traffic, "traffic_source": traffic_source,
                         "algorithmic_bytes": px_per_launch * BYTES_PER_PIXEL,
                         "kernel_ms_avg": round(k_avg, 4), "kernels_sharing_the_device": overlap,
                         # with frames in flight a launch's duration counts the time it shares the device with its neighbour(s):
                         "achieved_per_frame_of_device_time": round(achieved * overlap, 3),
                         "pixels_per_launch": px_per_launch,
                         "bytes_per_pixel": BYTES_PER_PIXEL,
                         "note": "north_star names the HBM-write roofline; the path is FP32-VALU-bound, see `valu`.  `traffic` above the "
                                 "algorithmic bytes is what the scheduling tables of a repeated view move (LABNOTES.md §3.9: a 4-byte pixel-table "
                                 "entry read and a 4-byte store of its own per lane) — memory traffic, of which this path uses about 1 % of the "
                                 "peak, traded for lanes that finish together"},
        }
        if world > 1 or (pipe is not None and not pipe.single):
            out.update(per_rank_fields(rank_stats, dt / max(steps, 1) * 1e3, overlap))
            if orbit:
                out["frames_equal_to_rank0_render"] = frames_equal
            else:
                out["frame_equal_to_single_launch"] = frame_equal
        if emulate:
            # not a benchmark line: ONE GPU playing rank 0 of an N-rank run.  `ms_per_step` is the root's cadence —
            # its share of the kernels + the gather through RCCL (1 rank: the self-copy) + the whole-frame assembly.
            out["metric"] = f"EMULATION: cadence of rank 0 of a {emulate}-rank {name} run, on one GPU (ms per frame)"
            out["value"] = round(dt / steps * 1e3, 4)
            out["unit"] = "ms/frame"
            out["higher_is_better"] = False
            out["emulated_world"] = emulate
            out["root_kernel_elapsed_ms"] = round(k_avg, 4)         # measured (HIP events)
            out["root_kernel_ms" if overlap == 1 else "root_kernel_ms_estimated"] = round(k_avg / overlap, 4)       # per frame: elapsed over the kernels that share the device
            out["implied_mpixels_per_s_if_root_is_the_critical_path"] = round(w * h / (dt / steps) / 1e6, 1)
        # The legs below come after the timed region and only add to the record: one that fails says so in its place instead
        # of taking the line — the metric the driver reads — down with it.
        def leg(key, fn):
            try:
                fn()
            except Exception as e:                              # noqa: BLE001 — whatever went wrong, the record says it
                out[key] = {"error": f"{type(e).__name__}: {e}"}
                print(f"[bench] the `{key}` leg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)

        if world == 1 and not orbit and local is not None:       # (local is None in the 1-rank gather rehearsal)
            leg("kernels", lambda: out.__setitem__("kernels", both_kernels(r, sc, local, w, h, max_steps, cams[0], stream, k_avg, px_per_launch)))
        # `value` is the rate of a view that REPEATS (the workload BASELINE.json names: one camera); the rate of a frame whose camera
        # has just moved stands beside it at the top level: the same frame in the better fixed tile order, same context, same run
        out["value_new_view"] = None
        if orbit:
            out["value_new_view"] = out["value"]          # every frame of the orbit has a new camera
        if world == 1 and not orbit and local is not None and tile["mode"] == "lpt" and os.environ.get("LOL_BENCH_SCHEDULING", "1") != "0":
            def sched_leg():
                out["scheduling"] = scheduling_rates(r, local, w, h, max_steps, cams[0], stream, k_avg, px_per_launch)
                out["value_new_view"] = out["scheduling"]["new_view_mpixels_per_s"]
                out["value_new_view_note"] = ("Mpixels/s of the same frame when its camera has just moved: no costs of a frame before, the better "
                                              "of the two fixed tile orders (`scheduling`); `value` is the repeated view")
            leg("scheduling", sched_leg)
        if world == 1 and not orbit and local is not None and os.environ.get("LOL_BENCH_FRAMES_IN_FLIGHT_LEG", "1") != "0":
            def fif_leg():
                out["frames_in_flight"] = frames_in_flight_rates(r, sc, cfg)
                mv = out["frames_in_flight"].get("moving_camera")
                if mv:
                    # the round-5 figure beside `value` (still camera, sequential) and `value_new_view` (one new frame, sequential):
                    # a camera that moves EVERY frame, two frames in flight on the library's own streams
                    out["value_moving_camera_2_in_flight"] = mv["2_in_flight_mpixels_per_s"]
            leg("frames_in_flight", fif_leg)
        if world == 1 and not args.no_cpu_baseline and orbit and orbit_timed:
            def orbit_parity_leg():
                out["orbit_parity"] = orbit_parity(r, sc, cfg, orbit_timed, stream)
            leg("orbit_parity", orbit_parity_leg)
        if world == 1 and not args.no_cpu_baseline and not orbit and local is not None:
            def cpu_leg():
                base, ctr = cpu_baseline(sc, cfg, gpu_frame=local.cpu().numpy().view(np.uint32))
                out["cpu_baseline"] = base
                out["valu"] = valu_fields(r.kernel_name(), name, px_per_launch, r.kernel_key(), px_per_launch / (k_avg * 1e-3) / 1e6,
                                          ctr, flops_per_sdf(r.program), dealt=tile["order"] == "lpt")
            leg("cpu_baseline", cpu_leg)
        if startup is not None:
            out["startup"] = startup
        if world == 1 and local is not None and os.environ.get("LOL_BENCH_HOST_SURFACE", "1") != "0":
            def host_leg():
                torch.cuda.synchronize()
                hs_cams = [orbit_camera(i, 256) for i in range(0, 256, 16)] if orbit else None
                out["host_surface"] = host_surface_rates(r, sc, cfg, hs_cams)
                # beside `value` (frames that stay on the device), at the top level: what a host gets through the reference's UNMODIFIED
                # protocol — render_thread as main.c:189-194 calls it: one frame at a time, into a host surface, kernel and PCIe copy in
                # series — and with --pipeline (frame i - 1 reaches the surface while frame i renders)
                out["value_through_render_thread_as_main_c_calls_it"] = out["host_surface"]["sync_mpixels_per_s"]
                out["value_through_render_thread_pipelined"] = out["host_surface"]["pipelined_mpixels_per_s"]
            leg("host_surface", host_leg)
        sys.stdout.flush()
        os.write(record_fd, (json.dumps(out) + "\n").encode())

    r.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    watchdog.done()
    if rank == 0 and (frame_equal is False or frames_equal is False):
        raise SystemExit("bench.py: the frames of the partitioned run differ from the single-launch render (record printed above)")


if __name__ == "__main__":
    main()
