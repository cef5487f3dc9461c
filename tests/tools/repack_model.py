#!/usr/bin/env python3
"""What would re-packing shadow rays inside a wave (or a block of waves) buy?  CPU model on the oracle's per-pixel, per-light
step counts (test infrastructure: runs the oracle, not the product).  For every 16x4 patch of sampled 4-row bands of C3:
  now      the wave marches light after light, each for the maximum over its lanes            (sum_l max_lanes steps[l])
  queue    the (pixel, light) marches of the patch are tasks; a lane that finishes takes the next one (list scheduling,
           refill only when at least R lanes are idle or nothing else is left)
  block-N  the same with N waves sharing one queue (N patches side by side)
  global-K (round 4, VERDICT round 3 next #2: trade idle HBM for lane efficiency) the render kernel marches every shadow ray for at
           most K steps; a ray that is not finished by then is a SURVIVOR: its state (origin, direction, res, t, L, steps, pixel,
           light: 40 B) is appended to a queue in global memory (one wave-level atomicAdd per wave), the next launch marches the
           queue — compacted: 64 survivors per wave, in arrival order — for at most K more steps, and so on (128 / K rounds); a
           last pass finishes the pixels that had survivors (Phong sum in light order + gamma + store; no SDF evaluations: the
           pixel's p, n, id and the per-light factors, 64 B, are parked in memory meanwhile).  Charged: the queue traffic at
           4 TB/s, 5 us per extra launch, and the finishing pass.
Usage: python tests/tools/repack_model.py [--every 4] [--size 3840x2160]"""
import argparse
import heapq
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from loltracer_amd import scene as S  # noqa: E402


def schedule(tasks, lanes, refill_at):
    """steps a group of `lanes` lanes executes for `tasks` (list of step counts, in queue order), refilling idle lanes only
    when at least `refill_at` of them are idle (or the queue would otherwise starve the group)."""
    tasks = [t for t in tasks if t > 0]
    if not tasks:
        return 0
    q = list(tasks)
    qi = 0
    now = 0
    busy = []                                   # finish times
    while qi < len(q) or busy:
        idle = lanes - len(busy)
        if qi < len(q) and (idle >= refill_at or not busy):
            take = min(idle, len(q) - qi)
            for _ in range(take):
                heapq.heappush(busy, now + q[qi])
                qi += 1
        # advance to the next finish
        now = heapq.heappop(busy)
        while busy and busy[0] == now:
            heapq.heappop(busy)
    return now


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--every", type=int, default=4, help="sample every n-th band of 4 rows")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    sc = S.Scene.parse_file(a.scene)
    nl = sc.c.n_lights if hasattr(sc.c, "n_lights") else 2
    res = {}
    acc = dict(now=0, need=0, march=0, march_need=0, pixels=0, rays=0)
    pol = {"queue_r1": (1, 1), "queue_r16": (1, 16), "queue_r32": (1, 32), "block4_r1": (4, 1), "block4_r16": (4, 16), "longest_first_r1": (1, 1)}
    tot = {k: 0 for k in pol}
    compact = {}
    KS = (8, 16, 24, 32, 48, 64)
    glob = {K: dict(pass_a=0, queues=[[] for _ in range(128 // K + 1)], surv_rays=0, pend_px=0) for K in KS}
    for band in range(0, h // 4, a.every):
        y0 = band * 4
        _, _, st = O.render_rows(sc, w, h, y0, y0 + 4, 256, want_steps=True)
        st = st[y0:y0 + 4].astype(np.int64)
        hit = st[..., 2] != 0
        dark = st[..., 3]
        per_light = []
        for li in range(4):
            s = st[..., 8 + li].copy()                              # settled steps (the kernel's exit)
            s[((dark >> li) & 1) == 1] = 0                          # dark lanes do not march
            s[~hit] = 0                                             # escaped rays do not march
            per_light.append(s)
        per_light = np.stack(per_light, axis=-1)                    # [4, w, lights]
        npx = w // 16
        pl = per_light.reshape(4, npx, 16, 4).transpose(1, 0, 2, 3).reshape(npx, 64, 4)     # [patch, lane, light]
        acc["now"] += int(pl.max(axis=1).sum())
        acc["need"] += int(pl.sum())
        acc["rays"] += int((pl > 0).sum())
        acc["pixels"] += 4 * w
        m = st[..., 0].reshape(4, npx, 16).transpose(1, 0, 2).reshape(npx, 64)
        acc["march"] += int(m.max(axis=1).sum()) * 64
        acc["march_need"] += int(m.sum())
        for name, (waves, r) in pol.items():
            for p0 in range(0, npx - waves + 1, waves):
                grp = pl[p0:p0 + waves]                              # [waves, 64, 4]
                # queue order: light 0 of every lane, then light 1, ...
                tasks = [int(grp[wv, ln, li]) for li in range(4) for wv in range(waves) for ln in range(64)]
                if name.startswith("longest"):
                    tasks.sort(reverse=True)
                tot[name] += schedule(tasks, 64 * waves, r) * waves   # wave-steps
        # global chunked compaction: pass A caps every light's march of a wave at K steps; survivors go to the queue in patch order
        for K in KS:
            G = glob[K]
            G["pass_a"] += int(np.minimum(pl.max(axis=1), K).sum())            # wave-steps of the render kernel's shadow loops
            rem = pl - K                                                         # [patch, lane, light]
            surv = rem[rem > 0]                                                  # arrival order: patch, lane, light
            G["queues"][0].append(surv)
            G["surv_rays"] += int(surv.size)
            G["pend_px"] += int(((rem > 0).any(axis=2)).sum())
        # ideal compaction inside a block of N waves: every lane runs its own lights back to back, and at every step the
        # block issues ceil(live lanes / 64) waves
        busy = pl.sum(axis=2)                                         # [patch, lane] steps of the lane's chain
        for N in (1, 2, 4, 8, 16):
            for p0 in range(0, npx - N + 1, N):
                b = np.sort(busy[p0:p0 + N].ravel())[::-1]            # descending
                # live(t) = number of lanes with busy > t; cost = sum_t ceil(live(t) / 64) = sum over k of b[64 k] (the (64k+1)-th longest)
                compact[N] = compact.get(N, 0) + int(b[::64].sum())
    px = acc["pixels"]
    out = dict(scene=os.path.basename(a.scene), size=a.size, sampled_pixels=px,
               shadow_steps_needed_per_pixel=acc["need"] / px,
               shadow_wave_steps_per_pixel_now=acc["now"] / px,
               shadow_lane_efficiency_now=acc["need"] / (acc["now"] * 64),
               march_lane_efficiency=acc["march_need"] / acc["march"])
    for k, v in tot.items():
        out[k] = dict(wave_steps_per_pixel=v / px, lane_efficiency=acc["need"] / (v * 64))
    for N, v in sorted(compact.items()):
        out["ideal_compaction_block%d" % N] = dict(wave_steps_per_pixel=v / px, lane_efficiency=acc["need"] / (v * 64))
    # ---- global chunked compaction (see the docstring): wave-steps of all passes, traffic, launches; scaled to the whole frame
    frame_px = w * h
    scale = frame_px / px
    other_wave_steps = acc["march"] / 64 + 4 * px / 64 / 0.96     # primary march (executed) + normal taps (waves with a hit: ~all)
    now_total = acc["now"] + other_wave_steps
    out["wave_evaluations_per_pixel_now"] = dict(shadow=acc["now"] / px, march_and_normal=other_wave_steps / px, total=now_total / px)
    for K in KS:
        G = glob[K]
        q = np.concatenate(G["queues"][0]) if G["queues"][0] else np.zeros(0, dtype=np.int64)
        rounds, wave_steps_b, rays_moved = 0, 0, 0
        while q.size:
            rounds += 1
            rays_moved += int(q.size)
            n_waves = (q.size + 63) // 64
            padded = np.zeros(n_waves * 64, dtype=np.int64)
            padded[:q.size] = q
            wave_steps_b += int(np.minimum(padded.reshape(n_waves, 64).max(axis=1), K).sum())
            q = q - K
            q = q[q > 0]
        shadow_ws = G["pass_a"] + wave_steps_b
        # a step of pass B is one SDF evaluation like any other; pass C costs no evaluation.  Traffic: a survivor is written and
        # read once per round it lives (40 B each way) and its factor written back (8 B); a pending pixel parks 64 B and reads it back
        bytes_frame = (rays_moved * 80 + G["surv_rays"] * 8 + G["pend_px"] * 128) * scale
        t_traffic_ms = bytes_frame / 4e12 * 1e3
        t_launch_ms = (rounds + 1) * 5e-3
        # what a wave-step costs: the frame's measured 1.06 ms over its executed wave-steps (march + normal + shadow)
        ms_per_wave_step = 1.06 / (now_total * scale)
        t_now = 1.06
        t_new = (shadow_ws + other_wave_steps) * scale * ms_per_wave_step + t_traffic_ms + t_launch_ms
        out["global_K%d" % K] = dict(
            shadow_wave_steps_per_pixel=shadow_ws / px, shadow_lane_efficiency=acc["need"] / (shadow_ws * 64),
            pass_a=G["pass_a"] / px, pass_b=wave_steps_b / px, rounds=rounds,
            survivors_share_of_rays=G["surv_rays"] / max(int((np.concatenate([np.zeros(1)])).size), 1) if False else G["surv_rays"] / max(acc["rays"], 1),
            pending_share_of_pixels=G["pend_px"] / px,
            wave_evaluations_saved_share=1 - (shadow_ws + other_wave_steps) / now_total,
            queue_traffic_mb_per_frame=bytes_frame / 1e6, traffic_ms=t_traffic_ms, launch_ms=t_launch_ms,
            modelled_frame_ms=t_new, modelled_gain=t_now / t_new - 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
