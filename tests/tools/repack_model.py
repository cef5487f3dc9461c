#!/usr/bin/env python3
"""What would re-packing shadow rays inside a wave (or a block of waves) buy?  CPU model on the oracle's per-pixel, per-light
step counts (test infrastructure: runs the oracle, not the product).  For every 16x4 patch of sampled 4-row bands of C3:
  now      the wave marches light after light, each for the maximum over its lanes            (sum_l max_lanes steps[l])
  queue    the (pixel, light) marches of the patch are tasks; a lane that finishes takes the next one (list scheduling,
           refill only when at least R lanes are idle or nothing else is left)
  block-N  the same with N waves sharing one queue (N patches side by side)
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
    acc = dict(now=0, need=0, march=0, march_need=0, pixels=0)
    pol = {"queue_r1": (1, 1), "queue_r16": (1, 16), "queue_r32": (1, 32), "block4_r1": (4, 1), "block4_r16": (4, 16), "longest_first_r1": (1, 1)}
    tot = {k: 0 for k in pol}
    compact = {}
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
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
