#!/usr/bin/env python3
"""Would a MOVING camera gain from pixels dealt to waves by what they cost in the frame BEFORE (round-4 review, next #2)?
For a camera that stands still the previous frame's per-pixel step counts are exact and dealing the pixels of every 64x16
region to its sixteen waves by cost removes 11 % of the wave-evaluations (sorted_region_model.py).  While the camera moves
those counts are one frame old.  CPU model on the oracle's per-pixel, per-light step counts (test infrastructure): frames of
the 256-frame orbit of BASELINE.json's config 5 taken `stride` frames apart (1.4 degrees per orbit frame; the reference's arrow
keys turn 5.7 degrees a frame, main.c:70-112); the pixels of frame i are dealt
    (a) as 16x4 rectangles (what a new view gets today),
    (b) sorted by frame i's own total step count (the repeated view: exact),
    (c) sorted by the total step count the SAME PIXEL had in frame i - stride (stale dealing),
and the wave-evaluations executed are  sum over waves of [max march + 4 (any lane hit) + sum over lights of max shadow steps].
The review's bar for building (c): >= 8 % fewer wave-evaluations than (a).
Usage: python tests/tools/stale_dealing_model.py [--size 3840x2160] [--every 4] [--frames 0,64,128,192] [--strides 1,2,4]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
import bench  # noqa: E402
from loltracer_amd import scene as S  # noqa: E402

RW, RH = 64, 16


def strip_counts(sc, w, h, y0, cam):
    _, _, st = O.render_rows(sc, w, h, y0, y0 + 64, 256, want_steps=True, camera=cam)
    st = st[y0:y0 + 64].astype(np.int64)
    hit = st[..., 2] != 0
    dark = st[..., 3]
    march = st[..., 0]
    sh = []
    for li in range(4):
        s = st[..., 8 + li].copy()
        s[((dark >> li) & 1) == 1] = 0
        s[~hit] = 0
        sh.append(s)
    return march, hit, np.stack(sh, axis=-1)


def wave_cost(march, hit, sh):
    return march.max(axis=1) + 4 * hit.any(axis=1) + sh.max(axis=1).sum(axis=1)


def cut(x, w):
    tail = x.shape[2:]
    return x.reshape(64 // RH, RH, w // RW, RW, *tail).swapaxes(1, 2).reshape(-1, RH * RW, *tail)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--every", type=int, default=4, help="sample every n-th strip of 64 rows")
    ap.add_argument("--frames", default="0,64,128,192")
    ap.add_argument("--strides", default="1,2,4")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    strides = [int(x) for x in a.strides.split(",")]
    frames = [int(x) for x in a.frames.split(",")]
    tot = {s: dict(rectangles=0, own_cost=0, stale_cost=0, need=0, px=0) for s in strides}
    for f in frames:
        for strip in range(0, h // 64, a.every):
            y0 = strip * 64
            m, hit, sh = strip_counts(sc, w, h, y0, bench.orbit_camera(f, 256))
            cm, chit, csh = cut(m, w), cut(hit, w), cut(sh, w)
            n_reg = cm.shape[0]
            nw = RW * RH // 64
            own = cm + 4 * chit + csh.sum(axis=2)

            def rect(x):
                tail = x.shape[2:]
                return x.reshape(n_reg, RH // 4, 4, RW // 16, 16, *tail).swapaxes(2, 3).reshape(n_reg * nw, 64, *tail)

            def dealt(key):
                idx = np.argsort(key, axis=1, kind="stable")
                g = lambda x: np.take_along_axis(x, idx if x.ndim == 2 else idx[..., None], axis=1).reshape(n_reg * nw, 64, *x.shape[2:])
                return int(wave_cost(g(cm), g(chit), g(csh)).sum())
            rect_cost = int(wave_cost(rect(cm), rect(chit), rect(csh)).sum())
            own_cost = dealt(own)
            for s in strides:
                pm, phit, psh = strip_counts(sc, w, h, y0, bench.orbit_camera((f - s) % 256, 256))
                prev = cut(pm, w) + 4 * cut(phit, w) + cut(psh, w).sum(axis=2)
                T = tot[s]
                T["rectangles"] += rect_cost
                T["own_cost"] += own_cost
                T["stale_cost"] += dealt(prev)
                T["need"] += int(own.sum())
                T["px"] += 64 * w
            print(f"frame {f} strip {strip} done", file=sys.stderr, flush=True)
    out = dict(scene="scene4.lol", size=a.size, region="64x16", orbit_frames=frames, sampled_strips_every=a.every,
               degrees_per_orbit_frame=360.0 / 256, reference_arrow_key_degrees_per_frame=5.71)
    for s, T in tot.items():
        out[f"stride_{s}"] = dict(
            degrees_per_frame=round(s * 360.0 / 256, 2),
            wave_evaluations_per_pixel=dict(rectangles=T["rectangles"] / T["px"], own_cost=T["own_cost"] / T["px"], stale_cost=T["stale_cost"] / T["px"]),
            lane_efficiency=dict(rectangles=T["need"] / (64 * T["rectangles"]), own_cost=T["need"] / (64 * T["own_cost"]), stale_cost=T["need"] / (64 * T["stale_cost"])),
            fewer_wave_evaluations_than_rectangles=dict(own_cost=1 - T["own_cost"] / T["rectangles"], stale_cost=1 - T["stale_cost"] / T["rectangles"]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
