#!/usr/bin/env python3
"""Would it pay to let a wave shade the 64 pixels of a REGION that cost about the same, instead of a 16x4 rectangle?
For a camera that stands still the per-pixel step counts of the frame before are exact, so pixels can be dealt to waves by
cost ahead of time (a permutation per region, no compaction at run time).  CPU model on the oracle's per-pixel, per-light step
counts (test infrastructure): for regions of W x H pixels (W*H/64 waves), the wave-evaluations a frame executes
    sum over waves of [ max_lanes march + 4 (if any lane hit) + sum over lights of max_lanes shadow steps ]
when the region's pixels are dealt to its waves  (a) as 16x4 rectangles (now),  (b) sorted by their total step count,
(c) sorted by shadow steps of light 0 then light 1 (lexicographic on the coarse counts).
Usage: python tests/tools/sorted_region_model.py [--size 3840x2160] [--every 4]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from loltracer_amd import scene as S  # noqa: E402


def wave_cost(march, hit, sh):
    """march [n,64], hit [n,64] bool, sh [n,64,L] → executed evaluations per wave [n]"""
    return march.max(axis=1) + 4 * hit.any(axis=1) + sh.max(axis=1).sum(axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--every", type=int, default=4, help="sample every n-th strip of 64 rows")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    sc = S.Scene.parse_file(a.scene)
    regions = [(16, 4), (32, 8), (64, 8), (64, 16), (128, 32)]
    tot = {f"{rw}x{rh}": dict(now=0, by_total=0, by_lights=0) for rw, rh in regions}
    need = 0
    px = 0
    for strip in range(0, h // 64, a.every):
        y0 = strip * 64
        _, _, st = O.render_rows(sc, w, h, y0, y0 + 64, 256, want_steps=True)
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
        sh = np.stack(sh, axis=-1)                                   # [64, w, 4]
        need += int(march.sum() + 4 * hit.sum() + sh.sum())
        px += 64 * w
        for rw, rh in regions:
            if w % rw or 64 % rh:
                continue
            # regions [n_regions, rh*rw]
            def cut(x):
                tail = x.shape[2:]
                return x.reshape(64 // rh, rh, w // rw, rw, *tail).swapaxes(1, 2).reshape(-1, rh * rw, *tail)
            m, hh, s4 = cut(march), cut(hit), cut(sh)
            n_reg, npx = m.shape
            nw = npx // 64
            key = f"{rw}x{rh}"
            # (a) 16x4 rectangles inside the region
            def rect(x):
                tail = x.shape[2:]
                return x.reshape(n_reg, rh // 4, 4, rw // 16, 16, *tail).swapaxes(2, 3).reshape(n_reg * nw, 64, *tail)
            tot[key]["now"] += int(wave_cost(rect(m), rect(hh), rect(s4)).sum())
            # (b) sorted by total steps
            total = m + 4 * hh + s4.sum(axis=2)
            idx = np.argsort(total, axis=1, kind="stable")
            g = lambda x: np.take_along_axis(x, idx if x.ndim == 2 else idx[..., None], axis=1).reshape(n_reg * nw, 64, *x.shape[2:])
            tot[key]["by_total"] += int(wave_cost(g(m), g(hh), g(s4)).sum())
            # (b') the same with runs of u horizontally adjacent pixels kept together (a wave then writes whole 4u-byte pieces of rows)
            for u in (2, 4, 8):
                if rw % u:
                    continue
                def runs(x):
                    tail = x.shape[2:]
                    return x.reshape(n_reg, rh, rw // u, u, *tail).reshape(n_reg, rh * rw // u, u, *tail)
                tu = runs(total).max(axis=2)                           # [n_reg, n_runs]
                ridx = np.argsort(tu, axis=1, kind="stable")
                def gr(x):
                    y = runs(x)
                    ix = ridx.reshape(n_reg, -1, *([1] * (y.ndim - 2)))
                    y = np.take_along_axis(y, ix, axis=1)
                    return y.reshape(n_reg * nw, 64, *x.shape[2:])
                tot[key].setdefault("runs_of_%d" % u, 0)
                tot[key]["runs_of_%d" % u] += int(wave_cost(gr(m), gr(hh), gr(s4)).sum())
            # (c) lexicographic: light 0 steps (coarse, /4), then light 1, then march
            k = (s4[..., 0] // 4) * 1_000_000 + (s4[..., 1] // 4) * 1000 + m
            idx = np.argsort(k, axis=1, kind="stable")
            tot[key]["by_lights"] += int(wave_cost(g(m), g(hh), g(s4)).sum())
    out = dict(scene=os.path.basename(a.scene), size=a.size, sampled_pixels=px, needed_evaluations_per_pixel=need / px)
    for key, v in tot.items():
        if v["now"] == 0:
            continue
        out[key] = {name: dict(wave_evaluations_per_pixel=val / px, lane_efficiency=need / (val * 64),
                               gain_over_rectangles=v["now"] / val - 1) for name, val in v.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
