#!/usr/bin/env python3
"""How much of a wave's work is done for lanes that no longer need it?  Renders the C3 frame with the per-pixel step
counters on (march steps, shadow steps) and compares, per wave patch, the steps the wave executes (the maximum over its
64 pixels) with the steps its pixels need (the mean), for several patch shapes.
Run on the GPU box:  python tools/divergence.py [--scene tests/golden/scenes/scene4.lol] [--size 3840x2160]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from loltracer_amd import gpu, scene as S  # noqa: E402
from test_gpu_parity import gpu_render  # noqa: E402


def patches(a, pw, ph):
    h, w = a.shape
    return a[: h // ph * ph, : w // pw * pw].reshape(h // ph, ph, w // pw, pw).transpose(0, 2, 1, 3).reshape(h // ph, w // pw, ph * pw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    ap.add_argument("--size", default="3840x2160")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    sc = S.Scene.parse_file(a.scene)
    r = gpu.Renderer(0)
    g = gpu_render(torch, r, sc, w, h)
    march = (g["steps"] & 0xFFFF).astype(np.float64)
    shadow = (g["steps"] >> 16).astype(np.float64)
    hit = g["id"] != 0
    out = dict(scene=os.path.basename(a.scene), size=a.size, march_steps_per_pixel=march.mean(), shadow_steps_per_pixel=shadow.mean(),
               hit_fraction=float(hit.mean()))
    for pw, ph in ((16, 4), (8, 8), (4, 16), (2, 32), (32, 2), (64, 1), (4, 4), (2, 2), (1, 1)):
        pm, ps, phit = patches(march, pw, ph), patches(shadow, pw, ph), patches(hit.astype(np.float64), pw, ph)
        lanes = pw * ph
        # what the wave executes: max over lanes of march steps; the 4 normal taps if any lane hit; shadow: the kernel marches
        # the lights one after another, so the per-pixel total is only a lower bound of the wave's count — use max of totals
        exe = pm.max(axis=2) + 4 * (phit.max(axis=2) > 0) + ps.max(axis=2)
        need = pm.mean(axis=2) + 4 * phit.mean(axis=2) + ps.mean(axis=2)
        out["%dx%d" % (pw, ph)] = dict(lanes=lanes, wave_evals_per_pixel=float(exe.mean()), needed_evals_per_pixel=float(need.mean()),
                                       lane_efficiency=float(need.sum() / exe.sum()),
                                       march_only=float(pm.mean(axis=2).sum() / pm.max(axis=2).sum()),
                                       shadow_only=float(ps.mean(axis=2).sum() / max(ps.max(axis=2).sum(), 1)))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
