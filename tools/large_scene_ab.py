#!/usr/bin/env python3
"""Large scenes (long smooth-union chains, 42 ... 256 ops) on the three forms of the SDF: specialised with the SDF
inlined into every loop, specialised with the SDF as one out-of-line function, and the macro-op interpreter.
Prints one JSON line per scene: Mpixels/s of each, render_prepare seconds, and whether the frames are identical.
Run on the GPU box:  python tools/large_scene_ab.py [--size 1920x1080]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loltracer_amd import gpu, scene as S  # noqa: E402


def num(x):
    return ("%.5f" % x).rstrip("0").rstrip(".")


def chain_scene(n_unions: int, seed: int = 3) -> S.Scene:
    rng = np.random.default_rng(seed)
    body = "sphere { point = (0,0,-5), radius = 1 }"
    for _ in range(n_unions):
        c = rng.normal(size=3) * [3, 2, 3] + [0, 0, -7]
        body = "smooth_union { smoothness = 1, a = sphere { point = (%s, %s, %s), radius = %s }, b = %s }" % (
            num(c[0]), num(c[1]), num(c[2]), num(rng.uniform(0.3, 1.2)), body)
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (0,0,0) },"
            " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
            "scene { camera { point = (0, 3, 6), direction = (0, -0.25, -1), fov = 100 },"
            " point_light { point = (0,9,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
            + body.replace("{", "{ material = #1,", 1) + ", plane { y = -4, material = #1 } }")
    return S.Scene.parse_string(text)


def run(sc, w, h, mode, inline_max, frames=5):
    os.environ["LOL_GPU_SPEC_INLINE_MAX"] = str(inline_max)
    r = gpu.Renderer(0, specialize=mode)
    t0 = time.perf_counter()
    r.prepare(sc)
    prep = time.perf_counter() - t0
    side = torch.cuda.Stream()
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    with torch.cuda.stream(side):
        r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        ev = []
        for _ in range(frames):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
            e1.record()
            ev.append((e0, e1))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    name = r.kernel_name()
    r.close()
    return dict(kernel=name, mpixels_per_s=round(w * h / ms / 1e3, 1), ms=round(ms, 3), prepare_s=round(prep, 2)), buf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--unions", default="20,69,126,250,510")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    for n in (int(v) for v in a.unions.split(",")):
        sc = chain_scene(n)
        out = {"scene": f"chain of {n} smooth unions", "n_ops": sc.flatten().n_ops, "size": a.size}
        res_in, f_in = run(sc, w, h, 1, 1 << 30)
        res_ool, f_ool = run(sc, w, h, 1, 0)
        res_it, f_it = run(sc, w, h, 4, 0)
        out["spec_inline"], out["spec_out_of_line"], out["interp"] = res_in, res_ool, res_it
        out["frames_identical"] = bool(torch.equal(f_in, f_ool) and torch.equal(f_in, f_it))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment"); only when RUN, not when a test imports the scene builders
    main()
