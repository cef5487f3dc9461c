#!/usr/bin/env python3
"""One object that is a balanced smooth-union tree of 2^d spheres (d = 5 ... 9: 64 ... 1024 ops), specialised kernel
with saturation culling inside the union off / on for operands of at least N primitives (LOL_GPU_SAT_CULL_MIN_PRIMS).
One JSON line per scene: Mpixels/s per setting and whether the frames are identical.
Run on the GPU box:  python tools/tree_scene_ab.py [--size 1920x1080] [--depths 5,7,9] [--min-prims 0,1,2,4,8]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loltracer_amd import gpu, scene as S  # noqa: E402


def num(x):
    return ("%.5f" % x).rstrip("0").rstrip(".")


def tree_scene(depth: int, smooth: float = 0.5, seed: int = 11) -> S.Scene:
    rng = np.random.default_rng(seed)

    def tree(d, lo, hi):
        """spatially coherent: each level halves the box along its longest axis"""
        if d == 0:
            c = rng.uniform(lo, hi)
            return "sphere { point = (%s, %s, %s), radius = %s }" % (num(c[0]), num(c[1]), num(c[2]), num(rng.uniform(0.2, 0.6)))
        ax = int(np.argmax(hi - lo))
        mid = 0.5 * (lo[ax] + hi[ax])
        h1, l2 = hi.copy(), lo.copy()
        h1[ax] = mid
        l2[ax] = mid
        return "smooth_union { smoothness = %s, a = %s, b = %s }" % (num(smooth), tree(d - 1, lo, h1), tree(d - 1, l2, hi))
    body = tree(depth, np.array([-9.0, -1.0, -16.0]), np.array([9.0, 4.0, -3.0]))
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.02,.02,.02) },"
            " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
            "scene { camera { point = (0, 3, 5), direction = (0, -0.2, -1), fov = 100 },"
            " point_light { point = (0,12,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
            + body.replace("{", "{ material = #1,", 1) + ", plane { y = -2, material = #1 } }")
    return S.Scene.parse_string(text)


def run(sc, w, h, min_prims, frames=3):
    os.environ["LOL_GPU_SAT_CULL_MIN_PRIMS"] = str(min_prims)
    r = gpu.Renderer(0)
    r.prepare(sc)
    side = torch.cuda.Stream()
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    with torch.cuda.stream(side):
        r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(frames):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / frames
    assert r.kernel_name() == "lol_render_spec"
    r.close()
    return round(w * h / ms / 1e3, 1), buf


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--depths", default="5,7,9")
    ap.add_argument("--min-prims", default="0,1,2,4,8,16")
    ap.add_argument("--smooth", type=float, default=0.5)
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    for d in (int(x) for x in a.depths.split(",")):
        sc = tree_scene(d, a.smooth)
        out = dict(spheres=2 ** d, ops=sc.flatten().n_ops, smoothness=a.smooth, size=a.size, mpixels_per_s={})
        ref = None
        same = True
        for mp in (int(x) for x in a.min_prims.split(",")):
            v, f = run(sc, w, h, mp)
            out["mpixels_per_s"][str(mp)] = v
            if ref is None:
                ref = f
            same = same and torch.equal(ref, f)
        out["identical"] = bool(same)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment"); only when RUN, not when a test imports the scene builders
    main()
