#!/usr/bin/env python3
"""Scenes of many separate objects (a field of spheres, rounded boxes and small smooth unions over a plane): the
specialised kernel with culling off, with the single run of all bounded objects (LOL_GPU_CULL_CLUSTERS=0), and with the
k-d clusters (default); the interpreter with and without clusters.  One JSON line per scene: Mpixels/s of each and whether the frames are identical.
Run on the GPU box:  python tools/flat_scene_ab.py [--size 1920x1080] [--objects 16,64,200]"""
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


def pt(v):
    return "(%s, %s, %s)" % tuple(num(x) for x in v)


def field_scene(n_objects: int, seed: int = 5) -> S.Scene:
    """n objects scattered over a 24 x 24 field in front of the camera, on a plane, two lights."""
    rng = np.random.default_rng(seed)
    objs = []
    for _ in range(n_objects):
        c = np.array([rng.uniform(-12, 12), rng.uniform(-0.5, 2.0), rng.uniform(-26, -2)])
        k = rng.integers(4)
        if k == 0:
            objs.append("sphere { material = #%d, point = %s, radius = %s }" % (rng.integers(1, 4), pt(c), num(rng.uniform(0.3, 0.9))))
        elif k == 1:
            objs.append("box { material = #%d, point = %s, point2 = %s, radius = %s }" % (
                rng.integers(1, 4), pt(c), pt(rng.uniform(0.2, 0.7, 3)), num(rng.uniform(0, 0.2))))
        else:
            d = rng.normal(size=3) * 0.5
            body = "smooth_union { material = #%d, smoothness = 0.5, a = sphere { point = %s, radius = %s }, b = sphere { point = %s, radius = %s } }" % (
                rng.integers(1, 4), pt(c), num(rng.uniform(0.3, 0.7)), pt(c + d), num(rng.uniform(0.3, 0.7)))
            if k == 3:
                body = body.replace("b = sphere", "b = smooth_union { smoothness = 0.25, a = box { point = %s, point2 = (0.3, 0.3, 0.3), radius = 0.1 }, b = sphere" % pt(c - d), 1)
                body = body[:-1] + "} }"
            objs.append(body)
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.05,.05,.08) },"
            " { shininess = 8, diffuse = (.6,.3,.2), specular = (.2,.2,.2), ambient = (.1,.1,.1) },"
            " { shininess = 16, diffuse = (.2,.6,.3), specular = (.3,.3,.3), ambient = (.1,.1,.1) },"
            " { shininess = 4, diffuse = (.3,.3,.7), specular = (.1,.1,.1), ambient = (.1,.1,.1) } }\n"
            "scene { ambient { color = (.2,.2,.2) }, camera { point = (0, 5, 6), direction = (0, -0.35, -1), fov = 90 },"
            " point_light { point = (8,12,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) },"
            " point_light { point = (-9,7,-20), diffuse_intensity = (1,1,2), specular_intensity = (1,1,1) },"
            " plane { y = -1, material = #1 }, " + ", ".join(objs) + " }")
    return S.Scene.parse_string(text)


def run(sc, w, h, cull, clusters, frames=5, mode=1):
    os.environ["LOL_GPU_CULL_CLUSTERS"] = str(clusters)
    r = gpu.Renderer(0, specialize=mode)
    r.set_cull(cull)
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
    ap.add_argument("--objects", default="4,16,64,150,250")
    ap.add_argument("--leaves", default="3", help="k-d leaf sizes to try (LOL_GPU_CULL_CLUSTERS), comma separated")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    for n in (int(x) for x in a.objects.split(",")):
        sc = field_scene(n)
        prog = sc.flatten()
        off, f0 = run(sc, w, h, 0, 0)
        flat, f1 = run(sc, w, h, 1, 0)
        same = torch.equal(f0, f1)
        kd = {}
        for leaf in (int(x) for x in a.leaves.split(",")):
            kd[leaf], f2 = run(sc, w, h, 1, leaf)
            same = same and torch.equal(f0, f2)
        out = dict(objects=n, ops=prog.n_ops, size=a.size, no_culling=off, one_run=flat)
        if len(kd) == 1:
            out["kd_clusters"] = list(kd.values())[0]
        else:
            out["kd_clusters_by_leaf"] = {str(k): v["mpixels_per_s"] for k, v in kd.items()}
        i0, g0 = run(sc, w, h, 1, 0, mode=4)
        i1, g1 = run(sc, w, h, 1, int(a.leaves.split(",")[0]), mode=4)
        out["interp_one_run"], out["interp_kd_clusters"] = i0["mpixels_per_s"], i1["mpixels_per_s"]
        out["identical"] = bool(same and torch.equal(f0, g0) and torch.equal(f0, g1))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment"); only when RUN, not when a test imports the scene builders
    main()
