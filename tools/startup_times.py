#!/usr/bin/env python3
"""Tiered start-up on the GPU box: render_prepare's cost, time to the first frame and to the scene's own kernel for scenes
of growing size (cold render_prepare < 50 ms for the 1024-op scene).
    python tools/startup_times.py > gpurun_out/r4_startup.json
Each scene is measured in a process-fresh state for the code caches (unique random scenes, disk cache off)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from loltracer_amd import gpu, scene as S


def tree_scene(depth, seed):
    rng = np.random.default_rng(seed)

    def tree(d):
        if d == 0:
            return "sphere { point = (%.3f, %.3f, %.3f), radius = %.3f }" % (*(rng.normal(size=3) * [4, 2, 3] + [0, 0, -9]), rng.uniform(0.2, 0.8))
        return "smooth_union { smoothness = 0.5, a = %s, b = %s }" % (tree(d - 1), tree(d - 1))
    return S.Scene.parse_string(
        "materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.02,.02,.02) },"
        " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
        "scene { camera { point = (0, 1, 4), direction = (0, -0.1, -1), fov = 100 },"
        " point_light { point = (0,9,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
        + tree(depth).replace("{", "{ material = #1,", 1) + " }")


def main():
    seed = int(time.time()) % 100000
    scenes = [("scene4.lol", S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))),
              ("scene.lol", S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene.lol"))),
              ("tree of 128 spheres (256 ops)", tree_scene(7, seed)),
              ("tree of 512 spheres (1024 ops)", tree_scene(9, seed + 1)),
              ("tree of 2048 spheres (4096 ops)", tree_scene(11, seed + 2))]
    gpu.Renderer(0).close()                              # HIP initialisation is not what is being measured
    out = []
    for name, sc in scenes:
        rec = {"scene": name, "ops": sc.flatten().n_ops}
        rec.update(bench.startup_times(sc, 1920, 1080, 256, 0))
        rec.pop("note", None)
        out.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)
    print(json.dumps({"what": "tiered start-up (lol_gpu_upload_program returns with hipRTC on a host thread), 1920x1080 first frame",
                      "scenes": out}, indent=1))


if __name__ == "__main__":
    main()
