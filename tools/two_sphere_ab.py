#!/usr/bin/env python3
"""Two-sphere culling bounds (lol_gpu.hip, cluster_bounds) on a scene built for them: ONE object that is a smooth union of two
groups of spheres `gap` apart over a plane, 1920x1080.  Run twice: LOL_GPU_CULL_TWO_SPHERES=1 / =0.  Prints Mpixels/s per gap."""
import json
import os
import sys
if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loltracer_amd import gpu, scene as S  # noqa: E402


def scene(gap):
    def blob(x):
        return ("smooth_union { smoothness = 1, a = sphere { point = (%g, 0.5, -8), radius = 1.2 }, b = smooth_union { smoothness = 1, "
                "a = sphere { point = (%g, 1.5, -9), radius = 0.9 }, b = sphere { point = (%g, 0, -7), radius = 0.8 } } }" % (x, x + 0.8, x - 0.6))
    obj = "smooth_union { smoothness = 1, a = %s, b = %s }" % (blob(-gap / 2), blob(gap / 2))
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (0,0,0) },"
            " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
            "scene { camera { point = (0, 4, 8), direction = (0, -0.25, -1), fov = 110 },"
            " point_light { point = (0,12,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
            + obj.replace("{", "{ material = #1,", 1) + ", plane { y = -1, material = #1 } }")
    return S.Scene.parse_string(text)


w, h = 1920, 1080
out = {"two_spheres": os.environ.get("LOL_GPU_CULL_TWO_SPHERES", "1")}
for gap in (4, 12, 30):
    r = gpu.Renderer(0)
    r.prepare(scene(gap))
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(20):
        r.render_into(buf.data_ptr(), w, h)
    r.sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    s = torch.cuda.current_stream().cuda_stream
    for a, b in ev:
        a.record(); r.render_into(buf.data_ptr(), w, h, stream=s); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
    out[f"gap_{gap}"] = {"mpixels_per_s": round(w * h / ms / 1e3, 1), "checksum": int(buf.to(torch.int64).sum().item())}
    r.close()
print(json.dumps(out))
