#!/usr/bin/env python3
"""Round 5: the register budget (amdgpu_waves_per_eu minimum) of the INLINED kernel on mid-size scenes, which round 2 swept on
chains only and before culling clusters, saturation shortcuts and pixel dealing existed: 504-op chain, fields of 250 / 420
objects, 1080p, LOL_GPU_WAVES_PER_EU = 2,8 / 3,8 / 4,8 (default above 96 ops) / 6,8 / 8,8."""
import json
import os
os.environ["LOL_GPU_CACHE_DIR"] = ""
os.environ["LOL_GPU_SPEC_INLINE_MAX"] = "100000"
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from loltracer_amd import gpu, scene as S  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import large_scene_ab as L  # noqa: E402


def run(sc, w, h, budget):
    os.environ["LOL_GPU_WAVES_PER_EU"] = budget
    r = gpu.Renderer(0)
    r.prepare(sc)
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e1.record()
    torch.cuda.synchronize()
    r.close()
    return round(4 * w * h / e0.elapsed_time(e1) / 1e3, 1)


def main():
    w, h = 1920, 1080
    for name, sc in (("chain of 250 smooth unions", L.chain_scene(250)),
                     ("field of 250 objects", S.Scene.parse_string(F.big_field_scene(250, 9, 2))),
                     ("field of 420 objects", S.Scene.parse_string(F.big_field_scene(420, 9, 2)))):
        out = {"scene": name, "n_ops": sc.flatten().n_ops}
        for b in ("2,8", "3,8", "4,8", "6,8", "8,8"):
            out[b] = run(sc, w, h, b)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h); only when RUN, not when a test imports the scene builders
    main()
