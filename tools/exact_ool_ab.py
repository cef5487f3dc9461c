#!/usr/bin/env python3
"""Round 5 experiment: the plain (fallback) SDF out of line where the fast SDF is inlined (LOL_GPU_EXACT_OOL=N: for programs above N
(the library hunk that read LOL_GPU_EXACT_OOL is not kept: LABNOTES.md §8)
ops) — compile seconds and Mpixels/s of the INLINED form, scene4 (C3), a 504-op chain and fields of 120 / 250 objects, 1080p / 4K."""
import json
import os
os.environ["LOL_GPU_CACHE_DIR"] = ""
os.environ["LOL_GPU_SPEC_INLINE_MAX"] = "100000"
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from loltracer_amd import gpu, scene as S  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import large_scene_ab as L  # noqa: E402


def run(sc, w, h, exact_ool):
    if exact_ool is None:
        os.environ.pop("LOL_GPU_EXACT_OOL", None)
    else:
        os.environ["LOL_GPU_EXACT_OOL"] = exact_ool
    r = gpu.Renderer(0)
    t0 = time.perf_counter()
    r.prepare(sc)
    prep = time.perf_counter() - t0
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(20):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e1.record()
    torch.cuda.synchronize()
    r.close()
    return dict(mpixels_per_s=round(10 * w * h / e0.elapsed_time(e1) / 1e3, 1), prepare_s=round(prep, 2)), buf


def main():
    scenes = [("scene4.lol at 4K", S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol")), 3840, 2160),
              ("chain of 250 smooth unions", L.chain_scene(250), 1920, 1080),
              ("field of 120 objects", S.Scene.parse_string(F.big_field_scene(120, 9, 2)), 1920, 1080),
              ("field of 250 objects", S.Scene.parse_string(F.big_field_scene(250, 9, 2)), 1920, 1080)]
    for name, sc, w, h in scenes:
        out = {"scene": name, "n_ops": sc.flatten().n_ops}
        for rep in (1, 2):
            out[f"default_{rep}"], a = run(sc, w, h, None)
            out[f"exact_out_of_line_{rep}"], b = run(sc, w, h, "0")
        out["frames_identical"] = bool(torch.equal(a, b))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment"); only when RUN, not when a test imports the scene builders
    main()
