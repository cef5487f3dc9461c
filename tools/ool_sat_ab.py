#!/usr/bin/env python3
"""Round 5: the out-of-line form of the scene's kernel (scenes above 1024 ops) with and without the saturated smooth-min
shortcut inside the function (LOL_GPU_SMIN_SAT=2; round 2 measured it as a loss there: 504-op chain 100 -> 42 Mpixels/s).
Re-measured on the round's kernels: a 2048-op chain and fields of 600 / 1200 objects, 960x540."""
import json
import os
os.environ["LOL_GPU_CACHE_DIR"] = ""
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from loltracer_amd import gpu, scene as S  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
import large_scene_ab as L  # noqa: E402


def run(sc, w, h, sat):
    if sat is None:
        os.environ.pop("LOL_GPU_SMIN_SAT", None)
    else:
        os.environ["LOL_GPU_SMIN_SAT"] = sat
    r = gpu.Renderer(0)
    t0 = time.perf_counter()
    r.prepare(sc)
    prep = time.perf_counter() - t0
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
    return dict(mpixels_per_s=round(4 * w * h / e0.elapsed_time(e1) / 1e3, 2), prepare_s=round(prep, 2)), buf


def main():
    w, h = 960, 540
    for name, sc in (("chain of 1022 smooth unions", L.chain_scene(1022)),
                     ("field of 600 objects", S.Scene.parse_string(F.big_field_scene(600, 9, 2))),
                     ("field of 1200 objects", S.Scene.parse_string(F.big_field_scene(1200, 9, 2)))):
        out = {"scene": name, "n_ops": sc.flatten().n_ops}
        out["default"], a = run(sc, w, h, None)
        out["smin_sat_in_the_function"], b = run(sc, w, h, "2")
        out["frames_identical"] = bool(torch.equal(a, b))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h); only when RUN, not when a test imports the scene builders
    main()
