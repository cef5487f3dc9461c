#!/usr/bin/env python3
"""Round 5: fields of separate objects (the culling plan's k-d tests around them) with the scene's SDF inlined into the three
loops against the one out-of-line function, 1080p — the counterpart of tools/large_scene_ab.py (chains: ONE object) for the
decision on LOL_SPEC_INLINE_MAX_OPS.  One JSON line per field: ops, Mpixels/s and seconds of render_prepare (disk cache off)."""
import json
import os
os.environ["LOL_GPU_CACHE_DIR"] = ""
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from loltracer_amd import gpu, scene as S  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402


def run(sc, w, h, inline_max, frames=5):
    os.environ["LOL_GPU_SPEC_INLINE_MAX"] = str(inline_max)
    r = gpu.Renderer(0)
    t0 = time.perf_counter()
    r.prepare(sc)
    prep = time.perf_counter() - t0
    side = torch.cuda.Stream()
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(6):                                # the view repeats: the library's tables settle
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
    r.close()
    return dict(mpixels_per_s=round(w * h / ms / 1e3, 1), prepare_s=round(prep, 2)), buf


def main():
    w, h = 1920, 1080
    for n in (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "120,250,420").split(",")):
        sc = S.Scene.parse_string(F.big_field_scene(n, 9, 2))
        out = {"scene": f"field of {n} objects", "n_ops": sc.flatten().n_ops}
        out["spec_inline"], a = run(sc, w, h, 1 << 30)
        out["spec_out_of_line"], b = run(sc, w, h, 0)
        out["frames_identical"] = bool(torch.equal(a, b))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    os.environ.setdefault("LOL_GPU_TUNING", "1")      # the library honours its A/B switches only beside this (include/lol_gpu.h, "The environment"); only when RUN, not when a test imports the scene builders
    main()
