#!/usr/bin/env python3
"""Round 5: the two tiers of the scene compiler on mid-size scenes (257 ... 1024 ops): when each kernel is there and how fast it
renders, 1080p, disk cache off.  One JSON line per scene."""
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


def rate(r, buf, w, h, frames=4):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(5):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(frames):
            r.render_into(buf.data_ptr(), w, h, 256, stream=side.cuda_stream)
        e1.record()
    torch.cuda.synchronize()
    return round(frames * w * h / e0.elapsed_time(e1) / 1e3, 1)


def main():
    w, h = 1920, 1080
    scenes = [("chain of 140 smooth unions", L.chain_scene(140)), ("chain of 510 smooth unions", L.chain_scene(510)),
              ("field of 250 objects", S.Scene.parse_string(F.big_field_scene(250, 9, 2))),
              ("field of 420 objects", S.Scene.parse_string(F.big_field_scene(420, 9, 2)))]
    for name, sc in scenes:
        buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        out = {"scene": name, "n_ops": sc.flatten().n_ops}
        r = gpu.Renderer(0)
        t0 = time.perf_counter()
        r.prepare(sc, wait=False)
        out["prepare_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        out["interpreter_mpixels_per_s"] = rate(r, buf, w, h) if r.kernel_name() == "render_interp" else None
        while r.specialize_state()[0] == 1:
            time.sleep(0.01)
        r.render_into(buf.data_ptr(), w, h, 256); r.sync()          # the frame boundary: the first kernel takes over
        out["first_kernel_after_s"] = round(time.perf_counter() - t0, 2)
        out["first_kernel_state"] = r.specialize_state()[0]
        k1 = r.kernel_key()
        out["first_kernel_mpixels_per_s"] = rate(r, buf, w, h)
        first = buf.clone()
        r.specialize_wait()
        r.render_into(buf.data_ptr(), w, h, 256); r.sync()
        out["last_kernel_after_s"] = round(time.perf_counter() - t0, 2)
        out["two_kernels"] = r.kernel_key() != k1
        out["last_kernel_mpixels_per_s"] = rate(r, buf, w, h)
        out["frames_identical"] = bool(torch.equal(first, buf))
        r.close()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
