#!/usr/bin/env python3
"""Do consecutive frames gain from being launched on more than one stream?  A frame is ONE launch of 129,600 one-wave blocks
whose tail leaves SIMDs half empty (6.2 of 8 waves resident on average, LABNOTES.md §8); on one stream the next frame's first
waves wait for the last one's to finish.  Frames of a stream of frames are independent (naive_renderer.c:216), so with two
destination buffers they may overlap.  Times N frames issued round-robin over 1, 2, 3 streams (same kernel, same frames).
    python tools/stream_overlap_ab.py [--workload c3|c2|c4|orbit|band]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from loltracer_amd import gpu, scene as S


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frames", type=int, default=120)
    a = ap.parse_args()
    band = a.workload == "band"                         # rank 0's share of a C4 frame over 8 ranks (540 rows of 7680), one launch
    cfg = bench.WORKLOADS["c4" if band else a.workload]
    w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", cfg["scene"] + ".lol"))
    r = gpu.Renderer(0)
    r.prepare(sc)
    rows = gpu.Rows(16, 128, 0) if band else None
    n_rows = gpu.part_rows(h, rows)
    cams = [sc.frame_camera(w, h, bench.orbit_camera(i, 256)) for i in range(256)] if a.workload == "orbit" else [sc.frame_camera(w, h)]
    out = {"workload": a.workload, "frames": a.frames, "pixels_per_frame": n_rows * w}
    for order in ("rows", "cols"):
        r.set_tile_order(order)
        for n_streams in (1, 2, 3):
            streams = [torch.cuda.Stream() for _ in range(n_streams)]
            bufs = [torch.zeros((n_rows, w), dtype=torch.int32, device="cuda") for _ in range(n_streams)]

            def run(k):
                for i in range(k):
                    r.render_into(bufs[i % n_streams].data_ptr(), w, h, ms, rows=rows, stream=streams[i % n_streams].cuda_stream,
                                  frame_camera=cams[i % len(cams)])
                torch.cuda.synchronize()
            run(40)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                run(a.frames)
                best = min(best, time.perf_counter() - t0)
            out[f"{order}_{n_streams}_streams_mpixels_per_s"] = round(a.frames * n_rows * w / best / 1e6, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
