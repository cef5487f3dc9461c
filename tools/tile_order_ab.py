#!/usr/bin/env python3
"""The order in which a launch hands out its tiles: row by row, column by column, longest first (lol_gpu_set_tile_order) —
Mpixels/s of back-to-back frames on ONE stream and frame equality, per workload and kernel.
    python tools/tile_order_ab.py [--workloads c3,c2,c4,orbit,band] [--kernels spec,interp]"""
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
    ap.add_argument("--workloads", default="c3,c2,c4,orbit,band")
    ap.add_argument("--kernels", default="spec,interp")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--orbit-stride", type=int, default=1, help="orbit: every n-th frame of the 256 (what one of n ranks renders)")
    a = ap.parse_args()
    for wl in a.workloads.split(","):
        band = wl == "band"                              # rank 0's share of a C4 frame over 8 ranks, one launch
        cfg = bench.WORKLOADS["c4" if band else wl]
        w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
        sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", cfg["scene"] + ".lol"))
        rows = gpu.Rows(16, 128, 0) if band else None
        n_rows = gpu.part_rows(h, rows)
        cams = [sc.frame_camera(w, h, bench.orbit_camera(i, 256)) for i in range(0, 256, a.orbit_stride)] if wl == "orbit" else [sc.frame_camera(w, h)]
        for kern in a.kernels.split(","):
            r = gpu.Renderer(0, specialize=1 if kern == "spec" else 4)
            r.prepare(sc)
            stream = torch.cuda.Stream()
            buf = torch.zeros((n_rows, w), dtype=torch.int32, device="cuda")
            out = {"workload": wl + (f" stride {a.orbit_stride}" if wl == "orbit" else ""), "kernel": r.kernel_name(), "pixels_per_frame": n_rows * w}
            ref = None
            for order in ("rows", "cols", "lpt"):
                r.set_tile_order(order)

                def run(k, first=0):
                    for i in range(first, first + k):
                        r.render_into(buf.data_ptr(), w, h, ms, rows=rows, stream=stream.cuda_stream, frame_camera=cams[i % len(cams)])
                    torch.cuda.synchronize()
                run(24)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    run(a.frames)
                    best = min(best, time.perf_counter() - t0)
                out[order] = round(a.frames * n_rows * w / best / 1e6, 1)
                buf.zero_()
                torch.cuda.synchronize()                     # (zero_ runs on torch's stream, the frame on ours)
                run(1, first=7)
                if ref is None:
                    ref = buf.clone()
                out[order + "_frame_equal"] = bool(torch.equal(buf, ref))
            print(json.dumps(out), flush=True)
            r.close()


if __name__ == "__main__":
    main()
