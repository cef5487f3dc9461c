#!/usr/bin/env python3
"""How do frames on two streams share the device — side by side from start to end (their tails coincide), or staggered (one
frame's first waves fill the other's tail)?  Renders N orbit frames alternately on two streams, every kernel between two
events, and prints the timeline (start / end of every kernel relative to the first start) plus the rate, for
  lockstep: the frames launched back to back, as a host loop does;
  stagger:  the FIRST frame as two launches of half the rows each, the second stream's first frame waiting for the first half —
            seeds an offset of half a frame between the streams.
    python tools/stagger_probe.py [--frames 64] [--order cols|rows] [--workload orbit|c3]"""
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
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--order", default="cols")
    ap.add_argument("--workload", default="orbit")
    ap.add_argument("--streams", type=int, default=2)
    a = ap.parse_args()
    cfg = bench.WORKLOADS["c3"]
    w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    r = gpu.Renderer(0)
    r.prepare(sc)
    r.set_tile_order(a.order)
    n_s = a.streams
    cams = [sc.frame_camera(w, h, bench.orbit_camera(2 * i, 256)) for i in range(a.frames)] if a.workload == "orbit" else [sc.frame_camera(w, h)] * a.frames
    if a.workload != "orbit":
        cams = [sc.frame_camera(w, h, bench.orbit_camera(i % 7, 256)) for i in range(a.frames)]      # a camera that moves a little, never repeats twice in a row
    streams = [torch.cuda.Stream() for _ in range(n_s)]
    bufs = [torch.zeros((h, w), dtype=torch.int32, device="cuda") for _ in range(n_s)]
    out = {"workload": a.workload, "order": a.order, "frames": a.frames, "streams": n_s}
    for mode in ("sequential", "lockstep", "stagger"):
        for rep in range(3):
            torch.cuda.synchronize()
            ev = []
            base = torch.cuda.Event(enable_timing=True)
            base.record(streams[0])
            t0 = time.perf_counter()
            for i, fc in enumerate(cams):
                k = 0 if mode == "sequential" else i % n_s
                s = streams[k]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if mode == "stagger" and i < n_s - 1:
                    # frame i as n_s launches of h / n_s rows; stream i + 1's first frame waits for the first of them
                    part = (h // n_s) // 4 * 4
                    e0.record(s)
                    for p in range(n_s):
                        rows = gpu.Rows(part if p < n_s - 1 else h - part * (n_s - 1), h, p * part)
                        r.render_into(bufs[k].data_ptr() + p * part * w * 4, w, h, ms, rows=rows, stream=s.cuda_stream, frame_camera=fc)
                        if p == 0:
                            first = torch.cuda.Event()
                            first.record(s)
                            streams[i + 1].wait_event(first)
                    e1.record(s)
                else:
                    e0.record(s)
                    r.render_into(bufs[k].data_ptr(), w, h, ms, stream=s.cuda_stream, frame_camera=fc)
                    e1.record(s)
                ev.append((e0, e1))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        out[mode + "_mpixels_per_s"] = round(a.frames * w * h / dt / 1e6, 1)
        tl = [(round(base.elapsed_time(e0), 3), round(base.elapsed_time(e1), 3)) for e0, e1 in ev]
        out[mode + "_timeline_ms_first_12"] = tl[:12]
        out[mode + "_timeline_ms_last_6"] = tl[-6:]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
