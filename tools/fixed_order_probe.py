"""Diagnostic (round 4): C3 frames in the fixed tile orders and in the scheduled mode, per-frame HIP events against wall clock,
with short and long warm-ups — how bench.py's `scheduling` leg came to warm up for 16 frames (a device that has idled for a
moment renders its first frames 8 % slow).  python tools/fixed_order_probe.py (on the GPU box)"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from loltracer_amd import gpu, scene as S
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = bench.WORKLOADS["c3"]; w, h, ms = cfg["w"], cfg["h"], cfg["max_steps"]
sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", cfg["scene"] + ".lol"))
r = gpu.Renderer(0); r.prepare(sc)
fc = sc.frame_camera(w, h)
side = torch.cuda.Stream()
buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
def timed(n, warm, stream):
    for _ in range(warm): r.render_into(buf.data_ptr(), w, h, ms, stream=stream, frame_camera=fc)
    torch.cuda.synchronize()
    ev = []
    t0 = time.perf_counter()
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r.render_into(buf.data_ptr(), w, h, ms, stream=stream, frame_camera=fc); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    per = [a.elapsed_time(b) for a, b in ev]
    return round(wall, 4), [round(x, 4) for x in per[:8]], round(sum(per) / len(per), 4)
with torch.cuda.stream(side):
    s = side.cuda_stream
    for order, warm, n in (("lpt", 8, 20), ("cols", 2, 5), ("cols", 0, 40), ("rows", 2, 5), ("rows", 0, 40), ("lpt", 8, 20), ("cols", 24, 100)):
        r.set_tile_order(order)
        print(order, warm, n, timed(n, warm, s), flush=True)
    # no events: wall only
    r.set_tile_order("cols")
    for _ in range(24): r.render_into(buf.data_ptr(), w, h, ms, stream=s, frame_camera=fc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): r.render_into(buf.data_ptr(), w, h, ms, stream=s, frame_camera=fc)
    torch.cuda.synchronize(); print("cols wall, no events", (time.perf_counter() - t0) * 10)
