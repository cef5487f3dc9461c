"""Time one rank's share of C4 (7680x4320 over 8 ranks) for several band heights, on one GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from loltracer_amd import gpu, scene as S
sc = S.Scene.parse_file(os.path.join(ROOT, "tests/golden/scenes/scene4.lol"))
r = gpu.Renderer(0); r.prepare(sc)
w, h, n = 7680, 4320, 8
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
fc = sc.frame_camera(w, h)
for band in (2, 4, 6, 12, 20, 36, 540):
    if h % (band * n): continue
    ts = []
    for part in (0, 3, 7):
        rows = gpu.Rows.equal(band, n, part); nr = gpu.part_rows(h, rows)
        buf = torch.zeros((nr, w), dtype=torch.int32, device="cuda")
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); r.render_into(buf.data_ptr(), w, h, 256, rows=rows, stream=st.cuda_stream, frame_camera=fc); e1.record()
            torch.cuda.synchronize()
            if i >= 2: ts.append(e0.elapsed_time(e1))
    print(f"band {band:4d}: avg part {sum(ts)/len(ts):.3f} ms  max {max(ts):.3f}  -> 8-GPU kernel-bound rate {w*h/ (max(ts)*1e-3)/1e6:.0f} Mpix/s")
