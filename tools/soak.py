"""One-off soak: many random scenes, GPU (spec + interp) vs the CPU oracle.  python tools/soak.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fuzz as F
from test_gpu_parity import check_against_oracle, gpu_render
from loltracer_amd import gpu, scene as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
rs = {m: gpu.Renderer(0, specialize=m) for m in (1, 4)}
bad = 0
for i in range(n):
    text = F.rand_scene(rng)
    sc = S.Scene.parse_string(text)
    w, h = int(rng.integers(17, 90)), int(rng.integers(9, 60))
    for m, r in rs.items():
        try:
            g = gpu_render(torch, r, sc, w, h)
            check_against_oracle(g, sc, w, h)
        except AssertionError as e:
            bad += 1
            print(f"FAIL scene {i} mode {m} {w}x{h}: {str(e)[:200]}\n{text}\n", flush=True)
    if i % 25 == 0:
        print("done", i, "bad", bad, flush=True)
print("total", n, "bad", bad)
sys.exit(1 if bad else 0)
