#!/usr/bin/env python3
"""How do the two kernels scale to scenes far beyond the old 1024-op capacity?  Fields of N objects (tests/test_gpu_fuzz.py:
big_field_scene), a small and a realistic frame, each configuration in a subprocess of its own under a time limit (a giant
straight-line kernel run by two waves can take minutes).   python tools/big_scene_probe.py [--objects 250,600,1200,2300]"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CHILD = r"""
import sys, time, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np, torch
from loltracer_amd import gpu, scene as S
src = open(%(root)r + "/tests/test_gpu_fuzz.py").read()
ns = {"np": np}
exec(src[src.index("def num(x)"):src.index("def rand_leaf")], ns)
exec(src[src.index("def big_field_scene"):src.index("@pytest.mark.parametrize(\"mode,name\"")], ns)
n, lights, mode, w, h = %(n)d, %(lights)d, %(mode)d, %(w)d, %(h)d
sc = S.Scene.parse_string(ns["big_field_scene"](n, 20, lights))
r = gpu.Renderer(0, specialize=mode)
t0 = time.perf_counter(); r.prepare(sc); t_prep = time.perf_counter() - t0
buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
r.render_into(buf.data_ptr(), w, h); r.sync()
t0 = time.perf_counter()
frames = %(frames)d
for _ in range(frames):
    r.render_into(buf.data_ptr(), w, h)
r.sync()
dt = (time.perf_counter() - t0) / frames
print(json.dumps(dict(objects=n, ops=r.program.n_ops, lights=lights, kernel=r.kernel_name(), size="%%dx%%d" %% (w, h), prepare_s=round(t_prep, 2),
                      ms_per_frame=round(dt * 1e3, 3), mpixels_per_s=round(w * h / dt / 1e6, 3))))
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--objects", default="250,600,1200,2300")
    ap.add_argument("--lights", type=int, default=4)
    ap.add_argument("--sizes", default="16x8,640x360")
    ap.add_argument("--limit", type=int, default=150)
    a = ap.parse_args()
    for n in (int(x) for x in a.objects.split(",")):
        for size in a.sizes.split(","):
            w, h = (int(x) for x in size.split("x"))
            for mode in (4, 1):
                code = CHILD % dict(root=ROOT, n=n, lights=a.lights, mode=mode, w=w, h=h, frames=2)
                t0 = time.time()
                try:
                    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=a.limit)
                    line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else json.dumps(dict(objects=n, mode=mode, size=size, error=p.stderr[-300:]))
                except subprocess.TimeoutExpired:
                    line = json.dumps(dict(objects=n, mode=mode, size=size, error=f"no result within {a.limit} s"))
                print(line, flush=True)


if __name__ == "__main__":
    main()
