import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import bench
from loltracer_amd import gpu, scene as S
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
cfg = bench.WORKLOADS["c3"]
sc = S.Scene.parse_file(os.path.join(ROOT, "tests/golden/scenes/scene4.lol"))
r = gpu.Renderer(0); r.prepare(sc)
print(json.dumps(bench.host_surface_rates(r, sc, cfg, None)))
