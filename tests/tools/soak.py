"""Soak: many random scenes, GPU (spec + interp) vs the CPU oracle.  python tests/tools/soak.py [n] [seed] [stress] [onek] [still]
still: every scene is rendered five times under one camera — the first frames in a fixed order and through rectangles, the
later ones with their pixels dealt to waves by cost and the waves handed out longest first (lol_gpu.hip) — and the LAST frame is
what is held against the oracle (pixels, colours, ids, distances, step counts).
flight (round 5): every scene is rendered seven times with TWO FRAMES IN FLIGHT on the library's own streams
(lol_gpu_set_frames_in_flight) — scene camera, scene camera, another camera, then the scene camera four more times, so that
new views (fixed order) and repeated views (one set of scheduling tables per stream) overlap — and the last frame of EACH stream is
held against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_fuzz as F
from test_gpu_parity import check_against_oracle, gpu_render
from loltracer_amd import gpu, scene as S

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
modes = sys.argv[3:]                                    # any of: stress onek still (e.g. "still stress": the repeated view of the stress scenes)
stress = "stress" in modes
onek = "onek" in modes                                  # every smooth union of a scene shares one k: the interpreter folds its pops (MOPB_POST)
still = "still" in modes
flight = "flight" in modes
rng = np.random.default_rng(seed)


def flight_render(r, sc, w, h):
    """seven frames, two in flight; returns the last frame of each of the two streams, as gpu_render() does for one frame"""
    dev = torch.device("cuda:0")
    sets = []
    for _ in range(2):
        frame = torch.full((h, w), 0x55AA55, dtype=torch.int32, device=dev)
        rgb = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
        dist = torch.zeros((h, w), dtype=torch.float32, device=dev)
        hid = torch.zeros((h, w), dtype=torch.int32, device=dev)
        steps = torch.zeros((h, w), dtype=torch.int32, device=dev)
        sets.append((frame, rgb, dist, hid, steps, gpu.Debug(rgb.data_ptr(), dist.data_ptr(), hid.data_ptr(), steps.data_ptr())))
    torch.cuda.synchronize()                              # (the fills run on torch's stream, the frames on the library's own)
    r.prepare(sc)
    other = S.Camera()
    other.point = S.V3(sc.c.camera.point.x + 0.37, sc.c.camera.point.y + 0.11, sc.c.camera.point.z - 0.23)
    other.direction = sc.c.camera.direction
    other.fov = sc.c.camera.fov
    r.set_frames_in_flight(2)
    for k, cam in enumerate([None, None, other, None, None, None, None]):
        f = sets[k % 2]
        r.render_into(f[0].data_ptr(), w, h, 256, camera=cam, debug=f[5])
    r.sync()
    r.set_frames_in_flight(1)
    return [dict(xrgb=f[0].cpu().numpy().view(np.uint32), rgb=f[1].cpu().numpy(), dist=f[2].cpu().numpy(),
                 id=f[3].cpu().numpy().view(np.uint32), steps=f[4].cpu().numpy().view(np.uint32), miss_skip=r.miss_skip_active()) for f in sets]


def stress_scene(rng):
    """Scenes that lean on the culling bounds: many top-level objects of very different sizes and distances, smoothness
    from 0.01 to 60 (and 0 / negative: no bound), coordinates up to 10^4, negative radii, cameras inside objects."""
    scale = float(rng.choice([1, 1, 1, 30, 1000]))

    def leaf():
        c = rng.normal(size=3) * [6, 3, 6] * scale + [0, 1, -8 * scale]
        if rng.random() < 0.6:
            return "sphere { point = %s, radius = %s }" % (F.fmt(c), F.num(rng.choice([-1, 0, 0.01, 0.5, 2, 9]) * scale))
        return "box { point = %s, point2 = %s, radius = %s }" % (F.fmt(c), F.fmt(rng.uniform(0, 4, 3) * scale), F.num(rng.choice([0, 0.3, 2]) * scale))

    def tree(d):
        if d == 0 or rng.random() < 0.35:
            return leaf() if rng.random() < 0.93 else "plane { y = %s }" % F.num(rng.uniform(-5, 0) * scale)
        k = rng.choice([0, -1, 0.01, 0.3, 1, 4, 15, 60]) * scale
        return "smooth_union { smoothness = %s, a = %s, b = %s }" % (F.num(k), tree(d - 1), tree(int(rng.integers(0, d))))

    mats = "materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.1,.1,.1) }, { shininess = 9, diffuse = (.5,.4,.3), specular = (.3,.3,.3), ambient = (.1,.1,.1) } }"
    comps = ["camera { point = %s, direction = %s, fov = %s }" % (F.fmt(rng.normal(size=3) * [3, 2, 3] * scale), F.fmt(rng.normal(size=3) * 0.3 + [0, -0.2, -1]), F.num(rng.uniform(50, 150)))]
    for _ in range(int(rng.integers(0, 3))):
        comps.append("point_light { point = %s, diffuse_intensity = (2,2,2), specular_intensity = (1,1,1) }" % F.fmt(rng.normal(size=3) * 8 * scale + [0, 9 * scale, 0]))
    order = [0, 1] if rng.random() < 0.5 else [1, 0]
    objs = []
    crowd = rng.random() < 0.25                      # many shallow objects: the k-d clusters of the culling plan
    for _ in range(int(rng.integers(10, 60)) if crowd else int(rng.integers(1, 9))):
        o = tree(int(rng.integers(0, 2 if crowd else 4)))
        head, rest = o.split("{", 1)
        objs.append("%s{ material = #1,%s" % (head, rest))
    if rng.random() < 0.7:
        objs.insert(int(rng.integers(0, len(objs) + 1)), "plane { material = #1, y = %s }" % F.num(rng.uniform(-6, -1) * scale))
    return mats + "\nscene { " + ",\n".join(comps + objs) + " }\n"
rs = {m: gpu.Renderer(0, specialize=m) for m in (1, 4)}
bad = 0
for i in range(n):
    text = stress_scene(rng) if stress else F.rand_scene(rng)
    if onek:
        import re
        k = F.num(rng.choice([0.5, 1, 2, 3, 0.25, 7.5, 0.01, 20]))
        text = re.sub(r"smoothness = [^,]+,", "smoothness = %s," % k, text)
    sc = S.Scene.parse_string(text)
    w, h = int(rng.integers(17, 90)), int(rng.integers(9, 60))
    for m, r in rs.items():
        try:
            if flight:
                for g in flight_render(r, sc, w, h):
                    check_against_oracle(g, sc, w, h)
                continue
            g = gpu_render(torch, r, sc, w, h, repeat=5 if still else 1)
            check_against_oracle(g, sc, w, h)
            if still:
                assert r.tile_order()["order"] == "lpt", r.tile_order()
        except AssertionError as e:
            bad += 1
            print(f"FAIL scene {i} mode {m} {w}x{h}: {str(e)[:200]}\n{text}\n", flush=True)
    if i % 25 == 0:
        print("done", i, "bad", bad, flush=True)
print("total", n, "bad", bad)
sys.exit(1 if bad else 0)
