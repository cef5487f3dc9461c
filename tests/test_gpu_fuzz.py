"""Randomised scenes: every node kind, nested smooth unions, several lights — GPU (all kernels) vs oracle.

The example scenes exercise one rounded box and one tree shape; this generates `.lol` text with random trees
(spheres, rounded boxes, planes, smooth unions nested up to 4 deep, both left- and right-leaning so that
SMIN and SMIN_R both occur), 0-3 lights, random materials and cameras (sometimes inside an object), renders
small frames and requires the same bit-level agreement as tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import gpu, scene as S
from test_gpu_parity import check_against_oracle, gpu_render

pytestmark = pytest.mark.gpu


def num(x):
    """`.lol` numbers are [-.0-9]+ (scene-lexer.l:12): no exponent syntax, so never print one."""
    return ("%.5f" % x).rstrip("0").rstrip(".") if "." in "%.5f" % x else "%.5f" % x


def fmt(v):
    return "(%s)" % ", ".join(num(x) for x in v)


def rand_leaf(rng):
    kind = rng.integers(3)
    if kind == 0:
        return "sphere { point = %s, radius = %s }" % (fmt(rng.normal(size=3) * [4, 2, 4] + [0, 1, -7]), num(rng.uniform(0.3, 3)))
    if kind == 1:
        return "box { point = %s, point2 = %s, radius = %s }" % (
            fmt(rng.normal(size=3) * [4, 2, 4] + [0, 1, -7]), fmt(rng.uniform(0.2, 2.5, size=3)), num(rng.uniform(0, 0.8)))
    return "plane { y = %s }" % num(rng.uniform(-3, 0))


def rand_tree(rng, depth):
    if depth == 0 or rng.random() < 0.3:
        return rand_leaf(rng)
    a, b = rand_tree(rng, depth - 1), rand_tree(rng, int(rng.integers(0, depth)))
    if rng.random() < 0.5:
        a, b = b, a
    return "smooth_union { smoothness = %s, a = %s, b = %s }" % (num(rng.choice([0.5, 1, 2, 3, 0.25, 7.5])), a, b)


def rand_scene(rng):
    n_mat = int(rng.integers(1, 5))
    mats = []
    for i in range(n_mat):
        z = i == 0 and rng.random() < 0.5
        mats.append("{ shininess = %s, diffuse = %s, specular = %s, ambient = %s }" % (
            num(rng.choice([0, 1, 2, 8, 30.5])), fmt(rng.uniform(0, 0 if z else 0.6, 3)), fmt(rng.uniform(0, 0 if z else 0.4, 3)),
            fmt(rng.uniform(0, 0.5, 3))))
    comps = ["ambient { color = %s }" % fmt(rng.uniform(0, 0.2, 3)),
             "camera { point = %s, direction = %s, fov = %s }" % (
                 fmt(rng.normal(size=3) * [2, 1, 2] + [0, 2, 2]), fmt(rng.normal(size=3) * 0.4 + [0, -0.3, -1]),
                 num(rng.uniform(40, 160)))]
    for _ in range(int(rng.integers(0, 4))):
        comps.append("point_light { point = %s, diffuse_intensity = %s, specular_intensity = %s }" % (
            fmt(rng.normal(size=3) * 5 + [0, 8, -3]), fmt(rng.uniform(0.5, 4, 3)), fmt(rng.uniform(0, 4, 3))))
    for _ in range(int(rng.integers(1, 5))):
        obj = rand_tree(rng, int(rng.integers(0, 5)))
        head, rest = obj.split("{", 1)
        comps.append("%s{ material = #%d,%s" % (head, rng.integers(n_mat), rest))
    return "materials { %s }\nscene { %s }\n" % (",\n".join(mats), ",\n".join(comps))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("mode", [1, 3, 0, 4], ids=["spec", "spec-plain", "interp-plain", "interp"])
def test_random_scenes(torch_cuda, mode):
    rng = np.random.default_rng(20261004)
    r = gpu.Renderer(0, specialize=mode)
    ops_seen = set()
    for i in range(24):
        text = rand_scene(rng)
        sc = S.Scene.parse_string(text)
        prog = sc.flatten()
        ops_seen |= {prog.ops[k].op for k in range(prog.n_ops)}
        w, h = int(rng.integers(17, 70)), int(rng.integers(9, 40))
        g = gpu_render(torch_cuda, r, sc, w, h)
        try:
            check_against_oracle(g, sc, w, h)
        except AssertionError as e:
            raise AssertionError(f"scene {i} ({w}x{h}) failed: {e}\n{text}") from e
    assert ops_seen == {S.OP_SPHERE, S.OP_RBOX, S.OP_PLANE, S.OP_SMIN, S.OP_SMIN_R, S.OP_TOP}
    r.close()


def test_degenerate_inputs(torch_cuda):
    """No lights, no objects, camera inside a sphere / on a plane, negative shininess (powf → inf → NaN → 1.0)."""
    cases = [
        "materials { { shininess = 2, diffuse = (1,1,1), specular = (1,1,1), ambient = (.5,.25,.125) } } scene { ambient { color = (1,1,1) } }",
        "materials { { shininess = 2, diffuse = (1,1,1), specular = (1,1,1), ambient = (.5,.5,.5) } } scene { ambient { color = (.3,.3,.3) }, sphere { radius = 5 }, point_light { point = (0,9,0), diffuse_intensity = (1,1,1), specular_intensity = (1,1,1) } }",
        "materials { { shininess = -1.5, diffuse = (.2,.2,.2), specular = (.3,.3,.3), ambient = (0,0,0) } } scene { camera { point = (0,1,0), direction = (0,-.2,-1), fov = 90 }, plane { y = 0 }, point_light { point = (2,5,-3), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) } }",
        "materials { { shininess = 4, diffuse = (.2,.2,.2), specular = (.3,.3,.3), ambient = (.1,.1,.1) } } scene { camera { point = (0,0,0), direction = (0,0,-1), fov = 120 }, plane { y = 0 }, box { point = (0,0,-4), point2 = (1,1,1), radius = 0 }, point_light { point = (0,0,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) } }",
    ]
    # a sphere so far away that |p - c|^2 overflows to +inf: sqrt(inf) = inf in the reference; the fast roots are not
    # proven there, so lol::Range must flag it and the wave re-shades through the plain path
    cases.append("materials { { shininess = 4, diffuse = (.2,.2,.2), specular = (.3,.3,.3), ambient = (.1,.1,.1) } } scene {"
                 " camera { point = (0,1,0), direction = (0,0,-1), fov = 90 }, sphere { point = (100000000000000000000, 0, 0), radius = 1 },"
                 " sphere { point = (0,1,-4), radius = 1 }, point_light { point = (3,5,0), diffuse_intensity = (2,2,2), specular_intensity = (1,1,1) } }")
    # the camera sits exactly on a sphere's centre (squared length 0: the fast root gives NaN there), next to a sphere
    # too small for the NaN flag (radius 2^-21: keeps the range tracker), inside a smooth union and on its own
    cases.append("materials { { shininess = 4, diffuse = (.2,.2,.2), specular = (.3,.3,.3), ambient = (.1,.1,.1) } } scene {"
                 " camera { point = (0,1,0), direction = (0,0,-1), fov = 90 }, sphere { point = (0,1,0), radius = 0.25 },"
                 " smooth_union { smoothness = 0.5, a = sphere { point = (0,1,0), radius = 0.125 }, b = sphere { point = (1,1,-3), radius = 1 } },"
                 " sphere { point = (0,1,0), radius = 0.000000476837158203125 }, plane { y = -1 },"
                 " point_light { point = (0,1,0), diffuse_intensity = (2,2,2), specular_intensity = (1,1,1) } }")
    for mode in (1, 4):
        r = gpu.Renderer(0, specialize=mode)
        for text in cases:
            sc = S.Scene.parse_string(text)
            g = gpu_render(torch_cuda, r, sc, 40, 24)
            check_against_oracle(g, sc, 40, 24)
        r.close()


def test_orbit_frames_match_the_oracle(torch_cuda, scenes):
    """BASELINE config 5's camera path (bench.orbit_camera): 12 of the 256 orbit frames, every pixel checked."""
    import bench
    sc = scenes["scene4"]
    r = gpu.Renderer(0)
    w, h = 256, 144
    for i in range(0, 256, 22):
        cam = bench.orbit_camera(i, 256)
        g = gpu_render(torch_cuda, r, sc, w, h, camera=cam)
        check_against_oracle(g, sc, w, h, camera=cam)
    r.close()


def test_orbit_frames_at_full_size_match_the_oracle_on_sampled_rows(torch_cuda, scenes):
    """BASELINE config 5 at ITS size (3840x2160; round-4 review: the full-size orbit was only ever compared HIP against HIP): three
    orbit cameras, five rows each against the oracle — pixels, float colours, ids, distances, step counts — exactly as
    test_full_size_sampled_rows_and_partition does for config 3.  And the same frames with two of them in flight
    (lol_gpu_set_frames_in_flight, what bench.py's orbit workload does) are the same frames."""
    import bench
    import torch
    sc = scenes["scene4"]
    w, h = 3840, 2160
    r = gpu.Renderer(0)
    frames = {}
    for i in (37, 128, 211):
        cam = bench.orbit_camera(i, 256)
        g = gpu_render(torch_cuda, r, sc, w, h, camera=cam)
        for y in (0, 540, 1079, 1620, 2159):
            sub = {k: (v[y:y + 1] if isinstance(v, np.ndarray) else v) for k, v in g.items()}
            check_against_oracle(sub, sc, w, h, y0=y, y1=y + 1, camera=cam)
        frames[i] = g["xrgb"]
    r.set_frames_in_flight(2)
    ring = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()                              # (the fills run on torch's stream, the frames on the library's own)
    ids = (37, 128, 211, 37)
    for k in range(0, len(ids), 2):
        for j in (0, 1):
            r.render_into(ring[j].data_ptr(), w, h, camera=bench.orbit_camera(ids[k + j], 256))
        r.sync()
        for j in (0, 1):
            assert np.array_equal(ring[j].cpu().numpy().view(np.uint32), frames[ids[k + j]]), (k, j)
    r.close()


def chain_scene(n_unions, seed=3):
    """One object: a right-leaning chain of n smooth unions of spheres (2n + 2 ops), over a plane, lit and in view."""
    rng = np.random.default_rng(seed)
    body = "sphere { point = (0,0,-5), radius = 1 }"
    for i in range(n_unions):
        body = "smooth_union { smoothness = 1, a = sphere { point = %s, radius = %s }, b = %s }" % (
            fmt(rng.normal(size=3) * [3, 2, 3] + [0, 0, -7]), num(rng.uniform(0.3, 1.2)), body)
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (0,0,0) },"
            " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
            "scene { camera { point = (0, 3, 6), direction = (0, -0.25, -1), fov = 100 },"
            " point_light { point = (0,9,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
            + body.replace("{", "{ material = #1,", 1) + ", plane { y = -4, material = #1 } }")
    return S.Scene.parse_string(text)


@pytest.mark.parametrize("n_unions,inline_max", [(69, None), (69, 0), (130, None), (130, 0)],
                         ids=["142ops-inline", "142ops-out-of-line", "264ops-default", "264ops-out-of-line"])
def test_large_scenes_are_specialised_too(torch_cuda, monkeypatch, n_unions, inline_max):
    """Round 1 left scenes above 128 ops to the interpreter.  Every program the library accepts is now specialised:
    the SDF inlined up to 1024 ops (256 before round 5), as one out-of-line function beyond (LOL_GPU_SPEC_INLINE_MAX) — same pixels."""
    if inline_max is not None:
        monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
        monkeypatch.setenv("LOL_GPU_SPEC_INLINE_MAX", str(inline_max))
    sc = chain_scene(n_unions)
    assert sc.flatten().n_ops == 2 * n_unions + 4
    r = gpu.Renderer(0)
    r.prepare(sc)
    assert r.kernel_name() == "lol_render_spec", r.specialize_log()
    g = gpu_render(torch_cuda, r, sc, 48, 32)
    check_against_oracle(g, sc, 48, 32)
    r.close()


def test_unions_of_unions_with_one_smoothness(torch_cuda, monkeypatch):
    """Random trees whose smooth unions all share one k: the interpreter then folds every "smooth min of two finished sub-unions"
    into the record that finished the second one (lol_gpu.hip build_mops, MOPB_POST) — chains of them, at every stack depth,
    with round boxes and planes in between.  Against the oracle, and against the same frames with the folding switched off."""
    rng = np.random.default_rng(2024)

    def tree(d, k):
        if d == 0 or rng.random() < 0.2:
            return rand_leaf(rng)
        a, b = tree(d - 1, k), tree(int(rng.integers(0, d)), k)
        if rng.random() < 0.5:
            a, b = b, a
        return "smooth_union { smoothness = %s, a = %s, b = %s }" % (num(k), a, b)
    mats = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.02,.02,.02) },"
            " { shininess = 8, diffuse = (.5,.4,.3), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n")
    for n in range(10):
        k = float(rng.choice([0.5, 1, 3, 0.25]))
        objs = []
        for _ in range(int(rng.integers(1, 4))):
            head, rest = tree(int(rng.integers(2, 6)), k).split("{", 1)
            objs.append("%s{ material = #1,%s" % (head, rest))
        text = mats + ("scene { camera { point = (0, 2, 3), direction = (0, -0.2, -1), fov = 110 },"
                       " point_light { point = (1,9,0), diffuse_intensity = (2,2,2), specular_intensity = (1,1,1) }, "
                       + ", ".join(objs) + " }")
        sc = S.Scene.parse_string(text)
        frames = []
        for fuse in ("1", "0"):
            monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
            monkeypatch.setenv("LOL_GPU_INTERP_FUSE_POPS", fuse)
            r = gpu.Renderer(0, specialize=4)
            g = gpu_render(torch_cuda, r, sc, 48, 24)
            check_against_oracle(g, sc, 48, 24)
            frames.append(g["xrgb"])
            r.close()
        assert np.array_equal(frames[0], frames[1]), "scene %d" % n


def test_deep_tree_needs_the_big_operand_stack(torch_cuda):
    """A perfectly balanced smooth-union tree of 512 spheres (1024 ops): operand stack depth 10 — the interpreter's
    largest instantiation and the out-of-line specialised SDF; both against the oracle."""
    rng = np.random.default_rng(11)

    def tree(d):
        if d == 0:
            return "sphere { point = %s, radius = %s }" % (fmt(rng.normal(size=3) * [4, 2, 3] + [0, 0, -9]), num(rng.uniform(0.2, 0.8)))
        return "smooth_union { smoothness = 0.5, a = %s, b = %s }" % (tree(d - 1), tree(d - 1))
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.02,.02,.02) },"
            " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
            "scene { camera { point = (0, 1, 4), direction = (0, -0.1, -1), fov = 100 },"
            " point_light { point = (0,9,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
            + tree(9).replace("{", "{ material = #1,", 1) + " }")
    sc = S.Scene.parse_string(text)
    prog = sc.flatten()
    assert prog.n_ops == 1024 and prog.max_stack == 10
    for mode, name in ((4, "render_interp"), (1, "lol_render_spec")):
        r = gpu.Renderer(0, specialize=mode)
        r.prepare(sc)
        assert r.kernel_name() == name, r.specialize_log()
        g = gpu_render(torch_cuda, r, sc, 32, 20)
        check_against_oracle(g, sc, 32, 20)
        r.close()


def big_field_scene(n_obj, n_mat, n_light, seed=5):
    """A field of n_obj separate objects (a tenth of them small unions) over a plane, n_mat materials, n_light lights."""
    rng = np.random.default_rng(seed)
    mats = ["{ shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.02,.02,.03) }"]
    for i in range(1, n_mat):
        mats.append("{ shininess = %s, diffuse = %s, specular = %s, ambient = %s }" % (
            num(rng.choice([1, 2, 8, 30.5])), fmt(rng.uniform(0.1, 0.6, 3)), fmt(rng.uniform(0, 0.4, 3)), fmt(rng.uniform(0, 0.3, 3))))
    comps = ["ambient { color = (.1,.1,.1) }", "camera { point = (0, 4, 6), direction = (0, -0.35, -1), fov = 110 }"]
    for i in range(n_light):
        comps.append("point_light { point = %s, diffuse_intensity = %s, specular_intensity = %s }" % (
            fmt(rng.normal(size=3) * [12, 1, 12] + [0, 10, -10]), fmt(rng.uniform(0.02, 0.08, 3)), fmt(rng.uniform(0, 0.05, 3))))
    comps.append("plane { material = #1, y = -1 }")
    for i in range(n_obj - 1):
        c = rng.uniform([-20, -0.5, -30], [20, 2, 2])
        m = int(rng.integers(1, n_mat)) if i % 3 else (n_mat - 1 - i % 7)          # the last materials of the table are used too
        if i % 10 == 0:
            comps.append("smooth_union { material = #%d, smoothness = 0.5, a = sphere { point = %s, radius = %s }, b = box { point = %s, point2 = (.3,.4,.3), radius = .1 } }"
                         % (m, fmt(c), num(rng.uniform(0.2, 0.6)), fmt(c + [0.4, 0.2, 0])))
        else:
            comps.append("sphere { material = #%d, point = %s, radius = %s }" % (m, fmt(c), num(rng.uniform(0.15, 0.7))))
    return "materials { %s }\nscene { %s }\n" % (",\n".join(mats), ",\n".join(comps))


@pytest.mark.parametrize("mode,name", [(4, "render_interp"), (0, "render_interp"), (1, "lol_render_spec")], ids=["interp", "interp-plain", "spec"])
def test_scene_beyond_the_old_capacity(torch_cuda, mode, name):
    """2300 objects / 5000+ ops, 300 materials, 100 lights — rounds 1-3 refused it (1024 ops, 256 materials, 64 lights).  The
    tables (100 x 9 + 300 x 10 + 2300 dwords) no longer fit a one-wave block's LDS share and are read from global memory
    (lol_kernel.h, TABLES_GLOBAL): every pixel, id, distance and step count equals the oracle's on both kernels."""
    sc = S.Scene.parse_string(big_field_scene(2300, 300, 100))
    prog = sc.flatten()
    assert prog.n_ops >= 5000 and prog.n_materials == 300 and prog.n_lights == 100 and prog.n_roots == 2300
    r = gpu.Renderer(0, specialize=mode)
    r.prepare(sc)
    assert r.kernel_name() == name, r.specialize_log()
    w, h = 16, 8
    # (the checker accounts the exact skips light by light for the first four lights only: first with every shadow ray marched
    # to the reference's own end — all step counts against the oracle — then the production configuration against that frame)
    r.set_exact_skips(0)
    g = gpu_render(torch_cuda, r, sc, w, h)
    check_against_oracle(g, sc, w, h)
    assert len(np.unique(g["id"])) > 3                            # the frame really shows several objects
    r.set_exact_skips(7)
    g2 = gpu_render(torch_cuda, r, sc, w, h)
    assert g2["miss_skip"] != 0
    for k in ("xrgb", "id"):
        assert np.array_equal(g[k], g2[k]), k
    assert np.array_equal(g["rgb"].view(np.uint32), g2["rgb"].view(np.uint32)) and np.array_equal(g["dist"].view(np.uint32), g2["dist"].view(np.uint32))
    assert np.array_equal(g["steps"] & 0xFFFF, g2["steps"] & 0xFFFF) and ((g2["steps"] >> 16) <= (g["steps"] >> 16)).all()
    assert ((g2["steps"] >> 16) < (g["steps"] >> 16)).any()
    # ... and the same view again and again: the scheduled frame (pixels dealt by cost, waves longest first) of the kernels that
    # read their tables from global memory — a frame wider than one region, neither dimension a multiple of it
    w2, h2 = 70, 20
    first = gpu_render(torch_cuda, r, sc, w2, h2)
    fifth = gpu_render(torch_cuda, r, sc, w2, h2, repeat=5)
    assert r.tile_order()["order"] == "lpt" and r.tile_order()["decisions"] >= 1
    for k in ("xrgb", "id", "steps"):
        assert np.array_equal(first[k], fifth[k]), k
    assert np.array_equal(first["rgb"].view(np.uint32), fifth["rgb"].view(np.uint32)) and np.array_equal(first["dist"].view(np.uint32), fifth["dist"].view(np.uint32))
    r.close()


def test_operand_stack_deeper_than_the_slot_fields(torch_cuda):
    """A balanced smooth-union tree of 4096 spheres in ONE object: operand stack 13 — one more than the interpreter's register
    stacks hold; the deep instantiation (slots in words of their own, lol_kernel.h MOP_DEEP_FROM) renders it like the oracle, and
    so does the specialised kernel, whose straight-line SDF needs no stack at all."""
    rng = np.random.default_rng(3)

    def tree(d):
        if d == 0:
            return "sphere { point = %s, radius = %s }" % (fmt(rng.normal(size=3) * [5, 2, 4] + [0, 0, -10]), num(rng.uniform(0.1, 0.5)))
        return "smooth_union { smoothness = 0.25, a = %s, b = %s }" % (tree(d - 1), tree(d - 1))
    text = ("materials { { shininess = 2, diffuse = (0,0,0), specular = (0,0,0), ambient = (.02,.02,.02) },"
            " { shininess = 8, diffuse = (.5,.5,.5), specular = (.2,.2,.2), ambient = (.1,.1,.1) } }\n"
            "scene { camera { point = (0, 1, 4), direction = (0, -0.1, -1), fov = 100 },"
            " point_light { point = (0,9,0), diffuse_intensity = (2,2,2), specular_intensity = (2,2,2) }, "
            + tree(12).replace("{", "{ material = #1,", 1) + " }")
    sc = S.Scene.parse_string(text)
    prog = sc.flatten()
    assert prog.n_ops == 8192 and prog.max_stack == 13
    for mode, name in ((4, "render_interp"), (0, "render_interp"), (1, "lol_render_spec")):
        r = gpu.Renderer(0, specialize=mode)
        r.set_specialize_max_ops(16384)                   # (above the default cap of 6144 ops: the scene compiler takes it on when the host says so)
        r.prepare(sc)
        assert r.kernel_name() == name, r.specialize_log()
        g = gpu_render(torch_cuda, r, sc, 16, 8)
        check_against_oracle(g, sc, 16, 8)
        r.close()
    r = gpu.Renderer(0)                                   # without: quietly on the interpreter, the reason in the log
    r.prepare(sc)
    assert r.kernel_name() == "render_interp" and "6144" in r.specialize_log()
    r.close()


def test_very_large_scenes_stay_on_the_interpreter(torch_cuda, monkeypatch):
    """Above LOL_GPU_SPEC_MAX_OPS the scene compiler does not take a program on (minutes of hipRTC): it renders on the
    interpreter, quietly, with the reason in the log."""
    monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
    monkeypatch.setenv("LOL_GPU_SPEC_MAX_OPS", "100")
    sc = S.Scene.parse_string(big_field_scene(120, 9, 1))
    r = gpu.Renderer(0)
    r.prepare(sc)
    assert r.kernel_name() == "render_interp" and "LOL_GPU_SPEC_MAX_OPS" in r.specialize_log()
    g = gpu_render(torch_cuda, r, sc, 32, 16)
    check_against_oracle(g, sc, 32, 16)
    r.close()


def test_both_tiers_of_mid_size_scenes_match_the_oracle(torch_cuda, monkeypatch):
    """Scenes of 257 ... 1024 ops get two kernels (lol_gpu.hip, start_specialise): the SDF as one out-of-line function first, the
    SDF inlined into the three loops behind it.  Chains, fields (the culling plan's nested tests in both forms) and a union tree:
    every pixel, id, distance and step count of a frame rendered on EACH tier equals the oracle's."""
    import time
    monkeypatch.setenv("LOL_GPU_CACHE_DIR", "")                      # both kernels are really compiled, one after the other
    scenes = [chain_scene(140, seed=int(time.time()) % 9973),        # (scenes no other test of this process has compiled)
              S.Scene.parse_string(big_field_scene(150, 7, 2, seed=int(time.time()) % 9967)),
              S.Scene.parse_string(big_field_scene(330, 9, 3, seed=int(time.time()) % 9949))]
    for sc in scenes:
        n_ops = sc.flatten().n_ops
        assert 256 < n_ops <= 1024, n_ops
        w, h = 48, 28
        r = gpu.Renderer(0)
        r.prepare(sc, wait=False)
        deadline = time.perf_counter() + 240
        while r.specialize_state()[0] == 1 and time.perf_counter() < deadline:
            time.sleep(0.01)
        dev = torch_cuda.device("cuda:0")
        bufs = [torch_cuda.zeros((h, w), dtype=dt, device=dev) for dt in (torch_cuda.int32, torch_cuda.float32, torch_cuda.int32, torch_cuda.int32)]
        rgb = torch_cuda.zeros((h, w, 3), dtype=torch_cuda.float32, device=dev)
        torch_cuda.cuda.synchronize()

        def frame():
            dbg = gpu.Debug(rgb.data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr())
            r.render_into(bufs[0].data_ptr(), w, h, debug=dbg)
            r.sync()
            return dict(xrgb=bufs[0].cpu().numpy().view(np.uint32), rgb=rgb.cpu().numpy(), dist=bufs[1].cpu().numpy(),
                        id=bufs[2].cpu().numpy().view(np.uint32), steps=bufs[3].cpu().numpy().view(np.uint32), miss_skip=r.miss_skip_active())
        g1 = frame()                                                 # the frame boundary at which the first kernel takes over
        assert r.kernel_name() == "lol_render_spec" and r.specialize_state()[0] in (5, 6), (n_ops, r.specialize_state(), r.specialize_log())
        key1 = r.kernel_key()
        g1 = frame()
        if r.kernel_key() == key1:                                   # (still the first tier: this frame ran on it)
            check_against_oracle(g1, sc, w, h)
        r.specialize_wait()
        g2 = frame()
        assert r.specialize_state()[0] == 2 and r.kernel_key() != key1 and "second tier" in r.specialize_log()
        check_against_oracle(g2, sc, w, h)
        r.close()


def test_a_failed_first_tier_does_not_cost_the_scene_its_kernel(torch_cuda, monkeypatch):
    """Round-5 advisor: when the FIRST run of a mid-size scene (the out-of-line form — the one the long-branch trip-wire and the
    dropped-options refusal of compile_spec are about) fails, the inlined form is compiled all the same while the interpreter
    renders; only when that fails too does the scene stay on the interpreter."""
    import time
    monkeypatch.setenv("LOL_GPU_CACHE_DIR", "")
    sc = chain_scene(140, seed=int(time.time()) % 9941)
    assert 256 < sc.flatten().n_ops <= 1024
    w, h = 48, 28
    r = gpu.Renderer(0)
    r.testing_fail_first_tier(1)
    r.prepare(sc, wait=False)
    r.specialize_wait()
    buf = torch_cuda.zeros((h, w), dtype=torch_cuda.int32, device="cuda")
    r.render_into(buf.data_ptr(), w, h)
    r.sync()
    log = r.specialize_log()
    assert r.kernel_name() == "lol_render_spec" and r.specialize_state()[0] == 2, (r.specialize_state(), log)
    assert "out-of-line form of the kernel was not to be had" in log and "second tier" in log, log
    want, _, _ = O.render(sc, w, h, threads=4)
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), want)
    r.close()


def test_a_host_may_decline_the_second_tier(torch_cuda, monkeypatch):
    """lol_gpu_set_specialize(ctx, 5): a mid-size scene keeps its first (out-of-line) kernel and nothing is compiled behind it —
    for a host that would rather exit or switch scenes promptly (the scene compiler cannot be interrupted; lol_gpu_destroy
    waits for it).  Same pixels."""
    import time
    monkeypatch.setenv("LOL_GPU_CACHE_DIR", "")
    sc = chain_scene(140, seed=int(time.time()) % 9929)
    assert 256 < sc.flatten().n_ops <= 1024
    w, h = 48, 28
    r = gpu.Renderer(0, specialize=5)
    r.prepare(sc)                                                    # (waits for the scene compiler: there is only one run to wait for)
    buf = torch_cuda.zeros((h, w), dtype=torch_cuda.int32, device="cuda")
    r.render_into(buf.data_ptr(), w, h)
    r.sync()
    assert r.kernel_name() == "lol_render_spec" and r.specialize_state()[0] == 2 and "second tier" not in r.specialize_log()
    want, _, _ = O.render(sc, w, h, threads=4)
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), want)
    t0 = time.perf_counter()
    r.close()
    assert time.perf_counter() - t0 < 1.0                            # nothing to wait for
