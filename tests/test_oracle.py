"""The CPU oracle against everything that pins it (no GPU needed).

1. tests/golden/ref_primitives.json — outputs of the REFERENCE's own float.h / vec.h / sdf.h
   (compiled in place as oracle/_ref) on 400+ inputs per function, incl. NaN / inf / ±0 / denormals:
   the oracle's restated primitives must agree bit for bit.
2. Known answers recorded by the survey from the unmodified naive_renderer.c:
   centre / corner pixels and the per-pixel work counters (sdf evaluations, march steps, shadow steps).
   (The survey's five whole-frame hashes are not used: they do not describe the shipped scene files.)
3. tests/golden/ref_sdf_points.json — the scene SDF composed from the reference's own object tree and primitives.
   tests/golden/ref_frames.npz, ref_pixels.json — whole pixels composed the same way: naive_renderer.c:48-236 followed
   statement by statement with the reference's compiled vec.h / float.h / sdf.h doing every operation.
4. tests/golden/oracle_frames.npz — regression frames of the oracle itself (pins it across toolchains).
"""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import scene as S

HERE = os.path.dirname(os.path.abspath(__file__))
PRIM = json.load(open(os.path.join(HERE, "golden", "ref_primitives.json")))["vectors"]


def h2f(h):
    return struct.unpack("<f", struct.pack("<I", int(h, 16)))[0]


def f2h(x):
    return "%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


def same(got, want_hex):
    w = h2f(want_hex)
    if w != w:
        return got != got
    return f2h(got) == want_hex


def a3(hs):
    return (C.c_float * 3)(*[h2f(h) for h in hs])


@pytest.mark.parametrize("fn", ["minf", "maxf"])
def test_minmax_match_reference(fn):
    f = getattr(O.lib(), "lol_oracle_" + fn)
    for a, b, want in PRIM[fn]:
        assert same(f(h2f(a), h2f(b)), want), (fn, a, b)


@pytest.mark.parametrize("fn", ["clamp", "sminf"])
def test_ternary_match_reference(fn):
    f = getattr(O.lib(), "lol_oracle_" + fn)
    for a, b, c, want in PRIM[fn]:
        assert same(f(h2f(a), h2f(b), h2f(c)), want), (fn, a, b, c)


def test_vector_primitives_match_reference():
    l = O.lib()
    out = (C.c_float * 3)()
    for p, q, want in PRIM["v3dot"]:
        assert same(l.lol_oracle_v3dot(a3(p), a3(q)), want)
    for p, want in PRIM["v3len"]:
        assert same(l.lol_oracle_v3len(a3(p)), want)
    for p, want in PRIM["v3normalize"]:
        l.lol_oracle_v3normalize(a3(p), out)
        assert all(same(o, w) for o, w in zip(out, want)), p
    for p, q, want in PRIM["v3cross"]:
        l.lol_oracle_v3cross(a3(p), a3(q), out)
        assert all(same(o, w) for o, w in zip(out, want))
    for p, lo, hi, want in PRIM["v3clamp"]:
        l.lol_oracle_v3clamp(a3(p), h2f(lo), h2f(hi), out)
        assert all(same(o, w) for o, w in zip(out, want))


def test_sdf_primitives_match_reference():
    l = O.lib()
    for p, r, want in PRIM["sd_sphere"]:
        assert same(l.lol_oracle_sd_sphere(a3(p), h2f(r)), want)
    for p, b, r, want in PRIM["sd_round_box"]:
        assert same(l.lol_oracle_sd_round_box(a3(p), a3(b), h2f(r)), want)


def test_sse_minmax_semantics_spelled_out():
    l = O.lib()
    nan = float("nan")
    assert l.lol_oracle_minf(1.0, nan) != l.lol_oracle_minf(1.0, nan)      # second operand on NaN
    assert l.lol_oracle_minf(nan, 1.0) == 1.0
    assert f2h(l.lol_oracle_minf(0.0, -0.0)) == "80000000"                 # second operand on equal
    assert f2h(l.lol_oracle_maxf(-0.0, 0.0)) == "00000000"
    assert l.lol_oracle_clamp(nan, 0.0, 1.0) == 0.0                         # clamp(NaN) → lo
    out = (C.c_float * 3)()
    l.lol_oracle_v3clamp((C.c_float * 3)(nan, -5.0, 5.0), 0.0, 1.0, out)
    assert list(out) == [1.0, 0.0, 1.0]                                     # v3clamp(NaN) → hi


# ---- survey known answers -----------------------------------------

def test_survey_known_pixels(scenes):
    x, _, _ = O.render(scenes["scene"], 256, 256, threads=4)
    assert x[128, 128] == 0x000018
    x4, _, _ = O.render(scenes["scene4"], 256, 256, threads=4)
    assert x4[128, 128] == 0xB6D8CA and x4[0, 0] == 0


@pytest.mark.parametrize("name,w,h,sdf_per_px", [("scene", 256, 256, 35.0), ("scene", 480, 270, 39.1),
                                                 ("scene4", 256, 256, 84.3)])
def test_survey_work_counters(scenes, name, w, h, sdf_per_px):
    _, _, c = O.render(scenes[name], w, h, threads=4, want_counters=True)
    assert round(c.sdf_evals / c.pixels, 1) == sdf_per_px
    assert c.sdf_evals == c.march_steps + c.shadow_steps + 4 * c.pixels
    nodes = {"scene": 4, "scene4": 10}[name]
    assert c.node_evals == nodes * c.sdf_evals


def test_survey_step_split_at_960x540(scenes):
    """SURVEY.md §3.3 / BASELINE.md §2: scene4 at 960x540 costs 21.9 primary march steps + 71.5 shadow steps + 4 normal
    taps = 97.4 sdf() calls per pixel, each walking 10 nodes; scene.lol at 480x270: 26.7 + 8.4 + 4 = 39.1 over 4 nodes."""
    _, _, c = O.render(scenes["scene4"], 960, 540, threads=8, want_counters=True)
    assert round(c.march_steps / c.pixels, 1) == 21.9
    assert abs(c.shadow_steps / c.pixels - 71.5) < 0.06          # the survey printed 71.5; this oracle gives 71.45
    assert round(c.sdf_evals / c.pixels, 1) == 97.4 and c.node_evals == 10 * c.sdf_evals
    _, _, c = O.render(scenes["scene"], 480, 270, threads=8, want_counters=True)
    assert round(c.march_steps / c.pixels, 1) == 26.7 and round(c.shadow_steps / c.pixels, 1) == 8.4


# ---- the scene SDF, composed from the reference's own pieces (tests/golden/ref_sdf_points.json) ---------

@pytest.mark.parametrize("name", ["scene", "scene2", "scene3", "scene4"])
def test_scene_sdf_matches_the_reference_composition(scenes, name):
    """Rows a4 / a5 of SURVEY.md §8: sdf() and get_obj_dist() (naive_renderer.c:11-44).  The fixture holds, for ~1900
    points per example scene (volume samples, points along view rays, NaN / inf / denormal coordinates), the distance
    and object id obtained by walking the reference's OWN struct object tree (built by its scene.c) with its OWN
    compiled v3sub / sdSphere / sdRoundBox / sminf — see tests/golden/make_golden.py.  The oracle's sdf (tree walk,
    smooth-union operand order, strict '<' with 1-based ids) must agree bit for bit, id included."""
    pts = json.load(open(os.path.join(HERE, "golden", "ref_sdf_points.json")))["points"][name]
    assert len(pts) > 1500
    l = O.lib()
    l.lol_oracle_sdf.restype = C.c_float
    l.lol_oracle_sdf.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_uint32)]
    ids_seen = set()
    for p, want, want_id in pts:
        oid = C.c_uint32(99)
        got = l.lol_oracle_sdf(scenes[name].ptr, h2f(p[0]), h2f(p[1]), h2f(p[2]), C.byref(oid))
        assert same(got, want), (name, p, f2h(got), want)
        assert oid.value == want_id, (name, p, oid.value, want_id)
        ids_seen.add(want_id)
    assert len(ids_seen) >= 2                       # more than one object wins somewhere


# ---- whole pixels and frames composed from the reference's compiled pieces ---------------------------

def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["scene", "scene2", "scene3", "scene4"])
def test_whole_frames_equal_the_reference_composition(scenes, name):
    """tests/golden/ref_frames.npz: every pixel of the 64x36 frame, produced by following naive_renderer.c:48-236
    statement by statement with every vector / min / max / clamp / smooth-min / distance operation executed by the
    reference's own compiled code (make_golden.py: RefPipeline).  The oracle must reproduce ALL of it bit for bit:
    packed pixel, gamma and linear colour, hit distance and id, march steps (rows a2, a3, a9 - a13 of SURVEY.md §8)."""
    g = np.load(os.path.join(HERE, "golden", "ref_frames.npz"))
    w, h = 64, 36
    x, rgb, steps = O.render_rows(scenes[name], w, h, 0, h, want_steps=True)
    assert np.array_equal(x, g[f"{name}_xrgb"])
    assert np.array_equal(_bits(rgb), _bits(g[f"{name}_rgb"]))
    assert np.array_equal(steps[..., 0], g[f"{name}_march_steps"])
    assert np.array_equal(steps[..., 2], g[f"{name}_hit_id"])
    assert np.array_equal(steps[..., 1], g[f"{name}_shadow_steps"].sum(axis=-1))
    # the checker's "settled" shadow steps (round 3) against the count made beside the reference's own loop, light by light
    nl = min(g[f"{name}_shadow_steps"].shape[-1], 4)
    assert np.array_equal(steps[..., 4:4 + nl], g[f"{name}_shadow_steps"][..., :nl])
    assert np.array_equal(steps[..., 8:8 + nl], g[f"{name}_shadow_settled_steps"][..., :nl])
    assert len(np.unique(x)) > 50                       # a real picture, not a constant


@pytest.mark.parametrize("name", ["scene", "scene2", "scene3", "scene4"])
def test_pixel_intermediates_equal_the_reference_composition(scenes, name):
    """The same composition, per pixel and per stage (tests/golden/ref_pixels.json, a 13 x 9 grid of the 64x36 frame and
    a few 256x256 pixels): camera ray, hit distance, normal, every light's shadow factor and step count, linear colour."""
    px = json.load(open(os.path.join(HERE, "golden", "ref_pixels.json")))["pixels"][name]
    n = 0
    for size, plist in px.items():
        w, h = (int(v) for v in size.split("x"))
        for q in plist:
            p = O.probe(scenes[name], w, h, q["x"], q["y"])
            where = (name, size, q["x"], q["y"])
            assert [f2h(v) for v in p.rd] == q["rd"], where
            assert f2h(p.hit_dist) == q["hit_dist"] and p.hit_id == q["hit_id"] and p.march_steps == q["march_steps"], where
            nl = len(q["shadow"])
            assert all(same(p.shadow[i], q["shadow"][i]) for i in range(nl)), where
            assert [p.shadow_steps[i] for i in range(nl)] == q["shadow_steps"], where
            assert all(same(a, b) for a, b in zip(p.normal, q["normal"])), where
            assert all(same(a, b) for a, b in zip(p.rgb_linear, q["rgb_linear"])), where
            assert all(same(a, b) for a, b in zip(p.rgb, q["rgb"])), where
            assert p.xrgb == q["xrgb"], where
            n += 1
    assert n >= 117


# ---- regression frames of the oracle ---------------------------------------------------------------

def test_oracle_regression_frames(scenes):
    gold = np.load(os.path.join(HERE, "golden", "oracle_frames.npz"))
    for key in gold.files:
        if not key.endswith("_xrgb"):
            continue
        name, size = key.split("_")[0], key.split("_")[1]
        w, h = (int(v) for v in size.split("x"))
        x, rgb, _ = O.render(scenes[name], w, h, threads=4, want_rgb=True)
        assert np.array_equal(x, gold[key]), key
        rk = key.replace("_xrgb", "_rgb")
        if rk in gold.files:
            assert np.array_equal(rgb.view(np.uint32), gold[rk].view(np.uint32)), rk


def test_threads_and_row_ranges_agree(scenes):
    sc = scenes["scene4"]
    a, _, _ = O.render(sc, 96, 54, threads=1)
    b, _, _ = O.render(sc, 96, 54, threads=7)
    c, _, _ = O.render_rows(sc, 96, 54, 10, 20)
    assert np.array_equal(a, b) and np.array_equal(a[10:20], c[10:20]) and not c[:10].any()


def test_max_steps_zero_and_empty_scene():
    sc = S.Scene.parse_string("materials { { shininess = 2, diffuse = (1,1,1), specular = (1,1,1), ambient = (1,1,1) } }"
                              " scene { ambient { color = (0.5, 0.25, 0.125) } }")
    x, rgb, c = O.render(sc, 8, 4, want_rgb=True, want_counters=True)
    # no objects: sdf = +inf, miss material #0, ambient * mat.ambient, gamma
    want = np.power(np.array([0.5, 0.25, 0.125], dtype=np.float32), np.float32(1 / 2.2))
    assert np.allclose(rgb[0, 0], want, atol=1e-6) and c.miss_pixels == 32


def test_a_settled_shadow_factor_stays_settled(scenes):
    """softshadow (naive_renderer.c:73-90) returns maxf(res, 0); once res <= 0 no later step brings it back above 0 (min
    only lowers it, and a finite scene produces no NaN).  The renderer ends a shadow march there (FLAG_SHADOW_SETTLED);
    the oracle counts the rays that would contradict it — none, on every example scene, several sizes, moved cameras."""
    import bench
    for name, sc in scenes.items():
        for (w, h) in ((64, 36), (160, 90)):
            _, _, ctr = O.render(sc, w, h, threads=4, want_counters=True)
            assert ctr.settle_violations == 0, (name, w, h)
    sc = scenes["scene4"]
    for i in range(0, 256, 32):
        _, _, ctr = O.render(sc, 96, 54, threads=4, camera=bench.orbit_camera(i, 256), want_counters=True)
        assert ctr.settle_violations == 0, i
    # the settled step counts are a prefix of the full ones, light by light
    _, _, steps = O.render_rows(sc, 96, 54, 0, 54, want_steps=True)
    assert (steps[..., 8:12] <= steps[..., 4:8]).all() and (steps[..., 8:12] < steps[..., 4:8]).any()
