"""Exact culling of top-level objects (lol_gpu.hip "exact culling", LABNOTES.md §3.6).

CPU part — the proof obligation behind the skip: for every object with a bound (C, R'), the object's distance
value(p) >= |p - C| - R' at every point.  Checked with the oracle's SDF on thousands of random points around random
single-object scenes (spheres, rounded boxes, smooth unions nested up to 5 deep, k from 0.05 to 8), in double precision
against the binary32 values the oracle (= the reference's arithmetic) produces.
GPU part — culling on == culling off, bit for bit (pixels, hit ids, distances, step counts), and the reference's tie
rule (the FIRST object of equal distance wins, naive_renderer.c:39) survives the re-ordering.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import gpu, scene as S


def num(x):
    return ("%.5f" % x).rstrip("0").rstrip(".") if "." in "%.5f" % x else "%.5f" % x


def fmt(v):
    return "(%s)" % ", ".join(num(x) for x in v)


def rand_bounded(rng, depth):
    if depth == 0 or rng.random() < 0.25:
        if rng.random() < 0.5:
            return "sphere { point = %s, radius = %s }" % (fmt(rng.normal(size=3) * 4), num(rng.uniform(0.1, 3)))
        return "box { point = %s, point2 = %s, radius = %s }" % (
            fmt(rng.normal(size=3) * 4), fmt(rng.uniform(0.05, 3, size=3)), num(rng.uniform(0, 1)))
    return "smooth_union { smoothness = %s, a = %s, b = %s }" % (
        num(rng.choice([0.05, 0.5, 1, 3, 8])), rand_bounded(rng, depth - 1), rand_bounded(rng, int(rng.integers(0, depth))))


MAT = "materials { { shininess = 1, diffuse = (0,0,0), specular = (0,0,0), ambient = (0,0,0) } }\n"


def bounds_of(prog, root):
    c, r = (C.c_float * 3)(), C.c_float()
    st = gpu.gpu_lib().lol_gpu_cull_bounds(C.byref(prog), root, c, C.byref(r))
    return st, np.array(list(c), dtype=np.float64), float(r.value)


def clusters_of(prog, root):
    out = ((C.c_float * 4) * 3)()
    n = gpu.gpu_lib().lol_gpu_cull_bounds_clusters(C.byref(prog), root, out)
    assert n in (0, 2)
    return [(np.array(list(out[j])[:3], dtype=np.float64), float(out[j][3])) for j in range(n)]


def test_the_two_sphere_bound_holds_too():
    """Round 3: unions of three or more primitives whose leaves fall into two clusters much smaller than the one enclosing
    sphere are tested with two spheres, and skipped where BOTH tests pass: value(p) >= min_j (|p - C_j| - R'_j) must hold for
    the oracle's SDF — points near the object, far away, and on shells just outside either sphere."""
    rng = np.random.default_rng(2026)
    l = O.lib()
    checked = with_clusters = 0
    for _ in range(260):
        obj = rand_bounded(rng, int(rng.integers(2, 7)))
        sc = S.Scene.parse_string(MAT + "scene { " + obj.replace("{", "{ material = #0,", 1) + " }")
        prog = sc.flatten()
        cl = clusters_of(prog, 0)
        if not cl:
            continue
        with_clusters += 1
        st, c1, r1 = bounds_of(prog, 0)
        assert st == 1 and max(r for _, r in cl) < r1                # each cluster is smaller than the single sphere
        unit = lambda d: d / np.linalg.norm(d, axis=1, keepdims=True)
        pts = [rng.normal(size=(50, 3)) * 6 + c1, rng.normal(size=(20, 3)) * 300 + c1]
        for cj, rj in cl:
            pts.append(cj + unit(rng.normal(size=(50, 3))) * rj * rng.uniform(0.9, 1.3, size=(50, 1)))
        for p in np.concatenate(pts).astype(np.float32):
            oid = C.c_uint32()
            v = l.lol_oracle_sdf(sc.ptr, float(p[0]), float(p[1]), float(p[2]), C.byref(oid))
            bound = min(float(np.linalg.norm(p.astype(np.float64) - cj)) - rj for cj, rj in cl)
            assert v >= bound, (obj, p, v, bound)
            checked += 1
    assert with_clusters > 40 and checked > 6000
    # scene4's blob: two spheres of 5.7 and 7.8 instead of one of 11.1
    prog = S.Scene.parse_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes", "scene4.lol")).flatten()
    cl = clusters_of(prog, 0)
    assert len(cl) == 2 and sorted(round(r, 1) for _, r in cl) == [5.7, 7.8] and clusters_of(prog, 1) == []


def test_the_bound_holds_on_random_objects():
    rng = np.random.default_rng(77)
    l = O.lib()
    checked = 0
    for _ in range(120):
        obj = rand_bounded(rng, int(rng.integers(0, 6)))
        sc = S.Scene.parse_string(MAT + "scene { " + obj.replace("{", "{ material = #0,", 1) + " }")
        prog = sc.flatten()
        st, c, rp = bounds_of(prog, 0)
        assert st == 1
        # points: near the object, far away, and on a shell just outside the bound (where the inequality is tightest)
        pts = np.concatenate([rng.normal(size=(60, 3)) * 6 + c, rng.normal(size=(30, 3)) * 300 + c,
                              c + (lambda d: d / np.linalg.norm(d, axis=1, keepdims=True))(rng.normal(size=(60, 3))) * rp * rng.uniform(0.9, 1.3, size=(60, 1))])
        for p in pts.astype(np.float32):
            oid = C.c_uint32()
            v = l.lol_oracle_sdf(sc.ptr, float(p[0]), float(p[1]), float(p[2]), C.byref(oid))
            d = float(np.linalg.norm(p.astype(np.float64) - c))
            assert v >= d - rp, (obj, p, v, d - rp)
            checked += 1
    assert checked > 15000


def test_what_has_no_bound():
    for obj in ("plane { y = -1, material = #0 }",
                "smooth_union { material = #0, smoothness = 1, a = sphere { radius = 1 }, b = plane { y = 0 } }",
                "smooth_union { material = #0, smoothness = 0, a = sphere { radius = 1 }, b = sphere { radius = 2 } }",
                "smooth_union { material = #0, smoothness = -2, a = sphere { radius = 1 }, b = sphere { radius = 2 } }",
                "box { material = #0, point2 = (-1, 1, 1), radius = 0.1 }",
                "sphere { material = #0, point = (10000000000000000000, 0, 0), radius = 1 }"):
        st, _, _ = bounds_of(S.Scene.parse_string(MAT + "scene { " + obj + " }").flatten(), 0)
        assert st == 0, obj
    st, c, r = bounds_of(S.Scene.parse_string(MAT + "scene { sphere { material = #0, point = (1,2,3), radius = -4 } }").flatten(), 0)
    assert st == 1 and r < 1e-3 and c.tolist() == [1, 2, 3]       # |q| - r with r < 0 is >= |q|


# ------------------------------------------------------------------ on the GPU

KERNELS = {1: "lol_render_spec", 4: "render_interp"}


def _render(torch, sc, w, h, cull, mode):
    from test_gpu_parity import gpu_render
    r = gpu.Renderer(0, specialize=mode)
    r.set_cull(cull)
    g = gpu_render(torch, r, sc, w, h)
    assert r.kernel_name() == KERNELS[mode]
    r.close()
    return g


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 4], ids=["spec", "interp"])
@pytest.mark.parametrize("name,w,h", [("scene4", 640, 360), ("scene", 480, 270), ("scene2", 320, 200), ("scene3", 320, 200)])
def test_culling_changes_nothing(scenes, name, w, h, mode):
    import torch
    a, b = _render(torch, scenes[name], w, h, True, mode), _render(torch, scenes[name], w, h, False, mode)
    for key in ("xrgb", "id", "steps"):
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32))
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))


@pytest.mark.gpu
def test_ties_go_to_the_first_object_after_reordering():
    """sphere (id 1, bounded → evaluated last) and plane (id 2, no bound → evaluated first): at (0,-2,0) both are exactly
    1 away, and the reference's strict '<' keeps the sphere.  Likewise two coincident spheres after a plane."""
    import torch
    text = MAT + ("scene { sphere { material = #0, radius = 1 }, plane { material = #0, y = -3 },"
                  " sphere { material = #0, radius = 1 }, sphere { material = #0, point = (8, 0, 0), radius = 2 } }")
    sc = S.Scene.parse_string(text)
    pts = np.array([[0, -2, 0], [0, -2.5, 0], [0, 2, 0], [8, 0, 0], [5, -2, 0], [4, -1.5, 0], [0, -1, 0], [3, 0, 0]], dtype=np.float32)
    l = O.lib()
    want = []
    for p in pts:
        oid = C.c_uint32()
        want.append((l.lol_oracle_sdf(sc.ptr, float(p[0]), float(p[1]), float(p[2]), C.byref(oid)), oid.value))
    assert want[0] == (1.0, 1) and want[4][1] in (2, 4)
    for cull, mode in ((True, 1), (False, 1), (True, 4), (False, 4)):
        r = gpu.Renderer(0, specialize=mode)
        r.set_cull(cull)
        r.prepare(sc)
        d_pts = torch.from_numpy(pts.copy()).cuda()
        d_dist = torch.zeros(len(pts), dtype=torch.float32, device="cuda")
        d_id = torch.zeros(len(pts), dtype=torch.int32, device="cuda")
        r.sdf_batch(d_pts.data_ptr(), d_dist.data_ptr(), d_id.data_ptr(), len(pts), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = list(zip(d_dist.cpu().tolist(), d_id.cpu().tolist()))
        assert got == want, (cull, mode, got, want)
        r.close()


def _field_scene(n, seed):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import flat_scene_ab
    return flat_scene_ab.field_scene(n, seed)


def test_many_objects_are_grouped_into_nested_clusters(tmp_path, monkeypatch):
    """A field of 64 objects: the generated SDF holds one test per k-d node (nested braces), and with
    LOL_GPU_CULL_CLUSTERS=0 only the one run of all bounded objects plus the heavy objects' own tests."""
    sc = _field_scene(64, 5)
    gpu.compile_offline(sc.flatten(), str(tmp_path / "kd"))
    kd = open(str(tmp_path / "kd.hip")).read()
    monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
    monkeypatch.setenv("LOL_GPU_CULL_CLUSTERS", "0")
    gpu.compile_offline(sc.flatten(), str(tmp_path / "flat"))
    flat = open(str(tmp_path / "flat.hip")).read()
    assert kd.count(")) != 0") > flat.count(")) != 0") > 1           # (every test ends in `(vote(...) | vote(...)) != 0`)
    assert kd.count("cool[0] = 3u") == flat.count("cool[0] = 3u") == 2      # only the outermost run cools down (in eval and in eval_dist)
    # every object is still evaluated exactly once
    import re
    for src in (kd, flat):
        assert len(re.findall(r"best_id = \d+u; }", src)) == 65 * src.count("void eval(")


@pytest.mark.gpu
@pytest.mark.parametrize("clusters", ["0", "3", "5"], ids=["one-run", "leaf3", "leaf5"])
def test_fields_of_many_objects_match_the_oracle(monkeypatch, clusters):
    """10 … 120 separate objects (spheres, rounded boxes, small smooth unions) over a plane: every pixel against the
    oracle, for each shape of the culling plan, specialised kernel and interpreter (which carries the same tests)."""
    import torch
    from test_gpu_parity import check_against_oracle, gpu_render
    monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
    monkeypatch.setenv("LOL_GPU_CULL_CLUSTERS", clusters)
    for n, seed, w, h in ((10, 1, 64, 40), (40, 2, 56, 32), (120, 3, 48, 28)):
        sc = _field_scene(n, seed)
        for mode in (1, 4):
            r = gpu.Renderer(0, specialize=mode)
            g = gpu_render(torch, r, sc, w, h)
            assert r.kernel_name() == KERNELS[mode]
            check_against_oracle(g, sc, w, h)
            r.close()


@pytest.mark.gpu
def test_clustered_sdf_matches_the_oracle_at_points():
    """lol_gpu_sdf_batch over points scattered through a 150-object field, culling on (clusters) vs the oracle's SDF:
    distance bits and object id at every point — ids are where a wrong tie or a wrong skip would show."""
    import torch
    sc = _field_scene(150, 9)
    rng = np.random.default_rng(4)
    pts = (rng.uniform(-1, 1, size=(4096, 3)) * [14, 3, 14] + [0, 1, -14]).astype(np.float32)
    l = O.lib()
    want_d = np.zeros(len(pts), dtype=np.float32)
    want_id = np.zeros(len(pts), dtype=np.int32)
    for i, p in enumerate(pts):
        oid = C.c_uint32()
        want_d[i] = l.lol_oracle_sdf(sc.ptr, float(p[0]), float(p[1]), float(p[2]), C.byref(oid))
        want_id[i] = oid.value
    for mode in (1, 4):
        r = gpu.Renderer(0, specialize=mode)
        r.prepare(sc)
        d_pts = torch.from_numpy(pts.copy()).cuda()
        d_dist = torch.zeros(len(pts), dtype=torch.float32, device="cuda")
        d_id = torch.zeros(len(pts), dtype=torch.int32, device="cuda")
        r.sdf_batch(d_pts.data_ptr(), d_dist.data_ptr(), d_id.data_ptr(), len(pts), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(d_dist.cpu().numpy().view(np.uint32), want_d.view(np.uint32)), mode
        assert np.array_equal(d_id.cpu().numpy(), want_id), mode
        r.close()


def _tree_scene(depth, smooth=0.5):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import tree_scene_ab
    return tree_scene_ab.tree_scene(depth, smooth)


def test_big_operands_of_a_smooth_union_get_a_saturation_test(tmp_path, monkeypatch):
    """A balanced union tree of 64 spheres: operands `a` of >= 32 primitives (LOL_GPU_SAT_CULL_MIN_PRIMS moves the
    threshold, 0 removes the tests) are evaluated behind a test of their own bounding sphere against b + ks — in the
    fast struct only."""
    sc = _tree_scene(6)

    def tests_in(src):
        fast = src[src.index("struct SpecSdfFast"):]
        return fast.count("+ 0.f;") // 2                    # (the body twice: eval and eval_dist)
    gpu.compile_offline(sc.flatten(), str(tmp_path / "d"), assume_fast=True)
    src = open(str(tmp_path / "d.hip")).read()
    assert tests_in(src) == 1                       # the root's `a` (32 spheres)
    exact = src[src.index("struct SpecSdfExact"):src.index("struct SpecSdfFast")]
    assert "+ 0.f;" not in exact and "sminf_fastdiv" not in exact
    monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
    monkeypatch.setenv("LOL_GPU_SAT_CULL_MIN_PRIMS", "16")
    gpu.compile_offline(sc.flatten(), str(tmp_path / "s"), assume_fast=True)
    assert tests_in(open(str(tmp_path / "s.hip")).read()) == 3      # + the `a` of both 32-sphere halves
    monkeypatch.setenv("LOL_GPU_SAT_CULL_MIN_PRIMS", "1")
    gpu.compile_offline(sc.flatten(), str(tmp_path / "a"), assume_fast=True)
    assert tests_in(open(str(tmp_path / "a.hip")).read()) == 63
    monkeypatch.setenv("LOL_GPU_SAT_CULL_MIN_PRIMS", "0")
    gpu.compile_offline(sc.flatten(), str(tmp_path / "n"), assume_fast=True)
    assert tests_in(open(str(tmp_path / "n.hip")).read()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("min_prims", ["1", "4", "32", "0"])
def test_saturation_culling_changes_nothing(monkeypatch, min_prims):
    """Union trees (32 / 128 spheres, smoothness 0.5 and 2), random nested scenes and scene4 with a saturation test on
    every operand that may have one / from 4 / from 32 primitives (default) / none: every pixel, distance, id and step count
    against the oracle; lol_gpu_sdf_batch at scattered points too."""
    import torch
    import test_gpu_fuzz as F
    from test_gpu_parity import check_against_oracle, gpu_render
    monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
    monkeypatch.setenv("LOL_GPU_SAT_CULL_MIN_PRIMS", min_prims)
    r = gpu.Renderer(0)
    for depth, smooth, w, h in ((5, 0.5, 64, 36), (7, 0.5, 40, 24), (5, 2.0, 48, 28)):
        sc = _tree_scene(depth, smooth)
        g = gpu_render(torch, r, sc, w, h)
        assert r.kernel_name() == "lol_render_spec"
        check_against_oracle(g, sc, w, h)
    rng = np.random.default_rng(77)
    for _ in range(10):
        sc = S.Scene.parse_string(F.rand_scene(rng))
        w, h = int(rng.integers(17, 60)), int(rng.integers(9, 36))
        check_against_oracle(gpu_render(torch, r, sc, w, h), sc, w, h)
    # the SDF alone, far from and inside the tree, incl. non-finite points
    sc = _tree_scene(6)
    pts = (rng.uniform(-1, 1, size=(2048, 3)) * [14, 6, 12] + [0, 1, -9]).astype(np.float32)
    pts[:4] = [[np.inf, 0, 0], [np.nan, 1, 1], [1e30, -1e30, 0], [0, 3e38, 0]]
    l = O.lib()
    want_d = np.zeros(len(pts), dtype=np.float32)
    want_id = np.zeros(len(pts), dtype=np.int32)
    for i, p in enumerate(pts):
        oid = C.c_uint32()
        want_d[i] = l.lol_oracle_sdf(sc.ptr, float(p[0]), float(p[1]), float(p[2]), C.byref(oid))
        want_id[i] = oid.value
    r.prepare(sc)
    d_pts = torch.from_numpy(pts.copy()).cuda()
    d_dist = torch.zeros(len(pts), dtype=torch.float32, device="cuda")
    d_id = torch.zeros(len(pts), dtype=torch.int32, device="cuda")
    r.sdf_batch(d_pts.data_ptr(), d_dist.data_ptr(), d_id.data_ptr(), len(pts), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_dist.cpu().numpy()
    both_nan = np.isnan(got) & np.isnan(want_d)
    assert np.array_equal(got.view(np.uint32)[~both_nan], want_d.view(np.uint32)[~both_nan])
    assert np.array_equal(d_id.cpu().numpy(), want_id)
    r.close()
