"""The device's scene SDF against the reference's own composition, with no march in between.

tests/golden/ref_sdf_points.json holds sdf(p) — distance and winning object id — for ~1900 points per example scene,
composed from the reference's scene.c object tree and its compiled v3sub / sdSphere / sdRoundBox / sminf
(tests/golden/make_golden.py).  lol_gpu_sdf_batch evaluates the same points through the SDF code the frames run:
the hipRTC-specialised module and the macro-op interpreter, each with and without the proven fast paths.
Rows a4 - a8 of SURVEY.md §8 (sdf, get_obj_dist, primitives, sminf, vector ops), bit for bit, NaN / inf points included.
"""
import json
import os
import struct

import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import gpu, scene as S

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def h2u(h):
    return int(h, 16)


@pytest.mark.parametrize("mode", [1, 3, 0, 4], ids=["spec", "spec-plain", "interp-plain", "interp"])
@pytest.mark.parametrize("name", ["scene", "scene2", "scene3", "scene4"])
def test_device_sdf_equals_the_reference_composition(scenes, name, mode):
    import torch
    pts = json.load(open(os.path.join(HERE, "golden", "ref_sdf_points.json")))["points"][name]
    xyz = np.array([[h2u(v) for v in p[0]] for p in pts], dtype=np.uint32).view(np.float32)
    want_d = np.array([h2u(p[1]) for p in pts], dtype=np.uint32)
    want_id = np.array([p[2] for p in pts], dtype=np.uint32)
    r = gpu.Renderer(0, specialize=mode)
    r.prepare(scenes[name])
    assert r.kernel_name() == ("lol_render_spec" if mode in (1, 3) else "render_interp")
    d_pts = torch.from_numpy(xyz.copy()).cuda()
    d_dist = torch.zeros(len(pts), dtype=torch.float32, device="cuda")
    d_id = torch.full((len(pts),), 77, dtype=torch.int32, device="cuda")
    r.sdf_batch(d_pts.data_ptr(), d_dist.data_ptr(), d_id.data_ptr(), len(pts), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got_d = d_dist.cpu().numpy().view(np.uint32)
    got_id = d_id.cpu().numpy().view(np.uint32)
    nan_w = np.isnan(want_d.view(np.float32))
    same = (got_d == want_d) | (nan_w & np.isnan(got_d.view(np.float32)))
    bad = np.flatnonzero(~same)
    assert bad.size == 0, (name, mode, [(xyz[i].tolist(), hex(got_d[i]), hex(want_d[i])) for i in bad[:5]])
    assert np.array_equal(got_id, want_id), (name, mode, np.flatnonzero(got_id != want_id)[:5])
    r.close()


@pytest.mark.parametrize("mode", [1, 3, 0, 4], ids=["spec", "spec-plain", "interp-plain", "interp"])
@pytest.mark.parametrize("name", ["scene", "scene2", "scene3", "scene4"])
def test_device_frames_equal_the_reference_composition(scenes, name, mode):
    """tests/golden/ref_frames.npz — 64x36 frames composed pixel by pixel from the reference's compiled vec.h / float.h /
    sdf.h following naive_renderer.c:48-236 — against the DEVICE, with no oracle in between: packed pixel, post-gamma
    float colour, hit distance, hit id, march steps and (where no exact skip removes them) shadow steps, bit for bit."""
    import torch
    from test_gpu_parity import gpu_render, HOST_LIBM_IS_FMA_VARIANT
    g = np.load(os.path.join(HERE, "golden", "ref_frames.npz"))
    r = gpu.Renderer(0, specialize=mode)
    w, h = 64, 36
    # production configuration first (exact skips on: fewer shadow steps, the same pixels) ...
    d = gpu_render(torch, r, scenes[name], w, h)
    assert d["miss_skip"] & 4, "the settled-shadow exit should be active on the example scenes"
    assert np.array_equal(d["id"], g[f"{name}_hit_id"])
    if HOST_LIBM_IS_FMA_VARIANT:
        assert np.array_equal(d["rgb"].view(np.uint32), g[f"{name}_rgb"].view(np.uint32))
        assert np.array_equal(d["xrgb"][:, :w], g[f"{name}_xrgb"])
    # ... then only the settled-shadow exit: every shadow ray is marched, each up to the first step that left its running
    # factor <= 0 — the count the fixture records beside the reference's own loop (make_golden.py: in_shadow)
    r.set_exact_skips(4)
    d = gpu_render(torch, r, scenes[name], w, h)
    assert d["miss_skip"] == 4
    assert np.array_equal(d["steps"] >> 16, g[f"{name}_shadow_settled_steps"].sum(axis=-1))
    assert np.array_equal(d["id"], g[f"{name}_hit_id"])
    if HOST_LIBM_IS_FMA_VARIANT:
        assert np.array_equal(d["xrgb"][:, :w], g[f"{name}_xrgb"])
    r.set_miss_skip(False)                     # ... and finally march every shadow ray to the reference's own exit
    d = gpu_render(torch, r, scenes[name], w, h)
    assert np.array_equal(d["id"], g[f"{name}_hit_id"])
    assert np.array_equal(d["dist"].view(np.uint32), g[f"{name}_hit_dist"].view(np.uint32))
    assert np.array_equal(d["steps"] & 0xFFFF, g[f"{name}_march_steps"])
    assert np.array_equal(d["steps"] >> 16, g[f"{name}_shadow_steps"].sum(axis=-1))
    if HOST_LIBM_IS_FMA_VARIANT:               # the fixture's powf is the build container's libm (FMA variant), like the kernel's
        assert np.array_equal(d["rgb"].view(np.uint32), g[f"{name}_rgb"].view(np.uint32))
        assert np.array_equal(d["xrgb"][:, :w], g[f"{name}_xrgb"])
    else:
        assert np.abs(d["rgb"] - g[f"{name}_rgb"]).max() <= 1e-4
    r.close()


@pytest.mark.parametrize("mode", [1, 4], ids=["spec", "interp"])
def test_points_where_the_fast_root_has_no_proof(scenes, mode):
    """scene4's spheres carry no range tracker in the specialised kernel (sd_sphere_fast_nr): a squared length of 0,
    a denormal one, one below 2^-96 and an overflowing one must all come out as the oracle's value — through the NaN
    vote and the plain path where the fast root gives NaN, directly where the offset is far below half an ulp of r."""
    import ctypes as C
    import torch
    import oracle_lib as O
    sc = scenes["scene4"]
    centres = np.array([[0, 1, -6], [-1, 0.5, -3], [-3, 4.5, -3], [2, 2, -10], [6, 2, -10]], dtype=np.float32)
    pts = [c.copy() for c in centres]                                     # l2 == 0
    for c in centres:
        for eps in (1e-45, 1e-30, 1e-22, 1e-20, 1e-18, 1e-17, 1e-16, 3e-15, 1e-14, 1e-10):   # l2 = 0, denormal, [2^-126, 2^-96), just above
            for ax in range(3):
                q = c.copy().astype(np.float64)
                if c[ax] == 0:                                             # (an offset that small only survives next to 0)
                    q[ax] += eps
                    pts.append(q.astype(np.float32))
    pts += [np.array([1e20, 0, 0], dtype=np.float32), np.array([0, -3e38, 0], dtype=np.float32),
            np.array([np.inf, 1, 1], dtype=np.float32), np.array([np.nan, 0, 0], dtype=np.float32)]
    pts = np.array(pts, dtype=np.float32)
    pts = np.concatenate([pts, np.tile(centres[1], (64, 1))])            # a whole wave on one centre
    l = O.lib()
    want_d = np.zeros(len(pts), dtype=np.float32)
    want_id = np.zeros(len(pts), dtype=np.int32)
    for i, p in enumerate(pts):
        oid = C.c_uint32()
        want_d[i] = l.lol_oracle_sdf(sc.ptr, float(p[0]), float(p[1]), float(p[2]), C.byref(oid))
        want_id[i] = oid.value
    r = gpu.Renderer(0, specialize=mode)
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


@pytest.mark.gpu
def test_shadow_ray_that_never_moves():
    """A shadow ray whose first sample lies EXACTLY on a surface: s = 0, 50 s / t = 0 / 0 = NaN, t stays 0, the reference's factor is
    NaN for all 128 steps and comes out maxf(NaN, 0) = 0 (naive_renderer.c:80-89).  The specialised kernel's fast pipeline keeps the
    running minimum with v_min_f32, which drops a NaN operand — it names this case by `t == 0` afterwards (lol_kernel.h, soft_shadow).
    Camera on the plane y = 0 (every primary ray 'hits' at distance 0, so every shadow ray starts at the camera), light at the same
    height: direction.y = 0 exactly, and every sample of the shadow ray has p.y - 0 = 0.  The zero-incidence skip is switched off so
    that the lanes do march (the plane's normal is perpendicular to the light's direction); step counts and pixels = the oracle's."""
    import torch
    from test_gpu_parity import check_against_oracle, gpu_render
    text = ("materials { { shininess = 4, diffuse = (.6,.5,.4), specular = (.3,.3,.3), ambient = (.2,.1,.3) } } scene {"
            " camera { point = (0,0,0), direction = (0,-.3,-1), fov = 90 }, plane { y = 0 }, sphere { point = (2,1,-6), radius = 1 },"
            " point_light { point = (3,0,-2), diffuse_intensity = (2,2,2), specular_intensity = (1,1,1) },"
            " point_light { point = (-4,6,1), diffuse_intensity = (1,1,1), specular_intensity = (1,1,1) } }")
    sc = S.Scene.parse_string(text)
    for mode in (1, 3, 4, 0):
        r = gpu.Renderer(0, specialize=mode)
        r.set_exact_skips(4 | 1)                 # settled shadows on, the zero-incidence skip off
        g = gpu_render(torch, r, sc, 48, 20)
        assert (g["steps"] >> 16).max() >= 128, "no shadow ray marched its 128 steps: the case this test is about did not occur"
        check_against_oracle(g, sc, 48, 20)
        r.close()
