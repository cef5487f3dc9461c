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

from loltracer_amd import gpu

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
    r.set_miss_skip(False)                     # march every shadow ray, so that the shadow step counts are comparable
    w, h = 64, 36
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
