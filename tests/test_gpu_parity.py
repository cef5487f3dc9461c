"""GPU parity: the HIP renderer (through the C ABI) against the CPU oracle on the same scenes.

Bar (BASELINE.json north_star): every pixel's float colour within 1e-4 of naive_renderer.c.
Stronger checks made here: the control flow is bit-faithful — hit distance, hit id, march step
count and shadow step count of every pixel are EQUAL to the oracle's — and the packed XRGB8888
differs by at most 1 LSB in any channel (powf is the only non-bit-exact operation).
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import gpu, scene as S

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4      # north_star tolerance on post-gamma float colour


def _host_has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


# glibc picks its FMA build of powf on x86-64 CPUs with FMA; that is the variant the device code restates
HOST_LIBM_IS_FMA_VARIANT = _host_has_fma()


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.fixture(scope="module", params=[1, 3, 0, 4], ids=["spec", "spec-plain", "interp-plain", "interp"])
def renderer(torch_cuda, request):
    """All kernels: the hipRTC scene-specialised one and the ahead-of-time interpreter (macro-op list fetched with scalar loads), each
    with and without the proven fast paths."""
    r = gpu.Renderer(0, specialize=request.param)
    r.want_kernel = "lol_render_spec" if request.param in (1, 3) else "render_interp"
    yield r
    r.close()


def gpu_render(torch, r, sc, w, h, max_steps=256, rows=None, camera=None, pitch_px=None, repeat=1):
    n_rows = gpu.part_rows(h, rows)
    pitch_px = pitch_px or w
    dev = torch.device("cuda:0")
    frame = torch.full((n_rows, pitch_px), 0x55AA55, dtype=torch.int32, device=dev)
    rgb = torch.zeros((n_rows, w, 3), dtype=torch.float32, device=dev)
    dist = torch.zeros((n_rows, w), dtype=torch.float32, device=dev)
    hid = torch.zeros((n_rows, w), dtype=torch.int32, device=dev)
    steps = torch.zeros((n_rows, w), dtype=torch.int32, device=dev)
    dbg = gpu.Debug(rgb.data_ptr(), dist.data_ptr(), hid.data_ptr(), steps.data_ptr())
    r.prepare(sc)
    assert r.kernel_name() == getattr(r, "want_kernel", r.kernel_name()), r.specialize_log()
    for i in range(repeat):            # repeat > 1: the same view again and again — the later frames go through the library's tables
        if i:
            frame.fill_(0x55AA55); rgb.zero_(); dist.zero_(); hid.zero_(); steps.zero_()
        r.render_into(frame.data_ptr(), w, h, max_steps, camera=camera, rows=rows, pitch_bytes=pitch_px * 4,
                      debug=dbg, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return dict(xrgb=frame.cpu().numpy().view(np.uint32), rgb=rgb.cpu().numpy(), dist=dist.cpu().numpy(),
                id=hid.cpu().numpy().view(np.uint32), steps=steps.cpu().numpy().view(np.uint32),
                miss_skip=r.miss_skip_active())


def channels(x):
    return ((x[..., None] >> np.array([16, 8, 0], dtype=np.uint32)) & 0xFF).astype(np.int32)


def check_against_oracle(g, sc, w, h, max_steps=256, y0=0, y1=None, camera=None):
    y1 = h if y1 is None else y1
    ox, orgb, osteps = O.render_rows(sc, w, h, y0, y1, max_steps, camera=camera, want_steps=True)
    ox, orgb, osteps = ox[y0:y1], orgb[y0:y1], osteps[y0:y1]
    # what the kernel's early shadow exit (skip & 4) rests on, checked by the reference arithmetic itself on these very rays:
    # a running shadow factor that was <= 0 once never comes out != 0
    assert O.last_counters.settle_violations == 0
    gx = g["xrgb"][:, :w]
    assert np.array_equal(g["steps"] & 0xFFFF, osteps[..., 0]), "march step counts differ"
    gsh, osh = g["steps"] >> 16, osteps[..., 1]
    skip = g.get("miss_skip", 0)
    want = osh.astype(np.int64)
    for li in range(4):
        full = osteps[..., 4 + li].astype(np.int64)
        # a shadow march ends once its factor can only be 0 (skip & 4: res <= 0 — the oracle counts those steps too) ...
        marched = osteps[..., 8 + li].astype(np.int64) if skip & 4 else full
        # ... and lanes whose diffuse incidence for a light is exactly 0 do not march that light's shadow ray at all
        if skip & 2:
            marched = np.where((osteps[..., 3] >> li) & 1, 0, marched)
        want = want - (full - marched)
    if skip & 1:
        # escaped rays never march shadows (whole waves of them skip the normal taps too)
        want = np.where(osteps[..., 2] == 0, 0, want)
    assert np.array_equal(gsh, want), "shadow step counts differ"
    if "id" in g:
        assert np.array_equal(g["id"], osteps[..., 2]), "hit ids differ"
    d = np.abs(g["rgb"] - orgb)
    assert np.nanmax(d) <= RGB_TOL, f"max |rgb delta| = {np.nanmax(d)}"
    assert not np.isnan(g["rgb"]).any()
    cd = np.abs(channels(gx) - channels(ox))
    assert cd.max() <= 1, f"XRGB channel delta {cd.max()}"
    if HOST_LIBM_IS_FMA_VARIANT:
        # the kernel's powf restates the libm algorithm this host runs (tests/test_gpu_powf.py): colours are bit-identical
        assert np.array_equal(g["rgb"].view(np.uint32), orgb.view(np.uint32)), "float colours are not bit-identical"
        assert np.array_equal(gx, ox), "packed pixels are not identical"
    return int((gx != ox).sum())


@pytest.mark.parametrize("name,w,h", [
    ("scene", 256, 256), ("scene4", 256, 256), ("scene2", 160, 120), ("scene3", 160, 120),
    ("scene4", 97, 61),          # ragged: neither dimension a multiple of the 16x4 wave patch (nor of the 64x16 regions)
    ("scene", 33, 9), ("scene4", 1, 1), ("scene4", 5, 300),
])
def test_frame_matches_oracle(torch_cuda, renderer, scenes, name, w, h):
    sc = scenes[name]
    g = gpu_render(torch_cuda, renderer, sc, w, h)
    mism = check_against_oracle(g, sc, w, h)
    # On an FMA host (the GPU box is one) check_against_oracle has already required bit-identical colours and packed
    # pixels; elsewhere the host's other powf variant may move a channel by one step where c*255 straddles an integer.
    assert mism == 0 if HOST_LIBM_IS_FMA_VARIANT else mism <= max(4, w * h // 2000), f"{mism} packed pixels differ"


@pytest.mark.parametrize("name,w,h", [
    ("scene4", 256, 256), ("scene", 200, 120), ("scene2", 160, 120), ("scene3", 130, 70),
    ("scene4", 97, 61),          # neither dimension a multiple of the 64x16 region: padding lanes at both edges
    ("scene", 33, 9), ("scene4", 1, 1), ("scene4", 5, 300), ("scene4", 64, 16), ("scene4", 65, 17),
])
def test_repeated_view_matches_oracle(torch_cuda, renderer, scenes, name, w, h):
    """A view rendered again and again (a camera that stands still): from the third frame on its pixels are dealt to the waves
    of their 64x16 region by what they cost, from the fourth the waves are handed out longest first (lol_gpu.hip) — all four
    kernel configurations: the fifth frame's pixels, colours, ids, distances and step counts equal the oracle's like the first's."""
    sc = scenes[name]
    g = gpu_render(torch_cuda, renderer, sc, w, h, repeat=5)
    assert renderer.tile_order()["order"] == "lpt" and renderer.tile_order()["decisions"] >= 1
    mism = check_against_oracle(g, sc, w, h)
    assert mism == 0 if HOST_LIBM_IS_FMA_VARIANT else mism <= max(4, w * h // 2000), f"{mism} packed pixels differ"
    # ... and with a row partition (the middle band of three) and a pitched destination
    part = gpu.Rows(4, 12, 4)
    if h >= 12:
        a = gpu_render(torch_cuda, renderer, sc, w, h, rows=part, pitch_px=w + 3, repeat=5)
        b = gpu_render(torch_cuda, renderer, sc, w, h, rows=part, pitch_px=w + 3)
        for k in ("xrgb", "rgb", "dist", "id", "steps"):
            assert np.array_equal(a[k], b[k]), k
        assert (a["xrgb"][:, w:] == 0x55AA55).all()


def test_hit_distance_and_id_bit_exact(torch_cuda, renderer, scenes):
    sc = scenes["scene4"]
    w, h = 128, 96
    g = gpu_render(torch_cuda, renderer, sc, w, h)
    for y in range(0, h, 7):
        for x in range(0, w, 5):
            p = O.probe(sc, w, h, x, y)
            assert np.float32(p.hit_dist).view(np.uint32) == g["dist"][y, x].view(np.uint32), (x, y)
            assert p.hit_id == g["id"][y, x], (x, y)


@pytest.mark.parametrize("max_steps", [0, 1, 7, 128])
def test_max_steps_parameter(torch_cuda, renderer, scenes, max_steps):
    # BASELINE config 2 runs the primary march with 128 steps; the reference hard-codes 256
    sc = scenes["scene"]
    w, h = 96, 54
    g = gpu_render(torch_cuda, renderer, sc, w, h, max_steps=max_steps)
    check_against_oracle(g, sc, w, h, max_steps=max_steps)


def test_pitch_is_honoured(torch_cuda, renderer, scenes):
    sc = scenes["scene"]
    w, h, pitch = 70, 20, 96
    g = gpu_render(torch_cuda, renderer, sc, w, h, pitch_px=pitch)
    check_against_oracle(g, sc, w, h)
    assert (g["xrgb"][:, w:] == 0x55AA55).all(), "row padding was written"


@pytest.mark.parametrize("n_parts,band", [(2, 8), (4, 4), (8, 4), (3, 16)])
def test_band_partition_reassembles_to_the_full_frame(torch_cuda, renderer, scenes, n_parts, band):
    sc = scenes["scene4"]
    w, h = 64, band * n_parts * 3
    full = gpu_render(torch_cuda, renderer, sc, w, h)["xrgb"]
    out = np.zeros_like(full)
    for part in range(n_parts):
        rows = gpu.Rows.equal(band, n_parts, part)
        g = gpu_render(torch_cuda, renderer, sc, w, h, rows=rows)["xrgb"]
        nb = g.shape[0] // band
        for b in range(nb):
            y = (b * n_parts + part) * band
            out[y:y + band] = g[b * band:(b + 1) * band]
    assert np.array_equal(out, full)


def test_moved_camera(torch_cuda, renderer, scenes):
    sc = scenes["scene4"]
    cam = S.Camera()
    cam.point = S.V3(4.0, 3.0, 2.5)
    d = np.array([-0.5, -0.3, -1.0], dtype=np.float32)
    # normalise like scene.c does (float ops)
    n = np.float32(1.0) / np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + np.float32(d[2] * d[2]), dtype=np.float32)
    cam.direction = S.V3(*(float(np.float32(v * n)) for v in d))
    cam.fov = float(np.float32(np.float32(100.0) / np.float32(180) * np.pi))
    w, h = 80, 48
    g = gpu_render(torch_cuda, renderer, sc, w, h, camera=cam)
    check_against_oracle(g, sc, w, h, camera=cam)


def test_tile_order_does_not_change_a_bit(torch_cuda, renderer, scenes):
    """lol_gpu_set_tile_order: tiles handed out column by column (the launch grid transposed) — whole frames whose size is not a
    multiple of the tile, a pitched destination, and a band partition, all equal to the row-by-row frames and to the oracle."""
    sc = scenes["scene4"]
    w, h = 83, 45
    try:
        renderer.set_tile_order(False)
        rows = gpu_render(torch_cuda, renderer, sc, w, h, pitch_px=w + 5)
        renderer.set_tile_order(True)
        cols = gpu_render(torch_cuda, renderer, sc, w, h, pitch_px=w + 5)
        check_against_oracle(cols, sc, w, h)
        for k in ("xrgb", "rgb", "dist", "id", "steps"):
            assert np.array_equal(rows[k], cols[k]), k
        part = gpu.Rows(4, 12, 4)                     # the middle band of three
        a = gpu_render(torch_cuda, renderer, sc, w, h, rows=part)
        renderer.set_tile_order(False)
        b = gpu_render(torch_cuda, renderer, sc, w, h, rows=part)
        assert np.array_equal(a["xrgb"], b["xrgb"]) and np.array_equal(a["steps"], b["steps"])
    finally:
        renderer.set_tile_order(False)


def test_tile_order_auto_decides_inside_the_library(torch_cuda, scenes):
    """lol_gpu_set_tile_order(AUTO), the default: the first frames of a (scene, size) alternate between the two orders between
    the library's own events; once they have all finished one order is kept; a resize or a new scene decides again; a pinned
    order stops the trials.  Every frame on the way — trial or not — is the same frame."""
    torch = torch_cuda
    sc = scenes["scene4"]
    r = gpu.Renderer(0)
    r.prepare(sc)
    assert r.tile_order()["mode"] == "lpt"                       # (the default is longest tiles first: next test)
    r.set_tile_order("auto")
    info = r.tile_order()
    assert info["mode"] == "auto" and info["decisions"] == 0 and not info["deciding"]
    w, h = 320, 180
    want = None
    buf = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    for i in range(gpu.TILE_TRIAL_FRAMES + 4):
        buf.zero_()
        r.render_into(buf.data_ptr(), w, h)
        r.sync()
        got = buf.clone()
        want = got if want is None else want
        assert torch.equal(got, want), i                       # rows, columns, rows, columns ...: one frame
        if i < gpu.TILE_TRIAL_FRAMES - 1:
            assert r.tile_order()["deciding"]
    info = r.tile_order()
    assert not info["deciding"] and info["decisions"] == 1 and info["order"] in ("rows", "cols")
    assert info["trial_ms"]["rows"] > 0 and info["trial_ms"]["cols"] > 0
    ox, _, _ = O.render(sc, w, h, threads=4)
    assert np.array_equal(want.cpu().numpy().view(np.uint32), ox)
    # the window is resized: another size, another decision
    w2, h2 = 200, 120
    buf2 = torch.zeros((h2, w2), dtype=torch.int32, device="cuda")
    r.render_into(buf2.data_ptr(), w2, h2)
    assert r.tile_order()["deciding"]
    for _ in range(gpu.TILE_TRIAL_FRAMES + 2):
        r.render_into(buf2.data_ptr(), w2, h2)
    r.sync()
    assert r.tile_order()["decisions"] == 2 and not r.tile_order()["deciding"]
    # a new scene: again
    r.prepare(scenes["scene"])
    r.render_into(buf2.data_ptr(), w2, h2)
    assert r.tile_order()["deciding"]
    # pinned: no trials, whatever was in flight is dropped
    r.set_tile_order("cols")
    info = r.tile_order()
    assert info["mode"] == "cols" and info["order"] == "cols" and not info["deciding"]
    r.render_into(buf2.data_ptr(), w2, h2)
    r.sync()
    ox2, _, _ = O.render(scenes["scene"], w2, h2, threads=4)
    assert np.array_equal(buf2.cpu().numpy().view(np.uint32), ox2)
    r.close()


@pytest.mark.parametrize("mode", [1, 4], ids=["spec", "interp"])
def test_longest_tiles_first_renders_every_tile_exactly_once(torch_cuda, scenes, mode):
    """lol_gpu_set_tile_order(LPT), the default: while the camera stands still a launch hands its tiles out in the order of what
    they cost in the frame before (sorted on the device from the run times the tiles report); while it moves, in a fixed order.  The order table must be a permutation of the frame's tiles
    whatever the costs were: every frame — the first (row by row), the one after the first sort, the ones after later sorts,
    after a resize, a new scene, a moving camera, with a row partition, into a pitched destination poisoned beforehand —
    equals the frame of a context that hands its tiles out row by row, and the oracle's."""
    torch = torch_cuda
    a = gpu.Renderer(0, specialize=mode)
    b = gpu.Renderer(0, specialize=mode)
    b.set_tile_order("rows")
    stream = torch.cuda.Stream()

    def frames(sc, w, h, n, rows=None, cams=None, pitch_px=None):
        pitch_px = pitch_px or w
        n_rows = gpu.part_rows(h, rows)
        got = torch.empty((n_rows, pitch_px), dtype=torch.int32, device="cuda")
        want = torch.empty((n_rows, pitch_px), dtype=torch.int32, device="cuda")
        for i in range(n):
            cam = cams[i % len(cams)] if cams else None
            with torch.cuda.stream(stream):
                got.fill_(0x5A5A5A5A)
                want.fill_(0x5A5A5A5A)
            for r, buf in ((a, got), (b, want)):
                r.render_into(buf.data_ptr(), w, h, rows=rows, camera=cam, pitch_bytes=pitch_px * 4, stream=stream.cuda_stream)
            stream.synchronize()
            assert torch.equal(got, want), (w, h, i)
        return got.cpu().numpy().view(np.uint32)

    sc = scenes["scene4"]
    a.prepare(sc); b.prepare(sc)
    assert a.tile_order()["mode"] == "lpt" and a.tile_order()["deciding"]
    last = frames(sc, 333, 187, 19)                              # sizes that are no multiple of the tile: partial tiles at both edges
    info = a.tile_order()
    assert info["order"] == "lpt" and not info["deciding"] and info["decisions"] == 2      # a still camera: sorted after frame 1, and at frame 16
    ox, _, _ = O.render(sc, 333, 187, threads=4)
    assert np.array_equal(last, ox)
    frames(sc, 640, 360, 6, pitch_px=647)                         # resized, pitched: nothing written past the rows (poison intact = equal)
    frames(sc, 200, 120, 6, rows=gpu.Rows(4, 12, 4))              # a row partition: the middle band of three
    cams = []
    for k in range(7):                                           # the camera moves every frame: costs of another view are no prediction
        cam = S.Camera()
        C.memmove(C.byref(cam), C.byref(sc.c.camera), C.sizeof(cam))
        cam.point.x += 0.4 * k
        cams.append(cam)
    sorts = a.tile_order()["decisions"]
    frames(sc, 320, 180, 14, cams=cams)
    info = a.tile_order()
    assert info["decisions"] == sorts and info["order"] in ("rows", "cols")      # a moving camera: a fixed order, nothing is sorted
    frames(sc, 320, 180, 4, cams=cams[3:4])                      # ... it stops: one frame reports its costs, the next is sorted
    info = a.tile_order()
    assert info["decisions"] == sorts + 1 and info["order"] == "lpt" and not info["deciding"]
    sc2 = scenes["scene"]
    a.prepare(sc2); b.prepare(sc2)
    last = frames(sc2, 256, 144, 6)
    ox, _, _ = O.render(sc2, 256, 144, threads=4)
    assert np.array_equal(last, ox)
    # a frame of the same geometry on ANOTHER stream is launched without the table (the tables belong to one stream): same frame
    other = torch.cuda.Stream()
    got = torch.zeros((144, 256), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    a.render_into(got.data_ptr(), 256, 144, stream=other.cuda_stream)
    other.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), ox)
    a.close(); b.close()


def test_camera_beyond_the_sane_range(torch_cuda, renderer, scenes):
    """A camera 10^16 away is not "sane" (lol_gpu.hip, camera_sane): the shadow marches run to the reference's own end and the
    interpreter walks its list WITH v_div_fixup (the proof of the shorter blend factor covers finite operands only)."""
    sc = scenes["scene4"]
    cam = S.Camera()
    cam.point = S.V3(1.0e16, 3.0, 2.5)
    cam.direction = S.V3(-1.0, 0.0, 0.0)
    cam.fov = float(np.float32(np.float32(60.0) / np.float32(180) * np.pi))
    w, h = 64, 16
    g = gpu_render(torch_cuda, renderer, sc, w, h, camera=cam)
    check_against_oracle(g, sc, w, h, camera=cam)


def test_first_step_is_given_or_taken(torch_cuda, renderer, scenes):
    """The primary march's first step, sdf(camera position), comes from the host once per camera position (lol_gpu.hip, first_step;
    lol_kernel.h, march) — or is taken by every pixel itself where the hand-over is not provably the same bits: a component of the
    position that is a negative zero, a first step that ends the march (camera inside an object / within 0.001 of a surface / more
    than 100 from everything), max_steps = 0 and 1.  Distances, ids and step counts (this step included) = the oracle's in each."""
    sc = scenes["scene4"]
    w, h = 72, 40

    def cam_at(x, y, z, dx, dy, dz, fov=90.0):
        cam = S.Camera()
        cam.point = S.V3(x, y, z)
        d = np.array([dx, dy, dz], dtype=np.float32)
        n = np.float32(1.0) / np.sqrt(np.float32(d[0] * d[0] + d[1] * d[1]) + np.float32(d[2] * d[2]), dtype=np.float32)
        cam.direction = S.V3(*(float(np.float32(v * n)) for v in d))
        cam.fov = float(np.float32(np.float32(fov) / np.float32(180) * np.pi))
        return cam
    cams = [
        cam_at(-2.0, 6.0, 3.0, 0.2, -0.5, -1.0),          # the ordinary case: the step is given
        cam_at(-0.0, 6.0, 3.0, 0.0, -0.5, -1.0),          # x = -0: taken per pixel
        cam_at(0.0, 6.0, -0.0, 0.0, -0.5, -1.0),          # z = -0
        cam_at(0.0, 1.0, -6.0, 0.0, 0.0, -1.0),           # inside the blob: the first step is negative and ends the march
        cam_at(0.0, -0.9995, 3.0, 0.0, 0.1, -1.0),        # 0.0005 above the floor: ends on the first step too
        cam_at(0.0, 150.0, 0.0, 0.0, -1.0, -0.01),        # 151 above the floor: the first step overshoots MAX_DIST
        cam_at(0.0, 99.0, 0.0, 0.0, -1.0, -0.01),         # exactly 100 above it: dist == MAX_DIST after one step
    ]
    for cam in cams:
        for max_steps in (256, 1, 0):
            g = gpu_render(torch_cuda, renderer, sc, w, h, max_steps=max_steps, camera=cam)
            check_against_oracle(g, sc, w, h, max_steps=max_steps, camera=cam)


def test_render_host_surface(torch_cuda, renderer, scenes):
    sc = scenes["scene"]
    w, h, pitch = 50, 30, 64 * 4
    renderer.prepare(sc)
    surf = np.full((h, pitch // 4), 0xDEADBEEF, dtype=np.uint32)
    renderer.render_host(surf.ctypes.data, w, h, 256, pitch_bytes=pitch)
    ox, _, _ = O.render_rows(sc, w, h, 0, h)
    assert np.abs(channels(surf[:, :w]) - channels(ox)).max() <= 1
    if HOST_LIBM_IS_FMA_VARIANT:
        assert np.array_equal(surf[:, :w], ox[:h])
    assert (surf[:, w:] == 0xDEADBEEF).all()


def test_full_size_sampled_rows_and_partition(torch_cuda, renderer, scenes):
    """BASELINE config 3 (scene4, 3840x2160, 256 steps): oracle on a few rows + partition invariance."""
    sc = scenes["scene4"]
    w, h = 3840, 2160
    g = gpu_render(torch_cuda, renderer, sc, w, h)
    for y in (0, 777, 1080, 1500, 2159):
        sub = {k: (v[y:y + 1] if isinstance(v, np.ndarray) else v) for k, v in g.items()}
        check_against_oracle(sub, sc, w, h, y0=y, y1=y + 1)
    # size-independent property: 8-way band partition reassembles bit-identically
    band, n_parts = 6, 8
    assert h % (band * n_parts) == 0
    out = np.zeros_like(g["xrgb"])
    for part in range(n_parts):
        pg = gpu_render(torch_cuda, renderer, sc, w, h, rows=gpu.Rows.equal(band, n_parts, part))["xrgb"]
        v = pg.reshape(-1, band, w)
        out.reshape(-1, n_parts, band, w)[:, part] = v
    assert np.array_equal(out, g["xrgb"])


def test_fast_paths_are_proven_exhaustively(torch_cuda, scenes):
    """All 2^32 float inputs: sqrt_fast == sqrtf (outside (0, 2^-96)) and clamp(.5 + x/k) via div_const == exact."""
    r = gpu.Renderer(0)
    for k in (3.0, 1.0, 0.1, 7.5, 1e-3, 1e20):
        sq, dv = r.verify_fast_paths(k)
        assert sq == [0, 0, 0], f"sqrt_pm / sqrt_gs / sqrt_r2 differ from sqrtf on {sq} inputs"
        assert dv == 0, f"smooth-min division by {k} differs on {dv} inputs"
    # the blend factor without its v_div_fixup: proven for the smoothness constants the example scenes use; where it is
    # not (the count says on how many inputs) the generated code keeps the fix-up
    for k in (3.0, 1.0, 0.5, 7.5):
        print("no-fixup mismatches for k =", k, r.verify_smin_no_fixup(k))
    assert r.verify_smin_no_fixup(3.0) == 0
    r.prepare(scenes["scene4"])
    assert "sqrt=3, smin divisors=1 (without div_fixup: 1)" in r.specialize_log()
    r.close()


def test_tiny_squared_length_takes_the_plain_path(torch_cuda, scenes):
    """A camera placed exactly on a sphere centre makes |p-c|^2 == 0 < 2^-96 on the first march step of
    every ray: the wave must re-shade through the plain SDF and still match the oracle."""
    sc = scenes["scene4"]
    cam = S.Camera()
    cam.point = S.V3(0.0, 1.0, -6.0)          # centre of scene4's first sphere
    cam.direction = S.V3(0.0, 0.0, -1.0)
    cam.fov = float(np.float32(np.float32(90.0) / np.float32(180) * np.pi))
    r = gpu.Renderer(0)
    w, h = 64, 32
    g = gpu_render(torch_cuda, r, sc, w, h, camera=cam)
    check_against_oracle(g, sc, w, h, camera=cam)
    r.close()


def test_miss_skip_is_exact_and_conditional(torch_cuda, scenes):
    """Escaped-ray waves skip normal + shadows only when material #0 makes that exact; pixels are identical
    with the skip on and off, and with it off every shadow step count equals the oracle's."""
    sc = scenes["scene4"]
    w, h = 200, 120
    r = gpu.Renderer(0)
    on = gpu_render(torch_cuda, r, sc, w, h)
    assert on["miss_skip"] == 7 and (on["steps"][on["id"] == 0] >> 16).max() == 0
    r.set_miss_skip(False)
    off = gpu_render(torch_cuda, r, sc, w, h)
    assert not off["miss_skip"]
    check_against_oracle(off, sc, w, h)
    assert np.array_equal(on["xrgb"], off["xrgb"]) and np.array_equal(on["rgb"].view(np.uint32), off["rgb"].view(np.uint32))
    # a miss material with a non-zero diffuse term (or a negative shininess) does not qualify
    text = open(__import__("os").path.join(__import__("conftest").SCENES, "scene4.lol")).read()
    for old, new in (("diffuse\t\t= (0, 0, 0)", "diffuse = (0, 0.5, 0)"), ("shininess\t= 0,", "shininess = -1,")):
        assert old in text
        sc2 = S.Scene.parse_string(text.replace(old, new, 1))
        g = gpu_render(torch_cuda, r if False else gpu.Renderer(0), sc2, 64, 40)
        assert not (g["miss_skip"] & 1)
        check_against_oracle(g, sc2, 64, 40)
    r.close()


def test_baseline_configs_2_and_4_at_full_size(torch_cuda, scenes):
    """C2: scene.lol 1920x1080 at 128 march steps; C4: scene4.lol 7680x4320 — oracle on sampled rows, and for C4
    the 8-way band partition the multi-GPU bench uses (band 6) against the single-launch frame."""
    r = gpu.Renderer(0)
    sc, w, h = scenes["scene"], 1920, 1080
    g = gpu_render(torch_cuda, r, sc, w, h, max_steps=128)
    for y in (0, 300, 540, 800, 1079):
        sub = {k: (v[y:y + 1] if isinstance(v, np.ndarray) else v) for k, v in g.items()}
        check_against_oracle(sub, sc, w, h, max_steps=128, y0=y, y1=y + 1)
    del g
    sc, w, h = scenes["scene4"], 7680, 4320
    g = gpu_render(torch_cuda, r, sc, w, h)
    for y in (1, 2160, 3000, 4319):
        sub = {k: (v[y:y + 1] if isinstance(v, np.ndarray) else v) for k, v in g.items()}
        check_against_oracle(sub, sc, w, h, y0=y, y1=y + 1)
    full = g["xrgb"]
    del g
    # ... and EVERY pixel of both frames as a product host gets them (round-5 review): no diagnostics buffers (the kernel without
    # step counters), the fifth frame of a view that repeats (scheduled: waves longest first, pixels dealt by cost), against the
    # oracle's whole frame (all host cores: C4's 33 Mpixels take ~9 s on the GPU box's 16)
    import torch
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4
    for name, fw, fh, steps in (("scene", 1920, 1080, 128), ("scene4", 7680, 4320, 256)):
        fsc = scenes[name]
        r.prepare(fsc)
        buf = torch.zeros((fh, fw), dtype=torch.int32, device="cuda")
        for _ in range(5):
            r.render_into(buf.data_ptr(), fw, fh, steps)
        r.sync()
        assert r.tile_order()["order"] == "lpt"                    # the fifth frame went through the library's tables
        want, _, _ = O.render(fsc, fw, fh, steps, threads=threads)
        got = buf.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), f"{name} {fw}x{fh}: {(got != want).sum()} pixels differ from the oracle"
        del buf, want, got
    from loltracer_amd import multi
    band, n_parts = multi.choose_band_rows(h, 8), 8
    for part in (0, 5):
        pg = gpu_render(torch_cuda, r, sc, w, h, rows=gpu.Rows.equal(band, n_parts, part))["xrgb"]
        assert np.array_equal(pg.reshape(-1, band, w), full.reshape(-1, n_parts, band, w)[:, part])
    r.close()


def test_errors_are_loud(torch_cuda, scenes):
    r = gpu.Renderer(0)
    import torch
    buf = torch.zeros(16, dtype=torch.int32, device="cuda:0")
    with pytest.raises(gpu.GpuError):
        # no program uploaded yet
        fc = scenes["scene"].frame_camera(4, 4)
        r.render_into(buf.data_ptr(), 4, 4, frame_camera=fc)
    r.prepare(scenes["scene"])
    with pytest.raises(gpu.GpuError):
        r.render_into(buf.data_ptr(), 4, 4, pitch_bytes=6)
    with pytest.raises(gpu.GpuError):
        gpu.Renderer(99)
    r.close()


def test_scheduling_options_do_not_change_a_bit(torch_cuda, scenes, monkeypatch):
    """The specialised kernel is compiled with -amdgpu-sched-strategy=max-ilp and without the post-RA scheduler
    (lol_gpu.hip: compile_spec) — instruction ORDER only.  LOL_GPU_SCHED=default compiles it the stock way: every pixel,
    colour bit, distance and step count must be the same."""
    sc = scenes["scene4"]
    w, h = 320, 180
    monkeypatch.setenv("LOL_GPU_CACHE_DIR", "")
    r = gpu.Renderer(0)
    a = gpu_render(torch_cuda, r, sc, w, h)
    r.close()
    monkeypatch.setenv("LOL_GPU_TUNING", "1")      # (the library honours A/B switches only beside this)
    monkeypatch.setenv("LOL_GPU_SCHED", "default")
    r = gpu.Renderer(0)
    b = gpu_render(torch_cuda, r, sc, w, h)
    r.close()
    for key in ("xrgb", "id", "steps"):
        assert np.array_equal(a[key], b[key]), key
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert np.array_equal(a["dist"].view(np.uint32), b["dist"].view(np.uint32))
