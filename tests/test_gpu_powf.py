"""The kernel's powf (lol_kernel.h: powf_glibc) against the CPU libm's powf, bit for bit.

naive_renderer.c calls powf for the specular term and the gamma curve; the device restates glibc's algorithm
(the FMA build x86-64 selects) so that colours — and hence packed pixels — round identically on both sides.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import gpu

pytestmark = pytest.mark.gpu


def has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available()
    r = gpu.Renderer(0)
    yield torch, r
    r.close()


def device_powf(torch, r, x, y):
    xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    yd = torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).cuda()
    out = torch.empty_like(xd)
    r.powf_batch(xd.data_ptr(), yd.data_ptr(), out.data_ptr(), xd.numel(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def same_bits(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


# LOL_TEST_EXHAUSTIVE=1: every float in [0, 1] for every exponent below (14 x 1.07e9 inputs, about two minutes on the
# GPU box); the default run takes every one for the gamma exponent and every 13th for the others.
EXHAUSTIVE = os.environ.get("LOL_TEST_EXHAUSTIVE") == "1"
OTHER_EXPONENTS = [16.0, 25.0, 0.0, 2.0, 3.0, 50.0, 10.0, 4.0, 30.5, 8.0, 1.0, -1.5, 0.3]


@pytest.mark.skipif(not has_fma(), reason="host libm would run its non-FMA powf variant here")
@pytest.mark.parametrize("y,stride", [(1.0 / 2.2, 1)] + [(e, 1 if EXHAUSTIVE else 13) for e in OTHER_EXPONENTS])
def test_every_colour_input_for_the_exponents_in_use(ctx, y, stride):
    """x runs over the floats in [0, 1] (what clamp() hands to powf): EVERY one for the gamma exponent 1/2.2f; for the
    shininess values of the example scenes and a few others every 13th, or every one under LOL_TEST_EXHAUSTIVE=1."""
    torch, r = ctx
    bad = 0
    step = 1 << 26
    for start in range(0, 0x3F800000 + 1, step):
        bits = np.arange(start, min(start + step, 0x3F800000 + 1), stride, dtype=np.uint32)
        x = bits.view(np.float32)
        yy = np.full(x.shape, np.float32(y), dtype=np.float32)
        bad += int((~same_bits(device_powf(torch, r, x, yy), O.powf(x, yy))).sum())
    assert bad == 0


@pytest.mark.skipif(not has_fma(), reason="host libm would run its non-FMA powf variant here")
def test_random_pairs_including_specials(ctx):
    torch, r = ctx
    rng = np.random.default_rng(11)
    n = 1 << 24
    x = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    y = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, 2.0, 3.0, -3.0, 1e-40, -1e-40, 255.0], dtype=np.float32)
    xs, ys = np.meshgrid(special, special)
    x = np.concatenate([x, xs.ravel(), rng.uniform(0, 1, 1 << 20).astype(np.float32)])
    y = np.concatenate([y, ys.ravel(), rng.uniform(-60, 60, 1 << 20).astype(np.float32)])
    ok = same_bits(device_powf(torch, r, x, y), O.powf(x, y))
    assert ok.all(), (x[~ok][:5], y[~ok][:5])


def test_gamma_table_equals_powf_for_every_colour_input(ctx):
    """Gamma + quantisation through the 256 thresholds (lol_kernel.h: gamma_u8_table) against (Uint8)(powf(c, 1/2.2f) * 255) for
    EVERY float c in [0, 1], on the device (the sweep a context runs before its frames use the table), and the thresholds
    themselves against the CPU's powf: T[k] is the first float whose channel value reaches k."""
    import ctypes
    torch, r = ctx
    bad, T = r.verify_gamma_table()
    assert bad == 0
    T = np.array(T, dtype=np.float32)
    assert T[0] == 0.0 and np.isinf(T[256]) and T[256] > 0 and T[255] <= 1.0
    assert np.all(np.diff(T[:256]) > 0), "thresholds strictly increasing"
    assert abs(float(T[1]) - (1 / 255) ** 2.2) < 1e-7 and abs(float(T[128]) - (128 / 255) ** 2.2) < 1e-5
    if has_fma():
        libm = ctypes.CDLL("libm.so.6")
        libm.powf.restype = ctypes.c_float
        libm.powf.argtypes = [ctypes.c_float, ctypes.c_float]
        g = np.float32(1.0) / np.float32(2.2)

        def level(c):
            return int(np.float32(libm.powf(float(c), float(g))) * np.float32(255.0)) & 0xFF
        for k in range(1, 256):
            below = np.nextafter(T[k], np.float32(0.0), dtype=np.float32)
            assert level(T[k]) == k and level(below) == k - 1, k


def test_frames_with_and_without_the_gamma_table_are_the_same(ctx):
    """A frame whose 8 bits come from the table (the default) equals the frame of a context that keeps powf (specialize mode 3 /
    0: no shortcuts), with and without the diagnostics buffers (with them the float colour is computed as well)."""
    import oracle_lib  # noqa: F401  (tests/ on the path)
    from loltracer_amd import scene as S
    from test_gpu_parity import gpu_render
    torch, _ = ctx
    sc = S.Scene.parse_file(os.path.join(os.path.dirname(__file__), "golden", "scenes", "scene4.lol"))
    w, h = 333, 187
    frames = {}
    for mode in (1, 3, 4, 0):
        r = gpu.Renderer(0, specialize=mode)
        try:
            frames[mode] = gpu_render(torch, r, sc, w, h)["xrgb"]
            plain = torch.zeros((h, w), dtype=torch.int32, device="cuda")
            r.render_into(plain.data_ptr(), w, h, 256)      # no diagnostics: the table route alone (modes 1, 4)
            r.sync()
            assert np.array_equal(plain.cpu().numpy().view(np.uint32), frames[mode][:, :w]), mode
        finally:
            r.close()
    for mode in (3, 4, 0):
        assert np.array_equal(frames[1], frames[mode]), mode
