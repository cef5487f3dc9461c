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
