"""ctypes loader for the CPU oracle (oracle/liblol_oracle.so). Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from loltracer_amd import scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liblol_oracle.so")


class Counters(C.Structure):
    _fields_ = [("pixels", C.c_uint64), ("sdf_evals", C.c_uint64), ("node_evals", C.c_uint64),
                ("march_steps", C.c_uint64), ("shadow_steps", C.c_uint64), ("miss_pixels", C.c_uint64),
                ("settle_violations", C.c_uint64)]


PROBE_LIGHTS = 64          # LOL_ORACLE_PROBE_LIGHTS (oracle/lol_oracle.h)


class Probe(C.Structure):
    _fields_ = [("rd", C.c_float * 3), ("hit_dist", C.c_float), ("hit_id", C.c_uint32),
                ("march_steps", C.c_uint32), ("normal", C.c_float * 3),
                ("shadow", C.c_float * PROBE_LIGHTS), ("shadow_steps", C.c_uint32 * PROBE_LIGHTS),
                ("rgb_linear", C.c_float * 3), ("rgb", C.c_float * 3), ("xrgb", C.c_uint32)]


class PixelFormat(C.Structure):
    _fields_ = [("r_shift", C.c_uint8), ("g_shift", C.c_uint8), ("b_shift", C.c_uint8),
                ("r_loss", C.c_uint8), ("g_loss", C.c_uint8), ("b_loss", C.c_uint8), ("a_mask", C.c_uint32)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError(f"{ORACLE_SO} missing: run __graft_entry__.build() or make -C oracle")
        l = C.CDLL(ORACLE_SO)
        P = C.POINTER
        sp, cp = P(S.SceneStruct), P(S.Camera)
        l.lol_oracle_render_rows.argtypes = [sp, cp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, P(Counters)]
        l.lol_oracle_render_rows.restype = None
        l.lol_oracle_render_frame.argtypes = [sp, cp, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_size_t, C.c_void_p, P(Counters)]
        l.lol_oracle_render_frame.restype = None
        l.lol_oracle_render_sample.argtypes = [sp, cp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.c_int, C.c_void_p, C.c_size_t, P(Counters)]
        l.lol_oracle_render_sample.restype = None
        l.lol_oracle_probe_pixel.argtypes = [sp, cp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(Probe)]
        l.lol_oracle_probe_pixel.restype = None
        l.lol_oracle_sdf.argtypes = [sp, C.c_float, C.c_float, C.c_float, P(C.c_uint32)]
        l.lol_oracle_sdf.restype = C.c_float
        l.lol_oracle_powf_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        l.lol_oracle_powf_batch.restype = None
        l.lol_oracle_hash_xrgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        l.lol_oracle_hash_xrgb.restype = C.c_uint64
        l.lol_oracle_set_pixel_format.argtypes = [P(PixelFormat)]
        l.lol_oracle_set_pixel_format.restype = None
        f3 = P(C.c_float)
        for name, args, res in [
            ("lol_oracle_minf", [C.c_float] * 2, C.c_float), ("lol_oracle_maxf", [C.c_float] * 2, C.c_float),
            ("lol_oracle_clamp", [C.c_float] * 3, C.c_float), ("lol_oracle_sminf", [C.c_float] * 3, C.c_float),
            ("lol_oracle_v3dot", [f3, f3], C.c_float), ("lol_oracle_v3len", [f3], C.c_float),
            ("lol_oracle_v3normalize", [f3, f3], None), ("lol_oracle_v3cross", [f3, f3, f3], None),
            ("lol_oracle_v3clamp", [f3, C.c_float, C.c_float, f3], None),
            ("lol_oracle_sd_sphere", [f3, C.c_float], C.c_float),
            ("lol_oracle_sd_round_box", [f3, f3, C.c_float], C.c_float),
        ]:
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = res
        _lib = l
    return _lib


def render(scene: S.Scene, w: int, h: int, max_steps: int = 256, threads: int = 1, camera=None,
           want_rgb: bool = False, want_counters: bool = False):
    """Whole frame → (xrgb uint32 [h,w], rgb float32 [h,w,3] | None, Counters | None)."""
    cam = camera if camera is not None else scene.c.camera
    xrgb = np.zeros((h, w), dtype=np.uint32)
    rgb = np.zeros((h, w, 3), dtype=np.float32) if want_rgb else None
    ctr = Counters() if want_counters else None
    lib().lol_oracle_render_frame(scene.ptr, C.byref(cam), w, h, max_steps, threads,
                                  xrgb.ctypes.data, w * 4,
                                  rgb.ctypes.data if rgb is not None else None,
                                  C.byref(ctr) if ctr is not None else None)
    return xrgb, rgb, ctr


def render_rows(scene: S.Scene, w: int, h: int, y0: int, y1: int, max_steps: int = 256, camera=None,
                want_steps: bool = False):
    """Rows [y0,y1) → (xrgb [h,w] with only those rows filled, rgb [h,w,3], steps [h,w,12] | None);
    steps[..., 0] march steps, [..., 1] shadow steps summed over lights, [..., 2] hit id, [..., 3] bit mask of
    lights with diffuse incidence exactly 0, [..., 4:8] shadow steps of lights 0..3, [..., 8:12] their "settled"
    shadow steps (up to and including the first step that left res <= 0)."""
    cam = camera if camera is not None else scene.c.camera
    xrgb = np.zeros((h, w), dtype=np.uint32)
    rgb = np.zeros((h, w, 3), dtype=np.float32)
    steps = np.zeros((h, w, 12), dtype=np.uint16) if want_steps else None
    global last_counters
    last_counters = Counters()
    lib().lol_oracle_render_rows(scene.ptr, C.byref(cam), w, h, max_steps, y0, y1,
                                 xrgb.ctypes.data, w * 4, rgb.ctypes.data,
                                 steps.ctypes.data if steps is not None else None, C.byref(last_counters))
    return xrgb, rgb, steps


last_counters = None          # the work counters of the last render_rows() call (settle_violations must stay 0)


def probe(scene: S.Scene, w: int, h: int, x: int, y: int, max_steps: int = 256, camera=None) -> Probe:
    cam = camera if camera is not None else scene.c.camera
    p = Probe()
    lib().lol_oracle_probe_pixel(scene.ptr, C.byref(cam), w, h, max_steps, x, y, C.byref(p))
    return p


def powf(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Host libm powf, elementwise on float32 arrays."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    lib().lol_oracle_powf_batch(x.ctypes.data, y.ctypes.data, out.ctypes.data, x.size)
    return out


def hash_xrgb(xrgb: np.ndarray) -> int:
    a = np.ascontiguousarray(xrgb, dtype=np.uint32)
    h, w = a.shape
    return int(lib().lol_oracle_hash_xrgb(a.ctypes.data, w, h, w * 4))


def set_pixel_format(fmt=None):
    """The checker's surface format (None = XRGB8888): shifts, losses, Amask of an SDL_PixelFormat, or a
    loltracer_amd.gpu.PixelFormat.  Process-wide: reset it (None) when done."""
    if fmt is None:
        lib().lol_oracle_set_pixel_format(None)
        return
    f = PixelFormat(fmt.r_shift, fmt.g_shift, fmt.b_shift, fmt.r_loss, fmt.g_loss, fmt.b_loss, fmt.a_mask)
    lib().lol_oracle_set_pixel_format(C.byref(f))
