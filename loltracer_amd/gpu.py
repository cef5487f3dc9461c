"""ctypes mirror of include/lol_gpu.h — the gfx950 renderer behind the renderer.h boundary.

`Renderer` plays the role of the reference's renderer plug-in
(render_prepare / render_thread / render_destroy, renderer.h:24-26):
    r = Renderer(device=0); r.prepare(scene)        # render_prepare
    r.render_into(dev_ptr, w, h, max_steps=256)     # one frame of render_thread
    r.close()                                       # render_destroy
All rendering happens in liblol_gpu.so (HIP).  There is no CPU path: if the
library or a device is missing this raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import scene as S

LOL_GPU_OK = 0
LOL_GPU_ABI_VERSION = 6          # include/lol_gpu.h
_STATUS = {0: "ok", -1: "no HIP device", -2: "HIP runtime error", -3: "bad argument",
           -4: "no scene program uploaded", -5: "unsupported"}


class Rows(C.Structure):
    """lol_gpu_rows: the rows one launch renders — `band_rows` rows at `offset_rows` of every `cycle_rows` (multi-GPU
    row tiles; include/lol_gpu.h)."""
    _fields_ = [("band_rows", C.c_int32), ("cycle_rows", C.c_int32), ("offset_rows", C.c_int32)]

    @classmethod
    def equal(cls, band_rows: int, n_parts: int, part: int) -> "Rows":
        """Part `part` of n_parts equal parts with bands of band_rows rows."""
        return cls(band_rows, band_rows * n_parts, band_rows * part)


class PixelFormat(C.Structure):
    """lol_gpu_pixel_format: the SDL_PixelFormat fields SDL_MapRGB reads (renderer.h:17-22)."""
    _fields_ = [("r_shift", C.c_uint8), ("g_shift", C.c_uint8), ("b_shift", C.c_uint8),
                ("r_loss", C.c_uint8), ("g_loss", C.c_uint8), ("b_loss", C.c_uint8),
                ("bytes_per_pixel", C.c_uint8), ("palettised", C.c_uint8), ("a_mask", C.c_uint32)]


# SDL's names for the packed 32-bit formats (SDL_pixels.h) + two the renderer must refuse
PIXEL_FORMATS = {
    "xrgb8888": PixelFormat(16, 8, 0, 0, 0, 0, 4, 0, 0x00000000),
    "argb8888": PixelFormat(16, 8, 0, 0, 0, 0, 4, 0, 0xFF000000),
    "bgrx8888": PixelFormat(8, 16, 24, 0, 0, 0, 4, 0, 0x00000000),
    "rgba8888": PixelFormat(24, 16, 8, 0, 0, 0, 4, 0, 0x000000FF),
    "abgr8888": PixelFormat(0, 8, 16, 0, 0, 0, 4, 0, 0xFF000000),
    "rgb565": PixelFormat(11, 5, 0, 3, 2, 3, 2, 0, 0),
    "index8": PixelFormat(0, 0, 0, 8, 8, 8, 1, 1, 0),
}


class TileOrderInfo(C.Structure):
    """lol_gpu_tile_order_info: what lol_gpu_set_tile_order(AUTO) measured and took."""
    _fields_ = [("mode", C.c_int32), ("order", C.c_int32), ("deciding", C.c_int32), ("decisions", C.c_int32),
                ("rows_ms", C.c_float), ("cols_ms", C.c_float)]


TILES_ROWS, TILES_COLS, TILES_AUTO, TILES_LPT = 0, 1, 2, 3
TILE_TRIALS = 16                     # LOL_GPU_TILE_TRIALS: trial frames per order (after 6 untimed ones)
TILE_TRIAL_FRAMES = 6 + 2 * TILE_TRIALS


def _tile_order_arg(order) -> int:
    if isinstance(order, str):
        return {"rows": TILES_ROWS, "cols": TILES_COLS, "columns": TILES_COLS, "auto": TILES_AUTO, "lpt": TILES_LPT}[order]
    return int(order)                # False / True = rows / columns (the round-3 meaning of the argument), 2 = auto


class Debug(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("hit_dist", C.c_void_p), ("hit_id", C.c_void_p), ("steps", C.c_void_p)]


STREAM_DEFAULT = 1        # LOL_GPU_STREAM_DEFAULT: HIP's legacy default stream (hipStreamLegacy)


def _stream_arg(stream):
    """None → NULL (the context's own stream).  A handle of 0 is how HIP (and torch.cuda.current_stream()
    .cuda_stream for the default stream) spells the legacy default stream: pass it on as such, never as NULL."""
    if stream is None:
        return None
    return C.c_void_p(STREAM_DEFAULT if stream == 0 else stream)


class GpuError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"lol_gpu: {_STATUS.get(status, status)}: {message}")
        self.status = status


_lib = None


def gpu_lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("LOL_GPU_LIB") or os.path.join(S.LIB_DIR, "liblol_gpu.so")     # LOL_GPU_LIB: A/B another build
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension was not built "
                               "(run __graft_entry__.build()); there is no CPU fallback")
        lib = C.CDLL(path)
        P = C.POINTER
        vp = C.c_void_p
        lib.lol_gpu_abi_version.argtypes = []
        lib.lol_gpu_abi_version.restype = C.c_int
        if lib.lol_gpu_abi_version() != LOL_GPU_ABI_VERSION:
            raise RuntimeError(f"{path} speaks ABI version {lib.lol_gpu_abi_version()}, this mirror {LOL_GPU_ABI_VERSION}: "
                               "rebuild (__graft_entry__.build())")
        lib.lol_gpu_device_count.argtypes = []
        lib.lol_gpu_device_count.restype = C.c_int
        lib.lol_gpu_create.argtypes = [C.c_int, P(vp)]
        lib.lol_gpu_create.restype = C.c_int
        lib.lol_gpu_destroy.argtypes = [vp]
        lib.lol_gpu_destroy.restype = None
        lib.lol_gpu_error.argtypes = [vp]
        lib.lol_gpu_error.restype = C.c_char_p
        lib.lol_gpu_upload_program.argtypes = [vp, P(S.Program)]
        lib.lol_gpu_upload_program.restype = C.c_int
        lib.lol_gpu_part_rows.argtypes = [C.c_int, P(Rows)]
        lib.lol_gpu_part_rows.restype = C.c_int
        lib.lol_gpu_render_device.argtypes = [vp, P(S.FrameCamera), C.c_int, C.c_int, C.c_int, P(Rows),
                                              vp, C.c_size_t, P(Debug), vp]
        lib.lol_gpu_render_device.restype = C.c_int
        lib.lol_gpu_render_host.argtypes = [vp, P(S.FrameCamera), C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
        lib.lol_gpu_render_host.restype = C.c_int
        lib.lol_gpu_render_host_begin.argtypes = [vp, P(S.FrameCamera), C.c_int, C.c_int, C.c_int]
        lib.lol_gpu_render_host_begin.restype = C.c_int
        lib.lol_gpu_render_host_end.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int]
        lib.lol_gpu_render_host_end.restype = C.c_int
        lib.lol_gpu_render_host_pending.argtypes = [vp]
        lib.lol_gpu_render_host_pending.restype = C.c_int
        lib.lol_gpu_render_host_pending_size.argtypes = [vp, P(C.c_int), P(C.c_int)]
        lib.lol_gpu_render_host_pending_size.restype = C.c_int
        lib.lol_gpu_render_host_discard.argtypes = [vp]
        lib.lol_gpu_render_host_discard.restype = C.c_int
        lib.lol_gpu_set_pixel_format.argtypes = [vp, P(PixelFormat)]
        lib.lol_gpu_set_pixel_format.restype = C.c_int
        lib.lol_gpu_kernel_key.argtypes = [vp]
        lib.lol_gpu_kernel_key.restype = C.c_char_p
        lib.lol_gpu_roctx_ranges.argtypes = []
        lib.lol_gpu_roctx_ranges.restype = C.c_long
        lib.lol_gpu_sync.argtypes = [vp]
        lib.lol_gpu_sync.restype = C.c_int
        lib.lol_gpu_set_frames_in_flight.argtypes = [vp, C.c_int]
        lib.lol_gpu_set_frames_in_flight.restype = C.c_int
        lib.lol_gpu_frames_in_flight.argtypes = [vp]
        lib.lol_gpu_frames_in_flight.restype = C.c_int
        lib.lol_gpu_next_stream.argtypes = [vp]
        lib.lol_gpu_next_stream.restype = vp
        lib.lol_gpu_tuning_switches.argtypes = []
        lib.lol_gpu_tuning_switches.restype = C.c_char_p
        lib.lol_gpu_malloc.argtypes = [vp, C.c_size_t, P(vp)]
        lib.lol_gpu_malloc.restype = C.c_int
        lib.lol_gpu_free.argtypes = [vp, vp]
        lib.lol_gpu_free.restype = C.c_int
        lib.lol_gpu_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        lib.lol_gpu_memcpy_d2h.restype = C.c_int
        lib.lol_gpu_kernel_name.argtypes = [vp]
        lib.lol_gpu_kernel_name.restype = C.c_char_p
        lib.lol_gpu_set_specialize.argtypes = [vp, C.c_int]
        lib.lol_gpu_set_specialize.restype = C.c_int
        lib.lol_gpu_set_specialize_max_ops.argtypes = [vp, C.c_uint]
        lib.lol_gpu_set_specialize_max_ops.restype = C.c_int
        lib.lol_gpu_specialize_log.argtypes = [vp]
        lib.lol_gpu_specialize_log.restype = C.c_char_p
        lib.lol_gpu_specialize_wait.argtypes = [vp]
        lib.lol_gpu_specialize_wait.restype = C.c_int
        lib.lol_gpu_specialize_state.argtypes = [vp, P(C.c_double)]
        lib.lol_gpu_specialize_state.restype = C.c_int
        lib.lol_gpu_multi_specialize_wait.argtypes = [vp]
        lib.lol_gpu_multi_specialize_wait.restype = C.c_int
        lib.lol_gpu_compile_offline.argtypes = [P(S.Program), C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
        lib.lol_gpu_compile_offline.restype = C.c_int
        lib.lol_gpu_verify_fast_paths.argtypes = [vp, C.c_float, P(C.c_ulonglong), P(C.c_ulonglong)]
        lib.lol_gpu_verify_fast_paths.restype = C.c_int
        lib.lol_gpu_verify_smin_no_fixup.argtypes = [vp, C.c_float, P(C.c_ulonglong)]
        lib.lol_gpu_verify_smin_no_fixup.restype = C.c_int
        lib.lol_gpu_verify_gamma_table.argtypes = [vp, P(C.c_ulonglong), P(C.c_float)]
        lib.lol_gpu_verify_gamma_table.restype = C.c_int
        lib.lol_gpu_powf_batch.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
        lib.lol_gpu_powf_batch.restype = C.c_int
        lib.lol_gpu_set_miss_skip.argtypes = [vp, C.c_int]
        lib.lol_gpu_set_miss_skip.restype = C.c_int
        lib.lol_gpu_set_exact_skips.argtypes = [vp, C.c_uint]
        lib.lol_gpu_set_exact_skips.restype = C.c_int
        lib.lol_gpu_miss_skip_active.argtypes = [vp]
        lib.lol_gpu_miss_skip_active.restype = C.c_int
        lib.lol_gpu_sdf_batch.argtypes = [vp, vp, vp, vp, C.c_size_t, vp]
        lib.lol_gpu_sdf_batch.restype = C.c_int
        lib.lol_gpu_cull_bounds.argtypes = [P(S.Program), C.c_uint32, P(C.c_float), P(C.c_float)]
        lib.lol_gpu_cull_bounds.restype = C.c_int
        lib.lol_gpu_cull_bounds_clusters.argtypes = [P(S.Program), C.c_uint32, P(C.c_float * 4)]
        lib.lol_gpu_cull_bounds_clusters.restype = C.c_int
        lib.lol_gpu_set_cull.argtypes = [vp, C.c_int]
        lib.lol_gpu_set_cull.restype = C.c_int
        lib.lol_gpu_set_tile_order.argtypes = [vp, C.c_int]
        lib.lol_gpu_set_tile_order.restype = C.c_int
        lib.lol_gpu_tile_order.argtypes = [vp, P(TileOrderInfo)]
        lib.lol_gpu_tile_order.restype = C.c_int
        lib.lol_gpu_device.argtypes = [vp]
        lib.lol_gpu_device.restype = C.c_int
        # several devices (include/lol_gpu.h, "Several devices behind the same boundary")
        lib.lol_gpu_multi_create.argtypes = [P(C.c_int), C.c_int, P(vp)]
        lib.lol_gpu_multi_create.restype = C.c_int
        lib.lol_gpu_multi_destroy.argtypes = [vp]
        lib.lol_gpu_multi_destroy.restype = None
        lib.lol_gpu_multi_error.argtypes = [vp]
        lib.lol_gpu_multi_error.restype = C.c_char_p
        lib.lol_gpu_multi_device_count.argtypes = [vp]
        lib.lol_gpu_multi_device_count.restype = C.c_int
        lib.lol_gpu_multi_context.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_context.restype = vp
        lib.lol_gpu_multi_upload_program.argtypes = [vp, P(S.Program)]
        lib.lol_gpu_multi_upload_program.restype = C.c_int
        lib.lol_gpu_choose_band_rows.argtypes = [C.c_int, C.c_int]
        lib.lol_gpu_choose_band_rows.restype = C.c_int
        lib.lol_gpu_multi_set_band_rows.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_set_band_rows.restype = C.c_int
        lib.lol_gpu_part_frame_row.argtypes = [C.c_int, P(Rows), C.c_int]
        lib.lol_gpu_part_frame_row.restype = C.c_int
        lib.lol_gpu_multi_render_device.argtypes = [vp, P(S.FrameCamera), C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
        lib.lol_gpu_multi_render_device.restype = C.c_int
        lib.lol_gpu_multi_render_host.argtypes = [vp, P(S.FrameCamera), C.c_int, C.c_int, C.c_int, vp, C.c_size_t]
        lib.lol_gpu_multi_render_host.restype = C.c_int
        lib.lol_gpu_multi_set_parts_per_device.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_set_parts_per_device.restype = C.c_int
        lib.lol_gpu_multi_set_host_via_root.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_set_host_via_root.restype = C.c_int
        lib.lol_gpu_multi_sync.argtypes = [vp]
        lib.lol_gpu_multi_sync.restype = C.c_int
        lib.lol_gpu_multi_malloc.argtypes = [vp, C.c_size_t, P(vp)]
        lib.lol_gpu_multi_malloc.restype = C.c_int
        lib.lol_gpu_multi_free.argtypes = [vp, vp]
        lib.lol_gpu_multi_free.restype = C.c_int
        lib.lol_gpu_multi_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
        lib.lol_gpu_multi_memcpy_d2h.restype = C.c_int
        lib.lol_gpu_assemble_parts.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
        lib.lol_gpu_assemble_parts.restype = C.c_int
        lib.lol_gpu_assemble_parts_at.argtypes = [vp, vp, P(Rows), P(C.c_uint32), C.c_int, C.c_int, C.c_int, vp, C.c_size_t, vp]
        lib.lol_gpu_assemble_parts_at.restype = C.c_int
        lib.lol_gpu_split_rows.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, P(Rows)]
        lib.lol_gpu_split_rows.restype = C.c_int
        lib.lol_gpu_multi_set_root_band_rows.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_set_root_band_rows.restype = C.c_int
        lib.lol_gpu_multi_set_pixel_format.argtypes = [vp, P(PixelFormat)]
        lib.lol_gpu_multi_set_pixel_format.restype = C.c_int
        lib.lol_gpu_multi_set_tile_order.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_set_tile_order.restype = C.c_int
        # include/lol_gpu_testing.h
        lib.lol_gpu_testing_fail_uploads.argtypes = [vp, C.c_int]
        lib.lol_gpu_testing_fail_uploads.restype = C.c_int
        lib.lol_gpu_testing_fail_first_tier.argtypes = [vp, C.c_int]
        lib.lol_gpu_testing_fail_first_tier.restype = C.c_int
        lib.lol_gpu_testing_has_return_clobbering_branch.argtypes = [C.c_char_p, C.c_size_t]
        lib.lol_gpu_testing_has_return_clobbering_branch.restype = C.c_int
        lib.lol_gpu_multi_testing_root_stride.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_testing_root_stride.restype = C.c_int
        lib.lol_gpu_multi_testing_force_copier_threads.argtypes = [vp, C.c_int]
        lib.lol_gpu_multi_testing_force_copier_threads.restype = C.c_int
        _lib = lib
    return _lib


TESTING_SYMBOLS = ["lol_gpu_testing_fail_uploads", "lol_gpu_testing_fail_first_tier", "lol_gpu_testing_has_return_clobbering_branch", "lol_gpu_multi_testing_root_stride",
                   "lol_gpu_multi_testing_force_copier_threads"]          # include/lol_gpu_testing.h

EXPORTED_SYMBOLS = [                                                    # include/lol_gpu.h
    "lol_gpu_abi_version", "lol_gpu_device_count", "lol_gpu_create", "lol_gpu_destroy", "lol_gpu_error",
    "lol_gpu_upload_program", "lol_gpu_part_rows", "lol_gpu_render_device", "lol_gpu_render_host", "lol_gpu_sync",
    "lol_gpu_malloc", "lol_gpu_free", "lol_gpu_memcpy_d2h", "lol_gpu_kernel_name", "lol_gpu_set_specialize",
    "lol_gpu_specialize_log", "lol_gpu_specialize_wait", "lol_gpu_specialize_state", "lol_gpu_multi_specialize_wait",
    "lol_gpu_compile_offline", "lol_gpu_set_miss_skip", "lol_gpu_miss_skip_active", "lol_gpu_set_exact_skips", "lol_gpu_device",
    "lol_gpu_set_cull", "lol_gpu_set_tile_order", "lol_gpu_tile_order", "lol_gpu_render_host_begin", "lol_gpu_render_host_end",
    "lol_gpu_render_host_pending", "lol_gpu_multi_create", "lol_gpu_multi_destroy", "lol_gpu_multi_error",
    "lol_gpu_multi_device_count", "lol_gpu_multi_context", "lol_gpu_multi_upload_program", "lol_gpu_choose_band_rows",
    "lol_gpu_multi_set_band_rows", "lol_gpu_part_frame_row", "lol_gpu_multi_render_device", "lol_gpu_multi_render_host",
    "lol_gpu_multi_sync", "lol_gpu_multi_malloc", "lol_gpu_multi_free", "lol_gpu_multi_memcpy_d2h", "lol_gpu_assemble_parts",
    "lol_gpu_multi_set_parts_per_device", "lol_gpu_multi_set_host_via_root", "lol_gpu_set_pixel_format",
    "lol_gpu_render_host_pending_size", "lol_gpu_render_host_discard", "lol_gpu_kernel_key", "lol_gpu_assemble_parts_at",
    "lol_gpu_split_rows", "lol_gpu_multi_set_root_band_rows", "lol_gpu_multi_set_pixel_format", "lol_gpu_multi_set_tile_order",
    "lol_gpu_set_frames_in_flight", "lol_gpu_frames_in_flight", "lol_gpu_next_stream", "lol_gpu_set_specialize_max_ops",
]

DIAG_SYMBOLS = [                                                        # include/lol_gpu_diag.h
    "lol_gpu_tuning_switches", "lol_gpu_roctx_ranges", "lol_gpu_verify_fast_paths", "lol_gpu_verify_smin_no_fixup",
    "lol_gpu_verify_gamma_table", "lol_gpu_cull_bounds", "lol_gpu_cull_bounds_clusters", "lol_gpu_powf_batch",
    "lol_gpu_sdf_batch",
]


def tuning_switches() -> str:
    """lol_gpu_tuning_switches: the LOL_GPU_* A/B switches this process has honoured so far (needs LOL_GPU_TUNING=1)."""
    return gpu_lib().lol_gpu_tuning_switches().decode()


def compile_offline(program: S.Program, out_base: str, arch: str = "gfx950", assume_fast: bool = False) -> str:
    """hipRTC-compile the scene-specialised kernel without a device; returns the compiler log."""
    log = C.create_string_buffer(1 << 16)
    st = gpu_lib().lol_gpu_compile_offline(C.byref(program), arch.encode(), os.fsencode(out_base),
                                           int(assume_fast), log, len(log))
    if st != LOL_GPU_OK:
        raise GpuError(st, "hipRTC compile failed:\n" + log.value.decode(errors="replace"))
    return log.value.decode(errors="replace")


def part_rows(h: int, rows: Rows | None) -> int:
    return gpu_lib().lol_gpu_part_rows(h, C.byref(rows) if rows is not None else None)


def split_rows(n_parts: int, band_rows: int, root_band_rows: int = 0, root_stride: int = 1) -> list:
    """lol_gpu_split_rows (pure host logic, no device needed): the Rows of every part of a frame cut into n_parts
    parts with bands of band_rows rows, those of parts p % root_stride == 0 root_band_rows tall (0 = band_rows)."""
    arr = (Rows * n_parts)()
    st = gpu_lib().lol_gpu_split_rows(n_parts, band_rows, root_band_rows, root_stride, arr)
    if st != LOL_GPU_OK:
        raise GpuError(st, f"lol_gpu_split_rows({n_parts}, {band_rows}, {root_band_rows}, {root_stride})")
    return [Rows(r.band_rows, r.cycle_rows, r.offset_rows) for r in arr]


def assemble_parts_at(ctx_renderer, parts_ptr: int, part_rows: list, part_row0: list, w: int, h: int, dst_ptr: int,
                      pitch_bytes: int, stream: int | None):
    """lol_gpu_assemble_parts_at on the renderer's device: un-interleave gathered parts (device pointers)."""
    n = len(part_rows)
    rows = (Rows * n)(*part_rows)
    row0 = (C.c_uint32 * n)(*part_row0)
    st = gpu_lib().lol_gpu_assemble_parts_at(ctx_renderer._ctx, C.c_void_p(parts_ptr), rows, row0, n, w, h,
                                             C.c_void_p(dst_ptr), pitch_bytes, _stream_arg(stream))
    if st != LOL_GPU_OK:
        raise GpuError(st, "lol_gpu_assemble_parts_at")


class Renderer:
    def __init__(self, device: int = 0, specialize: bool | int = True):
        self._lib = gpu_lib()
        self._ctx = C.c_void_p()
        st = self._lib.lol_gpu_create(device, C.byref(self._ctx))
        if st != LOL_GPU_OK:
            self._ctx = C.c_void_p()
            raise GpuError(st, f"lol_gpu_create(device={device}) failed")
        self._lib.lol_gpu_set_specialize(self._ctx, int(specialize))
        self.scene: S.Scene | None = None
        self.program: S.Program | None = None

    def _check(self, st: int):
        if st != LOL_GPU_OK:
            raise GpuError(st, self._lib.lol_gpu_error(self._ctx).decode())

    # render_prepare (renderer.h:25): flatten + upload the scene
    def prepare(self, scene: S.Scene, wait: bool = True):
        """wait=True (tests, benchmarks): also wait for the scene's own kernel, so that the frames that follow all run it.
        wait=False is what the C host does: return as soon as the interpreter can render (tiered start-up, lol_gpu.h)."""
        self.scene = scene
        self.program = scene.flatten()
        self._check(self._lib.lol_gpu_upload_program(self._ctx, C.byref(self.program)))
        if wait:
            self.specialize_wait()

    def upload_program(self, program: S.Program, wait: bool = True):
        self.program = program
        self._check(self._lib.lol_gpu_upload_program(self._ctx, C.byref(program)))
        if wait:
            self.specialize_wait()

    def set_specialize_max_ops(self, max_ops: int):
        """The largest program the scene compiler takes on (0 = the default, 6144 ops); takes effect at the next prepare()."""
        self._check(self._lib.lol_gpu_set_specialize_max_ops(self._ctx, max_ops))

    def specialize_wait(self):
        self._check(self._lib.lol_gpu_specialize_wait(self._ctx))

    def specialize_state(self):
        """(state, compile_ms): 0 none, 1 compiling, 3 compiled (takes over at the next frame), 2 in use, -1 failed; scenes of
        257 ... 1024 ops: 5 = the first kernel (SDF out of line) in use, the second (SDF inlined) compiling, 6 = the second compiled."""
        ms = C.c_double()
        st = self._lib.lol_gpu_specialize_state(self._ctx, C.byref(ms))
        return st, ms.value

    def render_into(self, dst_ptr: int, w: int, h: int, max_steps: int = 256, camera: S.Camera | None = None,
                    rows: Rows | None = None, pitch_bytes: int | None = None, debug: Debug | None = None,
                    stream: int | None = None, frame_camera: S.FrameCamera | None = None):
        """Asynchronously render (a part of) a frame into device memory at dst_ptr."""
        fc = frame_camera if frame_camera is not None else self.scene.frame_camera(w, h, camera)
        self._check(self._lib.lol_gpu_render_device(
            self._ctx, C.byref(fc), w, h, max_steps,
            C.byref(rows) if rows is not None else None,
            C.c_void_p(dst_ptr), pitch_bytes if pitch_bytes is not None else w * 4,
            C.byref(debug) if debug is not None else None,
            _stream_arg(stream)))

    def render_host(self, host_ptr: int, w: int, h: int, max_steps: int = 256, camera: S.Camera | None = None,
                    pitch_bytes: int | None = None):
        """What render_thread does with surf->pixels: whole frame into a host surface, synchronous."""
        fc = self.scene.frame_camera(w, h, camera)
        self._check(self._lib.lol_gpu_render_host(self._ctx, C.byref(fc), w, h, max_steps, C.c_void_p(host_ptr),
                                                  pitch_bytes if pitch_bytes is not None else w * 4))

    def render_host_begin(self, w: int, h: int, max_steps: int = 256, camera: S.Camera | None = None):
        """Queue a frame for the host-surface path (two may be in flight, up to four after set_frames_in_flight);
        render_host_end() delivers the oldest."""
        fc = self.scene.frame_camera(w, h, camera)
        self._check(self._lib.lol_gpu_render_host_begin(self._ctx, C.byref(fc), w, h, max_steps))

    def render_host_end(self, host_ptr: int, pitch_bytes: int, w: int, h: int):
        """Deliver the oldest queued frame into a surface of w x h (refused when the frame has another size)."""
        self._check(self._lib.lol_gpu_render_host_end(self._ctx, C.c_void_p(host_ptr), pitch_bytes, w, h))

    def render_host_pending(self) -> int:
        return int(self._lib.lol_gpu_render_host_pending(self._ctx))

    def render_host_pending_size(self):
        w, h = C.c_int(), C.c_int()
        self._check(self._lib.lol_gpu_render_host_pending_size(self._ctx, C.byref(w), C.byref(h)))
        return w.value, h.value

    def render_host_discard(self):
        self._check(self._lib.lol_gpu_render_host_discard(self._ctx))

    def set_pixel_format(self, fmt: "PixelFormat | str | None"):
        """The surface's SDL_PixelFormat (None = XRGB8888); raises for palettised / non-32-bit formats."""
        if isinstance(fmt, str):
            fmt = PIXEL_FORMATS[fmt]
        self._check(self._lib.lol_gpu_set_pixel_format(self._ctx, C.byref(fmt) if fmt is not None else None))


    def kernel_key(self) -> str:
        return self._lib.lol_gpu_kernel_key(self._ctx).decode()

    def testing_fail_first_tier(self, n: int):
        """lol_gpu_testing_fail_first_tier: the next n out-of-line first runs of the scene compiler count as failed."""
        self._check(self._lib.lol_gpu_testing_fail_first_tier(self._ctx, n))

    def testing_fail_uploads(self, n: int):
        """include/lol_gpu_testing.h: the next n uploads fail at their copy step."""
        self._check(self._lib.lol_gpu_testing_fail_uploads(self._ctx, n))

    def sync(self):
        self._check(self._lib.lol_gpu_sync(self._ctx))

    def set_frames_in_flight(self, n: int):
        """Frames launched with stream=None go to n streams of the context in turn (1 = sequential, the default; <= 4):
        consecutive frames overlap.  Give frames that may be in flight together destinations of their own."""
        self._check(self._lib.lol_gpu_set_frames_in_flight(self._ctx, n))

    def frames_in_flight(self) -> int:
        return int(self._lib.lol_gpu_frames_in_flight(self._ctx))

    def next_stream(self) -> int:
        """hipStream_t handle the next frame launched with stream=None goes to (for events around a frame)."""
        return int(self._lib.lol_gpu_next_stream(self._ctx) or 0)

    def kernel_name(self) -> str:
        return self._lib.lol_gpu_kernel_name(self._ctx).decode()

    def verify_fast_paths(self, k: float = 3.0):
        """([sqrt_pm, sqrt_gs, sqrt_r2], x/k) mismatch counts over all 2^32 float inputs; 0 means proven exact."""
        sq, dv = (C.c_ulonglong * 3)(), C.c_ulonglong()
        self._check(self._lib.lol_gpu_verify_fast_paths(self._ctx, k, sq, C.byref(dv)))
        return list(sq), dv.value

    def verify_smin_no_fixup(self, k: float = 3.0) -> int:
        """Mismatch count of the blend factor without v_div_fixup over all 2^32 inputs; 0 means proven."""
        n = C.c_ulonglong()
        self._check(self._lib.lol_gpu_verify_smin_no_fixup(self._ctx, k, C.byref(n)))
        return n.value

    def verify_gamma_table(self):
        """(mismatches, thresholds): floats in [0, 1] on which gamma through the table and through powf give different channel
        values (0 means proven), and the 257 thresholds of the table."""
        n = C.c_ulonglong()
        t = (C.c_float * 257)()
        self._check(self._lib.lol_gpu_verify_gamma_table(self._ctx, C.byref(n), t))
        return n.value, list(t)

    def powf_batch(self, x_ptr: int, y_ptr: int, out_ptr: int, n: int, stream: int | None = None):
        """out[i] = the kernel's powf(x[i], y[i]) on device arrays (diagnostic for tests)."""
        self._check(self._lib.lol_gpu_powf_batch(self._ctx, C.c_void_p(x_ptr), C.c_void_p(y_ptr), C.c_void_p(out_ptr), n,
                                                 _stream_arg(stream)))

    def sdf_batch(self, pts_ptr: int, dist_ptr: int, id_ptr: int, n: int, stream: int | None = None):
        """dist[i], id[i] = the scene SDF at pts[i] (n x 3 floats), through the SDF code the frames use (diagnostic)."""
        self._check(self._lib.lol_gpu_sdf_batch(self._ctx, C.c_void_p(pts_ptr), C.c_void_p(dist_ptr), C.c_void_p(id_ptr), n,
                                                _stream_arg(stream)))

    def set_cull(self, enable: bool):
        """Exact bounding-sphere culling of top-level objects in the specialised kernel; takes effect at the next prepare()."""
        self._check(self._lib.lol_gpu_set_cull(self._ctx, 1 if enable else 0))

    def set_tile_order(self, order):
        """0 / False / "rows", 1 / True / "cols", 2 / "auto" (the library times both fixed orders on the first frames of a scene
        and size and keeps the faster), 3 / "lpt" (the default: a repeated view is scheduled by what the frame before cost, any
        other frame runs in auto's fixed order) — same pixels either way (lol_gpu.h)."""
        self._check(self._lib.lol_gpu_set_tile_order(self._ctx, _tile_order_arg(order)))

    def tile_order(self) -> dict:
        """lol_gpu_tile_order: mode asked for, order in use, whether trials are still running, the two typical trial frames."""
        info = TileOrderInfo()
        self._check(self._lib.lol_gpu_tile_order(self._ctx, C.byref(info)))
        names = {TILES_ROWS: "rows", TILES_COLS: "cols", TILES_AUTO: "auto", TILES_LPT: "lpt"}
        return {"mode": names[info.mode], "order": names[info.order], "deciding": bool(info.deciding), "decisions": info.decisions,
                "trial_ms": {"rows": round(info.rows_ms, 4), "cols": round(info.cols_ms, 4)}}

    def set_miss_skip(self, enable: bool):
        self._check(self._lib.lol_gpu_set_miss_skip(self._ctx, 1 if enable else 0))

    def set_exact_skips(self, mask: int):
        """bit 0 escaped waves, bit 1 zero incidence, bit 2 settled shadows."""
        self._check(self._lib.lol_gpu_set_exact_skips(self._ctx, mask))

    def miss_skip_active(self) -> int:
        """bit 0: escaped-wave skip active; bit 1: zero-incidence shadow skip active."""
        return int(self._lib.lol_gpu_miss_skip_active(self._ctx))

    def specialize_log(self) -> str:
        return self._lib.lol_gpu_specialize_log(self._ctx).decode(errors="replace")

    # render_destroy (renderer.h:26)
    def close(self):
        if getattr(self, "_ctx", None) and self._ctx.value:
            self._lib.lol_gpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiRenderer:
    """ctypes mirror of the lol_gpu_multi_* entry points: one frame over several devices of this process,
    parts exchanged with RCCL and assembled on devices[0] (include/lol_gpu.h)."""

    def __init__(self, devices, specialize: bool | int = True):
        self._lib = gpu_lib()
        self._m = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        st = self._lib.lol_gpu_multi_create(arr, len(devices), C.byref(self._m))
        if st != LOL_GPU_OK:
            self._m = C.c_void_p()
            raise GpuError(st, f"lol_gpu_multi_create(devices={list(devices)}) failed")
        for i in range(len(devices)):
            self._lib.lol_gpu_set_specialize(self._lib.lol_gpu_multi_context(self._m, i), int(specialize))
        self.scene = None
        self.program = None

    def _check(self, st: int):
        if st != LOL_GPU_OK:
            raise GpuError(st, self._lib.lol_gpu_multi_error(self._m).decode())

    def prepare(self, scene: S.Scene, wait: bool = True):
        self.scene = scene
        self.program = scene.flatten()
        self._check(self._lib.lol_gpu_multi_upload_program(self._m, C.byref(self.program)))
        if wait:
            self._check(self._lib.lol_gpu_multi_specialize_wait(self._m))

    def set_band_rows(self, band_rows: int):
        self._check(self._lib.lol_gpu_multi_set_band_rows(self._m, band_rows))

    def set_parts_per_device(self, parts: int):
        self._check(self._lib.lol_gpu_multi_set_parts_per_device(self._m, parts))

    def set_host_via_root(self, enable: bool):
        self._check(self._lib.lol_gpu_multi_set_host_via_root(self._m, 1 if enable else 0))

    def testing_root_stride(self, stride: int):
        """include/lol_gpu_testing.h: every stride-th part gets the root's band height, even on one device."""
        self._check(self._lib.lol_gpu_multi_testing_root_stride(self._m, stride))

    def testing_force_copier_threads(self, enable: bool):
        """include/lol_gpu_testing.h: host-surface copies through the per-device threads even with one device."""
        self._check(self._lib.lol_gpu_multi_testing_force_copier_threads(self._m, 1 if enable else 0))

    def set_root_band_rows(self, rows: int):
        """Band height of the root's parts (0 = like the others): its smaller share of the rows."""
        self._check(self._lib.lol_gpu_multi_set_root_band_rows(self._m, rows))

    def set_pixel_format(self, fmt):
        if isinstance(fmt, str):
            fmt = PIXEL_FORMATS[fmt]
        self._check(self._lib.lol_gpu_multi_set_pixel_format(self._m, C.byref(fmt) if fmt is not None else None))

    def set_tile_order(self, order):
        self._check(self._lib.lol_gpu_multi_set_tile_order(self._m, _tile_order_arg(order)))

    def tile_order(self, i: int = 0) -> dict:
        """lol_gpu_tile_order of device index i's context."""
        info = TileOrderInfo()
        st = self._lib.lol_gpu_tile_order(self._lib.lol_gpu_multi_context(self._m, i), C.byref(info))
        if st != LOL_GPU_OK:
            raise GpuError(st, "lol_gpu_tile_order")
        names = {TILES_ROWS: "rows", TILES_COLS: "cols", TILES_AUTO: "auto", TILES_LPT: "lpt"}
        return {"mode": names[info.mode], "order": names[info.order], "deciding": bool(info.deciding), "decisions": info.decisions,
                "trial_ms": {"rows": round(info.rows_ms, 4), "cols": round(info.cols_ms, 4)}}

    def render_into(self, dst_ptr: int, w: int, h: int, max_steps: int = 256, camera: S.Camera | None = None,
                    pitch_bytes: int | None = None, frame_camera: S.FrameCamera | None = None):
        fc = frame_camera if frame_camera is not None else self.scene.frame_camera(w, h, camera)
        self._check(self._lib.lol_gpu_multi_render_device(self._m, C.byref(fc), w, h, max_steps, C.c_void_p(dst_ptr),
                                                          pitch_bytes if pitch_bytes is not None else w * 4))

    def render_host(self, host_ptr: int, w: int, h: int, max_steps: int = 256, camera: S.Camera | None = None,
                    pitch_bytes: int | None = None):
        fc = self.scene.frame_camera(w, h, camera)
        self._check(self._lib.lol_gpu_multi_render_host(self._m, C.byref(fc), w, h, max_steps, C.c_void_p(host_ptr),
                                                        pitch_bytes if pitch_bytes is not None else w * 4))

    def sync(self):
        self._check(self._lib.lol_gpu_multi_sync(self._m))

    def kernel_name(self, i: int = 0) -> str:
        return self._lib.lol_gpu_kernel_name(self._lib.lol_gpu_multi_context(self._m, i)).decode()

    def close(self):
        if getattr(self, "_m", None) and self._m.value:
            self._lib.lol_gpu_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
