"""Several devices behind the C ABI (include/lol_gpu.h, lol_gpu_multi_*).

CPU part: the band partition is pure host logic — every frame row belongs to exactly one local row of
exactly one part, for any frame height (the reference's workers likewise cover every row exactly once,
naive_renderer.c:216).
GPU part (one device on the box): (1) the parts of an N-way partition rendered one by one and put together
by the library's own assembly kernel equal the single-launch frame; (2) the whole multi-device path —
contexts, streams, double-buffered parts, the RCCL group of ncclSend/ncclRecv (a one-rank communicator:
the root sends its part to itself), assembly, D2H — through lol_gpu_multi_* and through the C host with
`--devices 0`, equals the single-device frame.  What a 1-GPU box cannot show is the same exchange between
DIFFERENT devices.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from loltracer_amd import gpu, multi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "loltracer_amd", "lib", "lol_headless")
SCENE4 = os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol")


# ------------------------------------------------------------------ host logic (no GPU)

@pytest.mark.parametrize("h,band,n", [(4320, 12, 8), (2160, 16, 8), (1081, 4, 3), (17, 4, 8), (5, 16, 2), (64, 8, 1)])
def test_every_frame_row_belongs_to_exactly_one_part_row(h, band, n):
    lib = gpu.gpu_lib()
    seen = np.zeros(h, dtype=np.int32)
    for r in range(n):
        rows = gpu.Rows.equal(band, n, r)
        k = gpu.part_rows(h, rows)
        ys = [lib.lol_gpu_part_frame_row(h, C.byref(rows), i) for i in range(k)]
        assert ys == sorted(ys) and all(0 <= y < h for y in ys)
        assert ys == multi.frame_rows_of_part(h, band, n, r).tolist()      # the torch side agrees
        assert all((y // band) % n == r for y in ys)
        seen[ys] += 1
        assert lib.lol_gpu_part_frame_row(h, C.byref(rows), k) == -1 and lib.lol_gpu_part_frame_row(h, C.byref(rows), -1) == -1
    assert (seen == 1).all()


def test_choose_band_rows():
    f = gpu.gpu_lib().lol_gpu_choose_band_rows
    assert f(4320, 8) == 12            # C4 on 8 devices: 45 bands each
    assert f(4320, 1) == 4320          # one device: the whole frame is one band
    assert f(2160, 8) == 16 and f(2160, 2) == 12 and f(2160, 4) == 12      # 2160 = 135 bands of 16: parts differ by one band
    for h in (1080, 1081, 97, 33, 7):
        for n in (2, 3, 8):
            b = f(h, n)
            assert b in (4, 8, 12, 16)
    assert f(1081, 8) == 16 and f(97, 8) == 4
    assert f(0, 2) == 0 and f(10, 0) == 0


def test_multi_create_argument_checks():
    lib = gpu.gpu_lib()
    m = C.c_void_p()
    assert lib.lol_gpu_multi_create(None, 1, C.byref(m)) == -3
    assert lib.lol_gpu_multi_create((C.c_int * 1)(0), 0, C.byref(m)) == -3
    if lib.lol_gpu_device_count() == 0:
        assert lib.lol_gpu_multi_create((C.c_int * 1)(0), 1, C.byref(m)) == -1     # loud, no fallback
        with pytest.raises(gpu.GpuError):
            gpu.MultiRenderer([0])
    assert not m.value


# ------------------------------------------------------------------ on the GPU

def _frame(r, torch, w, h, max_steps=256, **kw):
    t = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    r.render_into(t.data_ptr(), w, h, max_steps, **kw)
    torch.cuda.synchronize()
    r.sync()
    return t


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,band,n", [(640, 360, 12, 8), (333, 181, 4, 3), (256, 97, 16, 2), (128, 17, 4, 8)])
def test_parts_assembled_by_the_library_equal_the_whole_frame(scenes, w, h, band, n):
    import torch
    r = gpu.Renderer(0)
    r.prepare(scenes["scene4"])
    whole = _frame(r, torch, w, h)
    staging = torch.full((h, w), -1, dtype=torch.int32, device="cuda")
    row0 = 0
    for part in range(n):
        rows = gpu.Rows.equal(band, n, part)
        k = gpu.part_rows(h, rows)
        if k:
            r.render_into(staging[row0:].data_ptr(), w, h, 256, rows=rows)
        row0 += k
    assert row0 == h
    r.sync()
    pitch_px = w + 5                                                     # a destination wider than the frame
    out = torch.zeros((h, pitch_px), dtype=torch.int32, device="cuda")
    st = r._lib.lol_gpu_assemble_parts(r._ctx, C.c_void_p(staging.data_ptr()), n, band, w, h,
                                       C.c_void_p(out.data_ptr()), pitch_px * 4, None)
    assert st == 0
    torch.cuda.synchronize()
    assert torch.equal(out[:, :w], whole)
    assert int(out[:, w:].abs().sum()) == 0
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,world,band,root", [(640, 360, 8, 16, 14), (333, 181, 3, 8, 5), (512, 432, 8, 12, 10), (256, 97, 2, 12, 1),
                                                  (640, 360, 8, 12, 0), (128, 50, 4, 16, 15)])
def test_weighted_parts_padded_per_rank_assemble_to_the_whole_frame(scenes, w, h, world, band, root):
    """What rank 0 of the one-process-per-GPU host does when the root's bands are less tall than the others': every rank's
    part (ONE launch) in a buffer padded to the largest rank, the buffers gathered rank by rank, then
    lol_gpu_assemble_parts_at with the geometry of every part and where it starts.  One device renders every rank's part
    here."""
    import torch
    from loltracer_amd import multi
    r = gpu.Renderer(0)
    r.prepare(scenes["scene4"])
    whole = _frame(r, torch, w, h)
    P = multi.Partition(h, world, band, root)
    geometry = [gpu.Rows(*g) for g in P.geometry]
    assert [gpu.part_rows(h, g) for g in geometry] == P.rank_rows
    staging = torch.full((world * P.max_rows, w), -1, dtype=torch.int32, device="cuda")
    pitch_px = w + 3
    torch.cuda.synchronize()
    for p in range(world):
        if not P.rank_rows[p]:
            continue
        r.render_into(staging[P.part_row0[p]:].data_ptr(), w, h, 256, rows=geometry[p])
    r.sync()
    out = torch.zeros((h, pitch_px), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    gpu.assemble_parts_at(r, staging.data_ptr(), geometry, P.part_row0, w, h, out.data_ptr(), pitch_px * 4, None)
    torch.cuda.synchronize()
    assert torch.equal(out[:, :w], whole) and int(out[:, w:].abs().sum()) == 0
    # the CPU-side index of the same partition (what the gloo tests assemble with) agrees
    idx = P.staging_index().cuda()
    assert torch.equal(staging[idx], whole)
    r.close()


@pytest.mark.gpu
def test_multi_path_on_one_device_equals_the_single_device_frame(scenes):
    import torch
    single = gpu.Renderer(0)
    single.prepare(scenes["scene4"])
    m = gpu.MultiRenderer([0])
    m.prepare(scenes["scene4"])
    assert m.kernel_name() == single.kernel_name() == "lol_render_spec"
    for (w, h) in [(3840, 2160), (641, 359)]:
        want = _frame(single, torch, w, h)
        got = torch.zeros((h, w), dtype=torch.int32, device="cuda")
        m.render_into(got.data_ptr(), w, h)
        m.sync()
        assert torch.equal(got, want), (w, h)
        m.set_tile_order(True)                        # tiles column by column on every device: the same frame
        got.zero_()
        m.render_into(got.data_ptr(), w, h)
        m.sync()
        assert torch.equal(got, want), (w, h, "columns")
        m.set_tile_order(False)
    # two frames in flight (double-buffered parts), different cameras, different destinations
    sc = scenes["scene4"]
    cams = []
    for dx in (0.0, 1.5):
        cam = type(sc.c.camera)()
        C.memmove(C.byref(cam), C.byref(sc.c.camera), C.sizeof(cam))
        cam.point.x += dx
        cams.append(cam)
    w, h = 800, 450
    outs = [torch.zeros((h, w), dtype=torch.int32, device="cuda") for _ in cams]
    for _ in range(3):                                                    # slots get reused
        for cam, o in zip(cams, outs):
            m.render_into(o.data_ptr(), w, h, camera=cam)
    m.sync()
    for cam, o in zip(cams, outs):
        assert torch.equal(o, _frame(single, torch, w, h, camera=cam))
    assert not torch.equal(outs[0], outs[1])
    # the host-surface form (what render_thread uses), pitch wider than the row
    w, h, pitch = 320, 200, (320 + 7) * 4
    host = np.zeros((h, pitch // 4), dtype=np.uint32)
    m.render_host(host.ctypes.data, w, h, pitch_bytes=pitch)
    assert np.array_equal(host[:, :w], _frame(single, torch, w, h).cpu().numpy().view(np.uint32))
    assert not host[:, w:].any()
    m.close()
    single.close()


@pytest.mark.gpu
@pytest.mark.parametrize("via_root", [False, True], ids=["direct-copies", "via-root"])
@pytest.mark.parametrize("parts,band,w,h", [(1, 0, 640, 360), (2, 0, 640, 360), (3, 4, 333, 181), (8, 12, 512, 300), (5, 16, 200, 97)])
def test_host_surface_with_several_parts_per_device(scenes, parts, band, w, h, via_root):
    """The multi-part machinery on ONE device: `parts` parts owned by device 0 (bands dealt over them), each part
    rendered by its own launch, then either copied band by band straight into the host surface (strided 3-D copies —
    what N devices do in parallel over N PCIe links) or sent through the RCCL exchange and assembled on the root.
    Both must give the single-launch frame, for heights that are not a multiple of the band and a padded pitch."""
    import torch
    single = gpu.Renderer(0)
    single.prepare(scenes["scene4"])
    want = _frame(single, torch, w, h).cpu().numpy().view(np.uint32)
    m = gpu.MultiRenderer([0])
    m.prepare(scenes["scene4"])
    m.set_parts_per_device(parts)
    m.set_band_rows(band)
    m.set_host_via_root(via_root)
    pitch = (w + 9) * 4
    for _ in range(2):                                          # twice: both buffer slots
        host = np.full((h, pitch // 4), 0xDEADBEEF, dtype=np.uint32)
        m.render_host(host.ctypes.data, w, h, pitch_bytes=pitch)
        assert np.array_equal(host[:, :w], want)
        assert (host[:, w:] == 0xDEADBEEF).all()                # nothing written past the row
    # and the device-resident form with the same split
    got = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    m.render_into(got.data_ptr(), w, h)
    m.sync()
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
    m.close()
    single.close()


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,parts,band,root,stride", [(640, 360, 8, 12, 10, 8), (333, 181, 6, 8, 5, 3), (1280, 720, 8, 16, 15, 8)])
def test_unequal_bands_through_the_multi_device_path(scenes, monkeypatch, w, h, parts, band, root, stride):
    """The root's lighter bands (lol_gpu_multi_set_root_band_rows) need a second device to mean anything; the test switch
    lol_gpu_multi_testing_root_stride gives every stride-th PART of ONE device the root's band height instead, so that bands of
    unequal height go through the real split, the per-part launches, the RCCL exchange, the assembly kernel and the strided
    host copies: device-resident frame and host surface equal the single-launch frame."""
    import torch
    single = gpu.Renderer(0)
    single.prepare(scenes["scene4"])
    want = _frame(single, torch, w, h).cpu().numpy().view(np.uint32)
    m = gpu.MultiRenderer([0])
    m.prepare(scenes["scene4"])
    m.testing_root_stride(stride)
    m.set_parts_per_device(parts)
    m.set_band_rows(band)
    m.set_root_band_rows(root)
    got = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(2):                                          # both buffer slots
        m.render_into(got.data_ptr(), w, h)
        m.sync()
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want)
    for via_root in (False, True):
        m.set_host_via_root(via_root)
        pitch = (w + 5) * 4
        host = np.full((h, pitch // 4), 0xDEADBEEF, dtype=np.uint32)
        m.render_host(host.ctypes.data, w, h, pitch_bytes=pitch)
        assert np.array_equal(host[:, :w], want) and (host[:, w:] == 0xDEADBEEF).all()
    m.close()
    single.close()


@pytest.mark.gpu
@pytest.mark.parametrize("parts,band,w,h", [(1, 0, 640, 360), (3, 4, 333, 181), (8, 12, 1920, 1080)])
def test_host_copies_issued_by_the_per_device_threads(scenes, parts, band, w, h):
    """With several devices every device's copies into the host surface are issued by a host thread of its own
    (lol_multi.hip, Worker: submit / wait / stop, hipSetDevice inside the thread, errors merged); one device copies
    inline.  lol_gpu_multi_testing_force_copier_threads sends the one device of this box down the threaded path:
    same frames, repeatedly (the thread is reused), after a resize, and the context shuts its thread down cleanly."""
    import torch
    single = gpu.Renderer(0)
    single.prepare(scenes["scene4"])
    m = gpu.MultiRenderer([0])
    m.prepare(scenes["scene4"])
    m.testing_force_copier_threads(True)
    m.set_parts_per_device(parts)
    m.set_band_rows(band)
    for (fw, fh) in [(w, h), (w // 2 + 1, h // 2 + 3), (w, h)]:
        want = _frame(single, torch, fw, fh).cpu().numpy().view(np.uint32)
        pitch = (fw + 3) * 4
        for _ in range(3):
            host = np.full((fh, pitch // 4), 0xDEADBEEF, dtype=np.uint32)
            m.render_host(host.ctypes.data, fw, fh, pitch_bytes=pitch)
            assert np.array_equal(host[:, :fw], want) and (host[:, fw:] == 0xDEADBEEF).all()
    m.testing_force_copier_threads(False)                       # and back to the inline copy on the same context
    host = np.zeros((h, w), dtype=np.uint32)
    m.render_host(host.ctypes.data, w, h)
    assert np.array_equal(host, _frame(single, torch, w, h).cpu().numpy().view(np.uint32))
    m.close()
    single.close()


@pytest.mark.gpu
def test_duplicate_devices_are_refused():
    with pytest.raises(gpu.GpuError) as e:
        gpu.MultiRenderer([0, 0])
    assert e.value.status == -3


@pytest.mark.gpu
def test_c_host_devices_flag_renders_the_same_frame(tmp_path):
    outs = []
    for flags in (["--device", "0"], ["--devices", "0"]):
        out = tmp_path / ("f" + flags[0].strip("-") + ".ppm")
        p = subprocess.run([HOST, "3", SCENE4, "--size", "7680x4320", "--frames", "2", "--out", str(out)] + flags,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "hip_renderer" not in p.stderr, p.stderr
        assert "Frame 2" in p.stdout
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and len(outs[0]) > 7680 * 4320 * 3
