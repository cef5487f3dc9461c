"""Frames in flight (round 5): the kernels of consecutive frames on different streams of the context.

The reference renders frame i+1 when frame i has been shown (main.c:189-194).  A host that has the next camera in hand —
lol_gpu_render_host_begin / _end, an orbit, the stripes of BASELINE.json's config 5 — may have several frames in flight
(lol_gpu_set_frames_in_flight); their kernels then overlap.  Whatever overlaps, every frame must come out bit-equal to
the frame a sequential context renders of the same camera, and equal to the oracle's:
  - a moving camera (fixed tile order on every stream) and a camera that stands still (one set of scheduling tables per stream);
  - through the host-surface pipeline at depths 2, 3 and 4, frames delivered in order;
  - a changed number of streams, a discarded frame, a tile-order reset with frames in flight on another stream (the
    table-reuse hazard the round-4 advisor described);
  - more streams than the library keeps table sets for.
"""
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from loltracer_amd import gpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def orbit(i):
    import bench
    return bench.orbit_camera(i, 256)


def sequential_frames(torch, sc, w, h, cams, specialize=1):
    """what a context without frames in flight renders, one frame at a time"""
    r = gpu.Renderer(0, specialize=specialize)
    r.prepare(sc)
    r.set_tile_order("rows")
    out = []
    for cam in cams:
        f = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()                          # (the fill runs on torch's stream, the frame on the library's own)
        r.render_into(f.data_ptr(), w, h, camera=cam)
        r.sync()
        out.append(f.cpu().numpy().view(np.uint32))
    r.close()
    return out


@pytest.mark.parametrize("specialize", [1, 4], ids=["spec", "interp"])
@pytest.mark.parametrize("n_streams", [2, 3, 4])
def test_frames_on_several_streams_equal_the_sequential_frames(torch_cuda, scenes, n_streams, specialize):
    torch = torch_cuda
    sc = scenes["scene4"]
    w, h = 328, 188                                       # (not a multiple of the 64x16 regions)
    # a camera that moves every frame, then one that stands still for ten frames, then moves on
    cam_ids = list(range(0, 48, 4)) + [60] * 10 + [64, 68, 72] + [80] * 7
    cams = [orbit(i) for i in cam_ids]
    uniq = {i: f for i, f in zip(cam_ids, sequential_frames(torch, sc, w, h, [orbit(i) for i in cam_ids]))}
    ox, _, _ = O.render(sc, w, h, threads=8, camera=orbit(60))
    assert np.array_equal(uniq[60], ox)                   # and the sequential frame is the oracle's

    r = gpu.Renderer(0, specialize=specialize)
    r.prepare(sc)
    r.set_frames_in_flight(n_streams)
    assert r.frames_in_flight() == n_streams
    ring = [torch.full((h, w), 0x5A5A5A, dtype=torch.int32, device="cuda:0") for _ in range(n_streams)]
    torch.cuda.synchronize()                              # (the fills run on torch's stream, the frames on the library's own)
    seen = set()
    got = []
    for k, cam in enumerate(cams):
        seen.add(r.next_stream())
        r.render_into(ring[k % n_streams].data_ptr(), w, h, camera=cam)       # stream=None: the context's streams in turn
        if k % n_streams == n_streams - 1 or k == len(cams) - 1:
            r.sync()                                      # the ring is full: collect it
            for j in range(k - (k % n_streams), k + 1):
                got.append(ring[j % n_streams].cpu().numpy().view(np.uint32).copy())
    assert len(seen) == n_streams and 0 not in seen       # really n different streams
    assert len(got) == len(cams)
    for k, (frame, cid) in enumerate(zip(got, cam_ids)):
        assert np.array_equal(frame, uniq[cid]), f"frame {k} (orbit camera {cid}) differs from the sequential render"
    info = r.tile_order()
    assert info["mode"] == "lpt"
    # back to one stream: frames are sequential again and still the same
    r.set_frames_in_flight(1)
    f = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    for _ in range(4):
        r.render_into(f.data_ptr(), w, h, camera=orbit(60))
    r.sync()
    assert np.array_equal(f.cpu().numpy().view(np.uint32), uniq[60])
    with pytest.raises(gpu.GpuError):
        r.set_frames_in_flight(5)
    with pytest.raises(gpu.GpuError):
        r.set_frames_in_flight(0)
    r.close()


def test_a_still_camera_is_scheduled_on_every_stream(torch_cuda, scenes):
    """A repeated view keeps its tables per stream: with three frames in flight every stream's frames go through a table
    once the view has repeated on it, sorts happen on every stream, and the frames equal the oracle's."""
    torch = torch_cuda
    sc = scenes["scene4"]
    w, h = 256, 144
    ox, _, _ = O.render(sc, w, h, threads=8)
    r = gpu.Renderer(0)
    r.prepare(sc)
    r.set_frames_in_flight(3)
    ring = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(3)]
    torch.cuda.synchronize()
    for k in range(30):
        r.render_into(ring[k % 3].data_ptr(), w, h)
    r.sync()
    info = r.tile_order()
    assert info["order"] == "lpt" and info["decisions"] >= 3, info       # at least one sort per stream
    for f in ring:
        assert np.array_equal(f.cpu().numpy().view(np.uint32), ox)
    r.close()


def test_tile_order_reset_with_table_frames_in_flight_on_another_stream(torch_cuda, scenes):
    """Round-4 advisor: table frames on stream A, then lol_gpu_set_tile_order, then frames of a repeated view on stream B with no
    host synchronisation in between — B must never rewrite tables that A's frames still read."""
    torch = torch_cuda
    sc = scenes["scene4"]
    w, h = 640, 360
    want = sequential_frames(torch, sc, w, h, [None])[0]
    r = gpu.Renderer(0)
    r.prepare(sc)
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    fa = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(12)]
    fb = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(12)]
    torch.cuda.synchronize()
    for f in fa:                                          # the view repeats on A: tables, dealing, sorts
        r.render_into(f.data_ptr(), w, h, stream=a.cuda_stream)
    r.set_tile_order("lpt")                               # reset while those frames are in flight
    for f in fb:                                          # ... and the view repeats on B at once
        r.render_into(f.data_ptr(), w, h, stream=b.cuda_stream)
    for f in fa[:4]:                                      # and on A again, behind its own frames
        r.render_into(f.data_ptr(), w, h, stream=a.cuda_stream)
    torch.cuda.synchronize()
    for k, f in enumerate(fa + fb):
        assert np.array_equal(f.cpu().numpy().view(np.uint32), want), f"frame {k} differs"
    r.close()


def test_more_streams_than_table_sets(torch_cuda, scenes):
    """Six caller streams rendering one repeated view in turn: four get table sets, the others fixed orders until a set is
    taken over (after its stream has run dry); every frame is the same frame."""
    torch = torch_cuda
    sc = scenes["scene"]
    w, h = 320, 200
    rs = gpu.Renderer(0)
    rs.prepare(sc)
    rs.set_tile_order("rows")
    f0 = torch.zeros((h, w), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    rs.render_into(f0.data_ptr(), w, h, max_steps=128)
    rs.sync()
    want = f0.cpu().numpy().view(np.uint32)
    rs.close()
    r = gpu.Renderer(0)
    r.prepare(sc)
    streams = [torch.cuda.Stream() for _ in range(6)]
    frames = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    for rnd in range(6):
        for s, f in zip(streams, frames):
            r.render_into(f.data_ptr(), w, h, max_steps=128, stream=s.cuda_stream)
        if rnd >= 3:                                      # the last two streams on their own for a while: they take sets over
            for _ in range(3):
                for s, f in zip(streams[4:], frames[4:]):
                    r.render_into(f.data_ptr(), w, h, max_steps=128, stream=s.cuda_stream)
    torch.cuda.synchronize()
    for k, f in enumerate(frames):
        assert np.array_equal(f.cpu().numpy().view(np.uint32), want), f"stream {k}"
    r.close()


@pytest.mark.parametrize("depth", [2, 3, 4])
def test_host_pipeline_depths_deliver_the_sequential_frames_in_order(torch_cuda, scenes, depth):
    sc = scenes["scene4"]
    w, h, pitch = 200, 120, (200 + 5) * 4
    cam_ids = [0, 40, 80, 80, 80, 80, 80, 80, 120, 160, 200, 200, 200, 200, 240]
    r = gpu.Renderer(0)
    r.prepare(sc)
    want = []
    for cid in cam_ids:
        surf = np.zeros((h, pitch // 4), dtype=np.uint32)
        r.render_host(surf.ctypes.data, w, h, camera=orbit(cid), pitch_bytes=pitch)
        want.append(surf)
    if depth > 2:
        r.set_frames_in_flight(depth)
    got = []
    begun = 0
    while len(got) < len(cam_ids):
        while begun < len(cam_ids) and r.render_host_pending() < depth:
            r.render_host_begin(w, h, camera=orbit(cam_ids[begun]))
            begun += 1
        if begun < len(cam_ids):
            assert r.render_host_pending() == depth
            with pytest.raises(gpu.GpuError):
                r.render_host_begin(w, h)                 # every slot is taken
        surf = np.zeros((h, pitch // 4), dtype=np.uint32)
        r.render_host_end(surf.ctypes.data, pitch, w, h)
        got.append(surf)
    for k, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), f"frame {k}"
    # a discarded frame's kernel may still be running into its slot when the slot is used again, on another stream
    for cid in (0, 8, 16)[:depth]:
        r.render_host_begin(w, h, camera=orbit(cid))
    r.render_host_discard()
    assert r.render_host_pending() == 0
    r.render_host_begin(w, h, camera=orbit(40))
    surf = np.zeros((h, pitch // 4), dtype=np.uint32)
    r.render_host_end(surf.ctypes.data, pitch, w, h)
    assert np.array_equal(surf, want[1])
    r.close()


def test_c_host_pipeline_depths_show_the_same_last_frame(tmp_path):
    """lol_headless --pipeline-depth N through render_thread: the surface lags N - 1 frames; after the flush rounds it shows
    the last orbit frame, whatever the depth."""
    from test_headless_host import HOST, SCENE4
    outs = []
    for flags in ([], ["--pipeline"], ["--pipeline-depth", "3"], ["--pipeline-depth", "4"]):
        out = tmp_path / ("p%d.ppm" % len(outs))
        p = subprocess.run([HOST, "2", SCENE4, "--size", "320x180", "--frames", "9", "--orbit", "--out", str(out)] + flags,
                           capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and "hip_renderer" not in p.stderr, p.stderr
        outs.append(open(out, "rb").read())
    assert all(o == outs[0] for o in outs[1:])


def test_random_sequences_through_the_host_pipeline(torch_cuda, scenes):
    """A host that does everything at once: frames of three sizes and five cameras (a camera often repeats: still views follow
    their predecessor's stream, new ones move on), begun and ended in random interleavings up to the depth allowed, frames
    discarded with their kernels still running, the number of frames in flight changed whenever nothing is pending — every
    frame that is delivered must be the frame a sequential context renders for its size and camera, in order."""
    sc = scenes["scene4"]
    rng = np.random.default_rng(20261005)
    sizes = [(200, 120), (97, 61), (320, 180)]
    cams = [None] + [orbit(i) for i in (10, 50, 90, 130)]
    r = gpu.Renderer(0)
    r.prepare(sc)
    want = {}
    for si, (w, h) in enumerate(sizes):
        for ci, cam in enumerate(cams):
            surf = np.zeros((h, w), dtype=np.uint32)
            r.render_host(surf.ctypes.data, w, h, camera=cam)
            want[si, ci] = surf
    depth = 2
    queue = []                                            # (size index, camera index) of the frames begun and not yet ended
    delivered = 0
    ci = 0
    for step in range(400):
        op = rng.random()
        if not queue and op < 0.08:
            depth = int(rng.integers(1, 5))
            r.set_frames_in_flight(depth)
            depth = max(depth, 2)                         # (render_host_begin never accepts fewer than two)
        elif queue and op < 0.12:
            r.render_host_discard()
            queue.clear()
        elif len(queue) < depth and (op < 0.6 or not queue):
            si = int(rng.integers(0, len(sizes)))
            if rng.random() < 0.5:
                ci = int(rng.integers(0, len(cams)))      # else: the camera of the frame before (a repeated view, if the size repeats too)
            w, h = sizes[si]
            r.render_host_begin(w, h, camera=cams[ci])
            queue.append((si, ci))
        elif queue:
            si, cj = queue.pop(0)
            w, h = sizes[si]
            assert r.render_host_pending_size() == (w, h)
            pitch = (w + int(rng.integers(0, 9))) * 4
            surf = np.full((h, pitch // 4), 0xDEADBEEF, dtype=np.uint32)
            r.render_host_end(surf.ctypes.data, pitch, w, h)
            assert np.array_equal(surf[:, :w], want[si, cj]), f"step {step}: frame of size {sizes[si]}, camera {cj}"
            assert np.all(surf[:, w:] == 0xDEADBEEF)
            delivered += 1
        assert r.render_host_pending() == len(queue)
    assert delivered > 100
    r.close()


def test_table_sets_outlive_the_streams_they_lived_on(torch_cuda, scenes):
    """A host may destroy a stream it rendered a repeated view on.  The scheduling tables that lived on it are taken over by the
    next stream that needs a set — after whatever the dead stream still had queued has finished (the device is drained when
    the stream itself can no longer be waited for) — and the frames stay the same frames."""
    import ctypes as C
    torch = torch_cuda
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def new_stream():
        s = C.c_void_p()
        assert hip.hipStreamCreate(C.byref(s)) == 0
        return s

    sc = scenes["scene4"]
    w, h = 384, 216
    want = sequential_frames(torch, sc, w, h, [None])[0]
    r = gpu.Renderer(0)
    r.prepare(sc)
    first = [new_stream() for _ in range(4)]
    frames = [torch.zeros((h, w), dtype=torch.int32, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    for _ in range(5):                                    # the view repeats on four streams: all four table sets are taken
        for s, f in zip(first, frames):
            r.render_into(f.data_ptr(), w, h, stream=s.value)
    for s in first:                                       # ... and the streams go away with frames still queued on them
        assert hip.hipStreamDestroy(s) == 0
    later = [new_stream() for _ in range(2)]
    for _ in range(6):                                    # two new streams: homeless twice, then each takes a dead stream's set over
        for s, f in zip(later, frames[4:]):
            r.render_into(f.data_ptr(), w, h, stream=s.value)
    torch.cuda.synchronize()
    assert r.tile_order()["order"] == "lpt"               # the last frame went through a table again
    for k, f in enumerate(frames):
        assert np.array_equal(f.cpu().numpy().view(np.uint32), want), f"frame buffer {k}"
    r.close()
    for s in later:
        hip.hipStreamDestroy(s)
