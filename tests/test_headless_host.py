"""The C host side of the drop-in: integration/lol_headless.c drives integration/hip_renderer.c
(render_prepare / render_thread / render_destroy) with main.c's semaphore protocol."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "loltracer_amd", "lib", "lol_headless")
SCENE4 = os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol")


def read_ppm(path):
    data = open(path, "rb").read()
    parts = data.split(b"\n", 3)
    assert parts[0] == b"P6"
    w, h = (int(v) for v in parts[1].split())
    return np.frombuffer(parts[3], dtype=np.uint8).reshape(h, w, 3)


def test_protocol_completes_and_fails_loudly_without_a_gpu(tmp_path):
    from loltracer_amd import gpu
    if gpu.gpu_lib().lol_gpu_device_count() > 0:
        pytest.skip("a HIP device is present")
    p = subprocess.run([HOST, "3", SCENE4, "--size", "32x16", "--frames", "2"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0                      # the frame barrier protocol still balances
    assert "no usable HIP device" in p.stderr and "frame skipped" in p.stderr
    assert "Frame 2" in p.stdout and "Cerrando" in p.stdout


def test_bad_scene_is_reported():
    bad = os.path.join(ROOT, "tests", "golden", "scenes", "missing.lol")
    p = subprocess.run([HOST, "1", bad], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "cannot open scene file" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 5])
def test_c_host_renders_through_the_plugin(tmp_path, scenes, threads):
    out = tmp_path / "f.ppm"
    w, h = 160, 90
    p = subprocess.run([HOST, str(threads), SCENE4, "--size", f"{w}x{h}", "--frames", "3", "--out", str(out)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "hip_renderer" not in p.stderr
    img = read_ppm(out).astype(np.int32)
    ox, _, _ = O.render(scenes["scene4"], w, h, threads=4)
    want = np.stack([(ox >> 16) & 0xFF, (ox >> 8) & 0xFF, ox & 0xFF], axis=-1).astype(np.int32)
    assert np.abs(img - want).max() <= 1
    assert "Frame 3" in p.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [[], ["--pipeline"], ["--devices", "0"]])
def test_c_host_gets_the_librarys_scheduling_without_asking(tmp_path, scenes, flags):
    """A host that passes no tile-order flag gets the library's default (lol_gpu.h: LOL_GPU_TILES_LPT): with a camera that
    stands still the frames are scheduled by what the frames before cost (sorts > 0) and the surface still holds the oracle's
    frame; --tile-rows pins the fixed order.  (Round 4's adapter pinned the fixed-order trial by mistake.)"""
    out = tmp_path / "f.ppm"
    w, h = 200, 120
    base = [HOST, "2", SCENE4, "--size", f"{w}x{h}", "--frames", "9", "--wait-kernel", "--report", "--out", str(out)]
    p = subprocess.run(base + flags, capture_output=True, text=True, timeout=180)
    assert p.returncode == 0 and "hip_renderer" not in p.stderr, p.stderr
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("hip_renderer: kernel")]
    assert line and "kernel lol_render_spec" in line[0] and "tile order mode lpt, in use lpt" in line[0], p.stdout
    assert int(line[0].split(",")[-1].split()[0]) >= 1, line[0]
    img = read_ppm(out).astype(np.int32)
    ox, _, _ = O.render(scenes["scene4"], w, h, threads=4)
    want = np.stack([(ox >> 16) & 0xFF, (ox >> 8) & 0xFF, ox & 0xFF], axis=-1).astype(np.int32)
    assert np.abs(img - want).max() <= 1
    p = subprocess.run(base + flags + ["--tile-rows"], capture_output=True, text=True, timeout=180)
    assert p.returncode == 0 and "tile order mode rows, in use rows" in p.stdout, p.stdout


@pytest.mark.gpu
def test_c_host_max_steps_flag(tmp_path, scenes):
    out = tmp_path / "f.ppm"
    w, h = 96, 54
    p = subprocess.run([HOST, "2", SCENE4, "--size", f"{w}x{h}", "--out", str(out), "--max-steps", "9"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    img = read_ppm(out).astype(np.int32)
    ox, _, _ = O.render(scenes["scene4"], w, h, max_steps=9, threads=4)
    want = np.stack([(ox >> 16) & 0xFF, (ox >> 8) & 0xFF, ox & 0xFF], axis=-1).astype(np.int32)
    assert np.abs(img - want).max() <= 1


def test_dump_kernel_flag_is_the_jitdump_counterpart(tmp_path):
    """--dump-kernel BASE (cf. the JIT renderer's --jitdump): generated HIP source + gfx950 code object, no GPU needed."""
    base = tmp_path / "k"
    p = subprocess.run([HOST, "1", SCENE4, "--size", "8x8", "--dump-kernel", str(base)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0
    src = open(str(base) + ".hip").read()
    assert "lol_render_spec" in src and src.count("sd_sphere(") == 2 * 5      # (eval and eval_dist)
    assert os.path.getsize(str(base) + ".co") > 1000


@pytest.mark.gpu
def test_python_cli_renders_the_same_frame(tmp_path, scenes):
    import sys
    out = tmp_path / "p.ppm"
    p = subprocess.run([sys.executable, "-m", "loltracer_amd", SCENE4, "--size", "120x68", "-o", str(out)],
                       capture_output=True, text=True, timeout=180, cwd=ROOT)
    assert p.returncode == 0, p.stderr
    img = read_ppm(out).astype(np.int32)
    ox, _, _ = O.render(scenes["scene4"], 120, 68, threads=4)
    want = np.stack([(ox >> 16) & 0xFF, (ox >> 8) & 0xFF, ox & 0xFF], axis=-1).astype(np.int32)
    assert np.abs(img - want).max() <= 1


@pytest.mark.gpu
def test_pipelined_host_path_delivers_the_same_frames(tmp_path, scenes):
    """lol_gpu_render_host_begin / _end: frame i's copy overlaps frame i+1's kernel.  Frames come out in order and
    equal the synchronous path's; through the C host (--pipeline) the final surface is the last orbit frame."""
    import bench
    from loltracer_amd import gpu
    sc = scenes["scene4"]
    r = gpu.Renderer(0)
    r.prepare(sc)
    w, h, pitch = 200, 120, (200 + 3) * 4
    cams = [bench.orbit_camera(i, 256) for i in (0, 40, 80, 120, 160)]
    want = []
    for cam in cams:
        surf = np.zeros((h, pitch // 4), dtype=np.uint32)
        r.render_host(surf.ctypes.data, w, h, camera=cam, pitch_bytes=pitch)
        want.append(surf)
    got = []
    r.render_host_begin(w, h, camera=cams[0])
    for i in range(len(cams)):
        if i + 1 < len(cams):
            r.render_host_begin(w, h, camera=cams[i + 1])
            assert r.render_host_pending() == 2
        surf = np.zeros((h, pitch // 4), dtype=np.uint32)
        r.render_host_end(surf.ctypes.data, pitch, w, h)
        got.append(surf)
    assert r.render_host_pending() == 0
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    with pytest.raises(gpu.GpuError):
        r.render_host_end(got[0].ctypes.data, pitch, w, h)     # nothing in flight
    r.render_host_begin(w, h); r.render_host_begin(w, h)
    with pytest.raises(gpu.GpuError):
        r.render_host_begin(w, h)                              # a third frame is refused
    r.render_host_end(got[0].ctypes.data, pitch, w, h); r.render_host_end(got[0].ctypes.data, pitch, w, h)
    r.close()

    outs = []
    for flags in ([], ["--pipeline"]):
        out = tmp_path / ("p%d.ppm" % len(flags))
        p = subprocess.run([HOST, "2", SCENE4, "--size", "320x180", "--frames", "6", "--orbit", "--out", str(out)] + flags,
                           capture_output=True, text=True, timeout=120)
        assert p.returncode == 0 and "hip_renderer" not in p.stderr, p.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]


def test_keyboard_to_camera_step_matches_the_reference_arithmetic(tmp_path):
    """The caller side of the boundary: main.c moves the camera from the held keys once per frame
    (update_camera, main.c:70-112).  integration/lol_host_input.h restates that step; the fixture was stepped with the
    reference's own compiled vec.h functions (tests/golden/make_golden.py).  No GPU needed: frames are skipped loudly,
    the protocol and the camera still run."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_camera_path.json")))
    dump = tmp_path / "cam.txt"
    p = subprocess.run([HOST, "2", SCENE4, "--size", "8x8", "--frames", str(len(gold["script"])),
                        "--keys", ",".join(gold["script"]), "--dump-camera", str(dump)],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = [line.split() for line in open(dump).read().splitlines()]
    assert got == gold["path"]
    assert len({tuple(g) for g in got}) > 40                  # the script really moves and turns the camera
