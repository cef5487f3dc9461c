"""The C-ABI libraries load and export every symbol the headers declare (no compute, no GPU)."""
import ctypes as C
import os
import re

import numpy as np

from loltracer_amd import gpu, scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lol_[a-z0-9_]+)\s*\(", text)))


def test_lol_gpu_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(S.LIB_DIR, "liblol_gpu.so"))
    names = declared("lol_gpu.h")
    assert set(names) == set(gpu.EXPORTED_SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None


def test_lol_gpu_diag_header_is_exported_too():
    """The proofs, bounds and probes live in a header of their own, so that lol_gpu.h is what a host reads."""
    lib = C.CDLL(os.path.join(S.LIB_DIR, "liblol_gpu.so"))
    names = declared("lol_gpu_diag.h")
    assert set(names) == set(gpu.DIAG_SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None
    for host in ("hip_renderer.c", "lol_headless.c"):
        text = open(os.path.join(ROOT, "integration", host)).read()
        assert "lol_gpu_diag.h" not in text and not any(n + "(" in text for n in names), host


def test_lol_gpu_testing_header_is_exported_too():
    lib = C.CDLL(os.path.join(S.LIB_DIR, "liblol_gpu.so"))
    names = declared("lol_gpu_testing.h")
    assert set(names) == set(gpu.TESTING_SYMBOLS)
    for n in names:
        assert getattr(lib, n) is not None


def test_abi_version_of_the_library_is_the_headers():
    text = open(os.path.join(ROOT, "include", "lol_gpu.h")).read()
    want = int(re.search(r"#define\s+LOL_GPU_ABI_VERSION\s+(\d+)", text).group(1))
    assert gpu.gpu_lib().lol_gpu_abi_version() == want == gpu.LOL_GPU_ABI_VERSION


def test_no_test_switch_is_read_from_the_environment():
    """Fault injection and test geometry are per-context setters (include/lol_gpu_testing.h), never ambient variables."""
    for f in ("lol_gpu.hip", "lol_multi.hip"):
        src = open(os.path.join(ROOT, "loltracer_amd", "csrc", f)).read()
        assert not re.search(r'getenv\("LOL_GPU[A-Z_]*TEST', src), f


def test_lol_scene_exports_every_declared_symbol():
    lib = C.CDLL(os.path.join(S.LIB_DIR, "liblol_scene.so"))
    names = declared("lol_scene.h")
    assert "lol_scene_parse_file" in names and "lol_scene_flatten" in names
    for n in names:
        assert getattr(lib, n) is not None


def test_struct_sizes_match_the_headers():
    # the Python mirrors must match the C layouts (lol_op is 10 dwords, see lol_scene.h)
    assert C.sizeof(S.Op) == 40 and C.sizeof(S.Light) == 36 and C.sizeof(S.Material) == 40
    assert C.sizeof(S.Node) == 48 and C.sizeof(S.FrameCamera) == 56
    # a program is counts + four pointers (no capacity: include/lol_scene.h); the caps are sanity bounds only
    hdr = open(os.path.join(ROOT, "include", "lol_scene.h")).read()
    caps = {k: eval(v.replace("u", "")) for k, v in re.findall(r"#define\s+(LOL_MAX_[A-Z]+)\s+(\(?[0-9u <]+\)?)", hdr)}
    assert caps == {"LOL_MAX_OPS": S.LOL_MAX_OPS, "LOL_MAX_LIGHTS": S.LOL_MAX_LIGHTS, "LOL_MAX_MATERIALS": S.LOL_MAX_MATERIALS,
                    "LOL_MAX_STACK": S.LOL_MAX_STACK}
    assert min(S.LOL_MAX_OPS, S.LOL_MAX_LIGHTS, S.LOL_MAX_MATERIALS) >= 1 << 16
    assert C.sizeof(S.Program) == 5 * 4 + 12 + 4 * C.sizeof(C.c_void_p)


def test_part_rows_is_pure_host_logic():
    assert gpu.part_rows(2160, None) == 2160
    assert [gpu.part_rows(4320, gpu.Rows.equal(4, 8, r)) for r in range(8)] == [540] * 8
    assert [gpu.part_rows(20, gpu.Rows.equal(8, 2, r)) for r in range(2)] == [12, 8]
    assert gpu.part_rows(10, gpu.Rows.equal(0, 1, 0)) == -1
    # bands of different heights inside one cycle (the root's smaller share): 7 + 3 x 8 rows per cycle of 31
    split = gpu.split_rows(4, 8, 7, 4)
    assert [(r.band_rows, r.cycle_rows, r.offset_rows) for r in split] == [(7, 31, 0), (8, 31, 7), (8, 31, 15), (8, 31, 23)]
    assert [gpu.part_rows(100, r) for r in split] == [28, 24, 24, 24] and sum(gpu.part_rows(4320, r) for r in split) == 4320
    assert gpu.part_rows(10, gpu.Rows(4, 8, 5)) == -1          # the band sticks out of its cycle


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import pytest
    if gpu.gpu_lib().lol_gpu_device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(gpu.GpuError) as e:
        gpu.Renderer(0)
    assert e.value.status == -1


def test_scene_specialised_kernel_compiles_without_a_device(tmp_path, scenes):
    """hipRTC cross-compiles the generated straight-line SDF for gfx950 (no GPU needed)."""
    import subprocess
    for name in ("scene", "scene4"):
        base = str(tmp_path / name)
        gpu.compile_offline(scenes[name].flatten(), base)
        src = open(base + ".hip").read()
        assert "lol_render_spec" in src and "sd_sphere" in src
        assert os.path.getsize(base + ".co") > 1000
    src4 = open(str(tmp_path / "scene4") + ".hip").read()
    # (the body twice: eval() with the object id, eval_dist() — the distance alone — for the shadow marches)
    assert src4.count("sminf_(") == 2 * 4 and src4.count("sd_sphere(") == 2 * 5 and "best_id = 2u" in src4
    assert src4.count("void eval(") == 1 and src4.count("void eval_dist(") == 1 and src4.count("best = vmin_(") == 2
    # constants are emitted as exact bit patterns: sphere radius 0.5 and smoothness 3
    assert "0x3f000000u" in src4 and "0x40400000u" in src4


def test_code_objects_are_cached_on_disk(tmp_path, scenes):
    """hipRTC output is kept under LOL_GPU_CACHE_DIR: a second process (another rank, the next run) loads it instead of
    compiling; a damaged or foreign file is ignored, never trusted; an empty LOL_GPU_CACHE_DIR switches the cache off."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from loltracer_amd import gpu, scene as S\n"
            "sc = S.Scene.parse_file(%r)\n"
            "print(gpu.compile_offline(sc.flatten(), %r))\n") % (ROOT, os.path.join(ROOT, "tests", "golden", "scenes", "scene2.lol"),
                                                                 str(tmp_path / "k"))

    def run(cache):
        env = dict(os.environ, LOL_GPU_CACHE_DIR=cache)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        return p.stdout

    cache = str(tmp_path / "cache" / "nested")
    assert "disk cache" not in run(cache)
    files = os.listdir(cache)
    assert len(files) == 1 and files[0].endswith(".co")
    first = open(str(tmp_path / "k.co"), "rb").read()
    assert "disk cache" in run(cache)
    assert open(str(tmp_path / "k.co"), "rb").read() == first
    with open(os.path.join(cache, files[0]), "r+b") as f:         # damage the key stored in front of the code object
        f.seek(40)
        f.write(b"XXXX")
    assert "disk cache" not in run(cache)                         # recompiled, and the entry rewritten
    assert "disk cache" in run(cache)
    # a code object whose bytes were changed (the stored checksum no longer matches) is not loaded either
    path = os.path.join(cache, os.listdir(cache)[0])
    with open(path, "r+b") as f:
        f.seek(-9, os.SEEK_END)
        f.write(b"\x00\x01")
    assert "disk cache" not in run(cache)
    assert "disk cache" in run(cache)
    # GPU code is only taken from files and directories that are this user's alone
    os.chmod(os.path.join(cache, os.listdir(cache)[0]), 0o666)
    assert "disk cache" not in run(cache)                         # a file others may write: ignored (and replaced)
    assert "disk cache" in run(cache)
    os.chmod(cache, 0o777)
    assert "disk cache" not in run(cache)                         # a directory others may write: never read from
    os.chmod(cache, 0o700)
    assert "disk cache" in run(cache)
    # the options hipRTC is given are part of the key: another flag set is another kernel
    env_flags = dict(os.environ, LOL_GPU_CACHE_DIR=cache, LOL_GPU_TUNING="1", LOL_GPU_SCHED="default")
    p = subprocess.run([sys.executable, "-c", code], env=env_flags, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "disk cache" not in p.stdout
    assert "disk cache" not in run("")                            # switched off: compiles, writes nothing


def test_tuning_switches_are_fenced_and_recorded(tmp_path):
    """Round-4 review: a switch inherited from some shell must never silently change what is compiled (the one that could end
    parity, LOL_GPU_RTC_FLAGS, is gone since round 6; ten remain).  The library's A/B switches are honoured only beside
    LOL_GPU_TUNING=1: a plain process with switches set compiles the SAME code object as one without (and says on stderr that it
    ignored them); with LOL_GPU_TUNING=1 a switch takes effect, another code object comes out, and lol_gpu_tuning_switches()
    names it."""
    import subprocess
    import sys
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "from loltracer_amd import gpu, scene as S\n"
            "sc = S.Scene.parse_file(%r)\n"
            "gpu.compile_offline(sc.flatten(), sys.argv[1])\n"
            "print(hashlib.sha1(open(sys.argv[1] + '.co', 'rb').read()).hexdigest(), '|' + gpu.tuning_switches() + '|')\n"
            ) % (ROOT, os.path.join(ROOT, "tests", "golden", "scenes", "scene.lol"))
    base = {k: v for k, v in os.environ.items() if not k.startswith("LOL_GPU_")}
    base["LOL_GPU_CACHE_DIR"] = ""

    def run(tag, **extra):
        p = subprocess.run([sys.executable, "-c", code, str(tmp_path / tag)], env=dict(base, **extra), capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        digest, switches = p.stdout.split()[0], p.stdout.split("|")[1]
        return digest, switches, p.stderr

    plain, sw, err = run("plain")
    assert sw == "" and "LOL_GPU_TUNING" not in err
    fenced, sw, err = run("fenced", LOL_GPU_SCHED="default", LOL_GPU_CULL_CLUSTERS="0", LOL_GPU_SPEC_INLINE_MAX="0")
    assert fenced == plain and sw == ""                            # same kernel, nothing in effect ...
    for name in ("LOL_GPU_SCHED", "LOL_GPU_CULL_CLUSTERS", "LOL_GPU_SPEC_INLINE_MAX"):
        assert f"{name} is set but LOL_GPU_TUNING=1 is not: ignored" in err      # ... and said so, once each
        assert err.count(name + " is set") == 1
    tuned, sw, err = run("tuned", LOL_GPU_TUNING="1", LOL_GPU_SCHED="default")
    assert tuned != plain and "LOL_GPU_SCHED=default" in sw and "ignored" not in err
    # a switch of rounds 2 - 5 that no longer exists does nothing, fenced or not
    gone, sw, err = run("gone", LOL_GPU_TUNING="1", LOL_GPU_RTC_FLAGS="-ffp-contract=fast", LOL_GPU_SHADOW_FDIV="1", LOL_GPU_WAVE_SHAPE="8x8x1")
    assert gone == plain and sw == ""
    again, sw, _ = run("again", LOL_GPU_TUNING="1")                # the opt-in alone changes nothing
    assert again == plain and sw == ""
    _, sw, _ = run("yes", LOL_GPU_TUNING="yes", LOL_GPU_SCHED="default")         # only the literal 1 opts in
    assert sw == ""


def test_small_scenes_keep_their_tables_in_lds_and_large_ones_do_not(tmp_path):
    """lights | materials | root_material are staged into LDS per one-wave block while they fit 4 KB (lol_kernel.h,
    TABLES_LDS_MAX_DWORDS) — a field of 250 objects does, as before — and are read from global memory beyond: the generated
    kernel says which (the choice is a compile-time constant of the scene's kernel)."""
    def field(n):
        objs = ", ".join("sphere { material = #1, point = (%d, 0, -5), radius = 1 }" % i for i in range(n))
        return S.Scene.parse_string("materials { { shininess = 1 }, { shininess = 2 } } scene { point_light { point = (0,9,0) }, %s }" % objs)
    for n, want in ((250, "false"), (994, "false"), (996, "true"), (3000, "true")):
        base = str(tmp_path / ("f%d" % n))
        gpu.compile_offline(field(n).flatten(), base)
        src = open(base + ".hip").read()
        assert ("shade_pixel<lol::SpecSdfExact, %s, COUNT>" % want) in src and ("store_pixel<%s>" % want) in src, n
        # (the pipeline with and without its step counters: two kernels for a small scene, the counting one alone for a large one)
        assert ("lol_render_spec_steps" in src) == (field(n).flatten().n_ops <= 256), n
        assert ("stage_common" in src) == (want == "false")


def test_out_of_line_sdf_beyond_the_short_branch_range_is_compiled_correctly_or_refused(tmp_path, monkeypatch):
    """An out-of-line SDF function of more than 128 KB (a field of 600 objects: 211 KB) needs branches beyond s_cbranch's
    reach.  LLVM's AMDGPU backend (ROCm 7.0 and 7.2) then reserves a register pair for them ahead of time and, in a leaf
    function, takes s[30:31] — the return address: the function never returns (found in round 4, once scenes could exceed
    1024 ops: the kernel ran for ever).  compile_spec passes -amdgpu-long-branch-factor=0 (no reservation: a dead pair is
    scavenged at the branch) and refuses any code object that still relaxes a branch through s[30:31]."""
    objs = ", ".join("sphere { material = #1, point = (%d, %d, -5), radius = 0.4 }" % (i % 40, i // 40) for i in range(600))
    sc = S.Scene.parse_string("materials { { shininess = 1 }, { shininess = 2 } } scene { point_light { point = (0,9,0) }, plane { material = #0, y = -1 }, %s }" % objs)
    monkeypatch.setenv("LOL_GPU_CACHE_DIR", "")
    base = str(tmp_path / "ok")
    gpu.compile_offline(sc.flatten(), base, assume_fast=True)
    code = open(base + ".co", "rb").read()
    words = np.frombuffer(code[: len(code) // 4 * 4], dtype="<u4")
    getpc_ra = np.flatnonzero(words == 0xBE9E1C00)                 # s_getpc_b64 s[30:31]
    assert not any(words[i + 1] == 0x801EFF1E and words[i + 5] == 0xBE801D1E for i in getpc_ra if i + 5 < len(words))
    assert len(code) > 300_000                                     # the function really is beyond the short-branch range
    # (Rounds 4 - 5 also compiled this scene under LLVM's own default and saw the pattern appear and the library refuse the code
    # object; with round 6's out-of-line function — one argument less — LLVM's default no longer picks s[30:31] for this scene, so
    # the real instance is gone and the switch that kept the default reachable went with it.  The checker's own test follows.)
    # the checker itself on both code objects: clean with the workaround ...
    check = gpu.gpu_lib().lol_gpu_testing_has_return_clobbering_branch
    assert check(code, len(code)) == 0


def test_the_long_branch_tripwire_reads_instruction_fields_at_any_offset():
    """Round-4 review: the trip-wire matched four literal words and passed by default on a misaligned buffer.  It now recognises
    `s_getpc_b64 s[30:31] ... s_setpc_b64 s[30:31]` by the SOP1 fields — whatever arithmetic sits in between — at every byte
    offset of the buffer, and tells a relaxed branch from a call (s_swappc_b64) and from a plain return."""
    import struct
    check = gpu.gpu_lib().lol_gpu_testing_has_return_clobbering_branch

    def sop1(op, sdst, ssrc0):
        return 0xBE800000 | sdst << 16 | op << 8 | ssrc0

    def sop2(op, sdst, ssrc1, ssrc0):
        return 0x80000000 | op << 23 | sdst << 16 | ssrc1 << 8 | ssrc0

    GETPC, SETPC, SWAPPC, LIT = 28, 29, 30, 0xFF
    S_ADD, S_SUB, S_ADDC, S_SUBB = 0, 1, 4, 5
    assert sop1(GETPC, 30, 0) == 0xBE9E1C00 and sop1(SETPC, 0, 30) == 0xBE801D1E and sop2(S_ADD, 30, LIT, 30) == 0x801EFF1E      # (round 4's literals)
    nop = 0xBF800000
    pad = [nop] * 5

    def has(words, shift=0):
        buf = b"\x7f" * shift + struct.pack("<%dI" % len(words), *words)
        return check(buf, len(buf))

    # the sequence LLVM emits, literal in the second source position (what round 4 matched) ...
    relaxed = [sop1(GETPC, 30, 0), sop2(S_ADD, 30, LIT, 30), 0x00012344, sop2(S_ADDC, 31, LIT, 31), 0, sop1(SETPC, 0, 30)]
    for shift in (0, 1, 2, 3, 4, 7, 64, 1001):                       # ... found wherever the buffer and the text begin
        assert has(pad + relaxed + pad, shift) == 1, shift
    # ... with the literal in the FIRST source position, as a backward branch (sub / subb), with padding in between
    assert has(pad + [sop1(GETPC, 30, 0), sop2(S_ADD, 30, 30, LIT), 0x40, sop2(S_ADDC, 31, 31, LIT), 0, sop1(SETPC, 0, 30)] + pad) == 1
    assert has(pad + [sop1(GETPC, 30, 0), sop2(S_SUB, 30, LIT, 30), 0x99990, sop2(S_SUBB, 31, LIT, 31), 0, nop, nop, sop1(SETPC, 0, 30)] + pad) == 1
    # not the bug: a call through another pair, a plain return, a branch relaxed through a scavenged pair, the two far apart
    call = [sop1(GETPC, 4, 0), sop2(S_ADD, 4, LIT, 4), 0x1000, sop2(S_ADDC, 5, LIT, 5), 0, sop1(SWAPPC, 30, 4)]
    assert has(pad + call + pad + [sop1(SETPC, 0, 30)]) == 0
    assert has(pad + [sop1(SETPC, 0, 30)] + pad) == 0
    assert has(pad + [sop1(GETPC, 34, 0), sop2(S_ADD, 34, LIT, 34), 0x1000, sop2(S_ADDC, 35, LIT, 35), 0, sop1(SETPC, 0, 34)] + pad) == 0
    assert has([sop1(GETPC, 30, 0)] + [nop] * 40 + [sop1(SETPC, 0, 30)]) == 0
    # a call that reuses s[30:31] for the target (getpc, swappc before any setpc) is a call
    assert has(pad + [sop1(GETPC, 30, 0), sop2(S_ADD, 30, LIT, 30), 0x10, sop2(S_ADDC, 31, LIT, 31), 0, sop1(SWAPPC, 30, 30), sop1(SETPC, 0, 30)] + pad) == 0
    assert has([]) == 0 and has([sop1(GETPC, 30, 0)]) == 0
