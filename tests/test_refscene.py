"""integration/lol_refscene.c: the reference's `struct scene` pointer graph → lol_scene.

Runs where oracle/_ref/liblol_ref.so exists (built from /root/reference by `make -C oracle ref`;
the prebuilt .so travels to the GPU box).  The reference scene is built by the reference's own
scene.c builders (driven by the independent walker of tests/golden/make_golden.py), converted by
lol_scene_from_reference(), flattened, and compared with the program the build's own .lol reader
produces from the same file.
"""
import ctypes as C
import importlib.util
import os

import pytest

from loltracer_amd import scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblol_ref.so")

pytestmark = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")


def _mk():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name", ["scene", "scene2", "scene3", "scene4"])
def test_reference_scene_converts_to_the_same_program(name):
    mk = _mk()
    ref = mk.load_ref()
    ref.lol_scene_from_reference.argtypes = [C.c_void_p]
    ref.lol_scene_from_reference.restype = C.POINTER(S.SceneStruct)
    ref.lol_scene_flatten.argtypes = [C.POINTER(S.SceneStruct), C.POINTER(S.Program)]
    ref.lol_scene_flatten.restype = C.c_int
    ref.lol_scene_free.argtypes = [C.POINTER(S.SceneStruct)]
    text = open(os.path.join(ROOT, "tests", "golden", "scenes", name + ".lol")).read()
    rsc = mk.Walker(ref, text).run()
    conv = ref.lol_scene_from_reference(rsc)
    assert conv
    prog = S.Program()
    assert ref.lol_scene_flatten(conv, C.byref(prog)) == S.LOL_OK
    mine = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", name + ".lol"))
    want = mine.flatten()
    assert prog.tables() == want.tables()
    ref.lol_program_free.argtypes = [C.POINTER(S.Program)]
    ref.lol_program_free(C.byref(prog))
    c, m = conv.contents.camera, mine.c.camera
    assert bytes(c) == bytes(m)
    ref.lol_scene_free(conv)
    ref.ref_scene_free(rsc)
