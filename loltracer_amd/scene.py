"""ctypes mirror of include/lol_scene.h — the `.lol` reader, scene model and flattener.

The work is done by loltracer_amd/lib/liblol_scene.so (plain C,
loltracer_amd/csrc/lol_scene.c); this module only declares the structs and
wraps the entry points so tests and bench.py can drive them.  Names follow the
reference's scene.h / scene-parser.y (scene_parse → Scene.parse_file,
scene_validate_materials → Scene.validate_materials).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

# sanity caps of include/lol_scene.h (a program is as large as its scene; counts beyond these are taken for corruption)
LOL_MAX_OPS = 1 << 20
LOL_MAX_LIGHTS = 1 << 16
LOL_MAX_MATERIALS = 1 << 20
LOL_MAX_STACK = 64

(LOL_OK, LOL_ERR_IO, LOL_ERR_SYNTAX, LOL_ERR_PROPERTY, LOL_ERR_TYPE, LOL_ERR_COMPONENT,
 LOL_ERR_MATERIAL, LOL_ERR_NOMEM, LOL_ERR_UNSUPPORTED) = range(9)

NODE_SPHERE, NODE_BOX, NODE_PLANE, NODE_SMOOTH_UNION = range(4)
OP_SPHERE, OP_RBOX, OP_PLANE, OP_SMIN, OP_SMIN_R, OP_TOP = range(6)


class V3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Material(C.Structure):
    _fields_ = [("shininess", C.c_float), ("diffuse", V3), ("specular", V3), ("ambient", V3)]


class Light(C.Structure):
    _fields_ = [("point", V3), ("diffuse_intensity", V3), ("specular_intensity", V3)]


class Camera(C.Structure):
    _fields_ = [("point", V3), ("direction", V3), ("fov", C.c_float)]


class Node(C.Structure):
    _fields_ = [("type", C.c_int32), ("material", C.c_uint32), ("point", V3), ("radius", C.c_float),
                ("half_extent", V3), ("smoothness", C.c_float), ("a", C.c_int32), ("b", C.c_int32)]


class SceneStruct(C.Structure):
    _fields_ = [("materials", C.POINTER(Material)), ("n_materials", C.c_size_t),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_size_t),
                ("nodes", C.POINTER(Node)), ("n_nodes", C.c_size_t),
                ("roots", C.POINTER(C.c_int32)), ("n_roots", C.c_size_t),
                ("ambient_color", V3), ("camera", Camera)]


class Op(C.Structure):
    _fields_ = [("op", C.c_uint32), ("id", C.c_uint32), ("f", C.c_float * 7), ("_pad", C.c_uint32)]


class Program(C.Structure):
    """lol_program: counts + pointers to the four tables (allocated by lol_scene_flatten, released with the object)."""
    _fields_ = [("n_ops", C.c_uint32), ("n_lights", C.c_uint32), ("n_materials", C.c_uint32),
                ("n_roots", C.c_uint32), ("max_stack", C.c_uint32), ("ambient_color", V3),
                ("ops", C.POINTER(Op)), ("lights", C.POINTER(Light)),
                ("materials", C.POINTER(Material)), ("root_material", C.POINTER(C.c_uint32))]

    def tables(self) -> bytes:
        """Everything the program says, as bytes (two programs are the same scene iff these are equal)."""
        head = bytes(memoryview(self).cast("B")[:C.sizeof(C.c_uint32) * 5 + C.sizeof(V3)])
        parts = [head]
        for ptr, n, t in ((self.ops, self.n_ops, Op), (self.lights, self.n_lights, Light),
                          (self.materials, self.n_materials, Material), (self.root_material, self.n_roots, C.c_uint32)):
            parts.append(C.string_at(ptr, n * C.sizeof(t)) if n else b"")
        return b"".join(parts)

    def free(self):
        if getattr(self, "_owned", False):
            self._owned = False
            host_lib().lol_program_free(C.byref(self))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FrameCamera(C.Structure):
    _fields_ = [("origin", V3), ("dir", V3), ("right", V3), ("up", V3),
                ("width", C.c_float), ("height", C.c_float)]


class SceneError(Exception):
    def __init__(self, status: int, message: str):
        super().__init__(f"[{status}] {message}")
        self.status = status
        self.message = message


_lib = None


def host_lib() -> C.CDLL:
    """liblol_scene.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        path = os.path.join(LIB_DIR, "liblol_scene.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (or make -C loltracer_amd/csrc)")
        lib = C.CDLL(path)
        P = C.POINTER
        lib.lol_scene_parse_file.argtypes = [C.c_char_p, P(P(SceneStruct)), C.c_char_p, C.c_size_t]
        lib.lol_scene_parse_file.restype = C.c_int
        lib.lol_scene_parse_string.argtypes = [C.c_char_p, C.c_size_t, P(P(SceneStruct)), C.c_char_p, C.c_size_t]
        lib.lol_scene_parse_string.restype = C.c_int
        lib.lol_scene_free.argtypes = [P(SceneStruct)]
        lib.lol_scene_free.restype = None
        lib.lol_scene_new.argtypes = []
        lib.lol_scene_new.restype = P(SceneStruct)
        lib.lol_scene_validate_materials.argtypes = [P(SceneStruct)]
        lib.lol_scene_validate_materials.restype = C.c_int
        lib.lol_scene_flatten.argtypes = [P(SceneStruct), P(Program)]
        lib.lol_scene_flatten.restype = C.c_int
        lib.lol_program_free.argtypes = [P(Program)]
        lib.lol_program_free.restype = None
        lib.lol_frame_camera_init.argtypes = [P(FrameCamera), P(Camera), C.c_int, C.c_int]
        lib.lol_frame_camera_init.restype = None
        lib.lol_status_str.argtypes = [C.c_int]
        lib.lol_status_str.restype = C.c_char_p
        _lib = lib
    return _lib


class Scene:
    """Owning handle of a parsed `lol_scene*`."""

    def __init__(self, ptr):
        self._ptr = ptr

    @classmethod
    def parse_file(cls, path: str) -> "Scene":
        lib = host_lib()
        out = C.POINTER(SceneStruct)()
        err = C.create_string_buffer(256)
        st = lib.lol_scene_parse_file(os.fsencode(path), C.byref(out), err, len(err))
        if st != LOL_OK:
            raise SceneError(st, err.value.decode() or lib.lol_status_str(st).decode())
        return cls(out)

    @classmethod
    def parse_string(cls, text) -> "Scene":
        lib = host_lib()
        data = text.encode() if isinstance(text, str) else bytes(text)
        out = C.POINTER(SceneStruct)()
        err = C.create_string_buffer(256)
        st = lib.lol_scene_parse_string(data, len(data), C.byref(out), err, len(err))
        if st != LOL_OK:
            raise SceneError(st, err.value.decode() or lib.lol_status_str(st).decode())
        return cls(out)

    @property
    def ptr(self):
        return self._ptr

    @property
    def c(self) -> SceneStruct:
        return self._ptr.contents

    @property
    def camera(self) -> Camera:
        return self.c.camera

    def validate_materials(self) -> bool:
        return bool(host_lib().lol_scene_validate_materials(self._ptr))

    def flatten(self) -> Program:
        prog = Program()
        st = host_lib().lol_scene_flatten(self._ptr, C.byref(prog))
        if st != LOL_OK:
            raise SceneError(st, host_lib().lol_status_str(st).decode())
        prog._owned = True                  # its tables belong to this object (lol_program_free when it goes)
        return prog

    def frame_camera(self, w: int, h: int, camera: Camera | None = None) -> FrameCamera:
        fc = FrameCamera()
        cam = camera if camera is not None else self.c.camera
        host_lib().lol_frame_camera_init(C.byref(fc), C.byref(cam), w, h)
        return fc

    def nodes(self):
        return [self.c.nodes[i] for i in range(self.c.n_nodes)]

    def roots(self):
        return [self.c.roots[i] for i in range(self.c.n_roots)]

    def materials(self):
        return [self.c.materials[i] for i in range(self.c.n_materials)]

    def lights(self):
        return [self.c.lights[i] for i in range(self.c.n_lights)]

    def close(self):
        if self._ptr:
            host_lib().lol_scene_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
