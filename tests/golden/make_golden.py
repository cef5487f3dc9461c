#!/usr/bin/env python3
"""Generate the committed golden fixtures.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Outputs (all DATA — inputs and expected outputs, no reference source text):

  ref_primitives.json   inputs → outputs of the REFERENCE's own float.h / vec.h / sdf.h functions,
                        obtained by calling oracle/_ref/liblol_ref.so (those headers compiled where
                        they lie in /root/reference, see oracle/Makefile `ref`).  Bit patterns as hex.
  ref_scenes.json       what the REFERENCE's scene.c builders produce for the four example scenes:
                        the .lol text is walked by the small independent parser below (mirroring the
                        grammar actions of scene-parser.y:99-145) and every block is handed to
                        scene.c's *_from_definition_list through liblol_ref.so; the resulting
                        struct scene is dumped field by field (floats as hex bit patterns).
  ref_sdf_points.json   the scene SDF of the four example scenes at ~2000 points each, COMPOSED from the
                        reference's own compiled pieces: the `struct object` tree built by scene.c is walked
                        (this script, following naive_renderer.c:11-44: q = p - point; sphere / round box /
                        plane; smooth_union = sminf(a, b, k) on the untranslated p; top level: first strict
                        minimum, ids 1-based in file order) and every arithmetic step is a call into
                        liblol_ref.so (v3sub, sdSphere, sdRoundBox, sminf).  Extends the pin from the primitives
                        to rows a4/a5 of SURVEY.md §8 (tests/test_oracle.py checks lol_oracle_sdf against it).
  ref_pixels.json       whole pixels of the four example scenes (a 13 x 9 grid of a 64x36 frame each, + 256x256 probes
                        of scene / scene4), COMPOSED from the reference's compiled pieces exactly like the scene SDF
                        above: this script follows naive_renderer.c:48-236 statement by statement (camera ray, march
                        loop, tetrahedron normal, soft shadow loop, Phong sum, clamp, gamma, 8-bit pack) and every
                        vector / min / max / clamp / smooth-min / distance operation is a call into liblol_ref.so
                        (scalar +, *, / are IEEE binary32 via numpy; atanf and powf are the C library's, the latter
                        through the reference's v3pow).  Recorded per pixel: ray direction, hit distance, hit id,
                        march steps, normal, per-light shadow factor and steps, linear and gamma colour, packed pixel.
                        Pins rows a2, a3, a9 - a13 of SURVEY.md §8 for the oracle (tests/test_oracle.py).
  ref_frames.npz        the same composition for EVERY pixel of the 64x36 frame of each example scene (9216 pixels):
                        arrays <scene>_xrgb, _rgb, _rgb_linear, _hit_dist, _hit_id, _march_steps, _normal, _rd, _shadow,
                        _shadow_steps, _shadow_settled_steps (steps up to the first that left the running factor <= 0) —
                        whole reference-composed frames for the oracle to equal.
  ref_camera_path.json  the camera after every frame of a key script (W A S D Space LCtrl and the four arrows, singly and
                        held together), stepped the way main.c's update_camera does (main.c:70-112) with every vector
                        operation a call into the reference's compiled vec.h (v3cross, v3normalize, v3add, v3scale) —
                        pins the host-side keyboard → camera step of integration/lol_host_input.h.
  oracle_frames.npz     XRGB8888 frames + float RGB of the CPU ORACLE (not of the reference:
                        naive_renderer.c cannot be built here) — regression fixtures that
                        pin the oracle's output across toolchains / on the GPU box.
"""
import ctypes as C
import json
import os
import re
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblol_ref.so")

# enum property / enum components of the reference (scene.h:7-35), as plain numbers
PROPS = ["shininess", "diffuse", "specular", "ambient", "color", "point", "direction", "fov",
         "diffuse_intensity", "specular_intensity", "radius", "material", "point2", "y", "smoothness", "a", "b"]
TYPES = ["ambient", "camera", "point_light", "sphere", "box", "plane", "smooth_union"]


def f2h(x):
    return "%08x" % struct.unpack("<I", struct.pack("<f", x))[0]


def h2f(h):
    return struct.unpack("<f", struct.pack("<I", int(h, 16)))[0]


def load_ref():
    ref = C.CDLL(REF_SO)
    f, f3, vp = C.c_float, C.POINTER(C.c_float), C.c_void_p
    sig = {
        "ref_minf": ([f, f], f), "ref_maxf": ([f, f], f), "ref_clamp": ([f, f, f], f),
        "ref_lerp": ([f, f, f], f), "ref_sminf": ([f, f, f], f),
        "ref_v3sub": ([f3, f3, f3], None), "ref_v3add": ([f3, f3, f3], None), "ref_v3scale": ([f3, f, f3], None),
        "ref_v3mul": ([f3, f3, f3], None),
        "ref_v3dot": ([f3, f3], f), "ref_v3len": ([f3], f), "ref_v3normalize": ([f3, f3], None),
        "ref_v3cross": ([f3, f3, f3], None), "ref_v3clamp": ([f3, f, f, f3], None),
        "ref_v3pow": ([f3, f, f3], None),
        "ref_sd_sphere": ([f3, f], f), "ref_sd_box": ([f3, f3], f), "ref_sd_round_box": ([f3, f3, f], f),
        "ref_deflist_new": ([], vp), "ref_deflist_add_num": ([vp, C.c_int, f], None),
        "ref_deflist_add_list": ([vp, C.c_int, f3, C.c_int], None),
        "ref_deflist_add_id": ([vp, C.c_int, C.c_size_t], None),
        "ref_deflist_add_obj": ([vp, C.c_int, C.c_int, vp], None),
        "ref_scene_new": ([], vp), "ref_scene_free": ([vp], None),
        "ref_scene_add_component": ([vp, C.c_int, vp], None), "ref_scene_add_material": ([vp, vp], None),
        "ref_scene_validate_materials": ([vp], C.c_int), "ref_scene_counts": ([vp, C.c_int], C.c_size_t),
        "ref_scene_camera": ([vp, f3], None), "ref_scene_ambient": ([vp, f3], None),
        "ref_scene_material": ([vp, C.c_size_t, f3], None), "ref_scene_light": ([vp, C.c_size_t, f3], None),
        "ref_scene_object": ([vp, C.c_size_t], vp),
        "ref_object_fields": ([vp, C.POINTER(C.c_int), C.POINTER(C.c_size_t), f3, f3, f3, f3,
                               C.POINTER(vp), C.POINTER(vp)], None),
        "ref_sizeof": ([C.c_int], C.c_size_t),
    }
    for name, (args, res) in sig.items():
        fn = getattr(ref, name)
        fn.argtypes, fn.restype = args, res
    return ref


# ------------------------------------------------------------------ primitives

def gen_primitives(ref):
    rng = np.random.default_rng(20261003)
    special = [0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1.0, -1.0, 0.5, 1e-3, 100.0,
               1e-40, -1e-40, 3.0e38, 1.17549435e-38, 0.001, 50.0, 3.0, 2.2]

    def rf():
        r = rng.random()
        if r < 0.2:
            return float(np.float32(special[rng.integers(len(special))]))
        if r < 0.75:
            return float(np.float32(rng.normal() * 10.0 ** int(rng.integers(-3, 3))))
        return float(np.array([rng.integers(0, 2 ** 32)], dtype=np.uint32).view(np.float32)[0])

    def v3():
        return [rf(), rf(), rf()]

    A3 = C.c_float * 3
    out = {k: [] for k in ("minf", "maxf", "clamp", "lerp", "sminf", "v3dot", "v3len", "v3normalize",
                           "v3cross", "v3clamp", "sd_sphere", "sd_box", "sd_round_box")}
    for _ in range(400):
        a, b, c = rf(), rf(), rf()
        out["minf"].append([f2h(a), f2h(b), f2h(ref.ref_minf(a, b))])
        out["maxf"].append([f2h(a), f2h(b), f2h(ref.ref_maxf(a, b))])
        out["clamp"].append([f2h(a), f2h(b), f2h(c), f2h(ref.ref_clamp(a, b, c))])
        out["lerp"].append([f2h(a), f2h(b), f2h(c), f2h(ref.ref_lerp(a, b, c))])
        out["sminf"].append([f2h(a), f2h(b), f2h(c), f2h(ref.ref_sminf(a, b, c))])
        # realistic smooth-min inputs too: distances of a few units, k = 3 (scene4)
        x, y = float(np.float32(rng.normal() * 4)), float(np.float32(rng.normal() * 4))
        out["sminf"].append([f2h(x), f2h(y), f2h(3.0), f2h(ref.ref_sminf(x, y, 3.0))])
        p, q = v3(), v3()
        P, Q, O3 = A3(*p), A3(*q), A3()
        out["v3dot"].append([[f2h(v) for v in p], [f2h(v) for v in q], f2h(ref.ref_v3dot(P, Q))])
        out["v3len"].append([[f2h(v) for v in p], f2h(ref.ref_v3len(P))])
        ref.ref_v3normalize(P, O3)
        out["v3normalize"].append([[f2h(v) for v in p], [f2h(v) for v in O3]])
        ref.ref_v3cross(P, Q, O3)
        out["v3cross"].append([[f2h(v) for v in p], [f2h(v) for v in q], [f2h(v) for v in O3]])
        ref.ref_v3clamp(P, 0.0, 1.0, O3)
        out["v3clamp"].append([[f2h(v) for v in p], f2h(0.0), f2h(1.0), [f2h(v) for v in O3]])
        out["sd_sphere"].append([[f2h(v) for v in p], f2h(a), f2h(ref.ref_sd_sphere(P, a))])
        out["sd_box"].append([[f2h(v) for v in p], [f2h(v) for v in q], f2h(ref.ref_sd_box(P, Q))])
        out["sd_round_box"].append([[f2h(v) for v in p], [f2h(v) for v in q], f2h(a),
                                    f2h(ref.ref_sd_round_box(P, Q, a))])
    return out


# ---------------------------------------------- independent .lol walker → ref builders

TOKEN_RE = re.compile(
    r"(?P<nl>\n)|(?P<ws>[ \r\t]+)|(?P<num>[-.0-9]+)|(?P<id>#[0-9]+)|"
    r"(?P<kw>materials|scene|ambient|camera|point-light|point_light|sphere|box|plane|smooth_union|smooth-union|"
    r"shininess|diffuse_intensity|diffuse-intensity|diffuse|specular_intensity|specular-intensity|specular|color|"
    r"point2|point|direction|fov|radius|material|smoothness|y|a|b)|(?P<p>[,(){}=])|(?P<junk>.)", re.S)


def tokens(text):
    # Python's alternation is first-match, so longer keywords are listed before their prefixes
    for m in TOKEN_RE.finditer(text):
        k = m.lastgroup
        if k in ("nl", "ws", "junk"):
            continue
        yield k, m.group()


class Walker:
    def __init__(self, ref, text):
        self.ref, self.toks, self.i = ref, list(tokens(text)), 0

    def peek(self):
        return self.toks[self.i] if self.i < len(self.toks) else ("eof", "")

    def take(self, val=None):
        k, v = self.peek()
        assert val is None or v == val, (val, k, v)
        self.i += 1
        return k, v

    @staticmethod
    def norm(name):
        return name.replace("-", "_")

    def deflist(self):
        dl = self.ref.ref_deflist_new()
        while True:
            _, prop = self.take()
            pid = PROPS.index(self.norm(prop))
            self.take("=")
            k, v = self.peek()
            if k == "num":
                self.take()
                self.ref.ref_deflist_add_num(dl, pid, C.c_float(float(np.float32(v))))
            elif k == "id":
                self.take()
                self.ref.ref_deflist_add_id(dl, pid, int(v[1:]))
            elif v == "(":
                self.take()
                nums = []
                while True:
                    nums.append(float(np.float32(self.take()[1])))
                    if self.peek()[1] == ",":
                        self.take()
                        continue
                    break
                self.take(")")
                arr = (C.c_float * len(nums))(*nums)
                self.ref.ref_deflist_add_list(dl, pid, arr, len(nums))
            else:
                _, t = self.take()
                self.take("{")
                inner = self.deflist()
                self.take("}")
                self.ref.ref_deflist_add_obj(dl, pid, TYPES.index(self.norm(t)), inner)
            if self.peek()[1] == ",":
                self.take()
                continue
            return dl

    def run(self):
        ref = self.ref
        sc = ref.ref_scene_new()
        self.take("materials"); self.take("{")
        while True:
            self.take("{")
            ref.ref_scene_add_material(sc, self.deflist())
            self.take("}")
            if self.peek()[1] == ",":
                self.take()
                continue
            break
        self.take("}")
        self.take("scene"); self.take("{")
        while True:
            _, t = self.take()
            self.take("{")
            ref.ref_scene_add_component(sc, TYPES.index(self.norm(t)), self.deflist())
            self.take("}")
            if self.peek()[1] == ",":
                self.take()
                continue
            break
        self.take("}")
        assert self.peek()[0] == "eof"
        return sc


def dump_object(ref, optr):
    t, m = C.c_int(), C.c_size_t()
    pt, half = (C.c_float * 3)(), (C.c_float * 3)()
    rad, smooth = C.c_float(), C.c_float()
    a, b = C.c_void_p(), C.c_void_p()
    ref.ref_object_fields(optr, C.byref(t), C.byref(m), pt, C.cast(C.byref(rad), C.POINTER(C.c_float)), half,
                          C.cast(C.byref(smooth), C.POINTER(C.c_float)), C.byref(a), C.byref(b))
    d = {"type": TYPES[t.value], "material": m.value, "point": [f2h(v) for v in pt], "radius": f2h(rad.value),
         "half_extent": [f2h(v) for v in half], "smoothness": f2h(smooth.value)}
    if a.value:
        d["a"] = dump_object(ref, a)
        d["b"] = dump_object(ref, b)
    return d


def dump_scene(ref, sc):
    cam, amb = (C.c_float * 7)(), (C.c_float * 3)()
    ref.ref_scene_camera(sc, cam)
    ref.ref_scene_ambient(sc, amb)
    out = {"camera": [f2h(v) for v in cam], "ambient": [f2h(v) for v in amb], "materials": [], "lights": [],
           "objects": [], "valid_materials": int(ref.ref_scene_validate_materials(sc))}
    for i in range(ref.ref_scene_counts(sc, 0)):
        m = (C.c_float * 10)()
        ref.ref_scene_material(sc, i, m)
        out["materials"].append([f2h(v) for v in m])
    for i in range(ref.ref_scene_counts(sc, 1)):
        l = (C.c_float * 9)()
        ref.ref_scene_light(sc, i, l)
        out["lights"].append([f2h(v) for v in l])
    for i in range(ref.ref_scene_counts(sc, 2)):
        out["objects"].append(dump_object(ref, ref.ref_scene_object(sc, i)))
    return out


# ---------------------------------------------- composed scene SDF from the reference's pieces

def ref_object_dist(ref, optr, p3):
    """get_obj_dist (naive_renderer.c:11-28) over the reference's own struct object, arithmetic by liblol_ref.so."""
    t, m = C.c_int(), C.c_size_t()
    pt, half = (C.c_float * 3)(), (C.c_float * 3)()
    rad, smooth = C.c_float(), C.c_float()
    a, b = C.c_void_p(), C.c_void_p()
    ref.ref_object_fields(optr, C.byref(t), C.byref(m), pt, C.cast(C.byref(rad), C.POINTER(C.c_float)), half,
                          C.cast(C.byref(smooth), C.POINTER(C.c_float)), C.byref(a), C.byref(b))
    kind = TYPES[t.value]
    if kind == "smooth_union":                      # children get the untranslated p; a before b
        da = ref_object_dist(ref, a, p3)
        db = ref_object_dist(ref, b, p3)
        return ref.ref_sminf(da, db, smooth.value)
    q = (C.c_float * 3)()
    ref.ref_v3sub(p3, pt, q)
    if kind == "sphere":
        return ref.ref_sd_sphere(q, rad.value)
    if kind == "box":
        return ref.ref_sd_round_box(q, half, rad.value)
    assert kind == "plane", kind
    return q[1]


def ref_scene_sdf(ref, sc, p):
    """sdf() (naive_renderer.c:31-44): {+inf, 0}, strict '<' over the top-level objects, ids 1-based."""
    p3 = (C.c_float * 3)(*p)
    best, best_id = float("inf"), 0
    for i in range(ref.ref_scene_counts(sc, 2)):
        d = ref_object_dist(ref, ref.ref_scene_object(sc, i), p3)
        if d < best:
            best, best_id = d, i + 1
    return best, best_id


class RefPipeline:
    """naive_renderer.c:48-236 for one pixel, every vector / float.h / sdf.h operation done by liblol_ref.so."""

    def __init__(self, ref, sc, dump):
        self.ref, self.sc, self.dump = ref, sc, dump
        self.A3 = C.c_float * 3
        self.libm = C.CDLL("libm.so.6")
        self.libm.atanf.restype, self.libm.atanf.argtypes = C.c_float, [C.c_float]
        self.cam = [h2f(v) for v in dump["camera"]]
        self.lights = [[h2f(v) for v in l] for l in dump["lights"]]
        self.materials = [[h2f(v) for v in m] for m in dump["materials"]]
        self.obj_material = [o["material"] for o in dump["objects"]]
        self.ambient = [h2f(v) for v in dump["ambient"]]

    # --- vec.h through the harness
    def v2(self, fn, a, b):
        o = self.A3()
        fn(self.A3(*a), self.A3(*b), o)
        return list(o)

    def add(self, a, b): return self.v2(self.ref.ref_v3add, a, b)
    def sub(self, a, b): return self.v2(self.ref.ref_v3sub, a, b)
    def mul(self, a, b): return self.v2(self.ref.ref_v3mul, a, b)
    def cross(self, a, b): return self.v2(self.ref.ref_v3cross, a, b)
    def dot(self, a, b): return self.ref.ref_v3dot(self.A3(*a), self.A3(*b))
    def length(self, a): return self.ref.ref_v3len(self.A3(*a))

    def scale(self, a, k):
        o = self.A3()
        self.ref.ref_v3scale(self.A3(*a), C.c_float(k), o)
        return list(o)

    def normalize(self, a):
        o = self.A3()
        self.ref.ref_v3normalize(self.A3(*a), o)
        return list(o)

    def powf(self, x, y):                               # v3pow = 3 x powf (vec.h:66-67)
        o = self.A3()
        self.ref.ref_v3pow(self.A3(x, x, x), C.c_float(y), o)
        return o[0]

    def sdf(self, p):
        return ref_scene_sdf(self.ref, self.sc, p)

    # --- naive_renderer.c
    def camera_ray(self, vx, vy, aspect):              # :179-193
        f32 = np.float32
        d = self.cam[3:6]
        half_fov = f32(self.cam[6]) / f32(2.0)
        height = f32(self.libm.atanf(C.c_float(float(half_fov))))
        width = f32(aspect) * height
        right = self.normalize(self.cross(d, [0.0, 1.0, 0.0]))
        up = self.cross(right, d)
        r = self.add(self.scale(right, float(f32(vx) * width)), self.scale(up, float(f32(vy) * height)))
        return self.normalize(self.add(r, d))

    def intersection(self, ro, rd):                    # :48-69
        f32 = np.float32
        dist, oid, steps = f32(0.0), 0, 0
        for _ in range(256):
            p = self.add(ro, self.scale(rd, float(dist)))
            sd, sid = self.sdf(p)
            dist = dist + f32(sd)
            oid = sid
            steps += 1
            if f32(sd) < f32(0.001) or dist > f32(100.0):
                break
        if dist >= f32(100.0):
            oid = 0
        return float(dist), oid, steps

    def normal(self, p, dist):                         # :114-125
        h = float(np.float32(dist) / np.float32(100.0))
        ks = ([1.0, -1.0, -1.0], [-1.0, -1.0, 1.0], [-1.0, 1.0, -1.0], [1.0, 1.0, 1.0])
        ps = [self.scale(k, self.sdf(self.add(p, self.scale(k, h)))[0]) for k in ks]
        return self.normalize(self.add(ps[0], self.add(ps[1], self.add(ps[2], ps[3]))))

    def in_shadow(self, light_point, p):               # :73-100
        f32 = np.float32
        light_dist = self.length(self.sub(light_point, p))
        rd = self.normalize(self.sub(light_point, p))
        ro = self.add(p, rd)
        res, dist, steps = f32(1.0), f32(0.0), 0
        settled = 0       # bookkeeping beside the reference's loop: the step (counted from 1) that first left res <= 0
        with np.errstate(all="ignore"):
            for _ in range(128):
                sd = f32(self.sdf(self.add(ro, self.scale(rd, float(dist))))[0])
                res = f32(self.ref.ref_minf(C.c_float(float(res)), C.c_float(float(f32(50.0) * sd / dist))))
                dist = dist + sd
                steps += 1
                if not settled and res <= f32(0.0):
                    settled = steps
                if res < f32(-1.0) or dist > f32(light_dist):
                    break
        shadow = self.ref.ref_maxf(C.c_float(float(res)), C.c_float(0.0))
        # what the renderer's early exit rests on (LABNOTES.md §3.7), here with the reference's own minf / maxf: a factor that was
        # <= 0 once comes out 0
        assert not settled or shadow == 0.0, "a settled shadow factor came back"
        self.last_settled_steps = settled or steps
        return shadow, steps

    def light(self, p, n, oid, rec):                   # :129-175
        f32 = np.float32
        mat = self.materials[self.obj_material[oid - 1] if oid else 0]
        shininess, m_diff, m_spec, m_amb = mat[0], mat[1:4], mat[4:7], mat[7:10]
        total = [0.0, 0.0, 0.0]
        cam_pos = self.cam[:3]
        for l in self.lights:
            shadow, sh_steps = self.in_shadow(l[:3], p)
            rec["shadow"].append(f2h(shadow)); rec["shadow_steps"].append(sh_steps)
            rec["shadow_settled_steps"].append(self.last_settled_steps)
            light_dir = self.normalize(self.sub(l[:3], p))
            refl = self.sub(self.scale(n, float(f32(2.0) * f32(self.dot(light_dir, n)))), light_dir)
            camera_dir = self.normalize(self.sub(cam_pos, p))
            di = self.ref.ref_clamp(C.c_float(self.dot(n, light_dir)), C.c_float(0.0), C.c_float(1.0))
            ld = self.mul(self.scale(l[3:6], float(f32(shadow) * f32(di))), m_diff)
            total = self.add(total, ld)
            si = float(f32(di) * f32(self.powf(self.ref.ref_clamp(C.c_float(self.dot(refl, camera_dir)), C.c_float(0.0), C.c_float(1.0)), shininess)))
            ls = self.mul(self.scale(l[6:9], float(f32(shadow) * f32(si))), m_spec)
            total = self.add(total, ls)
        total = self.add(total, self.mul(self.ambient, m_amb))
        o = self.A3()
        self.ref.ref_v3clamp(self.A3(*total), C.c_float(0.0), C.c_float(1.0), o)
        return list(o)

    def pixel(self, x, y, w, h):                       # :207-236, renderer.h:17-22
        f32 = np.float32
        fw, fh = f32(w), f32(h)
        aspect = fw / fh
        vx = (f32(x) + f32(0.5)) / fw * f32(2.0) - f32(1.0)
        vy = f32(1.0) - (f32(y) + f32(0.5)) / fh * f32(2.0)
        ro = self.cam[:3]
        rd = self.camera_ray(vx, vy, aspect)
        dist, oid, steps = self.intersection(ro, rd)
        p = self.add(ro, self.scale(rd, dist))
        n = self.normal(p, dist)
        rec = {"x": x, "y": y, "rd": [f2h(v) for v in rd], "hit_dist": f2h(dist), "hit_id": oid, "march_steps": steps,
               "normal": [f2h(v) for v in n], "shadow": [], "shadow_steps": [], "shadow_settled_steps": []}
        lin = self.light(p, n, oid, rec)
        o = self.A3()
        self.ref.ref_v3pow(self.A3(*lin), C.c_float(float(f32(1.0) / f32(2.2))), o)
        rgb = list(o)
        ch = [int(f32(c) * f32(255.0)) & 0xFF for c in rgb]          # Uint8 r = colorf.x * 255
        rec.update(rgb_linear=[f2h(v) for v in lin], rgb=[f2h(v) for v in rgb], xrgb=ch[0] << 16 | ch[1] << 8 | ch[2])
        return rec


def gen_sdf_points(ref, sc, cam, seed):
    rng = np.random.default_rng(seed)
    pts = []
    for _ in range(1200):                                         # the volume the example scenes live in
        pts.append(rng.uniform([-16, -3, -22], [16, 12, 6]))
    o, d = np.array(cam[:3], dtype=np.float64), np.array(cam[3:6], dtype=np.float64)
    for _ in range(700):                                          # along view rays, where the renderer samples
        r = d / np.linalg.norm(d) + rng.normal(size=3) * 0.8
        pts.append(o + r / np.linalg.norm(r) * rng.uniform(0, 40))
    for v in ([0, 0, 0], [0, -1, 0], [0, 1, -6], [1e-30, -1, 3e-39], [1e6, 2e6, -3e6], [-0.0, -0.0, -0.0],
              [float("inf"), 0, 0], [float("nan"), 1, 1], [3e38, 3e38, 3e38]):
        pts.append(np.array(v, dtype=np.float64))
    out = []
    for p in pts:
        pf = [float(np.float32(v)) for v in p]
        dist, oid = ref_scene_sdf(ref, sc, pf)
        out.append([[f2h(v) for v in pf], f2h(dist), oid])
    return out


# ---------------------------------------------- keyboard → camera, stepped with the reference's vec.h

CAMERA_SCRIPT = ("W,W,W,A,A,S,D,D,_,_,c,^,^,v,<,<,<,>,W^,WA<,SD>v,_c,WASD_c^v<>,.,.,w<,w<,w<,w<,w<,w<,w<,w<,^,^,^,^,^,^,^,^,^,^,^,^,"
                 "D>,D>,D>,D>,Sv,Sv,Sv,A,W").split(",")


def gen_camera_path(ref, cam7):
    A3 = C.c_float * 3
    f32 = lambda x: float(np.float32(x))

    def call2(fn, a, b):
        o = A3()
        fn(A3(*a), A3(*b), o)
        return list(o)

    def scale(a, k):
        o = A3()
        ref.ref_v3scale(A3(*a), C.c_float(k), o)
        return list(o)

    def normalize(a):
        o = A3()
        ref.ref_v3normalize(A3(*a), o)
        return list(o)

    point, direction = [f32(v) for v in cam7[:3]], [f32(v) for v in cam7[3:6]]
    out = []
    for keys in CAMERA_SCRIPT:
        right = normalize(call2(ref.ref_v3cross, direction, [0.0, 1.0, 0.0]))          # main.c:73-75
        up = normalize(call2(ref.ref_v3cross, right, direction))
        k = set(keys.upper()) | ({"v"} if "v" in keys or "V" in keys else set()) | ({"c"} if "c" in keys or "C" in keys else set())
        if "W" in k: point = call2(ref.ref_v3add, point, scale(direction, f32(0.1)))
        if "A" in k: point = call2(ref.ref_v3add, point, scale(right, f32(-0.1)))
        if "S" in k: point = call2(ref.ref_v3add, point, scale(direction, f32(-0.1)))
        if "D" in k: point = call2(ref.ref_v3add, point, scale(right, f32(0.1)))
        if "_" in k: point[1] = f32(np.float32(point[1]) + np.float32(0.1))
        if "c" in k: point[1] = f32(np.float32(point[1]) - np.float32(0.1))
        if "^" in k: direction = normalize(call2(ref.ref_v3add, direction, scale(up, f32(0.1))))
        if "v" in k: direction = normalize(call2(ref.ref_v3add, direction, scale(up, f32(-0.1))))
        if "<" in k: direction = normalize(call2(ref.ref_v3add, direction, scale(right, f32(-0.1))))
        if ">" in k: direction = normalize(call2(ref.ref_v3add, direction, scale(right, f32(0.1))))
        out.append([f2h(v) for v in point + direction])
    return out


def main():
    if not os.path.exists(REF_SO):
        raise SystemExit("oracle/_ref/liblol_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    ref = load_ref()

    with open(os.path.join(HERE, "ref_primitives.json"), "w") as f:
        json.dump({"source": "reference float.h/vec.h/sdf.h via oracle/_ref/liblol_ref.so", "vectors": gen_primitives(ref)},
                  f, separators=(",", ":"))

    scenes, sdf_points, pixels, frames = {}, {}, {}, {}
    for k, name in enumerate(("scene", "scene2", "scene3", "scene4")):
        text = open(os.path.join(HERE, "scenes", name + ".lol")).read()
        sc = Walker(ref, text).run()
        scenes[name] = dump_scene(ref, sc)
        sdf_points[name] = gen_sdf_points(ref, sc, [h2f(v) for v in scenes[name]["camera"]], 20261004 + k)
        pipe = RefPipeline(ref, sc, scenes[name])
        grid = [(x, y) for y in range(2, 36, 4) for x in range(2, 64, 5)]
        pixels[name] = {"64x36": [pipe.pixel(x, y, 64, 36) for x, y in grid]}
        W, H = 64, 36
        full = [pipe.pixel(x, y, W, H) for y in range(H) for x in range(W)]
        u = lambda hs: np.array([int(v, 16) for v in hs], dtype=np.uint32)
        nl = len(scenes[name]["lights"])
        frames[f"{name}_xrgb"] = np.array([q["xrgb"] for q in full], dtype=np.uint32).reshape(H, W)
        frames[f"{name}_hit_id"] = np.array([q["hit_id"] for q in full], dtype=np.uint32).reshape(H, W)
        frames[f"{name}_march_steps"] = np.array([q["march_steps"] for q in full], dtype=np.uint32).reshape(H, W)
        frames[f"{name}_hit_dist"] = u([q["hit_dist"] for q in full]).view(np.float32).reshape(H, W)
        for key in ("rgb", "rgb_linear", "normal", "rd"):
            frames[f"{name}_{key}"] = u([v for q in full for v in q[key]]).view(np.float32).reshape(H, W, 3)
        frames[f"{name}_shadow"] = u([v for q in full for v in q["shadow"]]).view(np.float32).reshape(H, W, nl)
        frames[f"{name}_shadow_steps"] = np.array([v for q in full for v in q["shadow_steps"]], dtype=np.uint32).reshape(H, W, nl)
        frames[f"{name}_shadow_settled_steps"] = np.array([v for q in full for v in q["shadow_settled_steps"]], dtype=np.uint32).reshape(H, W, nl)
        if name in ("scene", "scene4"):
            pixels[name]["256x256"] = [pipe.pixel(x, y, 256, 256) for x, y in ((128, 128), (0, 0), (255, 255), (40, 200), (200, 60), (77, 131))]
        ref.ref_scene_free(sc)
    np.savez_compressed(os.path.join(HERE, "ref_frames.npz"), **frames)
    with open(os.path.join(HERE, "ref_pixels.json"), "w") as f:
        json.dump({"source": "naive_renderer.c:48-236 followed statement by statement in make_golden.py (RefPipeline), every vector / "
                             "float.h / sdf.h operation executed by the reference's compiled code via oracle/_ref/liblol_ref.so",
                   "format": "floats as binary32 hex; xrgb = r<<16|g<<8|b with Uint8 channel = c*255 truncated (renderer.h:17-22)",
                   "pixels": pixels}, f, separators=(",", ":"))
    with open(os.path.join(HERE, "ref_camera_path.json"), "w") as f:
        json.dump({"source": "main.c:70-112 stepped with the reference's vec.h functions via oracle/_ref/liblol_ref.so "
                             "(make_golden.py gen_camera_path); start = scene4.lol's camera",
                   "keys": "W A S D, _ = Space, c = LCtrl, ^ v < > = arrows; one entry per frame",
                   "script": CAMERA_SCRIPT, "format": "[px,py,pz,dx,dy,dz] after each frame, binary32 hex",
                   "path": gen_camera_path(ref, [h2f(v) for v in scenes["scene4"]["camera"]])}, f, indent=0)
    with open(os.path.join(HERE, "ref_sdf_points.json"), "w") as f:
        json.dump({"source": "scene SDF composed from the reference's scene.c object tree + float.h/vec.h/sdf.h functions "
                             "via oracle/_ref/liblol_ref.so (walk: make_golden.py ref_scene_sdf, after naive_renderer.c:11-44)",
                   "format": "[[px,py,pz], dist, id] — floats as binary32 hex", "points": sdf_points},
                  f, separators=(",", ":"))
    sizes = dict(zip(["material", "light", "object", "camera", "scene", "vector"], [ref.ref_sizeof(i) for i in range(6)]))
    with open(os.path.join(HERE, "ref_scenes.json"), "w") as f:
        json.dump({"source": "reference scene.c builders via oracle/_ref/liblol_ref.so", "abi_sizes": sizes,
                   "scenes": scenes}, f, indent=1)

    import oracle_lib as O
    from loltracer_amd import scene as S
    frames = {}
    for name, w, h in (("scene", 64, 36), ("scene2", 64, 36), ("scene3", 64, 36), ("scene4", 64, 36),
                       ("scene", 256, 256), ("scene4", 256, 256)):
        sc = S.Scene.parse_file(os.path.join(HERE, "scenes", name + ".lol"))
        x, rgb, _ = O.render(sc, w, h, threads=8, want_rgb=True)
        frames[f"{name}_{w}x{h}_xrgb"] = x
        if w == 64:
            frames[f"{name}_{w}x{h}_rgb"] = rgb
    np.savez_compressed(os.path.join(HERE, "oracle_frames.npz"), **frames)
    print("wrote ref_primitives.json, ref_scenes.json, ref_sdf_points.json, ref_pixels.json, ref_frames.npz, ref_camera_path.json, oracle_frames.npz")


if __name__ == "__main__":
    main()
