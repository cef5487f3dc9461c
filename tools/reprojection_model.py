#!/usr/bin/env python3
"""Could a MOVING camera's pixels be dealt to waves by the previous frame's costs RE-PROJECTED into the new view?
Same-pixel stale costs are useless (tests/tools/stale_dealing_model.py: -0.5 ... +1.4 % at 1.4 ... 5.6 degrees a frame): what makes
a pixel dear — a shadow ray grazing a surface — moves by more than a pixel.  But the dear part of a pixel's cost, its shadow
marches, belongs to the SURFACE POINT it sees, not to the pixel: the same point seen from the next camera costs the same shadow
steps.  Model, on the device's own per-pixel counters (lol_gpu_debug: hit distance, hit id, march | shadow steps — equal to the
oracle's, tests/test_gpu_parity.py): frames of the orbit `stride` apart at 3840x2160; every pixel of frame i - stride is carried
to where its hit point (or, for an escaped ray, its direction) lands in frame i's camera, holes take the same pixel's stale cost;
the pixels of frame i are then dealt to the sixteen waves of their 64x16 region by that prediction, and the wave-evaluations
    sum over waves of [max march steps + 4 (any lane hit) + max shadow steps (both lights summed)]
are counted with frame i's TRUE per-pixel counts — against 16x4 rectangles, the frame's own costs (the repeated view) and the
same pixel's stale cost.   python tools/reprojection_model.py [--frames 0,64,128,192] [--strides 1,2,4]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from loltracer_amd import gpu, scene as S  # noqa: E402

RW, RH = 64, 16
W, H = 3840, 2160


def render(r, sc, cam):
    dev = torch.device("cuda:0")
    frame = torch.zeros((H, W), dtype=torch.int32, device=dev)
    dist = torch.zeros((H, W), dtype=torch.float32, device=dev)
    hid = torch.zeros((H, W), dtype=torch.int32, device=dev)
    steps = torch.zeros((H, W), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    fc = sc.frame_camera(W, H, cam)
    r.render_into(frame.data_ptr(), W, H, 256, debug=gpu.Debug(None, dist.data_ptr(), hid.data_ptr(), steps.data_ptr()), frame_camera=fc)
    r.sync()
    st = steps.cpu().numpy().view(np.uint32)
    return dict(fc=fc, dist=dist.cpu().numpy(), hit=hid.cpu().numpy() != 0, march=(st & 0xFFFF).astype(np.int64), shadow=(st >> 16).astype(np.int64))


def basis(fc):
    f = lambda a: np.array([a.x, a.y, a.z], dtype=np.float64)
    return f(fc.origin), f(fc.dir), f(fc.right), f(fc.up), float(fc.width), float(fc.height)


def rays(fc):
    ro, d, right, up, cw, ch = basis(fc)
    x = (np.arange(W, dtype=np.float64) + .5) / W * 2 - 1
    y = 1 - (np.arange(H, dtype=np.float64) + .5) / H * 2
    rd = right[None, None, :] * (x[None, :, None] * cw) + up[None, None, :] * (y[:, None, None] * ch) + d[None, None, :]
    return ro, rd / np.linalg.norm(rd, axis=2, keepdims=True)


def reproject(prev, cur_fc):
    """predicted cost per pixel of the current frame: the previous frame's costs carried along their hit points"""
    ro, rd = rays(prev["fc"])
    cost = prev["march"] + prev["shadow"]
    ro2, d2, right2, up2, cw2, ch2 = basis(cur_fc)
    pts = ro[None, None, :] + rd * prev["dist"][..., None].astype(np.float64)
    v = np.where(prev["hit"][..., None], pts - ro2[None, None, :], rd)              # escaped rays: a point at infinity
    a = (v @ right2) / (right2 @ right2)
    b = (v @ up2) / (up2 @ up2)
    c = v @ d2
    ok = c > 1e-6
    vx = np.where(ok, a / np.where(ok, c, 1) / cw2, 9)
    vy = np.where(ok, b / np.where(ok, c, 1) / ch2, 9)
    px = np.rint((vx + 1) / 2 * W - .5).astype(np.int64)
    py = np.rint((1 - vy) / 2 * H - .5).astype(np.int64)
    inside = ok & (px >= 0) & (px < W) & (py >= 0) & (py < H)
    pred = np.full((H, W), -1, dtype=np.int64)
    flat = py[inside] * W + px[inside]
    np.maximum.at(pred.reshape(-1), flat, cost[inside])                             # (several sources on one pixel: the dearest)
    holes = pred < 0
    pred[holes] = cost[holes]                                                       # nothing landed here: the same pixel's stale cost
    return pred, float(holes.mean())


def cut(x):
    hh = (H // RH) * RH
    return x[:hh].reshape(hh // RH, RH, W // RW, RW).swapaxes(1, 2).reshape(-1, RH * RW)


def wave_evals(march, hit, shadow, key=None):
    m, hh, s = cut(march), cut(hit), cut(shadow)
    n_reg, nw = m.shape[0], RW * RH // 64
    if key is None:                                                                 # 16x4 rectangles
        f = lambda x: x.reshape(n_reg, RH // 4, 4, RW // 16, 16).swapaxes(2, 3).reshape(n_reg * nw, 64)
    else:
        idx = np.argsort(cut(key), axis=1, kind="stable")
        f = lambda x: np.take_along_axis(x, idx, axis=1).reshape(n_reg * nw, 64)
    return int((f(m).max(axis=1) + 4 * f(hh).any(axis=1) + f(s).max(axis=1)).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="0,64,128,192")
    ap.add_argument("--strides", default="1,2,4")
    a = ap.parse_args()
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    r = gpu.Renderer(0)
    r.prepare(sc)
    r.set_miss_skip(True)
    out = dict(scene="scene4.lol", size=f"{W}x{H}", region="64x16", degrees_per_orbit_frame=360.0 / 256)
    tot = {}
    for f in (int(v) for v in a.frames.split(",")):
        cur = render(r, sc, bench.orbit_camera(f, 256))
        own = cur["march"] + cur["shadow"]
        rect = wave_evals(cur["march"], cur["hit"], cur["shadow"])
        best = wave_evals(cur["march"], cur["hit"], cur["shadow"], own)
        need = int((cur["march"] + 4 * cur["hit"] + cur["shadow"])[: (H // RH) * RH].sum())
        for s in (int(v) for v in a.strides.split(",")):
            prev = render(r, sc, bench.orbit_camera((f - s) % 256, 256))
            pred, holes = reproject(prev, cur["fc"])
            T = tot.setdefault(s, dict(rect=0, own=0, stale=0, reproj=0, need=0, holes=[]))
            T["rect"] += rect; T["own"] += best; T["need"] += need
            T["stale"] += wave_evals(cur["march"], cur["hit"], cur["shadow"], prev["march"] + prev["shadow"])
            T["reproj"] += wave_evals(cur["march"], cur["hit"], cur["shadow"], pred)
            T["holes"].append(holes)
            print(f"frame {f} stride {s} done", file=sys.stderr, flush=True)
    for s, T in tot.items():
        out[f"stride_{s}"] = dict(degrees_per_frame=round(s * 360.0 / 256, 2), holes=round(float(np.mean(T["holes"])), 4),
                                  lane_efficiency={k: round(T["need"] / (64 * T[k]), 4) for k in ("rect", "own", "stale", "reproj")},
                                  fewer_wave_evaluations_than_rectangles={k: round(1 - T[k] / T["rect"], 4) for k in ("own", "stale", "reproj")})
    print(json.dumps(out, indent=1))
    r.close()


if __name__ == "__main__":
    main()
