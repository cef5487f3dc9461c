#!/usr/bin/env python3
"""A numpy model of the specialised kernel's culling on scene4: which wave-evaluations could skip the blob?

Marches the frame in float32 numpy (not bit-exact — statistics only), wave by wave (16x4 patches), through the three loops of
the pipeline (primary march, normal taps, shadow marches with the exact skips), and for every wave-evaluation asks whether
EVERY lane that still cares passes the bounding test — with the ONE enclosing sphere the kernel uses today, and with two or
three spheres (clusters of the blob's primitives, each inflated by the union slack).  CPU only.
    python tools/cull_model.py [--size 480x270]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from loltracer_amd import scene as S

F = np.float32


def smin(a, b, k):
    h = np.clip(F(0.5) + F(0.5) * (b - a) / k, 0, 1)
    return (b + (a - b) * h) - k * h * (1 - h)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="480x272")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    h -= h % 4; w -= w % 16
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    P = sc.flatten()
    ops = [(P.ops[i].op, [float(P.ops[i].f[j]) for j in range(7)]) for i in range(P.n_ops)]
    spheres = [(np.array(f[:3], dtype=F), F(f[3])) for op, f in ops if op == S.OP_SPHERE]
    k = F([f[0] for op, f in ops if op in (S.OP_SMIN, S.OP_SMIN_R)][0])
    plane_y = F([f[0] for op, f in ops if op == S.OP_PLANE][0])

    def blob(p):
        d = [np.sqrt(((p - c) ** 2).sum(-1)) - r for c, r in spheres]
        return smin(smin(d[0], d[1], k), smin(d[4], smin(d[2], d[3], k), k), k)      # scene4's tree (LABNOTES.md §3.1)

    def sdf(p):
        pl = p[..., 1] - plane_y
        b = blob(p)
        return np.minimum(pl, b), np.where(b < pl, 1, 2), pl

    # bounds: one sphere around everything (+ 3 levels of k/4), or clusters of primitives, each + the same slack
    def enclose(idx):
        c = np.mean([spheres[i][0] for i in idx], axis=0)
        for _ in range(50):                                      # crude minimal enclosing sphere by iteration
            far = max(idx, key=lambda i: np.linalg.norm(spheres[i][0] - c) + spheres[i][1])
            d = spheres[far][0] - c
            c = c + 0.05 * d
        R = max(np.linalg.norm(spheres[i][0] - c) + spheres[i][1] for i in idx)
        return c.astype(F), F(R + 3 * k / 4) * F(1.001)
    plans = {"1 sphere": [enclose(range(5))], "2 spheres": [enclose([0, 1, 4]), enclose([2, 3])],
             "3 spheres": [enclose([0, 1]), enclose([4]), enclose([2, 3])]}

    cam = sc.c.camera
    ro = np.array([cam.point.x, cam.point.y, cam.point.z], dtype=F)
    cd = np.array([cam.direction.x, cam.direction.y, cam.direction.z], dtype=F)
    cd /= np.linalg.norm(cd)
    height = F(np.arctan(cam.fov / 2)); width = F(w / h) * height
    right = np.cross(cd, [0, 1, 0]).astype(F); right /= np.linalg.norm(right)
    up = np.cross(right, cd).astype(F)
    ys, xs = np.mgrid[0:h, 0:w]
    vx = ((xs + F(0.5)) / F(w) * 2 - 1).astype(F); vy = (1 - (ys + F(0.5)) / F(h) * 2).astype(F)
    rd = right * (vx * width)[..., None] + up * (vy * height)[..., None] + cd
    rd /= np.linalg.norm(rd, axis=-1, keepdims=True)

    stats = {name: [0, 0] for name in plans}       # [wave-evals that may skip, wave-evals]

    def account(p, pl, care):
        """p [h,w,3], pl = the running minimum before the blob (the plane), care [h,w] bool"""
        cw = care.reshape(h // 4, 4, w // 16, 16)
        any_care = cw.any(axis=(1, 3))
        for name, spheres_b in plans.items():
            ok = np.ones((h, w), dtype=bool)
            for c, R in spheres_b:
                D = np.sqrt(((p - c) ** 2).sum(-1))
                ok &= (D - R > pl * F(1.0003)) & (pl + R > 0)
            okw = (ok | ~care).reshape(h // 4, 4, w // 16, 16).all(axis=(1, 3))
            stats[name][0] += int((okw & any_care).sum())
            stats[name][1] += int(any_care.sum())

    # primary march
    dist = np.zeros((h, w), dtype=F); alive = np.ones((h, w), dtype=bool); hit_id = np.zeros((h, w), dtype=int)
    for i in range(256):
        if not alive.any():
            break
        p = ro + rd * dist[..., None]
        d, idn, pl = sdf(p)
        account(p, pl, alive)
        dist = np.where(alive, dist + d, dist); hit_id = np.where(alive, idn, hit_id)
        alive &= ~((d < F(0.001)) | (dist > 100))
    hit = dist < 100
    p0 = ro + rd * dist[..., None]
    hh = dist / F(100)
    taps = []
    for kx, ky, kz in ((1, -1, -1), (-1, -1, 1), (-1, 1, -1), (1, 1, 1)):
        q = p0 + np.array([kx, ky, kz], dtype=F) * hh[..., None]
        d, _, pl = sdf(q)
        lit_wave = np.repeat(np.repeat(hit.reshape(h // 4, 4, w // 16, 16).any(axis=(1, 3)), 4, 0), 16, 1)
        account(q, pl, lit_wave)
        taps.append(np.array([kx, ky, kz], dtype=F) * d[..., None])
    n = sum(taps); n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-30)
    for li in range(P.n_lights):
        lp = np.array([P.lights[li].point.x, P.lights[li].point.y, P.lights[li].point.z], dtype=F)
        tl = lp - p0; L = np.linalg.norm(tl, axis=-1); ld = tl / L[..., None]
        di = np.clip((n * ld).sum(-1), 0, 1)
        alive = hit & (di > 0)
        o = p0 + ld; res = np.ones((h, w), dtype=F); t = np.zeros((h, w), dtype=F)
        for i in range(128):
            if not alive.any():
                break
            q = o + ld * t[..., None]
            s, _, pl = sdf(q)
            account(q, pl, alive)
            with np.errstate(divide="ignore", invalid="ignore"):
                v = F(50) * s / t
            res = np.where(alive, np.where(res < v, res, v), res)
            t = np.where(alive, t + s, t)
            alive &= ~((res <= 0) | (t > L))
    for name, (skip, total) in stats.items():
        print(f"{name}: {skip} of {total} wave-evaluations may skip the blob = {skip / total:.3f}")


if __name__ == "__main__":
    main()
