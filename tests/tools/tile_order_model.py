#!/usr/bin/env python3
"""A NEW view has no costs of a frame before it.  Could the ORDER in which its tiles are handed out be predicted inside the
frame itself?  (Round-5 review, next #2: "render one wave slot of every 64x16 region first — real pixels, nothing wasted —
take its run time as the region's cost, hand the other 15/16 out longest first".)

CPU model (test infrastructure; nothing here runs on the product path).  Per 16x4 wave of a frame, what it executes — the
maximum over its lanes of the march, 4 normal taps if any lane hit, per light the maximum over its lanes of the (settled)
shadow march, in SDF evaluations — from the oracle's per-pixel, per-light step counts, which the device's counters equal
(tests/test_gpu_parity.py).  The machine: 1024 SIMDs with 8 wave slots each; a SIMD issues one unit of work per unit of time,
shared equally by its resident waves, a single wave getting at most a quarter of it (the kernel is issue-bound from four waves
per SIMD: DESIGN.md); a freed slot is filled at once by the next wave of the launch's order.  A launch ends when its last wave
does; a second launch on the same stream starts then (plus a fixed gap).  Strategies:

  rows / cols      the fixed orders a new view gets today (the better of the two)
  lpt_exact        every wave's own cost known beforehand, longest first — what a REPEATED view gets (order only, no dealing)
  two_launches     the review's proposal: launch 1 = one wave of every 64x16 region (1/16 of the frame), launch 2 = the other
                   15 waves of every region, regions longest first by what their sample wave cost
  two_launches_free the same with launch 1 costing nothing beyond its share of the work (an upper bound on any cleverer
                   overlap of the sampling with other work: what the PREDICTION is worth if it were free)
  region_exact     regions longest first by their TRUE mean cost (the limit of any region-level prediction)
  one_launch_*     the most favourable form of sampling inside ONE launch: samples first, a fixed-order share `alpha` of the
                   frame behind them (what keeps the machine busy while the samples finish and are sorted — assumed free and
                   always in time), the rest longest first by the region's own sample / by the maximum over the region and its
                   eight neighbours

Calibration: on C3 the model's lpt_exact over the better fixed order must be near the +8 % measured on the device for
longest-first alone (DESIGN.md / LABNOTES.md, round 4: 7750 -> 8300 ... 8340 Mpixels/s).

Usage: python tests/tools/tile_order_model.py [--size 3840x2160] [--frames c3,0,64,128,192] [--procs 8]"""
import argparse
import heapq
import json
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_SIMD, SLOTS, SOLO_SHARE = 1024, 8, 0.25
LAUNCH_GAP = 0.0          # in work units; set from --gap-us and the frame's scale


def _strip(args):
    scene_path, w, h, y0, y1, cam_spec = args
    import oracle_lib as O
    from loltracer_amd import scene as S
    sc = S.Scene.parse_file(scene_path)
    cam = None
    if cam_spec != "c3":
        import bench
        cam = bench.orbit_camera(int(cam_spec), 256)
    _, _, st = O.render_rows(sc, w, h, y0, y1, 256, camera=cam, want_steps=True)
    st = st[y0:y1].astype(np.int32)
    hit = st[..., 2] != 0
    dark = st[..., 3]
    march = st[..., 0]
    sh = []
    for li in range(4):
        s = st[..., 8 + li].copy()
        s[((dark >> li) & 1) == 1] = 0          # zero-incidence lanes do not march (FLAG_DARK_SKIP)
        s[~hit] = 0                             # escaped lanes do not either (FLAG_MISS_SKIP)
        sh.append(s)
    sh = np.stack(sh, axis=-1)
    rows = y1 - y0

    def waves(x):                               # [rows, w, ...] -> [rows/4, w/16, 64, ...]
        tail = x.shape[2:]
        return x.reshape(rows // 4, 4, w // 16, 16, *tail).swapaxes(1, 2).reshape(rows // 4, w // 16, 64, *tail)
    m, hh, s4 = waves(march), waves(hit), waves(sh)
    any_hit = hh.any(axis=2)
    cost = m.max(axis=2) + 4 * any_hit + np.where(any_hit[..., None], s4.max(axis=2), 0).sum(axis=2)
    return y0, cost.astype(np.float64)          # (a wave whose rays all escaped skips normal and shadows altogether)


def wave_costs(scene_path, w, h, cam_spec, procs):
    strips = [(scene_path, w, h, y0, min(y0 + 64, h), cam_spec) for y0 in range(0, h, 64)]
    with Pool(procs) as pool:
        parts = pool.map(_strip, strips)
    parts.sort(key=lambda t: t[0])
    return np.concatenate([p[1] for p in parts], axis=0)          # [h/4, w/16]


def simulate(costs, start=0.0):
    """List scheduling of `costs` (in launch order) on N_SIMD processor-sharing SIMDs with SLOTS slots; returns the makespan."""
    n = len(costs)
    if n == 0:
        return start
    nxt = 0
    # per SIMD: remaining work of its residents, time of its last update
    rem = [[] for _ in range(N_SIMD)]
    last = [start] * N_SIMD
    ver = [0] * N_SIMD
    heap = []

    def rate(k):
        return min(SOLO_SHARE, 1.0 / k)

    def schedule(s):
        if rem[s]:
            r = rate(len(rem[s]))
            heapq.heappush(heap, (last[s] + min(rem[s]) / r, s, ver[s]))
    # the dispatcher fills the machine breadth first: one wave per SIMD in turn
    for slot in range(SLOTS):
        for s in range(N_SIMD):
            if nxt < n:
                rem[s].append(float(costs[nxt])); nxt += 1
    for s in range(N_SIMD):
        schedule(s)
    t = start
    while heap:
        t, s, v = heapq.heappop(heap)
        if v != ver[s]:
            continue
        r = rate(len(rem[s]))
        dt = t - last[s]
        done = min(rem[s])
        rem[s] = [x - dt * r for x in rem[s]]
        # retire every wave that has (numerically) finished
        rem[s] = [x for x in rem[s] if x > 1e-9 * (1.0 + done)]
        last[s] = t
        while len(rem[s]) < SLOTS and nxt < n:
            rem[s].append(float(costs[nxt])); nxt += 1
        ver[s] += 1
        schedule(s)
    return t


def strategies(cost):
    """cost [H4, W16] per 16x4 wave -> {name: makespan} (work units), total work, and the prediction's quality"""
    H4, W16 = cost.shape
    total = cost.sum()
    out = {}
    out["rows"] = simulate(cost.reshape(-1))
    out["cols"] = simulate(cost.T.reshape(-1))
    out["lpt_exact"] = simulate(np.sort(cost.reshape(-1))[::-1])
    # 64x16 regions = 4 x 4 waves (pad the frame's edge with zero-cost waves that are never launched)
    RH, RW = 4, 4
    Hp, Wp = -(-H4 // RH) * RH, -(-W16 // RW) * RW
    pad = np.full((Hp, Wp), -1.0)
    pad[:H4, :W16] = cost
    reg = pad.reshape(Hp // RH, RH, Wp // RW, RW).swapaxes(1, 2).reshape(-1, RH * RW)      # [n_regions, 16]
    sample_ix = 5                                                # a wave in the middle of the region
    sample = reg[:, sample_ix].copy()
    valid = reg >= 0
    # a region at the edge whose sample wave lies beyond the frame: its first wave inside instead
    first_valid = valid.argmax(axis=1)
    use = np.where(sample >= 0, sample_ix, first_valid)
    sample = reg[np.arange(len(reg)), use]
    rest_mask = valid.copy()
    rest_mask[np.arange(len(reg)), use] = False
    order = np.argsort(-sample, kind="stable")
    launch1 = sample                                             # regions in row order
    launch2 = np.concatenate([reg[r][rest_mask[r]] for r in order])
    t1 = simulate(launch1)
    out["two_launches"] = simulate(launch2, start=t1 + LAUNCH_GAP)
    out["two_launches_free"] = launch1.sum() / N_SIMD + simulate(launch2)
    # ... and the most favourable form of sampling INSIDE one launch (no barrier between the phases, the sort free and always in
    # time): the sample waves first, then the first `alpha` of the frame's regions in row order, then the other regions longest
    # first by prediction — the prediction being the region's own sample, or the maximum over the region and its 8 neighbours
    # (a region whose sample wave sees sky may still hold a silhouette that its neighbour's sample has seen)
    n_ry, n_rx = Hp // RH, Wp // RW
    grid = sample.reshape(n_ry, n_rx)
    dil = np.pad(grid, 1, mode="edge")
    dil = np.max(np.stack([dil[dy:dy + n_ry, dx:dx + n_rx] for dy in range(3) for dx in range(3)]), axis=0).reshape(-1)
    for tag, pred in (("own", sample), ("dilated", dil)):
        for alpha in (0.0, 0.25, 0.5):
            n_fixed = int(alpha * len(reg))
            fixed_part = [reg[r][rest_mask[r]] for r in range(n_fixed)]
            later = np.arange(n_fixed, len(reg))
            later = later[np.argsort(-pred[later], kind="stable")]
            seq = np.concatenate([launch1] + fixed_part + [reg[r][rest_mask[r]] for r in later])
            out[f"one_launch_{tag}_alpha{alpha}"] = simulate(seq)
    mean_true = np.where(valid, reg, 0).sum(axis=1) / valid.sum(axis=1)
    order_true = np.argsort(-mean_true, kind="stable")
    out["region_exact"] = simulate(np.concatenate([reg[r][valid[r]] for r in order_true]))
    rest_mean = np.where(rest_mask, reg, 0).sum(axis=1) / np.maximum(rest_mask.sum(axis=1), 1)
    corr = float(np.corrcoef(sample, rest_mean)[0, 1])
    return out, float(total), corr, float(t1), float(launch1.sum() / N_SIMD)


def main():
    global LAUNCH_GAP
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default=os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--frames", default="c3,0,64,128,192", help="c3 = the scene's own camera; a number = that frame of the 256-frame orbit")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--gap-us", type=float, default=10.0, help="between launch 1's end and launch 2's first wave: sort + launch (C3 frame = 830 us)")
    a = ap.parse_args()
    w, h = (int(x) for x in a.size.split("x"))
    res = {"size": a.size, "machine": {"simds": N_SIMD, "slots": SLOTS, "solo_share": SOLO_SHARE}, "frames": {}}
    for f in a.frames.split(","):
        cost = wave_costs(a.scene, w, h, f, a.procs)
        ideal = cost.sum() / N_SIMD
        LAUNCH_GAP = a.gap_us / 830.0 * ideal                    # the gap in work units, scaled by this frame's ideal time ~ a C3 frame
        out, total, corr, t1, t1_ideal = strategies(cost)
        fixed = min(out["rows"], out["cols"])
        res["frames"][f] = {
            "waves": int(cost.size), "work_per_simd": ideal,
            "makespan_over_ideal": {k: round(v / ideal, 4) for k, v in out.items()},
            "gain_over_better_fixed_order": {k: round(fixed / v - 1, 4) for k, v in out.items()},
            "sample_wave_vs_rest_of_region_correlation": round(corr, 3),
            "launch1_makespan_over_its_share_of_the_work": round(t1 / t1_ideal, 2),
        }
        print(f, json.dumps(res["frames"][f]), file=sys.stderr, flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
