#!/usr/bin/env python3
"""Frames in flight under the profiler: how much of the render kernels' time runs CONCURRENTLY with another render kernel?

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --workload orbit --no-cpu-baseline
    python tools/kernel_overlap.py DIR > profiles/r5_orbit_kernel_overlap.json

Reads the kernel trace of that run (start / end timestamps of every dispatch) and reports, over the LAST `--last` render kernels
(the timed pass): their average duration, the wall time they span per kernel, the share of the spanned time in which exactly one /
two / more render kernels were running, and the queues they ran on."""
import argparse
import csv
import glob
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--last", type=int, default=256)
    a = ap.parse_args()
    kt = glob.glob(a.dir + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(kt)) if "lol_render" in r["Kernel_Name"] or "render_interp" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-a.last:]
    iv = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
    events = sorted([(s, 1) for s, _ in iv] + [(e, -1) for _, e in iv])
    depth, last_t, at = 0, events[0][0], {}
    for t, d in events:
        at[depth] = at.get(depth, 0) + (t - last_t)
        depth += d
        last_t = t
    span = max(e for _, e in iv) - min(s for s, _ in iv)
    busy = sum(v for k, v in at.items() if k >= 1)
    queues = sorted({r.get("Queue_Id", "?") for r in rows})
    print(json.dumps({"render_kernels": len(iv), "kernel_ms_avg": round(sum(e - s for s, e in iv) / len(iv) / 1e6, 4),
                      "span_ms_per_kernel": round(span / len(iv) / 1e6, 4),
                      "share_of_span_with_0_1_2_3plus_kernels_running": [round(at.get(0, 0) / span, 4), round(at.get(1, 0) / span, 4), round(at.get(2, 0) / span, 4),
                                                                         round(sum(v for k, v in at.items() if k >= 3) / span, 4)],
                      "average_kernels_running_while_any_is": round(sum(k * v for k, v in at.items()) / max(busy, 1), 3),
                      "queues": queues}))


if __name__ == "__main__":
    main()
