#!/usr/bin/env python3
"""How much of the device-to-host copies of `lol_headless --pipeline` runs UNDER a render kernel?

    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -- ./loltracer_amd/lib/lol_headless … --pipeline
    python tools/copy_overlap.py DIR > profiles/r3_headless_pipeline_copy_overlap.json

Reads the kernel and memory-copy traces of that run and intersects their time intervals."""
import csv
import glob
import json
import sys


def intervals(path, pred):
    out = []
    for r in csv.DictReader(open(path)):
        if pred(r):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return sorted(out)


def main():
    d = sys.argv[1]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    mt = glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True)[0]
    kernels = intervals(kt, lambda r: "lol_render" in r["Kernel_Name"] or "render_interp" in r["Kernel_Name"])
    rows = list(csv.DictReader(open(mt)))
    copies = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "DEVICE_TO_HOST" in r.get("Direction", "").upper().replace("MEMORY_COPY_", ""))
    if not copies:
        copies = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
    big = [c for c in copies if c[1] - c[0] > 100_000]                  # the frame copies (> 0.1 ms), not the small uploads
    under = 0
    for a, b in big:
        for ka, kb in kernels:
            lo, hi = max(a, ka), min(b, kb)
            if hi > lo:
                under += hi - lo
    total = sum(b - a for a, b in big)
    span = (max(k[1] for k in kernels) - min(k[0] for k in kernels)) if kernels else 0
    print(json.dumps({"render_kernels": len(kernels), "kernel_ms_avg": round(sum(b - a for a, b in kernels) / max(len(kernels), 1) / 1e6, 4),
                      "frame_copies": len(big), "copy_ms_avg": round(total / max(len(big), 1) / 1e6, 4),
                      "copy_time_under_a_render_kernel": round(under / max(total, 1), 4),
                      "span_ms_per_kernel": round(span / max(len(kernels), 1) / 1e6, 4)}))


if __name__ == "__main__":
    main()
