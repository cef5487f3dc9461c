#!/usr/bin/env python3
"""rocprofv3 --pmc CSVs → per-kernel, per-dispatch averages → profiles/pmc_traffic.json (what bench.py reads for
`roofline.traffic`) or any other summary file.

    python tools/pmc_summary.py --kernel lol_render_spec --workload c3 --pixels 8294400 \
        --out profiles/pmc_traffic.json  profiles/r2_spec_pmc_*_counter_collection.csv

Every CSV is one rocprofv3 pass (one counter group per pass, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE do not fit one pass).  Rows of the named kernel are averaged per counter over its dispatches.
Traffic per launch = WRITE_SIZE + 2 x FETCH_SIZE in bytes (both are reported in KiB; on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes, hence the factor 2 — same guide, HBM section).  Also derives the issue view used in
LABNOTES.md §3.4: cycles per XCD (GRBM_GUI_ACTIVE / 8), clock, cycles per VALU wave-instruction per SIMD.
"""
import argparse
import collections
import csv
import json
import os
import sys


def summarise(paths, kernel):
    acc = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for path in paths:
        seen = set()
        for row in csv.DictReader(open(path)):
            if kernel not in row["Kernel_Name"]:
                continue
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            key = (path, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[os.path.basename(path)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6)
    # An instruction count is the same for every dispatch of a deterministic kernel; now and then one dispatch of a pass comes
    # back with another kernel's work counted in (seen: 1 of 107, every counter of that pass high at once).  Where at least
    # 90 % of the dispatches agree on one value, that value is the summary and the others are reported, not averaged in.
    counters, off_mode = {}, {}
    for k, v in sorted(acc.items()):
        value, n_same = collections.Counter(v).most_common(1)[0]
        if n_same >= 0.9 * len(v):
            counters[k] = value
            if n_same != len(v):
                off_mode[k] = len(v) - n_same
        else:
            counters[k] = sum(v) / len(v)
    launches = {k: len(v) for k, v in acc.items()}
    launches["_dispatches_off_the_common_value"] = off_mode
    ms = {k: sum(v) / len(v) for k, v in dur.items()}
    return counters, launches, ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv", nargs="+")
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--pixels", type=int, default=3840 * 2160)
    ap.add_argument("--out", required=True)
    ap.add_argument("--merge", action="store_true", help="keep the other kernels already in --out")
    ap.add_argument("--kernel-key", default=None, help="lol_gpu_kernel_key() of the code that was profiled (bench.py: config.kernel_key)")
    ap.add_argument("--min-dispatches", type=int, default=1, help="fail when a counter was seen on fewer dispatches of the kernel")
    a = ap.parse_args()
    c, n, ms = summarise(a.csv, a.kernel)
    if not c:
        sys.exit(f"no rows of kernel {a.kernel!r} in {a.csv}")
    off_mode = n.pop("_dispatches_off_the_common_value")
    few = {k: v for k, v in n.items() if v < a.min_dispatches}
    if few:
        sys.exit(f"kernel {a.kernel!r}: fewer than {a.min_dispatches} dispatches behind {few} — is this the profile of another kernel's run?")
    rec = {"workload": a.workload, "pixels_per_launch": a.pixels, "kernel_key": a.kernel_key, "counters_avg_per_dispatch": c,
           "dispatches_per_counter": n, "dispatches_off_the_common_value": off_mode, "kernel_ms_under_pmc": ms, "sources": [os.path.basename(p) for p in a.csv]}
    if "WRITE_SIZE" in c and "FETCH_SIZE" in c:
        rec["write_bytes"] = c["WRITE_SIZE"] * 1024
        rec["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
        rec["traffic_bytes"] = rec["write_bytes"] + 2 * rec["fetch_bytes_raw"]
        rec["algorithmic_bytes"] = a.pixels * 4
    if "GRBM_GUI_ACTIVE" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8                                   # the counter sums the 8 XCDs
        rec["cycles_per_xcd"] = cyc
        t = [v for k, v in ms.items()]
        if t:
            rec["clock_ghz_under_pmc"] = cyc / (sum(t) / len(t) * 1e-3) / 1e9
        simds = 256 * 4
        for name, key in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("branch", "SQ_INSTS_BRANCH"), ("smem", "SQ_INSTS_SMEM")):
            if key in c:
                rec[f"{name}_wave_instructions_per_simd"] = c[key] / simds
                rec[f"cycles_per_{name}_instruction_per_simd"] = cyc / (c[key] / simds)
        if "SQ_INSTS_VALU" in c:
            rec["valu_instructions_per_pixel"] = c["SQ_INSTS_VALU"] * 64 / a.pixels
        if "SQ_INSTS_VALU_TRANS_F32" in c and "SQ_INSTS_VALU" in c:
            rec["transcendental_share_of_valu"] = c["SQ_INSTS_VALU_TRANS_F32"] / c["SQ_INSTS_VALU"]
        if "SQC_ICACHE_REQ" in c and "SQC_ICACHE_MISSES" in c:
            rec["icache_miss_rate"] = c["SQC_ICACHE_MISSES"] / max(c["SQC_ICACHE_REQ"], 1.0)
    out = {"source": "rocprofv3 --pmc passes on MI355X (tools/final_profile.sh), summarised by tools/pmc_summary.py",
           "units": "WRITE_SIZE / FETCH_SIZE are KiB per dispatch as reported; bytes = KiB*1024; FETCH doubled per the gfx950 "
                    "correction (MI355X_MICROARCH.md, HBM); GRBM_GUI_ACTIVE sums the 8 XCDs", "kernels": {}}
    if a.merge and os.path.exists(a.out):
        out = json.load(open(a.out))
    out["kernels"][a.kernel] = rec
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
