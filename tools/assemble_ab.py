#!/usr/bin/env python3
"""The root's un-interleave, both ways, on one GPU (VERDICT round 2, next #1a): the gathered parts of a C4 frame
(7680x4320, 8 ranks) → the final framebuffer with (a) the round-2 torch permuted copy `frame.copy_(assemble(staging))`
and (b) the library's uint4 kernel lol_gpu_assemble_parts_at, equal and weighted splits; HIP-event times, GB/s of
2 x 132.7 MB moved, and frame equality.      python tools/assemble_ab.py > gpurun_out/assemble_ab.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from loltracer_amd import gpu, multi, scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def main():
    w, h, world = 7680, 4320, 8
    sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))
    r = gpu.Renderer(0)
    r.prepare(sc)
    whole = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    r.render_into(whole.data_ptr(), w, h)
    r.sync()
    out = {"frame": f"{w}x{h}", "bytes_moved": 2 * w * h * 4, "cases": []}
    stream = torch.cuda.current_stream().cuda_stream
    for band, root in ((12, 0), (16, 15), (12, 10), (16, 12)):
        P = multi.Partition(h, world, band, root)
        geometry = [gpu.Rows(*g) for g in P.geometry]
        staging = torch.zeros((world, P.max_rows, w), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for p in range(world):
            r.render_into(staging.view(-1, w)[P.part_row0[p]:].data_ptr(), w, h, rows=geometry[p])
        r.sync()
        frame = torch.empty((h, w), dtype=torch.int32, device="cuda")
        rec = dict(P.describe())
        if not root:
            ms = timed(lambda: frame.copy_(multi.assemble(staging, h, P.band)))
            rec["torch_permute_copy_ms"] = round(ms, 4)
            rec["torch_permute_copy_GBps"] = round(out["bytes_moved"] / ms / 1e6, 1)
            rec["torch_equal"] = bool(torch.equal(frame, whole))
            frame.zero_()
        idx = P.staging_index().cuda()
        ms = timed(lambda: torch.index_select(staging.view(-1, w), 0, idx, out=frame))
        rec["torch_index_select_ms"] = round(ms, 4)
        frame.zero_()
        ms = timed(lambda: gpu.assemble_parts_at(r, staging.data_ptr(), geometry, P.part_row0, w, h, frame.data_ptr(), w * 4, stream))
        rec["library_kernel_ms"] = round(ms, 4)
        rec["library_kernel_GBps"] = round(out["bytes_moved"] / ms / 1e6, 1)
        rec["library_equal"] = bool(torch.equal(frame, whole))
        out["cases"].append(rec)
    print(json.dumps(out, indent=1))
    r.close()


if __name__ == "__main__":
    main()
