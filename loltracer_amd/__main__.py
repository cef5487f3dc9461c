"""python -m loltracer_amd scene.lol [-o frame.ppm] [--size WxH] [--max-steps N] [--device D] [--frames N]

Renders a `.lol` scene on the GPU through the C ABI (liblol_gpu.so) and writes a binary PPM — the Python spelling of
`loltracer_amd/lib/lol_headless`.  There is no CPU rendering path."""
from __future__ import annotations

import argparse
import sys
import time

import numpy as np

from . import gpu, scene as S


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m loltracer_amd", description=__doc__.split("\n\n")[1])
    ap.add_argument("scene")
    ap.add_argument("-o", "--out", default=None)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--max-steps", type=int, default=256)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--frames", type=int, default=1)
    args = ap.parse_args(argv)
    w, h = (int(v) for v in args.size.lower().split("x"))
    try:
        sc = S.Scene.parse_file(args.scene)
    except S.SceneError as e:
        print(e.message, file=sys.stderr)
        return 1
    if not sc.validate_materials():
        print("scene_validate_materials failed", file=sys.stderr)
        return 1
    r = gpu.Renderer(args.device)
    r.prepare(sc)
    surf = np.zeros((h, w), dtype=np.uint32)
    for f in range(args.frames):
        t0 = time.perf_counter()
        r.render_host(surf.ctypes.data, w, h, args.max_steps)
        dt = (time.perf_counter() - t0) * 1e3
        print(f"Frame {f + 1}: {dt:.3f}ms  {w * h / dt / 1e3:.1f} Mpixels/s  [{r.kernel_name()}]")
    if args.out:
        rgb = np.stack([(surf >> 16) & 0xFF, (surf >> 8) & 0xFF, surf & 0xFF], axis=-1).astype(np.uint8)
        with open(args.out, "wb") as fp:
            fp.write(b"P6\n%d %d\n255\n" % (w, h))
            fp.write(rgb.tobytes())
    r.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
