"""N>1 path on CPU: row-band partition + gather + un-interleave under gloo, world_size 2 and 3.

The pixel source is injected: each rank renders ITS bands with the CPU oracle (tests may use it), the
product's loltracer_amd.multi does the partition / gather / assembly exactly as bench.py does on
nccl, and rank 0 compares the assembled frame with a single-process full frame.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from loltracer_amd import multi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, w, h, band, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from loltracer_amd import scene as S
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = S.Scene.parse_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene4.lol"))

        def render_part(band_rows, world_, rank_):
            ys = multi.frame_rows_of_part(h, band_rows, world_, rank_).tolist()
            out = np.zeros((len(ys), w), dtype=np.uint32)
            for i, y in enumerate(ys):
                x, _, _ = O.render_rows(sc, w, h, y, y + 1)
                out[i] = x[y]
            return torch.from_numpy(out.view(np.int32))

        frame = multi.render_frame_distributed(render_part, w, h, band)
        if rank == 0:
            full, _, _ = O.render(sc, w, h, threads=2)
            q.put(bool(np.array_equal(frame.numpy().view(np.uint32), full)))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,band", [(2, 32, 4), (3, 36, 2)])
def test_band_partition_gather_assembles_the_frame(world, h, band):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 48, h, band, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _pipe_worker(rank, world, port, w, h, band, frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pipe = multi.GatherPipeline(w, h, band, torch.device("cpu"), depth=2)
        ys = multi.frame_rows_of_part(h, band, world, rank)
        ok = True
        for f in range(frames):
            def render(part, f=f):
                # pixel value encodes (frame, row, column) so any mix-up of slots / bands shows
                part.copy_((f * 1000003 + ys.view(-1, 1) * 4099 + torch.arange(w).view(1, -1)).to(torch.int32))
            pipe.submit(render)
            if rank == 0 and f >= 2:
                # frames complete in order, two in flight: after submitting f, frame f-2 has been assembled
                want = ((f - 2) * 1000003 + torch.arange(h).view(-1, 1) * 4099 + torch.arange(w).view(1, -1)).to(torch.int32)
                ok = ok and torch.equal(pipe.frame, want)
        last = pipe.drain()
        if rank == 0:
            want = ((frames - 1) * 1000003 + torch.arange(h).view(-1, 1) * 4099 + torch.arange(w).view(1, -1)).to(torch.int32)
            q.put(bool(ok and torch.equal(last, want) and pipe.frames_done == frames))
        else:
            assert last is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,band", [(2, 24, 4), (4, 32, 2)])
def test_gather_pipeline_keeps_frames_in_order(world, h, band):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pipe_worker, args=(r, world, port, 20, h, band, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_partition_helpers():
    # multiples of the 4-row wave patch are preferred, so no wave straddles two bands
    assert multi.choose_band_rows(4320, 8) == 12 and multi.choose_band_rows(4320, 2) == 16
    assert multi.choose_band_rows(4320, 4) == 12 and multi.choose_band_rows(2160, 4) == 12
    assert multi.choose_band_rows(30, 3) == 5 and multi.choose_band_rows(7, 2) == 0
    h, band, world = 48, 4, 3
    seen = torch.cat([multi.frame_rows_of_part(h, band, world, r) for r in range(world)])
    assert sorted(seen.tolist()) == list(range(h))
    assert [multi.part_rows(h, band, world, r) for r in range(world)] == [16, 16, 16]
    # assemble() inverts the interleave
    parts = torch.stack([multi.frame_rows_of_part(h, band, world, r).view(-1, 1).repeat(1, 5) for r in range(world)])
    assert torch.equal(multi.assemble(parts, h, band)[:, 0], torch.arange(h))


def test_kernel_row_mapping_agrees_with_partition_helper():
    # include/lol_gpu.h: local row r of part p holds frame row ((r/band)*n_parts + p)*band + r%band
    h, band, world = 40, 4, 5
    for p in range(world):
        ys = multi.frame_rows_of_part(h, band, world, p).tolist()
        assert ys == [((r // band) * world + p) * band + r % band for r in range(len(ys))]
