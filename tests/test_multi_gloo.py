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


# ---------------------------------------------------------------- cost-weighted split (the root's bands are less tall)

def test_split_geometry_matches_the_library():
    """loltracer_amd.multi (one process per GPU) and lol_gpu_split_rows / lol_gpu_part_rows (behind the C ABI) must cut a
    frame identically."""
    from loltracer_amd import gpu
    for world in (1, 2, 3, 4, 8, 16):
        for band, root in ((16, 0), (16, 15), (12, 11), (12, 10), (8, 7), (16, 12), (4, 1), (7, 3)):
            geo = multi.split_rows(world, band, root)
            lib = gpu.split_rows(world, band, root, world)
            assert geo == [(r.band_rows, r.cycle_rows, r.offset_rows) for r in lib], (world, band, root)
            for h in (4320, 2160, 1080, 97, 720, 5):
                assert [multi.rows_of_split(h, *g) for g in geo] == [gpu.part_rows(h, r) for r in lib]
    assert multi.split_rows(3, 8, 5) == [(5, 21, 0), (8, 21, 5), (8, 21, 13)]
    with pytest.raises(ValueError):
        multi.split_rows(2, 0)


@pytest.mark.parametrize("world,band,root,h", [(8, 16, 15, 4320), (8, 12, 10, 4320), (4, 8, 7, 4320), (2, 16, 12, 4320), (3, 8, 5, 101),
                                                (8, 12, 0, 4320), (2, 12, 1, 50), (5, 4, 3, 17)])
def test_partition_covers_every_row_once_and_is_balanced(world, band, root, h):
    P = multi.Partition(h, world, band, root)
    seen = torch.cat([P.frame_rows_of_rank(r) for r in range(world)])
    assert sorted(seen.tolist()) == list(range(h))
    assert P.rank_rows == [len(P.frame_rows_of_rank(r)) for r in range(world)]
    idx = P.staging_index()
    assert len(set(idx.tolist())) == h and int(idx.max()) < world * P.max_rows
    # the others are equal to within one band, the root's share follows its band height
    others = P.rank_rows[1:]
    assert max(others) - min(others) <= band
    if h >= 1000:
        rb = root or band
        assert abs(P.rank_rows[0] / h - rb / (rb + (world - 1) * band)) < 0.01
    # the kernel's mapping (include/lol_gpu.h: lol_gpu_part_frame_row) is the inverse
    from loltracer_amd import gpu
    lib = gpu.gpu_lib()
    for r in (0, world // 2, world - 1):
        rows = gpu.Rows(*P.geometry[r])
        ys = P.frame_rows_of_rank(r).tolist()
        assert ys == [lib.lol_gpu_part_frame_row(h, rows, i) for i in range(len(ys))]
        assert lib.lol_gpu_part_frame_row(h, rows, len(ys)) == -1


def _weighted_worker(rank, world, port, w, h, band, root, frames, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        P = multi.Partition(h, world, band, root)
        pipe = multi.GatherPipeline(w, h, P.band, torch.device("cpu"), depth=2, partition=P)
        ys = P.frame_rows_of_rank(rank)
        assert pipe.local[0].shape == (P.max_rows, w) and len(ys) == P.rank_rows[rank]
        for f in range(frames):
            def render(buf, f=f):
                buf.fill_(-1)                          # the padding behind a short part must never reach the frame
                buf[:len(ys)] = (f * 1000003 + ys.view(-1, 1) * 4099 + torch.arange(w).view(1, -1)).to(torch.int32)
            pipe.submit(render)
        last = pipe.drain()
        if rank == 0:
            want = ((frames - 1) * 1000003 + torch.arange(h).view(-1, 1) * 4099 + torch.arange(w).view(1, -1)).to(torch.int32)
            q.put(bool(torch.equal(last, want)) and P.rank_rows[0] < P.rank_rows[1])
        else:
            assert last is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,h,band,root", [(2, 120, 8, 6), (3, 101, 8, 5), (4, 96, 4, 1),
                                               (8, 4320, 16, 15), (8, 4320, 12, 10)])      # BASELINE config 4's own geometry, eight ranks
def test_unequal_parts_gather_and_assemble(world, h, band, root):
    """The root's bands are less tall than the others' (its smaller share), heights that no cycle divides: the gathered,
    padded parts still assemble to the frame.  The last two cases are config 4 as `bench.py --gpus 8` cuts it — 4320 rows,
    bands of 16 (15 for rank 0) and of 12 (10) — as eight PROCESSES over gloo (round-5 review: world 8 was geometry only)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_weighted_worker, args=(r, world, port, 20, h, band, root, 4, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
