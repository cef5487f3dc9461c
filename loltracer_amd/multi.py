"""Row-tile partition of one frame over the ranks of a node, assembled with a gather.

The reference parallelises a frame by letting workers claim scan-lines from an
atomic counter (naive_renderer.c:216, main.c:189-194): every pixel is
independent.  Across GPUs the same independence is used statically: the frame's
rows are cut into bands of `band_rows`, band b goes to part b % n_parts (fine
interleave, so sky rows and blob rows are spread evenly), every part belongs to
one rank, each rank renders its parts compactly (include/lol_gpu.h,
lol_gpu_rows) and one gather over RCCL/xGMI brings them to rank 0, which
un-interleaves them.  One process per GPU; torch.distributed is only the
transport.

Cost-weighted split (`Partition`): rank 0 also receives and un-interleaves the
whole frame, so with an equal share it is the straggler.  Every rank owns one
band per cycle of rows and renders its part with ONE launch (eight small
launches took 2.7x as long as one, measured); the root's bands are simply less
tall than the others' — the geometry of lol_gpu_split_rows behind the C ABI,
so both hosts cut a frame identically.

Everything here is device-agnostic (tensors in, tensors out) so the same code
runs under gloo on CPU in the tests and under nccl (= RCCL) on MI355X; on a GPU
the un-interleave is the library's uint4 kernel (lol_gpu_assemble_parts_at,
injected as `assembler`), on the CPU an index_select.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def choose_band_rows(h: int, world: int, preferred: int = 16, patch_rows: int = 4) -> int:
    """Band height with h % (band * world) == 0 (equal parts): the largest multiple of the kernel's patch
    height (16x4 pixels per wave) up to `preferred`, so that no wave straddles two bands — frame rows that are
    `world` bands apart; failing that the largest divisor <= 8; 0 if none."""
    limit = max(h // world, 1)
    for band in range(min(preferred, limit) // patch_rows * patch_rows, 0, -patch_rows):
        if h % (band * world) == 0:
            return band
    for band in range(min(8, limit), 0, -1):
        if h % (band * world) == 0:
            return band
    return 0


def part_rows(h: int, band_rows: int, world: int, rank: int) -> int:
    bands = (h + band_rows - 1) // band_rows
    n = 0
    for b in range(rank, bands, world):
        n += min(band_rows, h - b * band_rows)
    return n


def frame_rows_of_part(h: int, band_rows: int, world: int, rank: int) -> torch.Tensor:
    """Frame row index of every local row of `rank` (the inverse of the kernel's row mapping)."""
    bands = (h + band_rows - 1) // band_rows
    ys = []
    for b in range(rank, bands, world):
        y0 = b * band_rows
        ys.extend(range(y0, min(y0 + band_rows, h)))
    return torch.tensor(ys, dtype=torch.long)


def assemble(parts: torch.Tensor, h: int, band_rows: int) -> torch.Tensor:
    """[world, rows_per_part, w] gathered parts → [h, w] frame (equal parts)."""
    world, rpp, w = parts.shape
    assert rpp * world == h and rpp % band_rows == 0
    nb = rpp // band_rows
    # parts[r, b, i] is frame row (b*world + r)*band + i
    return parts.view(world, nb, band_rows, w).permute(1, 0, 2, 3).reshape(h, w)


def split_rows(world: int, band_rows: int, root_band_rows: int = 0):
    """[(band_rows, cycle_rows, offset_rows)] of every rank: one band per rank per cycle, in rank order, rank 0's
    root_band_rows tall (0 = like the others).  Mirrors lol_gpu_split_rows (include/lol_gpu.h)."""
    if world < 1 or band_rows < 1 or root_band_rows < 0:
        raise ValueError(f"split_rows({world}, {band_rows}, {root_band_rows})")
    if world == 1:
        return [(band_rows, band_rows, 0)]
    heights = [root_band_rows or band_rows] + [band_rows] * (world - 1)
    cycle = sum(heights)
    out, at = [], 0
    for hgt in heights:
        out.append((hgt, cycle, at))
        at += hgt
    return out


def rows_of_split(h: int, band: int, cycle: int, offset: int) -> int:
    """Rows of a frame of height h that belong to the part (band rows at `offset` of every `cycle`): lol_gpu_part_rows."""
    n = 0
    for y0 in range(offset, h, cycle):
        n += min(band, h - y0)
    return n


class Partition:
    """How one frame of `h` rows is cut over `world` ranks: the same on every rank.

    geometry[r]   (band_rows, cycle_rows, offset_rows) of rank r — its lol_gpu_rows: ONE launch renders the rank's part
    rank_rows[r]  rows rank r renders
    max_rows      rows of every rank's local buffer (padded to the largest, so one gather of equal tensors serves)
    part_row0[r]  first row of rank r inside the gathered [world * max_rows, w] staging buffer
    """

    def __init__(self, h: int, world: int, band_rows: int = 0, root_band_rows: int = 0):
        self.h, self.world = h, world
        if world == 1:
            band_rows, root_band_rows = h, 0
        self.band = band_rows or choose_band_rows(h, world)
        if self.band <= 0:
            self.band = 4 if h >= 4 * world else 1
        self.root_band = root_band_rows if world > 1 else 0
        self.geometry = split_rows(world, self.band, self.root_band)
        self.rank_rows = [rows_of_split(h, *g) for g in self.geometry]
        assert sum(self.rank_rows) == h
        self.max_rows = max(self.rank_rows)
        self.part_row0 = [r * self.max_rows for r in range(world)]

    def frame_rows_of_rank(self, rank: int) -> torch.Tensor:
        """Frame row of every (used) local row of `rank`, in local order (the inverse of the kernel's row mapping)."""
        band, cycle, offset = self.geometry[rank]
        ys = []
        for y0 in range(offset, self.h, cycle):
            ys.extend(range(y0, min(y0 + band, self.h)))
        return torch.tensor(ys, dtype=torch.long)

    def staging_index(self) -> torch.Tensor:
        """index[y] = row of the gathered staging buffer that holds frame row y."""
        idx = torch.empty(self.h, dtype=torch.long)
        for r in range(self.world):
            ys = self.frame_rows_of_rank(r)
            idx[ys] = self.part_row0[r] + torch.arange(len(ys))
        return idx

    def describe(self) -> dict:
        return {"band_rows": self.band, "root_band_rows": self.root_band or self.band, "cycle_rows": self.geometry[0][1],
                "rows_per_rank": self.rank_rows}


def gather_frame(local: torch.Tensor, h: int, band_rows: int, group=None, dst: int = 0,
                 out: Optional[torch.Tensor] = None, staging: Optional[torch.Tensor] = None
                 ) -> Optional[torch.Tensor]:
    """Gather every rank's compact part [rows_per_part, w] to `dst` and return the [h, w] frame there.

    `staging` ([world, rows_per_part, w], on dst) and `out` ([h, w]) can be preallocated so the
    timed loop allocates nothing.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    rpp, w = local.shape
    if rank == dst:
        if staging is None:
            staging = torch.empty((world, rpp, w), dtype=local.dtype, device=local.device)
        dist.gather(local, [staging[i] for i in range(world)], dst=dst, group=group)
        frame = assemble(staging, h, band_rows)
        if out is not None:
            out.copy_(frame)
            return out
        return frame.contiguous()
    dist.gather(local, None, dst=dst, group=group)
    return None


class GatherPipeline:
    """Frames in flight: while frame i's parts travel to `dst`, frame i+1 is already rendering.

    Each rank owns `depth` local buffers of partition.max_rows rows.  submit(render) renders the next frame's parts
    into the next buffer (after making sure the gather that last read that buffer has finished) and starts an
    asynchronous gather of it; on `dst` the gathered parts are un-interleaved into `frame` when their gather is waited
    for.  drain() completes everything in flight.  A renderer that produces a stream of frames (the reference's frame
    loop, main.c:163-211) loses nothing by this: every frame still ends up assembled on `dst`, in order.

    `assembler(staging, frame, partition, stream_handle)`: the un-interleave on a GPU (bench.py passes the library's
    lol_gpu_assemble_parts_at); None = index_select (CPU tensors, tests).
    """

    def __init__(self, w: int, h: int, band_rows: int, device, group=None, dst: int = 0, depth: int = 2,
                 dtype=torch.int32, force_collective: bool = False, partition: Optional[Partition] = None,
                 assembler: Optional[Callable] = None, kernel_streams: Optional[list] = None):
        self.group, self.dst, self.h, self.w = group, dst, h, w
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # force_collective: run the gather even in a 1-rank group (exercises the backend's gather on one device)
        self.single = self.world == 1 and not (force_collective and dist.is_initialized())
        if dst != 0:
            raise ValueError("the root of the partition is rank 0")
        self.partition = partition if partition is not None else Partition(h, self.world, band_rows if self.world > 1 else h)
        P = self.partition
        if P.h != h:
            raise ValueError("partition is for another frame height")
        self.band = P.band
        # a partition cut for MORE ranks than the group has (bench.py's one-GPU rehearsal of rank 0's cadence):
        # this rank renders its share, the gather moves what the group has, the assembly covers the whole frame
        self.rows = P.rank_rows[self.rank]
        self.depth = depth if not self.single else 1
        self.local = [torch.zeros((P.max_rows, w), dtype=dtype, device=device) for _ in range(self.depth)]
        self.is_dst = self.rank == dst
        self.staging = ([torch.zeros((P.world, P.max_rows, w), dtype=dtype, device=device) for _ in range(self.depth)]
                        if not self.single and self.is_dst else None)
        self.frame = torch.empty((h, w), dtype=dtype, device=device) if not self.single and self.is_dst else None
        self.assembler = assembler
        self._index = P.staging_index().to(device) if (self.frame is not None and assembler is None) else None
        self.work = [None] * self.depth
        self.n = 0
        self.frames_done = 0
        # On a GPU the root un-interleaves on a stream of its own, so that neither the wait for the parts nor the
        # 4 B/pixel un-interleave sits in the render stream between two frame kernels; `assembled[slot]` orders the
        # next gather into staging[slot] (and the caller's reads of `frame`) after it.
        self.cuda = torch.device(device).type == "cuda"
        self.asm_stream = torch.cuda.Stream(device=device) if self.cuda and self.is_dst and not self.single else None
        self.assembled = [None] * self.depth
        # kernel_streams (GPU, optional): one torch stream per slot.  Slot k's rendering and its gather are issued on stream k, so
        # the KERNELS of consecutive frames run on different streams and overlap (a rank's launch of its bands is a small launch:
        # its ramp and tail are a larger share of it than of a whole frame's; bench.py: LOL_BENCH_KERNEL_STREAMS).  Everything about
        # one slot is still ordered on one stream; slots only share the root's frame, which the one assembly stream orders.
        self.kernel_streams = list(kernel_streams) if (kernel_streams and self.cuda and not self.single) else None
        if self.kernel_streams is not None and len(self.kernel_streams) < self.depth:
            raise ValueError("one kernel stream per slot, please")

    def _slot_stream(self, slot: int):
        import contextlib
        return torch.cuda.stream(self.kernel_streams[slot]) if self.kernel_streams is not None else contextlib.nullcontext()

    def _assemble(self, slot: int, stream_handle):
        if self.assembler is not None:
            self.assembler(self.staging[slot], self.frame, self.partition, stream_handle)
        else:
            torch.index_select(self.staging[slot].view(-1, self.w), 0, self._index, out=self.frame)

    def _finish(self, slot: int):
        w = self.work[slot]
        if w is None:
            return
        self.work[slot] = None
        if self.asm_stream is not None:
            with self._slot_stream(slot):
                w.wait()                               # the render stream may overwrite local[slot] only after the
                                                       # collective that reads it (two frames of slack at depth 2)
            with torch.cuda.stream(self.asm_stream):
                w.wait()                               # orders the assembly stream after the collective
                self._assemble(slot, self.asm_stream.cuda_stream)
                ev = torch.cuda.Event()
                ev.record()
            self.assembled[slot] = ev
        else:
            with self._slot_stream(slot):
                w.wait()                               # orders the (slot's) stream after the collective
                if self.is_dst:
                    self._assemble(slot, torch.cuda.current_stream().cuda_stream if self.cuda else None)
        self.frames_done += 1

    def submit(self, render):
        """render(local_buffer): enqueue the rendering of this rank's part into the [max_rows, w] tensor (compactly,
        from row 0; rows beyond partition.rank_rows[rank] are padding)."""
        slot = self.n % self.depth
        self.n += 1
        if self.single:
            render(self.local[0])
            self.frame = self.local[0]
            self.frames_done += 1
            return
        self._finish(slot)
        with self._slot_stream(slot):
            render(self.local[slot])
            self._gather(slot)

    def _gather(self, slot: int):
        if self.assembled[slot] is not None:           # staging[slot] is still being read by the assembly stream
            self.assembled[slot].wait()
            self.assembled[slot] = None
        if self.is_dst:
            self.work[slot] = dist.gather(self.local[slot], [self.staging[slot][i] for i in range(self.world)],
                                          dst=self.dst, group=self.group, async_op=True)
        else:
            self.work[slot] = dist.gather(self.local[slot], None, dst=self.dst, group=self.group, async_op=True)

    def regather(self, slot: int = 0):
        """Gather + assemble the part already sitting in local[slot] again, synchronously (for timing the
        exchange step on its own).  Every rank must call it."""
        if self.single:
            return
        self._finish(slot)
        with self._slot_stream(slot):
            self._gather(slot)
        self._finish(slot)
        self._join()

    def drain(self):
        """Finish every frame in flight, oldest first; returns the last assembled frame on dst (None elsewhere)."""
        if not self.single:
            for k in range(self.depth):
                self._finish((self.n + k) % self.depth)
            self._join()
        return self.frame if self.is_dst else None

    def _join(self):
        """Order the current stream after every assembly issued so far (before the caller reads `frame`)."""
        for ev in self.assembled:
            if ev is not None:
                ev.wait()


def render_frame_distributed(render_part: Callable[[int, int, int], torch.Tensor], w: int, h: int,
                             band_rows: int, group=None, dst: int = 0, out=None, staging=None):
    """render_part(band_rows, world, rank) → this rank's compact [rows, w] int32 tensor; returns the frame on dst."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world > 1 and h % (band_rows * world) != 0:
        raise ValueError(f"h={h} must be a multiple of band_rows*world={band_rows * world}")
    local = render_part(band_rows, world, rank)
    if world == 1:
        return local
    return gather_frame(local, h, band_rows, group=group, dst=dst, out=out, staging=staging)
