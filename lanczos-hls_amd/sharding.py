"""Frame-batch and row-strip sharding over ranks (one process per GPU).

The reference has no parallelism beyond HLS unrolling (ROW_WORKERS output rows per strip,
lanczos.cpp:72-82); this is the multi-GPU layer the north star adds.  The resample shards with NO data-path
collective: frames are independent, and output row strips of one frame are independent given an input halo
(lanczos_strip_input_rows).  The only exchange is an optional root scatter (inputs) / gather (outputs), done
with torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.

`compute` callbacks keep this module independent of where the arithmetic runs: the product passes
Context.resample / resample_strip; the CPU tests pass the oracle.
"""
from typing import Callable, List, Sequence, Tuple


def frame_shards(n_frames: int, world: int) -> List[Tuple[int, int]]:
    """Block partition of a batch: rank r gets frames [start, start+count). Earlier ranks take the remainder."""
    if n_frames < 0 or world < 1:
        raise ValueError("n_frames >= 0 and world >= 1 required")
    base, rem = divmod(n_frames, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((start, cnt))
        start += cnt
    return out


def strip_shards(out_h: int, world: int, strip_input_rows: Callable[[int, int], Tuple[int, int]],
                 min_first: int = 0) -> List[Tuple[int, int, int, int]]:
    """Row strips of ONE frame: rank r produces output rows [out_row0, out_row0+out_rows) from input rows
    [in_row0, in_row0+in_rows).  `strip_input_rows(out_row0, out_rows)` is lanczos_strip_input_rows.
    `min_first`: the first strip must hold at least this many rows (the in-place prefix recurrence of
    full_TB.h:67-77 needs output rows [0, M) in one place)."""
    if world < 1 or out_h < 1:
        raise ValueError("world >= 1 and out_h >= 1 required")
    bounds = [out_h * i // world for i in range(world + 1)]
    if world > 1 and bounds[1] < min_first:
        bounds[1] = min(min_first, out_h)
        for i in range(2, world):  # keep the remaining bounds monotone
            bounds[i] = max(bounds[i], bounds[1])
    shards = []
    for r in range(world):
        row0, rows = bounds[r], bounds[r + 1] - bounds[r]
        if rows <= 0:
            shards.append((row0, 0, 0, 0))
            continue
        in0, n = strip_input_rows(row0, rows)
        shards.append((row0, rows, in0, n))
    return shards


def scatter_frames(dist, frames_root, shards: Sequence[Tuple[int, int]], like, src: int = 0):
    """Root holds frames_root [F, ...]; every rank receives its block (a tensor like `like`, [count, ...]).
    Uses point-to-point sends because blocks may be unequal (F not divisible by the world size)."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    start, cnt = shards[rank]
    mine = torch.empty((cnt,) + tuple(like.shape[1:]), dtype=like.dtype, device=like.device)
    if rank == src:
        reqs = []
        for r in range(world):
            s, c = shards[r]
            if c == 0:
                continue
            if r == src:
                mine.copy_(frames_root[s:s + c])
            else:
                reqs.append(dist.isend(frames_root[s:s + c].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif cnt > 0:
        dist.recv(mine, src=src)
    return mine


def gather_frames(dist, mine, shards: Sequence[Tuple[int, int]], out_root=None, dst: int = 0):
    """Inverse of scatter_frames: root receives every rank's block into out_root [F, ...]."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        for r in range(world):
            s, c = shards[r]
            if c == 0:
                continue
            if r == dst:
                out_root[s:s + c].copy_(mine)
            else:
                dist.recv(out_root[s:s + c], src=r)
        return out_root
    if shards[rank][1] > 0:
        dist.send(mine.contiguous(), dst=dst)
    return None


def resample_batch(dist, frames_root, like_in, like_out, compute: Callable, src: int = 0):
    """Config 4 of BASELINE.json: a batch of frames sharded over the ranks.  frames_root / the result live on
    rank `src` only; `compute(block_in) -> block_out` runs on every rank."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    n = [int(frames_root.shape[0]) if rank == src else 0]
    t = torch.tensor(n, dtype=torch.int64, device=like_in.device)
    dist.broadcast(t, src=src)
    shards = frame_shards(int(t.item()), world)
    mine = scatter_frames(dist, frames_root, shards, like_in, src)
    out_block = compute(mine) if shards[rank][1] > 0 else like_out[:0]
    out_root = None
    if rank == src:
        out_root = torch.empty((int(t.item()),) + tuple(like_out.shape[1:]), dtype=like_out.dtype,
                               device=like_out.device)
    return gather_frames(dist, out_block, shards, out_root, src)


def resample_strips(dist, frame_root, out_h: int, strip_input_rows: Callable, compute_strip: Callable,
                    like_in, like_out, min_first: int = 0, src: int = 0):
    """Config 5 of BASELINE.json: ONE frame cut into output row strips with an input halo.
    `compute_strip(rows_in, out_row0, out_rows, in_row0) -> rows_out` runs on every rank."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    shards = strip_shards(out_h, world, strip_input_rows, min_first)
    row0, rows, in0, n = shards[rank]
    mine = torch.empty((n,) + tuple(like_in.shape[1:]), dtype=like_in.dtype, device=like_in.device)
    if rank == src:
        reqs = []
        for r in range(world):
            _, rr, i0, nn = shards[r]
            if rr == 0:
                continue
            if r == src:
                mine.copy_(frame_root[i0:i0 + nn])
            else:
                reqs.append(dist.isend(frame_root[i0:i0 + nn].contiguous(), dst=r))
        for q in reqs:
            q.wait()
    elif rows > 0:
        dist.recv(mine, src=src)
    out_rows = compute_strip(mine, row0, rows, in0) if rows > 0 else None
    if rank == src:
        out = torch.empty((out_h,) + tuple(like_out.shape[1:]), dtype=like_out.dtype, device=like_out.device)
        for r in range(world):
            r0, rr, _, _ = shards[r]
            if rr == 0:
                continue
            if r == src:
                out[r0:r0 + rr].copy_(out_rows)
            else:
                dist.recv(out[r0:r0 + rr], src=r)
        return out
    if rows > 0:
        dist.send(out_rows.contiguous(), dst=src)
    return None
