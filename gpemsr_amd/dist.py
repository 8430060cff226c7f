"""Multi-GPU inference: one process per GPU, tiles sharded contiguously over ranks
(tiles are fully independent in the forward: no BatchNorm, per-sample GroupNorm,
per-frame attention -- SURVEY 8(e)), weights replicated, and ONE collective per
step: an all-gather of the HR output slabs (4 MiB per tile in fp32) over
RCCL/xGMI (``backend='nccl'`` is RCCL on ROCm; ``gloo`` for the CPU tests).
The reference has no multi-GPU inference (output_GPEMSR.py is single-process);
this is the north_star's new capability."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """Returns (rank, world, local_rank); initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `total` tiles owned by `rank` (remainder spread over the first ranks)."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def volume_window_rows(n_slices: int, n_frames: int = 5):
    """Slice numbers of every output window of an n-slice volume in output order: the reference's edge rule
    (output_GPEMSR.py:54-84, 98-128) replicates the first / last slice, i.e. window k reads slices clamp(k-2 .. k+2) except
    that the two windows at each end are built from the first / last FULL window's files -- the same thing for n >= 5."""
    assert n_frames == 5 and n_slices >= 5, "the reference's edge handling is written for 5-slice windows and >= 5 slices"
    T = n_slices
    return ([[0, 0, 0, 1, 2], [0, 0, 1, 2, 3]] + [[i, i + 1, i + 2, i + 3, i + 4] for i in range(T - 4)]
            + [[T - 4, T - 3, T - 2, T - 1, T - 1], [T - 3, T - 2, T - 1, T - 1, T - 1]])


def plan_volume_shard(n_slices: int, rank: int, world: int, n_frames: int = 5):
    """z-sharding of a volume (second axis): rank r writes the contiguous output slices [lo, hi) and needs
    the LR slices [s_lo, s_hi) = its own plus a halo of n_frames // 2 = 2 slices on each interior side (none past the volume's
    ends, where the reference replicates).  Returns (lo, hi, s_lo, s_hi, rows) with rows [hi - lo, n_frames] numbering the
    window's slices RELATIVE to s_lo -- what forward_volume takes.  No data-path collective: every rank reads its own files
    and writes its own PNGs."""
    lo, hi = shard_range(n_slices, rank, world)
    rows = volume_window_rows(n_slices, n_frames)[lo:hi]
    if not rows:
        return lo, hi, 0, 0, torch.zeros(0, n_frames, dtype=torch.int32)
    s_lo, s_hi = min(min(r) for r in rows), max(max(r) for r in rows) + 1
    return lo, hi, s_lo, s_hi, torch.tensor([[v - s_lo for v in r] for r in rows], dtype=torch.int32)


def all_gather_slabs(local: torch.Tensor, world: int) -> torch.Tensor:
    """All-gather equally sized per-rank slabs [b,1,H,W] -> [world*b,1,H,W] (rank-major = tile order)."""
    if world == 1:
        return local
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local)
    return out


def all_gather_ragged(local: torch.Tensor, total: int, rank: int, world: int) -> torch.Tensor:
    """All-gather when `total` is not divisible by `world`: pad to the largest shard, gather, trim."""
    if world == 1:
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    g = all_gather_slabs(pad, world).view(world, mx, *local.shape[1:])
    return torch.cat([g[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


class _Mark:
    """A point in time on the stream the step runs on: a HIP event on the current stream (GPU), the host clock otherwise."""

    def __init__(self, cuda: bool):
        self.cuda = cuda
        if cuda:
            self.ev = torch.cuda.Event(enable_timing=True)
            self.ev.record()
        else:
            import time
            self.t = time.perf_counter()

    def ms_until(self, other: "_Mark") -> float:
        return self.ev.elapsed_time(other.ev) if self.cuda else 1e3 * (other.t - self.t)


def forward_sharded(model, x_all_or_local: torch.Tensor, rank: int, world: int, already_local: bool = False,
                    gather: bool = True, gather_u8: bool = False, timing: list = None):
    """Run the stage-3 forward on this rank's tiles and (optionally) all-gather the SR slabs.
    x: [B,N,1,H,W]; returns (out_all [B,1,sH,sW] on every rank, ref_img_local).
    gather_u8: exchange the 8-bit image the network's last kernel writes (``model(x, want_u8=True)``, the reference's tensor2img of
    SR: what output_GPEMSR.py saves) instead of the fp32 slab -- 1 MiB instead of 4 MiB per 1024^2 tile over xGMI; the first return
    value is then uint8 [B, sH, sW].
    timing: a list that receives one (start, after forward, after all-gather) triple of `_Mark`s per call (`step_phase_times`); the
    collective's mark sits on the launch stream, which waits for RCCL's stream before anything after it runs."""
    B = x_all_or_local.shape[0] * (world if already_local else 1)
    if already_local:
        x = x_all_or_local
    else:
        lo, hi = shard_range(B, rank, world)
        x = x_all_or_local[lo:hi]
    cuda = x.is_cuda
    m0 = _Mark(cuda) if timing is not None else None
    if gather_u8:
        _, ref, out = model(x, want_u8=True)
    else:
        out, ref = model(x)
    m1 = _Mark(cuda) if timing is not None else None
    if gather and world > 1:
        out = all_gather_slabs(out, world) if B % world == 0 else all_gather_ragged(out, B, rank, world)
    if timing is not None:
        timing.append((m0, m1, _Mark(cuda)))
    return out, ref


def step_phase_times(timing: list, world: int, device) -> dict:
    """Per-rank mean forward / all-gather milliseconds of the recorded steps and their max / min over ranks, so that an N-GPU line says
    whether a step is long because one rank's forward is slow (a slow device, a straggler) or because the exchange is."""
    if device.type == "cuda":
        torch.cuda.synchronize()
    n = max(len(timing), 1)
    fwd = sum(a.ms_until(b) for a, b, _ in timing) / n
    gat = sum(b.ms_until(c) for _, b, c in timing) / n
    mine = torch.tensor([fwd, gat], dtype=torch.float64, device=device)
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = torch.stack(allr).cpu()
    else:
        rows = mine.cpu().view(1, 2)
    f, g = rows[:, 0], rows[:, 1]
    return {"forward_ms_per_rank": [round(float(v), 3) for v in f], "gather_ms_per_rank": [round(float(v), 3) for v in g],
            "forward_ms_max": round(float(f.max()), 3), "forward_ms_min": round(float(f.min()), 3),
            "gather_ms_max": round(float(g.max()), 3), "gather_ms_min": round(float(g.min()), 3), "steps_recorded": len(timing)}


def average_gradients(flat_grad: torch.Tensor, world: int) -> torch.Tensor:
    """What DistributedDataParallel does for train_stage3.py:141: one all-reduce (RCCL on GPUs, gloo in the CPU tests) of the
    flat gradient buffer, then the mean over replicas.  In place; returns the buffer."""
    if world > 1:
        torch.distributed.all_reduce(flat_grad)
        flat_grad.mul_(1.0 / world)
    return flat_grad
