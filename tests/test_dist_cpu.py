"""CPU, 2 ranks, gloo: the N>1 path of bench.py / output_GPEMSR.py (tile sharding + all-gather of HR slabs)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _FakeModel:
    """Stands in for the HIP model on CPU: a per-tile function, so sharded == unsharded iff sharding is right."""
    def __call__(self, x):
        b = x.shape[0]
        out = x[:, 2].mean(dim=(1, 2, 3)).view(b, 1, 1, 1).expand(b, 1, 4, 4) + x[:, 2, :, :4, :4]
        return out.contiguous(), x.clone()


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gpemsr_amd import dist as gdist
    r, w, _ = gdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    x_all = torch.rand(total, 5, 1, 8, 8, generator=g)
    model = _FakeModel()
    want, _ = model(x_all)
    got, ref_local = gdist.forward_sharded(model, x_all, rank, world)           # ragged or even
    lo, hi = gdist.shard_range(total, rank, world)
    ok = torch.equal(got, want) and ref_local.shape[0] == hi - lo
    if total % world == 0:      # bench.py's weak-scaling form: every rank already holds its own tiles
        got2, _ = gdist.forward_sharded(model, x_all[lo:hi], rank, world, already_local=True)
        ok = ok and torch.equal(got2, want)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t) == float(world)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


@pytest.mark.parametrize("total", [8, 7])
def test_sharded_forward_allgather_gloo(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def _volume_worker(rank, world, port, n_slices, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from gpemsr_amd import dist as gdist
    gdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(1)
    vol = torch.rand(n_slices, 1, 4, 4, generator=g)                         # the whole LR volume (every rank can index it)
    lo, hi, s_lo, s_hi, rows = gdist.plan_volume_shard(n_slices, rank, world)
    local = vol[s_lo:s_hi]                                                   # what this rank reads: own slices + halo
    # a stand-in for forward_volume: window k -> weighted sum of its five slices (order matters: catches a wrong halo index)
    wts = torch.tensor([1.0, 2.0, 4.0, 8.0, 16.0]).view(1, 5, 1, 1, 1)
    out_local = (local[rows.long()] * wts).sum(1)                            # [hi-lo, 1, 4, 4]
    full_rows = torch.tensor(gdist.volume_window_rows(n_slices))
    want = (vol[full_rows] * wts).sum(1)
    ok = torch.equal(out_local, want[lo:hi])
    ok = ok and (s_hi - s_lo) <= (hi - lo) + 4 and s_lo >= max(0, lo - 2) and s_hi <= min(n_slices, hi + 2)
    got = gdist.all_gather_ragged(out_local, n_slices, rank, world)           # (the CLI writes PNGs instead; gathered here to compare)
    ok = ok and torch.equal(got, want)
    dist.barrier(); dist.destroy_process_group()
    q.put((rank, bool(ok), (lo, hi, s_lo, s_hi)))


@pytest.mark.parametrize("n_slices", [7, 12])
def test_volume_z_sharding_with_halo_gloo(n_slices):
    """output_GPEMSR.py volume mode over 2 ranks: contiguous output slices per rank, LR slices = own + 2-slice halo on the
    interior side, window tables relative to the local slice block; the ranks' outputs tile the volume exactly."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_volume_worker, args=(r, world, port, n_slices, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    (lo0, hi0, s0, e0), (lo1, hi1, s1, e1) = res[0][2], res[1][2]
    assert lo0 == 0 and hi0 == lo1 and hi1 == n_slices
    assert s0 == 0 and e0 == min(n_slices, hi0 + 2) and s1 == max(0, lo1 - 2) and e1 == n_slices     # the halos overlap by 4 slices


def test_volume_shard_plan_every_world_size():
    from gpemsr_amd import dist as gdist
    for n in (5, 6, 9, 125):
        rows = gdist.volume_window_rows(n)
        assert len(rows) == n and all(len(r) == 5 for r in rows) and rows[2] == [0, 1, 2, 3, 4] if n >= 5 else True
        for world in (1, 2, 3, 4, 8):
            covered = []
            for rank in range(world):
                lo, hi, s_lo, s_hi, rel = gdist.plan_volume_shard(n, rank, world)
                covered += list(range(lo, hi))
                for k in range(hi - lo):
                    assert [int(v) + s_lo for v in rel[k]] == rows[lo + k]
                    assert all(0 <= int(v) < s_hi - s_lo for v in rel[k])
            assert covered == list(range(n))
