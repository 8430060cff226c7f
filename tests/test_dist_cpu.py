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


def _volume8_worker(rank, world, port, n_slices, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from gpemsr_amd.dist import plan_volume_shard, volume_window_rows
    lo, hi, s_lo, s_hi, rows = plan_volume_shard(n_slices, rank, world)
    vol = torch.arange(n_slices, dtype=torch.float32)                    # "slice k" = the number k
    local = vol[s_lo:s_hi]                                               # what this rank reads from disk: its slices + halo
    got = local[rows.long()] if rows.numel() else torch.zeros(0, 5)
    want = vol[torch.tensor(volume_window_rows(n_slices)[lo:hi], dtype=torch.long)] if hi > lo else torch.zeros(0, 5)
    cnt = torch.tensor([hi - lo, s_hi - s_lo], dtype=torch.int64)
    dist.all_reduce(cnt)                                                 # every rank took part; totals checked by the parent
    q.put((rank, lo, hi, s_lo, s_hi, bool(torch.equal(got, want)), int(cnt[0]), int(cnt[1])))
    dist.destroy_process_group()


def test_ragged_125_slice_volume_over_eight_gloo_ranks():
    """CREMI's 125-slice volumes on the 8 ranks of one node: 125 = 8 x 15 + 5, so five ranks write 16 slices and three write 15; every
    rank's LR read set is its slices plus a 2-slice halo on interior sides only, and windows indexed relative to that read set are the
    reference's windows (R:output_GPEMSR.py:54-84,98-128)."""
    world, n = 8, 125
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_volume8_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(30)
    assert [r[2] - r[1] for r in res] == [16] * 5 + [15] * 3
    assert res[0][1] == 0 and res[-1][2] == n and all(res[i][2] == res[i + 1][1] for i in range(world - 1))
    assert all(r[5] for r in res)
    assert all(r[6] == n for r in res)
    for r in res:
        rank, lo, hi, s_lo, s_hi = r[:5]
        assert s_lo == max(lo - 2, 0) and s_hi == min(hi + 2, n)
    assert res[0][7] == n + 2 * 2 * (world - 1)                           # the halo costs 4 extra slice reads per interior boundary
