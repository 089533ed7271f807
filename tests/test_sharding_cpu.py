"""Multi-GPU path, CPU side: the strip planner and the torch.distributed transport (gloo, world sizes
2 and 3, CPU tensors).  The transport code is the one bench.py runs with backend nccl (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from style_transfer import sharding


def test_strip_rows():
    rows = sharding.strip_rows(2172, 8)             # 135 blocks + 12 rows
    assert rows[0] == (0, 272) and rows[-1] == (1904, 2172)
    assert [(e - b) // 16 for b, e in rows] == [17, 17, 17, 17, 17, 17, 17, 16]
    assert all(b % 16 == 0 for b, _ in rows) and all(e % 16 == 0 for _, e in rows[:-1])
    assert all(rows[i][1] == rows[i + 1][0] for i in range(7))
    assert sharding.strip_rows(512, 1) == [(0, 512)]
    assert sharding.strip_rows(2048, 4) == [(0, 512), (512, 1024), (1024, 1536), (1536, 2048)]
    assert sharding.strip_rows(135, 2) == [(0, 64), (64, 135)]
    with pytest.raises(ValueError):
        sharding.strip_rows(40, 3)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        fab = sharding.DistFabric(rank, world)
        for it in range(3):                                   # repeated exchanges must not deadlock
            up = torch.full((n,), 100.0 * rank + 1 + it) if rank > 0 else None
            down = torch.full((n,), 100.0 * rank + 2 + it) if rank < world - 1 else None
            r_up = torch.zeros(n) if rank > 0 else None
            r_down = torch.zeros(n) if rank < world - 1 else None
            fab.halo_exchange(up, down, r_up, r_down)
            if rank > 0:                                      # got the upper neighbour's "down" rows
                assert torch.all(r_up == 100.0 * (rank - 1) + 2 + it)
            if rank < world - 1:                              # got the lower neighbour's "up" rows
                assert torch.all(r_down == 100.0 * (rank + 1) + 1 + it)
        t = torch.arange(5, dtype=torch.float32) + rank
        fab.allreduce(t)
        assert torch.allclose(t, torch.arange(5, dtype=torch.float32) * world + sum(range(world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_fabric_over_gloo(world):
    mp.spawn(_worker, args=(world, _free_port(), 257), nprocs=world, join=True)


def _gather_worker(rank, world, port, h):
    """Scale-transition hand-off of the sharded stylize(): strips of DIFFERENT heights -> the full tensor on every
    rank, which is then resampled and cut again exactly like the single-GPU path."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from style_transfer.style_transfer import _dist_info, _gather_rows, interpolate
        assert _dist_info() == (rank, world)
        full = torch.arange(3 * h * 7, dtype=torch.float32).reshape(1, 3, h, 7)
        rows = sharding.strip_rows(h, world)
        b, e = rows[rank]
        got = _gather_rows(full[:, :, b:e].contiguous(), rows, rank)
        assert torch.equal(got, full)
        # resample + cut: concatenating every rank's new strip reproduces the resampled full tensor
        new_h = 2 * h
        big = interpolate(got, (new_h, 14), mode='bicubic')
        nrows = sharding.strip_rows(new_h, world)
        nb, ne = nrows[rank]
        back = _gather_rows(big[:, :, nb:ne].contiguous(), nrows, rank)
        assert torch.equal(back, big)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,h', [(2, 135), (3, 100)])
def test_scale_transition_gather_over_gloo(world, h):
    mp.spawn(_gather_worker, args=(world, _free_port(), h), nprocs=world, join=True)


def _resample_worker(rank, world, port, sizes):
    """Shard-aware scale transition: every rank resamples only its own rows (neighbour rows point to point) and the
    concatenated strips reproduce F.interpolate of the full tensor - through a whole / sharded / sharded chain of
    scales like stylize() runs them, for the image (bicubic) and the second Adam moment (bilinear)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from style_transfer.style_transfer import _gather_rows, interpolate
        for mode in ('bicubic', 'bilinear'):
            gen = torch.Generator().manual_seed(7)
            h0, w0 = sizes[0]
            full = torch.rand(1, 3, h0, w0, generator=gen)
            strip, rows = full, None                               # first scale: held whole on every rank
            for (h1, w1) in sizes[1:]:
                want = interpolate(full, (h1, w1), mode=mode)
                nrows = sharding.strip_rows(h1, world)
                got = sharding.resample_strip(strip, rows, rank, world, full.shape[2], nrows, (h1, w1), mode)
                nb, ne = nrows[rank]
                assert got.shape == want[:, :, nb:ne].shape
                err = (got - want[:, :, nb:ne]).abs().max().item()
                assert err <= 4e-6, (mode, (h1, w1), rank, err)         # fp32 rounding of a 16-tap sum, other order
                assert torch.allclose(_gather_rows(got, nrows, rank), want, atol=4e-6, rtol=0)
                full, strip, rows = want, want[:, :, nb:ne].contiguous(), nrows
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world,sizes', [
    (2, [(23, 31), (45, 61), (64, 87), (181, 241)]),       # whole -> strips -> strips (x1.41 steps like the scales)
    (3, [(48, 40), (68, 57), (96, 80), (96, 80)]),         # incl. an identity transition
    (2, [(96, 64), (40, 30)]),                             # downsampling needs rows far from the strip border
])
def test_shard_aware_resample_over_gloo(world, sizes):
    mp.spawn(_resample_worker, args=(world, _free_port(), sizes), nprocs=world, join=True)


def test_resample_rows_match_aten_taps():
    """The H taps restated in sharding.py are ATen's: one rank holding everything reproduces F.interpolate."""
    from torch.nn import functional as F
    x = torch.rand(1, 2, 37, 29, generator=torch.Generator().manual_seed(3))
    for mode in ('bicubic', 'bilinear'):
        for size in [(52, 41), (37, 29), (111, 90), (20, 16)]:
            got = sharding.resample_strip(x, None, 0, 1, 37, [(0, size[0])], size, mode)
            want = F.interpolate(x, size, mode=mode)
            assert (got - want).abs().max().item() <= 4e-6, (mode, size)
    got = sharding.resample_strip(x, None, 0, 1, 37, [(0, 37)], (37, 29), 'bicubic')
    assert torch.equal(got, x)                              # same size: the exact identity


def test_dist_info_without_process_group():
    from style_transfer.style_transfer import _dist_info
    assert _dist_info() == (0, 1)
