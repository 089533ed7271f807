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


def test_strip_rows_balanced_for_the_chain_owner(monkeypatch):
    """ST_STRIP_BALANCE=1 (opt-in since round 6: on hardware every rank waits for relu5_1's owner, equal strips are the better
    deal - profiles/r06_strip_breakdown.md): with the width given, rank 0 - the owner of relu5_1's Newton-Schulz chains -
    gets fewer rows; small images keep the even split."""
    blocks = lambda rows: [(e - b) // 16 for b, e in rows]
    monkeypatch.delenv('ST_STRIP_BALANCE', raising=False)
    assert sharding.strip_rows(2172, 8, 2896) == sharding.strip_rows(2172, 8)        # the default: even strips
    monkeypatch.setenv('ST_STRIP_BALANCE', '1')
    rows = sharding.strip_rows(2172, 8, 2896)       # config C5 on 8 GPUs: one block moves from rank 0 to the last rank
    assert blocks(rows) == [16, 17, 17, 17, 17, 17, 17, 17] and rows[-1][1] == 2172
    assert all(rows[i][1] == rows[i + 1][0] and rows[i][1] % 16 == 0 for i in range(7))
    assert blocks(sharding.strip_rows(2172, 4, 2896)) == [33, 34, 34, 34]
    # power-of-two images: the even strips are whole tile rows of the deepest layers - kept; 2 ranks: both own a chain
    for h, w, n in [(2048, 2048, 8), (2048, 2048, 4), (4096, 4096, 8), (2172, 2896, 2)]:
        assert sharding.strip_rows(h, n, w) == sharding.strip_rows(h, n)
    for h, w, n in [(512, 512, 2), (512, 512, 8), (128, 128, 2), (256, 128, 4), (96, 80, 3)]:
        assert sharding.strip_rows(h, n, w) == sharding.strip_rows(h, n)
    for h, w, n in [(2172, 2896, 8), (2048, 2048, 4), (4096, 4096, 8), (1024, 1024, 4)]:
        rows = sharding.strip_rows(h, n, w)
        assert rows[0][0] == 0 and rows[-1][1] == h and all(e - b >= 16 for b, e in rows)
        assert sum(blocks(rows)) == h // 16
        even = blocks(sharding.strip_rows(h, n))
        assert all(abs(x - y) <= max(1, even[0] // 8) + 1 for x, y in zip(blocks(rows), even))
    monkeypatch.setenv('ST_STRIP_BALANCE', '0')
    assert sharding.strip_rows(2172, 8, 2896) == sharding.strip_rows(2172, 8)


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


def _descriptor_worker(rank, world, port):
    """DistFabric.apply on st_exchange descriptors of every kind (ABI version 2): halo (1), all-reduce (2), reduce to a
    root (4), broadcast from a root (5), nothing (3) - on CPU tensors whose addresses stand in for the library's
    buffers (`view` is patched to wrap host memory), channel 0 and channel 1."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import ctypes
        from style_transfer import _hip
        pool = {}

        def host_view(ptr, count, device):
            return pool[ptr][:count] if ptr else None
        sharding.view = host_view

        def buf(values):
            t = torch.tensor(values, dtype=torch.float32)
            pool[t.data_ptr()] = t
            return t
        fab = sharding.DistFabric(rank, world)
        n = 6
        for channel in (0, 1):
            red = buf([float(rank + 1)] * n)
            ex = _hip.Exchange(kind=4, count=n, buffer=red.data_ptr(), root=world - 1, channel=channel)
            fab.apply(ex, 'cpu')
            if rank == world - 1:
                assert torch.all(red == sum(range(1, world + 1)))
            bc = buf([float(10 * rank + 3)] * n)
            ex = _hip.Exchange(kind=5, count=n, buffer=bc.data_ptr(), root=1, channel=channel)
            fab.apply(ex, 'cpu')
            assert torch.all(bc == 13.0)
            ar = buf([1.0] * n)
            fab.apply(_hip.Exchange(kind=2, count=n, buffer=ar.data_ptr(), channel=channel), 'cpu')
            assert torch.all(ar == float(world))
            fab.apply(_hip.Exchange(kind=3), 'cpu')
            su, sd = buf([100.0 * rank + 1] * n), buf([100.0 * rank + 2] * n)
            ru, rd = buf([0.0] * n), buf([0.0] * n)
            ex = _hip.Exchange(kind=1, count=n, send_up=su.data_ptr() if rank > 0 else None,
                               send_down=sd.data_ptr() if rank < world - 1 else None,
                               recv_up=ru.data_ptr() if rank > 0 else None,
                               recv_down=rd.data_ptr() if rank < world - 1 else None, channel=channel)
            fab.apply(ex, 'cpu')
            if rank > 0:
                assert torch.all(ru == 100.0 * (rank - 1) + 2)
            if rank < world - 1:
                assert torch.all(rd == 100.0 * (rank + 1) + 1)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_exchange_descriptors_over_gloo(world):
    mp.spawn(_descriptor_worker, args=(world, _free_port()), nprocs=world, join=True)


def _lbfgs_worker(rank, world, port):
    """sharding.StripLBFGS with its inner products completed over the ranks (gloo) against torch.optim.LBFGS on the
    whole vector: same iterates up to the summation order of those inner products."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        def objective(x):
            return ((x * x).sum() + torch.log1p(x.pow(4)).sum()) / 300

        g = torch.Generator().manual_seed(0)
        x0 = torch.randn(300, generator=g)
        whole = x0.clone().requires_grad_()
        opt = torch.optim.LBFGS([whole], max_iter=1, history_size=10)

        def closure_whole():
            opt.zero_grad()
            loss = objective(whole)
            loss.backward()
            return loss
        per = 300 // world
        lo, hi = rank * per, (300 if rank == world - 1 else (rank + 1) * per)
        mine = x0[lo:hi].clone()
        grad = torch.empty_like(mine)
        fab = sharding.DistFabric(rank, world)
        strip = sharding.StripLBFGS(mine, grad, fab.allreduce, fab.allmax, history_size=10)

        def closure_strip():                          # a separable objective: the strip's terms, summed over the ranks
            with torch.enable_grad():
                xs = mine.detach().clone().requires_grad_()
                part = ((xs * xs).sum() + torch.log1p(xs.pow(4)).sum()) / 300
                part.backward()
            grad.copy_(xs.grad)
            total = part.detach().reshape(1).clone()
            fab.allreduce(total)
            return total[0]
        for i in range(15):                            # (converges to ~1e-12 after nine steps: absolute floor on the loss)
            la, lb = float(opt.step(closure_whole).detach()), float(strip.step(closure_strip))
            assert abs(la - lb) <= 1e-5 * abs(la) + 1e-9, (i, la, lb)
            assert torch.allclose(whole.detach()[lo:hi], mine, rtol=2e-3, atol=2e-5), i
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_strip_lbfgs_over_gloo(world):
    mp.spawn(_lbfgs_worker, args=(world, _free_port()), nprocs=world, join=True)
