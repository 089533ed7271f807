"""`StyleTransfer.stylize()` under torch.distributed (SURVEY.md 8(f) 1-2 made shard-aware): world_size OS processes
share cuda:0 over gloo (a gpurun box has one GPU, and RCCL refuses two ranks on one device); the same call on
N GPUs runs over RCCL.  Covers: a first scale too small for strips (runs whole on every rank), strip-sharded
scales, the shard-aware scale transition (whole -> strips, strips -> strips; every rank resamples its own rows,
neighbour rows point to point), a style image with its own strip
geometry, a style image too small to cut (evaluated whole on every rank), weighted multi-style blending, and the
averaged-iterate hand-off.  The result must be identical on every rank and match the single-process run of the
same call up to the summation order of the Gram / loss partial sums."""
import os
import time
import socket
import sys
import traceback

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _pil(seed, h, w):
    from PIL import Image
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((1, 3, max(h // 12, 2), max(w // 12, 2)), generator=g)
    img = torch.nn.functional.interpolate(low, (h, w), mode='bicubic', align_corners=False)
    img = (img + (torch.rand((1, 3, h, w), generator=g) - 0.5) * 0.1).clamp(0, 1)
    return Image.fromarray((img[0].permute(1, 2, 0).numpy() * 255).round().astype(np.uint8), 'RGB')


KW = dict(style_weights=[0.7, 0.3], min_scale=24, end_scale=96, iterations=5, initial_iterations=6)


def _worker(rank, world, port, out, extra=None):
    try:
        sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))
        import torch.distributed as dist
        import style_transfer as st_pkg
        from style_transfer import vgg
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        weights = vgg.synthetic_vgg19_weights(0)
        kw = dict(KW, **(extra or {}))
        if kw.pop('_dominant_filters', False):
            # the network of tests/test_range_guard_gpu.py: one filter per non-tap layer at 2^22, compensated in the consumer
            sys.path.insert(0, HERE)
            sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))
            from test_range_guard_gpu import _dominant_filters
            weights = _dominant_filters(weights)
        if kw.pop('_unequal_strips', False):
            # what sharding.strip_rows(height, world, width) does at BASELINE sizes (the owner of relu5_1's chains gets a shorter
            # strip), forced at test sizes: the optimised image's first strip hands a 16-row block to the last one
            from style_transfer import sharding
            even = sharding.strip_rows

            def unequal(height, world, width=None):
                rows = even(height, world)
                if width is None or rows[0][1] - rows[0][0] < 32:
                    return rows
                cuts = [0] + [e - 16 for _, e in rows[:-1]] + [height]
                return [(cuts[r], cuts[r + 1]) for r in range(world)]
            sharding.strip_rows = unequal
        content, styles = _pil(1, 96, 80), [_pil(2, 120, 90), _pil(3, 28, 40)]
        trace = []
        st = st_pkg.StyleTransfer(devices=['cuda:0'], weights=weights)
        st.stylize(content, styles, callback=lambda it: trace.append((it.w, it.h, it.i, it.loss)), **kw)
        result = st.get_image_tensor().cpu()
        torch.cuda.synchronize()
        wide = st.model.net.wide_layers()
        flags = torch.tensor(wide[0] + wide[1], dtype=torch.float32)
        every = [torch.empty_like(flags) for _ in range(world)]
        dist.all_gather(every, flags)
        assert all(torch.equal(f, every[0]) for f in every), 'the ranks disagree on the bf16x6 layers'
        gathered = [torch.empty_like(result) for _ in range(world)] if rank == 0 else None
        dist.gather(result, gathered, dst=0)
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            same = all(torch.equal(g, gathered[0]) for g in gathered)
            trace1 = []
            st1 = st_pkg.StyleTransfer(devices=['cuda:0'], weights=weights)     # no process group: single-GPU path
            st1.stylize(content, styles, callback=lambda it: trace1.append((it.w, it.h, it.i, it.loss)), **kw)
            want = st1.get_image_tensor().cpu()
            diff = (result - want).abs()
            out.put(('ok', same, float(diff.mean()), float(diff.max()), trace, trace1, tuple(result.shape), wide,
                     st1.model.net.wide_layers()))
    except Exception:                            # noqa: BLE001 - reported to the parent
        out.put(('error', rank, traceback.format_exc()))
        raise


def _run_ranks(world, extra=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, extra)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=420)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()                                 # exact handles of the processes started above
    assert not alive, 'a rank hung'
    results = []
    while not out.empty():
        results.append(out.get())
    errors = [r for r in results if r[0] == 'error']
    assert not errors, errors[0][2]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    ok = [r for r in results if r[0] == 'ok']
    assert len(ok) == 1
    return ok[0]


@pytest.mark.parametrize('world', [2, 3])
def test_stylize_in_separate_processes_matches_single_gpu(world):
    _, same, mean_abs, max_abs, trace, trace1, shape, wide, wide1 = _run_ranks(world)
    print('[stylize-sharded] wide layers sharded', wide, 'single', wide1)
    assert wide == wide1, 'the sharded guard flags what the single-process guard flags on this network'
    assert shape == (3, 96, 80)
    assert [t[:3] for t in trace] == [t[:3] for t in trace1], 'same scales and iteration counts'
    sizes = sorted({(t[0], t[1]) for t in trace})
    rel = max(abs(a[3] - b[3]) / abs(b[3]) for a, b in zip(trace, trace1))
    print(f'[stylize-sharded] R={world}: scales {sizes}, identical across ranks {same}, image mean_abs {mean_abs:.2e} '
          f'max_abs {max_abs:.2e}, max rel loss-trace diff {rel:.2e}')
    assert same, 'every rank must hold the same gathered result'
    # Adam normalises the gradient, so summation-order noise in near-zero gradients moves single pixels by up to
    # ~lr per iteration; the images must still agree closely on average and the loss traces track each other
    assert mean_abs < 1e-4 and rel < 1e-3          # measured: 6e-8 ... 8e-7 and 4e-5 ... 8e-5


def test_stylize_with_strips_of_unequal_height_matches_single_gpu():
    """The last scale (96 rows, 3 ranks) runs on strips of 16 / 32 / 48 rows after a scale of 3 x 16: the shard-aware scale
    transition, the targets and the closure must not assume the even split."""
    _, same, mean_abs, max_abs, trace, trace1, shape, wide, wide1 = _run_ranks(3, {'_unequal_strips': True})
    rel = max(abs(a[3] - b[3]) / abs(b[3]) for a, b in zip(trace, trace1))
    print(f'[stylize-sharded] unequal strips R=3: identical across ranks {same}, image mean_abs {mean_abs:.2e} max_abs {max_abs:.2e}, '
          f'max rel loss-trace diff {rel:.2e}')
    assert same and shape == (3, 96, 80) and [t[:3] for t in trace] == [t[:3] for t in trace1]
    assert mean_abs < 1e-4 and rel < 1e-3


def test_stylize_lbfgs_in_separate_processes_matches_single_gpu():
    """optimizer='lbfgs' on strips (reference style_transfer.py:464-465: torch.optim.LBFGS(max_iter=1, history_size=10)):
    sharding.StripLBFGS completes every inner product of torch's recursion over the ranks.  Compared with the
    single-process run (torch.optim.LBFGS itself over the unsharded plan); the quasi-Newton recursion amplifies the
    summation-order difference of those inner products like any rounding-level change (the reference's own trace moves
    by up to 2e-2 after seven iterations between 1 and 8 threads), so the bar is a loss trace within 5e-2 and images
    that agree on average - and bit-identical results on every rank."""
    _, same, mean_abs, max_abs, trace, trace1, shape, _, _ = _run_ranks(2, dict(optimizer='lbfgs', iterations=3, initial_iterations=4))
    assert [t[:3] for t in trace] == [t[:3] for t in trace1], 'same scales and iteration counts'
    rels = [abs(a[3] - b[3]) / abs(b[3]) for a, b in zip(trace, trace1)]
    print(f'[stylize-sharded] lbfgs R=2: identical across ranks {same}, image mean_abs {mean_abs:.2e} max_abs {max_abs:.2e}, '
          f'loss-trace rel diffs {["%.1e" % r for r in rels]}')
    assert same, 'every rank must hold the same gathered result'
    assert max(rels) < 5e-2 and mean_abs < 5e-3


def test_sharded_range_guard_flags_on_every_rank():
    """The activation-aware fp16x3 range guard under strip sharding (round 4): every rank checks its own rows as an image of
    their own (StyleTransfer._guard_rows), the ranks take the union (st_net_mark_wide).  Network: one dominant filter (2^22)
    per non-tap layer, invisible to the weights-only rule - without the guard the losses are off by orders of magnitude
    (tests/test_range_guard_gpu.py).  The sharded run must flag layers, identically on every rank, and follow the
    single-process guarded run like the benchmark network does."""
    # a single, sharded scale: nothing is flagged by an earlier whole-image scale
    _, same, mean_abs, max_abs, trace, trace1, shape, wide, wide1 = _run_ranks(
        2, dict(_dominant_filters=True, min_scale=96, end_scale=96, initial_iterations=8))
    assert {(t[0], t[1]) for t in trace} == {(80, 96)}
    rel = max(abs(a[3] - b[3]) / abs(b[3]) for a, b in zip(trace, trace1))
    print(f'[stylize-sharded] dominant filters R=2: bf16x6 forward {[i for i, v in enumerate(wide[0]) if v]} / data gradient '
          f'{[i for i, v in enumerate(wide[1]) if v]} (single process: {[i for i, v in enumerate(wide1[0]) if v]} / '
          f'{[i for i, v in enumerate(wide1[1]) if v]}); image mean_abs {mean_abs:.2e}, max rel loss-trace diff {rel:.2e}')
    assert any(wide[0]) and any(wide1[0]), 'the guard must have flagged forward layers'
    assert same
    assert mean_abs < 1e-4 and rel < 1e-3


def _device_list_case(out):
    """In a process of its own (the device-list form creates and destroys a process group of its own)."""
    try:
        sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))
        import style_transfer as st_pkg
        from style_transfer import vgg
        weights = vgg.synthetic_vgg19_weights(0)
        content, styles = _pil(1, 96, 80), [_pil(2, 120, 90), _pil(3, 28, 40)]
        trace, shapes = [], []
        st = st_pkg.StyleTransfer(devices=['cuda:0', 'cuda:0'], weights=weights)       # one GPU named twice: two ranks over gloo

        def callback(it):
            trace.append((it.w, it.h, it.i, it.loss))
            if it.i % 3 == 0:
                shapes.append(tuple(st.get_image_tensor().shape))     # cli.py:124-133: the callback reads the image (a gather)
        pil = st.stylize(content, styles, callback=callback, **KW)
        result = st.get_image_tensor().cpu()
        import torch.distributed as dist
        assert not dist.is_initialized(), 'the job must leave no process group behind'
        st.stylize(content, styles, **dict(KW, iterations=2, initial_iterations=2))     # ... and can be called again, without a callback
        # cli.py:261-266: Ctrl-C between two iterations keeps what has been computed - here the workers gather once more and
        # leave with rank 0 (a sharded scale: 96 x 80 on two ranks)
        seen = []

        def interrupting(it):
            seen.append((it.w, it.h))
            if (it.w, it.h) == (80, 96) and it.i == 2:
                raise KeyboardInterrupt
        try:
            st.stylize(content, styles, callback=interrupting, **KW)
            raise AssertionError('the interrupt was swallowed')
        except KeyboardInterrupt:
            pass
        kept = st.get_image_tensor()
        assert tuple(kept.shape) == (3, 96, 80) and float(kept.min()) >= 0 and not dist.is_initialized(), kept.shape
        out.put(('ok', result.numpy(), trace, shapes, pil.size))      # (by value: a tensor's shared storage dies with this process)
    except Exception:                            # noqa: BLE001 - reported to the parent
        out.put(('error', traceback.format_exc()))
        raise


def test_device_list_in_one_process_matches_the_launcher_form():
    """StyleTransfer(devices=[d0, d1]).stylize(...) in ONE process (the reference's two-device call, style_transfer.py:326-333,
    needs no launcher): the extra ranks are spawned by the call itself.  With cuda:0 named twice the two ranks share the GPU
    over gloo - the emulation every multi-process test here uses.  The result must be BIT-IDENTICAL to the launcher form
    (two separately started ranks, _run_ranks): it is the same code on the same strips."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_device_list_case, args=(out,))
    p.start()
    got = out.get(timeout=600)
    p.join(timeout=120)
    assert got[0] == 'ok', got[1]
    _, result, trace, shapes, pil_size = got
    assert pil_size == (80, 96) and tuple(result.shape) == (3, 96, 80)
    assert shapes and all(s[0] == 3 for s in shapes), 'the callback read the image while strips were distributed'
    # the launcher form of the same call
    import torch.multiprocessing  # noqa: F401
    out2 = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_launcher_rank, args=(r, 2, port, out2)) for r in range(2)]
    for q in procs:
        q.start()
    want = out2.get(timeout=600)
    for q in procs:
        q.join(timeout=120)
    assert want[0] == 'ok', want[1]
    assert [t[:3] for t in trace] == [t[:3] for t in want[2]], 'same scales and iteration counts'
    print(f'[device list] max abs difference to the launcher form {float(np.abs(result - want[1]).max()):.2e}; '
          f'loss traces equal: {trace == want[2]}')
    assert np.array_equal(result, want[1]) and trace == want[2]


def _device_list_sigint_case(out, inside_callback):
    """A REAL SIGINT to the whole process group (what a terminal's Ctrl-C does): the workers must ignore it, rank 0 salvages."""
    try:
        import signal
        import threading
        os.setpgrp()                                 # this process and the workers it spawns: a process group of their own
        sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))
        import style_transfer as st_pkg
        from style_transfer import vgg
        import torch.distributed as dist
        content, styles = _pil(1, 96, 80), [_pil(2, 120, 90), _pil(3, 28, 40)]
        st = st_pkg.StyleTransfer(devices=['cuda:0', 'cuda:0'], weights=vgg.synthetic_vgg19_weights(0))
        seen = []

        def callback(it):
            seen.append((it.w, it.h, it.i))
            if (it.w, it.h) == (80, 96) and it.i == 2:                 # a sharded scale (96 x 80 on two ranks)
                if inside_callback:
                    os.killpg(os.getpgrp(), signal.SIGINT)
                    time.sleep(0.5)                                    # (delivered here, inside the callback)
                else:                                                  # ... or between two callbacks, while rank 0 iterates
                    threading.Timer(0.002, lambda: os.killpg(os.getpgrp(), signal.SIGINT)).start()
        try:
            st.stylize(content, styles, callback=callback, **dict(KW, iterations=40))
            raise AssertionError('the interrupt was swallowed')
        except KeyboardInterrupt:
            pass
        kept = st.get_image_tensor()
        assert tuple(kept.shape) == (3, 96, 80) and float(kept.min()) >= 0 and float(kept.max()) <= 1, kept.shape
        assert not dist.is_initialized()
        last = seen[-1]
        assert last[:2] == (80, 96) and last[2] < 40, f'stopped at {last}: the run must end early, inside the sharded scale'
        st.stylize(content, styles, **dict(KW, iterations=2, initial_iterations=2))     # the object is usable again
        out.put(('ok', last))
    except BaseException:                        # noqa: BLE001 - reported to the parent
        out.put(('error', traceback.format_exc()))
        raise


@pytest.mark.parametrize('inside_callback', [True, False])
def test_device_list_survives_a_real_sigint_to_the_process_group(inside_callback):
    """Advisor finding of round 5: a terminal's Ctrl-C reaches every process of the foreground group, so the workers used to
    die mid-collective while rank 0 tried to salvage the averaged iterate (reference cli.py:261-266 keeps the image).  Now
    the workers ignore SIGINT and rank 0 drives the stop - whether the signal lands inside the callback or between two."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_device_list_sigint_case, args=(out, inside_callback))
    p.start()
    got = out.get(timeout=600)
    p.join(timeout=120)
    assert got[0] == 'ok', got[1]
    print(f'[device list] SIGINT to the group ({"inside" if inside_callback else "between"} callbacks): stopped at {got[1]}')
    assert p.exitcode == 0


def _device_list_dead_worker_case(out):
    try:
        sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))
        import style_transfer as st_pkg
        from style_transfer import vgg
        os.environ['ST_DEVICE_LIST_INJECT_FAILURE'] = '1'
        os.environ['ST_DEVICE_LIST_TIMEOUT'] = '60'
        st = st_pkg.StyleTransfer(devices=['cuda:0', 'cuda:0'], weights=vgg.synthetic_vgg19_weights(0))
        t0 = time.time()
        try:
            st.stylize(_pil(1, 96, 80), [_pil(2, 120, 90)], **KW)
            out.put(('error', 'a dead worker went unnoticed'))
        except RuntimeError as exc:
            out.put(('ok', str(exc), time.time() - t0))
    except BaseException:                        # noqa: BLE001
        out.put(('error', traceback.format_exc()))
        raise


def test_device_list_reports_a_worker_that_died():
    """Advisor finding of round 5: rank 0 must not wait for a worker that is gone; the caller gets the worker's traceback."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_device_list_dead_worker_case, args=(out,))
    p.start()
    got = out.get(timeout=300)
    p.join(timeout=120)
    assert got[0] == 'ok', got[1]
    print(f'[device list] dead worker reported after {got[2]:.1f} s: {got[1][:120]!r}')
    assert 'injected worker failure' in got[1] and got[2] < 90


def _launcher_rank(rank, world, port, out):
    try:
        sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))
        import torch.distributed as dist
        import style_transfer as st_pkg
        from style_transfer import vgg
        torch.cuda.set_device(0)
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        content, styles = _pil(1, 96, 80), [_pil(2, 120, 90), _pil(3, 28, 40)]
        trace = []
        st = st_pkg.StyleTransfer(devices=['cuda:0'], weights=vgg.synthetic_vgg19_weights(0))
        st.stylize(content, styles, callback=lambda it: trace.append((it.w, it.h, it.i, it.loss)), **KW)
        result = st.get_image_tensor().cpu()
        dist.barrier()
        dist.destroy_process_group()
        if rank == 0:
            out.put(('ok', result.numpy(), trace))
    except Exception:                            # noqa: BLE001
        if rank == 0:
            out.put(('error', traceback.format_exc()))
        raise
