"""The sharded closure in REAL multi-process form on one MI355X: world_size OS processes share cuda:0 (a gpurun box
has a single GPU, and RCCL refuses two ranks on one device), the transport is torch.distributed's gloo backend on
device tensors, and everything else - StripPlan, the library's phase machine, DistFabric.apply with its cached
zero-copy views, set_targets - is exactly what `bench.py --gpus N` runs over RCCL.  Every rank's losses and the
concatenated gradient must reproduce the unsharded plan."""
import os
import socket
import sys
import traceback

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, h, w, precision, out):
    try:
        sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))
        sys.path.insert(0, os.path.join(HERE, '..', 'oracle'))
        sys.path.insert(0, HERE)
        import torch.distributed as dist
        from style_transfer import _hip as hip, sharding as sh, vgg
        import st_oracle as O
        from test_sharding_gpu import _smooth
        dev = torch.device('cuda', 0)
        torch.cuda.set_device(dev)
        dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
        weights = vgg.synthetic_vgg19_weights(0)
        content, style, image = _smooth(31, h, w), _smooth(32, h, w), _smooth(33, h, w)
        net = hip.Net(weights, 'max', dev, precision)
        b, e = sh.strip_rows(h, world)[rank]
        plan = sh.StripPlan(net, h, w, b, e).set_rank(rank, world)

        fabric = sh.DistFabric(rank, world)          # gloo: host-synchronous exchanges (DistFabric.host_sync)
        sh.set_targets(plan, content[:, :, b:e].contiguous().to(dev), [style[:, :, b:e].contiguous().to(dev)], [1.0],
                       lambda p: sh.run_phases(p, fabric), fabric.allreduce)
        plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
        img = image[:, :, b:e].contiguous().to(dev)
        grad = torch.empty_like(img)
        for _ in range(2):                       # second pass runs on the cached views / op lists
            plan.closure_begin(img, grad)
            sh.run_phases(plan, fabric)
        torch.cuda.synchronize()
        losses = plan.losses.cpu().clone()
        gathered_l = [torch.empty_like(losses) for _ in range(world)] if rank == 0 else None
        dist.gather(losses, gathered_l, dst=0)
        rows = [r1 - r0 for r0, r1 in sh.strip_rows(h, world)]
        g_cpu = grad.cpu()
        if rank == 0:
            parts = [g_cpu] + [torch.empty((1, 3, rows[r], w)) for r in range(1, world)]
            for r in range(1, world):
                dist.recv(parts[r], src=r)
            whole = hip.Plan(net, h, w)
            whole.forward(content.to(dev), 22)
            whole.set_content_target_from_forward()
            whole.forward(style.to(dev), 29)
            for i, layer in enumerate(O.STYLE_LAYERS):
                whole.set_style_target(i, *whole.moments(layer))
            whole.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
            losses_w, grad_w = whole.loss_and_grad(image.to(dev))
            losses_w, grad_w = losses_w.cpu(), grad_w.cpu()
            rel = max(float(((l - losses_w).abs() / losses_w.abs()).max()) for l in gathered_l)
            same = all(torch.equal(l, gathered_l[0]) for l in gathered_l)
            gs = torch.cat(parts, dim=2)
            err = float((gs - grad_w).double().norm() / grad_w.double().norm())
            out.put(('ok', rel, same, err))
        else:
            dist.send(g_cpu, dst=0)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:                            # noqa: BLE001 - reported to the parent
        out.put(('error', rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize('world', [2, 3])
@pytest.mark.parametrize('precision', ['fp16x3'])
def test_sharded_closure_in_separate_processes(world, precision):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 96, 80, precision, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
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
    _, rel, same, err = ok[0]
    print(f'[multiproc] R={world} {precision}: max rel loss diff {rel:.2e}, identical across ranks {same}, '
          f'gradient rel_l2 vs unsharded {err:.2e}')
    assert rel < 5e-5 and same and err < 2e-4
