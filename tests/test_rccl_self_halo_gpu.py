"""The RCCL point-to-point halo path on ONE GPU (round-4; replaces reference style_transfer.py:87): a middle strip whose
upper and lower neighbour is rank 0 itself (ST_FABRIC_SELF_HALO=1 - RCCL send / recv to self inside one group, i.e. a
periodic boundary).  Every halo exchange of the phase machine goes through `DistFabric.apply` kind 1 exactly as it would
between two GPUs: zero-copy views of the library's pack / halo buffers, `batch_isend_irecv` on an ExternalStream over
the library's communication stream, ordered by the library's events against the interior / boundary launches, the host
never waiting.  The reference result is the one-process emulation of the same wrap-around (device copies ordered on the
same streams): same kernels, same descriptors, so the two must agree bit for bit.  Runs in a child process with a hard
timeout, so a transport hang cannot take the suite (or the box) down."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, socket
sys.path.insert(0, os.path.join(%(root)r, 'style-transfer-pytorch_amd'))
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from bench import synthetic_image
from style_transfer import _hip, sharding, vgg

H, W = %(h)d, %(w)d
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
print('backend', dist.get_backend(), 'world', dist.get_world_size(), 'rccl', torch.cuda.nccl.version(), flush=True)
weights = vgg.synthetic_vgg19_weights(0)
net = _hip.Net(weights, 'max', dev, 'fp16x3')
content = synthetic_image(100, H, W).to(dev)
style = synthetic_image(200, H, W).to(dev)
image0 = synthetic_image(300, H, W).to(dev)

def run(transport):
    # a MIDDLE strip of a fake 3H-row image: both neighbours exist, and both are this rank
    plan = sharding.StripPlan(net, 3 * H, W, H, 2 * H).set_rank(0, 1)
    sharding.set_targets(plan, content, [style], [1.0], transport, lambda t: None)
    plan.set_loss_weights(0.015, [w / 341 for w in (256, 64, 16, 4, 1)], 2.0)
    image = image0.clone()
    grad = torch.empty_like(image)
    m, v = torch.zeros_like(image), torch.zeros_like(image)
    ema = torch.zeros_like(image)
    first = None
    for it in range(1, 4):                       # repeated exchanges: buffers are reused, nothing may deadlock
        plan.closure_begin(image, grad)
        transport(plan)
        if first is None:
            torch.cuda.synchronize()
            first = (plan.losses.clone(), grad.clone(), plan.feature(1), plan.feature(20), plan.feature(29))
        plan.apply_update(image, grad, m, v, ema, it, 0.02)
    torch.cuda.synchronize()
    return first, plan.losses.clone(), image.clone()

fabric = sharding.DistFabric(0, 1)
assert fabric.self_halo and not fabric.host_sync, 'self-halo over RCCL must run stream-ordered'
a = run(lambda p: sharding.run_phases(p, fabric))
n_p2p = sum(1 for k in fabric._cache if k[0] == 1)
b = run(lambda p: sharding.run_phases_lockstep([p], wrap=True))
# the in-library transport (csrc/st_fabric.hip: ncclSend / ncclRecv from C++ on the library's communication stream, the
# whole phase sequence in one call)
native = sharding.NativeFabric(0, 1, dev, cold=fabric)
assert native.self_halo
nat = run(lambda p: sharding.run_phases(p, native))
# round 6: exchanges whose consumer is not cut are issued in line on the caller's stream (st_exchange.stream = null); with
# ST_STRIP_INLINE=0 every exchange travels on the communication stream as in rounds 4 / 5 - both forms, both transports
with _hip.options(ST_STRIP_INLINE=0):
    a0 = run(lambda p: sharding.run_phases(p, fabric))
    nat0 = run(lambda p: sharding.run_phases(p, native))
for name, x, y, z in zip(names_all := ('losses', 'grad', 'relu1_1', 'relu4_1', 'relu5_1'), a0[0], nat0[0], b[0]):
    assert torch.equal(x, z) and torch.equal(y, z), 'ST_STRIP_INLINE=0: ' + name
assert torch.equal(a0[2], b[2]) and torch.equal(nat0[2], b[2]), 'ST_STRIP_INLINE=0: three iterations diverged'
print('[self-halo] every exchange on the communication stream (ST_STRIP_INLINE=0): bit-identical too', flush=True)
# and with NO exchange at all the result must differ: the comparison above is not vacuous
c = run(lambda p: sharding.run_phases_lockstep([p], stub=True))
names = ('losses', 'grad', 'relu1_1', 'relu4_1', 'relu5_1')
for name, x, y in zip(names, a[0], b[0]):
    d = float((x - y).abs().max())
    print(f'[self-halo] first closure {name}: max abs diff RCCL vs emulation {d:.3e}', flush=True)
    assert torch.equal(x, y), name
assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), 'three iterations diverged'
for name, x, y in zip(names, nat[0], b[0]):
    assert torch.equal(x, y), 'in-library transport: ' + name
assert torch.equal(nat[1], b[1]) and torch.equal(nat[2], b[2]), 'in-library transport: three iterations diverged'
print('[self-halo] in-library transport (st_plan_closure_run over st_fabric): bit-identical too', flush=True)
assert not torch.equal(a[0][1], c[0][1]), 'halos made no difference: the test is vacuous'
print(f'[self-halo] OK: {n_p2p} distinct P2P descriptors over RCCL, loss after 3 iterations {float(a[1][7]):.6f}', flush=True)
native.close()
# the pre-flight's failure path: a deadline nothing can meet -> an exception the caller can act on (stylize() and bench.py
# fall back to torch.distributed), the half-used communicators aborted, and a fresh fabric still works afterwards
os.environ['ST_FABRIC_SELFTEST_MS'] = '-1'
try:
    sharding.NativeFabric(0, 1, dev, cold=fabric)
    raise SystemExit('the pre-flight accepted an impossible deadline')
except RuntimeError as exc:
    assert 'did not complete' in str(exc), exc
    print('[self-halo] pre-flight failure path:', str(exc)[:120], flush=True)
del os.environ['ST_FABRIC_SELFTEST_MS']
torch.cuda.synchronize()
again = sharding.NativeFabric(0, 1, dev, cold=fabric)
again.close()
print('[self-halo] a fresh fabric after the aborted one: pre-flight passed', flush=True)
dist.destroy_process_group()
'''


@pytest.mark.parametrize('h,w', [(64, 80), (272, 256)])
def test_halo_exchange_to_self_over_rccl(h, w):
    env = dict(os.environ, ST_FABRIC_SELF_HALO='1', ST_FABRIC_FORCE_COLLECTIVES='1')
    r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT, 'h': h, 'w': w}], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    print(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert '[self-halo] OK' in r.stdout and 'backend nccl' in r.stdout
