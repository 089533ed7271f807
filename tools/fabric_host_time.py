"""Host time of the strip closure's exchanges over the REAL transport, on one GPU: a middle strip whose neighbours are rank 0
itself (ST_FABRIC_SELF_HALO=1, RCCL send / recv to self), timed with the fabric and with the exchanges stubbed.
    ST_FABRIC_SELF_HALO=1 ST_FABRIC_FORCE_COLLECTIVES=1 python tools/fabric_host_time.py [rows] [width]"""
import os, sys, time, socket
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch, torch.distributed as dist
from bench import synthetic_image
from style_transfer import _hip, sharding, vgg

H = int(sys.argv[1]) if len(sys.argv) > 1 else 272
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2896
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
with socket.socket() as s:
    s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1, device_id=dev)
net = _hip.Net(vgg.synthetic_vgg19_weights(0), 'max', dev, 'fp16x3')
content, style, image0 = (synthetic_image(100 + i, H, W).to(dev) for i in range(3))
fabric = sharding.DistFabric(0, 1)
native = sharding.NativeFabric(0, 1, dev, cold=fabric)
plan = sharding.StripPlan(net, 3 * H, W, H, 2 * H).set_rank(0, 1)
run_f = lambda p: sharding.run_phases(p, fabric)
run_s = lambda p: sharding.run_phases_lockstep([p], stub=True)
run_n = lambda p: sharding.run_phases(p, native)
sharding.set_targets(plan, content, [style], [1.0], run_f, lambda t: None)
plan.set_loss_weights(0.015, [w / 341 for w in (256, 64, 16, 4, 1)], 2.0)
image, grad = image0.clone(), torch.empty_like(image0)
m, v, ema = torch.zeros_like(image), torch.zeros_like(image), torch.zeros_like(image)
for name, run in (('torch.distributed fabric (RCCL, self-neighbour)', run_f), ('exchanges stubbed', run_s),
                  ('in-library fabric (RCCL, self-neighbour)', run_n), ('torch.distributed fabric (RCCL, self-neighbour)', run_f),
                  ('in-library fabric (RCCL, self-neighbour)', run_n)):
    for k in range(3):
        plan.closure_begin(image, grad); run(plan); plan.apply_update(image, grad, m, v, ema, 1 + k, 0.02)
    torch.cuda.synchronize()
    n, t0 = 20, time.perf_counter()
    for k in range(n):
        plan.closure_begin(image, grad); run(plan); plan.apply_update(image, grad, m, v, ema, 4 + k, 0.02)
        if k == 5:      # the host's share from the first six iterations: later ones wait for room in the hardware queues
            host = (time.perf_counter() - t0) / 6 * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / n * 1e3
    print(f'[fabric] {W}x{H} middle strip, {name}: host enqueue {host:.2f} ms per iteration, iteration {total:.2f} ms', flush=True)
dist.destroy_process_group()
