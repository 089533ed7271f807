#!/usr/bin/env python3
"""Per-rank critical path of the strip-sharded closure, measured on ONE GPU.

R strip plans of the same image are built and given their targets in lockstep (transport = device copies); then every
rank's plan is timed ALONE with the exchanges stubbed out (no data moves: the results of those runs are garbage, the
kernels and the host work are exactly one rank's).  Two models: (1) the stubbed broadcasts return at once - only the OWNER
of a head's Newton-Schulz chains waits for them (rounds 3 - 5; optimistic: on hardware every rank waits for relu5_1's owner);
(2) the reduction -> broadcast time of relu5_1's head on its owner is measured and replayed as a delay on the other ranks.
The slowest rank of (2) is the iteration time a perfect fabric would give; unsharded time / (R x that) is the modelled
strong-scaling efficiency before communication.

    python tools/strip_bench.py [size | WxH] [ranks] [precision]        (ST_STRIP_NS_OWNER=0, ST_STRIP_OVERLAP=0: A/B)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
sys.path.insert(0, os.path.join(R, 'tests'))
sys.path.insert(0, os.path.join(R, 'oracle'))
import torch
import bench
from style_transfer import _hip as hip, sharding as sh, vgg
from test_sharding_gpu import _targets_lockstep, _smooth

height, width = bench.parse_size(sys.argv[1] if len(sys.argv) > 1 else '512')
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prec = sys.argv[3] if len(sys.argv) > 3 else 'fp16x3'
DEV = 'cuda:0'
net = hip.Net(vgg.synthetic_vgg19_weights(0), 'max', DEV, prec)
content, style, image = _smooth(31, height, width), _smooth(32, height, width), _smooth(33, height, width)
rows = sh.strip_rows(height, world, width)
plans = [sh.StripPlan(net, height, width, b, e).set_rank(r, world) for r, (b, e) in enumerate(rows)]
_targets_lockstep(sh, plans, content, [style], [1.0])
imgs = [image[:, :, b:e].contiguous().to(DEV) for b, e in rows]
grads = [torch.empty_like(t) for t in imgs]
ms_ = [torch.zeros_like(t) for t in imgs]; vs_ = [torch.zeros_like(t) for t in imgs]
emas = [0.01 * t for t in imgs]


def step_all(k):
    for p, t, g in zip(plans, imgs, grads):
        p.closure_begin(t, g)
    sh.run_phases_lockstep(plans)
    for p, t, g, m, v, e in zip(plans, imgs, grads, ms_, vs_, emas):
        p.apply_update(t, g, m, v, e, k, 0.02)


def step_one(r, k, owner_chains=None):
    plans[r].closure_begin(imgs[r], grads[r])
    sh.run_phases_lockstep([plans[r]], stub=True, owner_chains=owner_chains)
    plans[r].apply_update(imgs[r], grads[r], ms_[r], vs_[r], emas[r], k, 0.02)


for k in range(1, 4):
    step_all(k)
torch.cuda.synchronize()
n = 20 if height * width <= 1024 * 1024 else 8
per_rank, host = [], []
if os.environ.get('STRIP_TRACE_RANK'):                        # a kernel trace of ONE rank's iterations (tools/strip_prof.sh)
    r = int(os.environ['STRIP_TRACE_RANK'])
    for k in range(16):
        step_one(r, 4 + k)
    torch.cuda.synchronize()
    sys.exit(0)
for r in range(world):
    for k in range(3):
        step_one(r, 4 + k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        step_one(r, 7 + k)
    host.append((time.perf_counter() - t0) / n * 1e3)          # the host's share: enqueue only (phase machine + launches)
    torch.cuda.synchronize()
    per_rank.append((time.perf_counter() - t0) / n * 1e3)
# Second model: a rank that does not own relu5_1's head gets its result only when the OWNER's chains are done - the one head
# whose chains nothing overlaps (the shallower heads' chains run beside the rest of the forward pass and are done long before
# the backward pass asks for them; their broadcasts are issued late, so reduction -> broadcast on the owner says nothing
# about them).  The owner's reduction -> broadcast time of head 4 is measured (events; its broadcast is the first one issued,
# behind the chain's last kernel) and replayed as a delay on the other ranks.
chain_us, waited = {}, []
if world > 1 and os.environ.get('ST_STRIP_NS_OWNER', '1') != '0':
    cpu = sh.sleep_cycles_per_us(DEV)
    oc = {'rank': 0, 'measure': {}}                    # head 4's owner: (4 - 4) % world
    for k in range(4):
        oc['measure'].clear()
        step_one(0, 60 + k, oc)
    torch.cuda.synchronize()
    chain_us = {h: us for h, us in sh.owner_chain_us(oc).items() if h == 4}
    for r in range(world):
        oc = {'rank': r, 'delay_us': chain_us, 'cycles_per_us': cpu}
        for k in range(3):
            step_one(r, 70 + k, oc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            step_one(r, 80 + k, oc)
        torch.cuda.synchronize()
        waited.append((time.perf_counter() - t0) / n * 1e3)
for r in sorted({0, world - 1}):
    plans[r].profile_enable(True)
    for k in range(3):
        step_one(r, 40 + k)
    launches, ms, flops = plans[r].profile_read()
    plans[r].profile_enable(False)
    print(f'[strip_bench] rank {r}: {launches / 3:.0f} conv launches / step, {ms / 3:.2f} ms of conv launches (sum of HIP-event '
          f'brackets) = {flops / (ms * 1e-3) / 1e12:.0f} TF fp32-equivalent; step {per_rank[r]:.2f} ms -> {per_rank[r] - ms / 3:.2f} ms outside them')
print(f'[strip_bench] host enqueue ms per step (no fabric calls: the phase machine and its launches) = '
      + ' '.join(f'{t:.2f}' for t in host) + f'; {100 * max(h / t for h, t in zip(host, per_rank)):.0f} % of the step at worst')
print(f'[strip_bench] {width}x{height}, {world} ranks, {prec}: per-rank ms (exchanges stubbed, one rank at a time) = '
      + ' '.join(f'{t:.2f}' for t in per_rank) + f'; critical path {max(per_rank):.2f} ms -> <= {1e3 / max(per_rank):.1f} it/s')
if waited:
    print(f'[strip_bench] {width}x{height}, {world} ranks, {prec}: with the wait for relu5_1\'s owner (reduction -> broadcast on rank 0: '
          f'{chain_us.get(4, 0):.0f} us) replayed on the other ranks: per-rank ms = '
          + ' '.join(f'{t:.2f}' for t in waited) + f'; critical path {max(waited):.2f} ms -> <= {1e3 / max(waited):.1f} it/s; rows '
          + ' '.join(str(e - b) for b, e in rows))
