#!/usr/bin/env python3
"""Per-rank cost of the strip-sharded closure, measured on ONE GPU: R strip plans of the same image run in
lockstep (transport = device copies), so wall / R approximates one rank's GPU + host time without the fabric.
    python tools/strip_bench.py [size] [ranks] [precision]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
sys.path.insert(0, os.path.join(R, 'tests'))
sys.path.insert(0, os.path.join(R, 'oracle'))
import torch
from style_transfer import _hip as hip, sharding as sh, vgg
import st_oracle as O
from test_sharding_gpu import _targets_lockstep, _smooth

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
prec = sys.argv[3] if len(sys.argv) > 3 else 'fp16x3'
DEV = 'cuda:0'
w = vgg.synthetic_vgg19_weights(0)
net = hip.Net(w, 'max', DEV, prec)
content, style, image = _smooth(31, size, size), _smooth(32, size, size), _smooth(33, size, size)
rows = sh.strip_rows(size, world)
plans = [sh.StripPlan(net, size, size, b, e) for b, e in rows]
_targets_lockstep(sh, plans, content, [style], [1.0])
imgs = [image[:, :, b:e].contiguous().to(DEV) for b, e in rows]
grads = [torch.empty_like(t) for t in imgs]
ms_ = [torch.zeros_like(t) for t in imgs]; vs_ = [torch.zeros_like(t) for t in imgs]
emas = [0.01 * t for t in imgs]
def step(k):
    for p, t, g in zip(plans, imgs, grads):
        p.closure_begin(t, g)
    sh.run_phases_lockstep(plans)
    for p, t, g, m, v, e in zip(plans, imgs, grads, ms_, vs_, emas):
        p.apply_update(t, g, m, v, e, k, 0.02)
for k in range(1, 6):
    step(k)
torch.cuda.synchronize()
n = 30
t0 = time.perf_counter()
for k in range(6, 6 + n):
    step(k)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f'{size}x{size}, {world} ranks in lockstep on one GPU, {prec}: {dt * 1e3:.2f} ms per iteration for all ranks '
      f'-> ~{dt * 1e3 / world:.2f} ms per rank (no fabric), i.e. <= {world / dt:.0f} it/s if ranks ran concurrently')
