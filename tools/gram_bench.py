#!/usr/bin/env python3
"""Isolated time of the Gram / mean kernels (gram_partial + gram_finalize) per style tap.  python tools/gram_bench.py [size]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch
from style_transfer import _hip, vgg
size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = 'cuda:0'
net = _hip.Net(vgg.synthetic_vgg19_weights(0), 'max', dev, 'fp16x3')
plan = _hip.Plan(net, size, size)
plan.forward(torch.rand((1, 3, size, size)).to(dev), 29)
torch.cuda.synchronize()
for layer, c, lvl in ((1, 64, 0), (6, 128, 1), (11, 256, 2), (20, 512, 3), (29, 512, 4)):
    npix = (size >> lvl) ** 2
    for _ in range(2):
        plan.moments(layer)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        plan.moments(layer)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    gflop, gb = 2 * c * c * npix / 1e9, c * npix * 4 / 1e9
    print(f'features[{layer:2d}] C={c:3d} pixels={npix:8d}: {us:8.1f} us  {gflop / us * 1e3:6.1f} TF fp32  {gb / us * 1e3:5.2f} TB/s read')
