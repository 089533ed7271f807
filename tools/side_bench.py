#!/usr/bin/env python3
"""HBM-bound side kernels in isolation (HIP events through torch on the launch stream): TV loss + gradient, achieved
GB/s against the 6.3 TB/s streaming rate (algorithmic bytes: read 3HW floats, write 3HW floats)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch  # noqa: E402
from style_transfer import _hip  # noqa: E402

dev = 'cuda:0'
for size in (512, 1024, 2048, (2172, 2896)):
    h, w = (size, size) if isinstance(size, int) else size
    img = torch.rand((1, 3, h, w), device=dev)
    for _ in range(3):
        _hip.op_tv_loss(img)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # op_tv_loss allocates its outputs: time a batch and subtract nothing (allocation is cached by torch)
    n = 20
    e0.record()
    for _ in range(n):
        _hip.op_tv_loss(img)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    gb = 2 * 3 * h * w * 4 / 1e9
    print(f'tv {w}x{h}: {us:8.1f} us per call (incl. the op wrapper sync), {gb / (us * 1e-6) / 1e3:6.2f} TB/s algorithmic')
