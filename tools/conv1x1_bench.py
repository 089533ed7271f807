#!/usr/bin/env python3
"""Isolated launches of the style heads' gradient step dF = Ssym F + b (st_op_conv1x1, fp16x3) at the five tap shapes of
an image; run under `rocprofv3 --kernel-trace --stats` for the kernel durations.  python tools/conv1x1_bench.py [size]"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch                                   # noqa: E402
from style_transfer import _hip                # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = 'cuda:0'
for c, lvl in ((64, 0), (128, 1), (256, 2), (512, 3), (512, 4)):
    npix = (size >> lvl) ** 2
    x = torch.rand((c, npix), device=dev)
    w = torch.randn((c, c), device=dev) * 0.05
    b = torch.randn((c,), device=dev)
    for _ in range(6):
        out = _hip.op_conv1x1(x, w, b, 4)
    torch.cuda.synchronize()
    print(f'C={c} npix={npix}: read + write {2 * c * npix * 4 / 1e9:.2f} GB', flush=True)
    del x, out
