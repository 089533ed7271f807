"""Sustained rate of the 16-bit matrix pipe on this chip (csrc/st_diag.hip): alone and next to the LDS operand stream
of an LDS-fed tile.  `python tools/mfma_rate.py` on the GPU box; numbers go to profiles/r02_mfma_sustained.md."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

torch.cuda.set_device(0)
print('| waves per CU | ds_read_b128 per 12 MFMA | TFLOP/s (fp16 dense) | fp16x3 fp32-equivalent TF | shader clock MHz |')
print('|---:|---:|---:|---:|---:|')
for waves in (4, 8, 12):
    for reads in (0, 4, 8):
        steps = 40000 // (waves // 4)
        tf, mhz = _hip.op_mfma_rate(reads, waves, steps, 10)
        print(f'| {waves} | {reads} | {tf:.0f} | {tf / 3:.0f} | {mhz:.0f} |', flush=True)
