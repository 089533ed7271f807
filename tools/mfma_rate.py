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

print()
print('| MFMA waves per CU | ds_read_b128 per 12 MFMA | VALU-only waves | 1 = s_setprio 3, 2 = VALU waves oldest | TFLOP/s | cycles per MFMA (one wave) | cycles per VALU instruction (VALU wave) | MHz |')
print('|---:|---:|---:|---:|---:|---:|---:|---:|')
for waves, reads in ((8, 0), (8, 8), (4, 8)):
    for vw, prio in ((0, 0), (4, 0), (4, 1), (4, 2)):
        steps = 20000
        # the VALU loop is sized to END inside the MFMA loop even at 32 cycles per instruction (32 instructions per step)
        tf, mhz, cv, cm = _hip.op_mfma_valu_rate(reads, waves, steps, vw, int(steps * 0.3 * (waves // 4)) if vw else 0, prio)
        print(f'| {waves} | {reads} | {vw} | {prio} | {tf:.0f} | {cm:.1f} | {cv:.1f} | {mhz:.0f} |', flush=True)
# VALU wave alone (no MFMA issued by the other waves: steps = 1)
tf, mhz, cv, cm = _hip.op_mfma_valu_rate(0, 8, 2, 4, 400000, 0)
print(f'| 8 (idle) | 0 | 4 | 0 | - | - | {cv:.1f} | {mhz:.0f} |')
