"""Round 5: a device-wide barrier WITHOUT cache maintenance (st_op_grid_barrier_time, cfg digit 10000 = 1): the payload
travels in agent-scope (sc1) buffer stores / loads - coherent at the memory side by themselves - so an arrival is
s_waitcnt vmcnt(0) + one relaxed atomic and a release is the poll: no buffer_wbl2 / buffer_inv per workgroup.  The building
block of the persistent Newton-Schulz chain kernel (csrc/st_nschain.hip).  `reads` = slots of other workgroups each
workgroup reads per round (32 x 1024 floats = the 128 KB of operand panels a 32 x 32 tile of a 512^3 product needs).

    gpurun -- python tools/grid_barrier2.py
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
import torch
from style_transfer import _hip

torch.cuda.init()
print('| workgroups | mode | groups | sleep | payload floats | slots read | us per round (two barriers) | stale reads |')
print('|---:|---|---:|---:|---:|---:|---:|---:|')
for wgs in (32, 49, 64, 136, 256):
    for mode in (0, 1):
        for groups in (0, 8):
            for nap in (1, 4):
                for reads in (1, 32):
                    if reads > 1 and (groups or nap > 1):
                        continue
                    cfg = groups + 1000 * nap + 10000 * mode + 100000 * reads
                    us, err = _hip.op_grid_barrier_time(wgs, 300, 1024, cfg)
                    print(f'| {wgs} | {"write-through, no fences" if mode else "release / acquire fences"} | {groups} | {nap} | 1024 | '
                          f'{reads} | {us:.2f} | {err} |', flush=True)
# barrier alone (no payload: one barrier per round)
for wgs in (32, 64, 136, 256):
    for mode in (0, 1):
        us, err = _hip.op_grid_barrier_time(wgs, 300, 0, 1000 + 10000 * mode)
        print(f'| {wgs} | {"no fences" if mode else "fences"} | 0 | 1 | 0 | - | {us:.2f} (one barrier) | {err} |', flush=True)
