"""What a device-wide barrier costs inside one launch on this chip (st_op_grid_barrier_time): the building block of the
persistent Newton-Schulz chain kernel.  gpurun -- python tools/grid_barrier.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
import torch
from style_transfer import _hip

torch.cuda.init()
print('| workgroups | groups | payload floats per workgroup | us per round (payload > 0: two barriers) | stale reads |')
print('|---:|---:|---:|---:|---:|')
for wgs in (128, 256):
    for groups in (0, 8, 16, 32):
        for per_wg in (0, 1):
            for nap in (1, 4, 16):
                for payload in (0, 1024):
                    us, err = _hip.op_grid_barrier_time(wgs, 400, payload, groups + 100 * per_wg + 1000 * nap)
                    print(f'| {wgs} | {groups} flags/wg={per_wg} sleep={nap} | {payload} | {us:.2f} | {err} |', flush=True)
