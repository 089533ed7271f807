#!/usr/bin/env python3
"""Per-layer microbenchmark of the MFMA convolution (same launch path as the plan).
    python tools/conv_bench.py [size]       env: ST_CONV_TUNE / ST_CONV_SHAPE / ST_CONV_KSPLIT"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch
from style_transfer import _hip
lib = _hip.load_library()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PREC = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LAYERS = [('conv1_2', 64, 64, 0), ('conv2_1', 64, 128, 1), ('conv2_2', 128, 128, 1), ('conv3_1', 128, 256, 2),
          ('conv3_2', 256, 256, 2), ('conv4_1', 256, 512, 3), ('conv4_2', 512, 512, 3), ('conv5_1', 512, 512, 4)]
tot = {0: 0.0, 1: 0.0}
mult = {'conv3_2': 3, 'conv4_2': 3}
ONLY = sys.argv[3] if len(sys.argv) > 3 else None
for name, cin, cout, lvl in LAYERS:
    if ONLY and name != ONLY:
        continue
    h = size >> lvl
    row = []
    for dgrad in (0, 1):
        us = ctypes.c_double()
        _hip._check(lib.st_op_conv3x3_time(cin, cout, h, h, dgrad, PREC, 20, ctypes.byref(us), None))
        gf = 2 * 9 * cin * cout * h * h / 1e9
        row.append(f'{"dgrad" if dgrad else "fwd  "} {us.value:7.1f} us {gf / us.value * 1e3:6.1f} TF')
        tot[dgrad] += us.value * mult.get(name, 1)
    print(f'{name} {cin:3d}->{cout:3d} @{h:4d}: ' + ' | '.join(row))
print(f'trunk total (12 convs): fwd {tot[0]:.0f} us, dgrad {tot[1]:.0f} us, sum {tot[0] + tot[1]:.0f} us '
      f'precision={PREC} tune={os.environ.get("ST_CONV_TUNE", "0")} shape={os.environ.get("ST_CONV_SHAPE", "-")} ks={os.environ.get("ST_CONV_KSPLIT", "-")}')
