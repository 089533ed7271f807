#!/usr/bin/env python3
"""Isolated timing of the Newton-Schulz forward (24 launches + 4) and Lyapunov backward (24 + 4) chains."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
from style_transfer import _hip
lib = _hip.load_library()
for n in (64, 128, 256, 512):
    f, b = ctypes.c_double(), ctypes.c_double()
    _hip._check(lib.st_op_sqrtm_time(n, 10, ctypes.byref(f), ctypes.byref(b), None))
    gf_f, gf_b = 35 * 2 * n ** 3 / 1e9, 71 * 2 * n ** 3 / 1e9
    print(f'n={n:4d}: fwd chain {f.value:8.1f} us ({f.value / 28:5.1f} us/launch, {gf_f / f.value * 1e3:6.1f} TF) | '
          f'bwd chain {b.value:8.1f} us ({b.value / 28:5.1f} us/launch, {gf_b / b.value * 1e3:6.1f} TF)')
