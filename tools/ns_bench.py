#!/usr/bin/env python3
"""Isolated timing of the Newton-Schulz forward and Lyapunov backward chains (HIP events, st_op_sqrtm_time), per
arithmetic: fp32 chains (ST_NS_F16=0) vs the fp16x3 chains of csrc/st_nsgemm.hip; backward = the plan's form
(gradient = multiple of I, reduced recurrence) and the general full recurrence."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch  # noqa: E402,F401
from style_transfer import _hip  # noqa: E402

_hip.load_library()


def line(tag, n, **opts):
    with _hip.options(**opts):
        f, b = _hip.op_sqrtm_time(n, 20)
    print(f'n={n:4d} {tag:34s}: fwd chain {f:7.1f} us | bwd chain {b:7.1f} us', flush=True)
    return f, b


for n in (64, 128, 256):
    line('fp32, diag (reduced) backward', n, ST_NS_TIME_DIAG=1)
line('fp32, full backward', 512, ST_NS_F16=0, ST_NS_TIME_DIAG=0)
line('fp32, diag (reduced) backward', 512, ST_NS_F16=0, ST_NS_TIME_DIAG=1)
line('shipped: fp32 fwd, fp16x3 diag bwd', 512, ST_NS_TIME_DIAG=1)
line('fp16x3 both (4 waves)', 512, ST_NS_F16_FWD=1, ST_NS_TIME_DIAG=1)
line('fp16x3 both at n=256', 256, ST_NS_F16_FWD=1, ST_NS_TIME_DIAG=1, ST_NS_F16_MIN_N=256)
