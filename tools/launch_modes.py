#!/usr/bin/env python3
"""How the ~430 launches of one iteration are issued: all eager (ST_HEAD_GRAPH=0), a replayed hipGraph of the whole
closure, or (default) one captured linear graph per style head launched when its tap is ready - it/s of the fused
step at the small scales, where the iteration is bound by the Newton-Schulz chains rather than by the trunk."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch  # noqa: E402
import bench  # noqa: E402


class A:
    precision = 'fp16x3'


def run(size, layout, at_tap, f16=1, steps=40):
    from style_transfer import _hip
    _hip.set_option('ST_HEAD_GRAPH', 1)
    _hip.set_option('ST_STREAM_LAYOUT', layout)
    _hip.set_option('ST_HEAD_AT_TAP', at_tap)
    _hip.set_option('ST_NS_F16', f16)
    a = A()
    a.height = a.width = size
    dev = torch.device('cuda:0')
    plan, step, _, _ = bench.run_single(a, dev, 0, 1)
    sec = bench.timed_run(step, steps if size < 1024 else 15, 8, dev)
    del plan, step
    torch.cuda.empty_cache()
    return 1.0 / sec


for size in [int(s) for s in (sys.argv[1:] or ['128', '256', '512', '1024', '2048'])]:
    res = {f'L{l}{"t" if t else ""}': run(size, l, t) for l, t in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1))}
    print(f'{size:5d}^2: ' + '  '.join(f'{m} {v:7.1f}' for m, v in res.items()) + '  it/s', flush=True)
