#!/usr/bin/env python3
"""Same-box A/B of library switches: one plan per variant, all alive at once; rounds of interleaved short runs
(A B C A B C ...), median ms per step per variant - box-to-box (+-5 %) and first-run effects cancel.

    python tools/ab.py SIZE name:OPT=V,OPT=V[;OPT=V at run time] ...
    python tools/ab.py 512 base: f16bwd:ST_NS_F16=1 fp32:ST_NS_F16=0
Options before ';' are set while the plan is created AND while it runs; the ones after ';' only while it runs."""
import os
import statistics
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from style_transfer import _hip  # noqa: E402


class A:
    precision = 'fp16x3'


def parse(spec):
    name, _, rest = spec.partition(':')
    create, _, run = rest.partition(';')
    kv = lambda t: {k: int(v) for k, v in (x.split('=') for x in t.split(',') if x)}   # noqa: E731
    c = kv(create)
    r = dict(c)
    r.update(kv(run))
    return name, c, r


def main():
    size = sys.argv[1]
    hw = bench.parse_size(size)
    variants = [parse(s) for s in sys.argv[2:]]
    dev = torch.device('cuda:0')
    plans = []
    for name, c, r in variants:
        with _hip.options(**c):
            a = A()
            a.height, a.width = hw
            plan, step, _, _ = bench.run_single(a, dev, 0, 1)
            for _ in range(4):
                step()
            torch.cuda.synchronize(dev)
        plans.append((name, r, plan, step))
    steps = 30 if hw[0] * hw[1] <= 512 * 512 else (12 if hw[0] * hw[1] <= 1024 * 1024 else 6)
    times = {name: [] for name, *_ in plans}
    for rnd in range(7):
        for name, r, plan, step in plans:
            with _hip.options(**r):
                for _ in range(3):
                    step()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    step()
                torch.cuda.synchronize(dev)
                times[name].append((time.perf_counter() - t0) / steps * 1e3)
    base = statistics.median(times[plans[0][0]])
    for name, *_ in plans:
        m = statistics.median(times[name])
        print(f'{size:>9s} {name:16s} median {m:7.3f} ms ({1e3 / m:7.1f} it/s)  min {min(times[name]):7.3f}  '
              f'vs {plans[0][0]}: {100 * (base / m - 1):+5.1f} %', flush=True)


if __name__ == '__main__':
    main()
