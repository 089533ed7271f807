"""Host enqueue time against device time per iteration at the small scales of the default run (128^2 / 181^2 / 256^2 / 362^2 /
512^2: 2 000 of its 3 000 iterations run at <= 256^2), eager launches against hipGraph replay of the closure (st_plan_set_graph).
VERDICT r5 next #7.   gpurun -- python tools/small_scales.py"""
import argparse
import os
import sys
import time

import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import bench  # noqa: E402

DEV = torch.device('cuda:0')
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=300)
ap.add_argument('--sizes', default='128,181,256,362,512')
a = ap.parse_args()
args = argparse.Namespace(precision='fp16x3', height=0, width=0)
print('| image | mode | host enqueue per step (us) | step (us, device-bound wall) | it/s | host share |')
print('|---|---|---:|---:|---:|---:|')
for size in [int(x) for x in a.sizes.split(',')]:
    plan, step, _, final = bench.run_single(args, DEV, 0, 1, (size, size))
    for mode in ('eager', 'graph', 'eager', 'graph'):
        plan.set_graph(mode == 'graph')
        for _ in range(30):
            step()
        torch.cuda.synchronize(DEV)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        t1 = time.perf_counter()                       # every launch of every step is enqueued
        torch.cuda.synchronize(DEV)
        t2 = time.perf_counter()
        host, wall = (t1 - t0) / a.steps * 1e6, (t2 - t0) / a.steps * 1e6
        print(f'| {size}^2 | {mode} | {host:.0f} | {wall:.0f} | {1e6 / wall:.1f} | {host / wall:.2f} |', flush=True)
    print(f'  (final loss {final():.6f})', flush=True)
    del plan

print()
print('unthrottled host cost of one step (device idle before every call: no queue back-pressure), us:')
for size in [int(x) for x in a.sizes.split(',')]:
    plan, step, _, final = bench.run_single(args, DEV, 0, 1, (size, size))
    for mode in ('eager', 'graph'):
        plan.set_graph(mode == 'graph')
        for _ in range(20):
            step()
        torch.cuda.synchronize(DEV)
        host = dev = 0.0
        for _ in range(100):
            t0 = time.perf_counter()
            step()
            t1 = time.perf_counter()
            torch.cuda.synchronize(DEV)
            t2 = time.perf_counter()
            host += t1 - t0
            dev += t2 - t0
        print(f'  {size}^2 {mode}: host {host / 100 * 1e6:.0f} us, step alone (enqueue + drain) {dev / 100 * 1e6:.0f} us', flush=True)
    del plan
