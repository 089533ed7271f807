#!/usr/bin/env python3
"""What do OTHER streams' kernel boundaries cost a Newton-Schulz chain?  (round 5, profiles/r05_ns_chain.md section 7)

The n = 512 chain of a head (st_op_sqrtm_time: forward and backward chain, HIP events) is timed alone and while 1 - 3 host
threads launch trivial kernels (torch: x.add_(1) on 1 element = one workgroup, or on 64 K elements) on streams of their own -
no data in common with the chain, next to no CU time.  Forms of the chain: one launch per product (shipped), and the persistent
kernel whose operands live in the L2 (ST_NS_CHAIN=8, ST_NS_CHAIN_L2=2).

    gpurun -- python tools/boundary_probe.py"""
import os
import sys
import threading
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
import torch  # noqa: E402
from style_transfer import _hip  # noqa: E402

_hip.load_library()
DEV = 'cuda:0'


def background(nthreads, elems, kind='add'):
    stop = threading.Event()
    counts = [0] * nthreads

    def work(k):
        s = torch.cuda.Stream(device=DEV)
        x = torch.zeros(elems, device=DEV)
        if kind == 'matmul':                     # elems = n: n x n fp32 products (rocBLAS; the matrix pipe and the LDS)
            torch.backends.cuda.matmul.allow_tf32 = False
            x = torch.randn((elems, elems), device=DEV)
            y = torch.empty_like(x)
        with torch.cuda.stream(s):
            while not stop.is_set():
                for _ in range(50 if kind == 'add' else 4):
                    if kind == 'add':
                        x.add_(1.0)
                    else:
                        torch.mm(x, x, out=y)
                counts[k] += 50 if kind == 'add' else 4
                if kind != 'add' or counts[k] % 2000 == 0:
                    s.synchronize()              # bound the queue depth
    threads = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(nthreads)]
    for t in threads:
        t.start()
    return stop, threads, counts


print('| chain form | background launches | forward chain (us) | backward chain (us) | background launches / ms |')
print('|---|---|---:|---:|---:|')
for form, opts in (('one launch per product', dict(ST_NS_CHAIN=0)),
                   ('persistent, L2 + arena, LDS-DMA', dict(ST_NS_CHAIN=8, ST_NS_CHAIN_SYM=0, ST_NS_CHAIN_L2=2)),
                   ('persistent, sc1 loads', dict(ST_NS_CHAIN=8, ST_NS_CHAIN_SYM=0, ST_NS_CHAIN_L2=0))):
    for nthreads, elems, kind in ((0, 1, 'add'), (1, 1, 'add'), (3, 1, 'add'), (3, 65536, 'add'), (1, 1 << 28, 'add'),
                                  (1, 512, 'matmul'), (2, 512, 'matmul'), (1, 4096, 'matmul')):
        with _hip.options(ST_NS_TIME_DIAG=1, **opts):
            _hip.op_sqrtm_time(512, 5)
            if nthreads:
                stop, threads, counts = background(nthreads, elems, kind)
                time.sleep(0.3)
            c0, t0 = (sum(counts), time.perf_counter()) if nthreads else (0, time.perf_counter())
            res = [_hip.op_sqrtm_time(512, 40) for _ in range(3)]
            rate = ((sum(counts) - c0) / ((time.perf_counter() - t0) * 1e3)) if nthreads else 0.0
            if nthreads:
                stop.set()
                for t in threads:
                    t.join()
                torch.cuda.synchronize()
        f = sorted(r[0] for r in res)[1]
        b = sorted(r[1] for r in res)[1]
        what = 'none' if not nthreads else (f'{nthreads} thread(s), fp32 torch.mm {elems}^3' if kind == 'matmul' else
                                            f'{nthreads} thread(s), x.add_(1) on {"1 element" if elems == 1 else f"{elems} elements"}')
        print(f'| {form} | {what} | {f:.1f} | {b:.1f} | {rate:.0f} |', flush=True)
