#!/usr/bin/env python3
"""Round 5: the persistent Newton-Schulz chain kernel (csrc/st_nschain.hip, ST_NS_CHAIN bit 3 for the operators) against the launch-per-product
chains (ST_NS_CHAIN=0) - isolated chain times (st_op_sqrtm_time, HIP events) and agreement of the results with float64."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'style-transfer-pytorch_amd'))
sys.path.insert(0, os.path.join(R, 'oracle'))
import torch  # noqa: E402
from style_transfer import _hip  # noqa: E402
import st_oracle as O  # noqa: E402

_hip.load_library()
print('| n | form | forward chain (us) | backward chain, seed g I (us) |')
print('|---:|---|---:|---:|')
for n in (64, 128, 256, 512):
    for chain, sym in ((0, 0), (1, 0), (1, 15)):
        with _hip.options(ST_NS_CHAIN=8 * chain, ST_NS_TIME_DIAG=1, ST_NS_CHAIN_SYM=sym):
            f, b = _hip.op_sqrtm_time(n, 20)
        print(f'| {n} | {("persistent kernel, " + ("symmetric tile pairs" if sym else "every tile")) if chain else "one launch per product"} | {f:.1f} | {b:.1f} |', flush=True)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


print()
print('| n | matrix | form | root vs float64 NS-12 | tr(root) rel shift | gradient (seed -2/n I) vs float64 |')
print('|---:|---|---|---:|---:|---:|')
for n in (64, 256, 512):
    for kind in ('well conditioned', 'rank deficient'):
        g = torch.Generator().manual_seed(n + len(kind))
        if kind == 'well conditioned':
            b = torch.randn((n, 2 * n), generator=g)
            a = (b @ b.t()) / (2 * n) + torch.eye(n) * 1e-2
        else:
            b = torch.randn((n, n // 4), generator=g)
            a = (b @ b.t()) / (n // 4) + torch.eye(n) * 1e-4
        gd = -2.0 / n
        want64 = O.ns_sqrt(a.double(), 12)
        wantb64 = O.ns_sqrt_bwd(want64, torch.eye(n, dtype=torch.float64) * gd, 12)
        cpu32 = O.ns_sqrt(a, 12)
        cpub32 = O.ns_sqrt_bwd(cpu32, torch.eye(n) * gd, 12)
        print(f'| {n} | {kind} | CPU fp32 (the reference\'s arithmetic) | {rel(cpu32, want64):.2e} | '
              f'{float((cpu32.double().trace() - want64.trace()) / want64.trace()):+.2e} | {rel(cpub32, wantb64):.2e} |')
        for chain, sym in ((0, 0), (1, 0), (1, 15)):
            with _hip.options(ST_NS_CHAIN=8 * chain, ST_NS_CHAIN_SYM=sym):
                root = _hip.op_sqrtm_ns(a.to('cuda:0'))
                gb = _hip.op_sqrtm_ns_backward_diag(root, gd)
            r = root.cpu()
            asym = float((r - r.t()).abs().max())
            print(f'| {n} | {kind} | {("persistent, " + ("symmetric" if sym else "every tile")) if chain else "per product"} (max |R - R^T| {asym:.1e}) | {rel(r, want64):.2e} | '
                  f'{float((r.double().trace() - want64.trace()) / want64.trace()):+.2e} | {rel(gb.cpu(), wantb64):.2e} |', flush=True)
