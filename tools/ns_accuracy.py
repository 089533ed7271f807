#!/usr/bin/env python3
"""Accuracy of the n = 512 style heads per Newton-Schulz arithmetic: the relu4_1 / relu5_1 loss terms and the image
gradient of one closure against the float64 oracle, for the fp32 chains, the fp16x3 chains (both directions) and the shipped combination."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, os.path.join(R, 'style-transfer-pytorch_amd'), os.path.join(R, 'oracle'), os.path.join(R, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import st_oracle as O  # noqa: E402
from style_transfer import _hip, vgg  # noqa: E402
from test_hot_path_gpu import _build_plan, _smooth  # noqa: E402

torch.set_num_threads(16)
W = vgg.synthetic_vgg19_weights(0)
W64 = [(w.double(), b.double()) for w, b in W]


def case(kind, size):
    if kind == 'photo':
        c, s, i = _smooth(21, size, size), _smooth(22, size * 200 // 256, size), _smooth(23, size, size)
    else:
        g = torch.Generator().manual_seed(21)
        c = torch.rand((1, 3, size, size), generator=g)
        s = torch.rand((1, 3, size * 200 // 256, size), generator=g)
        i = torch.rand((1, 3, size, size), generator=g)
    t32 = O.build_targets(c, [s], W)
    terms32, _, grad32 = O.loss_and_grad(i, W, t32)
    t64 = O.build_targets(c.double(), [s.double()], W64)
    terms64, _, grad64 = O.loss_and_grad(i.double(), W64, t64)
    floor = [abs(a - b) / abs(b) for a, b in zip(terms32, terms64)]
    print(f'== {kind} {size}^2: cpu fp32 vs fp64: relu4_1 {floor[4]:.2e} relu5_1 {floor[5]:.2e} grad '
          f'{float((grad32.double() - grad64).norm() / grad64.norm()):.2e}')
    for name, opts in (('fp32 chains', dict(ST_NS_F16=0)), ('fp16x3 chains', dict(ST_NS_F16=1, ST_NS_F16_FWD=1)),
                       ('fp16x3 backward only (shipped)', dict(ST_NS_F16=1, ST_NS_F16_FWD=0))):
        with _hip.options(**opts):
            net, plan = _build_plan(_hip, W, c, [s], [1.0], precision='fp16x3')
            losses, g = plan.loss_and_grad(i.to('cuda:0'))
            got = losses.cpu().double().numpy()
            gc = g.cpu().double()
        sg = [(got[k] - terms64[k]) / abs(terms64[k]) for k in range(7)]
        print(f'   {name:30s} vs fp64 (signed): relu4_1 {sg[4]:+.2e} relu5_1 {sg[5]:+.2e} | vs cpu32: relu4_1 '
              f'{abs(got[4] - terms32[4]) / abs(terms32[4]):.2e} relu5_1 {abs(got[5] - terms32[5]) / abs(terms32[5]):.2e} | '
              f'grad vs fp64 {float((gc - grad64).norm() / grad64.norm()):.2e}', flush=True)


case('white', 256)
case('photo', 256)
case('white', 512)
