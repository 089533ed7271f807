"""The fat single-role 3x3 convolution (csrc/st_conv_fat.hip, ST_CONV_FAT=2) against the producer / consumer kernel: bit-identity
of the results (forward, data gradient with the plan's epilogue), microseconds per launch on the trunk's layer shapes.
gpurun -- python tools/fat_conv_bench.py [size ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

DEV = 'cuda:0'
torch.manual_seed(0)
F = torch.nn.functional


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print('| case | fat vs float64 | identical to the producer / consumer kernel |')
print('|---|---:|---|')
for cin, cout, h, w in ((64, 64, 32, 64), (64, 128, 48, 32), (128, 128, 40, 68), (256, 256, 16, 32), (128, 256, 35, 100), (512, 512, 16, 32)):
    x = torch.relu(torch.randn(1, cin, h, w))
    wt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout) * 0.1
    want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    base = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4)
    with _hip.options(ST_CONV_FAT=2):
        got = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4)
    with _hip.options(ST_CONV_PC=2, ST_CONV_PC_SHAPE=1):
        xl = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4)
    print(f'| fwd {cin}->{cout} {w}x{h} | {rel(got.cpu(), want):.2e} | default {torch.equal(got, base)}, XL tile {torch.equal(got, xl)} |', flush=True)
    g = torch.randn(1, cout, h, w)
    prev, mask = torch.randn(1, cin, h, w), torch.randn(1, cin, h, w)
    wantd = F.conv_transpose2d(g.double(), wt.double(), padding=1)
    wantd = torch.where(mask > 0, wantd + prev.double(), torch.zeros_like(wantd))
    outs = []
    for fat in (0, 2):
        with _hip.options(ST_CONV_FAT=fat):
            out = prev.clone().to(DEV)
            _hip.op_conv3x3_strip_ex(g.to(DEV), None, 0, 0, wt.to(DEV), None, False, True, out=out, out_mask=mask.to(DEV), precision=4)
            outs.append(out)
    print(f'| dgrad {cout}->{cin} {w}x{h} (+=, mask) | {rel(outs[1].cpu(), wantd):.2e} | default {torch.equal(outs[0], outs[1])} |', flush=True)

sizes = [int(a) for a in sys.argv[1:]] or [2048, 1024]
_hip.set_option('ST_CONV_NOMASK', 1)
for size in sizes:
    print()
    print(f'| layer @{size}^2 | producer / consumer fwd (us) | fat fwd (us) | ratio | p / c dgrad | fat dgrad | ratio | fat fwd TF |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    tot = [0.0, 0.0]
    for name, cin, cout, lvl, mult in (('conv1_2', 64, 64, 0, 1), ('conv2_1', 64, 128, 1, 1), ('conv2_2', 128, 128, 1, 1), ('conv3_1', 128, 256, 2, 1),
                                       ('conv3_2', 256, 256, 2, 3), ('conv4_1', 256, 512, 3, 1), ('conv4_2', 512, 512, 3, 3), ('conv5_1', 512, 512, 4, 1)):
        s = size >> lvl
        t = {}
        for fat in (0, 2):
            with _hip.options(ST_CONV_FAT=fat):
                t[fat] = [min(_hip.op_conv3x3_time(cin, cout, s, s, d, 4, 20) for _ in range(2)) for d in (False, True)]
        flops = 2.0 * 9 * cin * cout * s * s
        tot[0] += mult * sum(t[0])
        tot[1] += mult * sum(t[2])
        print(f'| {name} {cin}->{cout} {s}x{s} | {t[0][0]:.1f} | {t[2][0]:.1f} | {t[0][0] / t[2][0]:.2f} | {t[0][1]:.1f} | {t[2][1]:.1f} | '
              f'{t[0][1] / t[2][1]:.2f} | {flops / t[2][0] / 1e6:.0f} |', flush=True)
    print(f'trunk total (12 convs, fwd + dgrad): producer / consumer {tot[0]:.0f} us, fat {tot[1]:.0f} us, ratio {tot[0] / tot[1]:.3f}')
