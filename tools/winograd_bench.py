"""The Winograd F(2x2, 3x3) fp16x3 convolution (csrc/st_conv_wino.hip, operator precision code 5) against the shipped direct
producer / consumer convolution (code 4): accuracy against float64 (forward and data gradient, with the plan's epilogue options,
ragged sizes, K split) and microseconds per launch on the trunk's layer shapes.   gpurun -- python tools/winograd_bench.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

DEV = 'cuda:0'
torch.manual_seed(0)
QUICK = 'quick' in sys.argv[1:]
TIME_ONLY = 'time' in sys.argv[1:]


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def accuracy():
    print('| case | Winograd fp16x3 vs float64 | direct fp16x3 vs float64 | worst element (Winograd) |')
    print('|---|---:|---:|---:|')
    F = torch.nn.functional
    for cin, cout, h, w in ((64, 64, 32, 48), (128, 128, 48, 32), (256, 256, 32, 32), (512, 512, 16, 16), (256, 512, 16, 32),
                            (64, 128, 20, 66), (128, 64, 7, 130), (512, 512, 8, 8), (64, 64, 135, 182), (256, 256, 64, 64)):
        x = torch.relu(torch.randn(1, cin, h, w))
        wt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout) * 0.1
        want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
        got5 = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 5).cpu()
        got4 = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4).cpu()
        worst = float((got5.double() - want).abs().max() / want.abs().max())
        print(f'| fwd {cin}->{cout} {w}x{h} | {rel(got5, want):.2e} | {rel(got4, want):.2e} | {worst:.1e} |', flush=True)
    for tx in (8, 16, 32):
        with _hip.options(ST_WINO_TX=tx):
            cin, cout, h, w = 64, 64, 36, 132
            x = torch.relu(torch.randn(1, cin, h, w))
            wt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
            b = torch.randn(cout) * 0.1
            want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
            got5 = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 5).cpu()
            print(f'| fwd {cin}->{cout} {w}x{h}, tile row of {tx} | {rel(got5, want):.2e} | | |', flush=True)
    # data gradient with the plan's epilogue: out = mask > 0 ? (out + dgrad) : 0
    for cin, cout, h, w in ((64, 128, 24, 40), (256, 256, 16, 16), (512, 256, 9, 34)):
        g = torch.randn(1, cout, h, w)
        wt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
        prev = torch.randn(1, cin, h, w)
        mask = torch.randn(1, cin, h, w)
        want = F.conv_transpose2d(g.double(), wt.double(), padding=1)
        want = torch.where(mask > 0, want + prev.double(), torch.zeros_like(want))
        outs = []
        for code in (5, 4):
            out = prev.clone().to(DEV)
            _hip.op_conv3x3_strip_ex(g.to(DEV), None, 0, 0, wt.to(DEV), None, False, True, out=out, out_mask=mask.to(DEV),
                                     precision=code)
            outs.append(out.cpu())
        print(f'| dgrad {cout}->{cin} {w}x{h} (+= , mask) | {rel(outs[0], want):.2e} | {rel(outs[1], want):.2e} | |', flush=True)
        plain = _hip.op_conv3x3_strip_ex(g.to(DEV), None, 0, 0, wt.to(DEV), None, False, True, precision=5).cpu()
        print(f'| dgrad {cout}->{cin} {w}x{h} (plain) | {rel(plain, F.conv_transpose2d(g.double(), wt.double(), padding=1)):.2e} | | |',
              flush=True)


def timing():
    print()
    print('| layer shape (image) | direct fwd (us) | Winograd fwd (us) | ratio | direct dgrad (us) | Winograd dgrad (us) | ratio | Winograd fwd TF-equivalent |')
    print('|---|---:|---:|---:|---:|---:|---:|---:|')
    shapes = (('conv3_2 @2048^2', 256, 256, 512), ('conv3_2 @512^2', 256, 256, 128), ('conv2_2 @2048^2', 128, 128, 1024),
              ('conv2_2 @512^2', 128, 128, 256), ('conv4_2 @2048^2', 512, 512, 256), ('conv4_2 @512^2', 512, 512, 64),
              ('conv1_2 @2048^2', 64, 64, 2048), ('conv1_2 @512^2', 64, 64, 512), ('conv2_1 @2048^2', 64, 128, 1024),
              ('conv3_1 @2048^2', 128, 256, 512), ('conv4_1 @2048^2', 256, 512, 256), ('conv5_1 @2048^2', 512, 512, 128),
              ('conv5_1 @512^2', 512, 512, 32), ('conv3_2 @1024^2', 256, 256, 256), ('conv4_2 @1024^2', 512, 512, 128))
    if QUICK:
        shapes = shapes[:4]
    _hip.set_option('ST_CONV_NOMASK', 1)
    for name, cin, cout, size in shapes:
        t4 = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 4, 20) for _ in range(2))
        t5 = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 5, 20) for _ in range(2))
        d4 = min(_hip.op_conv3x3_time(cin, cout, size, size, True, 4, 20) for _ in range(2))
        d5 = min(_hip.op_conv3x3_time(cin, cout, size, size, True, 5, 20) for _ in range(2))
        flops = 2.0 * 9 * cin * cout * size * size
        print(f'| {name}: {cin}->{cout}, {size}x{size} | {t4:.1f} | {t5:.1f} | {t4 / t5:.2f} | {d4:.1f} | {d5:.1f} | {d4 / d5:.2f} | '
              f'{flops / t5 / 1e6:.0f} |', flush=True)


if not TIME_ONLY:
    accuracy()
timing()
if not QUICK and not TIME_ONLY:
    print()
    print('tile row width (ST_WINO_TX) on conv3_2 / conv2_2 @2048^2, forward, us:')
    for tx in (8, 16, 32):
        with _hip.options(ST_WINO_TX=tx):
            a = min(_hip.op_conv3x3_time(256, 256, 512, 512, False, 5, 20) for _ in range(2))
            b = min(_hip.op_conv3x3_time(128, 128, 1024, 1024, False, 5, 20) for _ in range(2))
            c = min(_hip.op_conv3x3_time(256, 256, 128, 128, False, 5, 20) for _ in range(2))
        print(f'  TX={tx}: conv3_2@2048 {a:.1f}  conv2_2@2048 {b:.1f}  conv3_2@512 {c:.1f}', flush=True)
