"""The Winograd F(2x2, 3x3) fp16x3 prototype (csrc/st_conv_wino.hip, operator precision code 5) against the shipped
producer / consumer convolution (code 4): accuracy against float64 and microseconds per launch on the trunk's layer shapes.
VERDICT r4 next #5's kill criterion: rel-L2 <= 1e-5 AND >= 1.3 x at 512^2 and 2048^2.   gpurun -- python tools/winograd_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

DEV = 'cuda:0'
torch.manual_seed(0)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print('| case | Winograd fp16x3 vs float64 | shipped fp16x3 vs float64 |')
print('|---|---:|---:|')
for cin, cout, h, w in ((64, 64, 32, 48), (128, 128, 48, 32), (256, 256, 32, 32), (512, 512, 16, 16), (256, 512, 16, 32)):
    x = torch.relu(torch.randn(1, cin, h, w))
    wt = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout) * 0.1
    want = torch.relu(torch.nn.functional.conv2d(x.double(), wt.double(), b.double(), padding=1))
    got5 = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 5).cpu()
    got4 = _hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4).cpu()
    print(f'| {cin}->{cout} {w}x{h} | {rel(got5, want):.2e} | {rel(got4, want):.2e} |', flush=True)

print()
print('| layer shape (image) | shipped (us) | Winograd prototype (us) | ratio | shipped TF-equivalent | prototype TF-equivalent |')
print('|---|---:|---:|---:|---:|---:|')
for name, cin, cout, size in (('conv3_2 @512^2', 256, 256, 128), ('conv3_2 @2048^2', 256, 256, 512), ('conv2_2 @512^2', 128, 128, 256),
                              ('conv2_2 @2048^2', 128, 128, 1024), ('conv4_2 @512^2', 512, 512, 64), ('conv4_2 @2048^2', 512, 512, 256),
                              ('conv1_2 @512^2', 64, 64, 512), ('conv1_2 @2048^2', 64, 64, 2048)):
    t4 = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 4, 20) for _ in range(2))
    t5 = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 5, 20) for _ in range(2))
    flops = 2.0 * 9 * cin * cout * size * size
    print(f'| {name}: {cin}->{cout}, {size}x{size} | {t4:.1f} | {t5:.1f} | {t4 / t5:.2f} | {flops / t4 / 1e6:.0f} | {flops / t5 / 1e6:.0f} |', flush=True)
