"""Ablation variants of the Winograd convolution kernel (ST_WINO_TUNE, compile-time variants of wino_conv_kernel; wrong results,
only the time matters): where a K chunk's time goes.   gpurun -- python tools/winograd_phases.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

NAMES = {0: 'full kernel', 1: 'no transform (VALU + ds_write)', 2: 'no MFMA / operand fetch', 3: 'no transform, no MFMA',
         4: 'no weight DMA', 8: 'no patch loads', 12: 'no DMA, no patch loads', 13: 'only MFMA + barriers', 14: 'only transform + barriers',
         15: 'barriers + epilogue only', 31: 'epilogue only', 32: 'half the patch loads (8 B)', 64: 'one 16 B load per patch row',
         36: 'half the patch loads, no DMA', 68: '16 B patch loads, no DMA'}
for name, cin, cout, size in (('conv3_2 @2048^2', 256, 256, 512), ('conv2_2 @2048^2', 128, 128, 1024), ('conv3_2 @512^2', 256, 256, 128)):
    print(name)
    for tune in (0, 4, 8, 12, 32, 64, 36, 68):
        with _hip.options(ST_WINO_TUNE=tune, ST_WINO_TX=32):
            t = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 5, 10) for _ in range(2))
        print(f'  tune {tune:2d} {NAMES[tune]:36s} {t:8.1f} us', flush=True)
