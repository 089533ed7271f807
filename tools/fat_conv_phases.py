"""Ablation variants of the fat convolution kernel (ST_CONV_FAT_TUNE; wrong results, only the time matters).   gpurun -- python tools/fat_conv_phases.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

NAMES = {0: 'full kernel', 1: 'no activation staging', 2: 'no MFMA / operand fetches', 4: 'no weight DMA', 5: 'MFMA + barriers only (no staging, no DMA)',
         7: 'barriers + epilogue', 15: 'epilogue only', 16: 'no patch loads (split + LDS writes of stale registers)', 32: 'patch loads only (no split, no LDS writes)', 64: 'all patch loads in the first slot of stage A', 128: 'all splits + writes in the last slot of stage B', 192: 'both', 512: 'patch loads from one 16 KB window (cache hits)', 1024: 'no split (raw bits written)', 2048: 'no LDS writes', 3072: 'loads + waits only', 4096: 'LDS writes to linear (conflict-free) addresses', 8192: 'plane 0 written only', 16384: 'two ds_write_b64 instead of one ds_write2st64_b64', 32768: 'patch by 40 LDS-DMA pieces (wrong data) instead of load + split + write'}
for name, cin, cout, size in (('conv4_2 @2048^2', 512, 512, 256), ('conv3_2 @2048^2', 256, 256, 512)):
    print(name)
    with _hip.options(ST_CONV_FAT=0):
        t = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 4, 10) for _ in range(2))
    print(f'  producer / consumer kernel                       {t:8.1f} us')
    for tune in (0, 1, 4, 5):
        with _hip.options(ST_CONV_FAT=2, ST_CONV_FAT_TUNE=tune):
            t = min(_hip.op_conv3x3_time(cin, cout, size, size, False, 4, 10) for _ in range(2))
        print(f'  tune {tune:2d} {NAMES[tune]:44s} {t:8.1f} us', flush=True)
