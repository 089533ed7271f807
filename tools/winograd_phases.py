"""Ablation timing of the Winograd prototype's phases (ST_WINO_TUNE bits: 1 no transform, 2 no MFMA, 4 no weight DMA, 8 no raw patch)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip
for name, cin, cout, size in (('conv3_2 @512^2', 256, 256, 128), ('conv3_2 @2048^2', 256, 256, 512)):
    row = []
    for tune in (0, 1, 2, 4, 8, 3, 12, 15, 14, 13):
        with _hip.options(ST_WINO_TUNE=tune):
            row.append((tune, min(_hip.op_conv3x3_time(cin, cout, size, size, False, 5, 20) for _ in range(2))))
    print(name, ' | '.join(f'tune {t}: {us:.1f} us' for t, us in row), flush=True)
