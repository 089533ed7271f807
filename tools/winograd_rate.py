"""The matrix rate a CU sustains in the consumer pattern of a Winograd F(2x2, 3x3) fp16x3 tile (st_op_winograd_consumer_rate:
16 ds_read_b128 of fresh operands per 12 MFMAs, four waves per CU, 16 position accumulators per wave) next to the shipped direct
tile's pattern (8 reads per 12 MFMAs).  profiles/r05_winograd.md.   gpurun -- python tools/winograd_rate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
from style_transfer import _hip      # noqa: E402

torch.cuda.set_device(0)
w = [_hip.op_winograd_consumer_rate(8192, 8) for _ in range(3)]
d4 = [_hip.op_mfma_rate(8, 4, 40000, 10)[0] for _ in range(2)]
d8 = [_hip.op_mfma_rate(8, 8, 20000, 10)[0] for _ in range(2)]
print(f'Winograd consumer pattern (4 waves, 16 reads / 12 MFMAs): {["%.0f" % v for v in w]} TFLOP/s of MFMA work '
      f'= {max(w) / 3:.0f} TF of fp16x3 products = {max(w) / 3 * 2.25:.0f} TF of direct-convolution-equivalent work')
print(f'direct tile pattern (8 reads / 12 MFMAs): 4 waves {["%.0f" % v for v in d4]}, 8 waves {["%.0f" % v for v in d8]} TFLOP/s '
      f'= {max(d8) / 3:.0f} TF fp32-equivalent')
