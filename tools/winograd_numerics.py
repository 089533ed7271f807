#!/usr/bin/env python3
"""VERDICT r4 next #5, the numerical half of the kill criterion: would a Winograd F(2x2, 3x3) form of the fp16x3 trunk
convolution keep fp32-class accuracy (per-conv rel-L2 <= 1e-5 against float64)?  CPU emulation (numpy / torch float64 as the
exact reference): weights transformed in double (U = G g G^T) and then split into two fp16 planes under one power-of-two scale,
the input tiles transformed in fp32 (V = B^T d B: additions only) BEFORE the split, the 16 plane-GEMMs with the three products
h0 g0 + h0 g1 + h1 g0 accumulated in fp32, the output transform in fp32.  Next to it the shipped direct fp16x3 form.

    python tools/winograd_numerics.py            (CPU, ~1 minute)
"""
import numpy as np
import torch

torch.manual_seed(0)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def split_planes(x32, scale_exp):
    """two fp16 planes of a fp32 tensor under a power-of-two scale: x ~ (h0 + h1) / 2^e, |x - ...| <= 2^-22 |x|"""
    xs = x32 * (2.0 ** scale_exp)
    h0 = xs.half()
    h1 = (xs - h0.float()).half()
    return h0.float(), h1.float()


def scale_for(x):
    amax = float(x.abs().max())
    return 0 if amax == 0 else 13 - int(np.floor(np.log2(amax)))      # bound in [2^13, 2^14)


def gemm3(a32, b32):
    """sum_k a[m,k] b[k,n] as h0 g0 + h0 g1 + h1 g0 with fp32 accumulation (emulated: float64 products of fp16 values, rounded to
    fp32 per 16-term block like the MFMA's accumulator chain)"""
    ea, eb = scale_for(a32), scale_for(b32)
    a0, a1 = split_planes(a32, ea)
    b0, b1 = split_planes(b32, eb)
    acc = torch.zeros(a32.shape[0], b32.shape[1], dtype=torch.float32)
    K = a32.shape[1]
    for k0 in range(0, K, 16):
        s = slice(k0, k0 + 16)
        blk = (a0[:, s].double() @ b0[s].double() + a0[:, s].double() @ b1[s].double() + a1[:, s].double() @ b0[s].double())
        acc = (acc.double() + blk).float()
    return acc * (2.0 ** -(ea + eb))


def direct_fp16x3(x, w):
    cout, cin = w.shape[:2]
    H, W = x.shape[1:]
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    cols = torch.stack([xp[:, i:i + H, j:j + W] for i in range(3) for j in range(3)], 0).reshape(9 * cin, H * W)
    wm = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin)                 # [cout][tap][cin] matches cols' [tap][cin]
    return gemm3(wm, cols).reshape(cout, H, W)


def winograd_fp16x3(x, w):
    cout, cin = w.shape[:2]
    H, W = x.shape[1:]
    U = torch.einsum('ij,ocjk,lk->ocil', G, w.double(), G).float()    # transformed in double, stored fp32 -> planes
    xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
    th, tw = H // 2, W // 2
    tiles = torch.stack([torch.stack([xp[:, 2 * i:2 * i + 4, 2 * j:2 * j + 4] for j in range(tw)], 1) for i in range(th)], 1)
    # tiles: [cin][th][tw][4][4]; V = B^T d B in fp32
    Bt = BT.float()
    V = torch.einsum('ij,cxyjk,lk->cxyil', Bt, tiles, Bt)
    M = torch.zeros(cout, th, tw, 4, 4, dtype=torch.float32)
    for a in range(4):
        for b in range(4):
            M[:, :, :, a, b] = gemm3(U[:, :, a, b], V[:, :, :, a, b].reshape(cin, th * tw)).reshape(cout, th, tw)
    At = AT.float()
    Y = torch.einsum('ij,oxyjk,lk->oxyil', At, M, At)                 # [cout][th][tw][2][2]
    return Y.permute(0, 1, 3, 2, 4).reshape(cout, H, W)


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


print('| case | direct fp16x3 | Winograd F(2x2,3x3) fp16x3 | plain fp32 direct (torch) |')
print('|---|---:|---:|---:|')
for name, cin, cout, hw, gain in (('conv3_2-like 256->256, 32x32', 256, 256, 32, None),
                                  ('conv2_2-like 128->128, 48x48', 128, 128, 48, None),
                                  ('conv4_2-like 512->512, 16x16', 512, 512, 16, None),
                                  ('one dominant input channel (x 2^10), 128->128, 32x32', 128, 128, 32, 2.0 ** 10)):
    x = torch.relu(torch.randn(cin, hw, hw))
    if gain:
        x[3] *= gain
    w = torch.randn(cout, cin, 3, 3) * (2.0 / (9 * cin)) ** 0.5
    want = torch.nn.functional.conv2d(x[None].double(), w.double(), padding=1)[0]
    f32 = torch.nn.functional.conv2d(x[None], w, padding=1)[0]
    print(f'| {name} | {rel(direct_fp16x3(x, w), want):.2e} | {rel(winograd_fp16x3(x, w), want):.2e} | {rel(f32, want):.2e} |', flush=True)
