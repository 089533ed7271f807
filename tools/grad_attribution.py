#!/usr/bin/env python3
"""the image gradient's worst 32 x 32 block deviates more the larger the image (1.3e-4 at 512^2
... 7.6e-4 at 2896 x 2172 against a 1e-3 bar) while the oracle stays <= 2e-5 from the reference - which arithmetic carries it?
One closure per switch on the reference goldens (tests/golden/eval_*), worst-block L2 and rel-L2 of the gradient sample:

    gpurun -- python tools/grad_attribution.py [--cases eval_512,eval_2048,eval_2896x2172]
"""
import argparse
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ('style-transfer-pytorch_amd', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(R, sub))
import numpy as np
import torch

from style_transfer import _hip, vgg
import synth
import st_oracle as O

DEV = 'cuda:0'


def golden(name):
    return dict(np.load(os.path.join(R, 'tests', 'golden', name + '.npz'), allow_pickle=False))


def blocks(grad, b):
    block = int(b['block'])
    g = grad.double()[0]
    c, h, w = g.shape
    hb, wb = -(-h // block), -(-w // block)
    pad = torch.zeros(c, hb * block, wb * block, dtype=torch.float64, device=g.device)
    pad[:, :h, :w] = g
    t = pad.reshape(c, hb, block, wb, block)
    s2 = (t * t).sum((2, 4)).cpu().numpy()
    r2 = b['squares']
    scale = float(np.sqrt(r2.sum()))
    l2 = np.abs(np.sqrt(s2) - np.sqrt(r2)) / (np.sqrt(r2) + 1e-3 * scale / np.sqrt(r2.size))
    return float(l2.max()), np.unravel_index(l2.argmax(), l2.shape)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='eval_512,eval_1024,eval_2048,eval_2896x2172')
    a = ap.parse_args()
    weights = vgg.synthetic_vgg19_weights(0)
    configs = [
        ('shipped (fp16x3 convs, fp16x3 Gram + 1x1, fp16x3 reduced Lyapunov chain at n = 512)', 'fp16x3', {}),
        ('exact fp32 trunk convolutions', 'fp32', {}),
        ('fp32 Gram (moments)', 'fp16x3', dict(ST_GRAM_F32=1)),
        ('fp32 heads 1x1 step (dF = Ssym F + b)', 'fp16x3', dict(ST_HEAD_1X1_F32=1)),
        ('fp32 backward chains (ST_NS_F16=0)', 'fp16x3', dict(ST_NS_F16=0)),
        ('full Lyapunov recurrence, fp32 (ST_NS_FULL_BACKWARD=1)', 'fp16x3', dict(ST_NS_FULL_BACKWARD=1)),
        ('fp32 Gram + fp32 1x1 + fp32 chains', 'fp16x3', dict(ST_GRAM_F32=1, ST_HEAD_1X1_F32=1, ST_NS_F16=0)),
        ('everything fp32 (trunk + heads)', 'fp32', dict(ST_NS_F16=0)),
        ('shallow heads not in lockstep (per-head K split)', 'fp16x3', dict(ST_HEAD_LOCKSTEP=0)),
    ]
    for case in a.cases.split(','):
        g, b = golden(case), golden(case + '_blocks')
        seed, stride = int(g['seed']), int(g['grad_stride'])
        height, width = (int(g['height']), int(g['width'])) if 'height' in g else (int(g['size']), int(g['size']))
        content, style, image = (synth.smooth_image(seed + i, height, width) for i in range(3))
        print(f'## {case} ({width} x {height})\n')
        print('| arithmetic | worst 32 x 32 block L2 (block) | gradient sample rel-L2 | |g| rel | relu1_1 .. relu5_1 term rel to reference |')
        print('|---|---:|---:|---:|---|')
        ref_sub = torch.from_numpy(g['grad_sub'])
        for label, prec, opts in configs:
            with _hip.options(**opts):
                net = _hip.Net(weights, 'max', DEV, prec)
                plan = _hip.Plan(net, height, width)
                plan.forward(content.to(DEV), 22)
                plan.set_content_target_from_forward()
                plan.forward(style.to(DEV), 29)
                for i, layer in enumerate(O.STYLE_LAYERS):
                    plan.set_style_target(i, *plan.moments(layer))
                plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
                losses, grad = plan.loss_and_grad(image.to(DEV))
                torch.cuda.synchronize()
            worst, where = blocks(grad, b)
            gc = grad.cpu()
            sub = gc.flatten()[::stride]
            err = float((sub.double() - ref_sub.double()).norm() / ref_sub.double().norm())
            nerr = abs(float(gc.double().norm()) - float(g['grad_l2'])) / float(g['grad_l2'])
            terms = losses.cpu().double().numpy()
            rel = [abs(terms[k] - g['terms'][k]) / abs(g['terms'][k]) for k in range(1, 6)]
            print(f'| {label} | {worst:.2e} {tuple(int(x) for x in where)} | {err:.2e} | {nerr:.1e} | '
                  f'{" ".join("%.1e" % r for r in rel)} |', flush=True)
            del plan, net
            torch.cuda.empty_cache()
        print()


if __name__ == '__main__':
    main()
