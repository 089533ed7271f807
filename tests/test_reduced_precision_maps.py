"""BASELINE.json configs[2] names "bf16 feature maps".  This CPU study pins what 16-bit STORAGE of the post-ReLU feature
maps does to the hot path's results, with everything else exact (fp32 arithmetic, fp32 gradients flowing straight
through the rounding): the oracle's trunk (oracle/st_oracle.py vgg_features, reference style_transfer.py:78-90) is
re-run with every ReLU output rounded to fp16 / bf16 before the next layer, the pooling and the loss taps read it.

Result (asserted below, measured 128^2 / 256^2 photo-like inputs):
  * the seven loss terms move by <= 1e-4 (fp16) / <= 3e-4 (bf16) - rounding noise averages out over the pixels;
  * the image gradient moves by 1.5e-2 ... 5e-2 rel-L2 - 15 ... 50 x the 1e-3 bar of the parity tests.  Per term
    (fp16, 128^2): relu1_1 1.8e-4 (one rounding, no pooling crossed), relu2_1 4e-2, relu3_1 6e-2, relu4_1 9e-2,
    relu5_1 1.1e-1, content 8e-2.  Two causes: rounding creates TIES in the 2x2 max-pool windows and the pool backward
    then routes a pixel's whole gradient to another pixel (average pooling: 4 ... 10 x smaller deviations, second
    test), and the loss gradients are DIFFERENCES (feature - target, cov - target): noise of 2^-12 of a feature is a
    much larger fraction of the difference, and it accumulates over the layers (average pooling still 1e-2).
So a reduced-precision feature-map mode cannot be a drop-in under the reference's tolerance; the shipped path keeps
fp32 maps in HBM and gets its 16-bit MFMA rate from split planes instead
(DESIGN.md 2).  This file is the "empirically stated tolerance" for that configuration."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, '..'))
sys.path.insert(0, os.path.join(HERE, '..', 'style-transfer-pytorch_amd'))

import synth                                    # noqa: E402
from oracle import st_oracle as so              # noqa: E402
from style_transfer import vgg                  # noqa: E402


class _Stored(torch.autograd.Function):
    """x as it would come back from a 16-bit feature map; gradient passes unchanged (fp32 gradient maps)."""

    @staticmethod
    def forward(ctx, x, dtype):
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, grad):
        return grad, None


def _features(image, weights, layers, pooling, dtype):
    feats = {'input': image}
    x = (image - image.new_tensor(so.MEAN).view(1, 3, 1, 1)) / image.new_tensor(so.STD).view(1, 3, 1, 1)
    for idx, op, conv_no in so.layer_program():
        if idx > max(layers):
            break
        if op == 'conv':
            w, b = weights[conv_no]
            x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='replicate'), w, b) if conv_no == 0 else F.conv2d(x, w, b, padding=1)
        elif op == 'relu':
            x = torch.relu(x)
            if dtype is not None:
                x = _Stored.apply(x, dtype)
        elif pooling == 'max':
            x = F.max_pool2d(x, 2)
        else:
            x = F.avg_pool2d(x, 2) * so.POOL_SCALE['average']
        if idx in layers:
            feats[idx] = x
    return feats


def _closure(image, weights, targets, pooling, dtype, monkeypatch):
    monkeypatch.setattr(so, 'vgg_features',
                        lambda img, w, layers, pooling='max': _features(img, w, sorted(set(layers)), pooling, dtype))
    x = image.clone().requires_grad_(True)
    terms, total = so.loss_terms(x, weights, targets, pooling=pooling)
    total.backward()
    monkeypatch.undo()
    return [float(t.detach()) for t in terms], x.grad.detach()


def _deviation(size, pooling, dtype, monkeypatch):
    weights = vgg.synthetic_vgg19_weights(0)
    content, style, image = (synth.smooth_image(s, size, size) for s in (1, 2, 3))
    targets = so.build_targets(content, [style], weights, pooling=pooling)
    t0, g0 = _closure(image, weights, targets, pooling, None, monkeypatch)
    t1, g1 = _closure(image, weights, targets, pooling, dtype, monkeypatch)
    terms = max(abs(a - b) / abs(b) for a, b in zip(t1, t0))
    grad = ((g1 - g0).norm() / g0.norm()).item()
    print(f'[16-bit maps] {size}^2 pooling={pooling} {dtype}: max term deviation {terms:.1e}, gradient rel-L2 {grad:.1e}')
    return terms, grad


@pytest.mark.parametrize('dtype,term_bar', [(torch.float16, 1e-4), (torch.bfloat16, 3e-4)])
def test_16bit_feature_maps_keep_the_losses_but_not_the_gradient(dtype, term_bar, monkeypatch):
    terms, grad = _deviation(128, 'max', dtype, monkeypatch)
    assert terms <= term_bar
    assert 5e-3 < grad < 1e-1            # measured 1.5e-2 (fp16) / 4.2e-2 (bf16); the parity bar is 1e-3


def test_average_pooling_has_no_ties_but_still_misses_the_gradient_bar(monkeypatch):
    terms, grad = _deviation(128, 'average', torch.float16, monkeypatch)
    assert terms <= 1e-4 and 2e-3 < grad < 1.5e-2            # measured 9.6e-3 (max pooling: 1.5e-2)
