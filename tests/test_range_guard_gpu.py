"""fp16x3's dynamic-range guard, the activation-aware part (round 4).

Two fp16 planes under ONE power-of-two scale per tensor keep fp32-class accuracy only for elements within ~2^-16 of the
tensor's maximum.  Round 3's guard looks at the WEIGHTS (a channel whose largest weight lies 2^8 above the layer's median
channel); it cannot see a feature map that a few dominant filters push far above its other channels while the consuming
layer's weights for those filters are merely SMALL (max / median of the consumer's weights stays ordinary) - the shape a
trained VGG-19 has (a handful of filters with very large gain), and nobody here can load vgg19-dcbb9e9d.pth to find out.
`st_plan_range_guard` measures instead: every unflagged convolution once in bf16x6 on the operand the shipped pass feeds
it (forward maps, closure gradients), and flags what disagrees.  Cold path; zero cost in the hot loop."""
import pytest
import torch

import st_oracle as O
from conftest import rel_l2
from test_hot_path_gpu import DEV, GRAD_TOL, _check_terms, _smooth, _term_tols

pytestmark = pytest.mark.gpu
NON_TAPS = [1, 3, 5, 6, 7, 10, 11]            # convs whose output is no tap: may be rescaled without changing any tap


def _dominant_filters(weights, log2_gain=22, per_layer=1, seed=3):
    """The same network function with `per_layer` dominant filters in every non-tap convolution: output channel c (weights
    and bias) times 2^log2_gain, the next convolution's input channel c divided by it.  ReLU and pooling commute with a
    positive factor, so every tap, loss term and gradient is unchanged in exact arithmetic."""
    g = torch.Generator().manual_seed(seed)
    out = [(w.clone(), b.clone()) for w, b in weights]
    gain = float(2 ** log2_gain)
    for i in NON_TAPS:
        cout = out[i][0].shape[0]
        pick = torch.randperm(cout, generator=g)[:per_layer]
        sc = torch.ones(cout)
        sc[pick] = gain
        out[i] = (out[i][0] * sc.view(-1, 1, 1, 1), out[i][1] * sc)
        out[i + 1] = (out[i + 1][0] / sc.view(1, -1, 1, 1), out[i + 1][1])
    return out


def _vgg_like_gains(weights, sigma=1.2, dominant=2, log2_dominant=10, seed=9):
    """Per-channel gains of every non-tap layer drawn log-normally (sigma in natural-log units: e^(+-3 sigma) ~ 36 x either
    way) plus `dominant` filters at 2^log2_dominant, compensated in the consumer - the spread a trained, unnormalised VGG
    carries - function-preserving as above."""
    g = torch.Generator().manual_seed(seed)
    out = [(w.clone(), b.clone()) for w, b in weights]
    for i in NON_TAPS:
        cout = out[i][0].shape[0]
        sc = torch.exp(torch.randn(cout, generator=g) * sigma)
        sc[torch.randperm(cout, generator=g)[:dominant]] *= float(2 ** log2_dominant)
        out[i] = (out[i][0] * sc.view(-1, 1, 1, 1), out[i][1] * sc)
        out[i + 1] = (out[i + 1][0] / sc.view(1, -1, 1, 1), out[i + 1][1])
    return out


def _targets(hip, net, plan, content, style, guard):
    """What StyleTransfer._build_targets does: (guard the forward on each image,) content target, style targets."""
    flagged = []
    if guard:
        flagged.append(plan.range_guard(content.to(DEV)))
    plan.forward(content.to(DEV), 22)
    plan.set_content_target_from_forward()
    sp = hip.Plan(net, *style.shape[2:])
    if guard:
        flagged.append(sp.range_guard(style.to(DEV)))
    sp.forward(style.to(DEV), 29)
    for i, layer in enumerate(O.STYLE_LAYERS):
        plan.set_style_target(i, *sp.moments(layer))
    plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
    return flagged


def _worst(losses, terms, terms64):
    got = losses.cpu().double().numpy()
    tols = _term_tols(terms, terms64)
    return max(abs(got[k] - terms[k]) / abs(terms[k]) / tols[k] for k in range(7))


def test_dominant_filters_are_caught_by_the_activation_guard(vgg_weights):
    """One filter per non-tap layer with a gain of 2^22 and a consumer whose weights for it are 2^-22: the weights-only guard
    sees nothing in the consumers' FORWARD (max / median of their input-channel weights is ordinary), plain fp16x3 loses the
    other channels' low bits and misses the fp32 tolerances; the activation-aware guard flags exactly the consumers and the
    closure is back inside them."""
    from style_transfer import _hip as hip
    size = 192
    content, style, image = _smooth(61, size, size), _smooth(62, size, size), _smooth(63, size, size)
    weights = _dominant_filters(vgg_weights)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    targets = O.build_targets(content, [style], weights)
    terms, total, grad = O.loss_and_grad(image, weights, targets)
    w64 = [(w.double(), b.double()) for w, b in weights]
    terms64, _, grad64 = O.loss_and_grad(image.double(), w64, O.build_targets(content.double(), [style.double()], w64))

    net = hip.Net(weights, 'max', DEV, 'fp16x3')
    wide_f0, wide_b0 = net.wide_layers()
    consumers = [i + 1 for i in NON_TAPS]
    assert not any(wide_f0[i] for i in consumers), 'the weights-only guard was not supposed to see the consumers'
    plan = hip.Plan(net, size, size)
    _targets(hip, net, plan, content, style, guard=False)
    losses, g = plan.loss_and_grad(image.to(DEV))
    bad_terms, bad_grad = _worst(losses, terms, terms64), rel_l2(g.cpu(), grad)
    print(f'[range] dominant filters, NO activation guard: worst term at {bad_terms:.1f} x its tolerance, gradient rel-L2 '
          f'{bad_grad:.2e}')
    assert bad_terms > 1.0 or bad_grad > GRAD_TOL, 'plain fp16x3 was expected to miss the tolerances here'

    flagged = _targets(hip, net, plan, content, style, guard=True)
    flagged.append(plan.range_guard(image.to(DEV)))                 # forward again on the iterate + the closure's gradients
    fwd = sorted({i for f, _ in flagged for i, v in enumerate(f) if v})
    bwd = sorted({i for _, b in flagged for i, v in enumerate(b) if v})
    print(f'[range] dominant filters: activation guard flagged forward of convs {fwd}, data gradient of convs {bwd}; '
          f'weights-only guard had forward {[i for i, v in enumerate(wide_f0) if v]}, data gradient '
          f'{[i for i, v in enumerate(wide_b0) if v]}')
    assert set(consumers) <= set(fwd) | {i for i, v in enumerate(wide_f0) if v}
    losses, g = plan.loss_and_grad(image.to(DEV))
    losses, g = losses.clone(), g.clone()
    _check_terms('dominant192/fp16x3+guard', losses, terms, total, terms64)
    err = rel_l2(g.cpu(), grad)
    print(f'[range] dominant filters WITH the guard: gradient rel-L2 {err:.2e} (cpu32-vs-fp64 {rel_l2(grad, grad64):.2e})')
    assert err <= GRAD_TOL
    # idempotent: a second call finds nothing new
    assert plan.range_guard(image.to(DEV)) == ([0] * 13, [0] * 13)


def test_vgg_like_gain_statistics_pass_with_the_guards(vgg_weights):
    """Log-normal per-channel gains plus a few dominant filters per layer (what an unnormalised trained network looks like):
    with the guards the closure meets the fp32 tolerances and stays where the unscaled network puts it."""
    from style_transfer import _hip as hip
    size = 192
    content, style, image = _smooth(71, size, size), _smooth(72, size, size), _smooth(73, size, size)
    weights = _vgg_like_gains(vgg_weights)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    targets = O.build_targets(content, [style], weights)
    terms, total, grad = O.loss_and_grad(image, weights, targets)
    w64 = [(w.double(), b.double()) for w, b in weights]
    terms64, _, _ = O.loss_and_grad(image.double(), w64, O.build_targets(content.double(), [style.double()], w64))
    net = hip.Net(weights, 'max', DEV, 'fp16x3')
    plan = hip.Plan(net, size, size)
    flagged = _targets(hip, net, plan, content, style, guard=True)
    flagged.append(plan.range_guard(image.to(DEV)))
    wf, wb = net.wide_layers()
    print(f'[range] VGG-like gains: bf16x6 forward of convs {[i for i, v in enumerate(wf) if v]}, data gradient of convs '
          f'{[i for i, v in enumerate(wb) if v]} (activation guard added forward '
          f'{sorted({i for f, _ in flagged for i, v in enumerate(f) if v})}, data gradient '
          f'{sorted({i for _, b in flagged for i, v in enumerate(b) if v})})')
    losses, g = plan.loss_and_grad(image.to(DEV))
    losses, g = losses.clone(), g.clone()
    _check_terms('vgglike192/fp16x3+guard', losses, terms, total, terms64)
    err = rel_l2(g.cpu(), grad)
    print(f'[range] VGG-like gains: gradient rel-L2 {err:.2e}')
    assert err <= GRAD_TOL


@pytest.mark.parametrize('size', [128, 512])
def test_the_guard_flags_nothing_on_the_benchmark_network(size, vgg_weights):
    """The seeded synthetic weights every measured number uses: the activation-aware guard must not change the arithmetic."""
    from style_transfer import _hip as hip
    content, style, image = _smooth(81, size, size), _smooth(82, size, size), _smooth(83, size, size)
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    plan = hip.Plan(net, size, size)
    flagged = _targets(hip, net, plan, content, style, guard=True)
    flagged.append(plan.range_guard(image.to(DEV)))
    assert all(f == [0] * 13 and b == [0] * 13 for f, b in flagged), flagged
    assert net.wide_layers() == ([0] * 13, [0] * 13)
