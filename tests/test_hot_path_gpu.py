"""Hot-path parity on a real MI355X: the closure (7 loss terms + image gradient), the fused
Adam/clamp/EMA iteration and the end-to-end ``stylize()`` against (a) the golden vectors recorded
from the unmodified reference and (b) the CPU oracle run live on the same seeded inputs.

Stated tolerances (fp32; see DESIGN.md "Parity"):
  * each weighted loss term and the total: 1e-4 relative to the reference value, except that a
    style term may use the measured fp32 rounding floor of the Newton-Schulz chain (3e-4), which the
    fp64 cross-check in tests/test_oracle_golden.py bounds at ~1e-4 for the reference itself;
  * image gradient: rel-L2 <= 1e-3 (gradients are held to 1e-3, not 1e-4);
  * post-step image / Adam moments / EMA: max-abs 2e-5 on O(1) quantities after one step.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
import st_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TERM_TOL = [1e-4, 3e-4, 3e-4, 3e-4, 3e-4, 3e-4, 1e-4]
TOTAL_TOL = 1e-4
GRAD_TOL = 1e-3
GRAD_TOL_BF16X3 = 5e-3     # opt-in approximate conv arithmetic (measured 2-3e-3)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _build_plan(hip, weights, content, styles, style_w, pooling='max', precision='fp32'):
    net = hip.Net(weights, pooling, DEV, precision)
    h, w = content.shape[2:]
    plan = hip.Plan(net, h, w)
    plan.forward(content.to(DEV), 22)
    plan.set_content_target_from_forward()
    blended = {}
    for s, sw in zip(styles, style_w):
        sp = hip.Plan(net, *s.shape[2:])
        sp.forward(s.to(DEV), 29)
        for layer in O.STYLE_LAYERS:
            mean, srm = sp.moments(layer)
            mean, srm = mean * sw, srm * sw
            if layer not in blended:
                blended[layer] = [mean, srm]
            else:
                blended[layer][0] += mean
                blended[layer][1] += srm
    for i, layer in enumerate(O.STYLE_LAYERS):
        plan.set_style_target(i, *blended[layer])
    plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
    return net, plan


def _check_terms(name, losses, want_terms, want_total):
    got = losses.cpu().double().numpy()
    for k in range(7):
        rel = abs(got[k] - want_terms[k]) / abs(want_terms[k])
        print(f'[parity] {name} term[{O.TERM_NAMES[k]}]: got {got[k]:.8g} want {want_terms[k]:.8g} rel={rel:.2e}')
    rel_total = abs(got[7] - want_total) / abs(want_total)
    print(f'[parity] {name} total: got {got[7]:.8g} want {want_total:.8g} rel={rel_total:.2e}')
    for k in range(7):
        rel = abs(got[k] - want_terms[k]) / abs(want_terms[k])
        assert rel <= TERM_TOL[k], f'{name}: term {O.TERM_NAMES[k]} rel {rel:.2e} > {TERM_TOL[k]:.0e}'
    assert rel_total <= TOTAL_TOL, f'{name}: total rel {rel_total:.2e}'


@pytest.mark.parametrize('name', ['eval_tiny', 'eval_avgpool', 'eval_l2pool', 'eval_s128', 'eval_odd181'])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x6', 'fp16x3', 'bf16x3'])
def test_closure_against_reference_goldens(name, precision, vgg_weights):
    """Same tolerances in every conv arithmetic mode: the split-precision paths must meet the fp32 bar."""
    from style_transfer import _hip as hip
    g = load_golden(name)
    pooling = str(g['pooling'])
    styles = [_t(g[k]) for k in sorted(k for k in g if k.startswith('style') and k[5:].isdigit())]
    net, plan = _build_plan(hip, vgg_weights, _t(g['content']), styles, list(g['style_weights']), pooling,
                            precision)
    name = f'{name}/{precision}'
    losses, grad = plan.loss_and_grad(_t(g['image']).to(DEV))
    torch.cuda.synchronize()
    _check_terms(name, losses, g['terms'], float(g['total']))
    if 'grad' in g:
        err = rel_l2(grad.cpu(), g['grad'])
    else:
        err = rel_l2(grad.cpu().flatten()[::7], g['grad_sub'])
    print(f'[parity] {name} image gradient rel_l2={err:.3e}')
    # bf16x3 is the documented APPROXIMATE mode: it meets the loss bar but not the 1e-3 gradient bar
    assert err <= (GRAD_TOL_BF16X3 if precision == 'bf16x3' else GRAD_TOL)


def _smooth(seed, h, w):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((1, 3, max(h // 16, 2), max(w // 16, 2)), generator=g)
    img = torch.nn.functional.interpolate(low, (h, w), mode='bicubic', align_corners=False)
    return (img + (torch.rand((1, 3, h, w), generator=g) - 0.5) * (24 / 255)).clamp(0, 1).contiguous()


@pytest.mark.parametrize('kind,precision', [('photo_like', 'fp32'), ('white_noise', 'fp32'),
                                            ('photo_like', 'bf16x6'), ('white_noise', 'bf16x6'),
                                            ('photo_like', 'fp16x3'), ('white_noise', 'fp16x3'),
                                            ('photo_like', 'bf16x3')])
def test_closure_against_live_oracle_256(kind, precision, vgg_weights):
    """BASELINE config 1 size (256x256): oracle evaluated here on the host, HIP path on the GPU.

    The style terms go through the non-converged NS-12 recurrence, whose fp32 evaluation has a
    rounding floor: the reference's own fp32 value deviates from exact arithmetic by d_k (measured
    here with the fp64 run of the same oracle).  A term passes within max(1e-4, 3 d_k) of the fp32
    reference; the total must be within 1e-4 regardless.  'white_noise' is the stress case (every
    covariance is near-singular); 'photo_like' is the realistic one."""
    from style_transfer import _hip as hip
    if kind == 'photo_like':
        content, style, image = _smooth(21, 256, 256), _smooth(22, 200, 256), _smooth(23, 256, 256)
    else:
        gen = torch.Generator().manual_seed(21)
        content = torch.rand((1, 3, 256, 256), generator=gen)
        style = torch.rand((1, 3, 200, 256), generator=gen)
        image = torch.rand((1, 3, 256, 256), generator=gen)
    targets = O.build_targets(content, [style], vgg_weights)
    terms, total, grad = O.loss_and_grad(image, vgg_weights, targets)
    w64 = [(w.double(), b.double()) for w, b in vgg_weights]
    t64 = O.build_targets(content.double(), [style.double()], w64)
    terms64, total64, grad64 = O.loss_and_grad(image.double(), w64, t64)
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision=precision)
    kind = f'{kind}/{precision}'
    losses, g = plan.loss_and_grad(image.to(DEV))
    got = losses.cpu().double().numpy()
    for k in range(7):
        floor = abs(terms[k] - terms64[k]) / abs(terms64[k])
        rel = abs(got[k] - terms[k]) / abs(terms[k])
        rel64 = abs(got[k] - terms64[k]) / abs(terms64[k])
        print(f'[parity] live256/{kind} {O.TERM_NAMES[k]}: hip-vs-cpu32 {rel:.2e}  hip-vs-fp64 {rel64:.2e}  '
              f'cpu32-vs-fp64 {floor:.2e}')
        assert rel <= max(1e-4, 3 * floor), (kind, O.TERM_NAMES[k], rel, floor)
    rel_total = abs(got[7] - total) / abs(total)
    print(f'[parity] live256/{kind} total rel={rel_total:.2e}')
    assert rel_total <= TOTAL_TOL
    err, floor_g = rel_l2(g.cpu(), grad), rel_l2(grad, grad64)
    print(f'[parity] live256/{kind} image gradient rel_l2={err:.3e} (cpu32-vs-fp64 {floor_g:.3e})')
    assert err <= (GRAD_TOL_BF16X3 if precision == 'bf16x3' else GRAD_TOL)


def test_three_iterations_against_reference(vgg_weights):
    from style_transfer import _hip as hip
    g = load_golden('iter_tiny')
    net, plan = _build_plan(hip, vgg_weights, _t(g['content']), [_t(g['style0'])], [1.0])
    image = _t(g['image0']).to(DEV).clone()
    m, v = torch.zeros_like(image), torch.zeros_like(image)
    ema = torch.zeros_like(image)
    decay = torch.tensor(0.99)
    ema += (1 - decay).to(DEV) * image            # EMA.__init__ -> update(image)
    trace = []
    for step in range(1, 4):
        losses = plan.step(image, m, v, ema, step, 0.02)
        trace.append(float(losses[7]))
        if step == 1:
            for key, val, tol in (('image_1', image, 2e-5), ('exp_avg_1', m, 1e-6), ('exp_avg_sq_1', v, 1e-8),
                                  ('ema_value_1', ema, 1e-6)):
                d = float((val.cpu() - _t(g[key])).abs().max())
                print(f'[parity] step1 {key}: max_abs={d:.3e}')
                assert d <= tol, key
    print('[parity] loss trace', trace, 'reference', list(g['trace']))
    assert np.allclose(trace, g['trace'], rtol=2e-4)
    for key, val, tol in (('image_3', image, 2e-3), ('exp_avg_3', m, 1e-4), ('ema_value_3', ema, 1e-4)):
        d = float((val.cpu() - _t(g[key])).abs().max())
        print(f'[parity] step3 {key}: max_abs={d:.3e}')
        assert d <= tol, key


def test_stylize_end_to_end_against_reference(vgg_weights):
    """Drop-in API: same call as the reference's stylize(); compare the callback trace and result."""
    from PIL import Image
    import style_transfer as st_pkg
    g = load_golden('stylize_e2e')
    content = Image.fromarray(g['content_u8'], 'RGB')
    style = Image.fromarray(g['style_u8'], 'RGB')
    st = st_pkg.StyleTransfer(devices=[DEV], weights=vgg_weights)
    its = []
    torch.manual_seed(0)
    st.stylize(content, [style], min_scale=45, end_scale=64, iterations=3, initial_iterations=4,
               callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)))
    got = np.array(its, dtype=np.float64)
    want = g['iterates']
    print('[parity] stylize trace got ', got[:, 4])
    print('[parity] stylize trace want', want[:, 4])
    assert got.shape == want.shape and np.array_equal(got[:, :4], want[:, :4])
    assert np.allclose(got[:4, 4], want[:4, 4], rtol=5e-4)          # first scale: identical inputs
    assert np.allclose(got[4:, 4], want[4:, 4], rtol=2e-2)          # after GPU bicubic resampling (cold path)
    res = st.get_image_tensor().cpu()
    d = float((res - _t(g['result'])).abs().max())
    print(f'[parity] stylize result max_abs={d:.3e}')
    assert d < 2e-2
    assert st.get_image('pil').size == (64, 64)
    assert st.get_image('np_uint16').dtype == np.uint16
    with pytest.raises(ValueError):
        st.get_image('bmp')


def test_full_size_properties_512(vgg_weights):
    """BASELINE config 2 size: determinism, finiteness and linearity-in-weights at 512x512, where the
    CPU oracle would take too long for a unit test."""
    from style_transfer import _hip as hip
    gen = torch.Generator().manual_seed(5)
    content = torch.rand((1, 3, 512, 512), generator=gen)
    style = torch.rand((1, 3, 512, 512), generator=gen)
    image = torch.rand((1, 3, 512, 512), generator=gen).to(DEV)
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0])
    l1, g1 = plan.loss_and_grad(image)
    l1, g1 = l1.clone(), g1.clone()
    l2, g2 = plan.loss_and_grad(image)
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all()
    assert torch.equal(l1, l2) and torch.equal(g1, g2), 'hot path must be run-to-run deterministic'
    # doubling every Scale factor doubles every term and the gradient (exactly: powers of two)
    plan.set_loss_weights(0.03, [2 * w for w in O.STYLE_LAYER_WEIGHTS], 4.0)
    l3, g3 = plan.loss_and_grad(image)
    assert torch.allclose(l3, 2 * l1, rtol=1e-6)
    assert rel_l2(g3.cpu(), (2 * g1).cpu()) < 1e-6


def test_graph_replay_is_bit_identical_to_eager_launches(vgg_weights):
    """The closure is captured into a hipGraph on its second call; replay must not change a bit."""
    from style_transfer import _hip as hip
    g = load_golden('eval_s128')
    styles = [_t(g[k]) for k in sorted(k for k in g if k.startswith('style') and k[5:].isdigit())]
    net, plan = _build_plan(hip, vgg_weights, _t(g['content']), styles, list(g['style_weights']))
    image = _t(g['image']).to(DEV)
    grad = torch.empty_like(image)
    plan.set_graph(False)
    l0, g0 = plan.loss_and_grad(image, grad)
    l0, g0 = l0.clone(), g0.clone()
    plan.set_graph(True)
    for call in range(4):                       # eager warm-up, capture + launch, replay, replay
        grad.zero_()
        l1, g1 = plan.loss_and_grad(image, grad)
        assert torch.equal(l1, l0) and torch.equal(g1, g0), f'call {call} differs from eager'
    # a changed input through the same pointer must be seen by the replayed graph
    image.mul_(0.5)
    l2, _ = plan.loss_and_grad(image, grad)
    assert not torch.equal(l2, l0)
