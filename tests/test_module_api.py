"""The reference's importable module surface (style_transfer.py:93-234, sqrtm.py) in the drop-in package: the names
import from the same places, and a loss graph assembled from them the way the reference's stylize() assembles
its own (:425-456) reproduces the reference-generated golden vectors.  CPU only (plain torch modules)."""
import copy

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
import st_oracle as O

torch.set_num_threads(8)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_reference_names_import_from_the_same_places():
    import style_transfer
    from style_transfer import sqrtm                                         # noqa: F401
    from style_transfer.style_transfer import (EMA, ContentLoss, ContentLossMSE, LayerApply, Scale,   # noqa: F401
                                               ScaledMSELoss, STIterate, StyleLoss, StyleLossW2, StyleTransfer,
                                               SumLoss, TVLoss, VGGFeatures, eye_like, gen_scales, interpolate,
                                               scale_adam, size_to_fit)
    for name in ('sqrtm_ns', 'sqrtm_ns_lyap', 'sqrtm_eig'):
        assert callable(getattr(style_transfer.sqrtm, name))


def _reference_style_graph(content_feat, style_moments, content_weight=0.015, tv_weight=2.0):
    """The module graph of reference stylize(), :376,427-455."""
    from style_transfer.style_transfer import ContentLossMSE, LayerApply, Scale, StyleLossW2, SumLoss, TVLoss
    tv = Scale(LayerApply(TVLoss(), 'input'), tv_weight)
    content = [Scale(LayerApply(ContentLossMSE(content_feat), 22), content_weight)]
    style = [Scale(LayerApply(StyleLossW2(style_moments[layer]), layer), w)
             for layer, w in zip(O.STYLE_LAYERS, O.STYLE_LAYER_WEIGHTS)]
    return SumLoss([*content, *style, tv])


@pytest.mark.parametrize('name', ['eval_tiny', 'eval_s128'])
def test_module_graph_reproduces_reference_closure(name, vgg_weights):
    from style_transfer.style_transfer import StyleLossW2
    g = load_golden(name)
    styles = [_t(g[k]) for k in sorted(k for k in g if k.startswith('style') and k[5:].isdigit())]
    sw = list(g['style_weights'])
    with torch.no_grad():
        cfeat = O.vgg_features(_t(g['content']), vgg_weights, [22])[22]
        blended = {}
        for img, w in zip(styles, sw):
            feats = O.vgg_features(img, vgg_weights, O.STYLE_LAYERS)
            for layer in O.STYLE_LAYERS:
                mean, srm = StyleLossW2.get_target(feats[layer])
                mean, srm = mean * w, srm * w
                if layer in blended:
                    blended[layer][0] += mean
                    blended[layer][1] += srm
                else:
                    blended[layer] = [mean, srm]
    crit = _reference_style_graph(cfeat, blended)
    image = _t(g['image']).clone().requires_grad_(True)
    feats = O.vgg_features(image, vgg_weights, O.STYLE_LAYERS + [22])
    terms = [float(member(feats).detach()) for member in crit]
    total = crit(feats)
    total.backward()
    assert np.allclose(terms, g['terms'], rtol=3e-6, atol=0), (terms, g['terms'])
    assert abs(float(total) - float(g['total'])) <= 3e-6 * abs(float(g['total']))
    ref_grad = g['grad'] if 'grad' in g else None
    if ref_grad is not None:
        assert rel_l2(image.grad, ref_grad) < 2e-5
    else:
        assert rel_l2(image.grad.flatten()[::7], g['grad_sub']) < 2e-5


def test_sqrtm_module_against_reference_known_answer():
    from style_transfer import sqrtm
    g = load_golden('ns_kat')
    a = _t(g['a']).clone().requires_grad_(True)
    root = sqrtm.sqrtm_ns_lyap(a, num_iters=12)
    assert torch.allclose(root.detach(), _t(g['root']), rtol=1e-6, atol=1e-7)
    root.backward(_t(g['gout']))
    assert rel_l2(a.grad, g['ga']) < 1e-6
    assert torch.allclose(sqrtm.sqrtm_ns(_t(g['a']), 12), _t(g['root']), rtol=1e-6, atol=1e-7)
    # the oracle's restatement and the module are the same recurrences
    assert torch.equal(sqrtm.sqrtm_ns(_t(g['a']), 12), O.ns_sqrt(_t(g['a']), 12))
    with pytest.raises(RuntimeError):
        sqrtm.sqrtm_ns(torch.ones(3))
    with pytest.raises(RuntimeError):
        sqrtm.sqrtm_ns(torch.ones(3, 4))
    with pytest.raises(RuntimeError):
        sqrtm.sqrtm_ns_lyap(torch.eye(3), num_iters=2, num_iters_backward=-1)


def test_sqrtm_eig_root_and_gradient():
    from style_transfer import sqrtm
    gen = torch.Generator().manual_seed(5)
    m = torch.randn((6, 6), generator=gen, dtype=torch.float64)
    a = (m @ m.T + 0.5 * torch.eye(6, dtype=torch.float64)).requires_grad_(True)
    root = sqrtm.sqrtm_eig(a)
    assert torch.allclose(root @ root, a, rtol=1e-10, atol=1e-10)
    # the Sylvester solve of the backward: root X + X root = G
    gout = torch.randn((6, 6), generator=gen, dtype=torch.float64)
    gout = gout + gout.T
    (x,) = torch.autograd.grad(root, a, gout)
    assert torch.allclose(root.detach() @ x + x @ root.detach(), gout, rtol=1e-9, atol=1e-9)
    # NS with many iterations converges to the same root on a well-conditioned matrix
    assert torch.allclose(sqrtm.sqrtm_ns(a.detach(), 40), root.detach(), rtol=1e-8, atol=1e-8)


def test_scaled_losses_and_gram_style_loss():
    from style_transfer.style_transfer import ContentLoss, ScaledMSELoss, StyleLoss, eye_like
    gen = torch.Generator().manual_seed(3)
    x, y = torch.randn((1, 4, 5, 6), generator=gen), torch.randn((1, 4, 5, 6), generator=gen)
    d = x - y
    want = d.pow(2).sum() / (d.abs().sum() + 1e-8)
    assert torch.allclose(ScaledMSELoss()(x, y), want)
    assert torch.allclose(ContentLoss(y)(x), want)
    assert 'eps=1e-08' in repr(ScaledMSELoss())
    gram = StyleLoss.get_target(y)
    flat = y.flatten(-2)
    assert torch.allclose(gram, flat @ flat.transpose(-2, -1) / 30)
    gx = StyleLoss.get_target(x)
    dg = gx - gram
    assert torch.allclose(StyleLoss(gram)(x), dg.pow(2).sum() / (dg.abs().sum() + 1e-8))
    assert torch.equal(eye_like(torch.zeros(2, 3, 3)), torch.eye(3).expand(2, 3, 3))


def test_scale_adam_matches_reference_transition(vgg_weights):
    """scale_adam on a torch.optim.Adam state_dict against the reference's own scale transition (iter_tiny)."""
    from style_transfer.style_transfer import scale_adam
    g = load_golden('iter_tiny')
    image = _t(g['image_3']).clone().requires_grad_(True)
    opt = torch.optim.Adam([image], lr=0.02, betas=(0.9, 0.99))
    image.grad = torch.zeros_like(image)
    opt.step()                                                      # creates the state entries
    state = opt.state_dict()
    state['state'][0]['exp_avg'] = _t(g['exp_avg_3']).clone()
    state['state'][0]['exp_avg_sq'] = _t(g['exp_avg_sq_3']).clone()
    before = copy.deepcopy(state)
    new = scale_adam(state, (57, 68))
    assert torch.allclose(new['state'][0]['exp_avg'], _t(g['next_exp_avg']), rtol=1e-5, atol=1e-7)
    assert torch.allclose(new['state'][0]['exp_avg_sq'], _t(g['next_exp_avg_sq']), rtol=1e-5, atol=1e-9)
    assert float(new['state'][0]['exp_avg_sq'].min()) >= 0
    assert torch.equal(state['state'][0]['exp_avg'], before['state'][0]['exp_avg'])     # input untouched
