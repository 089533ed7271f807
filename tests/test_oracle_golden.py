"""Pin the CPU oracle (oracle/st_oracle.py) against vectors produced by the unmodified reference
(tests/golden/make_golden.py).  CPU only; this is what makes the oracle trustworthy as the checker
for the HIP path."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
import st_oracle as O

torch.set_num_threads(8)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_ns_sqrt_known_answer():
    g = load_golden('ns_kat')
    a, gout = _t(g['a']), _t(g['gout'])
    root = O.ns_sqrt(a, 12)
    assert torch.allclose(root, _t(g['root']), rtol=1e-6, atol=1e-7)
    ga = O.ns_sqrt_bwd(root, gout, 12)
    assert rel_l2(ga, g['ga']) < 1e-6


EVAL_CASES = ['eval_tiny', 'eval_avgpool', 'eval_l2pool', 'eval_s128', 'eval_odd181']


@pytest.mark.parametrize('name', EVAL_CASES)
def test_closure_evaluation_matches_reference(name, vgg_weights):
    g = load_golden(name)
    pooling = str(g['pooling'])
    styles = [_t(g[k]) for k in sorted(k for k in g if k.startswith('style') and k[5:].isdigit())]
    targets = O.build_targets(_t(g['content']), styles, vgg_weights, list(g['style_weights']), pooling)
    terms, total, grad = O.loss_and_grad(_t(g['image']), vgg_weights, targets, pooling=pooling)
    # same operator library, same order of operations: expect agreement to rounding
    assert np.allclose(terms, g['terms'], rtol=2e-6, atol=0), (terms, g['terms'])
    assert abs(total - float(g['total'])) <= 2e-6 * abs(float(g['total']))
    if 'grad' in g:
        assert rel_l2(grad, g['grad']) < 1e-5
    else:
        assert rel_l2(grad.flatten()[::7], g['grad_sub']) < 1e-5
    assert abs(float(grad.double().norm()) - float(g['grad_l2'])) < 1e-5 * float(g['grad_l2'])
    feats = O.vgg_features(_t(g['image']), vgg_weights, O.STYLE_LAYERS + O.CONTENT_LAYERS, pooling)
    for layer in O.STYLE_LAYERS + O.CONTENT_LAYERS:
        assert list(feats[layer].shape) == list(g[f'tap{layer}_shape'])
        assert torch.allclose(feats[layer].flatten()[:512], _t(g[f'tap{layer}_head']), rtol=1e-5, atol=1e-6)
        assert abs(float(feats[layer].double().mean()) - float(g[f'tap{layer}_mean'])) < 1e-6


def test_three_iterations_and_scale_transition(vgg_weights):
    g = load_golden('iter_tiny')
    targets = O.build_targets(_t(g['content']), [_t(g['style0'])], vgg_weights)
    state = O.State(_t(g['image0']))
    trace = []
    for i in range(1, 4):
        _, total = O.iterate(state, vgg_weights, targets)
        trace.append(total)
        if i == 1:
            for key, val in (('image_1', state.image), ('exp_avg_1', state.exp_avg),
                             ('exp_avg_sq_1', state.exp_avg_sq), ('ema_value_1', state.ema_value)):
                assert torch.allclose(val, _t(g[key]), rtol=1e-5, atol=1e-7), key
            assert abs(float(state.ema_accum) - float(g['ema_accum_1'])) < 1e-7
    assert np.allclose(trace, g['trace'], rtol=1e-5)
    assert state.step == int(g['step_3'])
    for key, val in (('image_3', state.image), ('exp_avg_3', state.exp_avg),
                     ('exp_avg_sq_3', state.exp_avg_sq), ('ema_value_3', state.ema_value),
                     ('average_3', state.average())):
        assert torch.allclose(val, _t(g[key]), rtol=1e-4, atol=2e-6), key
    O.rescale_state(state, (57, 68))
    assert state.step == int(g['next_step'])
    for key, val in (('next_image', state.image), ('next_exp_avg', state.exp_avg),
                     ('next_exp_avg_sq', state.exp_avg_sq), ('next_ema_value', state.ema_value)):
        assert torch.allclose(val, _t(g[key]), rtol=1e-4, atol=2e-6), key
    assert abs(float(state.ema_accum) - float(g['next_ema_accum'])) < 1e-7


@pytest.mark.parametrize('name', ['eval_512', 'eval_1024'])
def test_closure_evaluation_matches_reference_at_baseline_sizes(name, vgg_weights):
    """BASELINE configs[1] / [2] sizes: the oracle against the reference's values; inputs regenerated from seeds
    (tests/synth.py) and verified against the checksums the generator recorded."""
    import synth
    g = load_golden(name)
    size, seed, stride = int(g['size']), int(g['seed']), int(g['grad_stride'])
    content, style, image = (synth.smooth_image(seed + i, size, size) for i in range(3))
    for t, key in ((content, 'content_checksum'), (style, 'style_checksum'), (image, 'image_checksum')):
        assert np.array_equal(synth.checksum(t), g[key]), key
    targets = O.build_targets(content, [style], vgg_weights)
    terms, total, grad = O.loss_and_grad(image, vgg_weights, targets)
    assert np.allclose(terms, g['terms'], rtol=5e-6, atol=0), (terms, g['terms'])
    assert abs(total - float(g['total'])) <= 2e-6 * abs(float(g['total']))
    assert rel_l2(grad.flatten()[::stride], g['grad_sub']) < 2e-5
    # every element of the gradient, through the reference's 32 x 32 block moments (<name>_blocks.npz)
    b = load_golden(name + '_blocks')
    block = int(b['block'])
    t = grad.double()[0].reshape(3, size // block, block, size // block, block)
    s1, s2 = t.sum((2, 4)).numpy(), (t * t).sum((2, 4)).numpy()
    scale = float(np.sqrt(b['squares'].sum()))
    assert np.all(np.abs(s1 - b['sums']) <= 2e-5 * (b['abs_sums'] + 1e-3 * scale))
    assert np.all(np.abs(np.sqrt(s2) - np.sqrt(b['squares'])) <= 2e-5 * (np.sqrt(b['squares']) + 1e-4 * scale))
    # the float64 values really are "the same computation, exactly": within the fp32 floor of the fp32 ones
    assert np.all(np.abs(g['terms'] - g['terms64']) <= 5e-4 * np.abs(g['terms64']))


def test_synthetic_image_generator_is_exact_arithmetic():
    """tests/synth.py must give identical bits on every host: integer hashing + exactly rounded float64."""
    import synth
    a = synth.smooth_image(7, 33, 47)
    assert a.shape == (1, 3, 33, 47) and a.dtype == torch.float32
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0
    assert np.array_equal(synth.checksum(a), np.array([4902589818825, 311807966703911])), synth.checksum(a)


def test_fp64_cross_check_of_the_restatement(vgg_weights):
    """The same restatement in fp64 agrees with the fp32 reference vectors to fp32 accuracy."""
    g = load_golden('eval_tiny')
    w64 = [(w.double(), b.double()) for w, b in vgg_weights]
    targets = O.build_targets(_t(g['content']).double(), [_t(g['style0']).double()], w64)
    terms, total, grad = O.loss_and_grad(_t(g['image']).double(), w64, targets)
    assert abs(total - float(g['total'])) < 2e-4 * abs(float(g['total']))
    assert rel_l2(grad, g['grad']) < 5e-3


def test_min_size_error(vgg_weights):
    with pytest.raises(ValueError):
        O.vgg_features(torch.zeros(1, 3, 12, 40), vgg_weights, O.STYLE_LAYERS)
