"""Host-side (Python) logic of the drop-in: pyramid helpers, signature compatibility, EMA and the
scale transition of the optimiser state - checked on CPU tensors against values recorded from the
reference (tests/golden) or stated by its code."""
import inspect

import numpy as np
import torch

from conftest import load_golden
from style_transfer import style_transfer as S
from style_transfer import vgg


def test_gen_scales_and_size_to_fit():
    assert S.gen_scales(128, 512) == [128, 181, 256, 362, 512]
    assert S.gen_scales(128, 1024) == [128, 181, 256, 362, 512, 724, 1024]
    assert S.gen_scales(45, 64) == [45, 64]
    assert S.gen_scales(256, 256) == [256]
    assert S.size_to_fit((4000, 3000), 2896, scale_up=True) == (2896, 2172)
    assert S.size_to_fit((64, 64), 45, scale_up=True) == (45, 45)
    assert S.size_to_fit((72, 56), 64) == (64, 50)
    assert S.size_to_fit((72, 56), 128) == (72, 56)                      # never upscales styles
    assert S.size_to_fit((56, 72), 64) == (50, 64)


def test_stylize_signature_is_the_reference_one():
    """cli.py builds its options from these (reference cli.py:150-153)."""
    kw = S.StyleTransfer.stylize.__kwdefaults__
    assert kw == {'style_weights': None, 'content_weight': 0.015, 'tv_weight': 2.0, 'optimizer': 'adam',
                  'min_scale': 128, 'end_scale': 512, 'iterations': 500, 'initial_iterations': 1000,
                  'step_size': 0.02, 'avg_decay': 0.99, 'init': 'content', 'style_scale_fac': 1.0,
                  'style_size': None, 'callback': None}
    ann = S.StyleTransfer.stylize.__annotations__
    assert ann == {'content_weight': float, 'tv_weight': float, 'optimizer': str, 'min_scale': int,
                   'end_scale': int, 'iterations': int, 'initial_iterations': int, 'step_size': float,
                   'avg_decay': float, 'init': str, 'style_scale_fac': float, 'style_size': int}
    params = list(inspect.signature(S.StyleTransfer.stylize).parameters)
    assert params[:3] == ['self', 'content_image', 'style_images']
    fields = [f for f in S.STIterate.__dataclass_fields__]
    assert fields == ['w', 'h', 'i', 'i_max', 'loss', 'time', 'gpu_ram']


def test_ema_matches_reference_buffers():
    g = load_golden('iter_tiny')
    img = torch.from_numpy(g['image0'])
    ema = S.EMA(img, 0.99)
    assert abs(float(ema.accum) - float(np.float32(0.99))) < 1e-9
    assert torch.allclose(ema.value, (1 - torch.tensor(0.99)) * img)
    assert torch.allclose(ema.get(), img, rtol=1e-5, atol=1e-6)


def test_scale_transition_of_adam_state():
    g = load_golden('iter_tiny')
    st = S.AdamState(torch.from_numpy(g['image_3']))
    st.exp_avg = torch.from_numpy(g['exp_avg_3'])
    st.exp_avg_sq = torch.from_numpy(g['exp_avg_sq_3'])
    st.step = int(g['step_3'])
    nxt = st.rescaled((57, 68))
    assert nxt.step == int(g['next_step'])
    assert torch.allclose(nxt.exp_avg, torch.from_numpy(g['next_exp_avg']), rtol=1e-5, atol=1e-7)
    assert torch.allclose(nxt.exp_avg_sq, torch.from_numpy(g['next_exp_avg_sq']), rtol=1e-5, atol=1e-9)
    assert float(nxt.exp_avg_sq.min()) >= 0


def test_vgg_tables_and_weight_loader():
    assert len(vgg.CONV_INDICES) == 13 and vgg.CONV_INDICES[0] == 0 and vgg.CONV_INDICES[-1] == 28
    assert vgg.min_size_for([1, 6, 11, 20, 22, 29]) == 16 and vgg.min_size_for([22]) == 8
    params = vgg.synthetic_vgg19_weights(0)
    sd = {}
    for idx, (w, b) in zip(vgg.CONV_INDICES, params):
        sd[f'features.{idx}.weight'], sd[f'features.{idx}.bias'] = w, b
    back = vgg.weights_from_state_dict(sd)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(params, back))


def test_image_conversions_round_trip():
    arr = (np.arange(5 * 7 * 3) % 256).astype(np.uint8).reshape(5, 7, 3)
    from PIL import Image
    t = S.to_tensor(Image.fromarray(arr, 'RGB'))
    assert t.shape == (3, 5, 7) and float(t.max()) <= 1.0
    back = np.asarray(S.to_pil_image(t))
    assert np.array_equal(back, arr)


def test_strip_lbfgs_is_torch_lbfgs_when_there_is_one_rank():
    """sharding.StripLBFGS restates torch.optim.LBFGS.step for the reference's configuration (max_iter=1, history_size=10,
    no line search; reference style_transfer.py:465).  With one rank its global sums are the local ones, so the iterates
    must be torch's bit for bit - on a non-quadratic objective, through history saturation (25 > 10 iterations)."""
    import torch
    from style_transfer import sharding

    def objective(x):
        return ((x * x).sum() + 0.1 * (x[1:] * x[:-1]).sum() + torch.log1p(x.pow(4)).sum()) / x.numel()

    g = torch.Generator().manual_seed(0)
    x0 = torch.randn(300, generator=g)
    a = x0.clone().requires_grad_()
    opt = torch.optim.LBFGS([a], max_iter=1, history_size=10)

    def closure_a():
        opt.zero_grad()
        loss = objective(a)
        loss.backward()
        return loss

    b = x0.clone()
    grad = torch.empty_like(b)
    mine = sharding.StripLBFGS(b, grad, lambda t: None, lambda t: None, history_size=10)

    def closure_b():
        with torch.enable_grad():
            xb = b.detach().clone().requires_grad_()
            loss = objective(xb)
            loss.backward()
        grad.copy_(xb.grad)
        return loss.detach()

    for i in range(25):
        la, lb = opt.step(closure_a), mine.step(closure_b)
        assert float(la) == float(lb), (i, float(la), float(lb))
        assert torch.equal(a.detach(), b), i
