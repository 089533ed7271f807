#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  The reference imports torchvision and
tifffile at module level; neither is installed, so a minimal shim (tests/golden/_refshim) is put
on sys.path.  The pretrained VGG-19 file is not available (no network), so the reference model's
conv parameters are overwritten with the package's seeded synthetic weights.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

What is recorded (all fp32, CPU, torch.set_num_threads(8), deterministic):
  ns_kat       Newton-Schulz sqrtm forward + Lyapunov backward on a seeded SPD matrix (sqrtm.py)
  eval_*       one closure evaluation: 7 weighted loss terms, total, image gradient, tap statistics; plus
               `terms64` / `total64`: the SAME reference modules evaluated in float64 on the same fp32
               parameters and inputs - |terms - terms64| is the reference's own fp32 rounding floor, which the
               GPU tests use as max(1e-4, 3 floor) for the style terms (non-converged NS-12 chain)
  eval_512, eval_1024, eval_2048, eval_2896x2172
               the same at BASELINE.json's config sizes (512^2: configs[1], 1024^2: configs[2], 2048^2: configs[3],
               2896 x 2172 (W x H): configs[4]); the inputs are regenerated from seeds by tests/synth.py
               (platform-stable) instead of being stored; above 1024^2 `terms64` comes from a no-grad float64 pass
  iter_tiny    3 full hot-loop iterations (Adam + clamp + EMA) and the scale transition after them
  stylize_e2e  StyleTransfer.stylize() end to end on PIL inputs (2 scales), loss trace + result
  stylize_lbfgs, stylize_init_{gray,uniform,normal,style_stats}
               stylize() with optimizer='lbfgs' (2 scales) and with each random `init` mode under
               torch.manual_seed(0) (two style images, weights .7/.3): loss trace + result

    python tests/golden/make_golden.py [case ...]   # only the named cases (e.g. eval_1024 stylize_lbfgs)
"""

import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, '_refshim'))
sys.path.insert(1, '/root/reference')

import style_transfer as ref_pkg                         # noqa: E402  (the reference)
from style_transfer import style_transfer as ref        # noqa: E402
from style_transfer import sqrtm as ref_sqrtm           # noqa: E402

_spec = importlib.util.spec_from_file_location(
    'st_vgg', os.path.join(REPO, 'style-transfer-pytorch_amd', 'style_transfer', 'vgg.py'))
st_vgg = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(st_vgg)

sys.path.insert(0, os.path.dirname(HERE))
import synth                                             # noqa: E402  (tests/synth.py)

torch.set_num_threads(8)


def smooth_image(seed, h, w):
    """Seeded synthetic photo-like field in [0,1]: low-res uniform noise upsampled + fine noise."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((1, 3, max(h // 16, 2), max(w // 16, 2)), generator=g)
    img = torch.nn.functional.interpolate(low, (h, w), mode='bicubic', align_corners=False)
    img = img + (torch.rand((1, 3, h, w), generator=g) - 0.5) * (24 / 255)
    return img.clamp(0, 1).contiguous()


def make_reference(pooling='max', seed=0, dtype=torch.float32):
    st = ref.StyleTransfer(devices=['cpu'], pooling=pooling)
    params = st_vgg.synthetic_vgg19_weights(seed)
    with torch.no_grad():
        for idx, (w, b) in zip(st_vgg.CONV_INDICES, params):
            st.model.model[idx].weight.copy_(w)
            st.model.model[idx].bias.copy_(b)
    if dtype != torch.float32:
        st.model.model.to(dtype)       # the fp32 parameters, exactly representable: only the arithmetic changes
    return st, params


def evaluate64(pooling, content, styles, style_w, image):
    """The reference modules in float64 on the same fp32 parameters / inputs: the exact-arithmetic value the
    fp32 run approximates (its distance from `terms` is the reference's own rounding floor)."""
    st64, _ = make_reference(pooling, dtype=torch.float64)
    crit = build_crit(st64, content.double(), [s.double() for s in styles], style_w)
    terms, total, _, _ = evaluate(st64, crit, image.double())
    return np.array(terms, dtype=np.float64), np.float64(total)


def build_crit(st, content, styles, style_image_weights, content_weight=0.015, tv_weight=2.0):
    """The per-scale target construction of stylize (style_transfer.py:425-455) on tensors."""
    with torch.no_grad():
        content_feats = st.model(content, layers=st.content_layers)
        content_losses = []
        for layer in st.content_layers:
            target = content_feats[layer]
            content_losses.append(ref.Scale(ref.LayerApply(ref.ContentLossMSE(target), layer),
                                            content_weight / len(st.content_layers)))
        style_targets, style_losses = {}, []
        for i, style in enumerate(styles):
            style_feats = st.model(style, layers=st.style_layers)
            for layer in st.style_layers:
                tm, tc = ref.StyleLossW2.get_target(style_feats[layer])
                tm *= style_image_weights[i]
                tc *= style_image_weights[i]
                if layer not in style_targets:
                    style_targets[layer] = tm, tc
                else:
                    style_targets[layer][0].add_(tm)
                    style_targets[layer][1].add_(tc)
        for layer, weight in zip(st.style_layers, st.style_weights):
            style_losses.append(ref.Scale(ref.LayerApply(ref.StyleLossW2(style_targets[layer]), layer),
                                          weight))
        tv = ref.Scale(ref.LayerApply(ref.TVLoss(), 'input'), tv_weight)
    return ref.SumLoss([*content_losses, *style_losses, tv])


def evaluate(st, crit, image):
    image = image.detach().clone().requires_grad_(True)
    feats = st.model(image)
    terms = [loss(feats) for loss in crit]
    total = sum(terms)
    total.backward()
    taps = {}
    for k, v in feats.items():
        if k == 'input':
            continue
        v = v.detach()
        taps[f'tap{k}_shape'] = np.array(v.shape)
        taps[f'tap{k}_mean'] = np.float64(v.double().mean())
        taps[f'tap{k}_absmean'] = np.float64(v.double().abs().mean())
        taps[f'tap{k}_head'] = v.flatten()[:512].numpy().copy()
    return [float(t) for t in terms], float(total), image.grad.detach().clone(), taps


def case_eval(name, pooling, h, w, style_shapes, style_w, seed, full_grad):
    st, _ = make_reference(pooling)
    content = smooth_image(seed, h, w)
    styles = [smooth_image(seed + 10 + i, sh, sw) for i, (sh, sw) in enumerate(style_shapes)]
    image = smooth_image(seed + 100, h, w)
    crit = build_crit(st, content, styles, style_w)
    terms, total, grad, taps = evaluate(st, crit, image)
    terms64, total64 = evaluate64(pooling, content, styles, style_w, image)
    out = dict(content=content.numpy(), image=image.numpy(),
               style_weights=np.array(style_w, dtype=np.float64),
               terms=np.array(terms, dtype=np.float64), total=np.float64(total),
               terms64=terms64, total64=total64,
               grad_l2=np.float64(grad.double().norm()), grad_absmax=np.float64(grad.abs().max()),
               pooling=np.array(pooling), **taps)
    for i, s in enumerate(styles):
        out[f'style{i}'] = s.numpy()
    if full_grad:
        out['grad'] = grad.numpy()
    else:
        out['grad_sub'] = grad.flatten()[::7].numpy().copy()
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    print(f'{name}: total={total:.8g} terms={["%.6g" % t for t in terms]} |g|={float(grad.norm()):.6g}')


def evaluate64_terms(content, style, image):
    """float64 VALUES only (no autograd graph: a float64 backward of a 2896x2172 image does not fit this container's
    62 GB): the reference modules under torch.no_grad() on the same fp32 parameters / inputs."""
    st64, _ = make_reference('max', dtype=torch.float64)
    crit = build_crit(st64, content.double(), [style.double()], [1.0])
    with torch.no_grad():
        feats = st64.model(image.double())
        terms = [float(loss(feats)) for loss in crit]
    return np.array(terms, dtype=np.float64), np.float64(sum(terms))


def case_eval_large(name, size, seed, grad_stride=61):
    """BASELINE config sizes: images from tests/synth.py seeds (not stored), one style image of the same size.
    `size`: S (square) or (height, width)."""
    h, w = (size, size) if isinstance(size, int) else size
    st, _ = make_reference('max')
    content = synth.smooth_image(seed, h, w)
    style = synth.smooth_image(seed + 1, h, w)
    image = synth.smooth_image(seed + 2, h, w)
    crit = build_crit(st, content, [style], [1.0])
    terms, total, grad, taps = evaluate(st, crit, image)
    del st, crit
    if h * w > 1024 * 1024:
        terms64, total64 = evaluate64_terms(content, style, image)
    else:
        terms64, total64 = evaluate64('max', content, [style], [1.0], image)
    out = dict(size=np.int64(h if h == w else 0), height=np.int64(h), width=np.int64(w), seed=np.int64(seed),
               grad_stride=np.int64(grad_stride),
               content_checksum=synth.checksum(content), style_checksum=synth.checksum(style),
               image_checksum=synth.checksum(image),
               terms=np.array(terms, dtype=np.float64), total=np.float64(total), terms64=terms64, total64=total64,
               grad_l2=np.float64(grad.double().norm()), grad_absmax=np.float64(grad.abs().max()),
               grad_sub=grad.flatten()[::grad_stride].numpy().copy(), pooling=np.array('max'))
    for k, v in taps.items():
        if k.endswith('_mean') or k.endswith('_absmean') or k.endswith('_shape'):
            out[k] = v
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
    floors = np.abs(np.array(terms) - terms64) / np.abs(terms64)
    print(f'{name}: total={total:.8g} terms={["%.6g" % t for t in terms]} |g|={float(grad.norm()):.6g} '
          f'fp32-vs-fp64 floors={["%.1e" % f for f in floors]}', flush=True)


def _pil(t):
    from PIL import Image
    return Image.fromarray((t[0].permute(1, 2, 0) * 255).round().byte().numpy(), 'RGB')


def case_stylize_variant(name, spread=False, **kw):
    """stylize() with a non-default optimiser / init (style_transfer.py:380-406,464-467,482-483): 64x64 content,
    two style images (56x72 and 64x48) with weights .7/.3, torch.manual_seed(0) as the CLI sets it.

    spread=True (L-BFGS): the quasi-Newton recursion amplifies rounding-level differences of the gradient, so the
    reference's own trace is only reproducible to ~1e-3 after three iterations and ~2e-2 after seven.  The fixture
    records that: the same run with 1 instead of 8 threads (another summation order) and with conv1_1's bias
    scaled by 1 + 1e-6 / 1 + 1e-5, as `trace_spread` / `result_spread` (max relative / mean absolute deviation
    from the base run).  The GPU test allows 5x that spread."""
    content = _pil(smooth_image(5, 64, 64))
    styles = [_pil(smooth_image(15, 56, 72)), _pil(smooth_image(16, 64, 48))]

    def run(threads=8, perturb=0.0):
        torch.set_num_threads(threads)
        st, _ = make_reference('max')
        if perturb:
            with torch.no_grad():
                st.model.model[0].bias.mul_(1 + perturb)
        its = []
        torch.manual_seed(0)
        st.stylize(content, styles, style_weights=[0.7, 0.3],
                   callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)), **kw)
        torch.set_num_threads(8)
        return np.array(its, dtype=np.float64), st.get_image_tensor().numpy().copy()

    its, result = run()
    extra = {}
    if spread:
        ts, rs, fs = np.zeros(len(its)), 0.0, 0.0
        for alt in (run(threads=1), run(perturb=1e-6), run(perturb=1e-5)):
            ts = np.maximum(ts, np.abs(alt[0][:, 4] - its[:, 4]) / np.abs(its[:, 4]))
            rs = max(rs, float(np.abs(alt[1] - result).mean()))
            fs = max(fs, float((np.abs(alt[1] - result) > 1e-3).mean()))
        extra = dict(trace_spread=ts, result_spread=np.float64(rs), result_outlier_frac=np.float64(fs))
        print(f'{name}: reference self-spread of the trace {["%.1e" % t for t in ts]}, of the result (mean abs) {rs:.2e}, '
              f'{100 * fs:.3f}% of values move by > 1e-3')
    np.savez_compressed(os.path.join(HERE, f'{name}.npz'), content_u8=np.asarray(content),
                        style0_u8=np.asarray(styles[0]), style1_u8=np.asarray(styles[1]),
                        iterates=its, result=result, **extra)
    print(f'{name}:', [round(i[4], 6) for i in its])


def case_stylize_c1():
    """BASELINE.json configs[0]: 256x256 content and style, a single scale, 50 Adam iterations
    on the reference's CPU path - the per-iteration loss trace and the averaged result.  Inputs are regenerated from
    seeds by the test (tests/synth.py is bit-stable; checksums stored), the result is stored on a 4x4 sub-grid.
    The trace's reproducibility is recorded like for L-BFGS: the same run with 1 thread and with conv1_1's bias
    scaled by 1 + 1e-6 (`trace_spread`, `result_spread`)."""
    content_t, style_t = synth.smooth_image(21, 256, 256), synth.smooth_image(22, 256, 256)
    content, style = _pil(content_t), _pil(style_t)

    def run(threads=8, perturb=0.0):
        torch.set_num_threads(threads)
        st, _ = make_reference('max')
        if perturb:
            with torch.no_grad():
                st.model.model[0].bias.mul_(1 + perturb)
        its = []
        torch.manual_seed(0)
        st.stylize(content, [style], min_scale=256, end_scale=256, initial_iterations=50,
                   callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)))
        torch.set_num_threads(8)
        return np.array(its, dtype=np.float64), st.get_image_tensor().numpy().copy()

    its, result = run()
    ts, rs = np.zeros(len(its)), 0.0
    for alt in (run(threads=1), run(perturb=1e-6)):
        ts = np.maximum(ts, np.abs(alt[0][:, 4] - its[:, 4]) / np.abs(its[:, 4]))
        rs = max(rs, float(np.abs(alt[1] - result).mean()))
    print(f'stylize_c1: reference self-spread of the trace max {ts.max():.1e} (last {ts[-1]:.1e}), result {rs:.2e}')
    np.savez_compressed(os.path.join(HERE, 'stylize_c1.npz'), seeds=np.array([21, 22]),
                        content_checksum=synth.checksum(content_t), style_checksum=synth.checksum(style_t),
                        iterates=its, result_sub=result[:, ::4, ::4].copy(), result_mean=np.float64(result.mean()),
                        trace_spread=ts, result_spread=np.float64(rs))
    print('stylize_c1:', [round(i[4], 6) for i in its[::7]])


def case_stylize_c2():
    """BASELINE.json configs[1] as the CLI runs it: 512x512 content and style, every default -
    scales 128, 181, 256, 362, 512 with 1000 + 4 x 500 Adam iterations - on the reference's CPU path (~15 minutes on
    8 threads).  Stored: the loss of every 25th iteration, the averaged result on an 8x8 sub-grid and its mean, and
    the same for a second run with conv1_1's bias scaled by 1 + 1e-6 (3000 chaotic Adam steps: the two runs agree in
    the loss curve, not in the pixels).  Inputs are regenerated from seeds (tests/synth.py)."""
    content_t, style_t = synth.smooth_image(31, 512, 512), synth.smooth_image(32, 512, 512)
    content, style = _pil(content_t), _pil(style_t)

    def run(perturb=0.0):
        st, _ = make_reference('max')
        if perturb:
            with torch.no_grad():
                st.model.model[0].bias.mul_(1 + perturb)
        its = []
        torch.manual_seed(0)
        st.stylize(content, [style], callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)))
        return np.array(its, dtype=np.float64), st.get_image_tensor().numpy().copy()

    its, result = run()
    its2, result2 = run(perturb=1e-6)
    spread = np.abs(its2[:, 4] - its[:, 4]) / np.abs(its[:, 4])
    print(f'stylize_c2: {len(its)} iterations, final loss {its[-1, 4]:.6f} / {its2[-1, 4]:.6f}, max rel spread of the loss '
          f'{spread.max():.2e} (last {spread[-1]:.2e}), result mean-abs spread {np.abs(result2 - result).mean():.2e}')
    np.savez_compressed(os.path.join(HERE, 'stylize_c2.npz'), seeds=np.array([31, 32]),
                        content_checksum=synth.checksum(content_t), style_checksum=synth.checksum(style_t),
                        iterates=its[::25].copy(), last=its[-1].copy(), trace_spread=spread[::25].copy(),
                        max_spread=np.float64(spread.max()), result_sub=result[:, ::8, ::8].copy(),
                        result_mean=np.float64(result.mean()), result_std=np.float64(result.std()),
                        result_spread=np.float64(np.abs(result2 - result).mean()),
                        result2_mean=np.float64(result2.mean()))


def case_ns():
    g = torch.Generator().manual_seed(7)
    n = 64
    b = torch.randn((n, 2 * n), generator=g)
    a = (b @ b.t()) / (2 * n) + torch.eye(n) * 1e-3
    a = a.requires_grad_(True)
    root = ref_sqrtm.sqrtm_ns_lyap(a, num_iters=12)
    gout = torch.randn((n, n), generator=g)
    (root * gout).sum().backward()
    np.savez_compressed(os.path.join(HERE, 'ns_kat.npz'), a=a.detach().numpy(),
                        root=root.detach().numpy(), gout=gout.numpy(), ga=a.grad.numpy())
    print('ns_kat: |root@root - a| =', float((root @ root - a).detach().abs().max()))


def case_iter_tiny():
    h, w = 40, 48
    st, _ = make_reference('max')
    content = smooth_image(3, h, w)
    styles = [smooth_image(13, 36, 44)]
    crit = build_crit(st, content, styles, [1.0])
    # stylize lines 420-422, 457-458, 472-486 on tensors
    image = smooth_image(103, h, w).clamp(0, 1)
    image0 = image.clone()
    average = ref.EMA(image, 0.99)
    image.requires_grad_()
    opt = torch.optim.Adam([image], lr=0.02, betas=(0.9, 0.99))

    def closure():
        feats = st.model(image)
        loss = crit(feats)
        loss.backward()
        return loss

    trace, snaps = [], {}
    for i in range(1, 4):
        opt.zero_grad()
        loss = opt.step(closure)
        with torch.no_grad():
            image.clamp_(0, 1)
        average.update(image)
        trace.append(float(loss))
        if i == 1:
            s = opt.state_dict()['state'][0]
            snaps.update(image_1=image.detach().numpy().copy(), exp_avg_1=s['exp_avg'].numpy().copy(),
                         exp_avg_sq_1=s['exp_avg_sq'].numpy().copy(),
                         ema_value_1=average.value.detach().numpy().copy(),
                         ema_accum_1=np.float32(average.accum.detach()))
    s = opt.state_dict()['state'][0]
    snaps.update(image_3=image.detach().numpy().copy(), exp_avg_3=s['exp_avg'].numpy().copy(),
                 exp_avg_sq_3=s['exp_avg_sq'].numpy().copy(), ema_value_3=average.value.detach().numpy().copy(),
                 ema_accum_3=np.float32(average.accum.detach()), step_3=np.float64(float(s['step'])),
                 average_3=average.get().detach().numpy().copy())
    # scale transition (:496-497, :420-421, :460-462) to the next sqrt(2) scale
    with torch.no_grad():
        image.copy_(average.get())
    nh, nw = 57, 68
    image_n = ref.interpolate(image.detach(), (nh, nw), mode='bicubic').clamp(0, 1)
    average_n = ref.EMA(image_n, 0.99)
    state_n = ref.scale_adam(opt.state_dict(), (nh, nw))['state'][0]
    snaps.update(next_image=image_n.numpy().copy(), next_exp_avg=state_n['exp_avg'].numpy().copy(),
                 next_exp_avg_sq=state_n['exp_avg_sq'].numpy().copy(),
                 next_ema_value=average_n.value.detach().numpy().copy(),
                 next_ema_accum=np.float32(average_n.accum.detach()), next_step=np.float64(float(state_n['step'])))
    np.savez_compressed(os.path.join(HERE, 'iter_tiny.npz'), content=content.numpy(),
                        style0=styles[0].numpy(), image0=image0.numpy(),
                        trace=np.array(trace, dtype=np.float64), **snaps)
    print('iter_tiny: trace', trace)


def case_stylize_e2e():
    """Full reference stylize() on PIL inputs: 2 scales (45, 64), 4 + 3 iterations."""
    from PIL import Image
    st, _ = make_reference('max')
    cimg = (smooth_image(5, 64, 64)[0].permute(1, 2, 0) * 255).round().byte().numpy()
    simg = (smooth_image(15, 56, 72)[0].permute(1, 2, 0) * 255).round().byte().numpy()
    content, style = Image.fromarray(cimg, 'RGB'), Image.fromarray(simg, 'RGB')
    its = []
    torch.manual_seed(0)
    st.stylize(content, [style], min_scale=45, end_scale=64, iterations=3, initial_iterations=4,
               callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)))
    np.savez_compressed(os.path.join(HERE, 'stylize_e2e.npz'), content_u8=cimg, style_u8=simg,
                        iterates=np.array(its, dtype=np.float64),
                        result=st.get_image_tensor().numpy().copy())
    print('stylize_e2e:', its)


def case_fingerprint():
    params = st_vgg.synthetic_vgg19_weights(0)
    np.savez_compressed(os.path.join(HERE, 'weights_fingerprint.npz'),
                        fp=np.array(st_vgg.weights_fingerprint(params), dtype=np.float64))


def block_moments(grad, block):
    """Per (channel, block x block tile): sum, sum of squares and sum of |g| of the gradient, accumulated in float64 - a
    fixture in which EVERY element of a full-size gradient takes part (the strided samples of eval_* see one element in
    61 ... 499: a wrong row at a tile or strip seam can hide between them).  Ragged edges: zero padding."""
    g = grad.detach().double()[0]
    c, h, w = g.shape
    hb, wb = -(-h // block), -(-w // block)
    pad = torch.zeros(c, hb * block, wb * block, dtype=torch.float64)
    pad[:, :h, :w] = g
    t = pad.reshape(c, hb, block, wb, block)
    return t.sum((2, 4)), (t * t).sum((2, 4)), t.abs().sum((2, 4))


def case_grad_blocks(name, size, seed, block=32):
    """<name>_blocks.npz beside eval_<size>: the same inputs (tests/synth.py seeds), the reference's gradient reduced to
    block moments.  grad_l2 is stored again so the consumer can check that this run reproduced the eval_* one."""
    h, w = (size, size) if isinstance(size, int) else size
    st, _ = make_reference('max')
    content = synth.smooth_image(seed, h, w)
    style = synth.smooth_image(seed + 1, h, w)
    image = synth.smooth_image(seed + 2, h, w)
    crit = build_crit(st, content, [style], [1.0])
    terms, total, grad, _ = evaluate(st, crit, image)
    s1, s2, sa = block_moments(grad, block)
    np.savez_compressed(os.path.join(HERE, f'{name}_blocks.npz'), block=np.int64(block), height=np.int64(h), width=np.int64(w),
                        seed=np.int64(seed), total=np.float64(total), grad_l2=np.float64(grad.double().norm()),
                        sums=s1.numpy(), squares=s2.numpy(), abs_sums=sa.numpy())
    print(f'{name}_blocks: {tuple(s1.shape)} blocks of {block}x{block}, total={total:.8g} |g|={float(grad.norm()):.6g}')


def case_grad_blocks64(name, size, seed, block=32):
    """<name>_blocks64.npz (round 5): the block moments of the reference's gradient evaluated in FLOAT64 on
    the same fp32 parameters and inputs, and the distance of the reference's own fp32 gradient from it - the rounding floor of
    the image gradient.  512^2 and 1024^2 only (a float64 backward of the larger images does not fit a CPU container)."""
    st, _ = make_reference('max')
    content, style, image = (synth.smooth_image(seed + i, size, size) for i in range(3))
    crit = build_crit(st, content, [style], [1.0])
    _, total, grad, _ = evaluate(st, crit, image)
    st64, _ = make_reference('max', dtype=torch.float64)
    crit64 = build_crit(st64, content.double(), [style.double()], [1.0])
    _, total64, grad64, _ = evaluate(st64, crit64, image.double())
    s1, s2, sa = block_moments(grad64, block)
    _, r2, _ = block_moments(grad, block)
    scale = float(np.sqrt(s2.sum()))
    l2 = (r2.sqrt() - s2.sqrt()).abs() / (s2.sqrt() + 1e-3 * scale / np.sqrt(s2.numel()))
    rel = float((grad.double() - grad64).norm() / grad64.norm())
    np.savez_compressed(os.path.join(HERE, f'{name}_blocks64.npz'), block=np.int64(block), height=np.int64(size), width=np.int64(size),
                        seed=np.int64(seed), total64=np.float64(total64), grad_l2=np.float64(grad64.norm()),
                        sums=s1.numpy(), squares=s2.numpy(), abs_sums=sa.numpy(),
                        ref32_rel_l2=np.float64(rel), ref32_worst_block_l2=np.float64(l2.max()))
    print(f'{name}_blocks64: reference fp32 gradient vs its float64 evaluation: rel-L2 {rel:.3e}, worst {block}x{block} block L2 '
          f'{float(l2.max()):.3e}')


def grad64_banded(content, style, image, band=512, halo=128):
    """The reference's closure gradient in FLOAT64 for images whose float64 autograd graph does not fit this container (round 6,
    VERDICT r5 next #4): exact, not an approximation.
      1. taps of the image under no_grad (float64); the loss modules alone under autograd give dL / d tap for the six taps and
         dTV / d image (the Gram statistics are global, so this step sees the whole image);
      2. the image gradient is J^T of those tap gradients and J is local: for every band of `band` image rows the reference
         model runs under autograd on the rows [band - halo, band + halo), the surrogate sum(tap_local * dL/dtap) over the tap
         rows that belong to the band is back-propagated, the pixel gradients of all bands add up.  halo >= 85 px = the
         receptive radius of relu5_1 (band edges are multiples of 16, so pooling windows and tap rows line up); the replicate /
         zero padding a crop puts at its cut edges reaches only tap rows outside the band.
    Checked against the un-banded float64 gradient at 512^2 (case 'check_banded64')."""
    st64, _ = make_reference('max', dtype=torch.float64)
    for prm in st64.model.parameters():
        prm.requires_grad_(False)
    crit = build_crit(st64, content.double(), [style.double()], [1.0])
    img = image.double()
    h, w = img.shape[-2:]
    with torch.no_grad():
        feats = st64.model(img)
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in feats.items()}
    del feats
    terms = [loss(leaves) for loss in crit]
    total = sum(terms)
    total.backward()
    gtap = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    terms = [float(t) for t in terms]
    total = float(total)
    del leaves
    grad = gtap.pop('input').clone()                       # dTV / d image
    for r0 in range(0, h, band):
        r1 = min(h, r0 + band)
        a, b = max(0, r0 - halo), min(h, r1 + halo)
        x = img[:, :, a:b, :].clone().requires_grad_(True)
        local = st64.model(x)
        surrogate = 0.0
        for k, g in gtap.items():
            s = round(h / g.shape[-2]) if g.shape[-2] else 1
            s = 1 << (s.bit_length() - 1) if s & (s - 1) else s          # stride of the tap: 1, 2, 4, 8, 16
            g0, g1 = r0 // s, min(-(-r1 // s) if r1 == h else r1 // s, g.shape[-2])
            l0, l1 = g0 - a // s, g1 - a // s
            if g1 > g0:
                surrogate = surrogate + (local[k][:, :, l0:l1, :] * g[:, :, g0:g1, :]).sum()
        surrogate.backward()
        grad[:, :, a:b, :] += x.grad
        del local, surrogate, x
        print(f'  band rows {r0}..{r1} of {h} done', flush=True)
    return terms, total, grad


def case_check_banded64():
    """grad64_banded against the plain float64 autograd gradient at 512^2 (bands of 128 rows: four seams inside the image)."""
    content, style, image = (synth.smooth_image(40 + i, 512, 512) for i in range(3))
    st64, _ = make_reference('max', dtype=torch.float64)
    crit64 = build_crit(st64, content.double(), [style.double()], [1.0])
    _, total64, grad64, _ = evaluate(st64, crit64, image.double())
    _, total_b, grad_b = grad64_banded(content, style, image, band=128, halo=96)
    rel = float((grad_b - grad64).norm() / grad64.norm())
    print(f'check_banded64: total {total64:.15g} vs {total_b:.15g}; gradient rel-L2 {rel:.3e} max-abs {float((grad_b - grad64).abs().max()):.3e}')
    assert rel < 1e-12


def case_grad_blocks64_banded(name, size, seed, block=32, band=512, stride=331):
    """<name>_blocks64.npz at the sizes whose float64 backward does not fit in one piece: 2048^2 and 2896 x 2172.  Besides the
    block moments: every `stride`-th element of the float64 gradient (the sample positions of eval_<size>.npz's grad_sub) and
    the reference's own fp32 gradient's rel-L2 from it ON THAT SAMPLE - the worst block is a maximum over thousands of blocks
    and moves with the image; the sample measures the whole gradient."""
    h, w = (size, size) if isinstance(size, int) else size
    st, _ = make_reference('max')
    content, style, image = (synth.smooth_image(seed + i, h, w) for i in range(3))
    crit = build_crit(st, content, [style], [1.0])
    _, total, grad, _ = evaluate(st, crit, image)
    del st, crit
    _, total64, grad64 = grad64_banded(content, style, image, band=band)
    s1, s2, sa = block_moments(grad64, block)
    _, r2, _ = block_moments(grad, block)
    scale = float(np.sqrt(s2.sum()))
    l2 = (r2.sqrt() - s2.sqrt()).abs() / (s2.sqrt() + 1e-3 * scale / np.sqrt(s2.numel()))
    rel = float((grad.double() - grad64).norm() / grad64.norm())
    sub64, sub32 = grad64.flatten()[::stride], grad.flatten()[::stride].double()
    rel_sub = float((sub32 - sub64).norm() / sub64.norm())
    np.savez_compressed(os.path.join(HERE, f'{name}_blocks64.npz'), block=np.int64(block), height=np.int64(h), width=np.int64(w),
                        seed=np.int64(seed), total64=np.float64(total64), grad_l2=np.float64(grad64.norm()),
                        sums=s1.numpy(), squares=s2.numpy(), abs_sums=sa.numpy(),
                        ref32_rel_l2=np.float64(rel), ref32_worst_block_l2=np.float64(l2.max()),
                        grad_stride=np.int64(stride), grad64_sub=sub64.numpy().copy(), ref32_sub_rel_l2=np.float64(rel_sub),
                        ref32_block_l2_rms=np.float64(float((l2 * l2).mean().sqrt())))
    print(f'{name}_blocks64: reference fp32 gradient vs its float64 evaluation: rel-L2 {rel:.3e}, worst {block}x{block} block L2 '
          f'{float(l2.max()):.3e}', flush=True)


CASES = {
    'weights_fingerprint': case_fingerprint,
    'ns_kat': case_ns,
    'eval_tiny': lambda: case_eval('eval_tiny', 'max', 40, 48, [(36, 44)], [1.0], seed=3, full_grad=True),
    'eval_avgpool': lambda: case_eval('eval_avgpool', 'average', 40, 48, [(36, 44)], [1.0], seed=3, full_grad=True),
    'eval_l2pool': lambda: case_eval('eval_l2pool', 'l2', 40, 48, [(36, 44)], [1.0], seed=3, full_grad=True),
    'eval_s128': lambda: case_eval('eval_s128', 'max', 128, 128, [(96, 128), (128, 100)], [0.7, 0.3], seed=4,
                                   full_grad=False),
    'eval_odd181': lambda: case_eval('eval_odd181', 'max', 135, 181, [(181, 140)], [1.0], seed=6, full_grad=False),
    'eval_512': lambda: case_eval_large('eval_512', 512, seed=40),
    'eval_1024': lambda: case_eval_large('eval_1024', 1024, seed=50),
    # BASELINE configs[3] / configs[4]: ~2 / ~3 CPU-minutes each incl. the float64 values;
    # the gradient is stored on a coarser sub-grid (every 331st / 499th element: 38 k / 38 k floats)
    'eval_2048': lambda: case_eval_large('eval_2048', 2048, seed=60, grad_stride=331),
    'eval_2896x2172': lambda: case_eval_large('eval_2896x2172', (2172, 2896), seed=70, grad_stride=499),
    # every element of the full-size gradients, as 32 x 32 block moments
    'eval_512_blocks': lambda: case_grad_blocks('eval_512', 512, seed=40),
    'eval_1024_blocks': lambda: case_grad_blocks('eval_1024', 1024, seed=50),
    'eval_2048_blocks': lambda: case_grad_blocks('eval_2048', 2048, seed=60),
    'eval_2896x2172_blocks': lambda: case_grad_blocks('eval_2896x2172', (2172, 2896), seed=70),
    # ... and the same moments of the reference evaluated in float64: the image gradient's rounding floor (round 5)
    'eval_512_blocks64': lambda: case_grad_blocks64('eval_512', 512, seed=40),
    'eval_1024_blocks64': lambda: case_grad_blocks64('eval_1024', 1024, seed=50),
    # ... at the two largest configurations by bands of rows (round 6): ~15 / ~25 CPU-minutes
    'check_banded64': case_check_banded64,
    'eval_2048_blocks64': lambda: case_grad_blocks64_banded('eval_2048', 2048, seed=60, stride=331),
    'eval_2896x2172_blocks64': lambda: case_grad_blocks64_banded('eval_2896x2172', (2172, 2896), seed=70, stride=499),
    'iter_tiny': case_iter_tiny,
    'stylize_e2e': case_stylize_e2e,
    'stylize_c1': case_stylize_c1,
    'stylize_c2': case_stylize_c2,
    'stylize_lbfgs': lambda: case_stylize_variant('stylize_lbfgs', spread=True, optimizer='lbfgs', min_scale=45,
                                                  end_scale=64, iterations=3, initial_iterations=4),
    # non-default loss weights, step size, EMA decay and style scaling (style_transfer.py:366-369,433-437,458)
    'stylize_params': lambda: case_stylize_variant('stylize_params', spread=True, content_weight=0.05, tv_weight=5.0,
                                                   step_size=0.03, avg_decay=0.9, style_scale_fac=1.5, min_scale=45,
                                                   end_scale=64, iterations=3, initial_iterations=4),
    'stylize_style_size': lambda: case_stylize_variant('stylize_style_size', spread=True, style_size=40, min_scale=45,
                                                       end_scale=64, iterations=3, initial_iterations=4),
    'stylize_init_gray': lambda: case_stylize_variant('stylize_init_gray', spread=True, init='gray', min_scale=64, end_scale=64,
                                                      initial_iterations=4),
    'stylize_init_uniform': lambda: case_stylize_variant('stylize_init_uniform', spread=True, init='uniform', min_scale=64,
                                                         end_scale=64, initial_iterations=4),
    'stylize_init_normal': lambda: case_stylize_variant('stylize_init_normal', spread=True, init='normal', min_scale=64,
                                                        end_scale=64, initial_iterations=4),
    'stylize_init_style_stats': lambda: case_stylize_variant('stylize_init_style_stats', spread=True, init='style_stats',
                                                             min_scale=45, end_scale=64, iterations=3,
                                                             initial_iterations=4),
}


def main():
    names = sys.argv[1:] or list(CASES)
    for name in names:
        CASES[name]()


if __name__ == '__main__':
    main()
