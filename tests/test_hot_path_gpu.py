"""Hot-path parity on a real MI355X: the closure (7 loss terms + image gradient), the fused
Adam/clamp/EMA iteration and the end-to-end ``stylize()`` against (a) the golden vectors recorded
from the unmodified reference and (b) the CPU oracle run live on the same seeded inputs.

Stated tolerances (fp32; see DESIGN.md "Parity"):
  * content / TV terms and the total: 1e-4 relative to the reference value;
  * a style term: max(1e-4, 3 d_k) where d_k = |fp32 reference - float64 reference| / |float64 reference| is
    the reference's OWN rounding floor for that term on that input (the non-converged NS-12 chain amplifies
    fp32 rounding; d_k is recorded in every golden as terms64, and evaluated live where the oracle runs) -
    in practice 1e-4 everywhere except relu1_1 of two fixtures (floors 9.5e-5 and 1.1e-4);
  * every closure test also prints the north_star's literal contract per term ("[strict-1e-4] ... PASS / FAIL",
    `_strict_report`) and FAILS when a term that is not floor-limited (3 d_k <= 1e-4) misses the strict 1e-4;
  * image gradient: rel-L2 <= 1e-3 (gradients are held to 1e-3, not 1e-4);
  * post-step image / Adam moments / EMA: max-abs 2e-5 on O(1) quantities after one step.
The shipped conv arithmetic (fp16x3) is held to the same numbers as the exact-fp32 mode.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
import st_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TERM_TOL = 1e-4
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


def _term_tols(want_terms, terms64):
    """max(1e-4, 3 x the reference's own fp32-vs-fp64 deviation) for the five style terms, 1e-4 otherwise."""
    tols = []
    for k in range(7):
        floor = abs(want_terms[k] - terms64[k]) / abs(terms64[k])
        tols.append(max(TERM_TOL, 3 * floor) if 1 <= k <= 5 else TERM_TOL)
    return tols


def _check_terms(name, losses, want_terms, want_total, terms64):
    got = losses.cpu().double().numpy()
    tols = _term_tols(want_terms, terms64)
    for k in range(7):
        rel = abs(got[k] - want_terms[k]) / abs(want_terms[k])
        print(f'[parity] {name} term[{O.TERM_NAMES[k]}]: got {got[k]:.8g} want {want_terms[k]:.8g} rel={rel:.2e} '
              f'(tol {tols[k]:.1e})')
    rel_total = abs(got[7] - want_total) / abs(want_total)
    print(f'[parity] {name} total: got {got[7]:.8g} want {want_total:.8g} rel={rel_total:.2e}')
    for k in range(7):
        rel = abs(got[k] - want_terms[k]) / abs(want_terms[k])
        assert rel <= tols[k], f'{name}: term {O.TERM_NAMES[k]} rel {rel:.2e} > {tols[k]:.1e}'
    assert rel_total <= TOTAL_TOL, f'{name}: total rel {rel_total:.2e}'
    _strict_report(name, losses, want_terms, terms64)


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
    _check_terms(name, losses, g['terms'], float(g['total']), g['terms64'])
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


def _live_oracle_case(size, kind, precision, vgg_weights):
    from style_transfer import _hip as hip
    sh = size * 200 // 256                                  # a style image of another size (never upscaled)
    if kind == 'photo_like':
        content, style, image = _smooth(21, size, size), _smooth(22, sh, size), _smooth(23, size, size)
    else:
        gen = torch.Generator().manual_seed(21)
        content = torch.rand((1, 3, size, size), generator=gen)
        style = torch.rand((1, 3, sh, size), generator=gen)
        image = torch.rand((1, 3, size, size), generator=gen)
    torch.set_num_threads(min(16, torch.get_num_threads()))   # 512^3 GEMMs oversubscribe on a 128-thread host
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
        print(f'[parity] live{size}/{kind} {O.TERM_NAMES[k]}: hip-vs-cpu32 {rel:.2e}  hip-vs-fp64 {rel64:.2e}  '
              f'cpu32-vs-fp64 {floor:.2e}')
        assert rel <= (max(1e-4, 3 * floor) if 1 <= k <= 5 else 1e-4), (kind, O.TERM_NAMES[k], rel, floor)
    _strict_report(f'live{size}/{kind}', losses, terms, terms64)
    rel_total = abs(got[7] - total) / abs(total)
    print(f'[parity] live{size}/{kind} total rel={rel_total:.2e}')
    assert rel_total <= TOTAL_TOL
    err, floor_g = rel_l2(g.cpu(), grad), rel_l2(grad, grad64)
    print(f'[parity] live{size}/{kind} image gradient rel_l2={err:.3e} (cpu32-vs-fp64 {floor_g:.3e})')
    assert err <= (GRAD_TOL_BF16X3 if precision == 'bf16x3' else GRAD_TOL)


@pytest.mark.parametrize('kind,precision', [('photo_like', 'fp32'), ('white_noise', 'fp32'),
                                            ('photo_like', 'bf16x6'), ('white_noise', 'bf16x6'),
                                            ('photo_like', 'fp16x3'), ('white_noise', 'fp16x3'),
                                            ('photo_like', 'bf16x3')])
def test_closure_against_live_oracle_256(kind, precision, vgg_weights):
    """BASELINE config 1 size (256x256): oracle evaluated here on the host, HIP path on the GPU.

    The style terms go through the non-converged NS-12 recurrence, whose fp32 evaluation has a
    rounding floor: the reference's own fp32 value deviates from exact arithmetic by d_k (measured
    here with the fp64 run of the same oracle).  A style term passes within max(1e-4, 3 d_k) of the fp32
    reference; content, TV and the total must be within 1e-4 regardless.  'white_noise' is the stress case
    (every covariance is near-singular); 'photo_like' is the realistic one."""
    _live_oracle_case(256, kind, precision, vgg_weights)


@pytest.mark.parametrize('kind,precision', [('photo_like', 'fp16x3'), ('white_noise', 'fp16x3'), ('photo_like', 'fp32')])
def test_closure_against_live_oracle_512(kind, precision, vgg_weights):
    """BASELINE config 2 size (512x512) in the SHIPPED conv arithmetic: here the trunk runs the producer /
    consumer kernel's XL tile (conv1_2 ... conv2_2), its 256- and 128-pixel tiles (conv3_x ... conv5_1) and the
    fp16x3 Gram / 1x1 head kernels - the configuration bench.py measures."""
    _live_oracle_case(512, kind, precision, vgg_weights)


def _strict_report(name, losses, want_terms, terms64):
    """The north_star's literal contract - every loss term within 1e-4 relative of the reference - term by term, next to
    the floor-based tolerance the assertion uses.  A term counts as floor-limited when the reference's OWN fp32 value
    sits >= 1/3 of 1e-4 away from its float64 value on this input; a term that is NOT floor-limited must pass 1e-4."""
    got = losses.cpu().double().numpy()
    rows = []
    for k in range(7):
        rel = abs(got[k] - want_terms[k]) / abs(want_terms[k])
        floor = abs(want_terms[k] - terms64[k]) / abs(terms64[k])
        rel64 = abs(got[k] - terms64[k]) / abs(terms64[k])
        limited = 3 * floor > TERM_TOL
        rows.append((O.TERM_NAMES[k], rel, floor, rel64, limited))
        print(f'[strict-1e-4] {name} {O.TERM_NAMES[k]}: |hip - ref32| / ref32 = {rel:.2e} -> {"PASS" if rel <= TERM_TOL else "FAIL"}; '
              f'reference fp32-vs-fp64 {floor:.2e}; hip-vs-fp64 {rel64:.2e}'
              f'{" (floor-limited term: the reference itself is >= 3.3e-5 from exact arithmetic)" if limited else ""}')
    for tname, rel, floor, rel64, limited in rows:
        assert limited or rel <= TERM_TOL, f'{name}: {tname} is not floor-limited and misses the strict 1e-4 ({rel:.2e})'
    return rows


def _check_gradient_blocks(name, grad, base):
    """EVERY element of the full-size gradient, through the reference's 32 x 32 block moments (tests/golden/<base>_blocks.npz,
    make_golden.case_grad_blocks): a wrong row or tile anywhere - the seams the strided sample can miss - moves its block's
    sum of squares.  Per block: |sum - ref| <= tol x sum|g|_ref (+ the fp32 noise of a 1024-term sum) and the L2 norms
    within tol of each other; over all blocks the differences' L2 stays within tol of the gradient's norm."""
    b = load_golden(base + '_blocks')
    block = int(b['block'])
    g = grad.double()[0]
    c, h, w = g.shape
    assert (h, w) == (int(b['height']), int(b['width']))
    hb, wb = -(-h // block), -(-w // block)
    pad = torch.zeros(c, hb * block, wb * block, dtype=torch.float64, device=g.device)
    pad[:, :h, :w] = g
    t = pad.reshape(c, hb, block, wb, block)
    s1, s2 = t.sum((2, 4)).cpu().numpy(), (t * t).sum((2, 4)).cpu().numpy()
    r1, r2, ra = b['sums'], b['squares'], b['abs_sums']
    scale = float(np.sqrt(r2.sum()))                       # = |g| of the reference
    sum_err = np.abs(s1 - r1) / (ra + 1e-3 * scale)
    l2_err = np.abs(np.sqrt(s2) - np.sqrt(r2)) / (np.sqrt(r2) + 1e-3 * scale / np.sqrt(r2.size))
    print(f'[parity] {name} gradient block moments ({c} x {hb} x {wb} blocks of {block}^2): worst block sum {sum_err.max():.2e}, '
          f'worst block L2 {l2_err.max():.2e}')
    assert sum_err.max() <= GRAD_TOL, np.unravel_index(sum_err.argmax(), sum_err.shape)
    assert l2_err.max() <= GRAD_TOL, np.unravel_index(l2_err.argmax(), l2_err.shape)
    # Round 5 (the deviation grows with the image): the same measure against the reference evaluated in
    # FLOAT64 (<base>_blocks64: 512^2 / 1024^2 by plain autograd, make_golden.case_grad_blocks64; round 6: 2048^2 and
    # 2896 x 2172 by bands of rows, case_grad_blocks64_banded - exact, checked against the plain form at 512^2), next to the
    # reference's own fp32 gradient's.  The block deviations above ARE that floor, not a kernel's arithmetic
    # (profiles/r05_gradient_attribution.md):
    #   * over the WHOLE gradient (round 6, the banded fixtures carry every 331st / 499th element of the float64 gradient):
    #     HIP-vs-float64 rel-L2 <= 1.5 x reference-fp32-vs-float64 + 5e-5;
    #   * worst 32 x 32 block: <= 1.5 x the reference's worst block + 5e-5 at 512^2 / 1024^2 (measured 1.0 x) and <= 2 x at
    #     the two largest sizes - a maximum over 12 - 18 thousand blocks of two independent rounding-noise fields is not the
    #     same block in both and moves with the image (measured at 2048^2: 6.5e-4 against the reference's 3.8e-4 while the
    #     whole-gradient figures are equal).
    path64 = os.path.join(os.path.dirname(__file__), 'golden', base + '_blocks64.npz')
    if os.path.exists(path64):
        b64 = load_golden(base + '_blocks64')
        q2 = b64['squares']
        scale64 = float(np.sqrt(q2.sum()))
        l2_64 = np.abs(np.sqrt(s2) - np.sqrt(q2)) / (np.sqrt(q2) + 1e-3 * scale64 / np.sqrt(q2.size))
        ref_floor = float(b64['ref32_worst_block_l2'])
        print(f'[parity] {name} gradient blocks vs the reference in FLOAT64: worst block L2 {l2_64.max():.2e}; the reference\'s own '
              f'fp32 gradient: {ref_floor:.2e} (rel-L2 {float(b64["ref32_rel_l2"]):.2e})')
        if 'grad64_sub' in b64:
            stride = int(b64['grad_stride'])
            sub64 = b64['grad64_sub']
            sub = grad.flatten()[::stride].double().cpu().numpy()
            rel_hip = float(np.linalg.norm(sub - sub64) / np.linalg.norm(sub64))
            rel_ref = float(b64['ref32_sub_rel_l2'])
            print(f'[parity] {name} gradient vs the reference in FLOAT64 over every {stride}th element: HIP rel-L2 {rel_hip:.2e}; the '
                  f'reference\'s own fp32 gradient: {rel_ref:.2e}')
            assert rel_hip <= 1.5 * rel_ref + 5e-5
            assert l2_64.max() <= 2.0 * ref_floor + 5e-5
        else:
            assert l2_64.max() <= 1.5 * ref_floor + 5e-5


@pytest.mark.parametrize('name,precision', [('eval_512', 'fp16x3'), ('eval_512', 'fp32'), ('eval_1024', 'fp16x3'),
                                            ('eval_2048', 'fp16x3'), ('eval_2896x2172', 'fp16x3')])
def test_closure_against_reference_goldens_at_baseline_sizes(name, precision, vgg_weights):
    """512^2 (BASELINE configs[1]), 1024^2 (configs[2]), 2048^2 (configs[3]) and 2896 x 2172 (configs[4]) against the
    UNMODIFIED reference (style_transfer.py:472-476): the fixtures hold seeds, the reference's 7 terms / total (fp32 and
    float64) and a strided sample of the gradient; the input images are regenerated by tests/synth.py
    (platform-stable integer hashing; checksums verified here)."""
    import synth
    from style_transfer import _hip as hip
    g = load_golden(name)
    seed, stride = int(g['seed']), int(g['grad_stride'])
    height, width = (int(g['height']), int(g['width'])) if 'height' in g else (int(g['size']), int(g['size']))
    content, style, image = (synth.smooth_image(seed + i, height, width) for i in range(3))
    for t, key in ((content, 'content_checksum'), (style, 'style_checksum'), (image, 'image_checksum')):
        assert np.array_equal(synth.checksum(t), g[key]), f'{key}: synthetic image generator drifted'
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision=precision)
    losses, grad = plan.loss_and_grad(image.to(DEV))
    torch.cuda.synchronize()
    base, name = name, f'{name}/{precision}'
    _check_terms(name, losses, g['terms'], float(g['total']), g['terms64'])
    gc = grad.cpu()
    err = rel_l2(gc.flatten()[::stride], g['grad_sub'])
    nerr = abs(float(gc.double().norm()) - float(g['grad_l2'])) / float(g['grad_l2'])
    print(f'[parity] {name} image gradient (every {stride}th element) rel_l2={err:.3e}; |g| rel={nerr:.2e}')
    assert err <= GRAD_TOL and nerr <= GRAD_TOL
    _check_gradient_blocks(name, grad, base)
    for layer in O.STYLE_LAYERS + O.CONTENT_LAYERS:
        f = plan.feature(layer)
        assert list(f.shape) == list(g[f'tap{layer}_shape'])
        m, am = float(f.double().mean()), float(f.double().abs().mean())
        assert abs(m - float(g[f'tap{layer}_mean'])) <= 1e-5 * float(g[f'tap{layer}_absmean'])
        assert abs(am - float(g[f'tap{layer}_absmean'])) <= 1e-5 * float(g[f'tap{layer}_absmean'])


def _spread_weights(weights, decades=6.0, seed=5):
    """The same network function with per-channel scales spanning `decades` decades inside it: output channel c of every
    conv that is NOT a tap is multiplied by s_c = 10^u (u uniform in +-decades/2, bias too) and input channel c of the
    following conv by 1 / s_c.  ReLU and max pooling commute with positive per-channel factors, so every tap - hence
    every loss term and the image gradient - is unchanged in exact arithmetic, while the feature maps BETWEEN the taps
    carry channels 10^6 apart, as the maps of the real VGG-19 do.  That is the stress case for fp16x3's single
    power-of-two scale per tensor."""
    g = torch.Generator().manual_seed(seed)
    taps = {0, 2, 4, 8, 9, 12}                     # relu1_1, 2_1, 3_1, 4_1, 4_2 (content), 5_1
    out = [(w.clone(), b.clone()) for w, b in weights]
    for i in range(len(out) - 1):
        if i in taps:
            continue
        cout = out[i][0].shape[0]
        sc = torch.pow(10.0, (torch.rand(cout, generator=g) - 0.5) * decades)
        out[i] = (out[i][0] * sc.view(-1, 1, 1, 1), out[i][1] * sc)
        out[i + 1] = (out[i + 1][0] / sc.view(1, -1, 1, 1), out[i + 1][1])
    return out


def test_closure_with_six_decades_of_channel_scales(vgg_weights):
    """Closure-level dynamic-range stress in the SHIPPED arithmetic (fp16x3) at 256^2 against the live oracle: feature
    maps whose channels span six decades (see _spread_weights).  Loss terms under the usual tolerances, the image
    gradient under 1e-3 - and both must stay where the unscaled network puts them.  Plain fp16x3 FAILS this (round 3,
    first run: content term off by 6.5e-3, relu5_1 by 2e-2 - two fp16 planes under one scale per tensor cannot hold a
    channel 2^-20 below its neighbours); the library's range guard moves the affected layers to bf16x6."""
    from style_transfer import _hip as hip
    size = 256
    content, style, image = _smooth(41, size, size), _smooth(42, size, size), _smooth(43, size, size)
    spread = _spread_weights(vgg_weights)
    ratios = [float(w.flatten(1).norm(dim=1).max() / w.flatten(1).norm(dim=1).min()) for w, _ in spread]
    print(f'[parity] channel-scale stress: per-layer max/min output-channel weight norm {["%.1e" % r for r in ratios]}')
    assert max(ratios) >= 1e5
    torch.set_num_threads(min(16, torch.get_num_threads()))
    targets = O.build_targets(content, [style], spread)
    terms, total, grad = O.loss_and_grad(image, spread, targets)
    w64 = [(w.double(), b.double()) for w, b in spread]
    terms64, _, grad64 = O.loss_and_grad(image.double(), w64, O.build_targets(content.double(), [style.double()], w64))
    net, plan = _build_plan(hip, spread, content, [style], [1.0], precision='fp16x3')
    # the range guard (st_api.hip range_guard) must have recognised the compensating weights: forward of the conv AFTER a
    # rescaled layer (its input channels carry 1 / s), data gradient of the rescaled layer itself (its output channels
    # carry s) run bf16x6 - and with normalised weights nothing does
    wide_f, wide_b = net.wide_layers()
    print(f'[parity] channel-scale stress: bf16x6 fallback - forward of convs {[i for i, f in enumerate(wide_f) if f]}, '
          f'data gradient of convs {[i for i, f in enumerate(wide_b) if f]}')
    # the documented rule (include/st_amd.h, st_net_wide_layers): largest weight of a channel > 2^8 x the median channel's
    def over(v):
        return float(v.max() / v.sort().values[len(v) // 2]) > 256.0
    want_f = [int(i > 0 and over(w.abs().amax(dim=(0, 2, 3)))) for i, (w, _) in enumerate(spread)]
    want_b = [int(i > 0 and over(w.abs().amax(dim=(1, 2, 3)))) for i, (w, _) in enumerate(spread)]
    assert (wide_f, wide_b) == (want_f, want_b) and sum(wide_f) >= 5 and sum(wide_b) >= 5
    assert hip.Net(vgg_weights, 'max', DEV, 'fp16x3').wide_layers() == ([0] * 13, [0] * 13)
    losses, g = plan.loss_and_grad(image.to(DEV))
    losses, g = losses.clone(), g.clone()
    _check_terms('spread256/fp16x3', losses, terms, total, terms64)
    err, floor_g = rel_l2(g.cpu(), grad), rel_l2(grad, grad64)
    print(f'[parity] spread256/fp16x3 image gradient rel_l2={err:.3e} (cpu32-vs-fp64 {floor_g:.3e})')
    assert err <= GRAD_TOL
    # the unscaled network computes the same function: the scaled run must not be visibly worse than it
    net0, plan0 = _build_plan(hip, vgg_weights, content, [style], [1.0], precision='fp16x3')
    losses0, g0 = plan0.loss_and_grad(image.to(DEV))
    rel = ((losses - losses0).abs() / losses0.abs()).max().item()
    dg = rel_l2(g.cpu(), g0.cpu())
    print(f'[parity] spread256/fp16x3 vs the unscaled network on the same inputs: loss terms {rel:.2e}, gradient {dg:.2e}')
    assert rel <= 3e-4 and dg <= 2e-3


@pytest.mark.parametrize('size', [64, 256, 512])
def test_pool_argmax_codes_leave_the_closure_unchanged(size, vgg_weights):
    """In the closure the four convolutions that feed a max pool (relu1_2, 2_2, 3_4, 4_4) leave the pooled map and one
    byte per window (first maximum + ReLU mask) instead of their full-resolution output, and the pooling backward
    scatters from those codes (ConvProblem::pool_code, pool_bwd_codes_kernel).  Same arithmetic, less traffic: losses
    and gradient must be bit-identical to ST_POOL_CODES=0 (map written, pooling backward re-reads it) - on a smooth
    image and on one with many ties (five grey levels: the first-maximum rule decides most windows)."""
    from style_transfer import _hip as hip
    content, style = _smooth(51, size, size), _smooth(52, size, size)
    images = [_smooth(53, size, size)]
    images.append((_smooth(54, size, size) * 4).round() / 4)
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision='fp16x3')
    for k, image in enumerate(images):
        img = image.to(DEV)
        l1, g1 = plan.loss_and_grad(img)
        l1, g1 = l1.clone(), g1.clone()
        with hip.options(ST_POOL_CODES=0):
            l0, g0 = plan.loss_and_grad(img)
        assert torch.isfinite(l1).all() and torch.isfinite(g1).all()
        assert torch.equal(l1, l0) and torch.equal(g1, g0), (size, k, float((g1 - g0).abs().max()))


@pytest.mark.parametrize('size', [128, 512])
def test_head_launches_folded_into_their_neighbours(size, vgg_weights):
    """Launches taken off every style head's dependent chain (they sit on the iteration's critical path): the covariance
    is written by the Gram kernel's finalize pass (ST_GRAM_FUSED_COV), the W2 scalars by the kernel that opens the
    Lyapunov backward chain, and for the fp16x3 chains that kernel also writes a_0 / q_0 straight as planes
    (ST_NS_FUSED_ENTRY).  The same arithmetic in the same order: the closure must be bit-identical to the unfused forms."""
    from style_transfer import _hip as hip
    content, style, image = _smooth(61, size, size), _smooth(62, size, size), _smooth(63, size, size)
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision='fp16x3')
    img = image.to(DEV)
    l1, g1 = plan.loss_and_grad(img)
    l1, g1 = l1.clone(), g1.clone()
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all()
    for switches in (dict(ST_GRAM_FUSED_COV=0), dict(ST_NS_FUSED_ENTRY=0), dict(ST_GRAM_FUSED_COV=0, ST_NS_FUSED_ENTRY=0)):
        with hip.options(**switches):
            l0, g0 = plan.loss_and_grad(img)
            l0, g0 = l0.clone(), g0.clone()
        assert torch.equal(l1, l0), (switches, (l1 - l0).abs().max().item())
        assert torch.equal(g1, g0), (switches, float((g1 - g0).abs().max()))
    # relu5_1's gradient masked by its producer (the head's 1x1 launch) instead of by conv5_1's data gradient while staging:
    # the same operand values, but the fp16x3 scale of that operand now comes from the masked tensor and the launch runs
    # on the producer / consumer kernel - fp32-class agreement, identical losses
    with hip.options(ST_HEAD5_MASK=0):
        l0, g0 = plan.loss_and_grad(img)
    rel = float((g1 - g0).norm() / g0.norm())
    print(f'[parity] relu5_1 gradient masked by its producer, {size}: gradient rel-L2 {rel:.2e}')
    assert torch.equal(l1, l0) and rel < 2e-6
    # taps of <= 1024 pixels: dF = Ssym F + b in one small-GEMM launch (head_dgrad_small_kernel) instead of the convolution
    # launcher's split-K + reduce - fp32 either way, another summation order
    with hip.options(ST_HEAD_SMALL_GEMM=0):
        l0, g0 = plan.loss_and_grad(img)
    rel = float((g1 - g0).norm() / g0.norm())
    print(f'[parity] small taps\' 1x1 gradient as one launch, {size}: gradient rel-L2 {rel:.2e}')
    assert torch.equal(l1, l0) and rel < 2e-6


@pytest.mark.parametrize('kind', ['photo_like', 'white_noise'])
def test_reduced_lyapunov_backward_against_full_recurrence(kind, vgg_weights):
    """The plan's NS backward drops the commutator a^T(a^T q - q a) of sqrtm.py:44 when the incoming gradient is
    a multiple of I (st_smallgemm.hip ns_sqrt_backward); ST_NS_FULL_BACKWARD=1 runs the reference's recurrence
    step for step.  Both on the same plan (n = 64 ... 512 heads incl. the rank-deficient relu5_1 at 256^2) in the
    shipped arithmetic: the image gradients must agree far below the 1e-3 gradient bar, and the reduced form
    must not be further from the oracle than the full one by more than that difference."""
    from style_transfer import _hip as hip
    size = 256
    if kind == 'photo_like':
        content, style, image = _smooth(31, size, size), _smooth(32, size, size), _smooth(33, size, size)
    else:
        gen = torch.Generator().manual_seed(31)
        content, style, image = (torch.rand((1, 3, size, size), generator=gen) for _ in range(3))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    targets = O.build_targets(content, [style], vgg_weights)
    _, _, grad_ref = O.loss_and_grad(image, vgg_weights, targets)
    w64 = [(w.double(), b.double()) for w, b in vgg_weights]
    _, _, grad64 = O.loss_and_grad(image.double(), w64, O.build_targets(content.double(), [style.double()], w64))
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision='fp16x3')
    img = image.to(DEV)
    l_red, g_red = plan.loss_and_grad(img)
    l_red, g_red = l_red.clone(), g_red.clone()
    with hip.options(ST_NS_FULL_BACKWARD=1):
        l_full, g_full = plan.loss_and_grad(img)
        l_full, g_full = l_full.clone(), g_full.clone()
    assert torch.equal(l_red, l_full), 'the backward variant must not change the loss values'
    delta = rel_l2(g_red.cpu(), g_full.cpu())
    e_red, e_full, floor = rel_l2(g_red.cpu(), grad_ref), rel_l2(g_full.cpu(), grad_ref), rel_l2(grad_ref, grad64)
    print(f'[parity] NS backward {kind}: reduced-vs-full rel_l2={delta:.3e}; vs oracle: reduced {e_red:.3e}, '
          f'full {e_full:.3e}; oracle fp32-vs-fp64 {floor:.3e}')
    assert delta <= max(1e-4, 0.5 * floor)
    assert e_red <= GRAD_TOL and e_full <= GRAD_TOL


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


def test_step_tail_in_the_update_kernel_is_bit_identical(vgg_weights):
    """st_plan_step ends an iteration with ONE launch (round 6): conv1_1's fold kernel applies the Adam + clamp + EMA update to
    each gradient element it finishes, totals the loss terms and clears the fp16x3 operand bounds for the next pass
    (ST_STEP_TAIL=2, default); =1: total + clearing ride in a separate update kernel; =0: four launches, as the closure-only
    entry points do.  Every iterate, both moments, the EMA and all eight loss values must agree bit for bit - also with
    forward passes and closures without an update in between (they find the bounds cleared, or clear them themselves)."""
    from style_transfer import _hip as hip
    g = load_golden('iter_tiny')

    def run(tail):
        net, plan = _build_plan(hip, vgg_weights, _t(g['content']), [_t(g['style0'])], [1.0], precision='fp16x3')
        image = _t(g['image0']).to(DEV).clone()
        m, v, ema = torch.zeros_like(image), torch.zeros_like(image), 0.01 * image
        out = []
        with hip.options(ST_STEP_TAIL=tail):
            for step in range(1, 7):
                losses = plan.step(image, m, v, ema, step, 0.02)
                out.append((losses.clone(), image.clone(), m.clone(), v.clone(), ema.clone()))
                if step == 2:
                    plan.forward(image, 29)                      # a pass that consumes the cleared bounds ...
                    out.append((plan.feature(29).clone(),))
                if step == 4:
                    l, gr = plan.loss_and_grad(image)            # ... and a closure behind one: clears them itself
                    out.append((l.clone(), gr.clone()))
        torch.cuda.synchronize()
        return out

    a, b, c = run(2), run(0), run(1)
    assert len(a) == len(b) == len(c) == 8
    for other in (b, c):
        for i, (x, y) in enumerate(zip(a, other)):
            for j, (t, u) in enumerate(zip(x, y)):
                assert torch.equal(t, u), (i, j, float((t - u).abs().max()))
    total = float(a[-1][0][7])
    assert abs(total - float(a[-1][0][:7].sum())) <= 1e-6 * abs(total)
    print(f'[parity] step tail folded into the update kernel: 6 iterates + interleaved passes bit-identical, loss {total:.6f}')


def _check_result(name, res, want, golden=None):
    """Final averaged image against the reference's.  The first Adam updates are lr * sign(g): a pixel whose
    gradient is ~0 may step the other way under fp32 rounding, so the bulk of the image is judged (mean and
    the fraction of outliers), plus a loose cap on the worst pixel.  Where the fixture records how far the
    reference's OWN result moves under rounding-level perturbations (1 thread instead of 8, one bias scaled by
    1 + 1e-6: `result_spread`, `result_outlier_frac` - init='gray' starts from a constant image whose gradient is
    ~0 almost everywhere, and 1 % of its pixels flip), 3 x that spread is allowed."""
    diff = (res - want).abs()
    frac = float((diff > 1e-3).float().mean())
    mean_tol, frac_tol, max_tol = 5e-5, 2e-3, 2e-2
    if golden is not None and 'result_spread' in golden:
        mean_tol = max(mean_tol, 3 * float(golden['result_spread']))
        frac_tol = max(frac_tol, 3 * float(golden['result_outlier_frac']))
        if float(golden['result_outlier_frac']) > 2e-3:
            max_tol = 0.2                                    # flipped pixels differ by a few lr steps
    print(f'[parity] {name} result: max_abs={float(diff.max()):.3e} mean_abs={float(diff.mean()):.3e} '
          f'{100 * frac:.3f}% of values off by > 1e-3 (allowed {mean_tol:.1e}, {100 * frac_tol:.3f}%)')
    assert float(diff.mean()) < mean_tol and frac < frac_tol and float(diff.max()) < max_tol


@pytest.mark.parametrize('mode,scale', [('bicubic', (64, 64)), ('bilinear', (64, 64)), ('bicubic', (57, 68)),
                                        ('bilinear', (57, 68))])
def test_gpu_resampling_matches_cpu(mode, scale):
    """Scale transition (style_transfer.py:279-295,420): the drop-in resamples image and Adam moments with
    F.interpolate on the HIP device, the reference on the CPU.  Isolated here: same input, both devices."""
    g = torch.Generator().manual_seed(9)
    x = torch.rand((1, 3, 45, 45), generator=g)
    cpu = torch.nn.functional.interpolate(x, scale, mode=mode)
    gpu = torch.nn.functional.interpolate(x.to(DEV), scale, mode=mode).cpu()
    d = float((cpu - gpu).abs().max())
    print(f'[parity] F.interpolate {mode} 45x45 -> {scale}: HIP vs CPU max_abs={d:.3e}')
    assert d <= 2e-6


def _stylize_variant(name, vgg_weights, **kw):
    from PIL import Image
    import style_transfer as st_pkg
    g = load_golden(name)
    content = Image.fromarray(g['content_u8'], 'RGB')
    styles = [Image.fromarray(g['style0_u8'], 'RGB'), Image.fromarray(g['style1_u8'], 'RGB')]
    st = st_pkg.StyleTransfer(devices=[DEV], weights=vgg_weights)
    its = []
    torch.manual_seed(0)
    st.stylize(content, styles, style_weights=[0.7, 0.3],
               callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)), **kw)
    got, want = np.array(its, dtype=np.float64), g['iterates']
    assert got.shape == want.shape and np.array_equal(got[:, :4], want[:, :4])
    rels = np.abs(got[:, 4] - want[:, 4]) / np.abs(want[:, 4])
    print(f'[parity] {name} loss trace got {got[:, 4]} rel {rels}')
    return st, rels, g


def test_stylize_lbfgs_against_reference(vgg_weights):
    """optimizer='lbfgs' (style_transfer.py:464-465): torch.optim.LBFGS(max_iter=1, history_size=10) over the
    native loss_and_grad, no clamp (:482-483), two scales (history restarts per scale, no Adam warm start)."""
    st, rels, g = _stylize_variant('stylize_lbfgs', vgg_weights, optimizer='lbfgs', min_scale=45, end_scale=64,
                                   iterations=3, initial_iterations=4)
    # The quasi-Newton recursion amplifies rounding-level gradient differences: the reference's OWN trace moves by
    # `trace_spread` (up to 2e-2 at the 7th iterate) when it runs with 1 thread instead of 8 or with one bias
    # changed by 1e-6 (recorded by make_golden.py).  5x that spread, and never tighter than 5e-4.
    tol = np.maximum(5e-4, 5 * g['trace_spread'])
    print(f'[parity] stylize lbfgs tolerances {tol}')
    assert np.all(rels <= tol)
    res, want = st.get_image_tensor().cpu(), _t(g['result'])
    d = float((res - want).abs().mean())
    print(f'[parity] stylize lbfgs result mean_abs={d:.3e} (reference self-spread {float(g["result_spread"]):.3e})')
    assert d <= max(1e-4, 5 * float(g['result_spread']))


@pytest.mark.parametrize('init', ['gray', 'uniform', 'normal', 'style_stats'])
def test_stylize_init_modes_against_reference(init, vgg_weights):
    """The random `init` modes (style_transfer.py:380-406) under torch.manual_seed(0), as the CLI seeds them
    (cli.py:245): same RNG draws in the same order as the reference; style_stats blends the per-image channel
    statistics with the normalised style weights."""
    kw = dict(min_scale=45, end_scale=64, iterations=3, initial_iterations=4) if init == 'style_stats' else \
        dict(min_scale=64, end_scale=64, initial_iterations=4)
    st, rels, g = _stylize_variant(f'stylize_init_{init}', vgg_weights, init=init, **kw)
    assert np.all(rels <= 5e-4)
    _check_result(f'stylize init={init}', st.get_image_tensor().cpu(), _t(g['result']), g)


@pytest.mark.parametrize('name,kw', [
    ('stylize_params', dict(content_weight=0.05, tv_weight=5.0, step_size=0.03, avg_decay=0.9, style_scale_fac=1.5,
                            min_scale=45, end_scale=64, iterations=3, initial_iterations=4)),
    ('stylize_style_size', dict(style_size=40, min_scale=45, end_scale=64, iterations=3, initial_iterations=4)),
])
def test_stylize_non_default_parameters_against_reference(name, kw, vgg_weights):
    """Every numeric keyword of stylize() away from its default - loss weights, Adam step size, EMA decay, and the two
    ways of scaling the style images (style_scale_fac, style_size: style_transfer.py:433-437) - against reference runs."""
    st, rels, g = _stylize_variant(name, vgg_weights, **kw)
    tol = np.maximum(5e-4, 5 * g['trace_spread'])
    assert np.all(rels <= tol), (rels, tol)
    _check_result(name, st.get_image_tensor().cpu(), _t(g['result']), g)


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
    # one tolerance before and after the scale transition: the resampling (ROCm F.interpolate vs the
    # reference's CPU one, isolated in test_gpu_resampling_matches_cpu) does not add error
    rels = np.abs(got[:, 4] - want[:, 4]) / np.abs(want[:, 4])
    print('[parity] stylize trace rel', rels)
    assert np.all(rels <= 5e-4)
    res = st.get_image_tensor().cpu()
    _check_result('stylize e2e', res, _t(g['result']))
    assert st.get_image('pil').size == (64, 64)
    assert st.get_image('np_uint16').dtype == np.uint16
    with pytest.raises(ValueError):
        st.get_image('bmp')


def test_config1_50_iteration_trace_against_reference(vgg_weights):
    """BASELINE.json configs[0] / SURVEY.md 8(d) C1: 256x256 content + style, one scale, 50 Adam iterations through
    the drop-in stylize(), against the reference's CPU run of the same call (golden `stylize_c1`): the loss of every
    iteration and the averaged result.  Protocol of 8(d): 1e-4 rel per step "if it holds, else report the drift
    curve" - Adam's first steps are lr * sign(g), so rounding-level differences of near-zero gradients grow along the
    trajectory; the fixture records how far the reference's OWN trace moves (1 thread instead of 8, one bias scaled
    by 1 + 1e-6).  Allowed per step: max(1e-4, 5 x that spread); the curve is printed."""
    import synth
    import style_transfer as st_pkg
    from PIL import Image
    g = load_golden('stylize_c1')
    imgs = [synth.smooth_image(int(seed), 256, 256) for seed in g['seeds']]
    assert np.array_equal(synth.checksum(imgs[0]), g['content_checksum'])
    assert np.array_equal(synth.checksum(imgs[1]), g['style_checksum'])
    pil = [Image.fromarray((t[0].permute(1, 2, 0) * 255).round().byte().numpy(), 'RGB') for t in imgs]
    st = st_pkg.StyleTransfer(devices=[DEV], weights=vgg_weights)
    its = []
    torch.manual_seed(0)
    st.stylize(pil[0], [pil[1]], min_scale=256, end_scale=256, initial_iterations=50,
               callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)))
    got, want = np.array(its, dtype=np.float64), g['iterates']
    assert got.shape == want.shape == (50, 5) and np.array_equal(got[:, :4], want[:, :4])
    rels = np.abs(got[:, 4] - want[:, 4]) / np.abs(want[:, 4])
    tol = np.maximum(1e-4, 5 * g['trace_spread'])
    print('[parity] C1 drift curve (rel loss deviation, every 7th iteration):', [f'{r:.1e}' for r in rels[::7]])
    print('[parity] C1 reference self-spread                                :', [f'{r:.1e}' for r in g['trace_spread'][::7]])
    print(f'[parity] C1 loss {want[0, 4]:.4f} -> {want[-1, 4]:.4f}; max rel deviation {rels.max():.2e} at iteration '
          f'{int(rels.argmax()) + 1}')
    assert np.all(rels <= tol), (rels / tol).max()
    res = st.get_image_tensor().cpu()
    d = float((res[:, ::4, ::4] - _t(g['result_sub'])).abs().mean())
    print(f'[parity] C1 result mean_abs={d:.3e} (reference self-spread {float(g["result_spread"]):.3e}), '
          f'mean {float(res.mean()):.6f} vs {float(g["result_mean"]):.6f}')
    assert d <= max(1e-4, 5 * float(g['result_spread']))
    assert abs(float(res.mean()) - float(g['result_mean'])) <= 1e-4


def test_config2_default_run_against_reference(vgg_weights):
    """BASELINE.json configs[1] / SURVEY.md 8(d) C2 end to end: stylize() with EVERY default (scales 128, 181, 256, 362,
    512; 1000 + 4 x 500 Adam iterations) on 512x512 inputs, against the reference's CPU run of the same call (golden
    `stylize_c2`, ~15 CPU-minutes to generate).  3000 Adam steps are chaotic in the pixels - the reference's own result
    moves by `result_spread` when one bias is scaled by 1 + 1e-6 - so the comparison is the LOSS CURVE (every 25th
    iteration, allowed: max(2e-3, 5 x the reference's own spread at that iteration)) and the result's statistics."""
    import time
    import synth
    import style_transfer as st_pkg
    from PIL import Image
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'stylize_c2.npz')):
        pytest.skip('tests/golden/stylize_c2.npz not generated (make_golden.py stylize_c2, ~30 CPU-minutes)')
    g = load_golden('stylize_c2')
    imgs = [synth.smooth_image(int(seed), 512, 512) for seed in g['seeds']]
    assert np.array_equal(synth.checksum(imgs[0]), g['content_checksum'])
    assert np.array_equal(synth.checksum(imgs[1]), g['style_checksum'])
    pil = [Image.fromarray((t[0].permute(1, 2, 0) * 255).round().byte().numpy(), 'RGB') for t in imgs]
    st = st_pkg.StyleTransfer(devices=[DEV], weights=vgg_weights)
    its = []
    torch.manual_seed(0)
    t0 = time.time()
    st.stylize(pil[0], [pil[1]], callback=lambda it: its.append((it.w, it.h, it.i, it.i_max, it.loss)))
    seconds = time.time() - t0
    got = np.array(its, dtype=np.float64)
    assert got.shape == (3000, 5)
    sub, want = got[::25], g['iterates']
    assert np.array_equal(sub[:, :4], want[:, :4]), 'same scales, sizes and iteration counters as the reference'
    rels = np.abs(sub[:, 4] - want[:, 4]) / np.abs(want[:, 4])
    tol = np.maximum(2e-3, 5 * g['trace_spread'])
    worst = int((rels / tol).argmax())
    print(f'[parity] C2 default run: 3000 iterations in {seconds:.1f} s (callback + loss.item() every iteration); loss '
          f'{want[0, 4]:.4f} -> {want[-1, 4]:.5f}; final loss {got[-1, 4]:.6f} vs reference {float(g["last"][4]):.6f}')
    print(f'[parity] C2 loss curve: max rel deviation {rels.max():.2e} (reference self-spread max '
          f'{float(g["max_spread"]):.2e}); worst vs tolerance at sample {worst}: {rels[worst]:.2e} / {tol[worst]:.2e}')
    print('[parity] C2 deviation at the end of each scale:', [f'{rels[i]:.1e}' for i in (39, 59, 79, 99, 119)])
    assert np.all(rels <= tol)
    res = st.get_image_tensor().cpu()
    d = float((res[:, ::8, ::8] - _t(g['result_sub'])).abs().mean())
    print(f'[parity] C2 result: mean {float(res.mean()):.5f} vs {float(g["result_mean"]):.5f} (perturbed reference '
          f'{float(g["result2_mean"]):.5f}), std {float(res.std()):.5f} vs {float(g["result_std"]):.5f}, mean-abs '
          f'difference {d:.2e} (reference self-spread {float(g["result_spread"]):.2e})')
    assert d <= max(1e-3, 3 * float(g['result_spread']))
    assert abs(float(res.mean()) - float(g['result_mean'])) <= max(1e-3, 5 * abs(float(g['result2_mean']) - float(g['result_mean'])))


def test_full_size_properties_512(vgg_weights):
    """BASELINE config 2 size, shipped conv arithmetic: determinism, finiteness and linearity-in-weights at
    512x512 (size-independent properties; the values themselves are pinned by the 512^2 golden and live-oracle
    tests above)."""
    from style_transfer import _hip as hip
    gen = torch.Generator().manual_seed(5)
    content = torch.rand((1, 3, 512, 512), generator=gen)
    style = torch.rand((1, 3, 512, 512), generator=gen)
    image = torch.rand((1, 3, 512, 512), generator=gen).to(DEV)
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision='fp16x3')   # the shipped arithmetic
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
