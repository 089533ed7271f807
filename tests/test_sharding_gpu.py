"""Strip sharding on a real MI355X, emulating R ranks on ONE GPU (gpurun grants a single GPU): R strip
plans in one process run the same kernels and produce the same exchange descriptors as R processes
would; only the transport (RCCL) is replaced by device copies.  The sharded result must reproduce the
unsharded HIP path (which the other tests pin to the reference): conv sums are bit-identical, only the
Gram / loss partial sums are combined in a different order."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
import st_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _smooth(seed, h, w):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand((1, 3, max(h // 16, 2), max(w // 16, 2)), generator=g)
    img = torch.nn.functional.interpolate(low, (h, w), mode='bicubic', align_corners=False)
    return (img + (torch.rand((1, 3, h, w), generator=g) - 0.5) * (24 / 255)).clamp(0, 1).contiguous()


def _targets_lockstep(sh, plans, content, styles, weights):
    rows = [(p.row_begin, p.row_end) for p in plans]
    for p, (b, e) in zip(plans, rows):
        p.forward_begin(content[:, :, b:e].contiguous().to(DEV), 22)
    sh.run_phases_lockstep(plans)
    for p in plans:
        p.set_content_target_from_forward()
    blended = {}
    for style, sw in zip(styles, weights):
        assert style.shape == content.shape          # same strips; other sizes need their own strip plans
        for p, (b, e) in zip(plans, rows):
            p.forward_begin(style[:, :, b:e].contiguous().to(DEV), 29)
        sh.run_phases_lockstep(plans)
        for layer in O.STYLE_LAYERS:
            total = sum(p.moment_sums(layer) for p in plans)
            c = {1: 64, 6: 128, 11: 256, 20: 512, 29: 512}[layer]
            level = {1: 0, 6: 1, 11: 2, 20: 3, 29: 4}[layer]
            npix = float((content.shape[2] >> level) * (content.shape[3] >> level))
            srm, mean = (total[:c * c] / npix).reshape(c, c) * sw, (total[c * c:] / npix) * sw
            if layer not in blended:
                blended[layer] = [mean, srm]
            else:
                blended[layer][0] += mean
                blended[layer][1] += srm
    for p in plans:
        for i, layer in enumerate(O.STYLE_LAYERS):
            p.set_style_target(i, blended[layer][0].contiguous(), blended[layer][1].contiguous())
        p.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)


# (ST_STRIP_OVERLAP, ST_STRIP_NS_OWNER): default = the cost model decides which convolutions are cut into an interior
# and a boundary launch around their halo exchange, each head's chains on one owner rank; 2 = cut every convolution the
# kernel can (at these small sizes the model rarely does); (0, 0) = rounds 1 / 2: whole launches, every rank runs every chain
@pytest.mark.parametrize('overlap,owner', [(1, 1), (2, 1), (0, 0)])
@pytest.mark.parametrize('h,w,world', [(96, 80, 2), (96, 80, 3), (135, 181, 2), (256, 128, 4)])
@pytest.mark.parametrize('precision', ['fp32', 'bf16x6', 'fp16x3'])
def test_sharded_closure_and_update_match_unsharded(h, w, world, precision, overlap, owner, vgg_weights):
    _sharded_case(h, w, world, precision, overlap, owner, vgg_weights)


@pytest.mark.parametrize('h,w,rows', [(256, 128, [(0, 48), (48, 144), (144, 256)]), (135, 181, [(0, 16), (16, 135)])])
def test_strips_of_unequal_height_match_unsharded(h, w, rows, vgg_weights):
    """sharding.strip_rows(height, world, width) gives the owner of relu5_1's chains a shorter strip: nothing in the closure may
    assume strips of (nearly) equal height."""
    _sharded_case(h, w, len(rows), 'fp16x3', 1, 1, vgg_weights, rows=rows)


def test_halo_bounds_travel_with_the_rows(vgg_weights):
    """fp16x3: the consumer of a halo row scales its operand by max(|operand|, |halo rows|).  The sender measures its rows while
    it packs them and ships the word in the message's trailer (csrc/st_api.hip halo_exchange); the round-4 form measured them on
    the receiver (ST_STRIP_HALO_BOUND=0: a copy + an amax launch per exchange).  Same maxima, same exponents: the closure must
    agree bit for bit - also where the bound matters, an image whose upper strip is 1000 x darker than the rows below it."""
    from style_transfer import _hip as hip
    out = {}
    for dark in (False, True):
        for shipped in (1, 0):
            with hip.options(ST_STRIP_HALO_BOUND=shipped):
                out[shipped] = _sharded_case(96, 80, 3, 'fp16x3', 1, 1, vgg_weights, dark_top=dark)
        assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1]), f'dark_top={dark}'


def _sharded_case(h, w, world, precision, overlap, owner, vgg_weights, rows=None, dark_top=False):
    from style_transfer import _hip as hip, sharding as sh
    if precision != 'fp16x3' and (overlap, owner) == (2, 1):
        pytest.skip('only the fp16x3 producer / consumer kernel has interior / boundary launches')
    content, style, image = _smooth(31, h, w), _smooth(32, h, w), _smooth(33, h, w)
    if dark_top:
        # (VGG's normalisation maps black to large negative inputs: "dark" is relative to the channel means)
        mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
        top = h // world
        image = image.clone()
        image[:, :, :top] = mean + (image[:, :, :top] - mean) * 1e-3
    net = hip.Net(vgg_weights, 'max', DEV, precision)
    # unsharded reference run of the same HIP code
    whole = hip.Plan(net, h, w)
    whole.forward(content.to(DEV), 22)
    whole.set_content_target_from_forward()
    whole.forward(style.to(DEV), 29)
    for i, layer in enumerate(O.STYLE_LAYERS):
        whole.set_style_target(i, *whole.moments(layer))
    whole.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
    img_w = image.to(DEV).clone()
    losses_w, grad_w = whole.loss_and_grad(img_w)
    losses_w, grad_w = losses_w.clone(), grad_w.clone()

    rows = rows or sh.strip_rows(h, world)
    # (rank / world set: every style head's chains run on ONE owner plan, the others receive its broadcast)
    plans = [sh.StripPlan(net, h, w, b, e).set_rank(r, world) for r, (b, e) in enumerate(rows)]
    _targets_lockstep(sh, plans, content, [style], [1.0])
    imgs = [image[:, :, b:e].contiguous().to(DEV) for b, e in rows]
    grads = [torch.empty_like(t) for t in imgs]
    with hip.options(ST_STRIP_OVERLAP=overlap, ST_STRIP_NS_OWNER=owner):
        for p, t, g in zip(plans, imgs, grads):
            p.closure_begin(t, g)
        sh.run_phases_lockstep(plans)
        torch.cuda.synchronize()
    for r, p in enumerate(plans):
        rel = ((p.losses - losses_w).abs() / losses_w.abs()).max().item()
        print(f'[shard] {h}x{w} R={world} overlap={overlap} owner={owner} rank {r}: max rel loss diff {rel:.2e}')
        assert rel < 5e-5, (r, p.losses, losses_w)
        assert torch.equal(p.losses, plans[0].losses), 'every rank must report identical losses'
    grad_s = torch.cat(grads, dim=2)
    err = rel_l2(grad_s.cpu(), grad_w.cpu())
    print(f'[shard] {h}x{w} R={world}: gradient rel_l2 vs unsharded {err:.2e}')
    # overlap == 2 forces interior + boundary launches on layers so small that the unsharded plan runs them with a K
    # split: another summation order per pixel (3e-7 per conv, tests/test_large_strips_gpu.py
    # test_conv_interior_plus_boundary_equals_one_launch), which the non-converged NS chains of these tiny, rank-deficient
    # taps (relu5_1 of a 96 x 80 image has 30 pixels for 512 channels) amplify to 8e-4 - the same 8e-4 for 2 and for 3
    # strips, i.e. a property of the reference run, not of the seams.  The bar for that mode is the oracle-level 1e-3.
    assert err < (1e-3 if overlap == 2 else 2e-4)

    # one Adam/clamp/EMA update per strip == the same update on the whole image
    m_w, v_w = torch.zeros_like(img_w), torch.zeros_like(img_w)
    e_w = (1 - torch.tensor(0.99)).to(DEV) * img_w
    whole.step(img_w, m_w, v_w, e_w, 1, 0.02)
    parts = []
    for p, t, g in zip(plans, imgs, grads):
        m, v = torch.zeros_like(t), torch.zeros_like(t)
        e = (1 - torch.tensor(0.99)).to(DEV) * t
        p.apply_update(t, g, m, v, e, 1, 0.02)
        parts.append(t)
    diff = (torch.cat(parts, dim=2) - img_w).abs()
    frac = float((diff > 1e-4).float().mean())
    print(f'[shard] post-update image: mean_abs {float(diff.mean()):.2e}, {100 * frac:.3f}% pixels off > 1e-4')
    assert float(diff.mean()) < 1e-5 and frac < 5e-3
    return grad_s.cpu(), plans[0].losses.cpu().clone()


def test_stub_runs_replay_the_wait_for_the_chain_owner(vgg_weights):
    """tools/strip_bench.py times one rank at a time with stubbed exchanges; a stubbed broadcast returns at once, so a rank that
    does not own relu5_1's Newton-Schulz chains would never wait for them (rounds 3 - 5 modelled it that way).
    run_phases_lockstep(..., owner_chains=...) measures the owner's reduction -> broadcast time and replays it on the other
    ranks: the measurement must see the chains (a positive time for head 4 on rank 0, nothing for heads rank 0 does not own),
    and a stub run with the delay must complete and take at least the delay longer than one without."""
    import time
    from style_transfer import _hip as hip, sharding as sh
    h, w, world = 128, 96, 2
    content, style, image = _smooth(61, h, w), _smooth(62, h, w), _smooth(63, h, w)
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    rows = sh.strip_rows(h, world)
    plans = [sh.StripPlan(net, h, w, b, e).set_rank(r, world) for r, (b, e) in enumerate(rows)]
    _targets_lockstep(sh, plans, content, [style], [1.0])
    imgs = [image[:, :, b:e].contiguous().to(DEV) for b, e in rows]
    grads = [torch.empty_like(t) for t in imgs]

    def stub(r, oc=None):
        plans[r].closure_begin(imgs[r], grads[r])
        sh.run_phases_lockstep([plans[r]], stub=True, owner_chains=oc)

    oc = {'rank': 0, 'measure': {}}
    for _ in range(3):
        oc['measure'].clear()
        stub(0, oc)
    torch.cuda.synchronize()
    us = sh.owner_chain_us(oc)
    assert set(us) == {0, 2, 4} and us[4] > 50, us           # rank 0 of 2 owns heads 4, 2, 0: (4 - head) % world
    cycles = sh.sleep_cycles_per_us(DEV)
    assert cycles > 1

    def timed(oc):
        for _ in range(2):
            stub(1, oc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            stub(1, oc)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5 * 1e6

    delay = 3000.0                                            # far above any run-to-run noise
    plain, held = timed(None), timed({'rank': 1, 'delay_us': {4: delay}, 'cycles_per_us': cycles})
    print(f'[shard] stub run of rank 1: {plain:.0f} us; with head 4 held back {delay:.0f} us: {held:.0f} us; '
          f"owner's reduction -> broadcast of head 4: {us[4]:.0f} us")
    assert held > plain + 0.8 * delay


def test_single_strip_is_the_whole_image(vgg_weights):
    """world == 1 through the sharded driver: no neighbours, identical (bitwise) conv path."""
    from style_transfer import _hip as hip, sharding as sh
    h, w = 64, 96
    content, style, image = _smooth(41, h, w), _smooth(42, h, w), _smooth(43, h, w)
    net = hip.Net(vgg_weights, 'max', DEV)
    plan = sh.StripPlan(net, h, w, 0, h)
    _targets_lockstep(sh, [plan], content, [style], [1.0])
    img = image.to(DEV)
    grad = torch.empty_like(img)
    plan.closure_begin(img, grad)
    sh.run_phases_lockstep([plan])
    assert torch.isfinite(plan.losses).all() and torch.isfinite(grad).all()
    with pytest.raises(ValueError):
        sh.StripPlan(net, 100, 64, 8, 100)          # boundary not a multiple of 16
