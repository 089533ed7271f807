"""The flaky TV term of round 4, root-caused in round 5 (profiles/r05_tv_hazard.md).

Cause: ONE instruction of tv_interior_kernel as the optimiser had built it - `v_pk_add_f32 vdst, vsrc0, vsrc1 op_sel:[0,1]
neg_lo:[0,1] neg_hi:[0,1]`, a packed subtraction whose two lanes both read the HIGH half of vsrc1 - returned `vsrc0 - 0` in its
low lane for lanes 48 - 63 of a wave: the horizontal sum of squares got M[2]^2 instead of (M[2] - M[1])^2 for sixteen pixels.
It needs the kernel at the TAIL of a head stream (beside the launch-per-product Newton-Schulz chains) in the FIRST closure of
a fresh plan: 55 - 75 % of such closures, never on a warm plan, never in the slot the kernel ships in.  The shipped kernel
(ST_TV_VARIANT=1) does not contain that instruction (build.py refuses to build any kernel that does); the old code is kept as
ST_TV_VARIANT=0, the reproducer.

This test rebuilds the failing situation (ST_TV_SLOT=1, ST_HEAD_LOCKSTEP=0, ST_NS_CHAIN=0, a fresh plan per closure) 50 times
at 2048^2 and requires every workgroup's four partial sums to be bit-identical to a quiet run.  The reproducer's failure count
under the same conditions is printed, not asserted (it is a property of the hardware, and a quiet box may show none)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SIZE = 2048
RUNS = 50


def _plan(hip, net, content, style):
    plan = hip.Plan(net, SIZE, SIZE)
    plan.forward(content, 22)
    plan.set_content_target_from_forward()
    plan.forward(style, 29)
    for i, layer in enumerate([1, 6, 11, 20, 29]):
        plan.set_style_target(i, *plan.moments(layer))
    plan.set_loss_weights(0.015, [w / 341 for w in (256, 64, 16, 4, 1)], 2.0)
    return plan


def test_tv_partial_sums_in_the_failing_slot_are_bit_identical(vgg_weights):
    import synth
    from style_transfer import _hip as hip
    content, style = (synth.smooth_image(70 + i, SIZE, SIZE).to(DEV) for i in range(2))
    image = content.clone()
    grad = torch.empty_like(image)
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    nblocks = min(3 * SIZE, 2048 - 256)                 # tv_interior_kernel's grid (one image row per workgroup pass)
    count = 4 * nblocks

    # yardstick: the shipped configuration on a warm plan, and the operator alone on a quiet device
    warm = _plan(hip, net, content, style)
    warm.loss_and_grad(image, grad)
    warm.loss_and_grad(image, grad)
    torch.cuda.synchronize()
    ref = warm.debug_read(0, count).view(nblocks, 4).clone()
    tv_ref = float(warm.losses[6].item())
    alone = float(hip.op_tv_loss(image)[0].item())
    assert abs(alone * 2.0 - tv_ref) <= 1e-6 * tv_ref, (alone, tv_ref)      # (tv_weight 2.0)
    del warm

    def failures(variant, runs):
        bad = 0
        for _ in range(runs):
            plan = _plan(hip, net, content, style)
            with hip.options(ST_TV_SLOT=1, ST_TV_VARIANT=variant, ST_HEAD_LOCKSTEP=0, ST_NS_CHAIN=0):
                plan.loss_and_grad(image, grad)
            torch.cuda.synchronize()
            part = plan.debug_read(0, count).view(nblocks, 4)
            if not torch.equal(part, ref):
                bad += 1
            del plan
            torch.cuda.empty_cache()
        return bad

    shipped = failures(1, RUNS)
    print(f'[tv hazard] shipped kernel, failing slot, fresh plans: {shipped} / {RUNS} closures with a differing partial sum')
    assert shipped == 0
    if os.environ.get('ST_TEST_TV_REPRODUCER', '1') != '0' and hip.has_experiments():      # (the reproducer kernel: --experiments builds)
        old = failures(0, 16)
        print(f'[tv hazard] reproducer (v_pk_add_f32 ... op_sel:[0,1]): {old} / 16 closures with a differing partial sum '
              f'(55 - 75 % on the round-5 boxes)')
