"""Stress of the flaky TV term of round 4 (profiles/r04_streams.md section 4).

tv_interior_kernel reported, in about one run of two at 2048^2, a few workgroups whose horizontal sum was too large when the
kernel ran at the TAIL of the shallow heads' stream (beside the backward trunk).  ST_TV_SLOT=1 rebuilds that slot; this
script runs the closure REPS times per kernel variant (ST_TV_VARIANT, csrc/st_pointwise.hip) in both slots, compares every
workgroup's four partial sums bit for bit with the same variant's result in isolation (st_op_tv_loss-equivalent: the
shipped slot at the start of the iteration is the yardstick only after it has itself been checked against a quiet device),
and for every mismatch says which component moved, by how much, and which image rows' sums that amount matches.

    gpurun -- python tools/tv_hazard.py [--size 2048] [--reps 40]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from style_transfer import _hip, vgg


def interior_row_sums(img):
    """Per (channel, row) sums of squares the interior kernel owns: columns of groups 1 .. W/4 - 2, rows 1 .. H - 2."""
    x = img[0].double()
    H, W = x.shape[-2:]
    cols = slice(4, W - 4)
    d1 = (x[:, :, 5:W - 3] - x[:, :, cols]) ** 2                    # M[j+2] - cc
    d2 = torch.zeros_like(d1)
    d2[:, :-1] = (x[:, 1:, cols] - x[:, :-1, cols]) ** 2            # D[j+1] - cc
    d3 = torch.zeros_like(d1)
    d3[:, 1:] = (x[:, 1:, cols] - x[:, :-1, 3:W - 5]) ** 2          # cc - U[j]
    d4 = torch.zeros_like(d1)
    d4[:, 1:] = (x[:, 1:, 3:W - 5] - x[:, :-1, cols]) ** 2          # M[j] - U[j+1]
    sums = torch.stack([d.sum(-1) for d in (d1, d2, d3, d4)], 0)    # [4][3][H]
    sums[:, :, 0] = 0
    sums[:, :, H - 1] = 0
    return sums.reshape(4, 3 * H)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--variants', default='1,0', help='ST_TV_VARIANT: 1 shipped, 0 the reproducer')
    ap.add_argument('--lockstep', type=int, default=1, help='ST_HEAD_LOCKSTEP (0: one stream per head - the layout of the round-4 failure)')
    ap.add_argument('--cold', type=int, default=0, help='N > 0: also N single closures on FRESH plans with a temporary image tensor')
    a = ap.parse_args()
    dev = 'cuda:0'
    import bench
    S = a.size
    weights = vgg.synthetic_vgg19_weights(0)
    content = bench.synthetic_image(100, S, S)
    style = bench.synthetic_image(200, S, S)
    net = _hip.Net(weights, 'max', dev, 'fp16x3')
    plan = _hip.Plan(net, S, S)
    plan.forward(content.to(dev), 22)
    plan.set_content_target_from_forward()
    plan.forward(style.to(dev), 29)
    for i, layer in enumerate([1, 6, 11, 20, 29]):
        plan.set_style_target(i, *plan.moments(layer))
    plan.set_loss_weights(0.015, [w / 341 for w in (256, 64, 16, 4, 1)], 2.0)
    image = content.to(dev).clone()
    grad = torch.empty_like(image)
    rows = interior_row_sums(content)
    H = S
    nblocks = min(3 * H, 2048 - 256)                                # tv_interior_kernel's grid at widths >= 1024 (one row per block)
    count = 4 * nblocks
    print(f'# {S}x{S}, {a.reps} closures per cell, {nblocks} interior workgroups', flush=True)

    def run(slot, variant, reps):
        out = []
        with _hip.options(ST_TV_SLOT=slot, ST_TV_VARIANT=variant, ST_HEAD_LOCKSTEP=a.lockstep):
            for _ in range(reps):
                losses, _g = plan.loss_and_grad(image, grad)
                torch.cuda.synchronize()
                out.append((plan.debug_read(0, count).view(nblocks, 4), float(losses[6].item())))
        return out

    for variant in [int(v) for v in a.variants.split(',')]:
        # yardstick: TV alone on a quiet device (the operator), then the shipped slot
        with _hip.options(ST_TV_VARIANT=variant):
            tv_alone = [float(_hip.op_tv_loss(image)[0].item()) for _ in range(3)]
        ref_runs = run(0, variant, 6)
        ref = ref_runs[0][0]
        ref_ok = all(torch.equal(r[0], ref) for r in ref_runs)
        print(f'variant {variant}: operator alone (x weight 1) {tv_alone}; shipped slot bit-identical over 6 runs: {ref_ok}; '
              f'TV term {ref_runs[0][1]:.9g}', flush=True)
        for slot in (1, 0):
            runs = run(slot, variant, a.reps)
            bad_runs = 0
            for k, (part, term) in enumerate(runs):
                diff = part != ref
                if not bool(diff.any()):
                    continue
                bad_runs += 1
                if bad_runs > 6:
                    continue
                idx = diff.nonzero()
                print(f'  slot {slot} run {k}: TV term {term:.9g} ({(term - ref_runs[0][1]) / ref_runs[0][1]:+.2e}), '
                      f'{idx.shape[0]} partial(s) differ', flush=True)
                for wg, comp in idx[:8].tolist():
                    delta = float(part[wg, comp].double() - ref[wg, comp].double())
                    own = [r for r in range(wg, 3 * H, nblocks)]
                    own_sums = [float(rows[comp, r]) for r in own]
                    near = torch.argmin((rows[comp] - delta).abs()).item()
                    print(f'    workgroup {wg} component s{comp + 1}: {float(ref[wg, comp]):.6g} -> {float(part[wg, comp]):.6g} '
                          f'(delta {delta:+.6g}); its rows {own} sum {own_sums}; quarter rows (one wave) '
                          f'{[s / 4 for s in own_sums]}; closest single row: {near} ({float(rows[comp, near]):.6g})', flush=True)
            print(f'  variant {variant} slot {slot} ({"tail of the shallow heads stream" if slot else "shipped"}): '
                  f'{bad_runs} / {a.reps} runs with a differing partial', flush=True)
        # the round-4 failure was the FIRST closure of a fresh plan, called with a temporary image tensor
        # (tests/test_large_strips_gpu.py: whole.loss_and_grad(image.to(DEV)) under ST_HEAD_LOCKSTEP=0): which of the two matters?
        def make_plan():
            fresh = _hip.Plan(net, S, S)
            fresh.forward(content.to(dev), 22)
            fresh.set_content_target_from_forward()
            fresh.forward(style.to(dev), 29)
            for i, layer in enumerate([1, 6, 11, 20, 29]):
                fresh.set_style_target(i, *fresh.moments(layer))
            fresh.set_loss_weights(0.015, [w / 341 for w in (256, 64, 16, 4, 1)], 2.0)
            return fresh

        for fresh_plan, temp_image, slot in ((1, 1, 1), (1, 0, 1), (0, 1, 1), (1, 1, 0)):
            if not a.cold:
                break
            bad_cold = 0
            for k in range(a.cold):
                pl = make_plan() if fresh_plan else plan
                with _hip.options(ST_TV_SLOT=slot, ST_TV_VARIANT=variant, ST_HEAD_LOCKSTEP=a.lockstep):
                    if temp_image:
                        losses_w, grad_w = pl.loss_and_grad(content.to(dev))
                    else:
                        losses_w, grad_w = pl.loss_and_grad(image, grad)
                    losses_w, grad_w = losses_w.clone(), grad_w.clone()
                torch.cuda.synchronize()
                part = pl.debug_read(0, count).view(nblocks, 4)
                if not torch.equal(part, ref):
                    bad_cold += 1
                    idx = (part != ref).nonzero()
                    print(f'  fresh={fresh_plan} temp={temp_image} slot={slot} rep {k}: TV term {float(losses_w[6]):.9g}, {idx.shape[0]} partial(s) differ', flush=True)
                    for wg, comp in idx[:5].tolist():
                        delta = float(part[wg, comp].double() - ref[wg, comp].double())
                        own = [r for r in range(wg, 3 * H, nblocks)]
                        own_sums = [round(float(rows[comp, r]), 4) for r in own]
                        near = torch.argmin((rows[comp] - delta).abs()).item()
                        print(f'    workgroup {wg} s{comp + 1}: {float(ref[wg, comp]):.6g} -> {float(part[wg, comp]):.6g} (delta {delta:+.6g}); '
                              f'its rows {own} sum {own_sums}; closest single row: {near} ({float(rows[comp, near]):.6g})', flush=True)
                if fresh_plan:
                    del pl
                    torch.cuda.empty_cache()
            print(f'  variant {variant} fresh plan={fresh_plan} temporary image={temp_image} slot={slot}: {bad_cold} / {a.cold} closures with a '
                  f'differing partial', flush=True)

if __name__ == '__main__':
    main()
