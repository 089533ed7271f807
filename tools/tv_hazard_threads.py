"""Which THREADS of tv_interior_kernel carry the wrong horizontal sum (tools/tv_hazard.py found: first closure of a fresh plan,
TV at the tail of relu1_1's head stream, ~60 % of the closures)?  ST_TV_VARIANT=3 = the shipped kernel + every thread's four
accumulators and its group count dumped before the block reduction.

    gpurun -- python tools/tv_hazard_threads.py [--reps 12]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'style-transfer-pytorch_amd'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from style_transfer import _hip, vgg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=2048)
    ap.add_argument('--reps', type=int, default=12)
    a = ap.parse_args()
    dev = 'cuda:0'
    import bench
    S = a.size
    weights = vgg.synthetic_vgg19_weights(0)
    content = bench.synthetic_image(100, S, S)
    style = bench.synthetic_image(200, S, S)
    net = _hip.Net(weights, 'max', dev, 'fp16x3')
    nblocks = min(3 * S, 2048 - 256)
    image = content.to(dev).clone()
    grad = torch.empty_like(image)

    def make_plan():
        pl = _hip.Plan(net, S, S)
        pl.forward(content.to(dev), 22)
        pl.set_content_target_from_forward()
        pl.forward(style.to(dev), 29)
        for i, layer in enumerate([1, 6, 11, 20, 29]):
            pl.set_style_target(i, *pl.moments(layer))
        pl.set_loss_weights(0.015, [w / 341 for w in (256, 64, 16, 4, 1)], 2.0)
        return pl

    def closure(pl, slot):
        with _hip.options(ST_TV_SLOT=slot, ST_TV_VARIANT=3, ST_HEAD_LOCKSTEP=0):
            pl.loss_and_grad(image, grad)
        torch.cuda.synchronize()
        return pl.debug_read(1, nblocks * 256 * 5).view(nblocks, 256, 5).clone(), grad.clone()

    warm = make_plan()
    closure(warm, 0)
    ref, gref = closure(warm, 0)
    again, _ = closure(warm, 0)
    print(f'# reference: warm plan, shipped slot; repeat bit-identical: {torch.equal(ref, again)}', flush=True)
    # per (row, thread) contribution to s1: thread t owns groups t and t + 256 of a row (W = 2048), minus the first / last group
    x = content[0].double()
    d1 = (x[:, :, 5:S - 3] - x[:, :, 4:S - 4]) ** 2                     # columns 4 .. S - 5
    d1 = torch.nn.functional.pad(d1, (4, 4)).reshape(3 * S, S // 4, 4).sum(-1)     # [row][group]
    d1[:, 0] = 0
    d1[:, S // 4 - 1] = 0
    contrib = d1[:, :256] + d1[:, 256:512] if S == 2048 else None       # [row][thread]
    for rep in range(a.reps):
        pl = make_plan()
        got, g = closure(pl, 1)
        diff = (got != ref)
        bad = diff.any(-1).nonzero()
        print(f'rep {rep}: {bad.shape[0]} thread(s) differ; gradient bit-identical to the reference: {torch.equal(g, gref)}', flush=True)
        per_wg = {}
        for wg, t in bad.tolist():
            per_wg.setdefault(wg, []).append(t)
        for wg, ts in list(per_wg.items())[:6]:
            comps = diff[wg].any(0).tolist()
            waves = sorted({t // 64 for t in ts})
            dl = (got[wg, :, 0].double() - ref[wg, :, 0].double())
            print(f'  workgroup {wg}: {len(ts)} threads (waves {waves}, lanes {min(t % 64 for t in ts)}..{max(t % 64 for t in ts)}); components that differ '
                  f'[s1 s2 s3 s4 visited] = {comps}; sum of s1 deltas {float(dl.sum()):+.6g}; visited ref/got of the first: '
                  f'{float(ref[wg, ts[0], 4])}/{float(got[wg, ts[0], 4])}', flush=True)
            if contrib is not None:
                # does each thread's delta equal ITS OWN contribution of some row?
                own = list(range(wg, 3 * S, nblocks))
                t0 = ts[0]
                cand = (contrib[:, t0] - float(dl[t0])).abs()
                r = int(cand.argmin())
                print(f'    thread {t0}: delta {float(dl[t0]):+.8g}; its contribution of its own rows {[round(float(contrib[q, t0]), 6) for q in own]}; '
                      f'closest row contribution: row {r} ({float(contrib[r, t0]):.8g}, |diff| {float(cand[r]):.2e})', flush=True)
                # all threads against the same row?
                rows_hit = {}
                for t in ts:
                    c = (contrib[:, t] - float(dl[t])).abs()
                    rr = int(c.argmin())
                    if float(c[rr]) < 1e-6 * max(1.0, abs(float(dl[t]))) + 1e-7:
                        rows_hit[rr] = rows_hit.get(rr, 0) + 1
                print(f'    rows whose per-thread contribution explains a delta exactly: {dict(sorted(rows_hit.items())[:8])}', flush=True)
        # what exactly was added?  For the first failing threads: (delta + the true term of one of its pixels) against every
        # value the thread held for that group - a pixel, a pixel squared, a product or a squared difference of two pixels
        shown = 0
        xf = content[0].float()
        for wg, ts in list(per_wg.items())[:3]:
            for t in ts[:4]:
                dl = float(got[wg, t, 0].double() - ref[wg, t, 0].double())
                best = None
                for rowi in range(wg, 3 * S, nblocks):
                    ch, y = divmod(rowi, S)
                    if y < 1 or y > S - 2:
                        continue
                    for g4 in (t, t + 256):
                        if g4 == 0 or g4 == S // 4 - 1 or g4 >= S // 4:
                            continue
                        c0 = 4 * g4
                        px = {}
                        for name, yy in (('U', y - 1), ('M', y), ('D', y + 1)):
                            for k in range(6):
                                px[f'{name}{k}'] = float(xf[ch, yy, c0 - 1 + k])
                        names = list(px)
                        for j in range(4):
                            true = (px[f'M{j + 2}'] - px[f'M{j + 1}']) ** 2
                            wrong = dl + true
                            for a_ in names:
                                for cand, label in ((px[a_], a_), (px[a_] ** 2, a_ + '^2')):
                                    err = abs(cand - wrong)
                                    if best is None or err < best[0]:
                                        best = (err, f'row {rowi} group {g4} j={j}: term = {label} ({cand:.8g})')
                                for b_ in names:
                                    if b_ <= a_:
                                        continue
                                    for cand, label in (((px[a_] - px[b_]) ** 2, f'({a_}-{b_})^2'), (px[a_] * px[b_], f'{a_}*{b_}')):
                                        err = abs(cand - wrong)
                                        if err < best[0]:
                                            best = (err, f'row {rowi} group {g4} j={j}: term = {label} ({cand:.8g})')
                print(f'    workgroup {wg} thread {t} (lane {t % 64}): delta {dl:+.8g}; best explanation |err| {best[0]:.2e}: {best[1]}', flush=True)
        del pl
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
