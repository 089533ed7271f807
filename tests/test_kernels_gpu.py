"""Kernel-level parity on a real MI355X: every HIP kernel, called through the C ABI, against the CPU
oracle / the matching torch CPU operator on the same seeded inputs.  Tolerances are fp32 ones:
the kernels accumulate in fp32 FMA chains (MFMA or VALU) in a different order than MKL/oneDNN."""
import numpy as np
import pytest
import torch
from torch.nn import functional as F

from conftest import load_golden, rel_l2
import st_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _hip():
    from style_transfer import _hip
    return _hip


def _report(name, got, want, tol):
    err = rel_l2(got.detach().cpu(), want)
    amax = float((got.detach().cpu().double() - want.double()).abs().max())
    print(f'[parity] {name}: rel_l2={err:.3e} max_abs={amax:.3e} (tol {tol:.1e})')
    assert np.isfinite(err) and err <= tol, f'{name}: rel_l2 {err:.3e} > {tol:.1e} (max_abs {amax:.3e})'


CONV_SHAPES = [  # cin, cout, h, w
    (64, 64, 40, 48), (64, 128, 20, 24), (128, 256, 33, 45), (512, 512, 8, 8), (256, 256, 64, 64),
    (64, 64, 130, 70), (512, 512, 16, 11), (128, 128, 9, 100), (256, 512, 5, 5),
]


# conv arithmetic modes: exact fp32 MFMA / split-precision bf16 MFMA with 6 or 3 products / scaled fp16 planes
# with 3 products (tolerance vs fp32 CPU)
PRECISIONS = [(0, 2e-6), (3, 3e-6), (2, 4e-5), (4, 3e-6)]


@pytest.mark.parametrize('cin,cout,h,w', CONV_SHAPES)
@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('precision,tol', PRECISIONS)
def test_conv3x3_forward(cin, cout, h, w, relu, precision, tol):
    g = torch.Generator().manual_seed(cin * 7 + cout + h * 3 + w)
    x = torch.randn((1, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    want = F.conv2d(x, wt, b, padding=1)
    if relu:
        want = want.relu()
    got = _hip().op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), relu, precision)
    _report(f'conv3x3 fwd p{precision} {cin}->{cout} {h}x{w} relu={relu}', got, want, tol)


@pytest.mark.parametrize('cin,cout,h,w', CONV_SHAPES)
@pytest.mark.parametrize('precision,tol', PRECISIONS)
def test_conv3x3_data_gradient_with_relu_mask(cin, cout, h, w, precision, tol):
    g = torch.Generator().manual_seed(cin + cout * 5 + h + w * 11)
    x = torch.randn((1, cin, h, w), generator=g, requires_grad=True)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    y = F.conv2d(x, wt, b, padding=1).relu()
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    got = _hip().op_conv3x3_dgrad(gy.to(DEV), y.detach().to(DEV), wt.to(DEV), precision)
    _report(f'conv3x3 dgrad p{precision} {cout}->{cin} {h}x{w}', got, x.grad, tol)


# ---- producer / consumer kernel (st_conv_pc.hip): every tile variant, at the sizes the BASELINE configs run -----
# (cin, cout, h, w, variant the ROUND-1 rule picks (ST_CONV_PC_MODEL=0), ksplit the same in both kernels?)  Rules:
# st_conv_pc.hip launch_conv_pc / xl_tile_pays - XL <32,2,8> needs ceil(w/32) * ceil(h/16) * cout/64 >= 256 tiles;
# <TW,2,4> needs 256 <= ceil(hw/256) * cout/64 (< 512 and cin >= 256 to be preferred); <TW,1,4> below that.
# The shipped default chooses the tile by a cost model (choose_pc_tile); it is checked against float64 below, and
# test_conv_pc_forced_tiles runs every (shape, width, K split) combination.
PC_SHAPES = [
    (64, 64, 512, 512, 'XL<32,2,8>  conv1_2 @512^2', True),
    (128, 128, 256, 256, 'XL<32,2,8>  conv2_2 @512^2', True),
    (512, 512, 128, 128, 'XL<32,2,8>  conv4_2 @1024^2', True),
    (64, 128, 362, 543, 'XL<32,2,8>  ragged 543x362', True),
    (256, 256, 128, 128, '<32,2,4>    conv3_2 @512^2', True),
    (512, 512, 200, 48, '<16,2,4>    narrow', True),
    (512, 512, 240, 40, '<8,2,4>     narrow', True),
    (512, 512, 64, 64, '<32,1,4>    conv4_2 @512^2', False),
    (512, 512, 32, 32, '<32,1,4>    conv5_1 @512^2 (split-K)', False),
    (256, 512, 45, 23, '<8,1,4>     ragged small', False),
]


def _pc_operands(cin, cout, h, w, dgrad):
    g = torch.Generator().manual_seed(cin * 13 + cout * 7 + h * 3 + w + (1000 if dgrad else 0))
    # post-ReLU-like operand with a per-channel spread of scales (exercises both fp16 planes and the bound)
    x = torch.randn((1, cout if dgrad else cin, h, w), generator=g)
    if not dgrad:
        x = x.relu() * torch.exp(torch.randn((1, cin, 1, 1), generator=g))
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    b = torch.randn((cout,), generator=g) * 0.1
    return x, wt, b


@pytest.mark.parametrize('cin,cout,h,w,variant,same_k', PC_SHAPES)
@pytest.mark.parametrize('dgrad', [False, True])
def test_conv_pc_variants_against_fp64_and_single_role_kernel(cin, cout, h, w, variant, same_k, dgrad):
    """The kernel that runs the trunk at every BASELINE config.  (1) fp16x3 result against float64 F.conv2d /
    conv_transpose2d (fp32-class tolerance); (2) against conv_split_kernel (ST_CONV_PC=0) on the same operands:
    bit-identical wherever both use the same K partition (st_conv_pc.hip header), else within fp32 rounding."""
    hip = _hip()
    x, wt, b = _pc_operands(cin, cout, h, w, dgrad)
    xd, wd, bd = x.to(DEV), wt.to(DEV), b.to(DEV)
    if dgrad:
        want = F.conv_transpose2d(x.double(), wt.double(), None, padding=1).float()
        run = lambda: hip.op_conv3x3_dgrad(xd, None, wd, 4)                      # noqa: E731
    else:
        want = F.conv2d(x.double(), wt.double(), b.double(), padding=1).relu().float()
        run = lambda: hip.op_conv3x3(xd, wd, bd, True, 4)                        # noqa: E731
    with hip.options(ST_CONV_PC=2, ST_CONV_PC_MODEL=0):   # force the producer / consumer kernel, round-1 tile rule
        got_pc = run()
    with hip.options(ST_CONV_PC=0):                       # single-role split kernel
        got_split = run()
    got_default = run()
    name = f'conv_pc {variant} {"dgrad" if dgrad else "fwd"} {cin}->{cout} {h}x{w}'
    _report(name + ' vs fp64', got_pc, want, 3e-6)
    _report(name + ' single-role vs fp64', got_split, want, 3e-6)
    ident = torch.equal(got_pc, got_split)
    dmax = float((got_pc - got_split).abs().max())
    print(f'[parity] {name}: pc vs single-role kernel bit-identical={ident} max_abs={dmax:.3e}')
    if same_k:
        assert ident, f'{name}: producer/consumer kernel differs from conv_split_kernel (max_abs {dmax:.3e})'
    else:
        assert rel_l2(got_pc.cpu(), got_split.cpu()) <= 1e-6
    # the shipped default (tile and K split chosen by the cost model)
    _report(name + ' default tile choice vs fp64', got_default, want, 3e-6)
    assert rel_l2(got_default.cpu(), got_pc.cpu()) <= 1e-6


@pytest.mark.parametrize('cin,cout,h,w', [(128, 128, 90, 90), (256, 512, 45, 45), (512, 512, 22, 23), (64, 64, 181, 181)])
def test_conv_pc_forced_tiles(cin, cout, h, w):
    """Every tile the cost model can choose - XL, 256- and 128-pixel tiles in the widths 32 / 16 / 8, K split 1 ... 16 -
    on ragged pyramid sizes (181, 90, 45, 22 are the 362^2 scale's levels): float64 reference, and bit-identity
    between all shapes that share a K split (the tile shape never changes the order of a pixel's K sum)."""
    hip = _hip()
    x, wt, b = _pc_operands(cin, cout, h, w, False)
    xd, wd, bd = x.to(DEV), wt.to(DEV), b.to(DEV)
    want = F.conv2d(x.double(), wt.double(), b.double(), padding=1).relu().float()
    by_ks = {}
    for shape in (1, 2, 3):
        for tw in ((32,) if shape == 1 else (32, 16, 8)):
            for ks in ((1,) if shape == 1 else (1, 2, 4, 8, 16)):
                if ks > 1 and ((cin // 16) % ks or cin // 16 // ks < 2):
                    continue
                with hip.options(ST_CONV_PC=2, ST_CONV_PC_SHAPE=shape, ST_CONV_PC_TW=tw, ST_CONV_PC_KSPLIT=ks):
                    got = hip.op_conv3x3(xd, wd, bd, True, 4)
                err = rel_l2(got.cpu(), want)
                assert err <= 3e-6, (shape, tw, ks, err)
                if ks in by_ks:
                    assert torch.equal(got, by_ks[ks][1]), f'shape {(shape, tw, ks)} differs from {by_ks[ks][0]}'
                else:
                    by_ks[ks] = ((shape, tw, ks), got)
    print(f'[parity] conv_pc forced tiles {cin}->{cout} {h}x{w}: K splits {sorted(by_ks)} each bit-identical across shapes')


@pytest.mark.parametrize('cin,cout,h,w', [(64, 64, 362, 362), (128, 128, 181, 181), (256, 256, 90, 91), (64, 128, 181, 362),
                                          (512, 512, 45, 45), (64, 64, 543, 362)])
@pytest.mark.parametrize('dgrad', [False, True])
def test_conv_pc_two_shape_cover(cin, cout, h, w, dgrad):
    """Layers whose tile count is just above a multiple of the CU count are covered by TWO launches (whole rounds of a
    large tile over the first rows, another tile over the rest; st_conv_pc.hip choose_pc_tile, ConvProblem::row_begin
    / row_end).  The seam must be invisible: bit-identical to the single-launch result (no K split on either side, and
    the tile shape never changes a pixel's summation order), and fp32-class against float64."""
    hip = _hip()
    x, wt, b = _pc_operands(cin, cout, h, w, dgrad)
    xd, wd, bd = x.to(DEV), wt.to(DEV), b.to(DEV)
    if dgrad:
        want = F.conv_transpose2d(x.double(), wt.double(), None, padding=1).float()
        run = lambda: hip.op_conv3x3_dgrad(xd, None, wd, 4)                      # noqa: E731
    else:
        want = F.conv2d(x.double(), wt.double(), b.double(), padding=1).relu().float()
        run = lambda: hip.op_conv3x3(xd, wd, bd, True, 4)                        # noqa: E731
    got = run()
    with hip.options(ST_CONV_PC_SPLIT=0, ST_CONV_PC_KSPLIT=1, ST_CONV_PC_SHAPE=2):
        single = run()
    name = f'conv_pc two-shape cover {"dgrad" if dgrad else "fwd"} {cin}->{cout} {h}x{w}'
    _report(name + ' vs fp64', got, want, 3e-6)
    ident = torch.equal(got, single)
    print(f'[parity] {name}: identical to one 256-pixel-tile launch without K split: {ident}')
    assert ident or rel_l2(got.cpu(), single.cpu()) <= 1e-6      # (the model may still choose a K split for small layers)


@pytest.mark.parametrize('case', ['outlier', 'tiny', 'huge', 'zeros', 'wide'])
def test_conv3x3_fp16x3_dynamic_range(case):
    """fp16x3 rescales both operands by a power of two taken from max |x|: operands far outside fp16's range, a
    single outlier 1e6 times the bulk, and an all-zero tensor must neither overflow nor lose the fp32-class bound."""
    g = torch.Generator().manual_seed(7)
    cin, cout, h, w = 64, 64, 24, 40
    x = torch.randn((1, cin, h, w), generator=g)
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (cin * 9)) ** 0.5
    if case == 'outlier':
        x[0, 3, 5, 7] = 1e6
    elif case == 'tiny':
        x, wt = x * 1e-20, wt * 1e-12
    elif case == 'huge':
        x, wt = x * 1e15, wt * 1e10
    elif case == 'zeros':
        x = torch.zeros_like(x)
    elif case == 'wide':
        x = x * torch.logspace(-6, 4, cin).reshape(1, cin, 1, 1)         # channels spanning 10 decades
    b = torch.zeros((cout,))
    want = F.conv2d(x.double(), wt.double(), None, padding=1)
    got = _hip().op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), False, 4).cpu()
    assert torch.isfinite(got).all()
    if case == 'zeros':
        assert float(got.abs().max()) == 0.0
        return
    # fp32 reference error for scale: the fp16x3 result must be fp32-class relative to the output norm
    ref32 = F.conv2d(x, wt, None, padding=1)
    e32 = rel_l2(ref32, want.float())
    e16 = rel_l2(got, want.float())
    print(f'[parity] fp16x3 dynamic range {case}: rel_l2={e16:.3e} (fp32 CPU conv: {e32:.3e})')
    assert e16 <= max(3e-6, 4 * e32)


def test_net_rejects_unknown_precision(vgg_weights):
    hip = _hip()
    with pytest.raises((ValueError, KeyError, RuntimeError)):
        hip.Net(vgg_weights, 'max', DEV, 'fp8')


@pytest.mark.parametrize('h,w', [(16, 16), (40, 48), (135, 181), (17, 300)])
def test_tv_loss_and_gradient(h, w):
    g = torch.Generator().manual_seed(h * 1000 + w)
    img = torch.rand((1, 3, h, w), generator=g)
    want_loss, want_grad = O.tv_loss_grad_closed_form(img)
    loss, grad = _hip().op_tv_loss(img.to(DEV))
    rel = abs(float(loss.cpu()) - float(want_loss)) / float(want_loss)
    print(f'[parity] tv loss {h}x{w}: rel={rel:.3e}')
    assert rel < 2e-6
    _report(f'tv grad {h}x{w}', grad, want_grad, 2e-6)


def test_sqrtm_known_answer_from_reference():
    g = load_golden('ns_kat')
    a = torch.from_numpy(g['a'])
    root = _hip().op_sqrtm_ns(a.to(DEV))
    _report('sqrtm_ns fwd vs reference KAT', root, torch.from_numpy(g['root']), 1e-5)
    ga = _hip().op_sqrtm_ns_backward(torch.from_numpy(g['root']).to(DEV), torch.from_numpy(g['gout']).to(DEV))
    _report('sqrtm_ns bwd vs reference KAT', ga, torch.from_numpy(g['ga']), 1e-4)


@pytest.mark.parametrize('n', [64, 128, 256, 512])
def test_sqrtm_against_oracle(n):
    g = torch.Generator().manual_seed(n)
    b = torch.randn((n, 2 * n), generator=g)
    a = (b @ b.t()) / (2 * n) + torch.eye(n) * 1e-2
    want = O.ns_sqrt(a, 12)
    root = _hip().op_sqrtm_ns(a.to(DEV))
    _report(f'sqrtm_ns fwd n={n}', root, want, 2e-5)
    gout = torch.randn((n, n), generator=g)
    want_b = O.ns_sqrt_bwd(want, gout, 12)
    got_b = _hip().op_sqrtm_ns_backward(want.to(DEV), gout.to(DEV))
    _report(f'sqrtm_ns bwd n={n}', got_b, want_b, 2e-4)


@pytest.mark.parametrize('n', [64, 128, 256, 512])
def test_sqrtm_first_step_shortcut_is_bit_identical(n):
    """sqrtm.py:21-24's first step multiplies by z = I twice (z @ y and t @ z): exact in any fp32 GEMM, so the library takes
    t_0 = (3I - y_0) / 2 from its prologue kernel and runs ONE product for that step.  Same bits as the literal form."""
    hip = _hip()
    g = torch.Generator().manual_seed(900 + n)
    for rank in (2 * n, n // 4):                         # well conditioned / rank deficient (+ eps I, as the covariances)
        b = torch.randn((n, rank), generator=g)
        a = ((b @ b.t()) / rank + torch.eye(n) * 1e-8).to(DEV)
        with hip.options(ST_NS_SKIP_IDENTITY=0):
            literal = hip.op_sqrtm_ns(a)
        with hip.options(ST_NS_SKIP_IDENTITY=1):
            short = hip.op_sqrtm_ns(a)
        assert torch.isfinite(short).all()
        assert torch.equal(literal, short), f'n={n} rank={rank}: {(literal - short).abs().max().item():.3e}'


@pytest.mark.parametrize('n', [64, 256, 512])
@pytest.mark.parametrize('kind', ['well_conditioned', 'rank_deficient'])
def test_sqrtm_diag_backward_and_fp16x3_chains(n, kind):
    """The chains as the plan runs them - forward NS-12 in fp32, the Lyapunov backward for grad = g I in its reduced
    form with fp16x3 products at n = 512 (csrc/st_nsgemm.hip) - and the optional fp16x3 forward chain
    (ST_NS_F16_FWD=1), against the oracle's fp32 recurrences (full commutator form) and against the library's own
    fp32 chains (ST_NS_F16=0, ST_NS_FULL_BACKWARD=1) on the same operands.
    'rank_deficient' is a covariance of n/4 samples + 1e-4 I, as relu5_1 sees at small scales (cond ~1e4...1e5,
    NS-12 not converged: the recurrences' rounding behaviour matters)."""
    hip = _hip()
    g = torch.Generator().manual_seed(n + len(kind))
    if kind == 'well_conditioned':
        b = torch.randn((n, 2 * n), generator=g)
        a = (b @ b.t()) / (2 * n) + torch.eye(n) * 1e-2
    else:
        b = torch.randn((n, n // 4), generator=g)
        a = (b @ b.t()) / (n // 4) + torch.eye(n) * 1e-4
    gd = -2.0 / n
    want = O.ns_sqrt(a, 12)
    want_b = O.ns_sqrt_bwd(want, torch.eye(n) * gd, 12)
    a64 = a.double()
    want64 = O.ns_sqrt(a64, 12)
    want_b64 = O.ns_sqrt_bwd(want64, torch.eye(n, dtype=torch.float64) * gd, 12)
    floor_f, floor_b = rel_l2(want, want64), rel_l2(want_b, want_b64)
    ad = a.to(DEV)
    with hip.options(ST_NS_F16_FWD=1):                      # both chains in fp16x3 (n = 512)
        root = hip.op_sqrtm_ns(ad)
        gb = hip.op_sqrtm_ns_backward_diag(root, gd)
    root_shipped = hip.op_sqrtm_ns(ad)                      # shipped default: fp32 forward chain ...
    gb_shipped = hip.op_sqrtm_ns_backward_diag(root_shipped, gd)      # ... fp16x3 backward chain
    with hip.options(ST_NS_F16=0, ST_NS_FULL_BACKWARD=1):
        root32 = hip.op_sqrtm_ns(ad)
        gb32 = hip.op_sqrtm_ns_backward_diag(root32, gd)
    ef, eb = rel_l2(root.cpu(), want), rel_l2(gb.cpu(), want_b)
    ef32, eb32 = rel_l2(root32.cpu(), want), rel_l2(gb32.cpu(), want_b)
    print(f'[parity] NS chains n={n} {kind}: fwd {ef:.2e} (fp32 chain {ef32:.2e}, cpu32-vs-fp64 {floor_f:.2e}); '
          f'diag bwd {eb:.2e} (fp32 full chain {eb32:.2e}, cpu32-vs-fp64 {floor_b:.2e}); '
          f'shipped vs fp32 chains: fwd {rel_l2(root.cpu(), root32.cpu()):.2e} bwd {rel_l2(gb.cpu(), gb32.cpu()):.2e}')
    assert torch.isfinite(root).all() and torch.isfinite(gb).all()
    assert ef <= max(2e-5, 3 * floor_f) and eb <= max(2e-4, 3 * floor_b)
    assert ef32 <= max(2e-5, 3 * floor_f) and eb32 <= max(2e-4, 3 * floor_b)
    assert torch.equal(root_shipped, root32), 'the shipped forward chain is the fp32 one'
    ebs = rel_l2(gb_shipped.cpu(), want_b)
    print(f'[parity] NS chains n={n} {kind}: shipped (fp32 forward, fp16x3 backward) diag bwd {ebs:.2e}')
    assert ebs <= max(2e-4, 3 * floor_b)


@pytest.mark.parametrize('cin,cout,h,w', [(64, 64, 32, 64), (64, 128, 48, 32), (128, 128, 40, 68), (256, 256, 16, 32), (128, 256, 35, 100),
                                          (512, 512, 16, 32)])
def test_fat_conv_kernel_is_bit_identical_to_the_producer_consumer_kernel(cin, cout, h, w):
    """csrc/st_conv_fat.hip (round 6; launch_conv_split takes it for conv3_x / conv4_x on 2048^2-class maps,
    conv_fat_preferred): four waves of (32 CB) co x 128 px register tiles that stage their own patches between their MFMAs.
    Same K order, LDS images and swizzle as conv_pc_kernel (nn.Conv2d at style_transfer.py:35,87 and its data gradient), so
    forced onto small shapes (ST_CONV_FAT=2) the results must equal the producer / consumer kernel's XL tile BIT FOR BIT -
    forward with bias + ReLU (64- and 128-channel tiles, ragged right / bottom edges, one to many workgroups) and the data
    gradient with the plan's epilogue (out = mask > 0 ? out + dgrad : 0)."""
    hip = _hip()
    g = torch.Generator().manual_seed(cin + w)
    x = torch.relu(torch.randn((1, cin, h, w), generator=g))
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    with hip.options(ST_CONV_FAT=0, ST_CONV_PC=2, ST_CONV_PC_SHAPE=1):
        xl = hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4)
    with hip.options(ST_CONV_FAT=2):
        fat = hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4)
    _report(f'fat conv fwd {cin}->{cout} {w}x{h} vs fp64', fat, want, 3e-6)
    assert torch.equal(fat, xl), 'forward differs from the producer / consumer kernel'
    gr = torch.randn((1, cout, h, w), generator=g)
    prev, mask = torch.randn((1, cin, h, w), generator=g), torch.randn((1, cin, h, w), generator=g)
    outs = []
    for opts in (dict(ST_CONV_FAT=0, ST_CONV_PC=2, ST_CONV_PC_SHAPE=1), dict(ST_CONV_FAT=2)):
        with hip.options(**opts):
            out = prev.clone().to(DEV)
            hip.op_conv3x3_strip_ex(gr.to(DEV), None, 0, 0, wt.to(DEV), None, False, True, out=out, out_mask=mask.to(DEV), precision=4)
            outs.append(out)
    wantd = F.conv_transpose2d(gr.double(), wt.double(), padding=1)
    wantd = torch.where(mask > 0, wantd + prev.double(), torch.zeros_like(wantd))
    _report(f'fat conv dgrad {cout}->{cin} {w}x{h} (+=, mask) vs fp64', outs[1], wantd, 3e-6)
    assert torch.equal(outs[0], outs[1]), 'data gradient differs from the producer / consumer kernel'


def test_fat_conv_kernel_in_the_closure_is_bit_identical(vgg_weights):
    """... and in a closure with the kernel forced onto every layer it applies to (ST_CONV_FAT=2: also the pooled layers'
    fused max pool + argmax codes and every data gradient's accumulate / mask epilogue): losses and image gradient equal,
    bit for bit, to the same closure on the producer / consumer kernel's XL tile (forced: the shipped selection splits K on
    maps this small, another summation order)."""
    hip = _hip()
    h, w = 128, 160
    g = torch.Generator().manual_seed(9)
    content, style, image = (torch.rand((1, 3, h, w), generator=g) for _ in range(3))

    def closure(**opts):
        with hip.options(**opts):
            net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
            plan = hip.Plan(net, h, w)
            plan.forward(content.to(DEV), 22)
            plan.set_content_target_from_forward()
            plan.forward(style.to(DEV), 29)
            for i, layer in enumerate(O.STYLE_LAYERS):
                plan.set_style_target(i, *plan.moments(layer))
            plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
            losses, grad = plan.loss_and_grad(image.to(DEV))
            return losses.clone().cpu(), grad.clone().cpu()
    l0, g0 = closure(ST_CONV_FAT=0, ST_CONV_PC=2, ST_CONV_PC_SHAPE=1)      # the XL tile everywhere it applies: no K split, pool kernels
    l1, g1 = closure(ST_CONV_FAT=2, ST_CONV_PC=2, ST_CONV_PC_SHAPE=1)      # ... with the fat kernel in front of it (fused pools + codes)
    rel = float((g1.double() - g0.double()).norm() / g0.double().norm())
    print(f'[parity] fat conv in the closure {w}x{h}: losses identical {torch.equal(l0, l1)}, gradient identical {torch.equal(g0, g1)} (rel_l2 {rel:.1e})')
    assert torch.equal(l0, l1) and torch.equal(g0, g1)


def _wino_or_skip():
    hip = _hip()
    if not hip.has_experiments():
        pytest.skip('libst_amd.so was built without --experiments (no Winograd kernel)')
    return hip


@pytest.mark.parametrize('cin,cout,h,w', [(64, 64, 32, 48), (128, 256, 48, 32), (512, 512, 16, 16), (64, 128, 21, 66),
                                          (128, 64, 7, 130), (512, 512, 8, 8), (64, 64, 135, 182)])
def test_winograd_conv_forward_against_float64(cin, cout, h, w):
    """csrc/st_conv_wino.hip (operator precision code 5; the plan does not use it - profiles/r06_winograd.md): the 3 x 3
    convolution + bias + ReLU (nn.Conv2d at style_transfer.py:35,87) as Winograd F(2 x 2, 3 x 3) on fp16x3 planes, interleaved
    transform / MFMA kernel of round 6.  Per-conv rel-L2 <= 1e-6 against float64 (measured 1.6 - 1.9e-7, the direct kernel:
    2.0 - 3.8e-7); zero padding at all four borders, ragged tiles (odd heights, widths that are no multiple of the tile), one to
    many workgroups, K split (the 8 x 8 and 16 x 16 shapes) and every tile-row width are inside these shapes."""
    hip = _wino_or_skip()
    g = torch.Generator().manual_seed(cin + h)
    x = torch.relu(torch.randn((1, cin, h, w), generator=g))
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    want = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=1))
    got = hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 5)
    shipped = hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 4)
    e5, e4 = rel_l2(got.cpu(), want), rel_l2(shipped.cpu(), want)
    print(f'[parity] Winograd conv {cin}->{cout} {w}x{h}: rel_l2 vs float64 {e5:.2e} (direct fp16x3 kernel {e4:.2e})')
    assert e5 <= 1e-6
    for tx in (8, 16, 32):
        with hip.options(ST_WINO_TX=tx):
            again = hip.op_conv3x3(x.to(DEV), wt.to(DEV), b.to(DEV), True, 5)
        assert rel_l2(again.cpu(), want) <= 1e-6, tx


@pytest.mark.parametrize('cin,cout,h,w', [(64, 128, 24, 40), (256, 256, 16, 16), (512, 256, 9, 34)])
def test_winograd_conv_data_gradient_with_the_plan_epilogue(cin, cout, h, w):
    """The data gradient (the same kernel on the rotated, role-swapped weight planes) with the epilogue of the plan's backward
    launches: out = mask > 0 ? out + dgrad : 0 (accumulate into the tap's gradient, threshold_backward of the layer below)."""
    hip = _wino_or_skip()
    gen = torch.Generator().manual_seed(cout + w)
    g = torch.randn((1, cout, h, w), generator=gen)
    wt = torch.randn((cout, cin, 3, 3), generator=gen) * (2.0 / (9 * cin)) ** 0.5
    prev = torch.randn((1, cin, h, w), generator=gen)
    mask = torch.randn((1, cin, h, w), generator=gen)
    plain = F.conv_transpose2d(g.double(), wt.double(), padding=1)
    want = torch.where(mask > 0, plain + prev.double(), torch.zeros_like(plain))
    out = prev.clone().to(DEV)
    hip.op_conv3x3_strip_ex(g.to(DEV), None, 0, 0, wt.to(DEV), None, False, True, out=out, out_mask=mask.to(DEV), precision=5)
    got_plain = hip.op_conv3x3_strip_ex(g.to(DEV), None, 0, 0, wt.to(DEV), None, False, True, precision=5)
    e, ep = rel_l2(out.cpu(), want), rel_l2(got_plain.cpu(), plain)
    print(f'[parity] Winograd dgrad {cout}->{cin} {w}x{h}: rel_l2 vs float64 {e:.2e} (+=, mask), {ep:.2e} (plain)')
    assert e <= 1e-6 and ep <= 1e-6
    assert bool((out.cpu()[mask <= 0] == 0).all())


@pytest.mark.parametrize('n', [64, 128, 256, 512])
@pytest.mark.parametrize('kind', ['well_conditioned', 'rank_deficient'])
def test_persistent_chain_kernel_against_the_launch_per_product_chains(n, kind):
    """csrc/st_nschain.hip (round 5, ST_NS_CHAIN bit 3 routes the standalone operators through it; off by default - see
    profiles/r05_ns_chain.md): both recurrences as ONE persistent launch with fence-free grid barriers.
    'every tile' (ST_NS_CHAIN_SYM=0) follows the reference's products exactly like the launch-per-product kernels - the
    forward chain at n = 512 has the same K split and must agree to rounding; the backward chain is fp32 throughout.
    'symmetric tile pairs' (ST_NS_CHAIN_SYM=15) computes 136 of 256 tiles and mirrors: exactly symmetric iterates - and
    measurably LESS accurate on rank-deficient input (the reason it does not ship): asserted only against a loose bound,
    its distance to float64 is printed next to the others."""
    hip = _hip()
    if not hip.has_experiments():
        pytest.skip('libst_amd.so was built without --experiments (no persistent chain kernel)')
    g = torch.Generator().manual_seed(n + len(kind))
    if kind == 'well_conditioned':
        b = torch.randn((n, 2 * n), generator=g)
        a = (b @ b.t()) / (2 * n) + torch.eye(n) * 1e-2
    else:
        b = torch.randn((n, n // 4), generator=g)
        a = (b @ b.t()) / (n // 4) + torch.eye(n) * 1e-4
    gd = -2.0 / n
    want64 = O.ns_sqrt(a.double(), 12)
    wantb64 = O.ns_sqrt_bwd(want64, torch.eye(n, dtype=torch.float64) * gd, 12)
    cpu32 = O.ns_sqrt(a, 12)
    floor_f = rel_l2(cpu32, want64)
    floor_b = rel_l2(O.ns_sqrt_bwd(cpu32, torch.eye(n) * gd, 12), wantb64)
    ad = a.to(DEV)
    with hip.options(ST_NS_CHAIN=0):
        root0 = hip.op_sqrtm_ns(ad)
        gb0 = hip.op_sqrtm_ns_backward_diag(root0, gd)
    out = {}
    for label, sym in (('every tile', 0), ('symmetric tile pairs', 15)):
        with hip.options(ST_NS_CHAIN=8, ST_NS_CHAIN_SYM=sym):
            root = hip.op_sqrtm_ns(ad)
            gb = hip.op_sqrtm_ns_backward_diag(root, gd)
        assert torch.isfinite(root).all() and torch.isfinite(gb).all()
        out[label] = (rel_l2(root.cpu(), want64), rel_l2(gb.cpu(), wantb64), rel_l2(root.cpu(), root0.cpu()),
                      float((root - root.t()).abs().max()))
        # the same products with the operands through the L2: behind an acquire per barrier (1), or - every iterate in a matrix
        # of its own, staged by LDS-DMA - without one (2).  Same arithmetic in the same order: the same bits, also when the
        # workspace's lines are warm in every L2 from the call before
        for l2 in (1, 2):
            for rep in range(3):
                with hip.options(ST_NS_CHAIN=8, ST_NS_CHAIN_SYM=sym, ST_NS_CHAIN_L2=l2):
                    root_l2 = hip.op_sqrtm_ns(ad)
                    gb_l2 = hip.op_sqrtm_ns_backward_diag(root_l2, gd)
                assert torch.equal(root_l2, root) and torch.equal(gb_l2, gb), (label, l2, rep)
    e0f, e0b = rel_l2(root0.cpu(), want64), rel_l2(gb0.cpu(), wantb64)
    print(f'[parity] persistent NS chain n={n} {kind}: vs float64 fwd / bwd - reference fp32 arithmetic {floor_f:.2e} / {floor_b:.2e}, '
          f'launch per product {e0f:.2e} / {e0b:.2e}, every tile {out["every tile"][0]:.2e} / {out["every tile"][1]:.2e} '
          f'(vs launch per product {out["every tile"][2]:.2e}), symmetric {out["symmetric tile pairs"][0]:.2e} / '
          f'{out["symmetric tile pairs"][1]:.2e} (max |R - R^T| {out["symmetric tile pairs"][3]:.1e})')
    ef, eb, same, _ = out['every tile']
    assert ef <= max(2e-5, 3 * floor_f) and eb <= max(2e-4, 3 * floor_b)
    if n == 512:
        assert same <= 1e-6, 'n = 512: the same K split as gemm_staged_kernel<512, 4> - rounding-level agreement'
    sf, sb, _, asym = out['symmetric tile pairs']
    assert asym == 0.0 and sf <= 1e-3 and sb <= 1e-3


@pytest.mark.parametrize('l2', [0, 2])
def test_relu5_1_head_on_the_persistent_chain_kernel(l2, vgg_weights):
    """ST_NS_CHAIN=4 (every tile): relu5_1's two recurrences of the closure as one persistent launch - not shipped (no gain
    beside the other heads' chains, profiles/r05_ns_chain.md), kept working.  The closure must agree with the launch-per-product
    form to the chains' rounding, and - the plan's workspace is the same memory in every closure, its lines warm in the L2s -
    repeat bit for bit (a stale operand line would show as a run that differs)."""
    hip = _hip()
    if not hip.has_experiments():
        pytest.skip('libst_amd.so was built without --experiments (no persistent chain kernel)')
    h = w = 256
    g = torch.Generator().manual_seed(5)
    content, style = torch.rand((1, 3, h, w), generator=g), torch.rand((1, 3, h, w), generator=g)

    def closures(**opts):
        with hip.options(**opts):
            net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
            plan = hip.Plan(net, h, w)
            plan.forward(content.to(DEV), 22)
            plan.set_content_target_from_forward()
            plan.forward(style.to(DEV), 29)
            for i, layer in enumerate(O.STYLE_LAYERS):
                plan.set_style_target(i, *plan.moments(layer))
            plan.set_loss_weights(0.015, O.STYLE_LAYER_WEIGHTS, 2.0)
            image = (0.5 * content + 0.5 * style).to(DEV)
            runs = []
            for _ in range(6):
                losses, grad = plan.loss_and_grad(image)
                runs.append((losses.clone(), grad.clone()))
            torch.cuda.synchronize()
        return runs

    base = closures(ST_NS_CHAIN=0)
    runs = closures(ST_NS_CHAIN=4, ST_NS_CHAIN_SYM=0, ST_NS_CHAIN_L2=l2)
    for losses, grad in runs[1:]:
        assert torch.equal(losses, runs[0][0]) and torch.equal(grad, runs[0][1])
    rel = ((runs[0][0] - base[0][0]).abs() / base[0][0].abs().clamp_min(1e-30)).max().item()
    err = rel_l2(runs[0][1].cpu(), base[0][1].cpu())
    print(f'[parity] relu5_1 head on the persistent chain kernel (operands {"through the L2, LDS-DMA" if l2 else "from the memory side"}): '
          f'loss terms vs launch per product {rel:.2e}, gradient rel_l2 {err:.2e}')
    assert rel < 5e-5 and err < 2e-4


@pytest.mark.parametrize('h,w', [(64, 64), (40, 48), (128, 16), (96, 260)])
def test_conv1_1_four_pixel_kernel_is_bit_identical(h, w, vgg_weights):
    """conv_first_fwd4_kernel (four pixels per thread, packed fp32 FMAs, 16-byte stores; st_conv_first.hip) against the
    one-pixel kernel (ST_CONV1_WIDE=0): same FMA order per output, so relu1_1 must agree bit for bit - borders
    (replicate padding) and the last group of every row included."""
    hip = _hip()
    g = torch.Generator().manual_seed(h + w)
    image = torch.rand((1, 3, h, w), generator=g).to(DEV)
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    plan = hip.Plan(net, h, w)
    plan.forward(image, 1)
    wide = plan.feature(1).clone()
    with hip.options(ST_CONV1_WIDE=0):
        plan.forward(image, 1)
        narrow = plan.feature(1).clone()
    assert torch.equal(wide, narrow), f'max abs diff {float((wide - narrow).abs().max()):.3e}'


@pytest.mark.parametrize('pooling', ['max', 'average', 'l2'])
@pytest.mark.parametrize('h,w', [(40, 48), (135, 181)])
@pytest.mark.parametrize('precision', ['fp32', 'fp16x3'])
def test_trunk_forward_taps(pooling, h, w, precision, vgg_weights):
    hip = _hip()
    g = torch.Generator().manual_seed(h + w)
    img = torch.rand((1, 3, h, w), generator=g)
    layers = O.STYLE_LAYERS + O.CONTENT_LAYERS + [4, 9, 18, 27, 3, 26]
    want = O.vgg_features(img, vgg_weights, layers, pooling)
    net = hip.Net(vgg_weights, pooling, DEV, precision)
    plan = hip.Plan(net, h, w)
    plan.forward(img.to(DEV), 29)
    for layer in sorted(layers):
        _report(f'features[{layer}] {pooling} {h}x{w} {precision}', plan.feature(layer), want[layer], 5e-6)


def test_trunk_forward_taps_512_shipped_arithmetic(vgg_weights):
    """Every tap of the 512x512 trunk (BASELINE configs[1]) in fp16x3 - the layers run the XL / 256- / 128-pixel
    producer-consumer tiles - against the oracle's fp32 CPU features."""
    hip = _hip()
    g = torch.Generator().manual_seed(512)
    img = torch.rand((1, 3, 512, 512), generator=g)
    layers = O.STYLE_LAYERS + O.CONTENT_LAYERS + [3, 8, 13, 15, 17, 24, 26]
    want = O.vgg_features(img, vgg_weights, layers, 'max')
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    plan = hip.Plan(net, 512, 512)
    plan.forward(img.to(DEV), 29)
    for layer in sorted(layers):
        _report(f'features[{layer}] 512x512 fp16x3', plan.feature(layer), want[layer], 5e-6)


@pytest.mark.parametrize('h,w', [(512, 512), (256, 256), (364, 544), (362, 368), (1024, 128), (128, 132)])
def test_max_pool_fused_into_the_conv_epilogue(h, w, vgg_weights):
    """MaxPool2d(2) (reference style_transfer.py:21) is written by the epilogue of the conv that feeds it wherever
    the chosen tile allows (st_conv_pc.hip: conv_pc_fuses_pool): every pooled map AND every tap must be bit-identical
    to the run with the separate pool kernel (ST_CONV_POOL_FUSE=0) - odd pooled heights, ragged widths, layers covered
    by two tile shapes included."""
    hip = _hip()
    g = torch.Generator().manual_seed(h * 7 + w)
    img = torch.rand((1, 3, h, w), generator=g).to(DEV)
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    layers = [4, 9, 18, 27, 6, 11, 20, 22, 29]
    feats = {}
    for fuse in (1, 0):
        with hip.options(ST_CONV_POOL_FUSE=fuse):
            plan = hip.Plan(net, h, w)
            plan.forward(img, 29)
            feats[fuse] = {layer: plan.feature(layer).clone() for layer in layers}
        del plan
    for layer in layers:
        same = torch.equal(feats[1][layer], feats[0][layer])
        assert same, f'features[{layer}] {h}x{w}: fused pooling differs (max_abs ' \
                     f'{float((feats[1][layer] - feats[0][layer]).abs().max()):.3e})'
    want = O.vgg_features(img.cpu(), vgg_weights, [4, 9], 'max')
    for layer in (4, 9):
        _report(f'features[{layer}] {h}x{w} fused pool vs oracle', feats[1][layer], want[layer], 5e-6)


@pytest.mark.parametrize('precision,wide', [('fp32', 1), ('fp16x3', 1), ('fp16x3', 2)])
@pytest.mark.parametrize('h,w', [(40, 48), (135, 181), (128, 128)])
def test_moments_of_taps(h, w, precision, wide, vgg_weights):
    """fp32: exact fp32 MFMA Gram kernel; fp16x3: the split-precision Gram kernel (scaled fp16 planes, bound
    from the producing convolution); wide = 2 forces its 128 x 128-tile form (shipped for C >= 256 taps of >= 65 536
    pixels) onto every C % 128 == 0 tap.  All against the oracle's moments of the oracle's features, and against
    float64 moments of the plan's OWN features (isolates the Gram kernel from the trunk's arithmetic)."""
    hip = _hip()
    g = torch.Generator().manual_seed(h * 3 + w)
    img = torch.rand((1, 3, h, w), generator=g)
    want = O.vgg_features(img, vgg_weights, O.STYLE_LAYERS)
    net = hip.Net(vgg_weights, 'max', DEV, precision)
    plan = hip.Plan(net, h, w)
    plan.forward(img.to(DEV), 29)
    for layer in O.STYLE_LAYERS:
        with hip.options(ST_GRAM_WIDE=wide):
            mean, srm = plan.moments(layer)
        wm, ws = O.feature_moments(want[layer])
        _report(f'mean features[{layer}] {h}x{w} {precision}', mean, wm, 5e-6)
        _report(f'srm features[{layer}] {h}x{w} {precision}', srm, ws, 5e-6)
        assert torch.equal(srm, srm.t()), 'second raw moment must be exactly symmetric'
        f = plan.feature(layer)[0].cpu().double().flatten(1)
        own = (f @ f.t()) / f.shape[1]
        _report(f'srm of own features[{layer}] {h}x{w} {precision}', srm, own.float(), 1e-6)
        _report(f'mean of own features[{layer}] {h}x{w} {precision}', mean, f.mean(1).float(), 1e-6)


@pytest.mark.parametrize('c,npix', [(64, 512 * 300), (128, 256 * 260), (64, 224 * 224), (256, 181 * 135 + 3),
                                    (64, 40000 + 1), (512, 4096), (128, 37),
                                    # the 128-output-channel tile (st_conv1x1.hip CM = 4): >= 512 such workgroups
                                    (128, 1024 * 160), (256, 512 * 260 + 3), (512, 256 * 132)])
@pytest.mark.parametrize('precision', [0, 4])
def test_conv1x1_head_gradient_step(c, npix, precision):
    """dF = S F + b 1^T (the style heads' gradient step) against float64.  Large taps run the fp16x3 kernel
    (256- and 128-pixel tiles, ragged / unaligned pixel counts); small ones fall back to the fp32 split-K
    kernel inside the same launcher.  Operands with a wide dynamic range, as the real S has."""
    hip = _hip()
    g = torch.Generator().manual_seed(c + npix)
    x = torch.relu(torch.randn((c, npix), generator=g)) * torch.exp(2 * torch.randn((c, 1), generator=g))
    s = torch.randn((c, c), generator=g) * torch.exp(3 * torch.randn((c, c), generator=g)) * 1e-7
    s = s + s.t()
    b = torch.randn((c,), generator=g) * 1e-6
    want = (s.double() @ x.double() + b.double()[:, None]).float()
    got = hip.op_conv1x1(x.to(DEV), s.to(DEV), b.to(DEV), precision)
    _report(f'conv1x1 C={c} npix={npix} precision={precision}', got, want, 2e-6 if precision else 1e-6)


def test_plan_rejects_small_inputs(vgg_weights):
    hip = _hip()
    net = hip.Net(vgg_weights, 'max', DEV)
    with pytest.raises(ValueError):
        hip.Plan(net, 12, 40)


@pytest.mark.parametrize('h,w', [(40, 48), (136, 184), (512, 512), (724, 1024)])
def test_relu1_1_moments_out_of_the_conv1_1_launch(h, w, vgg_weights):
    """Round 4, opt-in (ST_CONV1_GRAM=1; measured a wash end to end, see st_conv_first.hip): conv1_1's launch also leaves
    relu1_1's partial Gram matrix and row sums (conv_first_fwd_gram_kernel: per-block power-of-two scales, fp16x3 products on
    the idle matrix pipe), which saves the Gram kernel's pass over the largest tap.  Here through st_plan_forward + st_plan_moments (ST_CONV1_GRAM_IN_FORWARD=1):
    the feature map must be bit-identical to the four-pixel kernel's, the moments exactly symmetric and as close to the
    float64 moments of that map as the stand-alone kernel's are."""
    hip = _hip()
    g = torch.Generator().manual_seed(h * 5 + w)
    img = torch.rand((1, 3, h, w), generator=g)
    img[:, :, : h // 3] *= 0.02                           # a dark band: blocks whose scale differs by several binades
    net = hip.Net(vgg_weights, 'max', DEV, 'fp16x3')
    plan = hip.Plan(net, h, w)
    plan.forward(img.to(DEV), 1)
    mean0, srm0 = plan.moments(1)                         # (allocates the head's workspace; two-kernel form)
    feat0 = plan.feature(1)
    with hip.options(ST_CONV1_GRAM_IN_FORWARD=1, ST_CONV1_GRAM=1):
        plan.forward(img.to(DEV), 1)
        mean1, srm1 = plan.moments(1)
    feat1 = plan.feature(1)
    assert torch.equal(feat0, feat1), 'the fused launch must write the same relu1_1 map'
    assert torch.equal(srm1, srm1.t()), 'second raw moment must be exactly symmetric'
    f = feat1[0].cpu().double().flatten(1)
    own_srm, own_mean = ((f @ f.t()) / f.shape[1]).float(), f.mean(1).float()
    e_fused, e_plain = rel_l2(srm1.cpu(), own_srm), rel_l2(srm0.cpu(), own_srm)
    m_fused, m_plain = rel_l2(mean1.cpu(), own_mean), rel_l2(mean0.cpu(), own_mean)
    print(f'[kernels] relu1_1 moments {h}x{w}: srm vs float64 fused {e_fused:.2e} / stand-alone {e_plain:.2e}; mean '
          f'{m_fused:.2e} / {m_plain:.2e}')
    assert e_fused <= 1e-6 and m_fused <= 1e-6
    assert e_fused <= 3 * e_plain + 1e-7


@pytest.mark.parametrize('size', [128, 512])
def test_closure_with_and_without_the_fused_relu1_1_moments(size, vgg_weights):
    """The closure with relu1_1's moments out of conv1_1's launch (ST_CONV1_GRAM=1) against the shipped two-kernel form:
    every term but relu1_1's identical, relu1_1's and the gradient within the noise the non-converged Newton-Schulz chain
    makes of a 1e-7 difference in its input."""
    hip = _hip()
    from test_hot_path_gpu import _build_plan, _smooth
    content, style, image = _smooth(91, size, size), _smooth(92, size, size), _smooth(93, size, size)
    net, plan = _build_plan(hip, vgg_weights, content, [style], [1.0], precision='fp16x3')
    with hip.options(ST_CONV1_GRAM=1):
        l1, g1 = plan.loss_and_grad(image.to(DEV))
        l1, g1 = l1.clone(), g1.clone()
    with hip.options(ST_CONV1_GRAM=0):
        l0, g0 = plan.loss_and_grad(image.to(DEV))
    assert not torch.equal(l1, l0), 'the opt-in path did not run'

    rel = ((l1 - l0).abs() / l0.abs()).cpu()
    print(f'[kernels] fused relu1_1 moments, closure {size}: term diffs {[f"{float(x):.1e}" for x in rel]}, gradient rel-L2 '
          f'{rel_l2(g1.cpu(), g0.cpu()):.2e}')
    for k in (0, 2, 3, 4, 5, 6):
        assert float(rel[k]) == 0.0, k
    assert float(rel[1]) <= 5e-5 and rel_l2(g1.cpu(), g0.cpu()) <= 2e-4
