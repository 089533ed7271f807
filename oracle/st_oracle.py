"""CPU ORACLE for the style-transfer hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, the algorithm of the per-iteration hot path of
crowsonkb/style-transfer-pytorch so that the HIP kernels can be checked against it.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
shipped package (``style-transfer-pytorch_amd/``) never does and fails loudly without its HIP library.

Why torch-on-CPU and not plain C: the path is dense fp32 contraction work (95 GFLOP per iteration
at 256x256); the reference itself is a thin layer over ``torch`` CPU operators, so
the restatement calls the same operator library for conv/matmul and spells out everything the
reference implements itself (losses, Newton-Schulz sqrtm and its Lyapunov backward, TV stencil,
Adam, EMA, the scale transition).  It is written functionally (no nn.Module graph) and takes a
``dtype`` so the same code doubles as an fp64 cross-check.

PARITY PINNING: the reference ships no tests or golden vectors.  This oracle is
pinned instead against outputs of the *unmodified reference itself*, imported in the build
container by ``tests/golden/make_golden.py`` (torchvision shim + seeded synthetic VGG weights);
the resulting vectors are committed under ``tests/golden/`` and ``tests/test_oracle_golden.py``
checks this file against them.

Reference citations are ``file:line`` into ``/root/reference/style_transfer/``.
"""

import math

import torch
from torch.nn import functional as F

# ----------------------------------------------------------------------------------------------
# network description (torchvision vgg19 cfg "E", truncated at features[29]); style_transfer.py:35
# ----------------------------------------------------------------------------------------------
_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512]
MEAN = (0.485, 0.456, 0.406)       # style_transfer.py:30
STD = (0.229, 0.224, 0.225)        # style_transfer.py:31
CONTENT_LAYERS = [22]              # style_transfer.py:316
STYLE_LAYERS = [1, 6, 11, 20, 29]  # style_transfer.py:317
STYLE_LAYER_WEIGHTS = [w / 341 for w in (256, 64, 16, 4, 1)]   # style_transfer.py:320-322
POOL_SCALE = {'max': 1.0, 'average': 2.0, 'l2': 0.78}          # style_transfer.py:22


def layer_program():
    """[(features_index, op, conv_number)] for indices 0..29."""
    prog, idx, conv_no = [], 0, 0
    for item in _CFG:
        if item == 'M':
            prog.append((idx, 'pool', None))
            idx += 1
        else:
            prog.append((idx, 'conv', conv_no))
            prog.append((idx + 1, 'relu', None))
            idx += 2
            conv_no += 1
    return prog


def vgg_features(image, weights, layers, pooling='max'):
    """VGGFeatures.forward (style_transfer.py:78-90) incl. the 'input' tap and the size check.

    image: [1,3,H,W] in [0,1] (un-normalised).  weights: 13 (w, b) pairs.  Returns {tap: tensor}.
    conv1_1 uses replicate padding (:39,52-59), every other conv zero padding.
    """
    layers = sorted(set(layers))
    h, w = image.shape[2:4]
    need = 1
    for bound in (4, 9, 18, 27, 36):                     # _get_min_size, :61-69
        if max(layers) < bound:
            break
        need *= 2
    if min(h, w) < need:
        raise ValueError(f'Input is {h}x{w} but must be at least {need}x{need}')
    feats = {'input': image}
    mean = image.new_tensor(MEAN).view(1, 3, 1, 1)
    std = image.new_tensor(STD).view(1, 3, 1, 1)
    x = (image - mean) / std                              # transforms.Normalize, :30-31,85
    for idx, op, conv_no in layer_program():
        if idx > max(layers):
            break
        if op == 'conv':
            wgt, bias = weights[conv_no]
            wgt, bias = wgt.to(x.dtype), bias.to(x.dtype)
            if conv_no == 0:
                x = F.conv2d(F.pad(x, (1, 1, 1, 1), mode='replicate'), wgt, bias)
            else:
                x = F.conv2d(x, wgt, bias, padding=1)
        elif op == 'relu':
            x = torch.relu(x)
        else:
            if pooling == 'max':
                x = F.max_pool2d(x, 2)
            elif pooling == 'average':                    # Scale(AvgPool2d(2), 2.0), :41-46
                x = F.avg_pool2d(x, 2) * POOL_SCALE['average']
            elif pooling == 'l2':                         # Scale(LPPool2d(2, 2), 0.78)
                x = F.lp_pool2d(x, 2, 2) * POOL_SCALE['l2']
            else:
                raise ValueError(pooling)
        if idx in layers:
            feats[idx] = x
    return feats


# ----------------------------------------------------------------------------------------------
# Newton-Schulz matrix square root and its Lyapunov backward; sqrtm.py:9-55
# ----------------------------------------------------------------------------------------------
def ns_sqrt(mat, iters=12):
    """sqrtm_ns (sqrtm.py:9-25): Frobenius-normalised coupled Newton-Schulz, fixed iteration count."""
    n = mat.shape[-1]
    fro = mat.pow(2).sum().sqrt()
    y = mat / fro
    eye3 = torch.eye(n, dtype=mat.dtype) * 3
    z = torch.eye(n, dtype=mat.dtype)
    for _ in range(iters):
        t = (eye3 - z @ y) / 2
        y = y @ t
        z = t @ z
    return y * fro.sqrt()


def ns_sqrt_bwd(root, grad, iters=12):
    """_MatrixSquareRootNSLyap.backward (sqrtm.py:36-47): iterative Lyapunov solve."""
    n = root.shape[-1]
    fro = root.pow(2).sum().sqrt()
    a = root / fro
    eye3 = torch.eye(n, dtype=root.dtype) * 3
    q = grad / fro
    for i in range(iters):
        e = eye3 - a @ a
        q = (q @ e - a.t() @ (a.t() @ q - q @ a)) / 2
        if i < iters - 1:
            a = a @ e / 2
    return q / 2


class _NSRoot(torch.autograd.Function):
    """Glue so that autograd uses the reference's custom backward (sqrtm.py:28-47)."""

    @staticmethod
    def forward(ctx, mat):
        root = ns_sqrt(mat, 12)
        ctx.save_for_backward(root)
        return root

    @staticmethod
    def backward(ctx, grad):
        (root,) = ctx.saved_tensors
        return ns_sqrt_bwd(root, grad, 12)


# ----------------------------------------------------------------------------------------------
# loss heads; style_transfer.py:119-126 (content), :149-181 (style), :184-195 (TV)
# ----------------------------------------------------------------------------------------------
def feature_moments(feat):
    """StyleLossW2.get_target (:162-168): per-channel mean and second raw moment of a [1,C,h,w] tap."""
    c, h, w = feat.shape[1:]
    mat = feat.reshape(c, h * w)
    mean = feat.mean(dim=(2, 3)).reshape(c)
    srm = (mat @ mat.t()) / (h * w)
    return mean, srm


def style_target(mean, srm, eps=1e-4):
    """StyleLossW2.__init__ (:152-160): (mean, cov, cov_sqrt) from blended (mean, srm)."""
    cov = srm - torch.outer(mean, mean) + torch.eye(srm.shape[0], dtype=srm.dtype) * eps
    return mean, cov, ns_sqrt(cov, 12)


def style_w2(feat, target, eps=1e-4, differentiable=True):
    """StyleLossW2.forward (:175-181)."""
    t_mean, t_cov, t_root = target
    mean, srm = feature_moments(feat)
    cov = srm - torch.outer(mean, mean) + torch.eye(srm.shape[0], dtype=srm.dtype) * eps
    mean_term = torch.mean((mean - t_mean) ** 2)
    inner = t_root @ cov @ t_root
    root = _NSRoot.apply(inner) if differentiable else ns_sqrt(inner, 12)
    cov_term = torch.diagonal(t_cov + cov - 2 * root).mean()
    return mean_term + cov_term


def content_mse(feat, target):
    """ContentLossMSE.forward (:119-126): nn.MSELoss with mean reduction."""
    return torch.mean((feat - target) ** 2)


def tv_loss(image):
    """TVLoss.forward (:187-195): nine-point L2 total variation on the replicate-padded image."""
    p = F.pad(image, (1, 1, 1, 1), mode='replicate')
    core = p[..., 1:-1, 1:-1]
    dx = (p[..., 1:-1, 2:] - core).pow(2).mean() / 3
    dy = (p[..., 2:, 1:-1] - core).pow(2).mean() / 3
    d_se = (p[..., 1:, 1:] - p[..., :-1, :-1]).pow(2).mean() / 12
    d_sw = (p[..., 1:, :-1] - p[..., :-1, 1:]).pow(2).mean() / 12
    return 2 * (dx + dy + d_se + d_sw)


def tv_loss_grad_closed_form(image):
    """Hand-derived gradient of tv_loss - used to check the HIP stencil."""
    img = image.detach().clone().requires_grad_(True)
    loss = tv_loss(img)
    loss.backward()
    return loss.detach(), img.grad.detach()


# ----------------------------------------------------------------------------------------------
# targets (cold path per scale) and the seven-term objective; style_transfer.py:416-455
# ----------------------------------------------------------------------------------------------
class Targets:
    """Everything ``stylize`` precomputes per scale: relu4_2 of the content image and the blended
    (mean, cov, cov_sqrt) of the style images for the five style taps (:425-453)."""

    def __init__(self, content_feat, style):
        self.content_feat = content_feat          # [1,512,h,w]
        self.style = style                        # {layer: (mean, cov, cov_sqrt)}


def build_targets(content, styles, weights, style_image_weights=None, pooling='max'):
    """content: [1,3,H,W]; styles: list of [1,3,h,w].  Follows :425-453."""
    with torch.no_grad():
        cfeat = vgg_features(content, weights, CONTENT_LAYERS, pooling)[22]
        if style_image_weights is None:
            style_image_weights = [1 / len(styles)] * len(styles)
        acc = {}
        for simg, sw in zip(styles, style_image_weights):
            feats = vgg_features(simg, weights, STYLE_LAYERS, pooling)
            for layer in STYLE_LAYERS:
                mean, srm = feature_moments(feats[layer])
                mean, srm = mean * sw, srm * sw
                if layer not in acc:
                    acc[layer] = [mean, srm]
                else:
                    acc[layer][0] += mean
                    acc[layer][1] += srm
        style = {layer: style_target(*acc[layer]) for layer in STYLE_LAYERS}
    return Targets(cfeat, style)


TERM_NAMES = ['content', 'style_relu1_1', 'style_relu2_1', 'style_relu3_1', 'style_relu4_1',
              'style_relu5_1', 'tv']


def loss_terms(image, weights, targets, content_weight=0.015, tv_weight=2.0, pooling='max'):
    """SumLoss over [content, 5 x style, tv] with their Scale factors (:198-234,376,427-455).

    Returns (list of 7 *weighted* terms in SumLoss order, total)."""
    feats = vgg_features(image, weights, STYLE_LAYERS + CONTENT_LAYERS, pooling)
    terms = [content_mse(feats[22], targets.content_feat) * (content_weight / len(CONTENT_LAYERS))]
    for layer, lw in zip(STYLE_LAYERS, STYLE_LAYER_WEIGHTS):
        terms.append(style_w2(feats[layer], targets.style[layer]) * lw)
    terms.append(tv_loss(feats['input']) * tv_weight)
    total = terms[0]
    for t in terms[1:]:
        total = total + t
    return terms, total


def loss_and_grad(image, weights, targets, **kw):
    """The closure of the hot loop (:472-476): forward, loss, backward to the pixels."""
    img = image.detach().clone().requires_grad_(True)
    terms, total = loss_terms(img, weights, targets, **kw)
    total.backward()
    return [float(t.detach()) for t in terms], float(total.detach()), img.grad.detach()


# ----------------------------------------------------------------------------------------------
# optimiser update, box constraint, iterate averaging; style_transfer.py:237-253,457-486
# ----------------------------------------------------------------------------------------------
class State:
    """image + Adam moments + step + EMA (value, accum).  All tensors [1,3,H,W]."""

    def __init__(self, image, avg_decay=0.99):
        self.image = image.detach().clone()
        self.exp_avg = torch.zeros_like(self.image)
        self.exp_avg_sq = torch.zeros_like(self.image)
        self.step = 0
        self.avg_decay = avg_decay
        self.new_average()

    def new_average(self):
        """EMA.__init__ (:240-245): zero value, accum=1, then one update with the current image."""
        self.ema_value = torch.zeros_like(self.image)
        self.ema_accum = torch.tensor(1.0, dtype=self.image.dtype)
        ema_update(self)

    def average(self):
        """EMA.get (:247-248)."""
        return self.ema_value / (1 - self.ema_accum)


def ema_update(state):
    """EMA.update (:250-253).  decay is held as a tensor of the image dtype, so (1 - decay) is
    evaluated in that precision exactly like the reference's registered buffer."""
    decay = torch.tensor(state.avg_decay, dtype=state.image.dtype)
    state.ema_accum = state.ema_accum * decay
    state.ema_value = state.ema_value * decay
    state.ema_value = state.ema_value + (1 - decay) * state.image


def adam_update(state, grad, lr=0.02, beta1=0.9, beta2=0.99, eps=1e-8):
    """torch.optim.Adam, single-tensor non-capturable branch (torch/optim/adam.py:414-547), as
    configured at style_transfer.py:458.  Bias corrections are Python doubles."""
    state.step += 1
    state.exp_avg = torch.lerp(state.exp_avg, grad, 1 - beta1)
    state.exp_avg_sq = state.exp_avg_sq * beta2 + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** state.step
    bc2 = 1 - beta2 ** state.step
    step_size = lr / bc1
    denom = state.exp_avg_sq.sqrt() / math.sqrt(bc2) + eps
    state.image = state.image - step_size * (state.exp_avg / denom)


def iterate(state, weights, targets, lr=0.02, **kw):
    """One pass of the hot loop body (:479-486): closure, Adam, clamp, EMA.  Returns (terms, total)."""
    terms, total, grad = loss_and_grad(state.image, weights, targets, **kw)
    adam_update(state, grad, lr=lr)
    state.image = state.image.clamp(0, 1)          # :483-485
    ema_update(state)
    return terms, total


# ----------------------------------------------------------------------------------------------
# scale transition (row §8(f)1); style_transfer.py:279-295,420-422,460-462,496-497
# ----------------------------------------------------------------------------------------------
def rescale_state(state, shape):
    """End-of-scale hand-off + start of the next scale: image <- EMA average (:496-497), bicubic
    resize + clamp (:420), fresh EMA (:421), Adam moments resampled with ``step`` kept (:285-295)."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', UserWarning)
        img = F.interpolate(state.average(), shape, mode='bicubic').clamp(0, 1)
        m = F.interpolate(state.exp_avg, shape, mode='bicubic')
        v = F.interpolate(state.exp_avg_sq, shape, mode='bilinear').relu()
    state.image, state.exp_avg, state.exp_avg_sq = img, m, v
    state.new_average()
    return state
