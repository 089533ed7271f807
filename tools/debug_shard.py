import sys, os, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(R, 'style-transfer-pytorch_amd'), os.path.join(R, 'oracle')]
from style_transfer import _hip as hip, sharding as sh, vgg
DEV = 'cuda:0'
h, w, world = 96, 80, 2
g = torch.Generator().manual_seed(1)
img = torch.rand((1, 3, h, w), generator=g)
net = hip.Net(vgg.synthetic_vgg19_weights(0), 'max', DEV)
whole = hip.Plan(net, h, w)
whole.forward(img.to(DEV), 29)
rows = sh.strip_rows(h, world)
plans = [sh.StripPlan(net, h, w, b, e) for b, e in rows]
for p, (b, e) in zip(plans, rows):
    p.forward_begin(img[:, :, b:e].contiguous().to(DEV), 29)
sh.run_phases_lockstep(plans)
for layer in [1, 3, 4, 6, 8, 9, 11, 20, 22, 29]:
    fw = whole.feature(layer)
    fs = torch.cat([p.feature(layer) for p in plans], dim=2)
    d = (fw - fs).abs()
    rowmax = d.amax(dim=(0, 1, 3))
    print('feat', layer, tuple(fw.shape), 'max abs diff', float(d.max()), 'rows with diff', [int(i) for i in torch.nonzero(rowmax > 0).flatten()[:12]])
for layer in [1, 6, 11, 20, 29]:
    c = {1: 64, 6: 128, 11: 256, 20: 512, 29: 512}[layer]
    lvl = {1: 0, 6: 1, 11: 2, 20: 3, 29: 4}[layer]
    mean, srm = whole.moments(layer)
    tot = sum(p.moment_sums(layer) for p in plans)
    npix = float((h >> lvl) * (w >> lvl))
    srm_s, mean_s = (tot[:c * c] / npix).reshape(c, c), tot[c * c:] / npix
    print('moments', layer, 'srm rel', float((srm_s - srm).norm() / srm.norm()), 'mean rel', float((mean_s - mean).norm() / mean.norm()),
          'ratio', float(srm_s.norm() / srm.norm()))
