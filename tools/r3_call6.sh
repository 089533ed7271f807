#!/bin/bash
mkdir -p gpurun_out/r3c6
O=gpurun_out/r3c6
timeout 300 python tools/strip_bench.py 2896x2172 8 2>&1 | grep strip_bench | tee $O/strip_2896x8.log
timeout 300 python tools/strip_bench.py 2048 4 2>&1 | grep strip_bench | tee $O/strip_2048x4.log
# kernel trace of ONE rank's stubbed step (rank 7 of 8 at 2896x2172): where does the time outside the convs go?
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
cat > /tmp/one_rank.py <<'PY'
import os, sys
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0] = [R, os.path.join(R, 'style-transfer-pytorch_amd'), os.path.join(R, 'tests'), os.path.join(R, 'oracle')]
import torch, bench
from style_transfer import _hip as hip, sharding as sh, vgg
from test_sharding_gpu import _targets_lockstep, _smooth
height, width, world, r = 2172, 2896, 8, 7
net = hip.Net(vgg.synthetic_vgg19_weights(0), 'max', 'cuda:0', 'fp16x3')
content, style, image = _smooth(31, height, width), _smooth(32, height, width), _smooth(33, height, width)
rows = sh.strip_rows(height, world)
plans = [sh.StripPlan(net, height, width, b, e).set_rank(i, world) for i, (b, e) in enumerate(rows)]
_targets_lockstep(sh, plans, content, [style], [1.0])
b, e = rows[r]
img = image[:, :, b:e].contiguous().to('cuda:0'); g = torch.empty_like(img)
m = torch.zeros_like(img); v = torch.zeros_like(img); ema = 0.01 * img
for k in range(1, 13):
    plans[r].closure_begin(img, g); sh.run_phases_lockstep([plans[r]], stub=True)
    plans[r].apply_update(img, g, m, v, ema, k, 0.02)
torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_rank7 -o r7 -- python /tmp/one_rank.py > $O/prof.log 2>&1
f=$(find $O/prof_rank7 -name "*kernel_stats.csv" | head -1); echo "stats: $f"; python tools/prof_summary.py "$f" 12 2>/dev/null | head -45 || head -30 "$f"
