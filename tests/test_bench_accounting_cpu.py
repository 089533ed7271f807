"""bench.py's work and traffic accounting against SURVEY.md 8(d) (CPU: pure arithmetic over the layer shapes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_conv_flops_match_the_survey():
    # conv fwd + dgrad 378.7 GFLOP @512^2, 1 514.8 @1024^2, 6 059.1 @2048^2, 9 079.5 @2896x2172
    for (h, w), want in (((512, 512), 378.7e9), ((1024, 1024), 1514.8e9), ((2048, 2048), 6059.1e9), ((2172, 2896), 9079.5e9)):
        assert abs(bench.conv_flops(h, w) - want) <= 1e-3 * want, (h, w, bench.conv_flops(h, w))


def test_algorithmic_bytes_of_the_trunk_launches():
    """Bare convolutions (operand + result + pre-split weights) and the closure's fused launches (mask / tap-gradient reads
    of the data-gradient epilogues, pooled writes): the figures DESIGN.md section 5 and the bench line quote."""
    bare, fused = bench.conv_algorithmic_bytes_per_launch(512, 512), bench.conv_fused_path_bytes_per_launch(512, 512)
    assert abs(bare - 41.36e6) < 0.01e6 and abs(fused - 50.69e6) < 0.01e6
    # the fused path moves more than the bare convolutions (8 mask reads + 5 tap reads outweigh 4 pooled writes) ...
    assert bare < fused < 1.3 * bare
    # ... and both scale with the pixel count up to the (constant) weights
    b4, f4 = bench.conv_algorithmic_bytes_per_launch(2048, 2048), bench.conv_fused_path_bytes_per_launch(2048, 2048)
    weights = sum(2 * 9 * ci * co * 4 for ci, co, _ in bench.CONV_SPECS[1:]) / (2 * len(bench.CONV_SPECS[1:]))
    assert abs((b4 - weights) - 16 * (bare - weights)) < 1.0
    assert abs((f4 - weights) - 16 * (fused - weights)) < 64.0          # (px // 4 rounding of the pooled maps: none at these sizes)


def test_gpus_n_without_enough_devices_fails_loudly(tmp_path):
    """`python bench.py --gpus 2` with fewer than 2 HIP devices visible (this CPU container: none): a JSON line with value
    null and a non-zero exit code - never an N = 1 measurement under an N = 2 label (bench.self_launch)."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'ST_BENCH_SAME_DEVICE')}
    env['HIP_VISIBLE_DEVICES'] = ''
    env['CUDA_VISIBLE_DEVICES'] = ''
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.stdout[-500:], r.stderr[-500:])
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['value'] is None and d['n_gpus'] == 2 and 'FAILED' in d['config']['parallelism']
