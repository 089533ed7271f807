"""bench.py's one-line JSON contract (what the round driver parses), on a real MI355X: the N = 1 line, and the N = 2
launch contract (torch.distributed.run, RANK / LOCAL_RANK / WORLD_SIZE from the environment, rank 0 prints) with both
ranks on the single available GPU over gloo - functional only, the numbers of that run mean nothing."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, f'expected exactly one JSON line, got {len(lines)}:\n{out[-2000:]}'
    return json.loads(lines[0])


def test_single_gpu_line():
    r = subprocess.run([sys.executable, 'bench.py', '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-extra'],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
        assert key in d, key
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['value'] > 0 and abs(d['value'] * d['ms_per_step'] / 1e3 - 1.0) < 1e-6
    # dtype names the arithmetic of the path: fp32 everywhere, the 3x3 convs as fp16x3 split planes on the MFMA
    assert d['dtype'].startswith('f32') and 'fp16x3' in d['dtype'] and d['data'] == 'synthetic' and d['vs_baseline'] is None
    assert d['scaling'] == 'strong'
    assert 'workload' in d['config'] and '512x512' in d['config']['workload']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and 0 < rf['frac'] < 1
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9


def test_two_rank_launch_contract():
    env = dict(os.environ, ST_BENCH_SAME_DEVICE='1')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--dist-backend', 'gloo', '--size', '512']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _last_json(r.stdout)
    # N > 1 defaults to STRONG scaling: the same image cut into N strips, value = that image's iterations/s
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['value'] > 0
    assert abs(d['value'] * d['ms_per_step'] / 1e3 - 1.0) < 1e-6
    par = d['config']['parallelism']
    assert 'row strips' in par and 'strong' in par and 'FAILED' not in par, par
    assert '512x512' in d['config']['workload']
    assert 'cpu_baseline' not in d                    # rank 0 at N = 1 only


def test_plain_command_with_gpus_2_launches_two_ranks():
    """VERDICT r4 missing #1: the driver's command is plain `python bench.py --gpus N` - without a launcher the script
    re-runs itself as N ranks under torch.distributed.run (bench.self_launch) instead of printing an N = 1 line."""
    env = dict(os.environ, ST_BENCH_SAME_DEVICE='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1', '--dist-backend', 'gloo',
                        '--size', '512'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['scaling'] == 'strong'
    par = d['config']['parallelism']
    assert '2 row strips' in par and 'world size 2' in par and 'backend gloo' in par and 'RCCL' in par and 'FAILED' not in par, par
    assert len(d['value_regions']) == 3 and d['value_best_of_3'] >= d['value'] * (1 - 1e-9)


def test_two_rank_fallback_to_the_conservative_transport():
    """The guard around the first sharded iteration (bench.py): a rank that raises takes EVERY rank to the conservative
    transport (host-synchronised exchanges, whole launches, replicated chains), once, and the line says so.  The failure
    is injected on every rank at the same point (the recoverable kind: an unsupported call, a transport error in one
    collective); both ranks share the one GPU over gloo (functional only)."""
    env = dict(os.environ, ST_BENCH_SAME_DEVICE='1', ST_BENCH_INJECT_FAILURE='all')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--dist-backend', 'gloo', '--size', '512']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    d = _last_json(r.stdout)
    assert d['n_gpus'] == 2 and d['value'] > 0
    par = d['config']['parallelism']
    assert 'CONSERVATIVE transport' in par and 'FAILED' not in par, par


def test_single_rank_strip_path_over_rccl():
    """The only piece of the RCCL transport one GPU can run: `--mode shard` under a one-process launcher cuts the image
    into ONE strip and, with ST_FABRIC_FORCE_COLLECTIVES=1, still issues every all-reduce (Gram moments, loss partials),
    reduce-to-owner and broadcast (owned style heads) of the phase machine - through a real RCCL communicator (plus the
    heads' second communicator), on zero-copy views of the library's buffers, ordered on the library's communication / head
    streams (torch.cuda.ExternalStream).  The point-to-point halos need a neighbour and stay untested on hardware.  The
    stream-ordered path must run as is (no fallback), and land on the unsharded run's loss."""
    env = dict(os.environ, ST_FABRIC_FORCE_COLLECTIVES='1')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '1', '--mode', 'shard', '--steps', '4', '--warmup', '1',
           '--size', '256', '--no-extra', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    par = d['config']['parallelism']
    assert d['value'] > 0 and 'FAILED' not in par and 'CONSERVATIVE' not in par, par
    # the unsharded run of the same image: 1 (first-iteration guard) + 1 + 4 iterations there = --warmup 2 --steps 4 here
    r1 = subprocess.run([sys.executable, 'bench.py', '--steps', '4', '--warmup', '2', '--size', '256', '--no-extra',
                         '--no-cpu-baseline'], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    d1 = _last_json(r1.stdout)
    rel = abs(d['final_loss'] - d1['final_loss']) / d1['final_loss']
    print(f"[rccl-1] strip path over RCCL, one rank: {d['value']:.1f} it/s, final loss {d['final_loss']:.6f} vs unsharded "
          f"{d1['final_loss']:.6f} (rel {rel:.1e}); parallelism: {par}")
    assert rel < 1e-3


def test_a_first_iteration_past_its_deadline_falls_back_to_the_next_transport():
    """bench.py awaits the first sharded iteration on the host with a deadline (ST_BENCH_FIRST_STEP_S): an attempt that does
    not finish in time has its in-library communicators aborted and the next transport takes over.  Injected here with a
    deadline of zero for the in-library attempt (ST_BENCH_INJECT_FAILURE=deadline): the line must come from the
    stream-ordered torch.distributed transport, with a sane value."""
    env = dict(os.environ, ST_FABRIC_FORCE_COLLECTIVES='1', ST_BENCH_INJECT_FAILURE='deadline')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr',
           '127.0.0.1', '--master-port', str(port), 'bench.py', '--gpus', '1', '--mode', 'shard', '--steps', '4', '--warmup', '1',
           '--size', '256', '--no-extra', '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _last_json(r.stdout)
    par = d['config']['parallelism']
    print('[deadline]', par[-260:])
    assert d['value'] > 0 and 'FAILED' not in par and 'CONSERVATIVE' not in par, par
    assert 'did not complete within' in par and 'stream-ordered torch.distributed' in par, par
