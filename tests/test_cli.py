"""CLI / presentation surface (reference cli.py:143-270, web_interface.py, setup.py:14-16):
option table, `-s N+`, TIFF/ICC output, trace.json schema, the web viewer, and one end-to-end run on the GPU."""
import json
import os
import socket
import struct
import time
import urllib.request

import numpy as np
import pytest
import torch

from style_transfer import cli
from style_transfer import style_transfer as S


def test_parser_takes_its_defaults_from_stylize():
    """cli.py:150-153: defaults and types of the stylize() options come from its keyword defaults / annotations."""
    args = cli.build_parser().parse_args(['content.png', 'a.png', 'b.png'])
    kw = S.StyleTransfer.stylize.__kwdefaults__
    for key in ('content_weight', 'tv_weight', 'optimizer', 'min_scale', 'iterations', 'initial_iterations',
                'step_size', 'avg_decay', 'init', 'style_scale_fac', 'style_size', 'style_weights'):
        assert getattr(args, key) == kw[key], key
    assert args.styles == ['a.png', 'b.png'] and args.output == 'out.png' and args.end_scale == '512'
    assert args.save_every == 50 and args.random_seed == 0 and args.pooling == 'max' and args.devices == []
    args = cli.build_parser().parse_args(['c.png', 's.png', '-s', '1024+', '-i', '7', '-ii', '9', '-cw', '0.1', '-tw', '3',
                                          '-ms', '64', '-ss', '0.05', '-ad', '0.9', '-sw', '2', '1', '--init', 'gray',
                                          '--optimizer', 'lbfgs', '--style-size', '300', '-r', '5', '-o', 'x.tif'])
    assert (args.end_scale, args.iterations, args.initial_iterations) == ('1024+', 7, 9)
    assert (args.content_weight, args.tv_weight, args.min_scale, args.step_size, args.avg_decay) == (0.1, 3.0, 64, 0.05, 0.9)
    assert args.style_weights == [2.0, 1.0] and args.init == 'gray' and args.optimizer == 'lbfgs'
    assert args.style_size == 300 and args.random_seed == 5 and args.output == 'x.tif'
    with pytest.raises(SystemExit):
        cli.build_parser().parse_args(['c.png', 's.png', '--init', 'style_mean'])


def test_safe_scale():
    assert cli.get_safe_scale(512, 512, 512) == 512                    # reference cli.py:87-90
    assert cli.get_safe_scale(4000, 3000, 2508) == int((4 / 3) ** 0.5 * 2508)
    assert cli.get_safe_scale(3000, 4000, 2508) == cli.get_safe_scale(4000, 3000, 2508)


def test_tiff16_writer_round_trip(tmp_path):
    import style_transfer
    arr = (np.arange(11 * 13 * 3, dtype=np.uint32) * 997 % 65536).astype(np.uint16).reshape(11, 13, 3)
    path = tmp_path / 'x.tif'
    cli.save_image(path, arr)
    raw = path.read_bytes()
    assert raw[:4] == b'II*\x00'
    n = struct.unpack('<H', raw[8:10])[0]
    tags = {}
    for i in range(n):
        tag, typ, count, value = struct.unpack('<HHII', raw[10 + i * 12:22 + i * 12])
        tags[tag] = (typ, count, value)
    assert list(tags) == sorted(tags)                                   # IFD entries must be sorted
    assert tags[256][2] == 13 and tags[257][2] == 11 and tags[277][2] & 0xffff == 3
    off, cnt = tags[273][2], tags[279][2]
    assert np.array_equal(np.frombuffer(raw[off:off + cnt], dtype='<u2').reshape(11, 13, 3), arr)
    icc_off, icc_len = tags[34675][2], tags[34675][1]
    assert raw[icc_off:icc_off + icc_len] == style_transfer.srgb_profile
    from PIL import Image
    im = Image.open(path)
    assert im.size == (13, 11) and im.info['icc_profile'] == style_transfer.srgb_profile
    with pytest.raises(ValueError):
        cli.save_image(tmp_path / 'x.png', arr)                        # uint16 arrays only go to TIFF


def test_load_image_honours_embedded_profiles(tmp_path):
    import style_transfer
    from PIL import Image
    arr = (np.arange(8 * 8 * 3) % 256).astype(np.uint8).reshape(8, 8, 3)
    Image.fromarray(arr, 'RGB').save(tmp_path / 'plain.png')
    Image.fromarray(arr, 'RGB').save(tmp_path / 'tagged.png', icc_profile=style_transfer.srgb_profile)
    Image.fromarray(arr[:, :, 0], 'L').save(tmp_path / 'gray.png')
    for name in ('plain.png', 'tagged.png'):
        im = cli.load_image(tmp_path / name)
        assert im.mode == 'RGB' and np.array_equal(np.asarray(im), arr)
    assert cli.load_image(tmp_path / 'gray.png').mode == 'RGB'


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_web_interface_serves_the_current_iterate():
    import style_transfer
    port = _free_port()
    wi = style_transfer.WebInterface('127.0.0.1', port)
    try:
        with pytest.raises(urllib.error.HTTPError):                      # no image yet -> 404
            urllib.request.urlopen(f'http://127.0.0.1:{port}/image', timeout=5)
        page = urllib.request.urlopen(f'http://127.0.0.1:{port}/', timeout=5).read().decode()
        assert '/websocket' in page and '/image' in page
        img = torch.rand(3, 24, 32)
        wi.put_iterate(S.STIterate(w=32, h=24, i=1, i_max=2, loss=0.5, time=time.time(), gpu_ram=0), img)
        for _ in range(50):
            try:
                body = urllib.request.urlopen(f'http://127.0.0.1:{port}/image', timeout=5).read()
                break
            except urllib.error.HTTPError:
                time.sleep(0.1)
        from PIL import Image
        import io
        im = Image.open(io.BytesIO(body))
        assert im.format == 'JPEG' and im.size == (32, 24)
        wi.put_done()
    finally:
        wi.close()
    assert not wi._thread.is_alive()


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path, monkeypatch):
    """`style_transfer content style -s 64 ...` on the GPU: output image, periodic saves, trace.json schema
    (cli.py:139-140,269-270) - the loss trace equals the one the Python API gives for the same arguments."""
    from PIL import Image
    import style_transfer
    from conftest import load_golden
    g = load_golden('stylize_e2e')
    Image.fromarray(g['content_u8'], 'RGB').save(tmp_path / 'content.png')
    Image.fromarray(g['style_u8'], 'RGB').save(tmp_path / 'style.png')
    monkeypatch.chdir(tmp_path)
    cli.main(['content.png', 'style.png', '-o', 'out.tif', '-s', '64', '-ms', '45', '-i', '3', '-ii', '4',
              '--save-every', '2', '--weights', 'synthetic', '--devices', 'cuda:0'])
    trace = json.load(open(tmp_path / 'trace.json'))
    assert set(trace) == {'args', 'iterates'} and trace['args']['end_scale'] == 64
    its = trace['iterates']
    assert [(i['w'], i['h'], i['i'], i['i_max']) for i in its] == [tuple(int(v) for v in r[:4]) for r in g['iterates']]
    assert set(its[0]) == {'w', 'h', 'i', 'i_max', 'loss', 'time', 'gpu_ram'}
    rel = np.abs(np.array([i['loss'] for i in its]) - g['iterates'][:, 4]) / g['iterates'][:, 4]
    assert np.all(rel <= 5e-4), rel
    raw = (tmp_path / 'out.tif').read_bytes()
    assert raw[:4] == b'II*\x00' and len(raw) > 64 * 64 * 6


# ---- one process per GPU under torchrun ----------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _dist_init_worker(rank, world, port, out):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), ST_DIST_BACKEND='gloo')
    from style_transfer import cli as c
    devices, seen_world = c.init_distributed([])
    from style_transfer.style_transfer import _dist_info
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    out.put((rank, [str(d) for d in devices], seen_world, _dist_info(), c._is_rank0(), float(t)))
    c._leave_distributed()
    assert not dist.is_initialized()


def test_cli_joins_the_process_group_under_torchrun():
    """RANK / WORLD_SIZE / LOCAL_RANK in the environment: main() binds the process to cuda:LOCAL_RANK and initialises
    the group stylize() shards over - without it every rank would run the whole job on cuda:0 (round-3 advisor
    finding).  gloo, world 2, no GPU needed: only the launch logic runs here."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dist_init_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = sorted(out.get() for _ in range(2))
    assert got[0] == (0, ['cuda:0'], 2, (0, 2), True, 3.0)
    assert got[1] == (1, ['cuda:1'], 2, (1, 2), False, 3.0)


def test_cli_plain_launch_needs_no_process_group(monkeypatch):
    import torch.distributed as dist
    for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(key, raising=False)
    devices, world = cli.init_distributed(['cuda:0'])
    assert [str(d) for d in devices] == ['cuda:0'] and world == 1 and not dist.is_initialized()
    assert cli._is_rank0()


def _cli_rank(rank, world, port, tmp, out):
    import os
    import traceback
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                          MASTER_PORT=str(port), ST_DIST_BACKEND='gloo', ST_CLI_SAME_DEVICE='1')
        os.chdir(tmp)
        from style_transfer import cli as c
        c.main(['content.png', 'style.png', '-o', 'out.png', '-s', '96', '-ms', '48', '-i', '3', '-ii', '4',
                '--save-every', '2', '--weights', 'synthetic'])
        out.put(('ok', rank))
    except BaseException:                        # noqa: BLE001 - reported to the parent (SystemExit included)
        out.put(('error', rank, traceback.format_exc()))
        raise


@pytest.mark.gpu
def test_cli_under_two_ranks_shards_and_rank0_writes(tmp_path):
    """The CLI as `torchrun --nproc-per-node 2` starts it (two OS processes sharing cuda:0 over gloo - a gpurun box
    has one GPU): the 96-pixel scale is cut into two strips, only rank 0 writes out.png / trace.json, and the loss
    trace follows the single-process run of the same command."""
    import torch.multiprocessing as mp
    from PIL import Image
    from conftest import load_golden
    g = load_golden('stylize_e2e')
    Image.fromarray(g['content_u8'], 'RGB').resize((96, 80)).save(tmp_path / 'content.png')
    Image.fromarray(g['style_u8'], 'RGB').resize((90, 96)).save(tmp_path / 'style.png')
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_rank, args=(r, 2, port, str(tmp_path), out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=420)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    assert not alive, 'a rank hung'
    results = [out.get() for _ in range(2)]
    assert all(r[0] == 'ok' for r in results), [r for r in results if r[0] != 'ok'][0][2]
    sharded = json.load(open(tmp_path / 'trace.json'))['iterates']
    assert Image.open(tmp_path / 'out.png').size == (96, 80)
    os.rename(tmp_path / 'trace.json', tmp_path / 'trace2.json')
    cwd = os.getcwd()
    try:
        os.chdir(tmp_path)
        cli.main(['content.png', 'style.png', '-o', 'single.png', '-s', '96', '-ms', '48', '-i', '3', '-ii', '4',
                  '--save-every', '2', '--weights', 'synthetic'])
    finally:
        os.chdir(cwd)
    single = json.load(open(tmp_path / 'trace.json'))['iterates']
    assert [(i['w'], i['h'], i['i']) for i in sharded] == [(i['w'], i['h'], i['i']) for i in single]
    rel = max(abs(a['loss'] - b['loss']) / abs(b['loss']) for a, b in zip(sharded, single))
    assert rel < 1e-3, rel
