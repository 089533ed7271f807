"""Command line front end of the MI355X style-transfer build (console script ``style_transfer``).

Same command line as the reference's ``style_transfer/cli.py:143-270`` - positional ``content style [style ...]``,
the same option names / short forms / defaults (those of ``stylize()`` are taken from its keyword defaults and
annotations exactly as the reference does, cli.py:150-153), ``-s N+`` safe-scale syntax (:87-90,236-239), ICC
colour management of inputs and outputs (:23-60), 16-bit TIFF output for ``.tif`` (:63-71), periodic saves
(:126-135), the web viewer (``--web``) and ``trace.json`` (:139-140,269-270: ``{"args": ..., "iterates": [...]}``).

Additions of this build: ``--weights`` (path of torchvision's ``vgg19-dcbb9e9d.pth`` or ``synthetic[:seed]`` -
there is no network here to download it) and ``--precision`` (conv arithmetic).  Cold path only:
nothing here is on the per-iteration hot path, which is one ``Plan.step`` call into libst_amd.so.
"""

import argparse
import atexit
import dataclasses
import io
import json
import os
import struct
import sys
from pathlib import Path

import numpy as np
import torch
from PIL import Image, ImageCms

STYLIZE_OPTIONS = [
    # (flags, stylize keyword, extra argparse settings, help)
    (('--content-weight', '-cw'), 'content_weight', {}, 'the content weight'),
    (('--tv-weight', '-tw'), 'tv_weight', {}, 'the smoothing weight'),
    (('--optimizer',), 'optimizer', {'choices': ['adam', 'lbfgs']}, 'the optimizer to use'),
    (('--min-scale', '-ms'), 'min_scale', {}, 'the minimum scale (max image dim), in pixels'),
    (('--iterations', '-i'), 'iterations', {}, 'the number of iterations per scale'),
    (('--initial-iterations', '-ii'), 'initial_iterations', {}, 'the number of iterations on the first scale'),
    (('--step-size', '-ss'), 'step_size', {}, 'the step size (learning rate) for Adam'),
    (('--avg-decay', '-ad'), 'avg_decay', {}, 'the EMA decay rate for iterate averaging'),
    (('--init',), 'init', {'choices': ['content', 'gray', 'uniform', 'normal', 'style_stats']}, 'the initial image'),
    (('--style-scale-fac',), 'style_scale_fac', {}, 'the relative scale of the style to the content'),
    (('--style-size',), 'style_size', {}, 'the fixed scale of the style at different content scales'),
]


def _say(text):
    try:
        from tqdm import tqdm
        tqdm.write(text)
    except ImportError:
        print(text)


def _fail(err):
    print('\033[31m{}:\033[0m {}'.format(type(err).__name__, err), file=sys.stderr)
    sys.exit(1)


# ---- colour management and image files (reference cli.py:23-84) ---------------------------------
def _srgb():
    from . import srgb_profile
    return srgb_profile


def _convert_profile(image, src, dst, mode):
    return ImageCms.profileToProfile(image, io.BytesIO(src), io.BytesIO(dst), outputMode=mode)


def load_image(path, proof_prof=None):
    """RGB PIL image in sRGB; an embedded ICC profile is honoured; ``proof_prof`` soft-proofs through a CMYK profile."""
    srgb = _srgb()
    try:
        image = Image.open(path)
        embedded = image.info.get('icc_profile')
        if embedded is None:
            image = image.convert('RGB')
        src = embedded or srgb
        if proof_prof is not None:
            proof = Path(proof_prof).read_bytes()
            return _convert_profile(_convert_profile(image, src, proof, 'CMYK'), proof, srgb, 'RGB')
        if src == srgb:
            return image.convert('RGB')
        return _convert_profile(image, src, srgb, 'RGB')
    except OSError as err:
        _fail(err)


def write_tiff16(path, arr, icc):
    """Baseline uncompressed 16-bit RGB TIFF with an embedded ICC profile (tag 34675), 72 dpi.
    ``arr``: H x W x 3 uint16 (what ``StyleTransfer.get_image('np_uint16')`` returns).  The reference uses
    tifffile (cli.py:63-71), which this image does not have; a single-strip TIFF needs no library."""
    arr = np.ascontiguousarray(arr, dtype='<u2')
    h, w, c = arr.shape
    if c != 3:
        raise ValueError('expected an H x W x 3 array')
    data = arr.tobytes()
    entries = []          # (tag, type, count, value bytes or None for offset data, payload)
    extra = bytearray()
    n_tags = 13
    ifd_offset = 8
    extra_offset = ifd_offset + 2 + n_tags * 12 + 4

    def place(payload):
        off = extra_offset + len(extra)
        extra.extend(payload)
        if len(extra) % 2:
            extra.append(0)
        return off

    def short(tag, v):
        entries.append(struct.pack('<HHIHH', tag, 3, 1, v, 0))

    def long_(tag, v):
        entries.append(struct.pack('<HHII', tag, 4, 1, v))

    def ref(tag, typ, count, payload):
        entries.append(struct.pack('<HHII', tag, typ, count, place(payload)))

    long_(256, w)                                                  # ImageWidth
    long_(257, h)                                                  # ImageLength
    ref(258, 3, 3, struct.pack('<HHH', 16, 16, 16))                # BitsPerSample
    short(259, 1)                                                  # Compression: none
    short(262, 2)                                                  # Photometric: RGB
    strip_entry_index = len(entries)
    entries.append(None)                                           # StripOffsets, patched below
    short(277, 3)                                                  # SamplesPerPixel
    long_(278, h)                                                  # RowsPerStrip
    long_(279, len(data))                                          # StripByteCounts
    ref(282, 5, 1, struct.pack('<II', 72, 1))                      # XResolution
    ref(283, 5, 1, struct.pack('<II', 72, 1))                      # YResolution
    short(296, 2)                                                  # ResolutionUnit: inch
    ref(34675, 7, len(icc), icc)                                   # InterColorProfile
    assert len(entries) == n_tags
    entries[strip_entry_index] = struct.pack('<HHII', 273, 4, 1, extra_offset + len(extra))
    with open(path, 'wb') as f:
        f.write(struct.pack('<2sHI', b'II', 42, ifd_offset))
        f.write(struct.pack('<H', n_tags))
        f.write(b''.join(entries))
        f.write(struct.pack('<I', 0))
        f.write(bytes(extra))
        f.write(data)


def _is_rank0():
    """Under torchrun (one process per GPU, strip sharding) only rank 0 writes files / feeds the web viewer.  Asked of
    the process group stylize() itself consults (style_transfer._dist_info), so the two cannot disagree."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank() == 0
    return True


def init_distributed(requested_devices):
    """One process per GPU under ``torchrun --nproc-per-node N -m style_transfer.cli ...`` (RANK / WORLD_SIZE /
    LOCAL_RANK in the environment): bind this process to ``cuda:LOCAL_RANK``, default ``--devices`` to it and join the
    process group stylize() shards over (backend nccl = RCCL over xGMI; ``ST_DIST_BACKEND=gloo`` for the CPU-side
    tests and ``ST_CLI_SAME_DEVICE=1`` to put every rank on cuda:0 - functional tests on a one-GPU box, where RCCL
    refuses two ranks on one device).  Returns (devices, world); (requested or [cuda:0], 1) for a plain launch.
    Replaces the reference's --devices pair (cli.py:199-215, style_transfer.py:326-333)."""
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 or 'RANK' not in os.environ:
        return [torch.device(d) for d in requested_devices] or [torch.device('cuda:0')], 1
    local = 0 if os.environ.get('ST_CLI_SAME_DEVICE') == '1' else int(os.environ.get('LOCAL_RANK', os.environ['RANK']))
    device = torch.device('cuda', local)
    if requested_devices and [torch.device(d) for d in requested_devices] != [device]:
        _say(f'--devices {" ".join(map(str, requested_devices))} ignored under torchrun: rank '
             f'{os.environ["RANK"]} runs on {device}')
    backend = os.environ.get('ST_DIST_BACKEND', 'nccl')
    if torch.cuda.is_available():
        torch.cuda.set_device(device)
    if not dist.is_initialized():
        kw = {'device_id': device} if backend == 'nccl' and torch.cuda.is_available() else {}
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, **kw)               # env:// rendezvous (torchrun's MASTER_ADDR / MASTER_PORT)
        atexit.register(_leave_distributed)
    return [device], dist.get_world_size()


def _leave_distributed():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        from . import sharding
        sharding.release_head_groups()
        dist.destroy_process_group()


def save_image(path, image):
    path = Path(path)
    _say(f'Writing image to {path}.')
    suffix = path.suffix.lower()
    try:
        if isinstance(image, Image.Image):
            options = {'icc_profile': _srgb()}
            if suffix in ('.jpg', '.jpeg'):
                options.update(quality=95, subsampling=0)
            elif suffix == '.webp':
                options.update(quality=95)
            image.save(path, **options)
        elif isinstance(image, np.ndarray) and suffix in ('.tif', '.tiff'):
            write_tiff16(path, image, _srgb())
        else:
            raise ValueError('Unsupported combination of image type and extension')
    except OSError as err:
        _fail(err)


def get_safe_scale(w, h, dim):
    """End scale for a w x h image such that its pixel count does not exceed a dim x dim square's
    (``-s N+``; reference cli.py:87-90)."""
    aspect = w / h if w > h else h / w
    return int(pow(aspect, 1 / 2) * dim)


# ---- per-iteration reporting (reference cli.py:107-140) -------------------------------------------
class Callback:
    def __init__(self, st, args, image_type='pil', web_interface=None):
        self.st, self.args, self.image_type, self.web_interface = st, args, image_type, web_interface
        self.iterates = []
        self.progress = None

    def _bar(self, total):
        try:
            from tqdm import tqdm
            return tqdm(total=total, dynamic_ncols=True)
        except ImportError:
            return None

    def __call__(self, iterate):
        self.iterates.append(dataclasses.asdict(iterate))
        if iterate.i == 1:
            self.progress = self._bar(iterate.i_max)
        _say('Size: {}x{}, iteration: {}, loss: {:g}'.format(iterate.w, iterate.h, iterate.i, iterate.loss))
        if self.progress is not None:
            self.progress.update()
        # get_image* gathers the row strips of a sharded scale: EVERY rank calls it (the same iterations on every rank),
        # rank 0 alone uses the result
        if getattr(self.args, 'web', False) or self.web_interface is not None:
            preview = self.st.get_image_tensor()
            if self.web_interface is not None:
                self.web_interface.put_iterate(iterate, preview)
        last_of_scale = iterate.i == iterate.i_max
        if last_of_scale:
            self.close()
            if max(iterate.w, iterate.h) != self.args.end_scale:
                self._save()
            elif self.web_interface is not None:
                self.web_interface.put_done()
        elif iterate.i % self.args.save_every == 0:
            self._save()

    def _save(self):
        image = self.st.get_image(self.image_type)
        if _is_rank0():
            save_image(self.args.output, image)

    def close(self):
        if self.progress is not None:
            self.progress.close()
            self.progress = None

    def get_trace(self):
        return {'args': self.args.__dict__, 'iterates': self.iterates}


def build_parser():
    from . import StyleTransfer
    defaults = StyleTransfer.stylize.__kwdefaults__
    types = StyleTransfer.stylize.__annotations__
    p = argparse.ArgumentParser(prog='style_transfer', description=__doc__.split('\n')[0],
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument('content', type=str, help='the content image')
    p.add_argument('styles', type=str, nargs='+', metavar='style', help='the style images')
    p.add_argument('--output', '-o', type=str, default='out.png', help='the output image')
    p.add_argument('--style-weights', '-sw', type=float, nargs='+', default=None, metavar='STYLE_WEIGHT',
                   help='the relative weights for each style image')
    p.add_argument('--devices', type=str, default=[], nargs='+', help='the device names to use (omit for auto)')
    p.add_argument('--random-seed', '-r', type=int, default=0, help='the random seed')
    for flags, key, extra, text in STYLIZE_OPTIONS:
        p.add_argument(*flags, default=defaults[key], type=types[key], help=text, **extra)
    p.add_argument('--end-scale', '-s', type=str, default='512', help='the final scale (max image dim), in pixels')
    p.add_argument('--save-every', type=int, default=50, help='save the image every SAVE_EVERY iterations')
    p.add_argument('--pooling', type=str, default='max', choices=['max', 'average', 'l2'],
                   help="the model's pooling mode")
    p.add_argument('--proof', type=str, default=None,
                   help='the ICC color profile (CMYK) for soft proofing the content and styles')
    p.add_argument('--web', default=False, action='store_true', help='enable the web interface')
    p.add_argument('--host', type=str, default='0.0.0.0', help='the host the web interface binds to')
    p.add_argument('--port', type=int, default=8080, help='the port the web interface binds to')
    p.add_argument('--browser', type=str, default='', nargs='?',
                   help='open a web browser (specify the browser if not system default)')
    p.add_argument('--weights', type=str, default=None,
                   help="torchvision vgg19-dcbb9e9d.pth, or 'synthetic[:seed]' (default: $STYLE_TRANSFER_VGG19 / the "
                        "torch hub cache)")
    p.add_argument('--precision', type=str, default='fp16x3', choices=['fp16x3', 'bf16x6', 'fp32', 'bf16x3'],
                   help='arithmetic of the 3x3 trunk convolutions')
    p.add_argument('--trace', type=str, default='trace.json', help='where to write the iteration trace')
    return p


def main(argv=None):
    from . import StyleTransfer
    args = build_parser().parse_args(argv)

    content_img = load_image(args.content, args.proof)
    style_imgs = [load_image(path, args.proof) for path in args.styles]
    image_type = 'np_uint16' if Path(args.output).suffix.lower() in ('.tif', '.tiff') else 'pil'

    devices, world = init_distributed(args.devices)
    if len({d.type for d in devices}) != 1:
        print('Devices must all be the same type.')
        sys.exit(1)
    if not 1 <= len(devices) <= 8:
        print('Only 1 to 8 devices are supported.')          # (reference cli.py:214-216: 1 or 2)
        sys.exit(1)
    if len(devices) > 1:
        _say(f'{len(devices)} devices: the image is cut into one row strip per device, one worker process per extra device')
    if devices[0].type != 'cuda' or not torch.cuda.is_available():
        print('This build needs a HIP device (MI355X; PyTorch-ROCm names it cuda:N): there is no CPU path.')
        sys.exit(1)
    if world > 1:
        _say(f'Rank {os.environ["RANK"]} of {world} on {devices[0]} (row strips, one process per GPU)')
    if _is_rank0():
        print('Using devices:', ' '.join(str(d) for d in devices))
        for i, device in enumerate(devices):
            props = torch.cuda.get_device_properties(device)
            print(f'GPU {i} type: {props.name} ({getattr(props, "gcnArchName", "")})')
            print(f'GPU {i} RAM:', round(props.total_memory / 1024 / 1024), 'MB')

    end_scale = int(args.end_scale.rstrip('+'))
    if args.end_scale.endswith('+'):
        end_scale = get_safe_scale(*content_img.size, end_scale)
    args.end_scale = end_scale

    web_interface = None
    if args.web and _is_rank0():
        from .web_interface import WebInterface
        web_interface = WebInterface(args.host, args.port)
        atexit.register(web_interface.close)
        if args.browser is None or args.browser:
            import webbrowser
            url = f'http://{args.host}:{args.port}/'
            (webbrowser.get(args.browser) if args.browser else webbrowser).open(url)

    torch.manual_seed(args.random_seed)
    print('Loading model...')
    st = StyleTransfer(devices=devices, pooling=args.pooling, weights=args.weights, precision=args.precision)
    callback = Callback(st, args, image_type=image_type, web_interface=web_interface)
    atexit.register(callback.close)

    accepted = StyleTransfer.stylize.__kwdefaults__
    st_kwargs = {k: v for k, v in vars(args).items() if k in accepted and k != 'callback'}
    interrupted = False
    try:
        st.stylize(content_img, style_imgs, **st_kwargs, callback=callback)
    except KeyboardInterrupt:
        interrupted = True                     # keep what has been computed so far (reference :261-266)

    if interrupted and world > 1:
        # The ranks left stylize() at different points of the phase machine: gathering the strips now would issue
        # mismatched collectives and hang until the process-group timeout.  Each rank keeps what it alone holds:
        # rank 0 writes the last image the periodic save gathered (--save-every), nothing is exchanged.
        _say('Interrupted in a sharded run: keeping the last periodically saved image (no gather after an interrupt).')
        result = None
    else:
        result = st.get_image(image_type)
    if result is not None and _is_rank0():
        save_image(args.output, result)
    if _is_rank0():
        with open(args.trace, 'w') as fp:
            json.dump(callback.get_trace(), fp, indent=4)


if __name__ == '__main__':
    main()
