"""Drop-in ``StyleTransfer`` / ``stylize()`` whose per-iteration hot path runs in libst_amd.so.

Mirrors the public surface of reference ``style_transfer/style_transfer.py:256-499`` (names, keyword-
only parameters, defaults, annotations, exceptions, callback protocol) so that ``cli.py`` - which
introspects ``stylize.__kwdefaults__`` / ``__annotations__`` (reference cli.py:150-153) - and user code
keep working.  What differs on purpose:

* the closure, Adam step, clamp and EMA update (reference :472-486) are ONE call into the HIP
  library per iteration (``Plan.step``); there is no autograd graph and no torchvision;
* devices must be HIP devices (PyTorch-ROCm names them ``cuda:N``).  There is no CPU path;
* the VGG-19 weights come from a user-supplied torchvision checkpoint (``vgg19-dcbb9e9d.pth``) or,
  for tests/benchmarks, from the seeded synthetic generator - this image has no network.

Python keeps only the cold path: the multi-scale pyramid, PIL resizing, target blending and the
scale-to-scale resampling of the optimiser state.
"""

import os
import time
import warnings
from dataclasses import dataclass

import numpy as np
import torch
from PIL import Image
from torch.nn import functional as F

from . import _hip, sqrtm, vgg  # noqa: F401  (sqrtm: importable as in the reference, :17)
# the reference defines its loss / bookkeeping modules in this file (:93-234); user code imports them from here
from .losses import (ContentLoss, ContentLossMSE, LayerApply, Scale, ScaledMSELoss, StyleLoss,  # noqa: F401
                     StyleLossW2, SumLoss, TVLoss, eye_like)

CONTENT_LAYERS = [22]
STYLE_LAYERS = [1, 6, 11, 20, 29]


# ---- small host helpers (reference :256-306) ---------------------------------------------------
def size_to_fit(size, max_dim, scale_up=False):
    """Fit (w, h) inside a max_dim square keeping aspect ratio (reference :256-265)."""
    w, h = size
    if not scale_up and max(h, w) <= max_dim:
        return w, h
    if h > w:
        return round(max_dim * w / h), max_dim
    return max_dim, round(max_dim * h / w)


def gen_scales(start, end):
    """end, end/sqrt2, end/2, ... down to >= start, ascending (reference :268-276)."""
    found, i, scale = set(), 0, end
    while scale >= start:
        found.add(scale)
        i += 1
        scale = round(end / pow(2, i / 2))
    return sorted(found)


def interpolate(*args, **kwargs):
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', UserWarning)
        return F.interpolate(*args, **kwargs)


def scale_adam(state, shape):
    """A torch.optim.Adam ``state_dict()`` resampled to a new image size - the warm start of the next scale
    (reference :285-295): first moments bicubic, second moments (and amsgrad's running maximum) bilinear and
    clipped at zero, the step count kept.  stylize() itself keeps its Adam state in AdamState (the fused HIP step
    owns the update); this function serves user code that drives a torch optimiser the reference's way."""
    import copy
    out = copy.deepcopy(state)
    for entry in out['state'].values():
        entry['exp_avg'] = interpolate(entry['exp_avg'], shape, mode='bicubic')
        for key in ('exp_avg_sq', 'max_exp_avg_sq'):
            if key in entry:
                entry[key] = interpolate(entry[key], shape, mode='bilinear').relu_()
    return out


def to_tensor(pil_image):
    """PIL RGB -> float CHW in [0,1] (what torchvision's TF.to_tensor does for uint8 images)."""
    arr = np.asarray(pil_image.convert('RGB'), dtype=np.uint8)
    return torch.from_numpy(arr.copy()).permute(2, 0, 1).contiguous().to(torch.float32).div(255)


def to_pil_image(tensor):
    """float CHW in [0,1] -> PIL RGB; mul(255).byte() truncation like torchvision's to_pil_image."""
    arr = tensor.detach().cpu().mul(255).byte().permute(1, 2, 0).numpy()
    return Image.fromarray(arr, 'RGB')


@dataclass
class STIterate:
    w: int
    h: int
    i: int
    i_max: int
    loss: float
    time: float
    gpu_ram: int


class EMA:
    """Bias-corrected exponential moving average of the iterates (reference :237-253).

    ``value`` lives on the device and is advanced inside the fused HIP step; ``accum`` and ``decay``
    are fp32 scalars kept on the host with the same fp32 arithmetic as the reference's buffers."""

    def __init__(self, input, decay):
        self.value = torch.zeros_like(input)
        self.decay = torch.tensor(decay, dtype=torch.float32)
        self.accum = torch.tensor(1., dtype=torch.float32)
        self.update(input)

    def get(self):
        return self.value / (1 - self.accum).to(self.value.device)

    def advance_accum(self):
        self.accum = self.accum * self.decay

    def update(self, input):
        self.advance_accum()
        d = self.decay.to(self.value.device)
        self.value *= d
        self.value += (1 - d) * input.detach()


class AdamState:
    """exp_avg / exp_avg_sq / step of torch.optim.Adam for the single image tensor."""

    def __init__(self, image):
        self.exp_avg = torch.zeros_like(image)
        self.exp_avg_sq = torch.zeros_like(image)
        self.step = 0

    def rescaled(self, shape):
        """Warm start at a new scale (reference scale_adam, :285-295): moments resampled, step kept."""
        new = AdamState.__new__(AdamState)
        new.exp_avg = interpolate(self.exp_avg, shape, mode='bicubic').contiguous()
        new.exp_avg_sq = interpolate(self.exp_avg_sq, shape, mode='bilinear').relu_().contiguous()
        new.step = self.step
        return new


def _dist_info():
    """(rank, world) of the default torch.distributed group; (0, 1) when not initialised."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _gather_rows(strip, rows, rank):
    """Full [1, C, H, W] tensor on EVERY rank from the ranks' row strips (cold path: once per scale).
    One broadcast per rank, because the strips have different heights."""
    import torch.distributed as dist
    parts = []
    for r, (b, e) in enumerate(rows):
        buf = strip.contiguous() if r == rank else strip.new_empty((strip.shape[0], strip.shape[1], e - b, strip.shape[3]))
        if strip.is_cuda:
            torch.cuda.synchronize(strip.device)          # gloo on device tensors is not stream-ordered
        dist.broadcast(buf, src=r)
        parts.append(buf)
    return torch.cat(parts, dim=2)


def _starting_image(init, content_image, style_images, style_weights, height, width):
    """The iterate the first scale starts from, [1, 3, height, width] on the host (reference style_transfer.py:380-406).
    The random modes draw from torch's global generator in the reference's order and with its calls, so that
    torch.manual_seed(n) (the CLI's --random-seed) reproduces the reference's starting point draw for draw:
    'gray' one rand of the full shape, 'uniform' the same, 'normal' one trunc_normal_ of the full shape,
    'style_stats' one trunc_normal_ per colour channel (R, G, B in turn)."""
    shape = [1, 3, height, width]
    if init == 'content':
        return to_tensor(content_image.resize((width, height), Image.BICUBIC))[None]
    if init == 'gray':                                        # mid-gray plus at most one 8-bit step of noise
        return torch.rand(shape) / 255 + 0.5
    if init == 'uniform':
        return torch.rand(shape)
    if init == 'normal':
        out = torch.empty(shape)
        torch.nn.init.trunc_normal_(out, mean=0.5, std=0.25, a=0, b=1)
        return out
    if init == 'style_stats':
        # per-channel mean and (unbiased) variance of every style image, blended with the style weights
        blend_mean, blend_var = torch.zeros(3), torch.zeros(3)
        for pil, weight in zip(style_images, style_weights):
            pixels = to_tensor(pil)
            blend_mean = blend_mean + pixels.mean(dim=(1, 2)) * weight
            blend_var = blend_var + pixels.var(dim=(1, 2)) * weight
        planes = []
        for c in range(3):
            plane = torch.empty([1, 1, height, width])
            torch.nn.init.trunc_normal_(plane, mean=blend_mean[c], std=blend_var[c].sqrt(), a=0, b=1)
            planes.append(plane)
        return torch.cat(planes, dim=1)
    raise ValueError("init must be one of 'content', 'gray', 'uniform', 'style_mean'")


def _resolve_weights(weights):
    if isinstance(weights, (list, tuple)):
        return list(weights)
    if isinstance(weights, str) and weights.startswith('synthetic'):
        seed = int(weights.split(':')[1]) if ':' in weights else 0
        return vgg.synthetic_vgg19_weights(seed)
    candidates = []
    if isinstance(weights, str):
        candidates.append(weights)
    if os.environ.get('STYLE_TRANSFER_VGG19'):
        candidates.append(os.environ['STYLE_TRANSFER_VGG19'])
    candidates.append(os.path.expanduser('~/.cache/torch/hub/checkpoints/vgg19-dcbb9e9d.pth'))
    for path in candidates:
        if os.path.exists(path):
            return vgg.load_weights(path)
    raise FileNotFoundError(
        'VGG-19 weights not found. Pass weights=<path to torchvision vgg19-dcbb9e9d.pth>, set '
        "STYLE_TRANSFER_VGG19, or use weights='synthetic' (seeded random weights, tests/benchmarks only).")


class VGGFeatures:
    """Feature extractor facade (reference :20-90) over the HIP trunk; keeps one plan per input size."""

    def __init__(self, layers, pooling='max', weights=None, device='cuda:0', precision='fp16x3'):
        if pooling not in vgg.POOLINGS:
            raise KeyError(pooling)
        self.layers = sorted(set(layers))
        self.pooling = pooling
        self.device = torch.device(device)
        self.net = _hip.Net(_resolve_weights(weights), pooling, self.device, precision)
        self._plans = {}

    def plan_for(self, height, width):
        key = (int(height), int(width))
        if key not in self._plans:
            if len(self._plans) >= 4:          # style images of a few sizes; keep the cache small
                self._plans.pop(next(iter(self._plans)))
            self._plans[key] = _hip.Plan(self.net, *key)
        return self._plans[key]

    def drop_plans(self):
        self._plans.clear()

    def __call__(self, input, layers=None):
        layers = self.layers if layers is None else sorted(set(layers))
        h, w = input.shape[2:4]
        min_size = vgg.min_size_for(layers)
        if min(h, w) < min_size:
            raise ValueError(f'Input is {h}x{w} but must be at least {min_size}x{min_size}')
        x = input.detach().to(self.device, torch.float32).contiguous()
        plan = self.plan_for(h, w)
        plan.forward(x, max(layers))
        feats = {'input': input}
        for layer in layers:
            feats[layer] = plan.feature(layer)
        return feats

    forward = __call__


class _DeviceListJob:
    """Control channel of a device-list stylize() (StyleTransfer(devices=[d0, d1, ...]) in ONE process, reference :326-333).

    Rank 0 is the calling process; every other device has a worker process running the same stylize() on its own strip.
    Callbacks fire on rank 0 only - the workers wait at the same iteration for a command on a gloo side group: 0 = go on,
    1 = rank 0's callback asked for the image (get_image / get_image_tensor gather the strips: a collective), 2 = rank 0 was
    interrupted (gather the averaged iterate once more, then everybody leaves), 3 = rank 0's callback failed (leave at once).
    Without a callback there is no per-iteration traffic at all.

    Ctrl-C (reference cli.py:261-266 keeps the image): the workers ignore SIGINT - a terminal delivers it to the whole
    foreground process group - and rank 0 alone decides: inside the callback KeyboardInterrupt is salvaged directly, anywhere
    else the signal only sets `interrupted`, which the next iteration's rendez-vous turns into the same salvage (advisor
    finding of round 5)."""

    def __init__(self, rank, group, st=None):
        self.rank, self.group, self.st = rank, group, st
        self.in_callback = False
        self.interrupted = False
        self.stopped = False

    def _send(self, value):
        import torch.distributed as dist
        dist.broadcast(torch.tensor([value], dtype=torch.int32), src=0, group=self.group)

    def request_gather(self):
        if not self.in_callback:
            raise RuntimeError('StyleTransfer(devices=[...]): while stylize() runs, get_image() / get_image_tensor() may only '
                               'be called from the callback (the strips live in the worker processes)')
        self._send(1)

    def rank0_callback(self, user_callback):
        def fire(it):
            self.in_callback = True
            clean = False
            try:
                user_callback(it)
                if self.interrupted:                     # SIGINT arrived between two rendez-vous
                    raise KeyboardInterrupt
                clean = True
            except KeyboardInterrupt:
                # cli.py:261-266: Ctrl-C keeps what has been computed.  The workers wait at this iteration's rendez-vous:
                # command 2 = gather the averaged iterate once more, then everybody leaves stylize()
                st = self.st
                if st is not None and st.average is not None:
                    self._send(2)
                    local = st.average.get().detach()
                    if st._strip_rows is not None:
                        local = _gather_rows(local, *st._strip_rows)
                    st.image, st.average, st._strip_rows = local, None, None
                    self.stopped = True
                raise
            finally:
                self.in_callback = False
                if not self.stopped:
                    # 0: go on.  3: the callback raised something else - the workers must not run on into collectives
                    # that rank 0 will never join
                    self._send(0 if clean else 3)
                    self.stopped = not clean
        return fire

    def worker_callback(self, st):
        import torch.distributed as dist

        def fire(_it):
            while True:
                cmd = torch.zeros(1, dtype=torch.int32)
                dist.broadcast(cmd, src=0, group=self.group)
                cmd = int(cmd.item())
                if cmd == 0:
                    return
                if cmd == 3:
                    raise _StopDeviceListJob          # rank 0's callback failed: leave without another collective
                if st._strip_rows is not None:
                    st.get_image_tensor()             # joins rank 0's gather
                if cmd == 2:
                    raise _StopDeviceListJob          # rank 0 was interrupted: leave stylize() with it
        return fire


class _StopDeviceListJob(Exception):
    pass


def _device_list_backend(devices):
    """RCCL when every rank has a GPU of its own; a list that names one device twice (tests, a one-GPU box) runs over gloo -
    RCCL refuses two ranks on one device - which is functional, not fast."""
    return 'nccl' if len({str(d) for d in devices}) == len(devices) else 'gloo'


def _device_list_timeout():
    """Seconds a rendez-vous / a gloo collective of the device-list job may take before it fails (a rank that died early must
    not leave the others waiting for the backend's default of 30 minutes)."""
    import datetime
    return datetime.timedelta(seconds=int(os.environ.get('ST_DEVICE_LIST_TIMEOUT', 300)))


def _device_list_worker(rank, world, port, backend, device, ctor, content_image, style_images, kw, has_callback, failures):
    try:
        import signal
        signal.signal(signal.SIGINT, signal.SIG_IGN)         # rank 0 decides (see _DeviceListJob)
        import torch.distributed as dist
        device = torch.device(device)
        torch.cuda.set_device(device)
        init = dict(init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world, timeout=_device_list_timeout())
        if backend == 'nccl':
            init['device_id'] = device
        dist.init_process_group(backend, **init)
        ctl = dist.new_group(backend='gloo')
        if os.environ.get('ST_DEVICE_LIST_INJECT_FAILURE') == str(rank):     # (tests: a worker that dies after the rendez-vous)
            raise RuntimeError('injected worker failure')
        st = StyleTransfer(devices=[device], **ctor)
        st._job = _DeviceListJob(rank, ctl)
        try:
            st.stylize(content_image, style_images, callback=st._job.worker_callback(st) if has_callback else None, **kw)
            torch.cuda.synchronize(device)
            dist.barrier()
        except _StopDeviceListJob:
            torch.cuda.synchronize(device)
        dist.destroy_process_group()
    except BaseException as exc:                             # noqa: BLE001 - reported to rank 0, which raises
        import traceback
        failures.put((rank, f'{type(exc).__name__}: {exc}\n{traceback.format_exc()}'))
        raise


def _device_list_stylize(st, content_image, style_images, kw, callback):
    """stylize() of a StyleTransfer built with a device list: one process per device, this one is rank 0.

    The reference's two-device form (style_transfer.py:326-333, cli.py:214-223) needs no launcher - neither does this: the
    extra ranks are spawned here, joined afterwards, and the result is what `torchrun --nproc-per-node N` gives for the same
    call (it is the same code: strip plans, halo exchange + Gram reductions over RCCL, shard-aware scale transitions)."""
    import socket
    import torch.distributed as dist
    import torch.multiprocessing as mp
    if dist.is_available() and dist.is_initialized():
        raise RuntimeError('StyleTransfer(devices=[...several...]) starts its own ranks; under torchrun pass this rank\'s device only')
    world = len(st.devices)
    backend = _device_list_backend(st.devices)
    if backend == 'nccl' and torch.cuda.device_count() < world:
        raise RuntimeError(f'{world} devices named, {torch.cuda.device_count()} HIP device(s) visible')
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context('spawn')
    failures = ctx.SimpleQueue()
    env_keep = {k: os.environ.get(k) for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')}
    for k in env_keep:
        os.environ.pop(k, None)                              # (a stale launcher environment must not reach the workers)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = [ctx.Process(target=_device_list_worker, daemon=True,
                         args=(r, world, port, backend, str(st.devices[r]), st._ctor, content_image, style_images, kw,
                               callback is not None, failures))
             for r in range(1, world)]
    for p in procs:
        p.start()
    device = st.devices[0]
    result, error = None, None
    # Watchdog (advisor finding of round 5): a worker that dies early - bad device, out of memory, import error - would leave
    # rank 0 waiting in the rendez-vous or in a collective.  A thread watches the workers; when one has died it aborts the
    # transports rank 0 may be blocked in (the in-library RCCL communicators, torch's RCCL group; a gloo peer's death closes
    # its sockets, which fails the pending operation by itself), and the worker's traceback is what the caller gets.
    import signal
    import threading
    watch = {'stop': False, 'dead': None}

    def watchdog():
        while not watch['stop']:
            dead = [p for p in procs if p.exitcode not in (None, 0)]
            if dead or not failures.empty():
                watch['dead'] = [p.exitcode for p in dead]
                fabric = getattr(st, '_fabric', None)
                try:
                    if fabric is not None and hasattr(fabric, 'close'):
                        fabric.close(abort=True)
                    if backend == 'nccl' and dist.is_initialized():
                        dist.distributed_c10d._abort_process_group()
                except Exception:                            # noqa: BLE001 - best effort: the timeout still bounds the wait
                    pass
                return
            time.sleep(0.25)
    threading.Thread(target=watchdog, daemon=True).start()
    prev_sigint = None
    try:
        torch.cuda.set_device(device)
        init = dict(init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=world, timeout=_device_list_timeout())
        if backend == 'nccl':
            init['device_id'] = device
        dist.init_process_group(backend, **init)
        st._job = _DeviceListJob(0, dist.new_group(backend='gloo'), st)
        if callback is not None and threading.current_thread() is threading.main_thread():
            job = st._job

            def on_sigint(signum, frame):
                if job.in_callback:
                    raise KeyboardInterrupt                  # inside the user's callback: as without this handler
                job.interrupted = True                       # elsewhere: at the next iteration's rendez-vous
            prev_sigint = signal.signal(signal.SIGINT, on_sigint)
        result = st.stylize(content_image, style_images,
                            callback=st._job.rank0_callback(callback) if callback is not None else None, **kw)
        torch.cuda.synchronize(device)
        dist.barrier()
    except BaseException as exc:                             # noqa: BLE001 - re-raised below, after the workers are dealt with
        error = exc
        if st._strip_rows is not None:
            # not salvaged (the exception did not come out of the callback): get_image() must not try a gather without
            # workers - it returns this rank's rows of the averaged iterate
            warnings.warn('device-list stylize() left early: get_image() holds only the first device\'s rows')
            st.image, st.average, st._strip_rows = st.average.get().detach(), None, None
    finally:
        watch['stop'] = True
        if prev_sigint is not None:
            signal.signal(signal.SIGINT, prev_sigint)
        st._job = None
        for k, v in env_keep.items():
            if v is not None:
                os.environ[k] = v
        try:
            if dist.is_initialized():
                from . import sharding
                sharding.release_head_groups()
                dist.destroy_process_group()
        except Exception:                                    # noqa: BLE001
            pass
        for p in procs:
            p.join(timeout=60 if error is None else 5)
            if p.is_alive():
                p.terminate()
    notes = []
    while not failures.empty():
        notes.append('rank %d: %s' % failures.get())
    bad = [p.exitcode for p in procs if p.exitcode not in (0, None)]
    if (notes or watch['dead']) and not isinstance(error, KeyboardInterrupt):
        # the worker's failure is the cause; what rank 0 raised when its transport was aborted under it is the consequence
        raise RuntimeError('a worker of the device-list stylize() failed: ' + ('; '.join(notes) or f'exit codes {bad}')) from error
    if error is not None:
        raise error
    if bad:
        raise RuntimeError(f'a worker of the device-list stylize() failed: exit codes {bad}')
    return result


class StyleTransfer:
    def __init__(self, devices=['cuda:0'], pooling='max', weights=None, precision='fp16x3'):
        # reference :310 defaults to ['cpu']; this build has no CPU path, so the default is the first HIP device
        self.devices = [torch.device(device) for device in devices]
        self.image = None
        self.average = None

        self.content_layers = list(CONTENT_LAYERS)
        self.style_layers = list(STYLE_LAYERS)
        style_weights = [256, 64, 16, 4, 1]                       # reference :320-322
        weight_sum = sum(abs(w) for w in style_weights)
        self.style_weights = [w / weight_sum for w in style_weights]

        if not 1 <= len(self.devices) <= 8:
            raise ValueError('Only 1 to 8 devices are supported.')           # reference :331: "Only 1 or 2 devices ..."
        if any(d.type != 'cuda' for d in self.devices):
            raise ValueError('This build runs on MI355X only: pass HIP devices (e.g. devices=["cuda:0"]); '
                             'there is no CPU path.')
        # A device LIST (reference :326-333: VGG-19's layers split over two devices, to fit a 24 GB card) is the one-process
        # form of the strip sharding here: stylize() cuts the image into one row strip per device and runs one worker PROCESS
        # per extra device (the MI355X model: one process per GPU over RCCL - the same code path as a torchrun launch), this
        # process being rank 0.  See _device_list_stylize.  The model below serves the single-device calls and rank 0.
        self._ctor = dict(pooling=pooling, weights=weights, precision=precision)
        self._job = None             # the device-list job while its stylize() runs
        # precision: arithmetic of the 3x3 trunk convolutions - 'fp16x3' (default: scaled fp16 planes, fp32-class
        # accuracy, meets the fp32 parity bar), 'bf16x6' (same accuracy, twice the matrix work), 'fp32' (exact
        # fp32 MFMA) or 'bf16x3' (approximate)
        self.model = VGGFeatures(self.style_layers + self.content_layers, pooling=pooling, weights=weights,
                                 device=self.devices[0], precision=precision)
        self._plan = None
        self._strip_rows = None      # (rows, rank) while a strip-sharded scale is running

    # ---- results (reference :335-347) ----
    def get_image_tensor(self):
        """The averaged iterate, 3 x H x W on the compute device (reference :335-336).  While a strip-sharded scale is
        running (one process per GPU) every rank holds only its rows: the strips are gathered here, which makes the
        call a COLLECTIVE in that situation - callbacks fire on every rank at the same iteration, so a callback that
        calls it on every rank (as the CLI's does) is safe; calling it on one rank only would hang."""
        if self.average is None:                      # stylize() has returned: self.image is the gathered result
            return self.image.detach()[0].clamp(0, 1)
        local = self.average.get().detach()
        if self._strip_rows is not None:
            rows, rank = self._strip_rows
            if self._job is not None and rank == 0:
                self._job.request_gather()            # device-list form: the workers join the gather (see _DeviceListJob)
            local = _gather_rows(local, rows, rank)
        return local[0].clamp(0, 1)

    def get_image(self, image_type='pil'):
        if self.average is not None or self.image is not None:
            image = self.get_image_tensor()
            if image_type.lower() == 'pil':
                return to_pil_image(image)
            elif image_type.lower() == 'np_uint16':
                arr = image.cpu().movedim(0, 2).numpy()
                return np.uint16(np.round(arr * 65535))
            else:
                raise ValueError("image_type must be 'pil' or 'np_uint16'")

    # ---- per-scale targets (reference :425-453) ----
    def _build_targets(self, plan, content, style_images, style_weights, scale, style_scale_fac, style_size):
        device = self.devices[0]
        # fp16x3's activation-aware range guard (cold path, st_plan_range_guard): the forward arithmetic is checked on every
        # image BEFORE its features become targets; the closure's gradients are checked in stylize() once the targets exist
        guarded = self.model.net.precision == 'fp16x3'
        if guarded:
            plan.range_guard(content)
        plan.forward(content, 22)
        plan.set_content_target_from_forward()
        blended = {}
        for i, image in enumerate(style_images):
            if style_size is None:
                sw, sh = size_to_fit(image.size, round(scale * style_scale_fac))
            else:
                sw, sh = size_to_fit(image.size, style_size)
            style = to_tensor(image.resize((sw, sh), Image.BICUBIC))[None].to(device)
            print(f'Processing style image ({sw}x{sh})...')
            if min(sh, sw) < 16:
                raise ValueError(f'Input is {sh}x{sw} but must be at least 16x16')
            splan = plan if (sh, sw) == (plan.height, plan.width) else self.model.plan_for(sh, sw)
            if guarded:
                splan.range_guard(style)
            splan.forward(style, 29)
            for layer in self.style_layers:
                mean, srm = splan.moments(layer)
                mean *= style_weights[i]
                srm *= style_weights[i]
                if layer not in blended:
                    blended[layer] = [mean, srm]
                else:
                    blended[layer][0].add_(mean)
                    blended[layer][1].add_(srm)
        for idx, layer in enumerate(self.style_layers):
            plan.set_style_target(idx, *blended[layer])

    def _guard_rows(self, rows_image, blended=None, content_weight=None, tv_weight=None):
        """st_plan_range_guard on a block of image rows taken as an image of its own (a whole-image plan of that size).
        With ``blended`` (the scale's style targets) the data gradients of one closure are checked too; the content target is
        the block's own forward.  Returns the (forward, backward) layers this call flagged."""
        device = self.devices[0]
        h, w = rows_image.shape[2:]
        probe = _hip.Plan(self.model.net, h, w)
        if blended is not None:
            probe.forward(rows_image, 22)
            probe.set_content_target_from_forward()
            for idx, layer in enumerate(self.style_layers):
                probe.set_style_target(idx, blended[layer][0].contiguous(), blended[layer][1].contiguous())
            probe.set_loss_weights(content_weight, self.style_weights, tv_weight)
        flagged = probe.range_guard(rows_image)
        del probe
        torch.cuda.synchronize(device)
        return flagged

    def _agree_on_wide_layers(self, fabric):
        """The union over the ranks of the layers the range guard has moved to bf16x6: every rank then computes a layer in the
        same arithmetic (a strip's halo rows come from its neighbour's kernels).  Returns True if this rank gained a layer."""
        net = self.model.net
        fwd, bwd = net.wide_layers()
        flags = torch.tensor(fwd + bwd, dtype=torch.float32, device=self.devices[0])
        fabric.allmax(flags)
        merged = [int(v) for v in flags.tolist()]
        net.mark_wide(merged[:13], merged[13:])
        return merged != fwd + bwd

    @staticmethod
    def _style_rows(plan, rows, sh, sw, world):
        """Strips of a style image's forward pass: the optimised image's own strips when the sizes agree (its plan is reused),
        an even split otherwise (a forward pass has no chain owner to relieve)."""
        from . import sharding
        if (sh, sw) == (plan.global_height, plan.width):
            return rows
        return sharding.strip_rows(sh, world)

    def _build_targets_sharded(self, plan, fabric, content, rows, rank, style_images, style_weights, scale,
                               style_scale_fac, style_size):
        """Per-scale targets when the image is cut into row strips: the content target
        from this rank's strip; every style image is cut into its OWN strips (its size is independent of the
        content's) and its raw moment sums are all-reduced - or, when it is too small to give every rank 16
        rows, evaluated whole on every rank (identical kernels on identical inputs: identical results)."""
        from . import sharding
        device, world = self.devices[0], len(rows)
        b, e = rows[rank]
        guarded = self.model.net.precision == 'fp16x3'
        if guarded:
            # the activation-aware range guard of a sharded scale (st_plan_range_guard works on whole-image plans): every rank
            # checks ITS rows as an image of their own - the same kernels on the same statistics, replicate padding where the
            # strip has neighbours - and the ranks take the union of their verdicts before anything becomes a target
            self._guard_rows(content[:, :, b:e].contiguous().to(device))
            for image in style_images:
                if style_size is None:
                    sw, sh = size_to_fit(image.size, round(scale * style_scale_fac))
                else:
                    sw, sh = size_to_fit(image.size, style_size)
                if min(sh, sw) >= 16 and sh // 16 >= world:
                    sb, se = self._style_rows(plan, rows, sh, sw, world)[rank]
                    style = to_tensor(image.resize((sw, sh), Image.BICUBIC))[None]
                    self._guard_rows(style[:, :, sb:se].contiguous().to(device))
            self._agree_on_wide_layers(fabric)
        plan.forward_begin(content[:, :, b:e].contiguous().to(device), 22)
        sharding.run_phases(plan, fabric)
        plan.set_content_target_from_forward()
        blended = {}
        for i, image in enumerate(style_images):
            if style_size is None:
                sw, sh = size_to_fit(image.size, round(scale * style_scale_fac))
            else:
                sw, sh = size_to_fit(image.size, style_size)
            style = to_tensor(image.resize((sw, sh), Image.BICUBIC))[None]
            if rank == 0:
                print(f'Processing style image ({sw}x{sh})...')
            if min(sh, sw) < 16:
                raise ValueError(f'Input is {sh}x{sw} but must be at least 16x16')
            if sh // 16 >= world:
                sb, se = self._style_rows(plan, rows, sh, sw, world)[rank]
                sp = plan if (sh, sw, sb, se) == (plan.global_height, plan.width, b, e) else \
                    sharding.StripPlan(self.model.net, sh, sw, sb, se)
                sp.forward_begin(style[:, :, sb:se].contiguous().to(device), 29)
                sharding.run_phases(sp, fabric)
            else:
                sp = self.model.plan_for(sh, sw)
                sp.forward(style.to(device), 29)
            for level, layer in enumerate(self.style_layers):
                if isinstance(sp, sharding.StripPlan):
                    sums = sp.moment_sums(layer)
                    fabric.allreduce(sums)
                    c = sums.numel()
                    c = int(round((-1 + (1 + 4 * c) ** 0.5) / 2))             # c*c + c entries
                    npix = float((sh >> level) * (sw >> level))
                    srm, mean = (sums[:c * c] / npix).reshape(c, c), sums[c * c:] / npix
                else:
                    mean, srm = sp.moments(layer)
                mean, srm = mean * style_weights[i], srm * style_weights[i]
                if layer not in blended:
                    blended[layer] = [mean, srm]
                else:
                    blended[layer][0] += mean
                    blended[layer][1] += srm
            if sp is not plan:
                del sp
                self.model.drop_plans()
        for idx, layer in enumerate(self.style_layers):
            plan.set_style_target(idx, blended[layer][0].contiguous(), blended[layer][1].contiguous())
        return blended

    def stylize(self, content_image, style_images, *,
                style_weights=None,
                content_weight: float = 0.015,
                tv_weight: float = 2.,
                optimizer: str = 'adam',
                min_scale: int = 128,
                end_scale: int = 512,
                iterations: int = 500,
                initial_iterations: int = 1000,
                step_size: float = 0.02,
                avg_decay: float = 0.99,
                init: str = 'content',
                style_scale_fac: float = 1.,
                style_size: int = None,
                callback=None):

        if len(self.devices) > 1 and self._job is None:
            return _device_list_stylize(self, content_image, style_images, dict(
                style_weights=style_weights, content_weight=content_weight, tv_weight=tv_weight, optimizer=optimizer,
                min_scale=min_scale, end_scale=end_scale, iterations=iterations, initial_iterations=initial_iterations,
                step_size=step_size, avg_decay=avg_decay, init=init, style_scale_fac=style_scale_fac, style_size=style_size),
                callback)
        min_scale = min(min_scale, end_scale)
        content_weights = [content_weight / len(self.content_layers)] * len(self.content_layers)

        if style_weights is None:
            style_weights = [1 / len(style_images)] * len(style_images)
        else:
            weight_sum = sum(abs(w) for w in style_weights)
            style_weights = [weight / weight_sum for weight in style_weights]
        if len(style_images) != len(style_weights):
            raise ValueError('style_images and style_weights must have the same length')
        if optimizer not in ('adam', 'lbfgs'):
            raise ValueError("optimizer must be one of 'adam', 'lbfgs'")

        device = self.devices[0]
        scales = gen_scales(min_scale, end_scale)
        # One process per GPU under torch.distributed: the image, the Adam/EMA state and every feature map are cut
        # into row strips (sharding.py) and STAY cut from scale to scale (sharding.resample_strip moves only the few
        # neighbour rows a strip's resample needs); self.image / self.average hold this rank's strip while a sharded
        # scale runs and the gathered full image once the last scale is done.  get_image_tensor() / get_image()
        # gather on demand (a collective: call them on every rank).
        rank, world = _dist_info()
        if world > 1:
            from . import sharding
            import torch.distributed as dist
            # ST_FABRIC_HOST_SYNC=1: the conservative transport (what bench.py falls back to, and what the gloo tests
            # run) - every exchange a host-synchronised step, whole convolution launches behind their halos, every
            # rank running every head's chains on all-reduced moments.  The way out if the stream-ordered RCCL path
            # misbehaves on a system: slower, no cross-stream ordering to get wrong.  INTEGRATION.md section 5.
            conservative = os.environ.get('ST_FABRIC_HOST_SYNC') == '1'
            if conservative:
                _hip.set_option('ST_STRIP_OVERLAP', 0)
                _hip.set_option('ST_STRIP_NS_OWNER', 0)
            fabric = sharding.DistFabric(rank, world, host_sync=True if conservative else None)
            # nccl (= RCCL) backend: the closure's exchanges go through the in-library transport (csrc/st_fabric.hip: RCCL
            # operations issued by the library on its own streams, one call per closure); torch.distributed keeps the cold
            # path.  ST_FABRIC_NATIVE=0: torch.distributed for everything (the descriptor form).
            if not conservative and dist.get_backend() == 'nccl' and os.environ.get('ST_FABRIC_NATIVE') != '0':
                try:
                    fabric = sharding.NativeFabric(rank, world, device, cold=fabric)
                except RuntimeError as exc:     # the pre-flight's verdict is agreed over the ranks: all of them land here
                    warnings.warn(f'{exc}; the exchanges go through torch.distributed instead')

        cw, ch = size_to_fit(content_image.size, scales[0], scale_up=True)
        self.image = _starting_image(init, content_image, style_images, style_weights, ch, cw)
        self.image = self.image.to(device)
        if world > 1:
            torch.cuda.synchronize(device)
            dist.broadcast(self.image, src=0)              # the random inits must agree across ranks

        adam = None
        prev_rows, prev_h = None, None       # row strips / height of the previous scale (None: held whole on every rank)
        # ST_STYLIZE_TIMING=1 (measurement aid): per scale, seconds of setup (resample, plan, targets, range guard) and of the
        # iteration loop, each closed by a device synchronisation; read them from self.timing afterwards
        timing = os.environ.get('ST_STYLIZE_TIMING') == '1'
        self.timing = []
        for scale in scales:
            if timing:
                torch.cuda.synchronize(device)
                t_scale = time.perf_counter()
            cw, ch = size_to_fit(content_image.size, scale, scale_up=True)
            content = to_tensor(content_image.resize((cw, ch), Image.BICUBIC))[None]

            # strips need >= 16 rows per rank; a smaller scale runs whole on every rank (same kernels on the same
            # inputs: every rank holds the same result, no exchange needed)
            sharded = world > 1 and ch // 16 >= world
            rows = sharding.strip_rows(ch, world, cw) if sharded else None
            if sharded:
                # shard-aware scale transition (reference :279-295,420,460-462): every rank resamples only its own
                # rows of the image and of the two Adam moments; the few source rows it needs from its neighbours'
                # strips travel point to point (sharding.resample_strip) - nothing is gathered
                b, e = rows[rank]
                old_h = prev_h if prev_h is not None else self.image.shape[2]

                def strip_of(t, mode):
                    return sharding.resample_strip(t.detach(), prev_rows, rank, world, old_h, rows, (ch, cw), mode,
                                                   host_sync=fabric.host_sync)
                self.image = strip_of(self.image, 'bicubic').clamp(0, 1).contiguous()
                if optimizer == 'adam':
                    if adam is None:
                        adam = AdamState(self.image)
                    else:
                        new = AdamState.__new__(AdamState)
                        new.exp_avg = strip_of(adam.exp_avg, 'bicubic')
                        new.exp_avg_sq = strip_of(adam.exp_avg_sq, 'bilinear').relu_().contiguous()
                        new.step = adam.step
                        adam = new
            else:
                if prev_rows is not None:                   # a smaller scale after a sharded one: whole again
                    self.image = _gather_rows(self.image, prev_rows, rank)
                    if adam is not None:
                        adam.exp_avg = _gather_rows(adam.exp_avg, prev_rows, rank)
                        adam.exp_avg_sq = _gather_rows(adam.exp_avg_sq, prev_rows, rank)
                self.image = interpolate(self.image.detach(), (ch, cw), mode='bicubic').clamp(0, 1).contiguous()
                if optimizer == 'adam':
                    adam = AdamState(self.image) if adam is None else adam.rescaled((ch, cw))
            prev_rows, prev_h = rows, ch
            self.average = EMA(self.image, avg_decay)
            self._strip_rows = (rows, rank) if sharded else None

            if rank == 0:
                print(f'Processing content image ({cw}x{ch})...')
            plan = closure = opt = None                    # free the previous scale's plan BEFORE allocating the next
            self._plan = None
            self.model.drop_plans()
            torch.cuda.empty_cache()
            if sharded:
                plan = self._plan = sharding.StripPlan(self.model.net, ch, cw, b, e).set_rank(rank, world)
                blended = self._build_targets_sharded(plan, fabric, content, rows, rank, style_images, style_weights, scale,
                                                      style_scale_fac, style_size)
                grad = torch.empty_like(self.image)
            else:
                plan = self._plan = _hip.Plan(self.model.net, ch, cw)
                self._build_targets(plan, content.to(device), style_images, style_weights, scale, style_scale_fac,
                                    style_size)
            plan.set_loss_weights(content_weights[0], self.style_weights, tv_weight)
            if sharded and self.model.net.precision == 'fp16x3':
                # ... sharded: on this rank's rows of the iterate, against the scale's style targets; the union of the ranks'
                # verdicts; forward layers flagged only now have shaped the targets: build them again
                before = self.model.net.wide_layers()
                self._guard_rows(self.image.detach(), blended, content_weights[0], tv_weight)
                self._agree_on_wide_layers(fabric)
                after = self.model.net.wide_layers()
                if after[0] != before[0]:
                    blended = self._build_targets_sharded(plan, fabric, content, rows, rank, style_images, style_weights,
                                                          scale, style_scale_fac, style_size)
                if rank == 0 and after != before:
                    print(f'fp16x3 range guard: bf16x6 for the forward of convs {[i for i, v in enumerate(after[0]) if v]} '
                          f'and the data gradient of convs {[i for i, v in enumerate(after[1]) if v]} from now on')
            if not sharded and self.model.net.precision == 'fp16x3':
                # ... and on the iterate itself, forward and data gradients (one extra closure per scale).  A forward layer
                # flagged only now has already shaped the targets: build them again in the corrected arithmetic.
                fwd, bwd = plan.range_guard(self.image)
                if any(fwd):
                    self._build_targets(plan, content.to(device), style_images, style_weights, scale, style_scale_fac,
                                        style_size)
                if any(fwd) or any(bwd):
                    print(f'fp16x3 range guard: bf16x6 for the forward of convs {[i for i, v in enumerate(fwd) if v]} and '
                          f'the data gradient of convs {[i for i, v in enumerate(bwd) if v]} from now on')
            self.model.drop_plans()

            if optimizer != 'adam' and sharded:
                # torch.optim.LBFGS(max_iter=1, history_size=10) with its inner products completed over the ranks
                opt = sharding.StripLBFGS(self.image, grad, fabric.allreduce, fabric.allmax, history_size=10)

                def closure(plan=plan, grad=grad):
                    plan.closure_begin(self.image, grad)
                    sharding.run_phases(plan, fabric)
                    return plan.losses[7].clone()
            elif optimizer != 'adam':
                self.image.requires_grad_()
                opt = torch.optim.LBFGS([self.image], max_iter=1, history_size=10)

                def closure(plan=plan):
                    with torch.no_grad():
                        losses, grad = plan.loss_and_grad(self.image.detach())
                    self.image.grad = grad
                    return losses[7].clone()

            actual_its = initial_iterations if scale == scales[0] else iterations
            if timing:
                torch.cuda.synchronize(device)
                t_loop = time.perf_counter()
            for i in range(1, actual_its + 1):
                if optimizer == 'adam' and sharded:
                    adam.step += 1
                    plan.closure_begin(self.image, grad)
                    sharding.run_phases(plan, fabric)
                    plan.apply_update(self.image, grad, adam.exp_avg, adam.exp_avg_sq, self.average.value,
                                      adam.step, step_size, 0.9, 0.99, 1e-8, avg_decay)
                    self.average.advance_accum()
                    loss = plan.losses[7]
                elif optimizer == 'adam':
                    adam.step += 1
                    losses = plan.step(self.image, adam.exp_avg, adam.exp_avg_sq, self.average.value,
                                       adam.step, step_size, 0.9, 0.99, 1e-8, avg_decay)
                    self.average.advance_accum()
                    loss = losses[7]
                else:
                    loss = opt.step(closure)                # no clamp for L-BFGS (reference :482-483)
                    self.average.update(self.image)
                if callback is not None:
                    gpu_ram = torch.cuda.max_memory_allocated(device) + plan.device_bytes()
                    callback(STIterate(w=cw, h=ch, i=i, i_max=actual_its, loss=loss.item(),
                                       time=time.time(), gpu_ram=gpu_ram))

            if timing:
                torch.cuda.synchronize(device)
                self.timing.append({'scale': scale, 'size': (cw, ch), 'iterations': actual_its,
                                    'setup_s': t_loop - t_scale, 'loop_s': time.perf_counter() - t_loop})
            # Initialize each new scale with the previous scale's averaged iterate (reference :496-497)
            with torch.no_grad():
                self.image = self.image.detach()
                self.image.copy_(self.average.get())
                if sharded and scale == scales[-1]:         # the result: the full image on every rank
                    self.image = _gather_rows(self.image, rows, rank)
                    self.average = None                     # get_image() falls back to the gathered image
                    self._strip_rows = None

        if world > 1 and isinstance(fabric, sharding.NativeFabric):
            torch.cuda.synchronize(device)
            fabric.close()                                  # every rank is here: a collective destroy of the communicators
        return self.get_image()
