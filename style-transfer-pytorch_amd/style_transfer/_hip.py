"""ctypes binding of libst_amd.so (C ABI declared in include/st_amd.h).

PyTorch is used for device memory and streams only: every call below hands raw ``data_ptr()``s and
the current HIP stream to the library.  There is NO fallback: if the library is missing, was not
built for gfx950, or no GPU is visible, loading fails with an explicit error.
"""

import ctypes
import os

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# in-tree (repo checkout, `pip install -e`): style-transfer-pytorch_amd/lib/; installed wheel / non-editable install:
# setup.py's build step copies the library INTO the package (style_transfer/lib/)
_LIB_CANDIDATES = (os.path.join(_PKG_ROOT, 'lib', 'libst_amd.so'),
                   os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libst_amd.so'))
LIB_PATH = next((c for c in _LIB_CANDIDATES if os.path.exists(c)), _LIB_CANDIDATES[0])

_c_float_p = ctypes.c_void_p      # device pointers travel as integers
_lib = None


class HipLibraryError(RuntimeError):
    pass


class Exchange(ctypes.Structure):
    """st_exchange (include/st_amd.h): what must be exchanged between two closure phases."""
    _fields_ = [('kind', ctypes.c_int), ('count', ctypes.c_longlong), ('send_up', ctypes.c_void_p),
                ('send_down', ctypes.c_void_p), ('recv_up', ctypes.c_void_p), ('recv_down', ctypes.c_void_p),
                ('buffer', ctypes.c_void_p), ('root', ctypes.c_int), ('channel', ctypes.c_int),
                ('stream', ctypes.c_void_p)]


def _declare(lib):
    vp, i32, i64, f64, f32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_double, ctypes.c_float
    pp = ctypes.POINTER(ctypes.c_void_p)
    ip = ctypes.POINTER(ctypes.c_int)
    sig = {
        'st_last_error': (ctypes.c_char_p, []),
        'st_abi_version': (i32, []),
        'st_compiled_arch': (ctypes.c_char_p, []),
        'st_has_experiments': (i32, []),
        'st_env_switches': (i32, [ctypes.POINTER(ctypes.c_char_p), i32]),
        'st_set_option': (i32, [ctypes.c_char_p, i32, i32]),
        'st_net_create': (i32, [pp, pp, pp, i32]),
        'st_net_create_ex': (i32, [pp, pp, pp, i32, i32]),
        'st_net_destroy': (i32, [vp]),
        'st_net_wide_layers': (i32, [vp, ip, ip]),
        'st_net_mark_wide': (i32, [vp, ip, ip]),
        'st_plan_create': (i32, [pp, vp, i32, i32]),
        'st_plan_range_guard': (i32, [vp, vp, ip, ip, vp]),
        'st_plan_destroy': (i32, [vp]),
        'st_plan_device_bytes': (i64, [vp]),
        'st_plan_forward': (i32, [vp, vp, i32, vp]),
        'st_plan_feature': (i32, [vp, i32, pp, ip, ip, ip]),
        'st_plan_moments': (i32, [vp, i32, vp, vp, vp]),
        'st_plan_set_content_target': (i32, [vp, vp, vp]),
        'st_plan_set_style_target': (i32, [vp, i32, vp, vp, vp]),
        'st_plan_set_loss_weights': (i32, [vp, f32, ctypes.POINTER(f32), f32]),
        'st_plan_loss_and_grad': (i32, [vp, vp, vp, vp, vp]),
        'st_plan_step': (i32, [vp, vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, vp, vp]),
        'st_plan_apply_update': (i32, [vp, vp, vp, vp, vp, vp, i64, f64, f64, f64, f64, f64, vp]),
        'st_plan_create_strip': (i32, [pp, vp, i32, i32, i32, i32]),
        'st_plan_closure_begin': (i32, [vp, vp, vp]),
        'st_plan_set_rank': (i32, [vp, i32, i32]),
        'st_plan_closure_next': (i32, [vp, ctypes.POINTER(Exchange), vp]),
        'st_fabric_unique_id': (i32, [ctypes.c_char_p]),
        'st_fabric_create': (i32, [pp, ctypes.c_char_p, ctypes.c_char_p, i32, i32, i32]),
        'st_fabric_destroy': (i32, [vp]),
        'st_fabric_abort': (i32, [vp]),
        'st_fabric_selftest': (i32, [vp, vp, i32]),
        'st_plan_closure_run': (i32, [vp, vp, vp]),
        'st_plan_losses': (i32, [vp, pp]),
        'st_plan_debug_read': (i32, [vp, i32, ctypes.POINTER(f32), i32]),
        'st_plan_forward_begin': (i32, [vp, vp, i32]),
        'st_plan_moment_sums': (i32, [vp, i32, vp, vp]),
        'st_plan_set_graph': (i32, [vp, i32]),
        'st_plan_profile_enable': (i32, [vp, i32]),
        'st_plan_profile_read': (i32, [vp, ctypes.POINTER(i64), ctypes.POINTER(f64), ctypes.POINTER(f64)]),
        'st_plan_profile_read_hbm': (i32, [vp, i32, ctypes.POINTER(i64), ctypes.POINTER(f64), ctypes.POINTER(f64)]),
        'st_op_sqrtm_ns': (i32, [vp, vp, i32, vp]),
        'st_op_sqrtm_ns_backward': (i32, [vp, vp, vp, i32, vp]),
        'st_op_sqrtm_ns_backward_diag': (i32, [vp, f32, vp, i32, vp]),
        'st_op_tv_loss': (i32, [vp, i32, i32, vp, vp, vp]),
        'st_op_sqrtm_time': (i32, [i32, i32, ctypes.POINTER(f64), ctypes.POINTER(f64), vp]),
        'st_op_mfma_rate': (i32, [i32, i32, i32, i32, ctypes.POINTER(f64), ctypes.POINTER(f64), vp]),
        'st_op_mfma_valu_rate': (i32, [i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(f64), ctypes.POINTER(f64),
                                       ctypes.POINTER(f64), vp]),
        'st_op_grid_barrier_time': (i32, [i32, i32, i32, i32, ctypes.POINTER(f64), ip, vp]),
        'st_op_winograd_consumer_rate': (i32, [i32, i32, ctypes.POINTER(f64), vp]),
        'st_op_conv3x3_time': (i32, [i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(f64), vp]),
        'st_op_conv3x3': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
        'st_op_conv3x3_dgrad': (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        'st_op_conv3x3_strip': (i32, [vp, vp, i32, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
        'st_op_conv3x3_strip_ex': (i32, [vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
        'st_op_conv1x1': (i32, [vp, vp, vp, vp, i32, i32, i64, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)       # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return sig


EXPORTED_SYMBOLS = None


def load_library(require_gpu=True):
    """Load libst_amd.so.  ``require_gpu=False`` is only for the CPU-side symbol/ABI check."""
    global _lib, EXPORTED_SYMBOLS
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f'{LIB_PATH} not found: build it with `python style-transfer-pytorch_amd/build.py` '
                '(needs hipcc, targets gfx950).  This package has no CPU or PyTorch fallback.')
        lib = ctypes.CDLL(LIB_PATH)
        EXPORTED_SYMBOLS = sorted(_declare(lib))
        if lib.st_abi_version() != 2:
            raise HipLibraryError('libst_amd.so ABI version mismatch')
        _lib = lib
    if require_gpu and not torch.cuda.is_available():
        raise HipLibraryError('no HIP device visible: the MI355X hot path cannot run (no CPU fallback)')
    return _lib


def has_experiments():
    """True when libst_amd.so was built with ``build.py --experiments`` (Winograd conv, persistent NS chain kernel, ...)."""
    return bool(load_library(require_gpu=False).st_has_experiments())


def env_switches():
    """The ST_* switches a default build reads from the environment (everything else: ``set_option`` / ``options``)."""
    lib = load_library(require_gpu=False)
    names = (ctypes.c_char_p * 64)()
    n = lib.st_env_switches(names, 64)
    return [names[i].decode() for i in range(n)]


_overrides = {}      # this process's current st_set_option overrides (the library has no getter)


def set_option(name, value=None):
    """Override (or, with value=None, clear) one of the library's ST_* switches for this process."""
    lib = load_library(require_gpu=False)
    _check(lib.st_set_option(name.encode(), int(value or 0), 1 if value is None else 0))
    if value is None:
        _overrides.pop(name, None)
    else:
        _overrides[name] = int(value)


class options:
    """Context manager: ``with options(ST_CONV_PC=0): ...`` runs the block with the switches overridden and puts the
    PREVIOUS overrides back afterwards (nested / outer overrides of the same switch survive)."""

    def __init__(self, **kv):
        self.kv = kv
        self.saved = {}

    def __enter__(self):
        self.saved = {k: _overrides.get(k) for k in self.kv}
        for k, v in self.kv.items():
            set_option(k, v)

    def __exit__(self, *exc):
        for k, prev in self.saved.items():
            set_option(k, prev)


def _check(rc):
    if rc != 0:
        raise HipLibraryError(load_library(False).st_last_error().decode('utf-8', 'replace'))


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), 'expected a contiguous fp32 HIP tensor'
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Net:
    """Frozen VGG-19 trunk on one device (st_net)."""

    PRECISIONS = {'fp32': 0, 'bf16x3': 2, 'bf16x6': 3, 'fp16x3': 4}

    def __init__(self, params, pooling, device, precision='fp32'):
        self.lib = load_library()
        self.device = torch.device(device)
        self.pooling = pooling
        self.precision = precision
        with torch.cuda.device(self.device):
            dev = [(w.to(self.device, torch.float32).contiguous(), b.to(self.device, torch.float32).contiguous())
                   for w, b in params]
            wa = (ctypes.c_void_p * 13)(*[w.data_ptr() for w, _ in dev])
            ba = (ctypes.c_void_p * 13)(*[b.data_ptr() for _, b in dev])
            h = ctypes.c_void_p()
            torch.cuda.synchronize(self.device)
            _check(self.lib.st_net_create_ex(ctypes.byref(h), wa, ba, {'max': 0, 'average': 1, 'l2': 2}[pooling],
                                             self.PRECISIONS[precision]))
        self.handle = h

    def wide_layers(self):
        """([13 ints], [13 ints]): convolutions whose forward / data gradient the fp16x3 dynamic-range guard moved to
        bf16x6 (st_net_wide_layers)."""
        fwd, bwd = (ctypes.c_int * 13)(), (ctypes.c_int * 13)()
        _check(self.lib.st_net_wide_layers(self.handle, fwd, bwd))
        return list(fwd), list(bwd)

    def mark_wide(self, forward13, backward13):
        """Add layers to the bf16x6 set (st_net_mark_wide): the union of the ranks' range-guard verdicts in a sharded run."""
        fwd, bwd = (ctypes.c_int * 13)(*[int(v) for v in forward13]), (ctypes.c_int * 13)(*[int(v) for v in backward13])
        with torch.cuda.device(self.device):
            _check(self.lib.st_net_mark_wide(self.handle, fwd, bwd))

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h and self.lib is not None:
            self.lib.st_net_destroy(h)


class Plan:
    """All device buffers and kernels for one image size (st_plan)."""

    def __init__(self, net, height, width):
        self.lib = net.lib
        self.net = net
        self.device = net.device
        self.height, self.width = int(height), int(width)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.st_plan_create(ctypes.byref(h), net.handle, self.height, self.width)
        if rc != 0:
            msg = self.lib.st_last_error().decode()
            if 'must be at least' in msg:
                raise ValueError(msg)          # same exception type as VGGFeatures.forward (:83)
            raise HipLibraryError(msg)
        self.handle = h
        self.losses = torch.zeros(8, device=self.device, dtype=torch.float32)

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h and self.lib is not None:
            self.lib.st_plan_destroy(h)

    def device_bytes(self):
        return int(self.lib.st_plan_device_bytes(self.handle))

    def debug_read(self, what, count):
        """Diagnostic: `count` floats of an internal buffer (0: the TV kernels' per-workgroup partial sums)."""
        import numpy as np
        buf = np.empty(int(count), dtype=np.float32)
        _check(self.lib.st_plan_debug_read(self.handle, int(what), buf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), int(count)))
        return torch.from_numpy(buf)

    def _img(self, image):
        assert image.shape[-3:] == (3, self.height, self.width), (image.shape, self.height, self.width)
        return _ptr(image)

    def forward(self, image, last_layer=29):
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_forward(self.handle, self._img(image), int(last_layer), _stream()))

    def feature(self, layer):
        """Copy of a tap as a [1, C, h, w] tensor."""
        data, c, h, w = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(self.lib.st_plan_feature(self.handle, int(layer), ctypes.byref(data), ctypes.byref(c),
                                        ctypes.byref(h), ctypes.byref(w)))
        out = torch.empty((1, c.value, h.value, w.value), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _copy_d2d(out, data.value)   # D2D on the current stream from the ABI's borrowed pointer
        return out

    def moments(self, layer):
        c = {1: 64, 6: 128, 11: 256, 20: 512, 29: 512}[int(layer)]
        mean = torch.empty(c, device=self.device, dtype=torch.float32)
        srm = torch.empty((c, c), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_moments(self.handle, int(layer), _ptr(mean), _ptr(srm), _stream()))
        return mean, srm

    def set_content_target(self, feat):
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_set_content_target(self.handle, _ptr(feat.contiguous()), _stream()))

    def set_content_target_from_forward(self):
        data = ctypes.c_void_p()
        _check(self.lib.st_plan_feature(self.handle, 22, ctypes.byref(data), None, None, None))
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_set_content_target(self.handle, data, _stream()))

    def set_style_target(self, index, mean, srm):
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_set_style_target(self.handle, int(index), _ptr(mean.contiguous()),
                                                     _ptr(srm.contiguous()), _stream()))

    def set_loss_weights(self, content_weight, style_layer_weights, tv_weight):
        arr = (ctypes.c_float * 5)(*[float(w) for w in style_layer_weights])
        _check(self.lib.st_plan_set_loss_weights(self.handle, float(content_weight), arr, float(tv_weight)))

    def loss_and_grad(self, image, grad_out=None):
        """Returns (losses[8] device tensor: 7 weighted terms + total, grad [like image])."""
        if grad_out is None:
            grad_out = torch.empty_like(image)
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_loss_and_grad(self.handle, self._img(image), _ptr(grad_out),
                                                  _ptr(self.losses), _stream()))
        return self.losses, grad_out

    def step(self, image, exp_avg, exp_avg_sq, ema_value, step, lr, beta1=0.9, beta2=0.99, eps=1e-8,
             ema_decay=0.99):
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_step(self.handle, self._img(image), _ptr(exp_avg), _ptr(exp_avg_sq),
                                         _ptr(ema_value), int(step), float(lr), float(beta1), float(beta2),
                                         float(eps), float(ema_decay), _ptr(self.losses), _stream()))
        return self.losses

    def range_guard(self, image):
        """Activation-aware dynamic-range check of the fp16x3 convolutions on ``image`` (st_plan_range_guard): returns the
        ([13], [13]) forward / data-gradient layers this call moved to bf16x6 (cold path, synchronous)."""
        fwd, bwd = (ctypes.c_int * 13)(), (ctypes.c_int * 13)()
        with torch.cuda.device(self.device):
            _check(self.lib.st_plan_range_guard(self.handle, self._img(image), fwd, bwd, _stream()))
        return list(fwd), list(bwd)

    def set_graph(self, on=True):
        _check(self.lib.st_plan_set_graph(self.handle, 1 if on else 0))

    def profile_enable(self, on=True):
        _check(self.lib.st_plan_profile_enable(self.handle, 1 if on else 0))

    HBM_KERNELS = ('conv1_1 forward (+ normalize)', 'conv1_1 data gradient (+ pad fold)', 'max-pool backward (4 levels)',
                   'Adam + clamp + EMA', 'TV loss + gradient', 'relu1_1 Gram + mean', 'content MSE + gradient',
                   "style heads' 1x1 gradient step (5 taps)")

    def profile_read_hbm(self):
        """{kernel: (launches, ms, algorithmic bytes)} of the HBM-bound kernels since the last profile_read()
        (call BEFORE profile_read, which recycles the events)."""
        out = {}
        for cat, name in enumerate(self.HBM_KERNELS):
            n, ms, by = ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
            _check(self.lib.st_plan_profile_read_hbm(self.handle, cat, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)))
            out[name] = (n.value, ms.value, by.value)
        return out

    def profile_read(self):
        n, ms, fl = ctypes.c_longlong(), ctypes.c_double(), ctypes.c_double()
        _check(self.lib.st_plan_profile_read(self.handle, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl)))
        return n.value, ms.value, fl.value


def _copy_d2d(dst, src_ptr):
    """Device-to-device copy from a borrowed raw pointer into a torch tensor (same device)."""
    hip = _hip_runtime()
    rc = hip.hipMemcpyAsync(ctypes.c_void_p(dst.data_ptr()), ctypes.c_void_p(src_ptr),
                            ctypes.c_size_t(dst.numel() * dst.element_size()), 3,
                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise HipLibraryError(f'hipMemcpyAsync failed with code {rc}')


_hiprt = None


def _hip_runtime():
    global _hiprt
    if _hiprt is None:
        for name in ('libamdhip64.so', '/opt/rocm/lib/libamdhip64.so', 'libamdhip64.so.7', 'libamdhip64.so.6'):
            try:
                _hiprt = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _hiprt is None:
            raise HipLibraryError('libamdhip64.so not found')
        _hiprt.hipMemcpyAsync.restype = ctypes.c_int
        _hiprt.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                          ctypes.c_void_p]
    return _hiprt


# ---- standalone operators (kernel-level parity tests) ------------------------------------------
def op_sqrtm_ns(a):
    lib = load_library()
    n = a.shape[-1]
    root = torch.empty(a.shape, dtype=a.dtype, device=a.device)       # (empty_like would keep a transposed view's strides)
    with torch.cuda.device(a.device):
        _check(lib.st_op_sqrtm_ns(_ptr(a.contiguous()), _ptr(root), n, _stream()))
    return root


def op_sqrtm_ns_backward(root, grad_root):
    lib = load_library()
    n = root.shape[-1]
    ga = torch.empty(root.shape, dtype=root.dtype, device=root.device)
    with torch.cuda.device(root.device):
        _check(lib.st_op_sqrtm_ns_backward(_ptr(root.contiguous()), _ptr(grad_root.contiguous()), _ptr(ga), n,
                                           _stream()))
    return ga


def op_sqrtm_ns_backward_diag(root, grad_diag):
    """Lyapunov backward for grad_root = grad_diag * I (the plan's code path)."""
    lib = load_library()
    n = root.shape[-1]
    ga = torch.empty(root.shape, dtype=root.dtype, device=root.device)
    with torch.cuda.device(root.device):
        _check(lib.st_op_sqrtm_ns_backward_diag(_ptr(root.contiguous()), float(grad_diag), _ptr(ga), n, _stream()))
    return ga


def op_sqrtm_time(n, iters=20):
    """(forward, backward) microseconds per NS-12 chain, HIP events (tools/ns_bench.py)."""
    lib = load_library()
    f, b = ctypes.c_double(), ctypes.c_double()
    _check(lib.st_op_sqrtm_time(int(n), int(iters), ctypes.byref(f), ctypes.byref(b), _stream()))
    return f.value, b.value


def op_mfma_rate(lds_reads=0, waves=8, steps=20000, launches=10):
    """(TFLOP/s, shader MHz) the fp16 matrix pipe sustains under the XL convolution tile's consumer pattern with
    `lds_reads` ds_read_b128 per 12 MFMAs (csrc/st_diag.hip; tools/mfma_rate.py)."""
    lib = load_library()
    t, m = ctypes.c_double(), ctypes.c_double()
    _check(lib.st_op_mfma_rate(int(lds_reads), int(waves), int(steps), int(launches), ctypes.byref(t),
                               ctypes.byref(m), _stream()))
    return t.value, m.value


def op_mfma_valu_rate(lds_reads, waves, steps, valu_waves, valu_steps, valu_prio=0, launches=10):
    """(TFLOP/s, MHz, cycles per VALU instruction of the VALU-only waves, cycles per MFMA of an MFMA wave)."""
    lib = load_library()
    t, m = ctypes.c_double(), ctypes.c_double()
    c = (ctypes.c_double * 2)()
    _check(lib.st_op_mfma_valu_rate(int(lds_reads), int(waves), int(steps), int(launches), int(valu_waves),
                                    int(valu_steps), int(valu_prio), ctypes.byref(t), ctypes.byref(m), c, _stream()))
    return t.value, m.value, c[0], c[1]


def op_winograd_consumer_rate(steps=4096, launches=5):
    """TFLOP/s of MFMA work sustained in the Winograd tile's consumer pattern (csrc/st_diag.hip wino_rate_kernel)."""
    lib = load_library()
    t = ctypes.c_double()
    _check(lib.st_op_winograd_consumer_rate(int(steps), int(launches), ctypes.byref(t), _stream()))
    return t.value


def op_grid_barrier_time(workgroups=256, rounds=200, payload_floats=1024, groups=0):
    """(microseconds per round, stale reads) of device-wide barriers inside one launch (csrc/st_diag.hip)."""
    lib = load_library()
    us, err = ctypes.c_double(), ctypes.c_int()
    _check(lib.st_op_grid_barrier_time(int(workgroups), int(rounds), int(payload_floats), int(groups), ctypes.byref(us),
                                       ctypes.byref(err), _stream()))
    return us.value, err.value


def op_tv_loss(image):
    lib = load_library()
    h, w = image.shape[-2:]
    loss = torch.zeros(1, device=image.device, dtype=torch.float32)
    grad = torch.empty_like(image)
    with torch.cuda.device(image.device):
        _check(lib.st_op_tv_loss(_ptr(image.contiguous()), h, w, _ptr(loss), _ptr(grad), _stream()))
    return loss, grad


def op_conv3x3_time(cin, cout, height, width, dgrad=False, precision=4, iters=20):
    """Average microseconds per launch of the 3x3 convolution on device-resident random operands (st_op_conv3x3_time)."""
    lib = load_library()
    us = ctypes.c_double()
    _check(lib.st_op_conv3x3_time(int(cin), int(cout), int(height), int(width), 1 if dgrad else 0, int(precision), int(iters),
                                  ctypes.byref(us), _stream()))
    return us.value


def op_conv3x3(x, weight, bias, relu, precision=0):
    lib = load_library()
    cout, cin = weight.shape[:2]
    h, w = x.shape[-2:]
    out = torch.empty((1, cout, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _check(lib.st_op_conv3x3(_ptr(x.contiguous()), _ptr(weight.contiguous()),
                                 _ptr(bias.contiguous()) if bias is not None else None, _ptr(out), cin, cout,
                                 h, w, 1 if relu else 0, int(precision), _stream()))
    return out


def op_conv3x3_strip(x, halo, has_up, has_down, weight, bias, relu, dgrad, precision=4):
    """The 3x3 convolution (or its data gradient) on a row strip: ``halo`` = [2, C, W] neighbour rows."""
    lib = load_library()
    cout, cin = weight.shape[:2]
    h, w = x.shape[-2:]
    out = torch.empty((1, cin if dgrad else cout, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _check(lib.st_op_conv3x3_strip(_ptr(x.contiguous()), _ptr(halo.contiguous()), int(bool(has_up)),
                                       int(bool(has_down)), _ptr(weight.contiguous()),
                                       _ptr(bias.contiguous()) if bias is not None else None, _ptr(out), cin, cout,
                                       h, w, 1 if relu else 0, 1 if dgrad else 0, int(precision), _stream()))
    return out


def op_conv3x3_strip_ex(x, halo, has_up, has_down, weight, bias, relu, dgrad, out=None, out_mask=None, overlap=False,
                        precision=4):
    """op_conv3x3_strip with the plan's epilogue options: ``out`` given = accumulate into it (in place, returned),
    ``out_mask``, ``overlap`` = the interior + boundary two-launch form."""
    lib = load_library()
    cout, cin = weight.shape[:2]
    h, w = x.shape[-2:]
    acc = out is not None
    if out is None:
        out = torch.empty((1, cin if dgrad else cout, h, w), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _check(lib.st_op_conv3x3_strip_ex(_ptr(x.contiguous()), _ptr(halo.contiguous()) if halo is not None else None, int(bool(has_up)),
                                          int(bool(has_down)), _ptr(weight.contiguous()),
                                          _ptr(bias.contiguous()) if bias is not None else None, _ptr(out),
                                          _ptr(out_mask.contiguous()) if out_mask is not None else None, cin, cout, h, w,
                                          1 if relu else 0, 1 if dgrad else 0, 1 if acc else 0, 1 if overlap else 0,
                                          int(precision), _stream()))
    return out


def op_conv1x1(x, weight, bias, precision=0):
    """x [Cin, npix], weight [Cout, Cin], bias [Cout] or None -> [Cout, npix] (the style heads' gradient step)."""
    lib = load_library()
    cout, cin = weight.shape
    npix = x.shape[1]
    out = torch.empty((cout, npix), device=x.device, dtype=torch.float32)
    with torch.cuda.device(x.device):
        _check(lib.st_op_conv1x1(_ptr(x.contiguous()), _ptr(weight.contiguous()),
                                 _ptr(bias.contiguous()) if bias is not None else None, _ptr(out), cin, cout,
                                 npix, int(precision), _stream()))
    return out


def op_conv3x3_dgrad(grad_out, relu_out, weight, precision=0):
    lib = load_library()
    cout, cin = weight.shape[:2]
    h, w = grad_out.shape[-2:]
    gin = torch.empty((1, cin, h, w), device=grad_out.device, dtype=torch.float32)
    with torch.cuda.device(grad_out.device):
        _check(lib.st_op_conv3x3_dgrad(_ptr(grad_out.contiguous()),
                                       _ptr(relu_out.contiguous()) if relu_out is not None else None,
                                       _ptr(weight.contiguous()), _ptr(gin), cin, cout, h, w, int(precision),
                                       _stream()))
    return gin
