"""Spatial strip sharding of the hot path across the GPUs of one node.

Replaces the reference's 2-device *layer* split (``style_transfer.py:326-333``: layers 0-4 on one
device, 5-29 on the other, no speed-up) with a split of the *image*: rank r of R owns a horizontal
strip of rows (boundaries at multiples of 16 so all four 2x2 poolings stay strip-local) of the image,
of the Adam/EMA state and of every feature map.  Per iteration the ranks exchange

  * one boundary row of every convolution operand with the strip above / below (forward: 13 halo
    exchanges, backward: 13, point-to-point over xGMI) - issued on a communication stream while the
    interior rows of the consuming convolution are computed (the library cuts that convolution into an
    interior and a boundary launch),
  * the raw Gram / mean sums of the five style taps (16 KB ... 1 MB each): reduced to the head's OWNER
    rank, which alone runs that head's Newton-Schulz chains and broadcasts (Ssym, b, loss term) back
    - the two n = 512 chains land on different GPUs instead of slowing each other on every GPU,
  * five scalars (content and TV partial sums).

The HIP library runs the closure as a sequence of compute phases and tells the caller what to
exchange between them and on which HIP stream (``st_exchange``); this module performs those exchanges
with ``torch.distributed`` (backend ``nccl`` = RCCL; stream-ordered, the host never waits) on zero-copy
tensor views of the library's buffers, or - for parity tests on a single GPU - between several strip
plans living in one process.
"""

import ctypes

import os

import torch

from . import _hip


# Fixed per-iteration work of a rank beyond its rows' convolutions, in milliseconds, by the rank's position in the owner
# table (set_rank: head h's Newton-Schulz chains run on rank (4 - h) % world).  relu5_1's two n = 512 chains sit at the head
# of the backward pass with nothing to hide behind; relu4_1's mostly hide behind the conv5 -> conv4 backward trunk; the
# n <= 256 heads cost a few tens of microseconds.  Measured with tools/strip_bench.py (profiles/r05_strip_bench.txt:
# rank 0 exceeds the median rank by 0.50-0.55 ms at 2048^2 and 2896x2172 on 8 ranks, rank 1 by 0.09-0.20 ms).
_HEAD_OWNER_MS = (0.01, 0.02, 0.04, 0.12, 0.50)
# One 16-row block of the whole closure (forward, backward, update) per image column, in milliseconds: 4.2 ms of
# convolution launches + 0.1 ms of pointwise work for a 17-block strip of 2896 columns (profiles/r05_strip_bench.txt).
_BLOCK_MS_PER_COLUMN = 0.25 / 2896


def strip_rows(height, world, width=None):
    """Row ranges [(begin, end)] of the ``world`` strips: whole 16-row blocks; the last strip also takes the H % 16
    remainder.

    Blocks are spread as evenly as possible (earlier ranks get the extra block).

    ``width`` given and ST_STRIP_BALANCE=1 in the environment (round 5's default, round 6: opt-in): blocks are dealt so that
    the largest (rows + chain-owner work) is as small as possible - the owner of relu5_1's Newton-Schulz chains (rank 0)
    gets fewer rows by what those chains cost, in blocks of this width (at most an eighth of an even strip; 4 ranks or
    more; not when the even strips are whole 128-row groups).  That minimises the slowest rank of a model in which only
    the OWNER waits for its chains (tools/strip_bench.py with stubbed broadcasts).  On hardware every rank waits for
    relu5_1's owner - forward pass, the owner's chains, backward pass, on every rank - so the iteration is
    max(forward) + chains + max(backward) and EQUAL strips are the better deal: with the owner's wait replayed on the other
    ranks 2896 x 2172 / 8 takes 5.68 ms even against 5.75 balanced, / 4 9.07 against 9.37
    (profiles/r06_strip_breakdown.md).  Every rank computes the same table from (height, world, width)."""
    blocks = height // 16
    if blocks < world:
        raise ValueError(f'an image of {height} rows cannot be cut into {world} strips of >= 16 rows')
    base, extra = divmod(blocks, world)
    counts = [base + (1 if r < extra else 0) for r in range(world)]
    # Not on 2 ranks (both own an n = 512 chain; rank 0's measured excess is 0.13 ms of 16), and not when the even strips
    # are whole groups of 128 rows (power-of-two images): those are whole tile rows of the conv5 kernels, a block more or
    # less adds a partly filled tile row to every deep layer (measured -0.5 ... -1.4 % at 2048^2 / 4, / 8 and 4096^2 / 8
    # against +5.6 % at 2896x2172 / 8 and +1.7 % at / 4: profiles/r05_strip_bench.txt).
    quantum = extra == 0 and base % 8 == 0
    if width is not None and world >= 4 and not quantum and os.environ.get('ST_STRIP_BALANCE', '0') == '1':
        block_ms = _BLOCK_MS_PER_COLUMN * width
        load = [0.0] * world
        for head, ms in enumerate(_HEAD_OWNER_MS):
            load[(4 - head) % world] += ms / block_ms
        load = [min(l, base / 8) for l in load]             # the model is for strips of many blocks: at most an eighth of a strip moves
        load[-1] += (height % 16) / 16
        dealt = [1] * world                                  # every strip keeps >= 16 rows
        for _ in range(blocks - world):
            r = min(range(world), key=lambda i: (load[i] + dealt[i], i))
            dealt[r] += 1
        # the deal is only taken when it shortens the longest rank by at least a quarter block: equal strips otherwise
        worst = lambda c: max(l + n for l, n in zip(load, c))
        if worst(dealt) <= worst(counts) - 0.25:
            counts = dealt
    rows, begin = [], 0
    for r in range(world):
        end = begin + 16 * counts[r]
        if r == world - 1:
            end = height
        rows.append((begin, end))
        begin = end
    return rows


class _Raw:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': '<f4', 'data': (int(ptr), False),
                                         'version': 2, 'strides': None}


def view(ptr, count, device):
    """Zero-copy fp32 tensor over ``count`` floats of library-owned device memory."""
    if not ptr:
        return None
    return torch.as_tensor(_Raw(ptr, count), device=device)


class StripPlan(_hip.Plan):
    """st_plan for rows [row_begin, row_end) of a global_height x width image."""

    def __init__(self, net, global_height, width, row_begin, row_end):
        self.lib = net.lib
        self.net = net
        self.device = net.device
        self.global_height, self.width = int(global_height), int(width)
        self.row_begin, self.row_end = int(row_begin), int(row_end)
        self.height = self.row_end - self.row_begin
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.st_plan_create_strip(ctypes.byref(h), net.handle, self.global_height, self.width,
                                               self.row_begin, self.row_end)
        if rc != 0:
            msg = self.lib.st_last_error().decode()
            raise ValueError(msg) if ('must be' in msg or 'strip' in msg) else _hip.HipLibraryError(msg)
        self.handle = h
        p = ctypes.c_void_p()
        _hip._check(self.lib.st_plan_losses(self.handle, ctypes.byref(p)))
        self.losses = view(p.value, 8, self.device)

    def set_rank(self, rank, world):
        """Position among the ranks: with world > 1 each style head's Newton-Schulz chains run on one owner rank
        ((4 - head) % world) and the result is broadcast (st_plan_set_rank)."""
        _hip._check(self.lib.st_plan_set_rank(self.handle, int(rank), int(world)))
        return self

    # ---- phase machine ----
    # The phase machine keeps the raw pointers until the sequence ends, so the tensors are pinned here
    # (a temporary passed by the caller would otherwise be recycled by torch's caching allocator).
    def forward_begin(self, image, last_layer):
        self._inflight = (image,)
        _hip._check(self.lib.st_plan_forward_begin(self.handle, self._img(image), int(last_layer)))

    def closure_begin(self, image, grad):
        self._inflight = (image, grad)
        _hip._check(self.lib.st_plan_closure_begin(self.handle, self._img(image), _hip._ptr(grad)))

    def next(self, stream=None, ex=None):
        """Run the next compute phase and return the exchange that must follow it.  ``stream`` / ``ex``: a cached
        stream handle and a reusable descriptor (run_phases passes both: ~30 phases per closure)."""
        ex = _hip.Exchange() if ex is None else ex
        if stream is None:
            with torch.cuda.device(self.device):
                _hip._check(self.lib.st_plan_closure_next(self.handle, ctypes.byref(ex), _hip._stream()))
        else:
            _hip._check(self.lib.st_plan_closure_next(self.handle, ctypes.byref(ex), stream))
        return ex

    def moment_sums(self, layer):
        c = {1: 64, 6: 128, 11: 256, 20: 512, 29: 512}[int(layer)]
        sums = torch.empty(c * c + c, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _hip._check(self.lib.st_plan_moment_sums(self.handle, int(layer), _hip._ptr(sums), _hip._stream()))
        return sums

    def apply_update(self, image, grad, exp_avg, exp_avg_sq, ema_value, step, lr, beta1=0.9, beta2=0.99,
                     eps=1e-8, ema_decay=0.99):
        with torch.cuda.device(self.device):
            _hip._check(self.lib.st_plan_apply_update(
                self.handle, self._img(image), _hip._ptr(grad), _hip._ptr(exp_avg), _hip._ptr(exp_avg_sq),
                _hip._ptr(ema_value), int(step), float(lr), float(beta1), float(beta2), float(eps),
                float(ema_decay), _hip._stream()))


# ---- transports ----------------------------------------------------------------------------------
_HEAD_GROUPS = {}      # parent group (None = default) -> the second communicator for the style heads' traffic


def _head_group_for(dist, group, world):
    """The heads' own process group (= their own RCCL communicator), created ONCE per parent group and process:
    stylize() builds a fabric per call and bench.py one per attempt - a new_group() each time would leak a
    communicator per call.  Creating it is a collective over the parent group (every rank builds its first fabric)."""
    # keyed by the parent's RANKS and guarded by a weak reference to the parent object: id(group) alone can be reused by a
    # new ProcessGroup after the old one was collected, which would hand out a communicator over other ranks (or a destroyed
    # one) - and a cache hit on some ranks with a miss on others would deadlock in new_group()
    import weakref
    ranks = tuple(dist.get_process_group_ranks(group)) if group is not None else tuple(range(world))
    key = (None,) + ranks if group is None else (id(group),) + ranks
    cached = _HEAD_GROUPS.get(key)
    if cached is not None and (group is None or cached[1]() is group):
        return cached[0]
    made = dist.new_group(ranks=list(ranks))
    ref = None
    if group is not None:
        try:
            ref = weakref.ref(group, lambda _r, k=key: _HEAD_GROUPS.pop(k, None))
        except TypeError:                                    # (not weak-referenceable: compare identities via a strong reference)
            ref = (lambda g=group: g)
    _HEAD_GROUPS[key] = (made, ref)
    return made


def release_head_groups():
    """Destroy the cached head communicators (before dist.destroy_process_group(), or between process groups)."""
    import torch.distributed as dist
    for key, (grp, _ref) in list(_HEAD_GROUPS.items()):
        try:
            if dist.is_initialized():
                dist.destroy_process_group(grp)
        except Exception:                                    # noqa: BLE001 - the parent group may already be gone
            pass
        _HEAD_GROUPS.pop(key, None)


class DistFabric:
    """Exchanges over torch.distributed (nccl = RCCL over xGMI on MI355X; gloo in the CPU tests).

    RCCL work is ordered on the stream the descriptor names (``torch.cuda.ExternalStream`` over the library's
    handle): the host only enqueues.  Channel 1 (style heads) gets its own process group = its own communicator, so
    that a head's broadcast - which waits for its owner's chain - never sits in front of the trunk's halo exchanges
    (operations of one communicator execute in issue order)."""

    def __init__(self, rank, world, group=None, host_sync=None):
        import torch.distributed as dist
        self.dist, self.rank, self.world, self.group = dist, int(rank), int(world), group
        self._cache = {}
        self._streams = {}
        # RCCL work is ordered against the current stream; gloo (CPU tests, single-GPU multi-process tests) is not:
        # there every exchange on device tensors becomes a host-synchronous step.  host_sync=True forces that
        # conservative form on any backend (device-wide synchronise before and after every exchange; bench.py's
        # fallback when the stream-ordered path fails on a system).
        self.host_sync = dist.is_initialized() and dist.get_backend(group) != 'nccl'
        if host_sync is not None:
            self.host_sync = bool(host_sync)
        self.head_group = group
        # ST_FABRIC_FORCE_COLLECTIVES=1: a single rank still issues its all-reduces / reductions / broadcasts (the one-GPU
        # smoke test of the RCCL descriptor path: communicators, zero-copy views of the library's buffers, ordering on the
        # library's streams - everything but the point-to-point halos, which need a neighbour)
        self.force = os.environ.get('ST_FABRIC_FORCE_COLLECTIVES') == '1'
        # ST_FABRIC_SELF_HALO=1 (one rank only): this rank is its own upper AND lower neighbour - the point-to-point
        # halo exchanges of a middle strip run as RCCL send / recv to self inside one group (periodic boundary), on the
        # library's communication stream like any neighbour exchange.  What one GPU can execute of the P2P path;
        # compared against run_phases_lockstep(..., wrap=True) in tests/test_rccl_self_halo_gpu.py.
        self.self_halo = os.environ.get('ST_FABRIC_SELF_HALO') == '1' and world == 1
        self.sync_p2p = os.environ.get('ST_FABRIC_SYNC_P2P') == '1'
        if dist.is_initialized() and (world > 1 or self.force) and dist.get_backend(group) == 'nccl':
            self.head_group = _head_group_for(dist, group, world)  # (first use is a collective over `group`)

    def _sync(self, tensor):
        if self.host_sync and tensor is not None and tensor.is_cuda:
            torch.cuda.synchronize(tensor.device)

    def _global(self, r):
        """Global rank of group rank r (P2P / rooted collectives take global ranks)."""
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _on(self, ex, device):
        """Context manager: the stream an exchange must be ordered on."""
        import contextlib
        if self.host_sync or not ex.stream:
            return contextlib.nullcontext()
        st = self._streams.get(ex.stream)
        if st is None:
            st = self._streams[ex.stream] = torch.cuda.ExternalStream(ex.stream, device=device)
        return torch.cuda.stream(st)

    def halo_exchange(self, send_up, send_down, recv_up, recv_down):
        """send_up -> rank-1 (lands in ITS recv_down); send_down -> rank+1 (ITS recv_up)."""
        dist, ops = self.dist, []
        if send_up is not None and self.rank > 0:
            ops.append(dist.P2POp(dist.isend, send_up, self._global(self.rank - 1), self.group))
            ops.append(dist.P2POp(dist.irecv, recv_up, self._global(self.rank - 1), self.group))
        if send_down is not None and self.rank < self.world - 1:
            ops.append(dist.P2POp(dist.isend, send_down, self._global(self.rank + 1), self.group))
            ops.append(dist.P2POp(dist.irecv, recv_down, self._global(self.rank + 1), self.group))
        if ops:
            self._sync(send_up if send_up is not None else send_down)
            for work in dist.batch_isend_irecv(ops):
                work.wait()
            self._sync(recv_up if recv_up is not None else recv_down)

    def allreduce(self, tensor, op=None):
        if self.world > 1 or self.force:
            self._sync(tensor)
            self.dist.all_reduce(tensor, op=op or self.dist.ReduceOp.SUM, group=self.group)
            self._sync(tensor)

    def allmax(self, tensor):
        self.allreduce(tensor, self.dist.ReduceOp.MAX)

    def apply(self, ex, device):
        """Perform one exchange descriptor of the phase machine.  The library's buffers are fixed for the life of
        a plan, so the zero-copy views and the P2P op lists are built once per distinct descriptor and reused
        (building four tensor views + ops costs more host time than the exchange itself at small strips)."""
        dist = self.dist
        group = self.head_group if ex.channel == 1 else self.group
        if ex.kind == 1:
            key = (1, ex.send_up, ex.send_down, ex.recv_up, ex.recv_down, int(ex.count), str(device))
            ops = self._cache.get(key)
            if ops is None:
                n = ex.count
                send_up, send_down = view(ex.send_up, n, device), view(ex.send_down, n, device)
                recv_up, recv_down = view(ex.recv_up, n, device), view(ex.recv_down, n, device)
                ops = []
                if self.self_halo:
                    # sends and receives between one pair of ranks match in issue order: my "up" rows land in the
                    # upper neighbour's recv_down (mine), my "down" rows in the lower neighbour's recv_up (mine)
                    me = self._global(self.rank)
                    assert None not in (send_up, send_down, recv_up, recv_down), 'self-halo needs a middle strip'
                    ops = [dist.P2POp(dist.isend, send_up, me, group), dist.P2POp(dist.irecv, recv_down, me, group),
                           dist.P2POp(dist.isend, send_down, me, group), dist.P2POp(dist.irecv, recv_up, me, group)]
                elif send_up is not None and self.rank > 0:
                    ops.append(dist.P2POp(dist.isend, send_up, self._global(self.rank - 1), group))
                    ops.append(dist.P2POp(dist.irecv, recv_up, self._global(self.rank - 1), group))
                if not self.self_halo and send_down is not None and self.rank < self.world - 1:
                    ops.append(dist.P2POp(dist.isend, send_down, self._global(self.rank + 1), group))
                    ops.append(dist.P2POp(dist.irecv, recv_down, self._global(self.rank + 1), group))
                self._cache[key] = ops
            if ops:
                self._sync(ops[0].tensor)
                with self._on(ex, device):
                    if self.sync_p2p:
                        # one coalesced group of SYNCHRONOUS sends / receives: c10d launches synchronous operations on
                        # the current stream (here: the library's communication stream) instead of its internal one
                        with dist.distributed_c10d._coalescing_manager(group, ops[0].tensor.device, async_ops=False):
                            for op in ops:
                                op.op(op.tensor, op.peer, group)
                    else:
                        for work in dist.batch_isend_irecv(ops):
                            work.wait()
                self._sync(ops[0].tensor)
        elif ex.kind in (2, 4, 5):
            if self.world == 1 and not self.force:
                return
            key = (2, ex.buffer, int(ex.count), str(device))
            t = self._cache.get(key)
            if t is None:
                t = self._cache[key] = view(ex.buffer, ex.count, device)
            self._sync(t)
            with self._on(ex, device):
                if ex.kind == 2:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                elif ex.kind == 4:
                    dist.reduce(t, dst=self._global(ex.root), op=dist.ReduceOp.SUM, group=group)
                else:
                    dist.broadcast(t, src=self._global(ex.root), group=group)
            self._sync(t)


class NativeFabric:
    """The in-library RCCL transport (csrc/st_fabric.hip): the phase machine's exchanges are issued by libst_amd.so itself -
    ncclSend / ncclRecv / collectives on the streams the descriptors name, two communicators of its own - and
    ``run_phases`` is ONE call per closure.  torch.distributed is only used to hand the two 128-byte communicator ids from
    rank 0 to the others (any backend) and stays the transport of the cold path (targets, scale transitions, L-BFGS
    scalars: ``cold`` is a DistFabric).  ST_FABRIC_SELF_HALO=1 with one rank: that rank is its own neighbour."""

    def __init__(self, rank, world, device, group=None, cold=None):
        import torch.distributed as dist
        self.lib = _hip.load_library()
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self.self_halo = os.environ.get('ST_FABRIC_SELF_HALO') == '1' and world == 1
        ids = torch.zeros(256, dtype=torch.uint8)
        if rank == 0:
            for c in range(2):
                buf = ctypes.create_string_buffer(128)
                _hip._check(self.lib.st_fabric_unique_id(buf))
                ids[128 * c:128 * (c + 1)] = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8)
        if world > 1:
            carrier = ids.to(self.device) if dist.get_backend(group) == 'nccl' else ids
            dist.broadcast(carrier, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ids = carrier.cpu()
        raw = bytes(ids.tolist())
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip._check(self.lib.st_fabric_create(ctypes.byref(handle), raw[:128], raw[128:], self.rank, self.world,
                                                  1 if self.self_halo else 0))
        self.handle = handle
        self.cold = cold if cold is not None else DistFabric(rank, world, group)
        self.host_sync = False
        # pre-flight with a host-side deadline (on a private stream of the library), then agree on the verdict: one rank
        # falling back alone would hang the rest.  A rank whose own test failed ABORTS its communicators first: that ends
        # the RCCL kernels that may still be in flight (its peers' tests then time out and abort too), so that neither the
        # verdict's exchange nor anything queued later can sit behind a stuck kernel.
        failure = None
        with torch.cuda.device(self.device):
            if self.lib.st_fabric_selftest(self.handle, _hip._stream(), int(os.environ.get('ST_FABRIC_SELFTEST_MS', 30000))):
                failure = self.lib.st_last_error().decode('utf-8', 'replace')
                self.close(abort=True)
                torch.cuda.synchronize(self.device)
        if world > 1:
            verdict = torch.tensor([1.0 if failure else 0.0], device=self.device if dist.get_backend(group) == 'nccl' else 'cpu')
            self.cold.allmax(verdict)
            if verdict.item() and not failure:
                failure = 'the self-test failed on another rank'
        if failure:
            self.close(abort=True)
            raise RuntimeError(f'in-library RCCL transport unusable: {failure}')

    # cold-path operations go through torch.distributed
    def allreduce(self, tensor, op=None):
        self.cold.allreduce(tensor, op)

    def allmax(self, tensor):
        self.cold.allmax(tensor)

    def close(self, abort=False):
        """Destroy the two communicators: collectively after a completed run (every rank calls it), or ``abort=True`` when
        the ranks may have diverged (an exception, Ctrl-C): ncclCommAbort does not wait for operations in flight."""
        h, self.handle = getattr(self, 'handle', None), None
        if h:
            (self.lib.st_fabric_abort if abort else self.lib.st_fabric_destroy)(h)

    def __del__(self):
        self.close(abort=True)       # not closed by its owner: the run did not end normally


def run_phases(plan, fabric):
    """Drive one rank's phase machine to completion (after forward_begin / closure_begin)."""
    if isinstance(fabric, NativeFabric):
        with torch.cuda.device(plan.device):
            _hip._check(plan.lib.st_plan_closure_run(plan.handle, fabric.handle, _hip._stream()))
        return
    ex = _hip.Exchange()
    with torch.cuda.device(plan.device):
        stream = _hip._stream()
        while True:
            plan.next(stream, ex)
            if ex.kind == 0:
                return
            fabric.apply(ex, plan.device)


def _ext(handle, dev, cache={}):
    st = cache.get(handle)
    if st is None:
        st = cache[handle] = torch.cuda.ExternalStream(handle, device=dev)
    return st


def _delay_stream(dev, head, cache={}):
    st = cache.get((str(dev), head))
    if st is None:
        st = cache[(str(dev), head)] = torch.cuda.Stream(device=dev)
    return st


def owner_chain_us(owner_chains):
    """{head: microseconds from the head's reduction to its broadcast on its owner} of a measured stub run (after a
    device synchronize)."""
    return {h: e0.elapsed_time(e1) * 1e3 for h, (e0, e1) in owner_chains['measure'].items() if e0 is not None and e1 is not None}


def sleep_cycles_per_us(dev):
    """Calibration of torch.cuda._sleep on this device (its argument counts clock ticks, not time)."""
    torch.cuda._sleep(1000)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ticks = 20_000_000
    e0.record()
    torch.cuda._sleep(ticks)
    e1.record()
    torch.cuda.synchronize(dev)
    return ticks / (e0.elapsed_time(e1) * 1e3)


def run_phases_lockstep(plans, stub=False, wrap=False, owner_chains=None):
    """Single-process emulation of len(plans) ranks (all strips on one GPU): same kernels, same
    exchange descriptors, the transport replaced by device copies / an explicit sum, ordered on the streams the
    descriptors name exactly as a stream-ordered transport would be.  ``stub=True`` skips the data movement (timing
    runs of tools/strip_bench.py: per-rank critical path without a fabric; results are wrong).  ``wrap=True``: periodic
    boundary - the last plan's lower neighbour is the first (all plans must be middle strips); with ONE plan this is the
    reference result for DistFabric's ST_FABRIC_SELF_HALO mode.

    ``owner_chains`` (stub runs of ONE plan, owned heads): a rank that does not own a style head receives the head's
    result only after the OWNER's Newton-Schulz chains - a stubbed broadcast that returns at once leaves that wait out.
    ``{'rank': r, 'measure': {}}`` on an owner records, per owned head, the time from its reduction to its broadcast
    (events; read them with ``owner_chain_us`` after a synchronize); ``{'rank': r, 'delay_us': {head: us}, 'cycles_per_us':
    c}`` on any rank holds every head it does not own back by that long (a spinning one-thread kernel on a stream of its
    own between the head's reduction and its broadcast)."""
    dev = plans[0].device
    n_plans = len(plans)
    cur = torch.cuda.current_stream(dev)
    n_reduce = n_bcast = 0
    held = {}
    while True:
        exs = [p.next() for p in plans]
        kind = exs[0].kind
        assert all(e.kind == kind for e in exs), 'ranks disagree on the phase sequence'
        if kind == 0:
            return
        if stub and owner_chains is not None and kind in (4, 5) and n_plans == 1:
            # reductions are issued head 0 ... 4 (tap order), broadcasts 4 ... 0 (the order the backward pass needs them)
            ex = exs[0]
            head = n_reduce if kind == 4 else 4 - n_bcast
            n_reduce, n_bcast = n_reduce + (kind == 4), n_bcast + (kind == 5)
            hs = _ext(ex.stream, dev) if ex.stream else cur
            mine = ex.root == owner_chains['rank']
            if mine and 'measure' in owner_chains:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(hs)
                owner_chains['measure'].setdefault(head, [None, None])[0 if kind == 4 else 1] = ev
            elif not mine and owner_chains.get('delay_us', {}).get(head):
                if kind == 4:
                    side = _delay_stream(dev, head)
                    side.wait_stream(hs)
                    with torch.cuda.stream(side):
                        torch.cuda._sleep(int(owner_chains['delay_us'][head] * owner_chains['cycles_per_us']))
                        held[head] = torch.cuda.Event()
                        held[head].record(side)
                elif head in held:
                    hs.wait_event(held.pop(head))
        if kind == 3 or stub:
            continue
        streams = [(_ext(e.stream, dev) if e.stream else cur) for e in exs]
        if kind == 1:
            # copy r -> r +- 1 on the RECEIVER's stream behind the sender's pack; then the sender's stream waits for
            # the copy (its send buffer is reused by the next layer), as a completed send would guarantee
            pairs = []
            for r, ex in enumerate(exs):
                dn, up = ((r + 1) % n_plans, (r - 1) % n_plans) if wrap else (r + 1, r - 1)
                if ex.send_down:
                    pairs.append((r, dn, ex.send_down, exs[dn].recv_up, ex.count))
                if ex.send_up:
                    pairs.append((r, up, ex.send_up, exs[up].recv_down, ex.count))
            for a, b, src, dst, n in pairs:
                if streams[b] is not streams[a]:
                    streams[b].wait_stream(streams[a])
                with torch.cuda.stream(streams[b]):
                    view(dst, n, dev).copy_(view(src, n, dev))
            for a, b, *_ in pairs:
                if streams[b] is not streams[a]:
                    streams[a].wait_stream(streams[b])
        else:
            for st in streams:
                if st is not cur:
                    cur.wait_stream(st)
            bufs = [view(e.buffer, e.count, dev) for e in exs]
            if kind == 5:
                root = exs[0].root
                for r, b in enumerate(bufs):
                    if r != root:
                        b.copy_(bufs[root])
            else:
                total = torch.stack(bufs).sum(0)
                if kind == 2:
                    for b in bufs:
                        b.copy_(total)
                else:                                   # reduce: only the root's buffer is defined afterwards
                    bufs[exs[0].root].copy_(total)
            for st in streams:
                if st is not cur:
                    st.wait_stream(cur)


# ---- scale transition on strips (cold path, once per scale) ----------------------
def _cubic_rows(h_in, h_out, first, last, device):
    """Source rows and weights of output rows [first, last) for F.interpolate(mode='bicubic', align_corners=False)
    along H (ATen upsample_bicubic2d: scale = in / out, src = scale (dst + 0.5) - 0.5, cubic convolution A = -0.75,
    border rows clamped).  Indices stay on the host (every rank derives every rank's source interval from them),
    weights go to ``device``."""
    a = -0.75
    dst = torch.arange(first, last, dtype=torch.float32)
    src = dst.add(0.5).mul(torch.tensor(h_in, dtype=torch.float32) / h_out).sub(0.5)
    i0 = torch.floor(src)
    t = src - i0
    i0 = i0.to(torch.int64)

    def conv1(x):
        return ((a + 2) * x - (a + 3)) * x * x + 1

    def conv2(x):
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a
    weights = [conv2(t + 1), conv1(t), conv1(1 - t), conv2(2 - t)]
    index = [(i0 + j).clamp(0, h_in - 1) for j in (-1, 0, 1, 2)]
    return index, [w.to(device) for w in weights]


def _linear_rows(h_in, h_out, first, last, device):
    """The same for mode='bilinear' (src clamped at 0, second row clamped at the border)."""
    dst = torch.arange(first, last, dtype=torch.float32)
    src = dst.add(0.5).mul(torch.tensor(h_in, dtype=torch.float32) / h_out).sub(0.5).clamp_min(0)
    i0 = torch.floor(src)
    l1 = src - i0
    i0 = i0.to(torch.int64).clamp(max=h_in - 1)
    return [i0, (i0 + 1).clamp(max=h_in - 1)], [(1 - l1).to(device), l1.to(device)]


def resample_strip(local, old_rows, rank, world, old_height, new_rows, new_size, mode, host_sync=False, group=None):
    """This rank's strip [new_rows[rank]) of F.interpolate(full, new_size, mode=mode), computed from row strips of the
    old tensor - the shard-aware form of the scale transition (reference style_transfer.py:279-295,420): nothing is
    gathered, every rank resamples only its own rows.

    ``local``: [1, C, h, W_old] - this rank's strip ``old_rows[rank]`` of the old tensor, or the full tensor when
    ``old_rows`` is None (the previous scale was too small to shard and every rank holds all of it).
    Width: F.interpolate on the local rows with an unchanged height (scale 1 along H is the identity: t = 0, weights
    0, 1, 0, 0) - exactly the full tensor's width pass.  Height: the few source rows this rank's output needs from
    its neighbours' strips travel point to point, then the 4 (bicubic) / 2 (bilinear) taps are applied with the
    weights ATen computes.  Agrees with the gather-resample-cut form to fp32 rounding (separable filter, same tap
    order)."""
    import torch.distributed as dist
    from torch.nn import functional as F
    new_h, new_w = new_size
    nb, ne = new_rows[rank]
    dev = local.device
    x = F.interpolate(local, size=(local.shape[2], new_w), mode=mode) if local.shape[3] != new_w else local
    index, weights = (_cubic_rows if mode == 'bicubic' else _linear_rows)(old_height, new_h, nb, ne, dev)

    def needed(r):
        fb, fe = new_rows[r]
        idx, _ = (_cubic_rows if mode == 'bicubic' else _linear_rows)(old_height, new_h, fb, fe, 'cpu')
        return int(min(i.min() for i in idx)), int(max(i.max() for i in idx))

    lo, hi = needed(rank)
    if old_rows is None:
        slab = x[:, :, lo:hi + 1]
    else:
        ob, oe = old_rows[rank]
        pieces, ops, recvs = {}, [], []
        for r in range(world):
            if r == rank:
                continue
            rlo, rhi = needed(r)
            sb, se = max(rlo, ob), min(rhi + 1, oe)               # rows I own that rank r needs
            if sb < se:
                ops.append(dist.P2POp(dist.isend, x[:, :, sb - ob:se - ob].contiguous(), r, group))
            rb, re = old_rows[r]
            gb, ge = max(lo, rb), min(hi + 1, re)                  # rows rank r owns that I need
            if gb < ge:
                buf = x.new_empty((x.shape[0], x.shape[1], ge - gb, new_w))
                recvs.append((gb, buf))
                ops.append(dist.P2POp(dist.irecv, buf, r, group))
        if ops:
            if host_sync and x.is_cuda:
                torch.cuda.synchronize(dev)
            for work in dist.batch_isend_irecv(ops):
                work.wait()
            if host_sync and x.is_cuda:
                torch.cuda.synchronize(dev)
        mb, me = max(lo, ob), min(hi + 1, oe)
        if mb < me:
            pieces[mb] = x[:, :, mb - ob:me - ob]
        for gb, buf in recvs:
            pieces[gb] = buf
        slab = torch.cat([pieces[k] for k in sorted(pieces)], dim=2)
        assert slab.shape[2] == hi + 1 - lo, 'strip resample: source rows missing'
    out = None
    for idx, w in zip(index, weights):
        term = slab.index_select(2, (idx - lo).to(dev)) * w.view(1, 1, -1, 1)
        out = term if out is None else out + term
    return out.contiguous()


# ---- L-BFGS on strips (reference style_transfer.py:464-465) ------------------------------------------
class StripLBFGS:
    """``torch.optim.LBFGS(params, max_iter=1, history_size=h)`` - the reference's configuration: one quasi-Newton
    update per ``step``, fixed step length (no line search) - on an image that is cut into row strips: every rank holds
    its strip of the iterate, of the gradient and of the curvature pairs, and every inner product / norm of
    ``LBFGS.step`` (``y.s``, ``y.y``, the two-loop recursion's ``s_i.q`` and ``y_i.r``, ``|g|_1``, ``|g|_inf``, ``g.d``) is
    completed by a sum / max over the ranks.  The recursion is torch's, statement for statement (torch/optim/lbfgs.py,
    ``step`` with ``line_search_fn=None`` and ``max_iter == 1``); sums are formed in another order than a single
    tensor's ``dot``, which this quasi-Newton recursion amplifies like any other rounding-level change (the reference's
    own trace moves by 2e-2 after seven iterations when it runs on 1 thread instead of 8, tests/golden/make_golden.py).

    ``grad``: the buffer ``closure()`` writes this rank's gradient strip into; ``allsum(t)`` / ``allmax(t)``: in-place
    sum / max of a small tensor over the ranks."""

    def __init__(self, param, grad, allsum, allmax, lr=1.0, history_size=10, tolerance_grad=1e-7, tolerance_change=1e-9):
        self.param, self.grad, self.allsum, self.allmax = param, grad, allsum, allmax
        self.lr, self.history_size = lr, history_size
        self.tolerance_grad, self.tolerance_change = tolerance_grad, tolerance_change
        self.n_iter = 0
        self.d = self.t = self.prev_flat_grad = None
        self.old_dirs, self.old_stps, self.ro, self.H_diag = [], [], [], 1
        self.al = [None] * history_size

    def _sum(self, value):
        value = value.reshape(1).clone()
        self.allsum(value)
        return value[0]

    def _max(self, value):
        value = value.reshape(1).clone()
        self.allmax(value)
        return value[0]

    def _dot(self, a, b):
        return self._sum(a.dot(b))

    @torch.no_grad()
    def step(self, closure):
        orig_loss = closure()                      # the global loss (identical on every rank); fills self.grad
        flat_grad = self.grad.reshape(-1)
        if self._max(flat_grad.abs().max()) <= self.tolerance_grad:
            return orig_loss
        self.n_iter += 1
        if self.n_iter == 1:
            d = flat_grad.neg()
            self.old_dirs, self.old_stps, self.ro, self.H_diag = [], [], [], 1
        else:
            y = flat_grad.sub(self.prev_flat_grad)
            s = self.d.mul(self.t)
            ys = self._dot(y, s)
            if ys > 1e-10:
                if len(self.old_dirs) == self.history_size:
                    self.old_dirs.pop(0)
                    self.old_stps.pop(0)
                    self.ro.pop(0)
                self.old_dirs.append(y)
                self.old_stps.append(s)
                self.ro.append(1.0 / ys)
                self.H_diag = ys / self._dot(y, y)
            num_old = len(self.old_dirs)
            q = flat_grad.neg()
            for i in range(num_old - 1, -1, -1):
                self.al[i] = self._dot(self.old_stps[i], q) * self.ro[i]
                q.add_(self.old_dirs[i], alpha=-self.al[i])
            d = r = torch.mul(q, self.H_diag)
            for i in range(num_old):
                be_i = self._dot(self.old_dirs[i], r) * self.ro[i]
                r.add_(self.old_stps[i], alpha=self.al[i] - be_i)
        if self.prev_flat_grad is None:
            self.prev_flat_grad = flat_grad.clone(memory_format=torch.contiguous_format)
        else:
            self.prev_flat_grad.copy_(flat_grad)
        if self.n_iter == 1:
            t = min(1.0, 1.0 / float(self._sum(flat_grad.abs().sum()))) * self.lr
        else:
            t = self.lr
        self.d, self.t = d, t
        if self._dot(flat_grad, d) > -self.tolerance_change:      # directional derivative below tolerance: no move
            return orig_loss
        self.param.reshape(-1).add_(d, alpha=t)                   # fixed-step move (no line search)
        return orig_loss


# ---- target construction on strips (cold path, once per scale) -------------------------------------
STYLE_LAYERS = [1, 6, 11, 20, 29]


def set_targets(plan, content_strip, style_strips, style_weights, run, allreduce, style_plans=None):
    """Content target from this rank's strip of the content image; style targets from raw moment sums
    all-reduced over the strips of each style image (``style_plans[i]`` = the strip plan of style i on
    this rank; defaults to ``plan`` when the style image has the content's size).
    ``run(plan)`` drives the phase machine, ``allreduce(tensor)`` sums over ranks."""
    plan.forward_begin(content_strip, 22)
    run(plan)
    plan.set_content_target_from_forward()
    blended = {}
    for i, strip in enumerate(style_strips):
        sp = plan if style_plans is None else style_plans[i]
        sp.forward_begin(strip, 29)
        run(sp)
        for layer in STYLE_LAYERS:
            sums = sp.moment_sums(layer)
            allreduce(sums)
            c = {1: 64, 6: 128, 11: 256, 20: 512, 29: 512}[layer]
            level = {1: 0, 6: 1, 11: 2, 20: 3, 29: 4}[layer]
            npix = float((sp.global_height >> level) * (sp.width >> level))
            srm = (sums[:c * c] / npix).reshape(c, c) * style_weights[i]
            mean = (sums[c * c:] / npix) * style_weights[i]
            if layer not in blended:
                blended[layer] = [mean, srm]
            else:
                blended[layer][0] += mean
                blended[layer][1] += srm
    for idx, layer in enumerate(STYLE_LAYERS):
        plan.set_style_target(idx, blended[layer][0].contiguous(), blended[layer][1].contiguous())
