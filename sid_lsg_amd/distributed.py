"""Process-group helpers with the reference's surface (torch_utils/distributed.py:14-58):
init / get_rank / get_local_rank / get_world_size / should_stop / update_progress / print0.
One process per GPU; backend 'nccl' is RCCL on ROCm (xGMI inside a node), 'gloo' on CPU.
"""
import os

import torch


def init(backend=None):
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('LOCAL_RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC for RCCL on this platform
    if backend is None:
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    if not torch.distributed.is_initialized():
        torch.distributed.init_process_group(backend=backend, init_method='env://')
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))


def get_rank():
    return torch.distributed.get_rank() if torch.distributed.is_initialized() else 0


def get_local_rank():
    return int(os.environ.get('LOCAL_RANK', '0'))


def get_world_size():
    return torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1


def should_stop():
    return False


def update_progress(cur, total):
    _ = cur, total


def print0(*args, **kwargs):
    if get_rank() == 0:
        print(*args, **kwargs)


def _grad_side_streams(t):
    if not t.is_cuda:
        return []
    from . import ops
    return ops.grad_streams(t.device)


_DRYRUN = os.environ.get('SIDLSG_EXCHANGE_DRYRUN', '0') == '1'      # A/B: FlatGradReducer issues no collective (WRONG for world > 1)


def _flush_deferred(t):
    """Gradients are about to be read: run the norm kernels' queued dgamma / dbeta reductions first (ops.flush_deferred)."""
    if t.is_cuda:
        from . import ops
        ops.flush_deferred()


class FlatGradReducer:
    """Data-parallel gradient exchange on a network's flat fp32 gradient buffer.

    The reference wraps each net in DistributedDataParallel (sid_training_loop.py:316-323): ~138
    25-MB NCCL buckets per network.  Here the gradients already live in one flat buffer, so the exchange
    is `nbuckets` large all-reduce(SUM) calls issued on a dedicated communication stream; the mean is folded
    into the fused optimizer kernel (grad_scale = 1/world), so the buffer is read exactly once afterwards.
    xGMI is point-to-point (7 links/GPU): few LARGE messages let RCCL use all rings/links at full rate.
    The caller overlaps: start() right after the backward of the last accumulation round, then prepares
    the next phase's noise / text conditioning (which does not read the parameters), then wait().
    """

    def __init__(self, nbuckets=4, group=None, exchange_dtype=None, min_world=2):
        """exchange_dtype: None / torch.float32 = all-reduce the fp32 gradients in place (the reference's DDP semantics);
        torch.bfloat16 (or SIDLSG_GRAD_EXCHANGE=bf16) = OPT-IN halved xGMI traffic: each message is rounded to bf16 into a
        staging buffer, all-reduced in bf16 and written back to the fp32 gradient buffer.  This changes the numbers (8
        mantissa bits per rank contribution, bf16 accumulation inside the collective) and is therefore off by default.
        min_world: the exchange is skipped below this world size (2: a single rank has nothing to exchange; 1 makes a
        one-rank job run the collectives anyway -- how the RCCL path is exercised on a one-GPU box)."""
        self.nbuckets, self.group, self.min_world = nbuckets, group, int(min_world)
        self.stream = torch.cuda.Stream() if torch.cuda.is_available() else None
        self.handles = []
        if exchange_dtype is None and os.environ.get('SIDLSG_GRAD_EXCHANGE', 'fp32').lower() in ('bf16', 'bfloat16'):
            exchange_dtype = torch.bfloat16
        if exchange_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError('exchange_dtype must be None, torch.float32 or torch.bfloat16')
        self.exchange_dtype = exchange_dtype if exchange_dtype is not None else torch.float32
        self._pending = []          # (fp32 view, bf16 staging tensor) pairs to write back in wait()
        # measurement (bench.py --gpus N, tests): with `timing` on, every message is bracketed by events on the communication
        # stream and every wait() by events on the stream that waits, so a multi-GPU run reports how long the compute stream
        # actually stood still for the exchange (`exposed`) next to the all-reduce rate it saw
        self.timing = False
        self._t_msgs, self._t_waits = [], []

    def enable_timing(self, on=True):
        self.timing = bool(on) and self.stream is not None
        self._t_msgs, self._t_waits = [], []

    def timing_report(self, world=None):
        """After a device synchronisation: dict(exposed_ms={tag: total}, waits={tag: count}, messages, bytes, comm_ms,
        algbw_GBps = bytes / time of the all-reduce messages, busbw_GBps = algbw * 2 (n - 1) / n)."""
        world = get_world_size() if world is None else world
        exposed, waits = {}, {}
        for tag, e0, e1 in self._t_waits:
            exposed[tag] = exposed.get(tag, 0.0) + e0.elapsed_time(e1)
            waits[tag] = waits.get(tag, 0) + 1
        nbytes = sum(b for b, _, _ in self._t_msgs)
        ms = sum(e0.elapsed_time(e1) for _, e0, e1 in self._t_msgs)
        alg = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None
        return dict(exposed_ms=exposed, waits=waits, messages=len(self._t_msgs), bytes=nbytes, comm_ms=ms, algbw_GBps=alg,
                    busbw_GBps=(alg * 2 * (world - 1) / world) if alg is not None else None)

    def _all_reduce(self, view):
        """One message (called with the communication stream current, if there is one)."""
        e0 = None
        if self.timing and view.is_cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if _DRYRUN:          # measurement only: the whole stream choreography of the exchange without the collective itself
            if e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self._t_msgs.append((view.numel() * view.element_size(), e0, e1))
            return
        if self.exchange_dtype == torch.bfloat16:
            stage = view.to(torch.bfloat16)
            h = torch.distributed.all_reduce(stage, group=self.group, async_op=True)
            self._pending.append((view, stage))
        else:
            stage = view
            h = torch.distributed.all_reduce(view, group=self.group, async_op=True)
        self.handles.append(h)
        if e0 is not None:
            # the backend runs the collective on its own stream: joining it into the communication stream here (device-side
            # only) makes the event pair bracket exactly this message, and the next message's start event wait for it
            h.wait()
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._t_msgs.append((stage.numel() * stage.element_size(), e0, e1))

    def start_range(self, flat_grad, lo, hi, max_elems=1 << 28):
        """all-reduce flat_grad[lo:hi] (in messages of <= max_elems elements = 1 GiB fp32) on the communication stream,
        ordered after everything already enqueued on the current stream.  Several ranges may be in flight; wait() joins all."""
        if get_world_size() < self.min_world or hi <= lo:
            return
        _flush_deferred(flat_grad)
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            for side in _grad_side_streams(flat_grad):      # weight gradients are produced on their own stream (ops.py)
                self.stream.wait_stream(side)
            with torch.cuda.stream(self.stream):
                for o in range(lo, hi, max_elems):
                    self._all_reduce(flat_grad[o:min(hi, o + max_elems)])
        else:
            for o in range(lo, hi, max_elems):
                self._all_reduce(flat_grad[o:min(hi, o + max_elems)])

    def start_segment(self, flat_grad, ranges):
        """start_range for every (lo, hi) of one segment of HipUNet2DCondition.grad_segments()."""
        for lo, hi in ranges:
            self.start_range(flat_grad, lo, hi)

    def start(self, flat_grad):
        if get_world_size() < self.min_world:
            return
        _flush_deferred(flat_grad)
        n = flat_grad.numel()
        step = (n + self.nbuckets - 1) // self.nbuckets
        step = (step + 1023) // 1024 * 1024
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            for side in _grad_side_streams(flat_grad):
                self.stream.wait_stream(side)
            with torch.cuda.stream(self.stream):
                for o in range(0, n, step):
                    self._all_reduce(flat_grad[o:o + step])
        else:
            for o in range(0, n, step):
                self._all_reduce(flat_grad[o:o + step])

    def wait(self, tag='all'):
        t0 = None
        if self.timing:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
        self._wait()
        if t0 is not None:
            t1 = torch.cuda.Event(enable_timing=True)
            t1.record()
            self._t_waits.append((tag, t0, t1))

    def _wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        if self._pending:                               # bf16 exchange: reduced values back into the fp32 gradients,
            for view, stage in self._pending:           # on the CURRENT stream -- the one h.wait() has just ordered after
                view.copy_(stage)                       # the collectives
                if stage.is_cuda:
                    stage.record_stream(torch.cuda.current_stream())     # allocated on the communication stream
            self._pending = []
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
