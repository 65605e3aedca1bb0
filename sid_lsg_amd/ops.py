"""Host side of the HIP kernels: thin launch wrappers + torch.autograd.Function glue.

Follows the reference's op-wrapper convention (torch_utils/ops/bias_act.py: public function ->
cached autograd.Function -> plugin call on the current stream) with one deliberate difference:
there is no `impl='ref'` fallback.  If the library is absent or a tensor is not on the GPU these
raise.  PyTorch is used only for device memory, the current stream and autograd bookkeeping.

Conventions: activations are bf16 (production) or fp32 (the fp32-accurate parity mode: same call graph, the `_f32`
entry-point family of include/sidlsg_hip.h, chosen by the dtype of the activation tensor), NHWC / token-major and
contiguous; parameters are fp32
"masters" whose `.grad` is a pre-allocated fp32 view into the network's flat gradient buffer --
weight/bias gradients are ACCUMULATED IN PLACE by the kernels (atomics / +=) and the autograd
functions return None for them, so no per-parameter gradient tensors are ever materialised.
"""
import ctypes
import weakref
import os

import torch

from ._lib import lib

BF16 = torch.bfloat16
F32 = torch.float32


def _p(t):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


ACT = 'act'   # _chk: an activation tensor (bf16 or fp32)


def _chk(t, dtype=None):
    if not t.is_cuda:
        raise RuntimeError('sid_lsg_amd ops need CUDA(HIP) tensors: there is no CPU fallback')
    if dtype is ACT:
        if t.dtype not in (BF16, F32):
            raise RuntimeError(f'expected a bf16 or fp32 activation tensor, got {t.dtype}')
    elif dtype is not None and t.dtype != dtype:
        raise RuntimeError(f'expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise RuntimeError('expected a contiguous tensor')
    return t


def _fn(name, dtype, bf16_suffix=''):
    """C entry point of the activation dtype: `sidlsg_<name><bf16_suffix>` for bf16, `sidlsg_<name>_f32` for fp32."""
    return getattr(lib, f'sidlsg_{name}_f32' if dtype == F32 else f'sidlsg_{name}{bf16_suffix}')


def _wants_grad(p):
    """Weight gradients are wanted iff the parameter requires grad.  Its `.grad` is the pre-bound view of the network's flat
    gradient buffer: HipUNet2DCondition re-binds (and zeroes) the views at the start of every forward if a foreign optimizer
    ran `zero_grad(set_to_none=True)` (sid_training_loop.py:390,469); a None that appears between a forward and its backward
    is re-bound here the same way."""
    if p is None or not p.requires_grad:
        return False
    if p.grad is None:
        # zero_grad(set_to_none=True) between a forward and its backward (a retained graph, an accumulation round of a
        # foreign optimizer): None means zero to the caller, so the flat-buffer view comes back zeroed
        flat = getattr(p, '_flat_grad', None)
        if flat is None:
            raise RuntimeError('parameter requires grad but has no gradient buffer to bind (not a HipUNet2DCondition parameter?): '
                               'weight gradients would be lost')
        flat.zero_()
        p.grad = flat
    return True


def _take_assign(weight, dtype):
    """True when this weight gradient may OVERWRITE `weight.grad`: the fused optimizer left the view un-zeroed after its last step
    and marked the parameter (`_grad_assign`, optim.FusedAdamEMA with a network that has an assign plan); the first weight gradient
    after that step takes the mark.  Saves the optimizer's zero stores and the read of dW here (sidlsg_*wgrad_assign_bf16: bit-identical
    to accumulating onto zeros).  A launch that cannot overwrite (fp32 activations) zeroes the view instead and accumulates."""
    if not getattr(weight, '_grad_assign', False):
        return False
    weight._grad_assign = False
    if dtype != BF16:
        weight.grad.zero_()
        return False
    return True



# ------------------------------------------------------------------------------------------------
_workspace = {}


def _dev_key(device):
    """Cache key of a device: its index; an index-less 'cuda' device means the CURRENT device (not device 0)."""
    idx = torch.device(device).index
    return torch.cuda.current_device() if idx is None else idx


def ensure_workspace(device, nbytes=512 << 20):
    """fp32 split-K scratch for the small-pixel-count convs/GEMMs (allocated once per device through torch's allocator)."""
    key = _dev_key(device)
    if key not in _workspace:
        ws = torch.empty(nbytes // 4, device=device, dtype=F32)
        lib.sidlsg_set_workspace(ws.data_ptr(), ws.numel() * 4)
        _workspace[key] = ws
    return _workspace[key]


_stream_ws = {}
_side_streams = {}


def side_stream(device):
    """The process-wide second compute stream of `device` (with its private split-K workspace): created once, shared by
    every SiDStep -- the C library keeps at most 4 stream workspaces."""
    key = _dev_key(device)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
        ensure_stream_workspace(_side_streams[key])
    return _side_streams[key]


# ---- weight gradients on their own stream ---------------------------------------------------------------------------------
# A weight-gradient launch is one round of equal-work blocks: they start together, wait for their tile DMAs together and end in
# a chip-wide burst of slab writes followed by a small reduction kernel (tools/ab/wgrad_trace.py) -- MFMA and HBM idle in
# turns.  dW of a layer depends only on (dY, saved x), not on the backward-data chain, so these launches go to a second stream
# and their idle phases are filled by the dgrad GEMMs / attention / norm kernels of the main stream (and vice versa).
# Ordering: the side stream waits for the main stream at each launch (dY exists), the tensors are record_stream()-ed, and
# the main stream (plus, through grad_streams(), the gradient-exchange stream) waits for the side stream at the end of the
# backward pass (autograd engine callback), i.e. before anything may read .grad.  SIDLSG_WGRAD_STREAM=0 turns it off.
_WGRAD_SIDE = os.environ.get('SIDLSG_WGRAD_STREAM', '1') != '0'
# SIDLSG_WGRAD_STREAMS = n > 1: weight-gradient launches rotate over n streams, so that n of them can share the chip (with
# SIDLSG_WGRAD_SLOTS = 512 / n each launch splits its pixel range for 1 / n of the chip: fewer, longer blocks and 1 / n of the
# partial-sum slab traffic per layer).  A/B knob; default 1.
_WGRAD_NSTREAMS = max(1, int(os.environ.get('SIDLSG_WGRAD_STREAMS', '1')))
_wgrad_streams = {}
_wgrad_rr = {}
_wgrad_join_armed = set()


# SIDLSG_WGRAD_PRIO=low: the weight-gradient streams are created with the LOWEST HIP stream priority (hipStreamCreateWithPriority; torch
# itself only offers priorities at or above its default stream's), so that the workgroup dispatcher serves the main stream -- the
# backward-data chain, i.e. the critical path -- first and the weight gradients fill what it leaves idle.  A/B knob (profiles/r06_*prio*).
_WGRAD_PRIO = os.environ.get('SIDLSG_WGRAD_PRIO', '')
_hip_rt = None


def _low_priority_stream(device):
    global _hip_rt
    if _hip_rt is None:
        _hip_rt = ctypes.CDLL('libamdhip64.so')          # already in the process (torch's own runtime: same soname)
    least, greatest = ctypes.c_int(0), ctypes.c_int(0)
    if _hip_rt.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) != 0:
        raise RuntimeError('hipDeviceGetStreamPriorityRange failed')
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = _hip_rt.hipStreamCreateWithPriority(ctypes.byref(h), ctypes.c_uint(1), ctypes.c_int(least.value))      # 1 = hipStreamNonBlocking
    if rc != 0 or not h.value:
        raise RuntimeError(f'hipStreamCreateWithPriority failed with {rc}')
    return torch.cuda.ExternalStream(h.value, device=device)


def _wgrad_stream_list(device):
    key = _dev_key(device)
    if key not in _wgrad_streams:
        lst = [(_low_priority_stream(device) if _WGRAD_PRIO == 'low' else torch.cuda.Stream(device=device)) for _ in range(_WGRAD_NSTREAMS)]
        for st in lst:
            ensure_stream_workspace(st, nbytes=(256 << 20) // _WGRAD_NSTREAMS if _WGRAD_NSTREAMS > 1 else 256 << 20)
        _wgrad_streams[key] = lst
    return _wgrad_streams[key]


def wgrad_stream(device):
    """The stream of the next weight-gradient launch (round robin when several are configured)."""
    lst = _wgrad_stream_list(device)
    key = _dev_key(device)
    i = _wgrad_rr.get(key, 0)
    _wgrad_rr[key] = (i + 1) % len(lst)
    return lst[i]


def grad_streams(device):
    """Side streams that may still be writing parameter gradients of `device` (a gradient exchange must wait for them)."""
    return list(_wgrad_streams.get(_dev_key(device), ()))


class _OnWgradStream:
    """with _OnWgradStream(dy, x): <wgrad launches> -- inside a torch.autograd.Function.backward only."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.on = _WGRAD_SIDE and tensors[0].dtype == BF16

    def __enter__(self):
        if not self.on:
            return self
        dev = self.tensors[0].device
        self.side = wgrad_stream(dev)
        self.side.wait_stream(torch.cuda.current_stream(dev))
        self.cm = torch.cuda.stream(self.side)
        self.cm.__enter__()
        return self

    def __exit__(self, *exc):
        if not self.on:
            return False
        self.cm.__exit__(*exc)
        for t in self.tensors:
            t.record_stream(self.side)
        # one join per backward pass and device, keyed by the autograd graph-task id (a pass that died with an exception
        # must not leave the next one un-joined)
        key = (_dev_key(self.tensors[0].device), torch._C._current_graph_task_id())
        if key not in _wgrad_join_armed:
            dev = self.tensors[0].device

            def join():
                _wgrad_join_armed.discard(key)
                for side in grad_streams(dev):
                    torch.cuda.current_stream(dev).wait_stream(side)
            if key[1] < 0:                       # not inside a backward pass: join right away
                join()
            else:
                # ids never repeat: anything left for THIS device is from a pass that did not finish
                for stale in [k for k in _wgrad_join_armed if k[0] == key[0]]:
                    _wgrad_join_armed.discard(stale)
                _wgrad_join_armed.add(key)
                torch.autograd.Variable._execution_engine.queue_callback(join)
        return False


# ---- deferred parameter-gradient reductions of the norm backward kernels ---------------------------------------------------
# (csrc/norm.hip "deferred parameter-gradient reductions"): the dgamma / dbeta reductions of a backward pass -- ~80 per trainable
# network and pass, each a tiny launch on the critical stream -- are queued by the library and run as ONE launch per stream when the
# backward pass ends (autograd engine callback), when a gradient-exchange marker fires (_GradReady) or when somebody is about to
# read gradients (flush_deferred(): the fused optimizer and the reducer call it).  The partial-sum workspaces are kept alive here
# until then.  SIDLSG_DEFER_REDUCE=0: one reduction launch per layer as before (A/B, tests).
_DEFER = os.environ.get('SIDLSG_DEFER_REDUCE', '1') != '0'
_defer_streams = {}          # stream handle -> [torch stream, [workspaces]]
_defer_armed = set()


def _defer_begin(device):
    """Before a norm-backward launch that reduces parameter gradients: deferral on for the current stream."""
    if not _DEFER:
        return None
    st = torch.cuda.current_stream(device)
    h = st.cuda_stream
    if h not in _defer_streams:
        _defer_streams[h] = [st, []]
    lib.sidlsg_defer_reductions.raw(h, 1)
    return h


def _defer_end(h, ws):
    """After the launch: keep its partial sums alive; flush at the end of this backward pass (right away outside one)."""
    if h is None:
        return
    lib.sidlsg_defer_reductions.raw(h, 2)        # only the call in between was deferred (direct users of the C ABI never are)
    _defer_streams[h][1].append(ws)
    if torch._C._current_graph_task_id() < 0:
        flush_deferred()
    else:
        _arm_end_of_backward_flush()


def flush_deferred():
    """Launch every queued reduction (one kernel per stream that has any) and order the current stream after them."""
    flush_wgrad_queues()
    for h, (st, keep) in _defer_streams.items():
        if not keep:
            continue
        if lib.sidlsg_flush_reductions.raw(h) < 0:
            raise RuntimeError('sidlsg_flush_reductions failed')
        keep.clear()
        cur = torch.cuda.current_stream(st.device)
        if cur.cuda_stream != h:
            cur.wait_stream(st)


# ---- grouped dense weight gradients (csrc/gemm.hip "grouped dense weight gradients") -----------------------------------------------
# The C x C projections of a transformer block (to_out of both attentions, the cross-attention's to_q, proj_in / proj_out) and its
# 77-token k|v projection each need ~56 pixel splits to fill the chip alone.  Their weight gradients are QUEUED here (operands kept
# alive) and launched eight at a time as one grid + one slab reduction (sidlsg_wgrad_group_bf16); whatever is queued is launched when a
# gradient-exchange marker fires, when the backward pass ends, or when somebody is about to read gradients (flush_deferred).
# Only layers that would take the 128 x 128-tile kernel anyway; the wide FF / q|k|v layers keep their 160 x 160-tile launches.
# SIDLSG_WGRAD_GROUP=0: one launch per layer (A/B, tests).
_WG_GROUP = os.environ.get('SIDLSG_WGRAD_GROUP', '1') != '0'
_WG_GROUP160 = os.environ.get('SIDLSG_WGRAD_GROUP160', '1') != '0'      # A/B: the wide layers launch alone (round-5 first version)
_WG_MAX = 8
_WG_MAX160 = 3         # the three wide layers of one transformer block (FF-out, FF-in, q|k|v in backward order): 60 tiles of 160 x 160
_wg_queues = {}          # (stream handle, tile class) -> [torch stream, [jobs], tile class]
_wg_queued_dw = set()    # data_ptr of every dW with a queued job


class _WgJob(ctypes.Structure):
    _fields_ = [('dY', ctypes.c_void_p), ('A', ctypes.c_void_p), ('dW', ctypes.c_void_p), ('dBias', ctypes.c_void_p),
                ('ldy', ctypes.c_int), ('lda', ctypes.c_int), ('M', ctypes.c_int), ('N', ctypes.c_int), ('K', ctypes.c_int),
                ('assign', ctypes.c_int), ('pad', ctypes.c_int * 2)]


def _queue_dense_wgrad(dy, x, dw, dbias, M, N, K, assign):
    """True: queued for a grouped launch (the caller must not launch it).
    The jobs of one grouped launch must have DISTINCT dW (include/sidlsg_hip.h): a parameter that reaches its weight gradient twice in
    one backward pass (a shared / re-applied layer; one use queued and another one launched directly) first flushes what is queued,
    so the two gradients stay ordered on the weight-gradient stream (the `assign` mark was taken by the earlier one)."""
    if dw.data_ptr() in _wg_queued_dw:
        flush_wgrad_queues()
    if not _WG_GROUP or dy.dtype != BF16 or x.dtype != BF16 or not dy.is_cuda:
        return False
    if (N | K | dy.stride(0) | x.stride(0)) & 7 or (dy.data_ptr() | x.data_ptr()) & 15:
        return False
    # two classes, as in launch_wgrad: the layers of the 160 x 160-tile kernel (q|k|v, FF-in, FF-out: N, K multiples of 160, N != K) are
    # grouped among themselves (sidlsg_wgrad_group160_bf16), everything else on 128 x 128 tiles
    t160 = N % 160 == 0 and K % 160 == 0 and N != K and (M >= 4096 or N * K >= (8 << 20))
    if t160 and not _WG_GROUP160:
        return False
    if torch._C._current_graph_task_id() < 0:
        return False          # outside a backward pass nobody would flush
    st = torch.cuda.current_stream(dy.device)
    q = _wg_queues.setdefault((st.cuda_stream, t160), [st, [], t160])
    q[1].append((dy, x, dw, dbias, M, N, K, 1 if assign else 0))
    _wg_queued_dw.add(dw.data_ptr())
    if len(q[1]) >= (_WG_MAX160 if t160 else _WG_MAX):
        _flush_wgrad_queue(q)
    _arm_end_of_backward_flush()
    return True


def _flush_wgrad_queue(q):
    st, jobs, t160 = q
    if not jobs:
        return
    arr = (_WgJob * len(jobs))()
    for i, (dy, x, dw, dbias, M, N, K, assign) in enumerate(jobs):
        arr[i].dY, arr[i].A, arr[i].dW, arr[i].dBias = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (dbias.data_ptr() if dbias is not None else None)
        arr[i].ldy, arr[i].lda, arr[i].M, arr[i].N, arr[i].K, arr[i].assign = dy.stride(0), x.stride(0), M, N, K, assign
    tensors = [t for j in jobs for t in j[:2]]
    with torch.cuda.stream(st):          # the stream the operands were produced on: the weight-gradient stream waits for IT
        with _OnWgradStream(*tensors):
            (lib.sidlsg_wgrad_group160_bf16 if t160 else lib.sidlsg_wgrad_group_bf16)(ctypes.addressof(arr), len(jobs), _s())
    for j in jobs:
        _wg_queued_dw.discard(j[2].data_ptr())
    jobs.clear()


def flush_wgrad_queues():
    for q in _wg_queues.values():
        _flush_wgrad_queue(q)


def _discard_stale_backward_state():
    """A backward pass that raised left its queues behind (weight-gradient jobs, dgamma / dbeta reductions on the C side, partial-sum
    workspaces, the shared column-gradient buffer): launching them into .grad during the NEXT pass -- possibly after an optimizer step
    re-marked the gradients -- would apply stale gradients silently.  Drop them."""
    for q in _wg_queues.values():
        q[1].clear()
    _wg_queued_dw.clear()
    for h, (st, keep) in _defer_streams.items():
        lib.sidlsg_defer_reductions.raw(h, 3)
        keep.clear()
    for holder in list(_ColumnGrads.live):
        holder.buf = None


def _arm_end_of_backward_flush():
    tid = torch._C._current_graph_task_id()
    if tid >= 0 and tid not in _defer_armed:
        if _defer_armed:                 # ids never repeat: anything left is from a pass that did not finish
            _defer_armed.clear()
            _discard_stale_backward_state()
        _defer_armed.add(tid)

        def done():
            _defer_armed.discard(tid)
            flush_deferred()
        torch.autograd.Variable._execution_engine.queue_callback(done)


def ensure_stream_workspace(stream, nbytes=256 << 20):
    """Private split-K scratch for a side stream that runs contractions concurrently with the main one."""
    key = stream.cuda_stream
    if key not in _stream_ws:
        ws = torch.empty(nbytes // 4, device=stream.device, dtype=F32)
        lib.sidlsg_set_stream_workspace(key, ws.data_ptr(), ws.numel() * 4)
        _stream_ws[key] = ws
    return _stream_ws[key]


class Fp8Weight:
    """A frozen GEMM / conv weight in the fp8-weight format of include/sidlsg_hip.h: e4m3 bytes [N][K] + one fp32 scale per
    output channel.  Stands where the bf16 compute copy stands in gemm() / conv3x3() (forward only: the backward-data
    operand of a layer stays bf16)."""
    dtype = 'fp8_e4m3'

    def __init__(self, w_bf16):
        w = w_bf16.detach()
        if w.dtype != BF16 or w.ndim != 2 or not w.is_contiguous() or w.shape[1] % 16:
            raise RuntimeError('Fp8Weight: needs a contiguous bf16 [N, K] matrix with K % 16 == 0')
        self.shape = w.shape
        self.q = torch.empty(w.shape, device=w.device, dtype=torch.uint8)
        self.scale = torch.empty(w.shape[0], device=w.device, dtype=F32)
        self.requantize(w)

    def requantize(self, w_bf16):
        lib.sidlsg_quantize_fp8_rows(_p(w_bf16), _p(self.q), _p(self.scale), self.shape[0], self.shape[1], _s())

    def dequantize(self):
        return self.q.view(torch.float8_e4m3fn).float() * self.scale[:, None]


# ---- grouped launches: two networks of identical architecture on one stacked batch ----------------------------------------
class Pair(tuple):
    """(set 0, set 1): the same parameter (forward weight copy, backward-data operand, bias, norm scale / shift) of the two
    networks of a grouped pass.  The first half of the stacked batch is evaluated with set 0, the second half with set 1
    (include/sidlsg_hip.h "grouped launches")."""
    __slots__ = ()

    def __new__(cls, a, b):
        return super().__new__(cls, (a, b))

    @property
    def shape(self):
        return self[0].shape


_dual = None     # while a grouped pass is being recorded: id(tensor of network 0) -> the same tensor of network 1


class dual_networks:
    """`with ops.dual_networks(partner_map): net0.forward_nhwc(stacked inputs)` -- every weight-bearing op issued inside looks
    its parameters' partners up in `partner_map` (HipUNet2DCondition.partner_map(other)) and launches the grouped (`_g2`) entry
    point with both sets; parameter-free ops just see the stacked batch.  The autograd nodes keep the pairs, so the backward
    (data gradients only: both networks are frozen) needs no context."""

    def __init__(self, partner_map):
        self.map = partner_map

    def __enter__(self):
        global _dual
        if _dual is not None:
            raise RuntimeError('grouped passes do not nest')
        _dual = self.map
        return self

    def __exit__(self, *exc):
        global _dual
        _dual = None
        return False


def _pair(t):
    """(t, partner of t) under dual_networks; None stays None."""
    if t is None:
        return None
    try:
        return Pair(t, _dual[id(t)])
    except KeyError:
        raise RuntimeError('grouped pass: a parameter of the first network has no partner in the second one (different '
                           'architectures, fp8 weights, or compute copies re-created after the partner map was built)') from None


# raw launches
def gemm(a, w16, out=None, bias=None, res=None, rowvec=None, rows_per_batch=1, alpha=1.0, out_f32=False, lda=None):
    """C[M,N] = alpha*A[M,K] W[N,K]^T + bias + rowvec[m//rpb] + res.  w16 / bias may be Pairs (grouped launch: rows of the
    first half of A with set 0, of the second half with set 1)."""
    M = a.shape[0]
    K = w16.shape[1]
    N = w16.shape[0]
    lda = a.stride(0) if lda is None else lda
    f32 = a.dtype == F32
    if isinstance(w16, Pair):
        if a.dtype != BF16 or w16[0].dtype != BF16 or w16[1].dtype != BF16 or w16[0].shape != w16[1].shape:
            raise RuntimeError('grouped GEMM: bf16 activations and two bf16 weight matrices of one shape')
        ensure_workspace(a.device)
        if out is None:
            out = torch.empty((M, N), device=a.device, dtype=F32 if out_f32 else BF16)
        lib.sidlsg_gemm_bf16_g2(_p(a), lda, _p(w16[0]), _p(w16[1]), _p(out), out.stride(0), _p(bias[0]) if bias is not None else None,
                                _p(bias[1]) if bias is not None else None, _p(res), res.stride(0) if res is not None else 0, _p(rowvec),
                                rowvec.stride(0) if rowvec is not None else 0, rows_per_batch, M, N, K, float(alpha), 1 if out_f32 else 0, _s())
        return out
    if f32 and w16.dtype != F32:
        raise RuntimeError('fp32 activations need the fp32 compute copy of the weights')
    ensure_workspace(a.device)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=F32 if (out_f32 or f32) else BF16)
    if isinstance(w16, Fp8Weight):
        if a.dtype != BF16:
            raise RuntimeError('fp8 weights take bf16 activations')
        lib.sidlsg_gemm_fp8w(_p(a), lda, _p(w16.q), _p(w16.scale), _p(out), out.stride(0), _p(bias), _p(res),
                             res.stride(0) if res is not None else 0, _p(rowvec), rowvec.stride(0) if rowvec is not None else 0,
                             rows_per_batch, M, N, K, float(alpha), 1 if out_f32 else 0, _s())
        return out
    _fn('gemm', a.dtype, '_bf16')(_p(a), lda, _p(w16), _p(out), out.stride(0), _p(bias), _p(res), res.stride(0) if res is not None else 0,
                                  _p(rowvec), rowvec.stride(0) if rowvec is not None else 0, rows_per_batch, M, N, K, float(alpha),
                                  0 if f32 else (1 if out_f32 else 0), _s())
    return out


def conv3x3(x, w16, bias=None, res=None, rowvec=None, stride=1, ups=0, out_f32=False):
    """x: [B,Hs,Ws,Cin] bf16 NHWC; w16: [Cout, 9*Cin]; -> [B,Ho,Wo,Cout]"""
    B, Hs, Ws, Cin = x.shape
    H, W = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    Cout = w16.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    f32 = x.dtype == F32
    if isinstance(w16, Pair):
        if x.dtype != BF16 or w16[0].dtype != BF16 or w16[1].dtype != BF16 or w16[0].shape != w16[1].shape:
            raise RuntimeError('grouped conv: bf16 activations and two bf16 weight matrices of one shape')
        ensure_workspace(x.device)
        out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=F32 if out_f32 else BF16)
        lib.sidlsg_conv3x3_bf16_g2(_p(x), Cin, _p(w16[0]), _p(w16[1]), _p(out), Cout, _p(bias[0]) if bias is not None else None,
                                   _p(bias[1]) if bias is not None else None, _p(res), Cout if res is not None else 0, _p(rowvec),
                                   rowvec.stride(0) if rowvec is not None else 0, B, H, W, Cin, Cout, stride, ups, 1.0, 1 if out_f32 else 0, _s())
        return out
    if f32 and w16.dtype != F32:
        raise RuntimeError('fp32 activations need the fp32 compute copy of the weights')
    ensure_workspace(x.device)
    out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=F32 if (out_f32 or f32) else BF16)
    if isinstance(w16, Fp8Weight):
        if x.dtype != BF16:
            raise RuntimeError('fp8 weights take bf16 activations')
        lib.sidlsg_conv3x3_fp8w(_p(x), Cin, _p(w16.q), _p(w16.scale), _p(out), Cout, _p(bias), _p(res), Cout if res is not None else 0,
                                _p(rowvec), rowvec.stride(0) if rowvec is not None else 0, B, H, W, Cin, Cout, stride, ups, 1.0,
                                1 if out_f32 else 0, _s())
        return out
    _fn('conv3x3', x.dtype, '_bf16')(_p(x), Cin, _p(w16), _p(out), Cout, _p(bias), _p(res), Cout if res is not None else 0, _p(rowvec),
                                     rowvec.stride(0) if rowvec is not None else 0, B, H, W, Cin, Cout, stride, ups, 1.0,
                                     0 if f32 else (1 if out_f32 else 0), _s())
    return out


def colsum(g2d, rows_per_batch, per_batch=False, total=None, slot=None):
    """g2d: [B*rows_per_batch, N].  total (fp32 [N]) is accumulated in place; returns per-batch sums if asked.
    slot: the (holder, column offset, width) of a split_columns() part -- the per-batch sums are then accumulated straight into the
    part's columns of the holder's shared [B, sum C] gradient buffer (zeroed once per backward pass) and a view of it is returned."""
    R, N = g2d.shape
    B = R // rows_per_batch
    if per_batch and slot is not None and slot[2] == N:
        holder, off, _ = slot
        buf = holder.buffer(B, g2d.device)
        if buf is not None:
            pb = buf[:, off:off + N]
            _fn('colsum_strided', g2d.dtype)(_p(g2d), g2d.stride(0), pb.data_ptr(), buf.stride(0), _p(total), B, rows_per_batch, N, _s())
            return pb
    pb = torch.zeros((B, N), device=g2d.device, dtype=F32) if per_batch else None
    _fn('colsum', g2d.dtype)(_p(g2d), g2d.stride(0), _p(pb), _p(total), None, B, rows_per_batch, N, _s())
    return pb


# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = x W^T + b (+ res) (+ rowvec broadcast over rows_per_batch rows).  x: [M,K] bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, w16t, res, rowvec, rows_per_batch, out_f32):
        _chk(x, ACT)
        y = gemm(x, w16, bias=bias, res=res, rowvec=rowvec, rows_per_batch=rows_per_batch, out_f32=out_f32)
        ctx.save_for_backward(x, weight, bias, w16t)
        ctx.rpb = rows_per_batch
        ctx.has_res = res is not None
        ctx.has_rv = rowvec is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, w16t = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = gemm(dy, w16t) if ctx.needs_input_grad[0] else None
        need_b = _wants_grad(bias)
        need_rv = ctx.has_rv and ctx.needs_input_grad[6]
        # bias gradient comes out of the bf16 wgrad kernel (the fp32 family computes it with a column sum)
        fused_b = need_b and not need_rv and _wants_grad(weight) and x.dtype == BF16
        if _wants_grad(weight):
            M, K = x.shape
            assign = _take_assign(weight, x.dtype)
            if not _queue_dense_wgrad(dy, x, weight.grad, bias.grad if fused_b else None, M, weight.shape[0], K, assign):
                wg = _fn('wgrad_assign', x.dtype, '_bf16') if assign else _fn('wgrad', x.dtype, '_bf16')
                with _OnWgradStream(dy, x):
                    wg(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(weight.grad), _p(bias.grad) if fused_b else None, M, weight.shape[0], K, _s())
        drv = None
        if need_rv:
            drv = colsum(dy, ctx.rpb, per_batch=True, total=bias.grad if need_b else None)
        elif need_b and not fused_b:
            colsum(dy, dy.shape[0], total=bias.grad)
        dres = dy if (ctx.has_res and ctx.needs_input_grad[5]) else None
        return dx, None, None, None, None, dres, drv, None, None


class _LinearG2(torch.autograd.Function):
    """_Linear for the grouped pass of two FROZEN networks: parameters arrive as Pairs, the backward is the data gradient only
    (nothing of the forward input is kept alive for a weight gradient)."""

    @staticmethod
    def forward(ctx, x, bias, w16, w16t, res, rowvec, rows_per_batch, out_f32):
        _chk(x, BF16)
        y = gemm(x, w16, bias=bias, res=res, rowvec=rowvec, rows_per_batch=rows_per_batch, out_f32=out_f32)
        ctx.w16t, ctx.rpb = w16t, rows_per_batch
        ctx.has_res, ctx.has_rv = res is not None, rowvec is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        dx = gemm(dy, ctx.w16t) if ctx.needs_input_grad[0] else None
        drv = colsum(dy, ctx.rpb, per_batch=True) if (ctx.has_rv and ctx.needs_input_grad[5]) else None
        dres = dy if (ctx.has_res and ctx.needs_input_grad[4]) else None
        return dx, None, None, None, dres, drv, None, None


def _frozen(*params):
    for p in params:
        for q in (p if isinstance(p, Pair) else (p,)):
            if q is not None and q.requires_grad:
                raise RuntimeError('a grouped pass evaluates FROZEN networks: call requires_grad_(False) on both first')


def linear(x, weight, bias, w16, w16t, res=None, rowvec=None, rows_per_batch=1, out_f32=False):
    if _dual is not None:
        wp, bp = _pair(weight), _pair(bias)
        _frozen(wp, bp)
        return _LinearG2.apply(x, Pair(bp[0], bp[1]) if bp is not None else None, _pair(w16), _pair(w16t), res, rowvec, rows_per_batch, out_f32)
    if isinstance(w16, Fp8Weight):
        _mx8_no_weight_grads(weight, None)
    return _Linear.apply(x, weight, bias, w16, w16t, res, rowvec, rows_per_batch, out_f32)


class _Conv3x3(torch.autograd.Function):
    """NHWC 3x3 conv, pad 1, stride 1|2, optional fused nearest-x2 upsample of the input."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, w16t, res, rowvec, stride, ups, out_f32, bias_p):
        _chk(x, ACT)
        y = conv3x3(x, w16, bias=bias_p if bias_p is not None else bias, res=res, rowvec=rowvec, stride=stride, ups=ups,
                    out_f32=out_f32)
        ctx.save_for_backward(x, weight, bias, w16t)
        ctx.cfg = (stride, ups, res is not None, rowvec is not None)
        ctx.rv_slot = getattr(rowvec, '_col_slot', None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, w16t = ctx.saved_tensors
        stride, ups, has_res, has_rv = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        B, Ho, Wo, Cout = dy.shape
        Cin = x.shape[3]
        dx = None
        if ctx.needs_input_grad[0]:
            g = dy
            if stride == 2:
                g = torch.empty((B, x.shape[1], x.shape[2], Cout), device=dy.device, dtype=x.dtype)
                _fn('zero_insert2', x.dtype)(_p(dy), _p(g), B, Ho, Wo, x.shape[1], x.shape[2], Cout, _s())
            dx = conv3x3(g, w16t)           # w16t: [Cin, 9*Cout], taps flipped
            if ups:
                full = dx
                dx = torch.empty_like(x)
                _fn('sumpool2x2', x.dtype)(_p(full), _p(dx), B, x.shape[1], x.shape[2], Cin, _s())
        co_w, ci_w = weight.shape[0], weight.shape[1]     # logical (unpadded) sizes of the master
        padded = (co_w != Cout) or (ci_w != Cin)           # conv_in (Cin 4->8) / conv_out (Cout 4->8)
        need_b = _wants_grad(bias)
        need_rv = has_rv and ctx.needs_input_grad[6]
        # bias gradient from the (bf16) wgrad kernel
        fused_b = need_b and not need_rv and not padded and _wants_grad(weight) and x.dtype == BF16
        if _wants_grad(weight):
            H, W = (2 * x.shape[1], 2 * x.shape[2]) if ups else (x.shape[1], x.shape[2])
            wgrad = _fn('conv3x3_wgrad', x.dtype, '_bf16')
            if not padded and _take_assign(weight, x.dtype):
                wgrad = lib.sidlsg_conv3x3_wgrad_assign_bf16
            if not padded:
                with _OnWgradStream(dy, x):
                    wgrad(_p(dy), Cout, _p(x), Cin, _p(weight.grad), _p(bias.grad) if fused_b else None,
                          B, H, W, Cin, Cout, stride, ups, _s())
            else:
                tmp = torch.zeros((Cout, 9, Cin), device=dy.device, dtype=F32)
                wgrad(_p(dy), Cout, _p(x), Cin, _p(tmp), None, B, H, W, Cin, Cout, stride, ups, _s())
                weight.grad.permute(0, 2, 3, 1).reshape(co_w, 9, ci_w).add_(tmp[:co_w, :, :ci_w])
        dy2 = dy.view(B * Ho * Wo, Cout)
        drv = None
        btot = None
        if need_b and not fused_b:
            btot = bias.grad if co_w == Cout else torch.zeros(Cout, device=dy.device, dtype=F32)
        if need_rv:
            drv = colsum(dy2, Ho * Wo, per_batch=True, total=btot, slot=ctx.rv_slot)
        elif need_b and not fused_b:
            colsum(dy2, dy2.shape[0], total=btot)
        if need_b and not fused_b and co_w != Cout:
            bias.grad.add_(btot[:co_w])
        dres = dy if (has_res and ctx.needs_input_grad[5]) else None
        return dx, None, None, None, None, dres, drv, None, None, None, None


class _Conv3x3G2(torch.autograd.Function):
    """_Conv3x3 for the grouped pass of two frozen networks (Pairs; data gradient only)."""

    @staticmethod
    def forward(ctx, x, bias, w16, w16t, res, rowvec, stride, ups, out_f32):
        _chk(x, BF16)
        y = conv3x3(x, w16, bias=bias, res=res, rowvec=rowvec, stride=stride, ups=ups, out_f32=out_f32)
        ctx.w16t = w16t
        ctx.cfg = (stride, ups, res is not None, rowvec is not None, x.shape)
        ctx.rv_slot = getattr(rowvec, '_col_slot', None)
        return y

    @staticmethod
    def backward(ctx, dy):
        stride, ups, has_res, has_rv, xs = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        B, Ho, Wo, Cout = dy.shape
        dx = None
        if ctx.needs_input_grad[0]:
            g = dy
            if stride == 2:
                g = torch.empty((B, xs[1], xs[2], Cout), device=dy.device, dtype=BF16)
                lib.sidlsg_zero_insert2(_p(dy), _p(g), B, Ho, Wo, xs[1], xs[2], Cout, _s())
            dx = conv3x3(g, ctx.w16t)
            if ups:
                full = dx
                dx = torch.empty(xs, device=dy.device, dtype=BF16)
                lib.sidlsg_sumpool2x2(_p(full), _p(dx), B, xs[1], xs[2], xs[3], _s())
        drv = colsum(dy.view(B * Ho * Wo, Cout), Ho * Wo, per_batch=True, slot=ctx.rv_slot) if (has_rv and ctx.needs_input_grad[5]) else None
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        return dx, None, None, None, dres, drv, None, None, None


def conv3x3_op(x, weight, bias, w16, w16t, res=None, rowvec=None, stride=1, ups=0, out_f32=False, bias_p=None):
    if _dual is not None:
        wp, bp = _pair(weight), _pair(bias)
        _frozen(wp, bp)
        bq = _pair(bias_p) if bias_p is not None else bp        # (conv_out: the zero-padded bias of the padded output channels)
        return _Conv3x3G2.apply(x, Pair(bq[0], bq[1]) if bq is not None else None, _pair(w16), _pair(w16t), res, rowvec, stride, ups, out_f32)
    if isinstance(w16, Fp8Weight):
        _mx8_no_weight_grads(weight, None)
    return _Conv3x3.apply(x, weight, bias, w16, w16t, res, rowvec, stride, ups, out_f32, bias_p)


class _GroupNorm(torch.autograd.Function):
    """fork=True: returns (y, x_keep) where x_keep aliases x and must be what the block's OTHER branch (residual /
    shortcut) consumes.  The gradient of that branch then arrives here as dkeep and is summed inside the norm-backward
    kernel instead of by a separate autograd accumulation kernel (~1.5 % of the step)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, fork):
        _chk(x, ACT)
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        n = lib.sidlsg_groupnorm_ws_floats.raw(B, HW, C, groups)
        if n < 0:
            raise RuntimeError(f'groupnorm: unsupported shape B={B} HW={HW} C={C} G={groups}')
        ws = torch.empty(n, device=x.device, dtype=F32)
        stats = torch.empty((B, groups, 2), device=x.device, dtype=F32)
        y = torch.empty_like(x)
        _fn('groupnorm_fwd', x.dtype)(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(ws), B, HW, C, groups, float(eps), int(silu), _s())
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (B, HW, C, groups, int(silu), n)
        if fork:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dkeep=None):
        x, gamma, beta, stats = ctx.saved_tensors
        B, HW, C, groups, silu, n = ctx.cfg
        if dy is None:                       # only the pass-through output was used
            return dkeep, None, None, None, None, None, None
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dkeep is not None:
            dkeep = dkeep.contiguous()
            if dkeep.dtype != x.dtype:
                dkeep = dkeep.to(x.dtype)
        ws = torch.empty(n, device=x.device, dtype=F32)
        dx = torch.empty_like(x)
        pg = _wants_grad(gamma) and _wants_grad(beta)
        h = _defer_begin(x.device) if pg else None
        _fn('groupnorm_bwd', x.dtype)(_p(x), _p(dy), _p(stats), _p(gamma), _p(beta), _p(dkeep) if dkeep is not None else None, _p(dx),
                                      _p(gamma.grad) if pg else None, _p(beta.grad) if pg else None, _p(ws), B, HW, C, groups, silu, _s())
        _defer_end(h, ws)
        return dx, None, None, None, None, None, None


class _GroupNormG2(torch.autograd.Function):
    """_GroupNorm for the grouped pass: samples of the first half of the batch use (gamma, beta) of set 0, the others set 1."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, silu, fork):
        _chk(x, BF16)
        B, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (B * C)
        n = lib.sidlsg_groupnorm_ws_floats.raw(B, HW, C, groups)
        if n < 0 or B % 2:
            raise RuntimeError(f'grouped groupnorm: unsupported shape B={B} HW={HW} C={C} G={groups}')
        ws = torch.empty(n, device=x.device, dtype=F32)
        stats = torch.empty((B, groups, 2), device=x.device, dtype=F32)
        y = torch.empty_like(x)
        lib.sidlsg_groupnorm_fwd_g2(_p(x), _p(gamma[0]), _p(beta[0]), _p(gamma[1]), _p(beta[1]), _p(y), _p(stats), _p(ws), B, HW, C, groups,
                                    float(eps), int(silu), _s())
        ctx.save_for_backward(x, stats)
        ctx.params = (gamma, beta)
        ctx.cfg = (B, HW, C, groups, int(silu), n)
        if fork:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dkeep=None):
        x, stats = ctx.saved_tensors
        gamma, beta = ctx.params
        B, HW, C, groups, silu, n = ctx.cfg
        if dy is None:
            return dkeep, None, None, None, None, None, None
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        if dkeep is not None:
            dkeep = dkeep.contiguous()
            if dkeep.dtype != BF16:
                dkeep = dkeep.to(BF16)
        ws = torch.empty(n, device=x.device, dtype=F32)
        dx = torch.empty_like(x)
        lib.sidlsg_groupnorm_bwd_g2(_p(x), _p(dy), _p(stats), _p(gamma[0]), _p(beta[0]), _p(gamma[1]), _p(beta[1]),
                                    _p(dkeep) if dkeep is not None else None, _p(dx), _p(ws), B, HW, C, groups, silu, _s())
        return dx, None, None, None, None, None, None


def group_norm(x, gamma, beta, groups, eps, silu, fork=False):
    if _dual is not None:
        gp, bp = _pair(gamma), _pair(beta)
        _frozen(gp, bp)
        return _GroupNormG2.apply(x, gp, bp, groups, eps, silu, fork)
    return _GroupNorm.apply(x, gamma, beta, groups, eps, silu, fork)


class _LayerNorm(torch.autograd.Function):
    """fork=True: see _GroupNorm."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork):
        _chk(x, ACT)
        C = x.shape[-1]
        rows = x.numel() // C
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), device=x.device, dtype=F32)
        _fn('layernorm_fwd', x.dtype)(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), rows, C, float(eps), _s())
        ctx.save_for_backward(x, gamma, beta, stats)
        if fork:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dkeep=None):
        x, gamma, beta, stats = ctx.saved_tensors
        if dy is None:
            return dkeep, None, None, None, None
        C = x.shape[-1]
        rows = x.numel() // C
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dkeep is not None:
            dkeep = dkeep.contiguous()
            if dkeep.dtype != x.dtype:
                dkeep = dkeep.to(x.dtype)
        dx = torch.empty_like(x)
        pg = _wants_grad(gamma) and _wants_grad(beta)
        ws = torch.empty(lib.sidlsg_layernorm_bwd_nblocks.raw(rows) * C * 2, device=x.device, dtype=F32) if pg else None
        h = _defer_begin(x.device) if pg else None
        _fn('layernorm_bwd', x.dtype)(_p(x), _p(dy), _p(stats), _p(gamma), _p(dkeep) if dkeep is not None else None, _p(dx),
                                      _p(gamma.grad) if pg else None, _p(beta.grad) if pg else None, _p(ws), rows, C, _s())
        _defer_end(h, ws)
        return dx, None, None, None, None


class _LayerNormG2(torch.autograd.Function):
    """_LayerNorm for the grouped pass: token rows of the first half use set 0, of the second half set 1.  Halves whose row count
    the kernel cannot align its per-wave row ranges with (tiny test networks) run as two ordinary launches on the half views."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork):
        _chk(x, BF16)
        C = x.shape[-1]
        rows = x.numel() // C
        if rows % 2:
            raise RuntimeError('grouped layernorm: odd row count')
        half = rows // 2
        y = torch.empty_like(x)
        stats = torch.empty((rows, 2), device=x.device, dtype=F32)
        grouped = half % 16 == 0
        if grouped:
            lib.sidlsg_layernorm_fwd_g2(_p(x), _p(gamma[0]), _p(beta[0]), _p(gamma[1]), _p(beta[1]), _p(y), _p(stats), rows, C, float(eps), _s())
        else:
            es = x.element_size()
            for h in (0, 1):
                lib.sidlsg_layernorm_fwd(x.data_ptr() + h * half * C * es, _p(gamma[h]), _p(beta[h]), y.data_ptr() + h * half * C * es,
                                         stats.data_ptr() + h * half * 8, half, C, float(eps), _s())
        ctx.save_for_backward(x, stats)
        ctx.params, ctx.grouped = gamma, grouped
        if fork:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dkeep=None):
        x, stats = ctx.saved_tensors
        gamma = ctx.params
        if dy is None:
            return dkeep, None, None, None, None
        C = x.shape[-1]
        rows = x.numel() // C
        half = rows // 2
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        if dkeep is not None:
            dkeep = dkeep.contiguous()
            if dkeep.dtype != BF16:
                dkeep = dkeep.to(BF16)
        dx = torch.empty_like(x)
        if ctx.grouped:
            lib.sidlsg_layernorm_bwd_g2(_p(x), _p(dy), _p(stats), _p(gamma[0]), _p(gamma[1]), _p(dkeep) if dkeep is not None else None, _p(dx),
                                        rows, C, _s())
        else:
            es = x.element_size()
            for h in (0, 1):
                o = h * half * C * es
                lib.sidlsg_layernorm_bwd(x.data_ptr() + o, dy.data_ptr() + o, stats.data_ptr() + h * half * 8, _p(gamma[h]),
                                         dkeep.data_ptr() + o if dkeep is not None else None, dx.data_ptr() + o, None, None, None, half, C, _s())
        return dx, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, fork=False):
    if _dual is not None:
        gp, bp = _pair(gamma), _pair(beta)
        _frozen(gp, bp)
        return _LayerNormG2.apply(x, gp, bp, eps, fork)
    return _LayerNorm.apply(x, gamma, beta, eps, fork)


def gemm_mx8(a8, w8, out=None, bias=None, res=None, out_f32=False):
    """C[M,N] = wscale[n] * A8[M,K] W8[N,K]^T + bias + res with BOTH operands e4m3 (sidlsg_gemm_mx8: MX MFMA).
    a8: uint8 [M,K] e4m3 bytes at unit scale; w8: Fp8Weight with N % 160 == 0."""
    if not isinstance(w8, Fp8Weight) or a8.dtype != torch.uint8:
        raise RuntimeError('gemm_mx8: e4m3 activations (uint8) and an Fp8Weight')
    M, K = a8.shape
    N = w8.shape[0]
    ensure_workspace(a8.device)
    if out is None:
        out = torch.empty((M, N), device=a8.device, dtype=F32 if out_f32 else BF16)
    lib.sidlsg_gemm_mx8(_p(a8), a8.stride(0), _p(w8.q), _p(w8.scale), _p(out), out.stride(0), _p(bias), _p(res),
                        res.stride(0) if res is not None else 0, None, 0, 1, M, N, K, 1.0, 1 if out_f32 else 0, _s())
    return out


def conv3x3_mx8(x8, w8, bias=None, res=None, rowvec=None, out_f32=False):
    """3x3 conv, pad 1, stride 1, on an e4m3 NHWC image x8 [B,H,W,Cin] (uint8) with an Fp8Weight [Cout, 9*Cin] (Cout % 160 == 0)."""
    if not isinstance(w8, Fp8Weight) or x8.dtype != torch.uint8:
        raise RuntimeError('conv3x3_mx8: e4m3 activations (uint8) and an Fp8Weight')
    B, H, W, Cin = x8.shape
    Cout = w8.shape[0]
    ensure_workspace(x8.device)
    out = torch.empty((B, H, W, Cout), device=x8.device, dtype=F32 if out_f32 else BF16)
    lib.sidlsg_conv3x3_mx8(_p(x8), x8.stride(2), _p(w8.q), _p(w8.scale), _p(out), Cout, _p(bias), _p(res), res.stride(2) if res is not None else 0,
                           _p(rowvec), rowvec.stride(0) if rowvec is not None else 0, B, H, W, Cin, Cout, 1.0, 1 if out_f32 else 0, _s())
    return out


def cast_fp8(x):
    """bf16 [.., K] -> e4m3 bytes (uint8, same shape): clamp to +-448, round to nearest even, unit scale."""
    _chk(x, BF16)
    y = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
    lib.sidlsg_cast_fp8(_p(x), _p(y), x.numel(), _s())
    return y


def mx8_ok(w):
    """Can this forward weight take e4m3 activations through sidlsg_gemm_mx8?"""
    return isinstance(w, Fp8Weight) and w.shape[0] % 160 == 0 and w.shape[1] % 16 == 0


class _NormLinearMX8(torch.autograd.Function):
    """FROZEN networks with e4m3 weights: y = Linear(Norm(x)) as ONE autograd node -- the GroupNorm / LayerNorm kernel writes
    its output as e4m3 bytes (half the bytes of the bf16 output it replaces), the contraction runs on the MX-fp8 MFMA
    (sidlsg_gemm_mx8).  One node because the e4m3 intermediate is an integer tensor autograd cannot carry.  Backward (the data
    gradient the generator's update needs through the frozen networks): dnorm_out = dy W through the bf16 backward-data
    operand, then the ordinary norm backward on the saved input.  fork: as in _GroupNorm / _LayerNorm."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, groups, silu, fork, w8, bias, w16t, weight):
        _chk(x, BF16)
        C = x.shape[-1]
        y8 = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        if groups:
            B = x.shape[0]
            HW = x.numel() // (B * C)
            n = lib.sidlsg_groupnorm_ws_floats.raw(B, HW, C, groups)
            if n < 0:
                raise RuntimeError(f'groupnorm: unsupported shape B={B} HW={HW} C={C} G={groups}')
            ws = torch.empty(n, device=x.device, dtype=F32)
            stats = torch.empty((B, groups, 2), device=x.device, dtype=F32)
            lib.sidlsg_groupnorm_fwd_fp8(_p(x), _p(gamma), _p(beta), _p(y8), _p(stats), _p(ws), B, HW, C, groups, float(eps), int(silu), _s())
            ctx.cfg = (B, HW, C, groups, int(silu), n)
        else:
            rows = x.numel() // C
            stats = torch.empty((rows, 2), device=x.device, dtype=F32)
            lib.sidlsg_layernorm_fwd_fp8(_p(x), _p(gamma), _p(beta), _p(y8), _p(stats), rows, C, float(eps), _s())
            ctx.cfg = None
        y = gemm_mx8(y8.view(-1, C), w8, bias=bias)
        ctx.save_for_backward(x, gamma, beta, stats, w16t)
        if fork:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dkeep=None):
        x, gamma, beta, stats, w16t = ctx.saved_tensors
        none = (None,) * 10
        if dy is None:
            return (dkeep,) + none
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        dn = gemm(dy, w16t)                                        # gradient at the norm's output, [M, C]
        if dkeep is not None:
            dkeep = dkeep.contiguous()
            if dkeep.dtype != BF16:
                dkeep = dkeep.to(BF16)
        dx = torch.empty_like(x)
        C = x.shape[-1]
        if ctx.cfg is not None:
            B, HW, C, groups, silu, n = ctx.cfg
            ws = torch.empty(n, device=x.device, dtype=F32)
            lib.sidlsg_groupnorm_bwd(_p(x), _p(dn), _p(stats), _p(gamma), _p(beta), _p(dkeep) if dkeep is not None else None, _p(dx),
                                     None, None, _p(ws), B, HW, C, groups, silu, _s())
        else:
            rows = x.numel() // C
            lib.sidlsg_layernorm_bwd(_p(x), _p(dn), _p(stats), _p(gamma), _p(dkeep) if dkeep is not None else None, _p(dx), None, None, None,
                                     rows, C, _s())
        return (dx,) + none


class _NormConvMX8(torch.autograd.Function):
    """FROZEN networks: y = conv3x3(SiLU(GroupNorm(x))) + bias + rowvec (+ res) as one autograd node with the e4m3 image in
    between (see _NormLinearMX8).  Backward: data gradient of the conv through the bf16 backward-data operand (flipped taps),
    then the GroupNorm + SiLU backward on the saved input; res receives dy."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, groups, fork, w8, bias, w16t, weight, res, rowvec):
        _chk(x, BF16)
        B, H, W, C = x.shape
        HW = H * W
        n = lib.sidlsg_groupnorm_ws_floats.raw(B, HW, C, groups)
        if n < 0:
            raise RuntimeError(f'groupnorm: unsupported shape B={B} HW={HW} C={C} G={groups}')
        ws = torch.empty(n, device=x.device, dtype=F32)
        stats = torch.empty((B, groups, 2), device=x.device, dtype=F32)
        y8 = torch.empty(x.shape, device=x.device, dtype=torch.uint8)
        lib.sidlsg_groupnorm_fwd_fp8(_p(x), _p(gamma), _p(beta), _p(y8), _p(stats), _p(ws), B, HW, C, groups, float(eps), 1, _s())
        y = conv3x3_mx8(y8, w8, bias=bias, res=res, rowvec=rowvec)
        ctx.save_for_backward(x, gamma, beta, stats, w16t)
        ctx.cfg = (B, HW, C, groups, n, res is not None)
        if fork:
            return y, x.view(x.shape)
        return y

    @staticmethod
    def backward(ctx, dy, dkeep=None):
        x, gamma, beta, stats, w16t = ctx.saved_tensors
        B, HW, C, groups, n, has_res = ctx.cfg
        if dy is None:
            return (dkeep,) + (None,) * 11
        dy = dy.contiguous()
        if dy.dtype != BF16:
            dy = dy.to(BF16)
        dn = conv3x3(dy, w16t)                                     # gradient at the conv's input, [B,H,W,C]
        if dkeep is not None:
            dkeep = dkeep.contiguous()
            if dkeep.dtype != BF16:
                dkeep = dkeep.to(BF16)
        dx = torch.empty_like(x)
        ws = torch.empty(n, device=x.device, dtype=F32)
        lib.sidlsg_groupnorm_bwd(_p(x), _p(dn), _p(stats), _p(gamma), _p(beta), _p(dkeep) if dkeep is not None else None, _p(dx),
                                 None, None, _p(ws), B, HW, C, groups, 1, _s())
        dres = dy if (has_res and ctx.needs_input_grad[10]) else None
        return (dx,) + (None,) * 9 + (dres, None)


def _mx8_no_weight_grads(weight, gamma):
    """The e4m3 forward has no weight-gradient backward: fine under no_grad and for frozen parameters, an error otherwise."""
    if torch.is_grad_enabled() and ((weight is not None and weight.requires_grad) or (gamma is not None and gamma.requires_grad)):
        raise RuntimeError('the MX-fp8 path is for passes without weight gradients (frozen network, or torch.no_grad())')


def norm_conv_mx8(x, gamma, beta, eps, groups, w8, bias, w16t, weight, res=None, rowvec=None, fork=False):
    """conv3x3(SiLU(GroupNorm(x))) for a frozen network, e4m3 in between."""
    _mx8_no_weight_grads(weight, gamma)
    return _NormConvMX8.apply(x, gamma, beta, eps, groups, fork, w8, bias, w16t, weight, res, rowvec)


def norm_linear_mx8(x, gamma, beta, eps, w8, bias, w16t, weight, groups=0, silu=False, fork=False):
    """Linear(GroupNorm(x)) (groups > 0) or Linear(LayerNorm(x)) (groups = 0) for a frozen network, e4m3 in between."""
    _mx8_no_weight_grads(weight, gamma)
    return _NormLinearMX8.apply(x, gamma, beta, eps, groups, silu, fork, w8, bias, w16t, weight)


_ATTN_ALWAYS_KV = os.environ.get('SIDLSG_ATTN_SKIP_KV', '1') == '0'      # A/B switch: compute dK / dV even when nobody wants them


class _Attention(torch.autograd.Function):
    """q: [B,Nq,*] view with heads*D channels starting at column qoff of a row of width ldq; same for k, v."""

    @staticmethod
    def forward(ctx, qbuf, kvbuf, heads, D, qoff, koff, voff, prescaled=False):
        _chk(qbuf, ACT)
        _chk(kvbuf, qbuf.dtype)
        # prescaled: the queries arrive multiplied by D^-1/2 log2(e) (folded into the projection's forward weight copy,
        # HipUNet2DCondition._build_prescale_plan) -> the `_ps` entry points; bf16 only
        sfx = '_ps' if (prescaled and qbuf.dtype == BF16) else ''
        if prescaled and not sfx:
            raise RuntimeError('pre-scaled queries exist in the bf16 compute mode only')
        B, Nq, ldq = qbuf.shape
        Nk, ldk = kvbuf.shape[1], kvbuf.shape[2]
        C = heads * D
        o = torch.empty((B, Nq, C), device=qbuf.device, dtype=qbuf.dtype)
        lse = torch.empty((B, heads, Nq), device=qbuf.device, dtype=F32)
        es = qbuf.element_size()
        _fn('attn_fwd' + sfx, qbuf.dtype)(qbuf.data_ptr() + qoff * es, kvbuf.data_ptr() + koff * es, kvbuf.data_ptr() + voff * es, _p(o), _p(lse),
                                    B, heads, Nq, Nk, D, ldq, ldk, ldk, C, Nq * ldq, Nk * ldk, Nk * ldk, Nq * C, _s())
        ctx.save_for_backward(qbuf, kvbuf, o, lse)
        ctx.cfg = (heads, D, qoff, koff, voff, sfx)
        return o

    @staticmethod
    def backward(ctx, do):
        qbuf, kvbuf, o, lse = ctx.saved_tensors
        heads, D, qoff, koff, voff, sfx = ctx.cfg
        B, Nq, ldq = qbuf.shape
        Nk, ldk = kvbuf.shape[1], kvbuf.shape[2]
        C = heads * D
        do = do.contiguous()
        if do.dtype != qbuf.dtype:
            do = do.to(qbuf.dtype)
        same = qbuf.data_ptr() == kvbuf.data_ptr()
        # cross-attention whose keys / values nobody differentiates (frozen k|v projection of the text states): dQ only
        need_kv = same or ctx.needs_input_grad[1] or _ATTN_ALWAYS_KV
        dq = torch.empty_like(qbuf)
        dkv = dq if same else (torch.empty_like(kvbuf) if need_kv else None)
        delta = torch.empty((B, heads, Nq), device=qbuf.device, dtype=F32)
        es = qbuf.element_size()
        _fn('attn_bwd' + sfx, qbuf.dtype)(qbuf.data_ptr() + qoff * es, kvbuf.data_ptr() + koff * es, kvbuf.data_ptr() + voff * es, _p(o), _p(do),
                            _p(lse), dq.data_ptr() + qoff * es, dkv.data_ptr() + koff * es if need_kv else None,
                            dkv.data_ptr() + voff * es if need_kv else None, _p(delta),
                            B, heads, Nq, Nk, D, ldq, ldk, ldk, C, Nq * ldq, Nk * ldk, Nk * ldk, Nq * C, _s())
        return dq, (None if same else dkv), None, None, None, None, None, None


def self_attention(qkv, heads, prescaled=False):
    """qkv: [B,N,3C] (fused projection output) -> [B,N,C]"""
    C = qkv.shape[2] // 3
    return _Attention.apply(qkv, qkv, heads, C // heads, 0, C, 2 * C, prescaled)


def cross_attention(q, kv, heads, prescaled=False):
    """q: [B,N,C]; kv: [B,L,2C] (fused k|v projection of the text states) -> [B,N,C]"""
    C = q.shape[2]
    return _Attention.apply(q, kv, heads, C // heads, 0, 0, C, prescaled)


class _GEGLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        _chk(h, ACT)
        F2 = h.shape[-1]
        M = h.numel() // F2
        y = torch.empty(h.shape[:-1] + (F2 // 2,), device=h.device, dtype=h.dtype)
        _fn('geglu_fwd', h.dtype)(_p(h), _p(y), M, F2 // 2, _s())
        ctx.save_for_backward(h)
        return y

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        F2 = h.shape[-1]
        dh = torch.empty_like(h)
        _fn('geglu_bwd', h.dtype)(_p(h), _p(dy.contiguous().to(h.dtype)), _p(dh), h.numel() // F2, F2 // 2, _s())
        return dh


def geglu(h):
    return _GEGLU.apply(h)


# A/B knob: smallest K at which the fused FF-in + GEGLU kernel is used when h is kept.  640 while the fusion lived on gemm_v3_kernel only (K = 320 with h: A-stationary
# GEMM + stand-alone GEGLU 164 + 87 us against 262 us fused, batch 16); 320 since the fusion runs on the 256 x 320 kernel (225-229 us: tools/geglu_p8_bench.py, round 6)
_GEGLU_FUSE_MIN_K = int(os.environ.get('SIDLSG_GEGLU_FUSE_MIN_K', '320'))


class _LinearGEGLU(torch.autograd.Function):
    """y = GEGLU(x W^T + b) with the projection and the gating in ONE kernel (sidlsg_gemm_geglu_bf16): the separate GEGLU pass over
    h = x W^T + b [M, 2F] -- at the 64x64 stage of SD1.5 a 335 MB read that took longer than the projection itself -- disappears.
    h is still written when a backward will need it (GEGLU's derivative needs both halves); under no_grad it is not even stored.
    Backward = sidlsg_geglu_bwd followed by the ordinary Linear backward (data gradient through w16t, weight + bias gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, w16, w16t, keep_h, with_h=False):
        _chk(x, BF16)
        M, K = x.shape
        N2 = w16.shape[0]
        y = torch.empty((M, N2 // 2), device=x.device, dtype=BF16)
        h = torch.empty((M, N2), device=x.device, dtype=BF16) if keep_h else None
        lib.sidlsg_gemm_geglu_bf16(_p(x), x.stride(0), _p(w16), _p(h), N2, _p(y), N2 // 2, _p(bias), M, N2, K, _s())
        if keep_h:
            ctx.save_for_backward(x, weight, bias, w16t, h)
        if with_h:
            # h as a second output: the consumer (_GegluLinear) returns the gradient with respect to h itself -- its data-gradient
            # GEMM applies the GEGLU derivative in the epilogue -- and no gradient for y
            ctx.set_materialize_grads(False)
            return y, h
        return y

    @staticmethod
    def backward(ctx, dy, dh=None):
        x, weight, bias, w16t, h = ctx.saved_tensors
        F2 = h.shape[-1]
        if dy is not None:
            dh_y = torch.empty_like(h)
            lib.sidlsg_geglu_bwd(_p(h), _p(dy.contiguous().to(BF16)), _p(dh_y), h.shape[0], F2 // 2, _s())
            dh = dh_y if dh is None else add(dh.contiguous().to(BF16), dh_y)
        elif dh is None:
            return None, None, None, None, None, None, None
        else:
            dh = dh.contiguous()
            if dh.dtype != BF16:
                dh = dh.to(BF16)
        dx = gemm(dh, w16t) if ctx.needs_input_grad[0] else None
        if _wants_grad(weight):
            M, K = x.shape
            need_b = _wants_grad(bias)
            assign = _take_assign(weight, BF16)
            if not _queue_dense_wgrad(dh, x, weight.grad, bias.grad if need_b else None, M, weight.shape[0], K, assign):
                wg = lib.sidlsg_wgrad_assign_bf16 if assign else lib.sidlsg_wgrad_bf16
                with _OnWgradStream(dh, x):
                    wg(_p(dh), dh.stride(0), _p(x), x.stride(0), _p(weight.grad), _p(bias.grad) if need_b else None, M, weight.shape[0], K, _s())
        elif _wants_grad(bias):
            colsum(dh, dh.shape[0], total=bias.grad)
        return dx, None, None, None, None, None, None


def linear_geglu(x, weight, bias, w16, w16t, with_h=False):
    """GEGLU(Linear(x)): the fused kernel where it applies (bf16, a shape the direct-to-LDS GEMM takes), else the two ops.
    with_h: return (y, h) -- h = None when no backward will run, y = None when the caller has to apply the GEGLU itself
    (feed_forward_out does, inside the node whose backward fuses the GEGLU derivative into the FF-out data gradient)."""
    if (_dual is None and x.dtype == BF16 and isinstance(w16, torch.Tensor) and w16.dtype == BF16 and x.is_contiguous()
            and lib.sidlsg_gemm_geglu_ok.raw(x.shape[0], w16.shape[0], w16.shape[1])):
        keep_h = torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad))
        # measured (tools/ab/geglu_fused.py, MI355X, batch 16): K = 640 / 1280 (32x32 / 16x16 stages) 177 -> 165 us / 134 -> 120 us with h
        # kept; K = 320 (64x64 stage) only pays when h is NOT kept (251 -> 190 us): with h the A-stationary GEMM + the stand-alone
        # GEGLU kernel (164 + 87 us) beat the fused direct-to-LDS kernel (262 us); inside the step the two are equal (SIDLSG_GEGLU_FUSE_MIN_K=320
        # vs 640, eight alternations: 208.5 vs 208.7 ms, tools/_run50.sh)
        if not keep_h or w16.shape[1] >= _GEGLU_FUSE_MIN_K:
            if with_h and keep_h:
                return _LinearGEGLU.apply(x, weight, bias, w16, w16t, keep_h, True)
            y = _LinearGEGLU.apply(x, weight, bias, w16, w16t, keep_h)
            return (y, None) if with_h else y
    h = linear(x, weight, bias, w16, w16t)
    return (None, h) if with_h else geglu(h)


class _GegluLinear(torch.autograd.Function):
    """out = GEGLU(h) W^T + b (+ res): the GEGLU and the FF-out projection of a transformer block as ONE autograd node, so that the
    backward is sidlsg_gemm_geglu_bwd_bf16 -- the projection's data gradient with the GEGLU derivative in its epilogue: dy [M, F]
    is neither written nor re-read (2 x 168 MB at the 64x64 stage of SD1.5, batch 16) and sidlsg_geglu_bwd is not launched.
    y: GEGLU(h) when the fused FF-in kernel has already produced it (then no gradient flows back through y), else None."""

    @staticmethod
    def forward(ctx, h, y, weight, bias, w16, w16t, res):
        _chk(h, BF16)
        M, F2 = h.shape
        if y is None:
            y = torch.empty((M, F2 // 2), device=h.device, dtype=BF16)
            lib.sidlsg_geglu_fwd(_p(h), _p(y), M, F2 // 2, _s())
        out = gemm(y, w16, bias=bias, res=res)
        wg = weight.requires_grad
        ctx.save_for_backward(h, y if wg else None, weight, bias, w16t)
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        h, y, weight, bias, w16t = ctx.saved_tensors
        dout = dout.contiguous()
        if dout.dtype != BF16:
            dout = dout.to(BF16)
        M, F2 = h.shape
        K = dout.shape[1]
        dh = None
        if ctx.needs_input_grad[0]:
            dh = torch.empty_like(h)
            lib.sidlsg_gemm_geglu_bwd_bf16(_p(dout), dout.stride(0), _p(w16t), _p(h), _p(dh), F2, M, F2 // 2, K, _s())
        need_b = _wants_grad(bias)
        if _wants_grad(weight):
            assign = _take_assign(weight, BF16)
            if not _queue_dense_wgrad(dout, y, weight.grad, bias.grad if need_b else None, M, weight.shape[0], F2 // 2, assign):
                wgk = lib.sidlsg_wgrad_assign_bf16 if assign else lib.sidlsg_wgrad_bf16
                with _OnWgradStream(dout, y):
                    wgk(_p(dout), dout.stride(0), _p(y), y.stride(0), _p(weight.grad), _p(bias.grad) if need_b else None, M, weight.shape[0],
                        F2 // 2, _s())
        elif need_b:
            colsum(dout, dout.shape[0], total=bias.grad)
        dres = dout if (ctx.has_res and ctx.needs_input_grad[6]) else None
        return dh, None, None, None, None, None, dres


_FF_G2 = os.environ.get('SIDLSG_FF_G2', '1') != '0'      # A/B: the grouped pass's FeedForward as one node with the fused GEGLU kernels


class _FeedForwardG2(torch.autograd.Function):
    """The FeedForward block of the GROUPED frozen pass (two networks, stacked batch) as ONE autograd node over the grouped forms of the two
    GEGLU fusions: forward = sidlsg_gemm_geglu_bf16_g2 (FF-in projection + gating in one kernel; h is written only when a backward will need
    it) where the fused kernel pays -- K >= 640 with h kept, every admissible shape without -- else grouped GEMM + sidlsg_geglu_fwd, then the
    grouped FF-out GEMM (+ bias + residual); backward (data gradient only: both networks are frozen) = sidlsg_gemm_geglu_bwd_bf16_g2 (FF-out
    data gradient with the GEGLU derivative in its epilogue: dy [M, F] is neither written nor re-read) and the grouped FF-in data gradient.
    Until round 6 this pass -- 64 of an iteration's 144 sample-passes -- ran the unfused chain (grouped GEMMs + stand-alone GEGLU kernels),
    because the fusions only existed for single weight sets."""

    @staticmethod
    def forward(ctx, x, b1, w1_16, w1_16t, b2, w2_16, w2_16t, res, keep_h):
        _chk(x, BF16)
        M, K = x.shape
        N2 = w1_16[0].shape[0]
        F = N2 // 2
        ensure_workspace(x.device)
        y = torch.empty((M, F), device=x.device, dtype=BF16)
        h = None
        if x.is_contiguous() and lib.sidlsg_gemm_geglu_ok.raw(M, N2, K) and (not keep_h or K >= _GEGLU_FUSE_MIN_K):
            h = torch.empty((M, N2), device=x.device, dtype=BF16) if keep_h else None
            lib.sidlsg_gemm_geglu_bf16_g2(_p(x), x.stride(0), _p(w1_16[0]), _p(w1_16[1]), _p(h), N2, _p(y), F, _p(b1[0]), _p(b1[1]), M, N2, K, _s())
        else:
            h = gemm(x, w1_16, bias=b1)
            lib.sidlsg_geglu_fwd(_p(h), _p(y), M, F, _s())
            if not keep_h:
                h = None
        out = gemm(y, w2_16, bias=b2, res=res)
        ctx.save_for_backward(h)
        ctx.ops = (w1_16t, w2_16t)
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        w1_16t, w2_16t = ctx.ops
        dout = dout.contiguous()
        if dout.dtype != BF16:
            dout = dout.to(BF16)
        dx = None
        if ctx.needs_input_grad[0]:
            M, F2 = h.shape
            Kd = dout.shape[1]
            dh = torch.empty_like(h)
            if lib.sidlsg_gemm_geglu_bwd_ok.raw(M, F2 // 2, Kd):
                lib.sidlsg_gemm_geglu_bwd_bf16_g2(_p(dout), dout.stride(0), _p(w2_16t[0]), _p(w2_16t[1]), _p(h), _p(dh), F2, M, F2 // 2, Kd, _s())
            else:
                dy = gemm(dout, w2_16t)
                lib.sidlsg_geglu_bwd(_p(h), _p(dy), _p(dh), M, F2 // 2, _s())
            dx = gemm(dh, w1_16t)
        dres = dout if (ctx.has_res and ctx.needs_input_grad[7]) else None
        return dx, None, None, None, None, None, None, dres, None


def feed_forward(x, w1, b1, w1_16, w1_16t, w2, b2, w2_16, w2_16t, res=None):
    """diffusers FeedForward (GEGLU projection -> Linear) + the block's residual: out = GEGLU(x W1^T + b1) W2^T + b2 + res."""
    if (_dual is not None and _FF_G2 and x.dtype == BF16 and isinstance(w1_16, torch.Tensor) and w1_16.dtype == BF16
            and isinstance(w2_16, torch.Tensor) and w2_16.dtype == BF16 and b1 is not None and b2 is not None and w1_16.shape[0] % 2 == 0
            and x.shape[0] % 2 == 0):
        _frozen(_pair(w1), _pair(b1), _pair(w2), _pair(b2))
        keep_h = torch.is_grad_enabled() and x.requires_grad      # (grad mode is off inside Function.forward: decided here)
        return _FeedForwardG2.apply(x, _pair(b1), _pair(w1_16), _pair(w1_16t), _pair(b2), _pair(w2_16), _pair(w2_16t), res, keep_h)
    fused_bwd = (_dual is None and x.dtype == BF16 and torch.is_grad_enabled() and isinstance(w2_16, torch.Tensor) and w2_16.dtype == BF16
                 and isinstance(w1_16, torch.Tensor)
                 and lib.sidlsg_gemm_geglu_bwd_ok.raw(x.shape[0], w2_16.shape[1], w2_16.shape[0]))
    if not fused_bwd:
        return linear(linear_geglu(x, w1, b1, w1_16, w1_16t), w2, b2, w2_16, w2_16t, res)
    y, h = linear_geglu(x, w1, b1, w1_16, w1_16t, with_h=True)
    return geglu_linear(h, y, w2, b2, w2_16, w2_16t, res)


def geglu_linear(h, y, weight, bias, w16, w16t, res=None):
    """GEGLU(h) -> Linear (+ res) with the fused backward where it applies; h = None: y is all there is (no backward will run)."""
    if h is None:
        return linear(y, weight, bias, w16, w16t, res)
    if (_dual is None and h.dtype == BF16 and h.requires_grad and torch.is_grad_enabled() and isinstance(w16, torch.Tensor) and w16.dtype == BF16
            and h.is_contiguous() and lib.sidlsg_gemm_geglu_bwd_ok.raw(h.shape[0], h.shape[1] // 2, w16.shape[0])):
        return _GegluLinear.apply(h, y, weight, bias, w16, w16t, res)
    return linear(y if y is not None else geglu(h), weight, bias, w16, w16t, res)


class _SiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x, ACT)
        y = torch.empty_like(x)
        _fn('silu_fwd', x.dtype)(_p(x), _p(y), x.numel(), _s())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _fn('silu_bwd', x.dtype)(_p(x), _p(dy.contiguous().to(x.dtype)), _p(dx), x.numel(), _s())
        return dx


def silu(x):
    return _SiLU.apply(x)


class _Concat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _chk(a, ACT)
        _chk(b, a.dtype)
        C1, C2 = a.shape[-1], b.shape[-1]
        M = a.numel() // C1
        out = torch.empty(a.shape[:-1] + (C1 + C2,), device=a.device, dtype=a.dtype)
        _fn('concat2', a.dtype)(_p(a), _p(b), _p(out), M, C1, C2, 0, _s())
        ctx.shapes = (a.shape, b.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        sa, sb = ctx.shapes
        g = g.contiguous()
        da = torch.empty(sa, device=g.device, dtype=g.dtype)
        db = torch.empty(sb, device=g.device, dtype=g.dtype)
        _fn('concat2', g.dtype)(_p(da), _p(db), _p(g), da.numel() // sa[-1], sa[-1], sb[-1], 1, _s())
        return da, db


def concat_channels(a, b):
    return _Concat.apply(a, b)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        o = torch.empty_like(a)
        _fn('add', a.dtype, '_bf16')(_p(_chk(a, ACT)), _p(_chk(b, a.dtype)), _p(o), a.numel(), _s())
        return o

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(a, b)


def timestep_embed(t, dim, dtype=BF16):
    out = torch.empty((t.shape[0], dim), device=t.device, dtype=dtype)
    _fn('timestep_embed', dtype)(_p(_chk(t, torch.int64)), _p(out), t.shape[0], dim, _s())
    return out


# ------------------------------------------------------------------------------------------------
# scheduler / guidance glue and losses
class _NoisyInput(torch.autograd.Function):
    """x_t = s0*x0 + s1*noise -> (NHWC activations [dup*B,H,W,8] of `dtype`, x_t fp32 NCHW).  x0 may be None."""

    @staticmethod
    def forward(ctx, x0, noise, s0, s1, dup, dtype):
        B, C, H, W = noise.shape
        out = torch.empty((dup * B, H, W, 8), device=noise.device, dtype=dtype)
        xt = torch.empty_like(noise)
        _fn('noisy_input', dtype)(_p(x0), _p(_chk(noise, F32)), _p(s0), _p(s1), _p(out), _p(xt), B, C, H * W, 8, dup, _s())
        ctx.save_for_backward(s0, s1)
        ctx.cfg = (B, C, H, W, dup)
        return out, xt

    @staticmethod
    def backward(ctx, g, gxt):
        s0, s1 = ctx.saved_tensors
        B, C, H, W, dup = ctx.cfg
        g = g.contiguous()
        outs = []
        for idx, sc in ((0, s0), (1, s1)):
            if not ctx.needs_input_grad[idx]:
                outs.append(None)
                continue
            d = torch.empty((B, C, H, W), device=g.device, dtype=F32)
            _fn('noisy_input_bwd', g.dtype)(_p(g), _p(sc), _p(d), B, C, H * W, 8, dup, 0, _s())
            if gxt is not None:
                d = d + gxt * sc.view(B, 1, 1, 1)
            outs.append(d)
        return outs[0], outs[1], None, None, None, None


def noisy_input(x0, noise, s0, s1, dup, dtype=BF16):
    return _NoisyInput.apply(x0, noise, s0, s1, dup, dtype)


class _CfgX0(torch.autograd.Function):
    """eps [dup*B,HW,C] fp32 (+ x_t) -> NCHW fp32 guided eps or x0 prediction."""

    @staticmethod
    def forward(ctx, eps, xt, s0, s1, kappa, predict_x0, act_dtype):
        B, C, H, W = xt.shape
        dup = eps.shape[0] // B
        out = torch.empty_like(xt)
        lib.sidlsg_cfg_x0(_p(_chk(eps, F32)), _p(_chk(xt, F32)), _p(s0), _p(s1), _p(out), B, C, H * W, eps.shape[-1], dup,
                          float(kappa), int(predict_x0), _s())
        ctx.save_for_backward(s0, s1)
        ctx.cfg = (B, C, H, W, dup, float(kappa), int(predict_x0), act_dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        s0, s1 = ctx.saved_tensors
        B, C, H, W, dup, kappa, px0, act_dtype = ctx.cfg
        deps = torch.empty((dup * B, H * W, 8), device=g.device, dtype=act_dtype)    # the network's activation dtype
        dxt = torch.empty((B, C, H, W), device=g.device, dtype=F32) if ctx.needs_input_grad[1] else None
        _fn('cfg_x0_bwd', act_dtype)(_p(g.contiguous()), _p(s0), _p(s1), _p(deps), _p(dxt), B, C, H * W, 8, dup, kappa, px0, _s())
        return deps, dxt, None, None, None, None, None


def cfg_x0(eps, xt, s0, s1, kappa, predict_x0, act_dtype=BF16):
    return _CfgX0.apply(eps, xt, s0, s1, kappa, predict_x0, act_dtype)


class _GLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, yr, yf, alpha, scale):
        S = x.shape[0]
        n = x.numel() // S
        dx, dyr, dyf = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        loss = torch.empty(1, device=x.device, dtype=F32)
        ws = torch.empty(10 * S, device=x.device, dtype=F32)
        lib.sidlsg_g_loss(_p(_chk(x, F32)), _p(_chk(yr, F32)), _p(_chk(yf, F32)), _p(dx), _p(dyr), _p(dyf), _p(loss), _p(ws), S, n,
                          float(alpha), float(scale), _s())
        ctx.save_for_backward(dx, dyr, dyf)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dx, dyr, dyf = ctx.saved_tensors
        return dx * g, dyr * g, dyf * g, None, None


def sid_generator_loss(x, y_real, y_fake, alpha, scale):
    return _GLoss.apply(x, y_real, y_fake, alpha, scale)


class _FakeLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, noise, scale):
        S = e.shape[0]
        n = e.numel() // S
        de = torch.empty_like(e)
        loss = torch.empty(1, device=e.device, dtype=F32)
        ws = torch.empty(10 * S, device=e.device, dtype=F32)
        lib.sidlsg_fake_loss(_p(_chk(e, F32)), _p(_chk(noise, F32)), _p(de), _p(loss), _p(ws), S, n, float(scale), _s())
        ctx.save_for_backward(de)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (de,) = ctx.saved_tensors
        return de * g, None, None


def sid_fake_score_loss(e, noise, scale):
    return _FakeLoss.apply(e, noise, scale)


# ------------------------------------------------------------------------------------------------
def transpose_w(src_f32, n, k, taps=1, dtype=BF16):
    """fp32 master [N][T][K] -> [K][T reversed][N] of the compute dtype (the dgrad operand)"""
    dst = torch.empty((k, taps * n), device=src_f32.device, dtype=dtype)
    _fn('transpose_w', dtype)(_p(src_f32), _p(dst), n, k, taps, _s())
    return dst


class _ColumnGrads:
    """The gradient of a split_columns() source, built in place: one [n, sum C] fp32 buffer per backward pass (zeroed ONCE, on
    first use) into whose column slices the consumers' backward kernels accumulate (ops.colsum(slot=...)); _SplitColumns.backward
    hands it on as it is.  22 fills + a concat kernel per pass become one fill."""

    live = weakref.WeakSet()         # every holder, so that a pass that raised can be cleaned up after (_discard_stale_backward_state)

    def __init__(self, sizes):
        self.total, self.buf = sum(sizes), None
        _ColumnGrads.live.add(self)

    def buffer(self, n, device):
        if self.buf is None:
            self.buf = torch.zeros((n, self.total), device=device, dtype=F32)
        return self.buf if self.buf.shape[0] == n else None

    def take(self):
        b, self.buf = self.buf, None
        return b


class _SplitColumns(torch.autograd.Function):
    """x [N, sum C] -> views x[:, off_i:off_i+C_i] (no copies); backward = the shared column-gradient buffer when every part's
    gradient is its slice of it (the normal case), else one torch.cat of the column gradients."""

    @staticmethod
    def forward(ctx, x, sizes, holder):
        ctx.sizes, ctx.meta, ctx.holder = sizes, (x.shape[0], x.dtype, x.device), holder
        return tuple(x.split(sizes, dim=1))

    @staticmethod
    def backward(ctx, *grads):
        n, dtype, dev = ctx.meta
        buf = ctx.holder.take()
        if buf is not None and dtype == F32 and buf.shape[0] == n:
            off, ok = 0, True
            for g, c in zip(grads, ctx.sizes):
                ok = ok and g is not None and g.dtype == F32 and g.data_ptr() == buf.data_ptr() + 4 * off and g.stride(0) == buf.stride(0) and g.shape == (n, c)
                off += c
            if ok:
                return buf, None, None
        gs = [g if g is not None else torch.zeros((n, c), device=dev, dtype=dtype) for g, c in zip(grads, ctx.sizes)]
        return torch.cat([g.to(dtype) for g in gs], dim=1), None, None


def split_columns(x, sizes):
    """-> tuple of column views; part._col_slot = (holder, offset, width) lets a consumer's backward accumulate its gradient in place."""
    sizes = list(sizes)
    holder = _ColumnGrads(sizes)
    parts = _SplitColumns.apply(x, sizes, holder)
    off = 0
    for p, c in zip(parts, sizes):
        p._col_slot = (holder, off, c)
        off += c
    return parts


class _GradReady(torch.autograd.Function):
    """Identity whose backward calls `cb()` before passing the gradient on: placed at a block boundary in the forward, it
    fires when the backward has finished everything AFTER that boundary (autograd runs ready nodes with the highest
    sequence number first, and the marker is older than every node of the blocks behind it)."""

    @staticmethod
    def forward(ctx, x, cb):
        ctx.cb = cb
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        flush_deferred()        # queued dgamma / dbeta reductions belong to the segment that is about to be declared final
        ctx.cb()
        return g, None


def grad_ready_marker(x, cb):
    return _GradReady.apply(x, cb) if (cb is not None and x.requires_grad) else x


class _AfterBackward(torch.autograd.Function):
    """Identity on the network output; its backward (the FIRST node of that network's backward) queues `cb` on the
    autograd engine, which runs it once the whole backward pass has finished."""

    @staticmethod
    def forward(ctx, x, cb):
        ctx.cb = cb
        return x.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        torch.autograd.Variable._execution_engine.queue_callback(ctx.cb)
        return g, None


def after_backward(x, cb):
    return _AfterBackward.apply(x, cb) if x.requires_grad else x


def transpose_w_batched(jobs, njobs, nblocks, dtype=BF16, src16=False):
    """jobs: device uint8 tensor holding njobs sidlsg_tw_job records (see include/sidlsg_hip.h); dtype: of the destinations;
    src16: the records' sources are the bf16 compute copies (64x64-tile job table)."""
    if src16:
        lib.sidlsg_transpose_w16_batched(_p(jobs), njobs, nblocks, _s())
    else:
        _fn('transpose_w_batched', dtype)(_p(jobs), njobs, nblocks, _s())


def scale_cast_ranges(jobs, njobs, nblocks):
    """jobs: device uint8 tensor of njobs records (include/sidlsg_hip.h): dst = bf16(scale * src) per range."""
    lib.sidlsg_scale_cast_ranges(_p(jobs), njobs, nblocks, _s())


def cast_bf16(src_f32, out=None):
    if out is None:
        out = torch.empty(src_f32.shape, device=src_f32.device, dtype=BF16)
    lib.sidlsg_cast_f32_bf16(_p(src_f32), _p(out), src_f32.numel(), _s())
    return out
