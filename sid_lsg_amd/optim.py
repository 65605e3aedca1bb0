"""Fused optimizer for a network whose parameters live in one flat buffer.

Selectable by class name through `dnnlib.util.construct_class_by_name(params=..., class_name=
'sid_lsg_amd.optim.FusedAdamEMA', lr=, betas=, eps=)` exactly like the reference selects
`torch.optim.Adam` (sid_train.py:219-226 -> sid_training_loop.py:291-292).  One kernel launch per
step does: nan_to_num(grad) (sid_training_loop.py:458-460,541-543) -> optional clip (:546-547) ->
Adam/AdamW (torch single-tensor semantics) -> EMA lerp (:553-565) -> bf16 compute copy -> zero_grad.
"""
import math

import torch

from . import ops
from ._lib import lib


FLAT_LAYOUT = 2     # parameter order of HipUNet2DCondition's flat buffers (2: GEMM / conv weights first inside each gradient segment)


def _flat_of(tensors):
    tensors = list(tensors)
    st = tensors[0].untyped_storage()
    for t in tensors:
        if t.untyped_storage().data_ptr() != st.data_ptr():
            raise ValueError('FusedAdamEMA needs all parameters in ONE flat buffer (HipUNet2DCondition.flat_params)')
    return torch.empty(0, dtype=tensors[0].dtype, device=tensors[0].device).set_(st)


class FusedAdamEMA:
    def __init__(self, params, lr=1e-6, betas=(0.0, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False, clip_value=None):
        params = [p for p in params]
        if isinstance(params[0], dict):
            params = [p for g in params for p in g['params']]
        flat = _flat_of([p.data for p in params])
        if any(p.grad is None for p in params):
            raise ValueError('parameters must carry pre-allocated .grad views of the flat gradient buffer')
        grad = _flat_of([p.grad for p in params])
        self._setup(flat, grad, lr, betas, eps, weight_decay, decoupled, clip_value)
        self.param_groups = [dict(params=params, lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]

    @classmethod
    def from_flat(cls, flat, grad, lr=1e-6, betas=(0.0, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False, clip_value=None,
                  ema=None, w16=None):
        self = cls.__new__(cls)
        self._setup(flat, grad, lr, betas, eps, weight_decay, decoupled, clip_value)
        self.param_groups = [dict(params=[flat], lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.attach(ema=ema, w16=w16)
        return self

    def _setup(self, flat, grad, lr, betas, eps, weight_decay, decoupled, clip_value):
        if not flat.is_cuda:
            raise RuntimeError('FusedAdamEMA is a HIP kernel: parameters must be on the GPU (no CPU fallback)')
        assert flat.numel() == grad.numel() and flat.dtype == torch.float32
        self.flat, self.grad = flat, grad
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.weight_decay, self.decoupled = float(weight_decay), bool(decoupled)
        self.clip_value = clip_value
        self.step_count = 0
        self.exp_avg_sq = torch.zeros_like(flat)
        self.exp_avg = torch.zeros_like(flat) if self.betas[0] != 0 else None   # beta1 = 0 (the recipe) needs no first moment
        self.hyper = torch.zeros(16, device=flat.device, dtype=torch.float32)
        # the step's scalars travel through a small ring of PINNED host slots with non-blocking copies: a pageable
        # host-to-device copy blocks the host until the stream has drained, i.e. it was a device sync per optimizer step
        self._hyper_host = [torch.zeros(16, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._hyper_evt = [None] * 4
        self._hyper_slot = 0
        self.ema = self.w16 = None
        self.grad_scale = 1.0

    def attach(self, ema=None, w16=None, owner=None):
        """ema: flat fp32 buffer of the EMA network; w16: flat bf16 compute copy of this network; owner: the network that owns
        w16 -- told (`mark_w16_rewritten`) when a step has rewritten the WHOLE copy as plain bf16(master), which is the
        precondition of its next refresh_compute_weights(cast=False)."""
        self.ema, self.w16, self._owner = ema, w16, owner
        self._covered = 0
        # Weight ranges whose gradients need no zeroing (HipUNet2DCondition.assign_plan): the first weight gradient after a step
        # OVERWRITES its view (ops._take_assign), so this kernel skips its 4 B / parameter of zero stores there and the weight-gradient
        # kernels their read of dW.  parts: sorted cover of the flat buffer, (lo, hi, [parameters] | None = small parameters: zeroed)
        self._parts = None
        plan = owner.assign_plan() if owner is not None and hasattr(owner, 'assign_plan') else None
        if plan is not None and self.grad is not None and owner.flat_grads is not None and \
                owner.flat_grads.data_ptr() == self.grad.data_ptr() and owner.flat_grads.numel() == self.grad.numel():
            ranges, ws = plan
            parts, pos = [], 0
            for lo, hi in sorted(ranges):
                if lo > pos:
                    parts.append((pos, lo, None))
                parts.append((lo, hi, [(o, w) for o, w in ws if lo <= o < hi]))
                pos = hi
            if pos < self.flat.numel():
                parts.append((pos, self.flat.numel(), None))
            self._parts = parts
        return self

    def set_owner(self, net):
        """The network whose flat buffers these are (load_state_dict / state_dict need its name -> offset table); attach() sets it too."""
        self._owner = net

    _owner = None
    _covered = 0
    _parts = None

    def _w16_written(self, n):
        if self.w16 is None or self._owner is None:
            return
        self._covered += n
        if self._covered >= self.flat.numel():
            self._covered = 0
            self._owner.mark_w16_rewritten()

    def zero_grad(self, set_to_none=False):
        self.grad.zero_()   # normally unnecessary: step() zeroes the gradient buffer (or leaves weight ranges to be overwritten)

    def set_hyper(self, ema_beta=0.0):
        """Writes the step's scalars to device memory (outside any captured graph)."""
        b1, b2 = self.betas
        t = self.step_count
        h = [self.lr, b1, b2, self.eps, 1 - b1 ** t, math.sqrt(1 - b2 ** t), float(ema_beta), self.weight_decay,
             1.0 if self.decoupled else 0.0, float(self.clip_value) if self.clip_value else 0.0, self.grad_scale]
        k = self._hyper_slot = (self._hyper_slot + 1) % len(self._hyper_host)
        if self._hyper_evt[k] is not None:
            self._hyper_evt[k].synchronize()          # the copy that last read this slot (4 optimizer steps ago) has completed
        host = self._hyper_host[k]
        host[:len(h)] = torch.tensor(h, dtype=torch.float32)
        self.hyper.copy_(host, non_blocking=True)
        evt = self._hyper_evt[k] or torch.cuda.Event()
        evt.record()
        self._hyper_evt[k] = evt

    def launch(self, use_ema=True, zero_grad=True):
        ops.flush_deferred()          # (no-op unless a backward pass left reductions queued outside the autograd engine's callback)
        self._covered = 0
        self._launch_parts(0, self.flat.numel(), use_ema, zero_grad)

    def launch_range(self, lo, hi, use_ema=True, zero_grad=True):
        """The same kernel on elements [lo, hi) of the flat buffers (the update is elementwise: any partition of the buffer
        into ranges gives bit-identical results).  SiDStep updates a segment of a network as soon as the backward has
        finished with it, on its own stream beside the rest of the backward."""
        if hi <= lo:
            return
        ops.flush_deferred()
        if lo % 4:
            raise ValueError('range start must be a multiple of 4 elements (16-byte vector accesses)')
        self._launch_parts(lo, hi, use_ema, zero_grad)

    def _launch_parts(self, lo, hi, use_ema, zero_grad):
        if self._parts is None:
            return self._kernel(lo, hi, use_ema, zero_grad)
        for a, b, ws in self._parts:
            s, e = max(a, lo), min(b, hi)
            if e <= s:
                continue
            if ws is None:                            # small parameters: zeroed as ever
                self._kernel(s, e, use_ema, zero_grad)
                continue
            # A weight whose mark is still set got NO gradient since the last step took... nothing: its view still holds the
            # gradient the previous step consumed (the range is never zeroed), so this step's gradient of it is zero -- whatever
            # the caller does with the range (whole or cut, zeroing or not), the stale values must not be applied again.
            for o, w in ws:
                if o < e and o + w.numel() > s and getattr(w, '_grad_assign', False) and w.grad is not None:
                    w.grad.zero_()
            if not zero_grad or s != a or e != b:     # a range cut by the caller, or gradients the caller wants to keep: plain launch;
                self._kernel(s, e, use_ema, zero_grad)    # marks stay as they are (set: the next gradient assigns; clear: it accumulates)
                continue
            self._kernel(s, e, use_ema, False)
            for _, w in ws:
                w._grad_assign = True

    def _kernel(self, lo, hi, use_ema, zero_grad):
        off = lambda t, sz: None if t is None else t.data_ptr() + sz * lo      # noqa: E731
        lib.sidlsg_adam_ema(off(self.flat, 4), off(self.grad, 4), off(self.exp_avg, 4), off(self.exp_avg_sq, 4),
                            off(self.ema, 4) if use_ema else None, off(self.w16, 2), self.hyper.data_ptr(), hi - lo,
                            1 if zero_grad else 0, ops._s())
        self._w16_written(hi - lo)

    def begin_step(self, ema_beta=None):
        """Host half of a step: advance the step counter and put the step's scalars (bias corrections, EMA beta, 1/world) in
        device memory.  step() does this itself unless `external_scalars` is set -- the captured HIP graph of the training
        step (SiDStep.iteration_graphed) contains only launch(), and its owner calls begin_step() before every replay."""
        self.step_count += 1
        self._covered = 0
        self.set_hyper(0.0 if ema_beta is None else ema_beta)

    external_scalars = False

    @torch.no_grad()
    def step(self, ema_beta=None, zero_grad=True):
        if not self.external_scalars:
            self.begin_step(ema_beta)
        self.launch(use_ema=ema_beta is not None and self.ema is not None, zero_grad=zero_grad)

    def _layout_table(self):
        """name -> (offset, numel) of the owner's flat buffers (None without an owner network)."""
        fn = getattr(self._owner, 'flat_layout_table', None)
        return fn() if fn is not None else None

    def state_dict(self):
        return dict(step=self.step_count, exp_avg_sq=self.exp_avg_sq, exp_avg=self.exp_avg, lr=self.lr, betas=self.betas,
                    eps=self.eps, weight_decay=self.weight_decay, flat_layout=FLAT_LAYOUT, layout=self._layout_table())

    def load_state_dict(self, sd):
        """The moments are stored in the order of the network's flat buffer.  Files carry the name -> (offset, numel) table of the
        buffer they were written from (`layout`; files of rounds <= 4 only a version number, whose table the network can rebuild:
        HipUNet2DCondition.flat_layout_table(version)), and a state written under another parameter order is PERMUTED by name --
        never loaded positionally, which would pair every parameter with some other parameter's second moment."""
        mine = self._layout_table()
        theirs = sd.get('layout')
        if theirs is None and mine is not None and sd.get('flat_layout', 1) != FLAT_LAYOUT:
            theirs = self._owner.flat_layout_table(version=sd.get('flat_layout', 1))
        if theirs is not None and mine is None:
            raise ValueError('optimizer state with a name -> offset table, but this optimizer has no owner network to compare it with: '
                             'attach(owner=net) / set_owner(net) before load_state_dict (the moments are never loaded positionally '
                             'under an unchecked parameter order)')
        if theirs is None or theirs == mine:
            if theirs is None and sd.get('flat_layout', 1) != FLAT_LAYOUT:
                raise ValueError(f"optimizer state of flat-buffer layout {sd.get('flat_layout', 1)} without a name table, this build uses "
                                 f'layout {FLAT_LAYOUT}, and the optimizer has no owner network to rebuild the table from: attach(owner=net) first')
            if sd['exp_avg_sq'].numel() != self.exp_avg_sq.numel():
                raise ValueError('optimizer state of another network (size of the flat buffer differs)')
            self.exp_avg_sq.copy_(sd['exp_avg_sq'])
            if self.exp_avg is not None and sd.get('exp_avg') is not None:
                self.exp_avg.copy_(sd['exp_avg'])
        else:
            if set(theirs) != set(mine) or any(theirs[k][1] != mine[k][1] for k in mine):
                raise ValueError('optimizer state of another architecture (parameter names / sizes differ)')
            for dst, src in ((self.exp_avg_sq, sd['exp_avg_sq']), (self.exp_avg, sd.get('exp_avg'))):
                if dst is None or src is None:
                    continue
                src = src.to(dst.device)
                for k, (o, n) in mine.items():
                    dst[o:o + n].copy_(src[theirs[k][0]:theirs[k][0] + n])
        self.step_count = int(sd['step'])


class FusedAdamWEMA(FusedAdamEMA):
    """`--optimizer adamw` variant (sid_train.py:223-226): decoupled weight decay."""

    def __init__(self, params, lr=1e-6, betas=(0.0, 0.999), eps=1e-8, weight_decay=0.01, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=True, **kw)
