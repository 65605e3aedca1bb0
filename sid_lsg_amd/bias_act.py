"""`bias_act` with the reference's Python API (torch_utils/ops/bias_act.py:58), backed by a gfx950 kernel.

Same argument meaning, same `activation_funcs` table (names, default alpha/gain, plugin index 1..9).
impl='cuda' (default) runs the HIP kernel (sidlsg_bias_act) and RAISES if the tensor is not on the
GPU or the library is missing -- unlike the reference, whose CUDA plugin is disabled outright
(`_init()` returns False, bias_act.py:53-54) and silently uses the slow path.  impl='ref' is the same
explicit plain-PyTorch formulation the reference exposes under that name.
First-order gradients (dx, db) are supported; second-order ones (reference grad=2) are not needed
on the SiD-LSG path and raise.
"""
import math
from types import SimpleNamespace

import torch

from . import ops
from ._lib import lib

activation_funcs = {
    'linear':   SimpleNamespace(func=lambda x, **_: x, def_alpha=0, def_gain=1, cuda_idx=1),
    'relu':     SimpleNamespace(func=lambda x, **_: torch.nn.functional.relu(x), def_alpha=0, def_gain=math.sqrt(2), cuda_idx=2),
    'lrelu':    SimpleNamespace(func=lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), def_alpha=0.2, def_gain=math.sqrt(2), cuda_idx=3),
    'tanh':     SimpleNamespace(func=lambda x, **_: torch.tanh(x), def_alpha=0, def_gain=1, cuda_idx=4),
    'sigmoid':  SimpleNamespace(func=lambda x, **_: torch.sigmoid(x), def_alpha=0, def_gain=1, cuda_idx=5),
    'elu':      SimpleNamespace(func=lambda x, **_: torch.nn.functional.elu(x), def_alpha=0, def_gain=1, cuda_idx=6),
    'selu':     SimpleNamespace(func=lambda x, **_: torch.nn.functional.selu(x), def_alpha=0, def_gain=1, cuda_idx=7),
    'softplus': SimpleNamespace(func=lambda x, **_: torch.nn.functional.softplus(x), def_alpha=0, def_gain=1, cuda_idx=8),
    'swish':    SimpleNamespace(func=lambda x, **_: torch.sigmoid(x) * x, def_alpha=0, def_gain=math.sqrt(2), cuda_idx=9),
}


def _parse(act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    return spec, float(alpha if alpha is not None else spec.def_alpha), float(gain if gain is not None else spec.def_gain), \
        float(clamp if clamp is not None else -1)


def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    spec, alpha, gain, clamp = _parse(act, alpha, gain, clamp)
    if b is not None:
        assert b.ndim == 1 and 0 <= dim < x.ndim and b.shape[0] == x.shape[dim]
        x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
    x = spec.func(x, alpha=alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


_cache = {}


def _bias_act_hip(dim, act, alpha, gain, clamp):
    key = (dim, act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    idx = activation_funcs[act].cuda_idx

    def launch(x, b, dy, grad):
        out = torch.empty_like(x)
        dt = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        step = 1
        for s in x.shape[dim + 1:]:
            step *= s
        lib.sidlsg_bias_act(x.data_ptr(), ops._p(b), ops._p(dy), out.data_ptr(), x.numel(), step, x.shape[dim], idx, alpha, gain,
                            clamp, grad, dt, ops._s())
        return out

    class BiasActHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            x = x.contiguous()
            b = b.contiguous().to(x.dtype) if b is not None else None
            ctx.save_for_backward(x, b)
            return launch(x, b, None, 0)

        @staticmethod
        def backward(ctx, dy):
            x, b = ctx.saved_tensors
            if torch.is_grad_enabled() and (dy.requires_grad or x.requires_grad):
                raise NotImplementedError('second-order bias_act gradients are not implemented in the HIP plugin')
            dx = launch(x, b, dy.contiguous(), 1)
            db = None
            if b is not None and ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    _cache[key] = BiasActHip
    return BiasActHip


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'ref':
        return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)
    if x.device.type != 'cuda':
        raise RuntimeError("bias_act(impl='cuda') needs a GPU tensor; pass impl='ref' explicitly for the PyTorch formulation")
    _, alpha, gain, clamp = _parse(act, alpha, gain, clamp)
    return _bias_act_hip(dim, act, alpha, gain, clamp).apply(x, b)
