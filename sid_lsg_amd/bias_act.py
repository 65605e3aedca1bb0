"""`bias_act` with the reference's Python API (torch_utils/ops/bias_act.py:58), backed by a gfx950 kernel.

Same argument meaning, same `activation_funcs` table (names, default alpha/gain, plugin index 1..9).
impl='cuda' (default) runs the HIP kernel (sidlsg_bias_act) and RAISES if the tensor is not on the
GPU or the library is missing -- unlike the reference, whose CUDA plugin is disabled outright
(`_init()` returns False, bias_act.py:53-54) and silently uses the slow path.  impl='ref' is the same
explicit plain-PyTorch formulation the reference exposes under that name.
The kernel is reached the way the reference reaches its own: `custom_ops.get_plugin('bias_act_plugin', sources=...)`
(bias_act.py:41-51) returns a module and the autograd classes call `_plugin.bias_act(x, b, xref, yref, dy, grad, ...)`
(:146-212) for the forward (grad 0), first-order (grad 1) and second-order (grad 2) passes.
"""
import math
import os
from types import SimpleNamespace

import torch

from . import custom_ops

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
_plugin = None
_null_tensor = torch.empty([0])
_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')


def _init():
    """Build / load the plugin through the loader seam, as the reference's wrapper does (bias_act.py:41-51) -- except that a
    build failure is an error here, not a silent fall-back to the slow path."""
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin('bias_act_plugin', sources=[os.path.join(_CSRC, 'plugins', 'bias_act_plugin.hip')],
                                        headers=[os.path.join(_CSRC, 'bias_act_kernel.h'), os.path.join(_CSRC, 'common.h')])
    return True


def _bias_act_hip(dim, act, alpha, gain, clamp):
    """Autograd classes per static configuration (cached like bias_act.py:130-145): forward, first-order (dx, db) and
    second-order gradients, each a call of the plugin's `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`."""
    key = (dim, act, alpha, gain, clamp)
    if key in _cache:
        return _cache[key]
    idx = activation_funcs[act].cuda_idx
    nul = _null_tensor

    class BiasActHip(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            x = x.contiguous()
            b = b.contiguous().to(x.dtype) if b is not None else nul
            ctx.save_for_backward(x, b)
            return _plugin.bias_act(x, b, nul, nul, nul, 0, dim, idx, alpha, gain, clamp)

        @staticmethod
        def backward(ctx, dy):
            x, b = ctx.saved_tensors
            dx = BiasActHipGrad.apply(dy.contiguous(), x, b)
            db = None
            if b.numel() and ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActHipGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b):
            ctx.save_for_backward(dy, x, b)
            return _plugin.bias_act(dy, b, x, nul, nul, 1, dim, idx, alpha, gain, clamp)

        @staticmethod
        def backward(ctx, d_dx):
            dy, x, b = ctx.saved_tensors
            d_dx = d_dx.contiguous()
            d_dy = BiasActHipGrad.apply(d_dx, x, b) if ctx.needs_input_grad[0] else None     # the op is linear in dy
            d_x = d_b = None
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                d_x = _plugin.bias_act(d_dx, b, x, nul, dy, 2, dim, idx, alpha, gain, clamp)
                if b.numel() and ctx.needs_input_grad[2]:
                    d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b

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
    _init()
    return _bias_act_hip(dim, act, alpha, gain, clamp).apply(x, b)
