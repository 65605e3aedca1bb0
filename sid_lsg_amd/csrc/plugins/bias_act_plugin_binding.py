"""Tensor-level binding of bias_act_plugin.so: what PYBIND11_MODULE does for the reference's plugin
(torch_utils/ops/bias_act.cpp:94-97 `m.def("bias_act", &bias_act)`), done with ctypes on the C ABI.
`bind(dll)` returns {name: callable}; custom_ops.get_plugin turns it into the module it hands back."""
import ctypes

import torch


# the reference plugin takes fp64 / fp32 / fp16 (bias_act.cpp:77); bf16 is this package's production activation type
_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2, torch.float64: 3}


def bind(dll):
    fn = dll.bias_act_plugin_launch
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]

    def ptr(t):
        return t.data_ptr() if (t is not None and t.numel()) else None

    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        """Argument order and meaning of the reference plugin (bias_act.cpp:32): grad 0: x = input; grad 1: x = dy, xref = the
        saved input; grad 2: x = the second-order seed d(dx), xref = saved input, dy = the first-order upstream gradient.
        Empty tensors stand for absent optionals (bias_act.py:39 `_null_tensor`).  yref is accepted and unused: derivatives
        are evaluated from the recomputed pre-activation."""
        src = x if grad == 0 else xref
        if not src.is_cuda:
            raise RuntimeError('bias_act plugin: tensors must be on the GPU')
        if src.dtype not in _DTYPES:
            raise RuntimeError(f'bias_act plugin: unsupported dtype {src.dtype}')
        src = src.contiguous()
        out = torch.empty_like(src)
        step = 1
        for s in src.shape[dim + 1:]:
            step *= s
        bb = b.contiguous().to(src.dtype) if (b is not None and b.numel()) else None
        g1 = (x if grad == 1 else dy)
        g1 = g1.contiguous().to(src.dtype) if (grad >= 1) else None
        g2 = x.contiguous().to(src.dtype) if grad == 2 else None
        rc = fn(src.data_ptr(), ptr(bb), ptr(g1), ptr(g2), out.data_ptr(), src.numel(), step, src.shape[dim] if src.ndim else 1,
                int(act), float(alpha), float(gain), float(clamp), int(grad), _DTYPES[src.dtype],
                torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f'bias_act_plugin_launch failed with code {rc}')
        return out
    return dict(bias_act=bias_act)
