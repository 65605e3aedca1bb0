// Stand-alone HIP plugin built by sid_lsg_amd.custom_ops.get_plugin('bias_act_plugin', sources=[this file]) -- the ROCm
// counterpart of the reference's plugin pair torch_utils/ops/bias_act.cpp (host entry, :32-97) + bias_act.cu (kernel).
// C ABI instead of pybind: the tensor-level function `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`
// the reference's wrapper calls on `_plugin` is provided by bias_act_plugin_binding.py next to this file.
#include "../bias_act_kernel.h"

extern "C" int bias_act_plugin_launch(const void* x, const void* b, const void* dy, const void* ddx, void* out, long long n, int stepB,
                                      int sizeB, int act, float alpha, float gain, float clamp, int grad, int dtype, void* stream) {
    return bias_act_launch(x, b, dy, ddx, out, n, stepB, sizeB, act, alpha, gain, clamp, grad, dtype, stream);
}
