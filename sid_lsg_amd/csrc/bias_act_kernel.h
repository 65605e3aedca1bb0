// bias_act device code shared by the main library (sidlsg_bias_act, elementwise.hip) and the stand-alone plugin that
// sid_lsg_amd.custom_ops.get_plugin builds (plugins/bias_act_plugin.hip) -- the ROCm counterpart of the reference's
// torch_utils/ops/bias_act.cu kernel (one thread per element, runtime activation switch instead of template expansion,
// derivatives evaluated from the recomputed pre-activation instead of saved outputs).
//   grad 0: out = clamp(act(x + b[(i/stepB)%sizeB]) * gain)
//   grad 1: out = dy * gain * act'(x + b)                 (0 where the forward output was clamped)
//   grad 2: out = ddx * dy * gain * act''(x + b)          (second order: ddx is passed through the `ddx` pointer)
#pragma once
#include "common.h"

DEVFN float act_fwd(int a, float x, float alpha) {
    switch (a) {
        case 1: return x;
        case 2: return x > 0.f ? x : 0.f;
        case 3: return x > 0.f ? x : x * alpha;
        case 4: return tanhf(x);
        case 5: return 1.f / (1.f + expf(-x));
        case 6: return x > 0.f ? x : expm1f(x);
        case 7: return 1.0507009873554805f * (x > 0.f ? x : 1.6732632423543772f * expm1f(x));
        case 8: return x > 20.f ? x : log1pf(expf(x));
        case 9: return x / (1.f + expf(-x));
    }
    return x;
}
// derivative of act at pre-activation x (y = act(x))
DEVFN float act_grad(int a, float x, float y, float alpha) {
    switch (a) {
        case 1: return 1.f;
        case 2: return x > 0.f ? 1.f : 0.f;
        case 3: return x > 0.f ? 1.f : alpha;
        case 4: return 1.f - y * y;
        case 5: return y * (1.f - y);
        case 6: return x > 0.f ? 1.f : y + 1.f;
        case 7: return x > 0.f ? 1.0507009873554805f : y + 1.0507009873554805f * 1.6732632423543772f;
        case 8: return 1.f / (1.f + expf(-x));
        case 9: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f + x * (1.f - s)); }
    }
    return 1.f;
}
// second derivative
DEVFN float act_grad2(int a, float x, float y, float alpha) {
    (void)alpha;
    switch (a) {
        case 4: return -2.f * y * (1.f - y * y);
        case 5: return y * (1.f - y) * (1.f - 2.f * y);
        case 6: return x > 0.f ? 0.f : y + 1.f;
        case 7: return x > 0.f ? 0.f : y + 1.0507009873554805f * 1.6732632423543772f;
        case 8: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
        case 9: { const float s = 1.f / (1.f + expf(-x)); return s * (1.f - s) * (2.f + x * (1.f - 2.f * s)); }
    }
    return 0.f;     // linear, relu, lrelu
}
template <typename T>
__global__ void bias_act_kernel(const T* __restrict__ x, const T* __restrict__ b, const T* __restrict__ dy, const T* __restrict__ ddx,
                                T* __restrict__ out, size_t n, int stepB, int sizeB, int act, float alpha, float gain, float clamp,
                                int grad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = (float)x[i];
    if (b) v += (float)b[(i / stepB) % sizeB];
    const float y = act_fwd(act, v, alpha);
    float o;
    if (grad == 0) {
        o = y * gain;
        if (clamp >= 0.f) o = fminf(fmaxf(o, -clamp), clamp);
    } else {
        o = (float)dy[i] * gain * (grad == 1 ? act_grad(act, v, y, alpha) : (float)ddx[i] * act_grad2(act, v, y, alpha));
        if (clamp >= 0.f) { const float yy = y * gain; if (yy > clamp || yy < -clamp) o = 0.f; }
    }
    out[i] = (T)o;
}

// dtype 0 = fp32, 1 = bf16, 2 = fp16, 3 = fp64 (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF: bias_act.cpp:77; every
// type computes in fp32 registers like the reference's `scalar_t = float` for half, fp64 included: the plugin is an activation epilogue)
#define SIDLSG_BIAS_ACT_LAUNCH(T)                                                                                                      \
    hipLaunchKernelGGL(bias_act_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)b, (const T*)dy, (const T*)ddx, \
                       (T*)out, (size_t)n, stepB, sizeB, act, alpha, gain, clamp, grad)
static inline int bias_act_launch(const void* x, const void* b, const void* dy, const void* ddx, void* out, long long n, int stepB,
                                  int sizeB, int act, float alpha, float gain, float clamp, int grad, int dtype, void* stream) {
    if (act < 1 || act > 9 || grad < 0 || grad > 2 || (grad >= 1 && !dy) || (grad == 2 && !ddx) || n < 0 || dtype < 0 || dtype > 3) return SIDLSG_EINVAL;
    if (n == 0) return SIDLSG_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == 0) SIDLSG_BIAS_ACT_LAUNCH(float);
    else if (dtype == 1) SIDLSG_BIAS_ACT_LAUNCH(bf16);
    else if (dtype == 2) SIDLSG_BIAS_ACT_LAUNCH(_Float16);
    else SIDLSG_BIAS_ACT_LAUNCH(double);
    return sidlsg_last_error();
}
