// Fused optimizer step over a network's FLAT parameter buffer (gfx950, HBM-bound).
// SURVEY.md section 8 rows A9 + A10 in one pass:
//   g <- nan_to_num(g, 0, +-1e5)            (sid_training_loop.py:458-460, 541-543)
//   g <- clamp(g, +-clip)  if clip > 0       (fp16-only branch, :546-547)
//   Adam / AdamW (torch.optim single-tensor semantics; beta1 = 0 needs no first-moment buffer)
//   p_ema <- lerp(p, p_ema, ema_beta)        (:553-565)      [optional]
//   w_bf16 <- bf16(p)                        (compute copy for the MFMA kernels) [optional]
//   g <- 0                                   (zero_grad folded in)               [optional]
// One read of g,p,v(,m,ema) and one write of p,v(,m,ema,w,g): 28-36 B per parameter instead of
// the ~60 B of the reference's foreach kernels + python nan_to_num loop.
// Scalars live in device memory (hyper[]) so a captured HIP graph can be replayed with a new step.
#include "common.h"

// hyper: [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=bias_corr1 [5]=sqrt(bias_corr2) [6]=ema_beta [7]=weight_decay
//        [8]=decoupled(0/1) [9]=clip (<=0: off) [10]=grad_scale (e.g. 1/world for a summed all-reduce)
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, float* __restrict__ ema, bf16* __restrict__ w,
                                                       const float* __restrict__ hyper, size_t n, int zero_grad) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], bc1 = hyper[4], bc2 = hyper[5];
    const float eb = hyper[6], wd = hyper[7], clip = hyper[9], gs = hyper[10];
    const bool decoupled = hyper[8] != 0.f;
    const float step = lr / bc1;  // hyper[5] holds sqrt(bias_corr2), as torch computes it on the host
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + i), gv = *reinterpret_cast<f32x4*>(g + i), vv = *reinterpret_cast<f32x4*>(v + i);
            f32x4 mv = {0, 0, 0, 0}, ev = {0, 0, 0, 0};
            if (m) mv = *reinterpret_cast<f32x4*>(m + i);
            if (ema) ev = *reinterpret_cast<f32x4*>(ema + i);
            bf16x4 wv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float gg = gv[e] * gs;
                gg = (gg != gg) ? 0.f : fminf(fmaxf(gg, -1e5f), 1e5f);
                if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
                float pp = pv[e];
                if (wd != 0.f) { if (decoupled) pp *= (1.f - lr * wd); else gg += wd * pp; }
                float mm = gg;
                if (m) { mm = mv[e] + (gg - mv[e]) * (1.f - b1); mv[e] = mm; }
                const float v2 = b2 * vv[e] + (1.f - b2) * gg * gg;
                vv[e] = v2;
                pp -= step * mm / (sqrtf(v2) / bc2 + eps);
                pv[e] = pp;
                if (ema) { const float d = ev[e] - pp; ev[e] = eb < 0.5f ? pp + eb * d : ev[e] - d * (1.f - eb); }
                wv[e] = f2bf(pp);
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
            if (m) *reinterpret_cast<f32x4*>(m + i) = mv;
            if (ema) *reinterpret_cast<f32x4*>(ema + i) = ev;
            if (w) *reinterpret_cast<bf16x4*>(w + i) = wv;
            if (zero_grad) *reinterpret_cast<f32x4*>(g + i) = (f32x4){0, 0, 0, 0};
        } else {
            for (size_t j = i; j < n; j++) {
                float gg = g[j] * gs;
                gg = (gg != gg) ? 0.f : fminf(fmaxf(gg, -1e5f), 1e5f);
                if (clip > 0.f) gg = fminf(fmaxf(gg, -clip), clip);
                float pp = p[j];
                if (wd != 0.f) { if (decoupled) pp *= (1.f - lr * wd); else gg += wd * pp; }
                float mm = gg;
                if (m) { mm = m[j] + (gg - m[j]) * (1.f - b1); m[j] = mm; }
                const float v2 = b2 * v[j] + (1.f - b2) * gg * gg;
                v[j] = v2;
                pp -= step * mm / (sqrtf(v2) / bc2 + eps);
                p[j] = pp;
                if (ema) { const float d = ema[j] - pp; ema[j] = eb < 0.5f ? pp + eb * d : ema[j] - d * (1.f - eb); }
                if (w) w[j] = f2bf(pp);
                if (zero_grad) g[j] = 0.f;
            }
        }
    }
}

extern "C" {

int sidlsg_adam_ema(float* p, float* g, float* m, float* v, float* ema, void* w_bf16, const float* hyper, long long n,
                    int zero_grad, void* stream) {
    if (!p || !g || !v || !hyper || n <= 0) return SIDLSG_EINVAL;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)v) & 15) return SIDLSG_EINVAL;
    size_t blocks = ((size_t)n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;   // 16 blocks/CU, grid-stride the rest (1 / 2 / 4 per CU measured the same in the step)
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (bf16*)w_bf16,
                       hyper, (size_t)n, zero_grad);
    return sidlsg_last_error();
}

}  // extern "C"
