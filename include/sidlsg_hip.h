/* sidlsg_hip.h -- C ABI of libsidlsg_hip.so: the MI355X (gfx950) kernels of the SiD-LSG
 * distillation inner step.  Plain pointers and sizes only (device pointers unless noted);
 * `stream` is a hipStream_t passed as void*.  Every function returns 0 on success, a positive
 * hipError_t if the launch failed, or -22 (EINVAL) if the arguments violate the stated
 * constraints.  No function allocates, synchronises or touches the host heap, so all of them may
 * be captured into a HIP graph.
 *
 * What each entry point replaces in the reference (paths relative to mingyuanzhou/SiD-LSG):
 * the reference has no native code on this path -- every op below is executed there by
 * diffusers/ATen kernels reached through `unet(...).sample` (training/sid_sd_util.py:184,194,
 * 245,263), `torch.optim.Adam.step` (training/sid_training_loop.py:462,549) and inline torch
 * expressions in the loop.  Its only native-operator seam is the plugin API of
 * torch_utils/custom_ops.py:46 + torch_utils/ops/bias_act.cpp:32-97, which sidlsg_bias_act
 * mirrors.  Activations are NHWC bf16 ([tokens][channels], channels contiguous); conv weights
 * are [Cout][3][3][Cin] (= torch channels_last storage of the diffusers [Cout][Cin][3][3]
 * parameter); parameters, statistics, losses and gradients of parameters are fp32.
 */
#ifndef SIDLSG_HIP_H
#define SIDLSG_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* epilogue flags of the GEMM / conv entry points */
#define SIDLSG_OUT_F32 1 /* C is fp32 instead of bf16 */
#define SIDLSG_SILU 2    /* y = silu(...) */
#define SIDLSG_ACCUM 4   /* C += ... (fp32 only) */

/* ---- contractions (MFMA bf16 -> fp32) ------------------------------------------------------
 * torch.nn.functional.linear / conv2d(k=1) inside diffusers Attention / FeedForward /
 * Transformer2DModel / ResnetBlock2D.time_emb_proj / conv_shortcut (call site sid_sd_util.py:184).
 * C[M][ldc] = act(alpha * A[M][K] W[N][K]^T + bias[N] + rowvec[m/rows_per_batch][ld_rowvec] + res[M][ldres])
 * K, lda multiples of 8.  bias/res/rowvec may be NULL; ld_rowvec <= 0 means N (a slice of a wider matrix otherwise). */
int sidlsg_gemm_bf16(const void* A, int lda, const void* W, void* C, int ldc, const float* bias, const void* res, int ldres,
                     const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha, int flags,
                     void* stream);

/* torch conv2d(k=3, pad=1, stride 1|2) of ResnetBlock2D.conv1/conv2, Downsample2D, Upsample2D
 * (ups=1 fuses the nearest x2 interpolate), conv_in/conv_out.  X: [B][Hs][Ws][ldx] with the virtual
 * (post-upsample) size H x Wd; W: [Cout][3][3][Cin]; Y: [B][Ho][Wo][ldc].  Same epilogue as the GEMM;
 * rowvec is the per-sample time-embedding projection [B][ld_rowvec] (all ResBlocks' projections come out of ONE GEMM;
 * each conv reads its column slice).  Cin, ldx multiples of 8.
 * The backward-data pass is the same entry point fed with dY and the transposed/flipped weights
 * made by sidlsg_transpose_w. */
int sidlsg_conv3x3_bf16(const void* X, int ldx, const void* W, void* Y, int ldc, const float* bias, const void* res, int ldres,
                        const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout, int stride, int ups,
                        float alpha, int flags, void* stream);

/* diffusers GEGLU (`FeedForward.net[0]`: proj -> chunk(2) -> hidden * gelu(gate)) with its projection, in ONE kernel: the GEMM's
 * output tile holds 80 features of the value half and the same 80 features of the gate half, so the epilogue emits
 * Y[M][F] = h[:, :F] * gelu(h[:, F:]) next to h[M][N2] = A W^T + bias (N2 = 2 F).  h is what the backward needs (sidlsg_geglu_bwd);
 * H = NULL skips it (a pass without backward).  Y is computed from the bf16-rounded h: bit for bit sidlsg_geglu_fwd(h).  Saves the
 * separate GEGLU pass over h (335 MB read at the 64x64 stage of SD1.5, batch 16).  sidlsg_gemm_geglu_ok: host query -- the fused kernel
 * takes a shape only where the direct-to-LDS GEMM would be dispatched without split-K. */
int sidlsg_gemm_geglu_ok(int M, int N2, int K);
int sidlsg_gemm_geglu_bf16(const void* A, int lda, const void* W, void* H, int ldh, void* Y, int ldy, const float* bias, int M, int N2,
                           int K, void* stream);

/* The backward of the same block (diffusers `FeedForward`: net[2] = Linear(4C -> C) after the GEGLU): data gradient of the FF-out
 * projection with the GEGLU derivative in its epilogue -- dy = dOut Wt^T (Wt [F][K]: the backward-data operand of the [K][F] FF-out
 * weight) stays in the output tile (rounded to bf16 there: bit for bit what sidlsg_geglu_bwd computes from a stored dy),
 * dH[:, :F] = dy * gelu(h[:, F:]), dH[:, F:] = dy * h[:, :F] * gelu'(h[:, F:]); H, dH: [M][ldh], ldh >= 2 F.  Saves the write and
 * the re-read of dy [M][F] (2 x 168 MB at the 64x64 stage of SD1.5, batch 16) and one launch.  autograd reaches it through
 * `loss.backward()` (sid_training_loop.py:450,533).  sidlsg_gemm_geglu_bwd_ok: host query, same admission rule as the forward. */
int sidlsg_gemm_geglu_bwd_ok(int M, int F, int K);
int sidlsg_gemm_geglu_bwd_bf16(const void* dOut, int lda, const void* Wt, const void* H, void* dH, int ldh, int M, int F, int K,
                               void* stream);
/* Grouped variants of the two GEGLU fusions (two frozen networks of one shape on one stacked batch, M even: rows [0, M/2) are contracted
 * with (W, bias) / Wt, rows [M/2, M) with (W1, bias1) / Wt1; otherwise as the single-set entry points above).  The grouped frozen pass of
 * phase B (sid_training_loop.py:494-506: fake-score network and teacher evaluate the same images) takes its FeedForward blocks through
 * these; its backward is the data gradient only. */
int sidlsg_gemm_geglu_bf16_g2(const void* A, int lda, const void* W, const void* W1, void* H, int ldh, void* Y, int ldy, const float* bias,
                              const float* bias1, int M, int N2, int K, void* stream);
int sidlsg_gemm_geglu_bwd_bf16_g2(const void* dOut, int lda, const void* Wt, const void* Wt1, const void* H, void* dH, int ldh, int M, int F,
                                  int K, void* stream);

/* Optional fp32 scratch (device memory owned by the caller): per-split partial-sum slabs for split-K GEMMs/convs with few
 * output tiles and long K (8x8 / 16x16 stages) and for the pixel-split weight gradients (without it they fall back to
 * fp32 atomics, ~2-3x slower on MI355X).  Default for every stream without a private workspace (below): launches sharing
 * it must be ordered on one stream.  NULL disables. */
int sidlsg_set_workspace(void* ptr, long long bytes);
/* A private workspace for launches on `stream` (ptr = NULL removes it; at most 4): needed when two streams run
 * contractions concurrently -- the teacher beside the fake-score network in phase B of the step. */
int sidlsg_set_stream_workspace(void* stream, void* ptr, long long bytes);

/* weight gradients (autograd of the two ops above in the reference: loss.backward(),
 * sid_training_loop.py:450,533).  dW[N][K] += dY[M][N]^T A[M][K] in fp32 (pixel-split partial sums, reduced through the
 * workspace).  dBias (may be NULL): dBias[N] += sum_m dY[m][N], the bias gradient of the same layer, produced by the
 * same kernel (one extra MFMA against a vector of ones per dY fragment) instead of a separate column-sum pass. */
int sidlsg_wgrad_bf16(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K,
                      void* stream);
/* The same with dW = (not +=), dBias still +=: for gradient storage whose previous contents are dead.  The fused optimizer
 * (sidlsg_adam_ema, zero_grad = 0 on the weight ranges) leaves the weight gradients of a network un-zeroed and the first weight
 * gradient after it overwrites them: 4 B / parameter of zero stores and 4 B / parameter of dW reads less per optimizer step.  Same
 * summation order as the accumulating entry points on a zeroed dW: bit-identical results. */
int sidlsg_wgrad_assign_bf16(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K,
                             void* stream);
/* Several dense weight gradients as ONE launch + one slab-reduction launch (round 5: the C x C projections of a transformer block each
 * need ~56 pixel splits to fill the chip alone; grouped, the group fills it with ~6 each).  jobs: HOST array of njobs <= 8 records of
 * 64 bytes  { const void* dY; const void* A; float* dW; float* dBias; int ldy, lda, M, N, K, assign; int pad[2]; }  with the meaning
 * of sidlsg_wgrad_bf16 (assign = 0) / sidlsg_wgrad_assign_bf16 (assign = 1) per record; N, K, ldy, lda % 8 == 0, 16-byte aligned operands.
 * The dW of the records of one call must be DISTINCT buffers (the jobs run as one grid: two jobs on one dW would race). */
int sidlsg_wgrad_group_bf16(const void* jobs, int njobs, void* stream);
int sidlsg_wgrad_group160_bf16(const void* jobs, int njobs, void* stream); /* the same on 160 x 160 tiles: every N and K a multiple of 160 */
int sidlsg_conv3x3_wgrad_assign_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                                     int Cin, int Cout, int stride, int ups, void* stream);
int sidlsg_debug_wgrad_blocks_per_cu(int which); /* host diagnostic: resident blocks per CU of the weight-gradient kernels (0: 128x128, 1: 160x128, 2: 160x160 tiles) */
int sidlsg_debug_set_p8(int mode); /* A/B switch: which GEMM / conv calls take the 256-row-tile kernels (csrc/gemm_p8.h): -1 by rule (default), 0 never, 1 / 2 always the 256x160 / 256x320 configuration when admissible; returns the previous setting */
int sidlsg_conv3x3_wgrad_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                              int Cin, int Cout, int stride, int ups, void* stream);

/* ---- normalisation (HBM bound) ------------------------------------------------------------
 * torch.nn.GroupNorm(32, C, eps)+SiLU of ResnetBlock2D.norm1/norm2, Transformer2DModel.norm,
 * conv_norm_out; torch.nn.LayerNorm of BasicTransformerBlock.norm1-3. */
/* backward: dres (may be NULL, same shape as x) is the gradient reaching x through its other consumer (the residual /
 * shortcut branch of the block that owns the norm); dx = norm-backward + dres in the same pass. */
int sidlsg_groupnorm_ws_floats(int B, int HW, int C, int G); /* host: workspace size in floats, <0 on bad shape */
int sidlsg_groupnorm_nchunks(int B, int HW, int C, int G);   /* host */
int sidlsg_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, float* ws, int B,
                         int HW, int C, int G, float eps, int silu, void* stream);
int sidlsg_groupnorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                         const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G,
                         int silu, void* stream);
int sidlsg_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C,
                         float eps, void* stream);
int sidlsg_layernorm_bwd_nblocks(int rows); /* host: ws = nblocks*C*2 floats */
/* Deferred dgamma / dbeta reductions (csrc/norm.hip "deferred parameter-gradient reductions"): with deferral on for a stream the norm
 * backward entry points queue their final reduction; sidlsg_flush_reductions launches everything queued for the stream as ONE kernel.
 * The caller keeps every `ws` passed to a deferred call allocated and untouched until the flush.  Return: reductions flushed, < 0 error;
 * sidlsg_pending_reductions: queued reductions, -1 when deferral is off for the stream. */
int sidlsg_defer_reductions(void* stream, int on); /* 1: queue from now on; 2: stop queuing, keep the queue; 0: flush + forget */
int sidlsg_flush_reductions(void* stream);
int sidlsg_pending_reductions(void* stream);
int sidlsg_layernorm_bwd(const void* x, const void* dy, const float* stats, const float* gamma, const void* dres, void* dx,
                         float* dgamma, float* dbeta, float* ws, int rows, int C, void* stream);

/* ---- grouped launches: TWO networks of identical architecture evaluated on ONE stacked batch ---------------------------
 * Phase B of the step evaluates the frozen fake-score network and the frozen teacher on identical inputs
 * (`sid_sd_denoise(unet=fake_score, ...)` and `sid_sd_denoise(unet=true_score, ...)`, sid_training_loop.py:494-506, each a CFG
 * batch through sid_sd_util.py:259-265).  Here both run as ONE pass over a stacked batch [set 0 samples ; set 1 samples]:
 * every weight-bearing kernel takes a second parameter set and picks it per block (rows / samples of the second half), every
 * parameter-free kernel (attention, GEGLU, SiLU, concat) simply sees twice the batch.  One grid of 2x the blocks fills the
 * chip where either network's grid alone does not (16x16 / 8x8 stages, the time-embedding MLP, the 77-token K/V
 * projections), and the frozen passes issue half the launches.  M (rows) resp. B (samples) must be even; the halves need not
 * be tile aligned.  Same epilogue contract as the ungrouped entry points; res / rowvec / outputs are stacked like the input.
 * The backward-data pass of a layer is the same entry point with (dY, transposed weight set 0, transposed weight set 1).
 * bf16 only; frozen networks only (there is no grouped weight gradient). */
int sidlsg_gemm_bf16_g2(const void* A, int lda, const void* W, const void* W1, void* C, int ldc, const float* bias, const float* bias1,
                        const void* res, int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K,
                        float alpha, int flags, void* stream);
int sidlsg_conv3x3_bf16_g2(const void* X, int ldx, const void* W, const void* W1, void* Y, int ldc, const float* bias, const float* bias1,
                           const void* res, int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout,
                           int stride, int ups, float alpha, int flags, void* stream);
int sidlsg_groupnorm_fwd_g2(const void* x, const float* gamma, const float* beta, const float* gamma1, const float* beta1, void* y,
                            float* stats, float* ws, int B, int HW, int C, int G, float eps, int silu, void* stream);
int sidlsg_groupnorm_bwd_g2(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta, const float* gamma1,
                            const float* beta1, const void* dres, void* dx, float* ws, int B, int HW, int C, int G, int silu,
                            void* stream);
int sidlsg_layernorm_fwd_g2(const void* x, const float* gamma, const float* beta, const float* gamma1, const float* beta1, void* y,
                            float* stats, int rows, int C, float eps, void* stream);
int sidlsg_layernorm_bwd_g2(const void* x, const void* dy, const float* stats, const float* gamma, const float* gamma1, const void* dres,
                            void* dx, int rows, int C, void* stream);

/* ---- in-library kernel timing (measurement only; bench.py) -------------------------------------------------------------------
 * Kernel durations as `rocprofv3 --kernel-trace` reports them, taken live: while tracing is enabled every `stride`-th CALL of a
 * kernel family launches its kernels with start / stop events bound to the kernel's own dispatch packet (hipExtLaunchKernelGGL) --
 * no extra packets in the stream, unlike events recorded around a launch.  Families: 0 dense GEMM fwd + dgrad, 1 conv3x3 fwd +
 * dgrad, 2 attention forward, 3 attention backward, 4 dense weight gradient, 5 conv weight gradient, 6 GroupNorm forward,
 * 7 GroupNorm backward, 8 LayerNorm forward, 9 LayerNorm backward.  Work = algorithmic flop (0-5) or bytes (6-9) of the call.
 * sidlsg_trace_read (after a device synchronisation; out: 9 doubles, out[7] / out[8] = peak flop/s and bytes/s on input):
 * out[0] summed kernel ms of the sampled calls, out[1] their summed work, out[2] sampled calls, out[3] all calls of the family since
 * enable, out[4] timed kernels, out[5] their summed algorithmic bytes, out[6] their summed roofline time in ms (per call
 * max(flop / peak flop/s, bytes / peak bytes/s): every call is graded against its own bound). */
int sidlsg_trace_enable(int max_kernels);
int sidlsg_trace_pause(int paused);
int sidlsg_trace_set_stride(int family, int stride);
int sidlsg_trace_read(int family, double* out);

/* ---- attention (diffusers Attention + AttnProcessor2_0 / xformers; sid_sd_util.py:102-113) --
 * O = softmax(Q K^T D^-1/2) V per head; Q/K/V/O are strided views ([b][token][h*D + d], token
 * stride ld*, batch stride bs*, in elements) so the fused QKV projection is consumed in place.
 * D multiple of 8, <= 160.  LSE [B][H][Nq] fp32 (log2 domain) is saved for backward. */
int sidlsg_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                    int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                    void* stream);
/* backward: dK = dV = NULL -> only dQ (and delta) are produced. */
int sidlsg_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                    void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo,
                    long long bsq, long long bsk, long long bsv, long long bso, void* stream);
/* The same with PRE-SCALED queries: Q holds q * (D^-1/2 * log2 e) -- the caller folds the factor into the q rows of the
 * projection weight's FORWARD compute copy (sidlsg_scale_cast_ranges: one rounding, like the unscaled copy).  The QK^T
 * accumulators are then seeded with -max / -LSE and feed v_exp_f32 directly.  The backward returns the SAME tensors as
 * sidlsg_attn_bwd: gradients with respect to the unscaled q (= Q / (D^-1/2 log2 e)), k and v -- so the projection's weight
 * gradient is the ordinary one and its backward-data operand is the UNSCALED transposed weight. */
int sidlsg_attn_fwd_ps(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                       int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                       void* stream);
int sidlsg_attn_bwd_ps(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                       void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv, int ldo,
                       long long bsq, long long bsk, long long bsv, long long bso, void* stream);

/* ---- scheduler / guidance glue (sid_sd_util.py:182-185, 242, 259-272) ----------------------
 * noisy_input: x_t = s0[b]*x0 + s1[b]*noise (x0 NULL -> zeros: the one-step generator input),
 *   NCHW fp32 in, NHWC bf16 [dup*B][HW][Cp] out (dup=2: [uncond ; cond] halves of the CFG batch),
 *   optional fp32 NCHW copy of x_t.
 * cfg_x0: eps [dup*B][HW][Ce>=C] fp32 -> out NCHW fp32 = predict_x0 ? (x_t - s1*e)/s0 : e with
 *   e = u + kappa*(c-u) when dup=2. */
int sidlsg_noisy_input(const float* x0, const float* noise, const float* s0, const float* s1, void* out, float* xt, int B,
                       int C, int HW, int Cp, int dup, void* stream);
int sidlsg_noisy_input_bwd(const void* g, const float* s0, float* dx0, int B, int C, int HW, int Cp, int dup, int accumulate,
                           void* stream);
int sidlsg_cfg_x0(const float* eps, const float* xt, const float* s0, const float* s1, float* out, int B, int C, int HW,
                  int Ce, int dup, float kappa, int predict_x0, void* stream);
int sidlsg_cfg_x0_bwd(const float* g, const float* s0, const float* s1, void* deps, float* dxt, int B, int C, int HW, int Cp,
                      int dup, float kappa, int predict_x0, void* stream);

/* ---- losses with closed-form gradients (sid_training_loop.py:423-445, 508-530) -------------
 * Per-sample NaN filtering is done in-kernel (a sample containing NaN contributes 0 and gets zero
 * gradients), `scale` = loss_scaling / batch_gpu_total.  ws >= 2*S + 8*S floats.  loss: 1 float. */
int sidlsg_g_loss(const float* x, const float* yr, const float* yf, float* dx, float* dyr, float* dyf, float* loss, float* ws,
                  int S, int n, float alpha, float scale, void* stream);
int sidlsg_fake_loss(const float* e, const float* noise, float* de, float* loss, float* ws, int S, int n, float scale,
                     void* stream);

/* ---- optimizer: nan_to_num + clip + Adam/AdamW + EMA + bf16 weight copy + zero_grad --------
 * (sid_training_loop.py:458-462, 541-565; sid_train.py:219-226).  hyper: 11 floats in DEVICE memory:
 * lr, beta1, beta2, eps, bias_corr1, sqrt(bias_corr2), ema_beta, weight_decay, decoupled, clip, grad_scale.
 * m may be NULL when beta1 == 0; ema, w_bf16 may be NULL. */
int sidlsg_adam_ema(float* p, float* g, float* m, float* v, float* ema, void* w_bf16, const float* hyper, long long n,
                    int zero_grad, void* stream);

/* ---- small elementwise / layout kernels ---------------------------------------------------- */
int sidlsg_timestep_embed(const long long* t, void* out, int B, int dim, void* stream); /* Timesteps(flip_sin_to_cos) */
int sidlsg_silu_fwd(const void* x, void* y, long long n, void* stream);
int sidlsg_silu_bwd(const void* x, const void* dy, void* dx, long long n, void* stream);
int sidlsg_geglu_fwd(const void* h, void* y, long long M, int F, void* stream);           /* diffusers GEGLU */
int sidlsg_geglu_bwd(const void* h, const void* dy, void* dh, long long M, int F, void* stream);
int sidlsg_concat2(void* a, void* b, void* out, long long M, int C1, int C2, int to_parts, void* stream); /* skip concat */
int sidlsg_sumpool2x2(const void* g, void* out, int B, int H, int W, int C, void* stream);   /* bwd of nearest x2 */
int sidlsg_zero_insert2(const void* g, void* out, int B, int Ho, int Wo, int H, int W, int C, void* stream); /* bwd-data of stride 2: out [B][H][W][C] */
int sidlsg_add_bf16(const void* a, const void* b, void* o, long long n, void* stream);
int sidlsg_colsum_nchunks(int B, int rows_per_batch); /* host: grid.x of the reduction */
int sidlsg_colsum(const void* g, int ldg, float* per_batch, float* total, float* ws, int B, int rows_per_batch, int N,
                  void* stream); /* bias / time-embedding gradients: per_batch[B][N] += (zero it first), total[N] +=; ws unused */
/* the same with a row stride for per_batch (elements, >= N): the 22 ResBlocks of a pass accumulate their time-embedding column
 * gradients straight into their column slices of ONE [B][sum Cout] buffer (one fill per pass instead of 22, no concat kernel) */
int sidlsg_colsum_strided(const void* g, int ldg, float* per_batch, int ld_pb, float* total, int B, int rows_per_batch, int N, void* stream);
int sidlsg_cast_f32_bf16(const float* x, void* y, long long n, void* stream);
int sidlsg_cast_bf16_f32(const void* x, float* y, long long n, void* stream);
int sidlsg_transpose_w(const float* src, void* dst, int N, int K, int T, void* stream); /* [N][T][K] -> [K][T rev][N] bf16 */
/* All backward-data operands of a network in ONE launch.  jobs: DEVICE array of njobs records
 *   struct sidlsg_tw_job { const float* src; void* dst; int N, K, T, blk0; };   (32 bytes, no padding)
 * sorted by blk0 = index of the job's first 32x32 tile (job i covers T*ceil(N/32)*ceil(K/32) tiles); nblocks = total. */
int sidlsg_transpose_w_batched(const void* jobs, int njobs, int nblocks, void* stream);
/* Same, from the bf16 COMPUTE copies (src of a record points at bf16 [N][T][K]; 4 instead of 6 bytes per parameter):
 * 64x64 tiles -> blk0 / nblocks count ceil(N/64)*ceil(K/64) tiles per tap; N and K multiples of 8. */
int sidlsg_transpose_w16_batched(const void* jobs, int njobs, int nblocks, void* stream);
/* Scaled re-cast of parameter ranges, dst[i] = bf16(scale * src[i]): folds the attention's D^-1/2 log2(e) factor into the q
 * rows of a projection weight's bf16 COMPUTE copy (sidlsg_attn_fwd_ps).  jobs: DEVICE array of
 *   struct { const float* src; void* dst; int n, blk0; float scale; int pad; }   (32 bytes)
 * sorted by blk0 = index of the job's first block of 2048 elements; nblocks = total. */
int sidlsg_scale_cast_ranges(const void* jobs, int njobs, int nblocks, void* stream);

/* ---- reference plugin op: torch_utils/ops/bias_act.cpp:32 `bias_act(x,b,xref,yref,dy,grad,dim,act,alpha,gain,clamp)`
 * act: 1 linear 2 relu 3 lrelu 4 tanh 5 sigmoid 6 elu 7 selu 8 softplus 9 swish (bias_act.py:23-33).
 * grad 0: out = clamp(act(x + b[(i/stepB)%sizeB]) * gain); grad 1: out = dL/dx from dy (x, b = saved inputs).
 * dtype 0 fp32, 1 bf16. */
int sidlsg_bias_act(const void* x, const void* b, const void* dy, void* out, long long n, int stepB, int sizeB, int act,
                    float alpha, float gain, float clamp, int grad, int dtype, void* stream);

/* ---- fp8-weight contractions (BASELINE.json configs[4] "fp8 MFMA weights"; no counterpart in the reference, which only
 * knows fp32 / fp16: training/sid_training_loop.py:205) ----------------------------------------------------------------
 * For FROZEN networks: W8 = OCP e4m3 bytes [N][K] with one fp32 scale per output channel (sidlsg_quantize_fp8_rows from
 * the bf16 compute copy); A / X stay bf16 in HBM and are converted to e4m3 in the kernel's loader (static unit scale,
 * saturating); v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulation, C = act(alpha * wscale[n] * acc + bias + rowvec + res).
 * Same remaining arguments as sidlsg_gemm_bf16 / sidlsg_conv3x3_bf16; K (Cin) multiple of 16. */
int sidlsg_quantize_fp8_rows(const void* src_bf16, void* dst_fp8, float* scale, int rows, int cols, void* stream);
int sidlsg_gemm_fp8w(const void* A, int lda, const void* W8, const float* wscale, void* C, int ldc, const float* bias, const void* res,
                     int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha,
                     int flags, void* stream);
int sidlsg_conv3x3_fp8w(const void* X, int ldx, const void* W8, const float* wscale, void* Y, int ldc, const float* bias,
                        const void* res, int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout,
                        int stride, int ups, float alpha, int flags, void* stream);
/* Both operands e4m3 (MX MFMA v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales: twice the bf16 rate on gfx950).
 * A8: e4m3 bytes [M][lda] at unit scale -- what the GroupNorm / LayerNorm kernels write with an e4m3 output
 * (sidlsg_cast_fp8 converts a bf16 matrix: clamp to +-448, round to nearest even); W8 / wscale as above.  N % 160 == 0,
 * K % 16 == 0, lda % 16 == 0; the epilogue arguments are those of sidlsg_gemm_bf16. */
int sidlsg_cast_fp8(const void* src_bf16, void* dst_fp8, long long n, void* stream);
/* conv3x3, pad 1, stride 1 on an e4m3 NHWC image (X8: [B][H][Wd][ldx] bytes); W8: [Cout][9 * Cin] e4m3; Cin % 16 == 0,
 * Cout % 160 == 0; rowvec: the per-image row vector (time-embedding broadcast), one row per image. */
int sidlsg_conv3x3_mx8(const void* X8, int ldx, const void* W8, const float* wscale, void* Y, int ldc, const float* bias, const void* res,
                       int ldres, const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout, float alpha, int flags,
                       void* stream);
int sidlsg_layernorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y8, float* stats, int rows, int C,
                             float eps, void* stream);
int sidlsg_groupnorm_fwd_fp8(const void* x, const float* gamma, const float* beta, void* y8, float* stats, float* ws, int B,
                             int HW, int C, int G, float eps, int silu, void* stream);
int sidlsg_gemm_mx8(const void* A8, int lda, const void* W8, const float* wscale, void* C, int ldc, const float* bias, const void* res,
                    int ldres, const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha,
                    int flags, void* stream);

/* ---- fp32-accurate compute mode ------------------------------------------------------------------------------
 * The reference's default precision is fp32 (training/sid_training_loop.py:205 `dtype = float16 if use_fp16 else float32`,
 * run_sid.sh:63-88) and BASELINE.json configs[0] is fp32.  Every entry point above that touches bf16 activations also
 * exists with the suffix _f32: same arguments and semantics, but activations (A/X/C/Y/res, x/y/dy/dx, Q/K/V/O, ...) and
 * the compute copies of the weights (W, transposed W) are fp32 and the contractions run on the f32-input matrix
 * cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulation).  Selected per network by
 * HipUNet2DCondition(compute_dtype=torch.float32); used by the parity suite to assert north_star's 1e-3 bound.
 * Differences: K, lda, ldx, ldy multiples of 4; flags: only SIDLSG_SILU / SIDLSG_ACCUM (C is always fp32); the
 * weight-gradient entry points take dBias = NULL (the bias gradient is a sidlsg_colsum_f32 call); no workspace is used;
 * sidlsg_transpose_w(_batched)_f32 write fp32 destinations. */
int sidlsg_gemm_f32(const void* A, int lda, const void* W, void* C, int ldc, const float* bias, const void* res, int ldres,
                    const float* rowvec, int ld_rowvec, int rows_per_batch, int M, int N, int K, float alpha, int flags,
                    void* stream);
int sidlsg_conv3x3_f32(const void* X, int ldx, const void* W, void* Y, int ldc, const float* bias, const void* res, int ldres,
                       const float* rowvec, int ld_rowvec, int B, int H, int Wd, int Cin, int Cout, int stride, int ups,
                       float alpha, int flags, void* stream);
int sidlsg_wgrad_f32(const void* dY, int ldy, const void* A, int lda, float* dW, float* dBias, int M, int N, int K,
                     void* stream);
int sidlsg_conv3x3_wgrad_f32(const void* dY, int ldy, const void* X, int ldx, float* dW, float* dBias, int B, int H, int Wd,
                             int Cin, int Cout, int stride, int ups, void* stream);
int sidlsg_groupnorm_fwd_f32(const void* x, const float* gamma, const float* beta, void* y, float* stats, float* ws, int B,
                             int HW, int C, int G, float eps, int silu, void* stream);
int sidlsg_groupnorm_bwd_f32(const void* x, const void* dy, const float* stats, const float* gamma, const float* beta,
                             const void* dres, void* dx, float* dgamma, float* dbeta, float* ws, int B, int HW, int C, int G,
                             int silu, void* stream);
int sidlsg_layernorm_fwd_f32(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C,
                             float eps, void* stream);
int sidlsg_layernorm_bwd_f32(const void* x, const void* dy, const float* stats, const float* gamma, const void* dres, void* dx,
                             float* dgamma, float* dbeta, float* ws, int rows, int C, void* stream);
int sidlsg_attn_fwd_f32(const void* Q, const void* K, const void* V, void* O, float* LSE, int B, int H, int Nq, int Nk, int D,
                        int ldq, int ldk, int ldv, int ldo, long long bsq, long long bsk, long long bsv, long long bso,
                        void* stream);
int sidlsg_attn_bwd_f32(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, void* dQ,
                        void* dK, void* dV, float* delta, int B, int H, int Nq, int Nk, int D, int ldq, int ldk, int ldv,
                        int ldo, long long bsq, long long bsk, long long bsv, long long bso, void* stream);
int sidlsg_noisy_input_f32(const float* x0, const float* noise, const float* s0, const float* s1, void* out, float* xt, int B,
                           int C, int HW, int Cp, int dup, void* stream);
int sidlsg_noisy_input_bwd_f32(const void* g, const float* s0, float* dx0, int B, int C, int HW, int Cp, int dup,
                               int accumulate, void* stream);
int sidlsg_cfg_x0_bwd_f32(const float* g, const float* s0, const float* s1, void* deps, float* dxt, int B, int C, int HW,
                          int Cp, int dup, float kappa, int predict_x0, void* stream);
int sidlsg_timestep_embed_f32(const long long* t, void* out, int B, int dim, void* stream);
int sidlsg_silu_fwd_f32(const void* x, void* y, long long n, void* stream);
int sidlsg_silu_bwd_f32(const void* x, const void* dy, void* dx, long long n, void* stream);
int sidlsg_geglu_fwd_f32(const void* h, void* y, long long M, int F, void* stream);
int sidlsg_geglu_bwd_f32(const void* h, const void* dy, void* dh, long long M, int F, void* stream);
int sidlsg_concat2_f32(void* a, void* b, void* out, long long M, int C1, int C2, int to_parts, void* stream);
int sidlsg_sumpool2x2_f32(const void* g, void* out, int B, int H, int W, int C, void* stream);
int sidlsg_zero_insert2_f32(const void* g, void* out, int B, int Ho, int Wo, int H, int W, int C, void* stream);
int sidlsg_add_f32(const void* a, const void* b, void* o, long long n, void* stream);
int sidlsg_colsum_f32(const void* g, int ldg, float* per_batch, float* total, float* ws, int B, int rows_per_batch, int N,
                      void* stream);
int sidlsg_colsum_strided_f32(const void* g, int ldg, float* per_batch, int ld_pb, float* total, int B, int rows_per_batch, int N, void* stream);
int sidlsg_transpose_w_f32(const float* src, void* dst, int N, int K, int T, void* stream);
int sidlsg_transpose_w_batched_f32(const void* jobs, int njobs, int nblocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif
