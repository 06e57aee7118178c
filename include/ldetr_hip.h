/* layoutdetr_amd C ABI — hand-written HIP kernels for gfx950 (MI355X) behind plain pointers + sizes.
 *
 * All pointers are device pointers (HBM) unless noted; `stream` is a hipStream_t passed as void*.
 * Every function returns 0 on success or a non-zero LDETR_ERR_* code; `ldetr_last_error()` returns
 * the message (the Python host raises RuntimeError with it, mirroring TORCH_CHECK in the reference).
 * Activations are fp32.  "NHWC" means pixels are rows and channels are contiguous.
 *
 * Which reference interface each entry point replaces (paths relative to salesforce/LayoutDETR):
 *   ldetr_bias_act_f32            torch_utils/ops/bias_act.cpp:33-92  (bias_act_plugin.bias_act)
 *   ldetr_upfirdn2d_f32           torch_utils/ops/upfirdn2d.cpp:17-99 (upfirdn2d_plugin.upfirdn2d)
 *   ldetr_gemm_f32                ATen addmm/matmul behind nn.Linear: training/detr_transformer.py:187-189,
 *                                 training/networks_detr.py:50-62, training/networks_stylegan2.py:117-123
 *   ldetr_conv2d_*_f32            ATen conv2d + its autograd: training/detr_backbone.py:98-114 (torchvision
 *                                 ResNet-50), networks_detr.py:82 (input_proj), ops/conv2d_resample.py:133-135
 *   ldetr_conv_transpose2d_*_f32  ATen conv_transpose2d: torch_utils/ops/conv2d_resample.py:113-130
 *   ldetr_attention_*_f32         nn.MultiheadAttention core: training/detr_transformer.py:208-209,273-274,277-280
 *   ldetr_layernorm_*_f32         nn.LayerNorm + residual/dropout chain: training/detr_transformer.py:210-214,275-285
 *   ldetr_act_bwd_reduce_f32      bias_act backward + dx.sum(): torch_utils/ops/bias_act.py:160-173
 *   ldetr_mul_reduce_f32          `x * styles` / `x * dcoefs` backward: training/networks_stylegan2.py:66-72
 *   ldetr_torgb_bwd_f32           ToRGBLayer backward: training/networks_stylegan2.py:349-353
 *   ldetr_maxpool3x3s2_*_f32      torchvision ResNet stem max-pool (called via detr_backbone.py:105)
 *   ldetr_grad_sanitize_f32, ldetr_adam_step_f32, ldetr_ema_lerp_f32
 *                                 training/training_loop.py:303-313, 320-328
 *   ldetr_softmax_xent_*_f32, ldetr_embedding_*_f32
 *                                 LM text decoder head: training/med.py:60-61,88-94 (embeddings), 911-916 (loss)
 *   ldetr_demod_*_f32             demodulation coefficients of modulated_conv2d: training/networks_stylegan2.py:57-61
 *   ldetr_layout_losses_*_f32     compute_overlap / compute_alignment / generalized_iou_loss / mse on the generated boxes:
 *                                 metrics/metric_layoutnet.py:153-201,245-275, training/loss.py:94-97
 *   ldetr_resample_coeffs, ldetr_resize_normalize_u8
 *                                 PIL resize + normalise of the page background: training/dataset_layoutganpp.py:330-338
 *   ldetr_lsap_f64                scipy.optimize.linear_sum_assignment as used at metrics/metric_layoutnet.py:111,125,240
 *   ldetr_box_giou_pairwise_f32   box_cxcywh_to_xyxy + box_iou + generalized_box_iou: detr_util/box_ops.py:19-71
 *   ldetr_bmm_strided_f32         torch.bmm inside nn.MultiheadAttention when the regulariser phases differentiate it twice:
 *                                 training/loss.py:119-142 (path length), 207-215 (R1); hip/composite.py
 */
#ifndef LDETR_HIP_H
#define LDETR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Human-readable message of the last failing call on this thread. */
const char* ldetr_last_error(void);
/* ABI version; bumped whenever a signature changes. */
int ldetr_abi_version(void);

/* Development aid (tools/trace_tiles.py): while `buffer` (device memory, 5 int64 per block of the traced launch) is non-NULL,
 * every block of the LDS-tiled contraction kernel records wall-clock stamps (100 MHz) at entry, after its prologue, after its
 * main loop and at exit, and its (XCC_ID << 32 | HW_ID) placement word.  Pass NULL to switch it off (the default). */
int ldetr_debug_trace_tiles(int64_t* buffer);

/* Which tiles of the contraction engine run on the bf16 matrix pipe with the exact three-way operand split (fp32 operands and
 * results, fp32-equivalent accuracy: csrc/gemm_conv.hip, gemm_f32_kernel<..., SPLIT>).  Bit 0: 128x128, bit 1: 128x64, bit 2:
 * 256x32, bit 3: 64x64; 0 = f32 MFMA everywhere; -1 = back to the default / LDETR_DEBUG="SPLIT_BF16=..".  Process-wide; returns the previous override.
 * Used by the parity tests to run the same contraction on both pipes.  Non-finite values: the split of +-Inf is Inf + NaN + NaN, so a
 * tile whose accumulators come out non-finite is recomputed on the f32 pipe inside the same launch -- both settings return the same
 * Inf / NaN classes as an fp32 matmul (what training_loop.py:306-309's nan_to_num(0, 1e5, -1e5) then sees is therefore the same). */
int ldetr_set_split_bf16(int tiles);

/* Measurement aid (bench.py's roofline leg): how many contraction kernels the calling thread has launched so far on the f32 MFMA pipe
 * and on the bf16 pipe with the exact operand split; the difference across one C-ABI call tells which ceiling that call is priced against. */
int ldetr_engine_launch_counts(int64_t* f32_pipe, int64_t* bf16_split_pipe);

/* Scratch memory for the contraction engine's in-kernel split-K reduction on the calling thread's current device.
 * `ptr`: zero-filled, 16-byte aligned device memory the caller keeps alive and uses from one stream at a time
 * (the first 1 MiB holds per-tile arrival counters, the rest partial tiles; both are handed out as rings, a fresh
 * slice per launch, so that independent branches of a captured graph do not share scratch); (NULL, 0) unregisters it and the
 * engine falls back to fp32 atomics into a zero-filled C plus a second epilogue launch.  No reference counterpart:
 * the reference leaves split-K decisions to cuBLAS/cuDNN workspaces (torch.backends.cudnn.benchmark, train.py:330). */
int ldetr_set_workspace(void* ptr, int64_t bytes);

/* Strided 4-D activation view (sizes + element strides); sc == 1 selects the NHWC fast path. */
typedef struct ldetr_tensor4 {
    int N, C, H, W;
    int64_t sn, sc, sh, sw;
} ldetr_tensor4;

/* Epilogue applied to every GEMM / conv output element (row m, column n), in this order:
 *   v = acc * alpha
 *   v *= col_scale[n]                      (FrozenBatchNorm scale)
 *   v *= samp_scale[sample(m)][n]          (StyleGAN2 demodulation / style scale)
 *   v += col_bias[n]
 *   v += residual[m][n]
 *   v = act(v)                             (1: relu, 2: leaky-relu(act_alpha) * act_gain)
 *   v *= dact(mask_src[m][n])              (backward masks: 1 relu, 2 leaky-relu * act_gain)
 *   v *= dropout_keep(seed, m*ldc + n)/(1-p_drop)
 *   v *= out_scale
 *   C[m][n] = v   (or += when accumulate != 0)
 * NULL pointers / zero modes disable a step. */
typedef struct ldetr_epilogue {
    float alpha;
    const float* col_scale;
    const float* col_bias;
    const float* samp_scale;
    int64_t samp_ld;
    const float* residual;
    int64_t ldr;
    int act;
    float act_alpha;
    float act_gain;
    const float* mask_src;
    int64_t ldm;
    int mask_mode;
    float out_scale;
    float p_drop;
    uint64_t seed;
    const uint64_t* seed_ptr; /* optional device word added to `seed` at run time (keeps dropout fresh under hipGraph replay) */
    int accumulate;
    /* ldetr_gemm_f32 with ta == 1 only: a_rowsum[m] += sum_k op(A)[m, k] (no alpha) as a by-product of the contraction — the
     * bias gradient of a linear layer falls out of its weight-gradient GEMM dW = dY^T X (A = dY stored [K, M], lda == M).
     * The latency-bound kernel sums it from the A fragments it streams anyway; other sizes run the column-sum kernel. */
    float* a_rowsum;
} ldetr_epilogue;

int ldetr_bias_act_f32(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* y,
                       int64_t sizeX, int sizeB, int64_t stepB, int grad, int act, float alpha, float gain, float clamp,
                       void* stream);

int ldetr_upfirdn2d_f32(const float* x, const float* f, float* y, int N, int C, int inH, int inW,
                        const int64_t* x_strides_nchw, int fh, int fw, int64_t f_stride_h, int64_t f_stride_w,
                        int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip,
                        float gain, int outH, int outW, const int64_t* y_strides_nchw,
                        const float* act_bias, int has_act, float act_alpha, float act_gain, void* stream);

/* splitk: 0 = the launch policy decides (latency-bound sizes take a register-streaming 32x32 kernel; otherwise the
 * reduction may be split over grid.z, reduced in-kernel through the ldetr_set_workspace scratch or, without one, by
 * fp32 atomics into a zero-filled C plus an epilogue pass), 1 = never split, > 1 = explicit number of K slices. */
int ldetr_gemm_f32(const float* A, int64_t lda, int ta, const float* B, int64_t ldb, int tb, float* C, int64_t ldc,
                   int M, int N, int K, int splitk, const ldetr_epilogue* ep, int pix_per_sample, void* stream);

/* Two ldetr_gemm_f32 problems in one call, executed in order g0 then g1 (they may read the same operands; their outputs must not
 * overlap each other's inputs).  With g0 = (ta 0, tb 1), g1 = (ta 1, tb 1) -- the data and the weight gradient of a linear layer -- or
 * both (ta 1, tb 1) -- the two weight gradients of the fused feed-forward block -- and both in the small-tile class, they run as ONE
 * kernel launch; otherwise as two.  Same results either way. */
typedef struct ldetr_gemm_desc {
    const float* A; int64_t lda; int ta;
    const float* B; int64_t ldb; int tb;
    float* C; int64_t ldc;
    int M, N, K, splitk;
    const ldetr_epilogue* ep;
    int pix_per_sample;
} ldetr_gemm_desc;
int ldetr_gemm_pair_f32(const ldetr_gemm_desc* g0, const ldetr_gemm_desc* g1, void* stream);
/* 1 if the pair would run as one launch, else 0. */
int ldetr_gemm_pair_is_single_launch(const ldetr_gemm_desc* g0, const ldetr_gemm_desc* g1);

int ldetr_conv2d_fwd_f32(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW, int stride,
                         int pad, float* y, int64_t ldy, int OH, int OW, const float* in_scale, int64_t in_scale_ld,
                         const ldetr_epilogue* ep, void* stream);
int ldetr_conv2d_bwd_data_f32(const float* dy, const ldetr_tensor4* dyt, const float* w, int Cin, int KH, int KW,
                              int stride, int pad, float* dx, int64_t lddx, int IH, int IW, const float* dy_scale,
                              int64_t dy_scale_ld, const ldetr_epilogue* ep, void* stream);
/* accumulate != 0: dw += gradient (fp32 atomics onto the existing buffer, e.g. a view of the flat .grad buffer) instead of dw = gradient.
 * splitk: 0 = the library picks the tile shape and the number of pixel slices together; >= 1 explicit. */
int ldetr_conv2d_bwd_weight_f32(const float* x, const ldetr_tensor4* xt, const float* dy, const ldetr_tensor4* dyt,
                                float* dw, int KH, int KW, int stride, int pad, int splitk, const float* x_scale,
                                int64_t x_scale_ld, const float* dy_scale, int64_t dy_scale_ld, int accumulate, void* stream);

int ldetr_conv_transpose2d_fwd_f32(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW,
                                   int stride, int pad, float* y, int64_t ldy, int OH, int OW, const float* in_scale,
                                   int64_t in_scale_ld, const ldetr_epilogue* ep, void* stream);
int ldetr_conv_transpose2d_bwd_data_f32(const float* dy, const ldetr_tensor4* dyt, const float* w, int Cin, int KH,
                                        int KW, int stride, int pad, float* dx, int64_t lddx, int IH, int IW,
                                        const float* dy_scale, int64_t dy_scale_ld, const ldetr_epilogue* ep,
                                        void* stream);
int ldetr_conv_transpose2d_bwd_weight_f32(const float* x, const ldetr_tensor4* xt, const float* dy,
                                          const ldetr_tensor4* dyt, float* dw, int KH, int KW, int stride, int pad,
                                          int splitk, const float* x_scale, int64_t x_scale_ld, const float* dy_scale,
                                          int64_t dy_scale_ld, int accumulate, void* stream);

/* head_dim: multiple of 32 up to 192.  causal != 0 (self-attention only, Lq == Lk): key j is visible to query i iff j <= i
 * (BertSelfAttention of an is_decoder config, training/med.py:704-739), on top of the key-padding mask.
 * Lk <= 256: the score row stays in registers; 256 < Lk <= 16384: 256-key chunks with an online softmax (same results). */
int ldetr_attention_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const unsigned char* key_padding_mask, float* out, int64_t ldo, float* lse, int B, int H,
                            int Lq, int Lk, int head_dim, float scale, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                            int causal, void* stream);
int ldetr_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                            const unsigned char* key_padding_mask, const float* out, int64_t ldo, const float* lse,
                            const float* dout, int64_t lddo, float* dq, int64_t lddq, float* dk, int64_t lddk,
                            float* dv, int64_t lddv, int B, int H, int Lq, int Lk, int head_dim, float scale,
                            float p_drop, uint64_t seed, const uint64_t* seed_ptr, int causal, void* stream);

/* ypos (optional, with pos [pos_rows, D]): second output ypos[row] = y[row] + pos[row % pos_rows] — the position-embedded copy the
 * next attention block projects q and k from (training/detr_transformer.py:207,277 form it with one extra kernel per layer);
 * dy2 (optional): gradient that arrived through ypos, summed into dy as it is loaded. */
int ldetr_layernorm_fwd_pos_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                                float* z, float* mean, float* rstd, int64_t rows, int D, float eps, float p_drop,
                                uint64_t seed, const uint64_t* seed_ptr, const float* pos, int64_t pos_rows, float* ypos,
                                void* stream);

/* The same with the residual branch given as `n_parts` partial sums parts[s][rows][D] (pitch part_stride floats between slices) plus a
 * column bias: r = part_bias + sum_s parts[s], added in slice order (deterministic).  This is where the hidden-slice contributions of the
 * fused feed-forward block (ldetr_ffn_fwd_f32) are reduced.  n_parts = 0: `parts` is the plain residual (== ldetr_layernorm_fwd_pos_f32). */
int ldetr_layernorm_fwd_parts_f32(const float* x, const float* parts, int n_parts, int64_t part_stride, const float* part_bias,
                                  const float* gamma, const float* beta, float* y, float* z, float* mean, float* rstd,
                                  int64_t rows, int D, float eps, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                  const float* pos, int64_t pos_rows, float* ypos, void* stream);

/* Self-attention sub-block of the short-sequence stacks (nn.MultiheadAttention(256, 8) with q = k = v = x, L <= 16 tokens per sample:
 * training/detr_transformer.py:273-274; nn.TransformerEncoderLayer at training/util.py:21-26, networks_detr.py:243,269,275) as ONE
 * forward launch (csrc/mha_small.hip): packed projection, masked softmax attention with dropout on the probabilities, and the output
 * projection's per-head contributions.
 *   x [B*L][ldx], w_in [768][256], b_in [768], w_out [256][256], kpm [B][L] (nonzero = masked key) or NULL
 *   -> qkv [B*L][768] (projection incl. bias, q unscaled), o [B*L][256] (attention output), lse [B][8][L]: what
 *      ldetr_attention_bwd_f32 reads (same dropout element index, same seed);
 *   -> ypart [8][B*L][256]: out_proj contributions per head WITHOUT its bias (reduce with ldetr_layernorm_fwd_parts_f32). */
int ldetr_mha_small_fwd_f32(const float* x, int64_t ldx, const float* w_in, const float* b_in, const float* w_out,
                            const uint8_t* kpm, float* qkv, float* o, float* lse, float* ypart,
                            int B, int L, int D, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                            void* stream);

/* Cross-attention sub-block of a decoder layer (training/detr_transformer.py:277-280) with <= 16 queries per sample onto <= 64 memory
 * tokens whose K / V projections already exist (k, v: [B*Lk][ldk / ldv], head h in columns 32 h .. 32 h + 31): query projection
 * (w_q, b_q = rows 0..255 of in_proj_weight / in_proj_bias), attention, per-head output projection in one launch.
 *   -> q [B*Lq][256] (projected queries incl. bias, unscaled), o [B*Lq][256], lse [B][8][Lq]: what ldetr_attention_bwd_f32 reads;
 *   -> ypart [8][B*Lq][256] as in ldetr_mha_small_fwd_f32. */
int ldetr_mha_cross_fwd_f32(const float* x, int64_t ldx, const float* w_q, const float* b_q,
                            const float* k, int64_t ldk, const float* v, int64_t ldv, const float* w_out,
                            const uint8_t* kpm, float* q, float* o, float* lse, float* ypart,
                            int B, int Lq, int Lk, int D, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                            void* stream);

/* Position-wise feed-forward block linear2(dropout(relu(linear1(x)))) of the DETR layers (training/detr_transformer.py:212-214, 283-285;
 * d_model D = 256, hidden width F a multiple of 64), one launch per direction (csrc/ffn_fused.hip).
 * fwd: x [M][ldx], w1 [F][D], b1 [F], w2 [D][F] -> h [M][F] (hidden after relu + dropout, kept for the backward) and
 *      ypart [F/64][M][D]: per-hidden-slice contributions to the output WITHOUT b2 (reduce with ldetr_layernorm_fwd_parts_f32).
 * bwd: dy [M][D] (gradient of the block output) -> dxpart [F/64][M][D]: per-hidden-slice contributions to the input gradient (reduce with
 *      ldetr_layernorm_bwd_parts_f32) and, if dh is not NULL, dh [M][F] = gradient of the hidden pre-activation (relu and dropout masks
 *      applied).  The weight gradients are plain contractions over the tokens: dW2 += dy^T h, dW1 += dh^T x, db2 / db1 = their row sums
 *      (one ldetr_gemm_pair_f32 call).  p_drop / seed as in ldetr_gemm_f32's epilogue (element index = row * F + column of h). */
int ldetr_ffn_fwd_f32(const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2, float* h, float* ypart,
                      int64_t M, int D, int F, float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream);
int ldetr_ffn_bwd_f32(const float* dy, const float* x, int64_t ldx, const float* h, const float* w1, const float* w2,
                      float* dxpart, float* dh, int64_t M, int D, int F, float p_drop, void* stream);
int ldetr_layernorm_bwd2_f32(const float* dy, const float* dy2, const float* z, const float* mean, const float* rstd,
                             const float* gamma, float* dx, float* dresidual, float* dgamma, float* dbeta, int64_t rows,
                             int D, float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream);
/* ... with the incoming gradient given as dy (+ dy2) + sum_s dy_parts[s][rows][D] (pitch part_stride floats), added in slice order: where the
 * hidden-slice contributions of ldetr_ffn_bwd_f32 to the feed-forward block's input gradient are reduced. */
int ldetr_layernorm_bwd_parts_f32(const float* dy, const float* dy2, const float* dy_parts, int n_parts, int64_t part_stride,
                                  const float* z, const float* mean, const float* rstd, const float* gamma,
                                  float* dx, float* dresidual, float* dgamma, float* dbeta, int64_t rows, int D,
                                  float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream);
int ldetr_layernorm_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta, float* y,
                            float* z, float* mean, float* rstd, int64_t rows, int D, float eps, float p_drop,
                            uint64_t seed, const uint64_t* seed_ptr, void* stream);
int ldetr_layernorm_bwd_f32(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                            float* dx, float* dresidual, float* dgamma, float* dbeta, int64_t rows, int D,
                            float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream);

/* Second stage of ldetr_torgb_bwd_f32 (ToRGBLayer backward, training/networks_stylegan2.py:349-353): dws [B][3][C] ->
 * dw[o][c] += sum_b dws[b][o][c] s[b][c] (accumulated) and ds[b][c] = sum_o dws[b][o][c] w[o][c]. */
int ldetr_torgb_bwd_finish_f32(const float* dws, const float* s, const float* w, int B, int C, float* dw, float* ds, void* stream);

/* The tail of a loss phase (StyleGAN2Loss.accumulate_gradients, training/loss.py:84-116, 146-218: sum of ~10 weighted terms, .mean(), backward) as one
 * launch per direction: total = sum_k w[k] * c_k * sum_i f_k(x[k][i]), c_k = 1 / n[k] (red[k] = 0: a per-sample term that is averaged) or 1 (red[k] = 1:
 * per-sample contributions to a sum); fn[k]: 0 identity, 1 softplus(x), 2 softplus(-x) (F.softplus: threshold 20), 3 ratio x[0] / x[1] (n = 2: the
 * (loss sum, count) pair of ldetr_softmax_xent_fwd_f32; its gradient is passed UNdivided, ldetr_softmax_xent_bwd_f32 divides by the count).
 * x / w / n / fn / red are HOST arrays of K <= 16 entries (x[k]: device pointers); vals [K][ld] (weighted values, for the reported statistics),
 * sums [K], total [1]; backward: g [1] -> grads [K][ld]. */
int ldetr_loss_combine_fwd_f32(const float* const* x, const float* w, const int* n, const int* fn, const int* red, int K, int ld,
                               float* vals, float* sums, float* total, void* stream);
int ldetr_loss_combine_bwd_f32(const float* const* x, const float* w, const int* n, const int* fn, const int* red, int K, int ld,
                               const float* g, float* grads, void* stream);
/* F.mse_loss(a[valid], b[valid]) of the static-shape heads without the gather: a [rows][D], b [rows / bdiv][D] (one reference row per bdiv rows),
 * valid [rows] (nonzero = counted) -> out2 = {sum |a - b|^2 / (max(count, 1) * D), count}; backward: da = 2 (a - b) g / (count * D) on valid rows. */
int ldetr_masked_mse_fwd_f32(const float* a, const float* b, const uint8_t* valid, int64_t rows, int D, int bdiv, float* out2, void* stream);
int ldetr_masked_mse_bwd_f32(const float* a, const float* b, const uint8_t* valid, int64_t rows, int D, int bdiv, const float* out2,
                             const float* g, float* da, void* stream);

/* ---- Group launches of the short token stacks (round 5).
 * D's conditional and unconditional reconstruction decoders (networks_detr.py:269, 275-276 via training/util.py:13-43), and D's layout decoder
 * beside its unconditional encoder (networks_detr.py:242-243), are structurally identical, independent stacks that the reference runs one after the
 * other; every launch of such a stack is a fraction of a wave of work per CU.  The entry points below take ONE or TWO argument blocks: the second
 * problem's blocks follow the first's in the same grid (the ldetr_p3_conv2d_fwd_dual idea), so two stacks advance with one launch per sub-block.
 * The argument blocks are plain structs of device pointers and sizes; the single-problem entry points above are these with n = 1. */
typedef struct ldetr_ln_args {
    const float* x; const float* r; const float* gamma; const float* beta;       /* forward: y = LN(x + dropout(r)); r may be NULL */
    float* y; float* z; float* mean; float* rstd;                                 /* z = pre-norm sum (kept for the backward), row statistics */
    const float* dy; float* dx; float* dr; float* dgamma; float* dbeta;            /* backward: dx = dz, dr = dz * dropout mask (optional); dgamma / dbeta += (atomics) */
    int64_t rows; int D; float eps, p_drop; uint64_t seed; const uint64_t* seed_ptr;
    const float* pos; int64_t pos_rows; float* ypos; const float* dy2;             /* second output y + pos[row % pos_rows] and its gradient */
    int r_parts; int64_t r_part_stride; const float* r_bias;                      /* forward: r = r_bias + sum of r_parts slices r[s][rows][D] */
    const float* dy_parts; int dy_nparts; int64_t dy_part_stride;                 /* backward: incoming gradient = dy (+ dy2) + sum of dy_nparts slices */
} ldetr_ln_args;
int ldetr_layernorm_fwd_group_f32(const ldetr_ln_args* a, int n, void* stream);
int ldetr_layernorm_bwd_group_f32(const ldetr_ln_args* a, int n, void* stream);
/* out = base (or 0 when NULL) + sum_s parts[s], in slice order; n elements (multiple of 4), 16-byte aligned buffers. */
int ldetr_sum_parts_f32(const float* base, const float* parts, int n_parts, int64_t part_stride, float* out, int64_t n, void* stream);

typedef struct ldetr_ffn_args {
    const float* x; int64_t ldx;              /* [M][ldx] block input */
    const float* w1; const float* b1;         /* [F][256], [F] */
    const float* w2;                          /* [256][F] */
    float* h;                                 /* [M][F] hidden after relu (+ dropout): written by the forward, read by the backward */
    float* ypart;                             /* forward out: [F/64][M][256] */
    int M, F;
    float p_drop; uint64_t seed; const uint64_t* seed_ptr;
    const float* dy;                          /* backward: [M][256] gradient of the block output */
    float* dxpart;                            /* backward out: [F/64][M][256] */
    float* dh;                                /* backward out (optional): [M][F] gradient of the hidden pre-activation */
} ldetr_ffn_args;
int ldetr_ffn_fwd_group_f32(const ldetr_ffn_args* a, int n, void* stream);
int ldetr_ffn_bwd_group_f32(const ldetr_ffn_args* a, int n, void* stream);

/* Self-attention sub-block (ldetr_mha_small_fwd_f32's arguments) and its BACKWARD as one launch: per (sample, head) block
 *   dO_h = dr_b W_out[:, 32h : 32h+32]; attention backward on the saved projection (dropout mask regenerated from the seed); the head's packed
 *   gradient dqkv [B*L][768] (operand of the weight gradient dW_in += dqkv^T x, written once); and the head's contribution to the input gradient
 *   dqkv_h W_in,h -> dxpart[h][B*L][256] (reduce with ldetr_layernorm_bwd_group_f32's dy_parts or ldetr_sum_parts_f32).
 * Replaces {out_proj data gradient, ldetr_attention_bwd_f32, in_proj data gradient} = three launches of the unfused path. */
typedef struct ldetr_mha_small_args {
    const float* x; int64_t ldx;
    const float* w_in; const float* b_in; const float* w_out;
    const unsigned char* kpm;
    float* qkv; float* o; float* lse; float* ypart;       /* forward outputs; the backward reads qkv, o, lse */
    int B, L;
    float scale, p_drop; uint64_t seed; const uint64_t* seed_ptr;
    const float* dr;                                       /* backward: [B*L][256] gradient of the sub-block output (before out_proj's bias) */
    float* dqkv; float* dxpart;                            /* backward outputs */
} ldetr_mha_small_args;
int ldetr_mha_small_fwd_group_f32(const ldetr_mha_small_args* a, int n, void* stream);
int ldetr_mha_small_bwd_group_f32(const ldetr_mha_small_args* a, int n, void* stream);

/* Cross-attention sub-block backward (forward: ldetr_mha_cross_fwd_f32) as one launch per (sample, head): dO_h = dr_b W_out[:, head]; attention backward
 * on the saved q and the projected memory K / V (dK / dV written to dk / dv with pitches lddk / lddv: the grouped projection's gradient buffer);
 * dq [B*Lq][256]; dxpart[h][B*Lq][256] = dq_h W_q,h. */
typedef struct ldetr_mha_cross_args {
    const float* x; int64_t ldx;
    const float* w_q; const float* b_q;
    const float* k; int64_t ldk; const float* v; int64_t ldv;
    const float* w_out;
    const unsigned char* kpm;
    float* q; float* o; float* lse; float* ypart;
    int B, Lq, Lk;
    float scale, p_drop; uint64_t seed; const uint64_t* seed_ptr;
    const float* dr;
    float* dq; float* dk; int64_t lddk; float* dv; int64_t lddv; float* dxpart;
} ldetr_mha_cross_args;
int ldetr_mha_cross_bwd_f32(const ldetr_mha_cross_args* a, void* stream);

/* Up to 8 weight gradients dW[rows][cols] += A^T B over the tokens (A [M][lda] holds the `rows` output features in columns, B [M][ldb] the `cols`
 * input features; both row-major over tokens), bias gradient db[rows] += column sums of A (optional), as ONE launch: the weight gradients of a
 * transformer layer (feed-forward W1 / W2, attention in_proj / out_proj) are independent contractions with K = tokens. */
typedef struct ldetr_wgrad_desc {
    const float* A; int64_t lda; const float* B; int64_t ldb;
    float* dW; int64_t ldw; float* db;
    int M, rows, cols;
} ldetr_wgrad_desc;
int ldetr_wgrad_multi_f32(const ldetr_wgrad_desc* d, int n, void* stream);

int ldetr_colsum_f32(const float* a, float* red, int B, int64_t P, int C, void* stream);
int ldetr_act_bwd_reduce_f32(const float* dy, const float* y, float* dv, const float* bias, const float* demod,
                             float* dbias, float* ddemod, int B, int64_t P, int C, int act, float alpha, float gain,
                             void* stream);
int ldetr_mul_reduce_f32(const float* a, const float* x, const float* scale, float* out, float* red, int B, int64_t P,
                         int C, void* stream);
/* ToRGBLayer forward (training/networks_stylegan2.py:349-353): y[b][p][o] = bias[o] + sum_c x[b][p][c] w[o][c] styles[b][c], 3 colour channels;
 * x [B][P][C] NHWC pixels, C = 4 x a power of two <= 512. */
int ldetr_torgb_fwd_f32(const float* x, const float* w, const float* styles, const float* bias, float* y,
                        int B, int64_t P, int C, void* stream);
int ldetr_torgb_bwd_f32(const float* x, const float* dy, const float* w, const float* styles, float* dx, float* dws,
                        float* dbias, int B, int64_t P, int C, void* stream);

int ldetr_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, void* stream);
int ldetr_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C,
                               void* stream);

int ldetr_grad_sanitize_f32(float* g, int64_t n, float scale, float nan_value, float posinf, float neginf,
                            void* stream);
int ldetr_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1,
                        float beta2, float eps, int fuse_sanitize, float gscale, float nan_value, float posinf,
                        float neginf, void* stream);
/* The same pass with the parameters' exponential moving average updated alongside (training_loop.py:320-328):
 * p_ema = p_new + ema_beta * (p_ema - p_new); p_ema == NULL: plain Adam. */
int ldetr_adam_ema_step_f32(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, float lr, float beta1, float beta2,
                            float eps, int fuse_sanitize, float gscale, float nan_value, float posinf, float neginf,
                            float* p_ema, float ema_beta, void* stream);
int ldetr_ema_lerp_f32(float* p_ema, const float* p, int64_t n, float beta, void* stream);

/* Label-smoothed softmax cross entropy of the LM text decoder (CrossEntropyLoss(reduction='mean', label_smoothing) on the shifted
 * prediction scores, training/med.py:911-916; ignore_index marks padded tokens).  fwd: one pass over logits [rows, V] (row pitch
 * ld): row_lse[i] = logsumexp, *loss_sum += sum_i loss_i, *count += #(targets != ignore_index) (both zeroed by the caller; the mean
 * is loss_sum / count).  bwd: dlogits[i, c] = (softmax - (1 - eps) onehot - eps / V) * (*grad_out) / (*count), zero rows for
 * ignored targets; dlogits may alias logits. */
int ldetr_softmax_xent_fwd_f32(const float* logits, int64_t ld, const int64_t* targets, float* row_lse, float* loss_sum, float* count,
                               int64_t rows, int V, int64_t ignore_index, float label_smoothing, void* stream);
int ldetr_softmax_xent_bwd_f32(const float* logits, int64_t ld, const int64_t* targets, const float* row_lse, const float* count,
                               const float* grad_out, float* dlogits, int64_t ldd, int64_t rows, int V, int64_t ignore_index,
                               float label_smoothing, void* stream);

/* Token embedding of the LM text decoder (nn.Embedding(vocab, hidden, padding_idx=pad) + the position rows added right after it,
 * training/med.py:60-61,88-94).  fwd: out[i, :] = weight[ids[i], :] (+ pos[i % T, :] when pos != NULL); ids outside [0, V) give
 * a zero row.  bwd: dweight[ids[i], :] += dy[i, :] (fp32 atomics into a caller-zeroed or accumulating buffer); rows with
 * ids[i] == padding_idx or outside [0, V) contribute nothing.  d % 4 == 0 and 16-byte aligned rows for fwd. */
int ldetr_embedding_fwd_f32(const float* weight, const float* pos, const int64_t* ids, float* out, int64_t n, int d, int V, int T,
                            void* stream);
int ldetr_embedding_bwd_f32(const float* dy, const int64_t* ids, float* dweight, int64_t n, int d, int V, int64_t padding_idx,
                            void* stream);

/* Page-background preprocessing of a dataset item (training/dataset_layoutganpp.py:330-338): Pillow's 8-bit Lanczos ("ANTIALIAS")
 * resize of a decoded uint8 RGB page, then (x / 255 - mean) / std in fp32, CHW.  Bit-identical to Pillow + numpy.
 * ldetr_resample_coeffs (HOST memory, no GPU work): window bounds [out][2] = (first input index, tap count) and 22-bit fixed-point
 * weights, stored tap-major [ksize][out]; call with bounds = weights = NULL to query ksize.
 * ldetr_resize_normalize_u8: src [images][H][W][3] uint8 -> tmp [images][H][out_w][3] uint8 (scratch, horizontal pass) ->
 * out_u8 [images][out_h][out_w][3] (optional) and out_chw [images][3][out_h][out_w] fp32 (optional).  hbounds/hweights are
 * ldetr_resample_coeffs(W, out_w) and vbounds/vweights ldetr_resample_coeffs(H, out_h), copied to device memory. */
int ldetr_resample_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* weights, int64_t weights_capacity, int* ksize_out);
int ldetr_resize_normalize_u8(const uint8_t* src, int64_t images, int H, int W, int out_h, int out_w, const int32_t* hbounds,
                              const int32_t* hweights, int hksize, const int32_t* vbounds, const int32_t* vweights, int vksize,
                              uint8_t* tmp, uint8_t* out_u8, float* out_chw, float mean0, float mean1, float mean2, float std0,
                              float std1, float std2, void* stream);

/* Demodulation coefficients of the modulated convolution (training/networks_stylegan2.py:57-61), forward and backward:
 * dcoefs[b][o] = rsqrt(sum_{i,kh,kw} (weight[o][i][kh][kw] * styles[b][i])^2 + eps).  weight is addressed through its element
 * strides (so, si, sh, sw): OIHW or channels_last memory.  fwd also writes w2[o][i] = sum_taps weight^2 for the backward.
 * bwd: dweight (same strides as weight; += when accumulate_dweight != 0; may be NULL) and dstyles [B][I] (may be NULL) from
 * grad_dcoefs [B][O].  B <= 64. */
int ldetr_demod_fwd_f32(const float* weight, int64_t so, int64_t si, int64_t sh, int64_t sw, const float* styles, float* dcoefs,
                        float* w2, int B, int O, int I, int KH, int KW, float eps, void* stream);
int ldetr_demod_bwd_f32(const float* weight, int64_t so, int64_t si, int64_t sh, int64_t sw, const float* styles, const float* dcoefs,
                        const float* w2, const float* grad_dcoefs, float* dweight, int accumulate_dweight, float* dstyles, int B, int O,
                        int I, int KH, int KW, void* stream);

/* The generator phase's four layout losses on one set of generated boxes, fused with their gradients (training/loss.py:94,
 * metrics/metric_layoutnet.py:153-201,245-275): bbox, bbox_ref [B][N][4] (xc, yc, w, h), valid [B][N] (non-zero = real element),
 * N <= 64.  losses [4][B]: per-sample shares of (0) mse_loss(bbox[valid], bbox_ref[valid]), (1) generalized_iou_loss(same),
 * (2) compute_overlap(bbox, valid)[b], (3) compute_alignment(bbox, valid)[b] -- rows 0 and 1 sum to the reference's scalars.
 * grads [4][B][N][4] = d losses[t][b] / d bbox[b] (autograd's subgradient conventions).  bbox_ref may be NULL (rows 0, 1 = 0).
 * bwd: dbbox[b] = sum_t grad_losses[t][b] * grads[t][b]. */
int ldetr_layout_losses_f32(const float* bbox, const float* bbox_ref, const uint8_t* valid, int B, int N, float* losses,
                            float* grads, void* stream);
int ldetr_layout_losses_bwd_f32(const float* grads, const float* grad_losses, int B, int N, float* dbbox, void* stream);

/* Batched linear-sum-assignment (Hungarian / shortest augmenting path) on device.
 * cost: [batch][n][n] float64 row-major; maximize != 0 negates the costs first;
 * row_ind / col_ind: [batch][n] int32 outputs with scipy's ordering (row_ind sorted ascending). n <= 64. */
int ldetr_lsap_f64(const double* cost, int batch, int n, int maximize, int* row_ind, int* col_ind, void* stream);

/* Pairwise box IoU / union / generalised IoU of detr_util/box_ops.py (box_iou :35-48, generalized_box_iou :51-71; with cxcywh != 0
 * the boxes go through box_cxcywh_to_xyxy :19-23 first), batched: boxes1 [B][N][4], boxes2 [B][M][4] fp32 (16-byte aligned rows)
 * -> iou / uni / giou [B][N][M] fp32 (any may be NULL) and, if `cost` is not NULL, cost[b][i][j] = cost_sign * giou as float64: the
 * matrix ldetr_lsap_f64 takes (the DETR matcher's cost_giou = -generalized_box_iou).  fp32 arithmetic in the reference's operation
 * order without fma contraction: bit-identical to the reference's CPU values.  Degenerate pairs (union or enclosing area 0) give
 * the reference's Inf / NaN; the reference's `assert x1 >= x0` is the caller's (layoutdetr_amd/detr_util/box_ops.py). */
int ldetr_box_giou_pairwise_f32(const float* boxes1, const float* boxes2, int B, int N, int M, int cxcywh, float* iou, float* uni,
                                float* giou, double* cost, double cost_sign, void* stream);

/* Batched small matrix product on strided views: C[b1][b2][m][n] = alpha * sum_k A[b1][b2][m][k] * B[b1][b2][k][n]; sa / sb / sc hold the
 * (b1, b2, row, column) ELEMENT strides of each operand, so transposed operands and head-split views of a [rows][heads * dh] activation need
 * no copy.  fp32 FMAs in k order (deterministic).  For the regulariser phases only (second-order autograd through the attention products,
 * training/loss.py:119-142, 207-215: every product's backward is this product on transposed views); the hot path's attention is
 * ldetr_attention_* / ldetr_mha_*.  nb1, nb2 <= 65535. */
int ldetr_bmm_strided_f32(const float* A, const int64_t* sa, const float* B, const int64_t* sb, float* C, const int64_t* sc,
                          int nb1, int nb2, int M, int N, int K, float alpha, void* stream);

/* ---- Plane-format ("P3") convolutions of the ResNet-50 trunk (csrc/p3_engine.hip; ATen conv2d + its autograd behind
 * torchvision's resnet50 at training/detr_backbone.py:98-114).
 * P3 storage of an fp32 tensor [rows][C], C % 8 == 0: the exact three-way bf16 split x = hi + mid + lo in 48-byte groups of
 * 8 channels, [rows][C/8][3][8 x bf16] (6 bytes per element; `void*` below).  The convolutions contract such operands on the bf16
 * matrix pipe with fp32 accumulation: fp32-equivalent results (the same six-product scheme as ldetr_set_split_bf16's tiles).
 * Every fp32 value converts exactly (bit for bit through split -> merge), incl. +-0, +-Inf, NaN and values next to FLT_MAX: a leading part that
 * would round up to Inf is truncated instead, and a non-finite element is stored as (x, 0, 0).  The six-product contraction itself is exact for
 * finite operands only, so every kernel checks its accumulators after the k-loop and a wave that finds a non-finite value recomputes its tile with
 * fp32 FMAs on the merged planes: outputs are NaN / +Inf / -Inf exactly where an fp32 convolution puts them (training_loop.py:308 maps the classes
 * to different gradients).
 * Determinism: forward and data gradient are bit-reproducible run to run (split-K partials are summed in slice order by the last-arriving block);
 * the weight gradient (ldetr_p3_conv2d_bwd_weight, and the dw half of ldetr_p3_conv2d_bwd_pair) accumulates its pixel slices with fp32 atomics onto
 * the caller's buffer, so its low-order bits depend on the arrival order of the slices. */
int ldetr_p3_split_f32(const float* src, int64_t ld, void* dst, int64_t rows, int C, void* stream);
int ldetr_p3_merge_f32(const void* src, float* dst, int64_t ld, int64_t rows, int C, void* stream);
/* w [O][KH][KW][I] fp32 (times o_scale[o] when not NULL: FrozenBN's factor of the output gradient) -> P3 [I][KH*KW][O], taps in the
 * original order: the B operand of ldetr_p3_conv2d_bwd_data. */
int ldetr_p3_weight_bwd(const float* w_ohwi, const float* o_scale, void* dst, int O, int KH, int KW, int I, void* stream);

/* Every conv weight of a module in one launch: table_dev (device memory) holds nconv rows of 8 int64 = {w fp32 [O][T][I] pointer,
 * o_scale pointer or 0, dst_fwd pointer or 0 (P3 of w as stored), dst_bwd pointer or 0 (what ldetr_p3_weight_bwd writes), O, T, I,
 * index of the row's first block}; a row owns ceil(O*T*I/8 / 256) consecutive blocks, total_blocks = their sum.  O % 8 == 0, I % 8 == 0. */
int ldetr_p3_weight_prep(const int64_t* table_dev, int nconv, int total_blocks, void* stream);

/* v = acc * alpha * col_scale[n] + col_bias[n] + residual[m][n];  relu;  v = relu_mask[m][n] > 0 ? v : 0;  store as P3 and / or fp32. */
typedef struct ldetr_p3_epilogue {
    float alpha;
    const float* col_scale;
    const float* col_bias;
    const void* residual_p3;
    const float* residual_f32;
    const void* relu_mask_p3;
    int relu;
} ldetr_p3_epilogue;

/* y[n][oy][ox][co] = sum x[n][oy*stride - pad + kh][ox*stride - pad + kw][ci] * w[co][kh][kw][ci]; x P3 [N][H][W][Cin] (Cin % 32 == 0),
 * w P3 [Cout][KH][KW][Cin] (Cout % 8 == 0), KH*KW <= 32, stride 1 or 2. */
int ldetr_p3_conv2d_fwd(const void* x, int N, int H, int W, int Cin, const void* w, int Cout, int KH, int KW, int stride, int pad,
                        const ldetr_p3_epilogue* ep, void* out_p3, float* out_f32, void* stream);
/* ldetr_p3_conv2d_fwd on two (activations, weights, epilogue, output) sets of the same geometry as ONE launch (the second set is the z = 1 half of the
 * grid): G's and D's trunks convolve the same backgrounds with different weights, ATen runs them as two convolutions. */
int ldetr_p3_conv2d_fwd_dual(const void* x1, const void* x2, int N, int H, int W, int Cin, const void* w1, const void* w2, int Cout, int KH, int KW,
                             int stride, int pad, const ldetr_p3_epilogue* ep1, const ldetr_p3_epilogue* ep2, void* out1_p3, float* out1_f32,
                             void* out2_p3, float* out2_f32, void* stream);
/* dx[n][iy][ix][ci] = sum_{kh,kw,co} dy[n][(iy + pad - kh) / stride][(ix + pad - kw) / stride][co] * wb[ci][kh][kw][co] over the taps whose
 * quotients are exact and in range; dy P3 [N][OH][OW][Cout] (Cout % 32 == 0), wb from ldetr_p3_weight_bwd, dx [N][IH][IW][Cin].  The epilogue's
 * [m][n] operands (residual, mask) are indexed like dx. */
int ldetr_p3_conv2d_bwd_data(const void* dy, int N, int OH, int OW, int Cout, const void* wb, int Cin, int KH, int KW, int stride, int pad,
                             int IH, int IW, const ldetr_p3_epilogue* ep, void* out_p3, float* out_f32, void* stream);
/* dw[co][kh][kw][ci] += dy_scale[co] * sum_pixels dy[n][oy][ox][co] * x[n][oy*stride - pad + kh][ox*stride - pad + kw][ci]  (fp32 atomics
 * onto the existing buffer, e.g. a view of the flat .grad buffer); x P3 [N][H][W][Cin], dy P3 [N][OH][OW][Cout], both channel counts % 32 == 0. */
int ldetr_p3_conv2d_bwd_weight(const void* x, int N, int H, int W, int Cin, const void* dy, int Cout, int KH, int KW, int stride, int pad,
                               const float* dy_scale, float* dw, void* stream);
/* Both gradients of one convolution -- ldetr_p3_conv2d_bwd_data(dy, ..., ep, dx_p3, dx_f32) and ldetr_p3_conv2d_bwd_weight(x, ..., dy_scale, dw),
 * the two halves of ATen's convolution_backward for one layer -- as ONE kernel launch when the two grids can share a launch (the two are
 * independent and each alone leaves the chip idle through its fill and its store burst; parallel branches of a captured hipGraph do not
 * overlap, one grid does), else as the two launches.  x is the layer's input [N][IH][IW][Cin]; *launches (optional) receives 1 or 2. */
int ldetr_p3_conv2d_bwd_pair(const void* dy, int N, int OH, int OW, int Cout, const void* wb, const void* x, int Cin, int KH, int KW, int stride, int pad,
                             int IH, int IW, const ldetr_p3_epilogue* ep, void* dx_p3, float* dx_f32, const float* dy_scale, float* dw,
                             int* launches, void* stream);

/* sizeof of {ldetr_ln_args, ldetr_ffn_args, ldetr_mha_small_args, ldetr_mha_cross_args, ldetr_wgrad_desc, ldetr_p3_epilogue}: lets a host binding check
 * its mirror of the argument blocks. */
int ldetr_struct_sizes(int32_t* out6);

/* Introspection for the parity tests: what the launch policy chose for the calling thread's most recent plane-format launch.
 * info10 = {kind (1 gather, 2 patch, 3 weight gradient, 4 paired gather + weight gradient, 5 paired patch + weight gradient), tile rows, tile columns,
 * waves per block, split-K factor, XCD array rows, XCD array columns, grid z (parity classes / groups), weight-gradient pixel slices, blocks}. */
int ldetr_p3_last_launch(int32_t* info10);

/* The same for the fp32-operand engine (ldetr_gemm_f32 / ldetr_gemm_pair_f32 / ldetr_conv2d_* / ldetr_conv_transpose2d_*): the calling thread's most
 * recent contraction launch.  info10 = {kind (1 gemm_f32_kernel tile, 2 gemm_small_kernel, 3 gemm_small_pair_kernel, 4 conv3x3_c32(_split)_kernel,
 * 5 wgrad_c32_3x3_kernel, 6 stem_conv7x7_kernel), tile rows, tile columns, k-tile, waves per block, FAST (block-uniform address parts in SGPRs),
 * SPLIT (1 = bf16 pipe with the exact 3-way operand split), split-K factor, blocks, operand modes (AMODE * 16 + BMODE; kinds 2 / 3: ta * 2 + tb)}. */
int ldetr_engine_last_launch(int32_t* info10);

#ifdef __cplusplus
}
#endif
#endif /* LDETR_HIP_H */
