// Position-wise feed-forward block of the DETR layers as ONE launch per direction.
// Reference: training/detr_transformer.py:212-214 / 283-285 -- src2 = linear2(dropout(relu(linear1(src)))) with d_model 256 and
// dim_feedforward 2048 (networks_detr.py:101-103, 243, 269, 275), followed by src = norm(src + dropout(src2)).
//
// On the decoder-side stacks the block sees 144..320 tokens: linear1 and linear2 are two latency-bound launches (14 us each for
// 0.15 GFLOP) and their backward is two paired launches plus an activation-gradient pass.  Here a block owns a 32-token row tile and a
// 64-wide slice of the hidden layer:
//   forward   H = dropout(relu(X W1_s^T + b1_s)) (kept in LDS, written once for the backward), then the slice's contribution
//             Y_s = H W2_s^T to all 256 outputs -> ypart[s][M][256].  The 32 partial sums (+ b2) are added where they are consumed: in
//             the residual + LayerNorm launch (ldetr_layernorm_fwd_parts_f32), in slice order, so the forward stays deterministic.
//   backward  dH = (dY W2_s) * [H > 0] / keep (written once: operand of the weight gradients), then the slice's contribution dH W1_s to
//             the input gradient -> dxpart[s][M][256], summed (in slice order, with the residual-path gradient) by the LayerNorm
//             backward of the sub-block in front (ldetr_layernorm_bwd_parts_f32): no atomics, results reproducible run to run.
//             The weight gradients dW2 += dY^T H and dW1 += dH^T X (K = tokens) are a different parallelisation (1024 output tiles,
//             no reduction across blocks): one paired small-tile launch of the contraction engine (ldetr_gemm_pair_f32, TN + TN, bias
//             gradients as row sums).  (First versions: weight gradients accumulated here with 6.5 M fp32 atomics per launch -- 50 us,
//             no faster than the unfused path; dX by atomics -- run-to-run noise of 3e-7 that a saturated softmax upstream amplified to 1e-3.)
// Operands stream global -> registers in MFMA operand order (v_mfma_f32_32x32x2_f32, exact fp32) like gemm_small_kernel; the only
// LDS traffic is the hidden tile.  D must be 256 (8 column tiles = 2 per wave), the hidden width a multiple of 64.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

void note_engine_launch(bool bf16_split_pipe);   // gemm_conv.hip (ldetr_engine_launch_counts): these kernels contract on the f32 MFMA pipe

// One problem = the public argument block (include/ldetr_hip.h: ldetr_ffn_args).  A launch carries one or two: the second problem's blocks follow
// the first's in a 1-D grid (two independent stacks' feed-forward blocks as ONE launch); block c of a problem = (row tile c % gx, hidden slice c / gx).
typedef ldetr_ffn_args FfnParams;

constexpr int FD = 256, FHS = 64, FHP = 68;   // model width, hidden slice, LDS pitch of the hidden tile

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float ffn_acc_t;

__device__ __forceinline__ void ld_k16(const __amdgpu_buffer_rsrc_t& rs, int voff, int soff, float (&f)[16]) {   // 16 consecutive k of one row
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * j, soff, 0);
        f[4 * j] = __int_as_float(v[0]); f[4 * j + 1] = __int_as_float(v[1]); f[4 * j + 2] = __int_as_float(v[2]); f[4 * j + 3] = __int_as_float(v[3]);
    }
}
__device__ __forceinline__ void ld_r16(const __amdgpu_buffer_rsrc_t& rs, int voff, int soff, int ld4, float (&f)[16]) {   // 16 consecutive k of one column (pitch ld4 bytes)
#pragma unroll
    for (int t = 0; t < 16; t++) f[t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff + t * ld4, 0));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const float* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7fffffff, 0x00020000);
}
#define FFN_MFMA16(A, B, ACC) _Pragma("unroll") for (int t_ = 0; t_ < 16; t_++) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t_], B[t_], ACC, 0, 0, 0)

__global__ __launch_bounds__(256) void ffn_fwd_kernel(FfnParams pa, FfnParams pb, int nb0) {
    __shared__ float red[2][32][33];
    __shared__ float Hs[32][FHP];
    const bool second = (int)blockIdx.x >= nb0;
    const FfnParams& p = second ? pb : pa;
    const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, gx = (p.M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kl = lane >> 5;
    const int s = bid / gx, m0 = (bid - s * gx) * 32, j0 = s * FHS;
    const int OOB = (int)0x80000000;
    // ---- phase 1: wave (ct, kh) = 32 hidden units x half of the 256-long reduction
    const int ct = wave & 1, kh = wave >> 1;
    float b2[2][2][16];      // phase 2's weights: W2[(2 wave + q) * 32 + cl][j0 + 32 c + 16 kl ..]
    {
        const __amdgpu_buffer_rsrc_t rsX = rsrc(p.x), rsW = rsrc(p.w1), rsW2 = rsrc(p.w2);
        const int vA = (m0 + cl < p.M) ? (int)(((long)(m0 + cl) * p.ldx + kh * 128 + 16 * kl) * 4) : OOB;
        const int vB = ((j0 + ct * 32 + cl) * FD + kh * 128 + 16 * kl) * 4;
        ffn_acc_t acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        // every operand of the block is requested before the first MFMA (the whole reduction of a wave is 4 chunks: ONE memory round
        // trip instead of three -- at 18..320 tokens the launch is that round trip), the second GEMM's weights included
        float a[4][16], b[4][16];
#pragma unroll
        for (int c = 0; c < 4; c++) { ld_k16(rsX, vA, c * 128, a[c]); ld_k16(rsW, vB, c * 128, b[c]); }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int c = 0; c < 2; c++) ld_k16(rsW2, (((wave * 2 + q) * 32 + cl) * p.F + j0 + 16 * kl) * 4, c * 128, b2[q][c]);
        __builtin_amdgcn_sched_barrier(0);      // keep every request ahead of the first MFMA (the scheduler otherwise sinks them next to their use)
#pragma unroll
        for (int c = 0; c < 4; c++) FFN_MFMA16(a[c], b[c], acc);
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) red[ct][(r & 3) + 8 * (r >> 2) + 4 * kl][cl] = acc[r];
        }
        __syncthreads();
        if (kh == 0) {
            const float bias = p.b1[j0 + ct * 32 + cl];
            const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
            const uint64_t seed = p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kl;
                float v = acc[r] + red[ct][row][cl] + bias;
                v = v > 0.f ? v : 0.f;
                const long m = m0 + row;
                if (p.p_drop > 0.f) v *= drop_scale(seed, (uint64_t)(m * p.F + j0 + ct * 32 + cl), p.p_drop, inv_keep);
                Hs[row][ct * 32 + cl] = v;
                if (m < p.M) p.h[m * p.F + j0 + ct * 32 + cl] = v;
            }
        }
        __syncthreads();
    }
    // ---- phase 2: wave w -> output column tiles 2w, 2w+1 of Y_s = H (32 x 64) W2[:, slice]^T
    {
        ffn_acc_t acc[2];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float a[16];
#pragma unroll
            for (int t = 0; t < 16; t++) a[t] = Hs[cl][c * 32 + 16 * kl + t];
            FFN_MFMA16(a, b2[0][c], acc[0]);
            FFN_MFMA16(a, b2[1][c], acc[1]);
        }
        float* dst = p.ypart + ((long)s * p.M) * FD;
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                if (m < p.M) dst[m * FD + (wave * 2 + q) * 32 + cl] = acc[q][r];
            }
    }
}

__global__ __launch_bounds__(256) void ffn_bwd_kernel(FfnParams pa, FfnParams pb, int nb0) {
    __shared__ float red[2][32][33];
    __shared__ float dHs[32][FHP];
    const bool second = (int)blockIdx.x >= nb0;
    const FfnParams& p = second ? pb : pa;
    const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, gx = (p.M + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kl = lane >> 5;
    const int s = bid / gx, m0 = (bid - s * gx) * 32, j0 = s * FHS;
    const int OOB = (int)0x80000000;
    const __amdgpu_buffer_rsrc_t rsDY = rsrc(p.dy), rsW1 = rsrc(p.w1), rsW2 = rsrc(p.w2), rsX = rsrc(p.x);
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    float b2[2][2][16];
    // ---- phase 1: dH (32 tokens x 64 hidden) = dY (32 x 256) W2[:, slice]; wave (ct, kh) as in the forward
    {
        const int ct = wave & 1, kh = wave >> 1;
        const int vA = (m0 + cl < p.M) ? ((m0 + cl) * FD + kh * 128 + 16 * kl) * 4 : OOB;
        const int vB = (j0 + ct * 32 + cl + (kh * 128 + 16 * kl) * p.F) * 4;       // element (k = n, row = j): w2[n * F + j]
        const int ld4 = p.F * 4;
        ffn_acc_t acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        float a[4][16], b[4][16];
#pragma unroll
        for (int c = 0; c < 4; c++) { ld_k16(rsDY, vA, c * 128, a[c]); ld_r16(rsW2, vB, c * 32 * ld4, ld4, b[c]); }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int c = 0; c < 2; c++)      // phase 2's weights, element (k = j, row = c): w1[(j0 + j) * 256 + c]
                ld_r16(rsW1, ((wave * 2 + q) * 32 + cl + (j0 + c * 32 + 16 * kl) * FD) * 4, 0, FD * 4, b2[q][c]);
        float hvs[16];      // the saved hidden values of this wave's output tile (relu / dropout mask), requested with everything else
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            hvs[r] = (kh == 0 && m < p.M) ? p.h[m * p.F + j0 + ct * 32 + cl] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; c++) FFN_MFMA16(a[c], b[c], acc);
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) red[ct][(r & 3) + 8 * (r >> 2) + 4 * kl][cl] = acc[r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kl;
                const long m = m0 + row;
                const float hv = hvs[r];
                // relu'(pre) and the dropout mask in one test: the saved hidden value is positive exactly where both let the gradient through
                const float dh = hv > 0.f ? (acc[r] + red[ct][row][cl]) * inv_keep : 0.f;
                dHs[row][ct * 32 + cl] = dh;
                if (p.dh && m < p.M) p.dh[m * p.F + j0 + ct * 32 + cl] = dh;      // pre-activation gradient: operand of the weight-gradient launch
            }
        }
        __syncthreads();
    }
    // ---- phase 2: dX (32 x 256) += dH (32 x 64) W1[slice, :]; wave w -> input-feature tiles 2w, 2w+1
    {
        ffn_acc_t acc[2];
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[q][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; c++) {
            float a[16];
#pragma unroll
            for (int t = 0; t < 16; t++) a[t] = dHs[cl][c * 32 + 16 * kl + t];
            FFN_MFMA16(a, b2[0][c], acc[0]);
            FFN_MFMA16(a, b2[1][c], acc[1]);
        }
        float* dst = p.dxpart + ((long)s * p.M) * FD;
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                if (m < p.M) dst[m * FD + (wave * 2 + q) * 32 + cl] = acc[q][r];
            }
    }
}

}  // namespace ldetr

using namespace ldetr;

static int ffn_check(const char* what, int64_t M, int D, int F, const void* x, int64_t ldx) {
    LDETR_CHECK(D == FD, "%s: the fused feed-forward block is built for d_model = 256 (got %d)", what, D);
    LDETR_CHECK(F >= FHS && F % FHS == 0, "%s: hidden width must be a multiple of 64 (got %d)", what, F);
    LDETR_CHECK(M >= 0 && M <= (1 << 20), "%s: bad row count", what);
    LDETR_CHECK(x && ldx >= D && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0, "%s: x must be 16-byte aligned rows with a pitch that is a multiple of 4", what);
    LDETR_CHECK((long)M * ldx * 4 < 0x7fffffffL && (long)M * F * 4 < 0x7fffffffL && (long)(F / FHS) * M * D * 4 < 0x7fffffffL, "%s: operand above 2 GiB", what);
    return LDETR_OK;
}

static int ffn_check_args(const char* what, const FfnParams& p, bool bwd) {
    if (int rc = ffn_check(what, p.M, FD, p.F, p.x, p.ldx)) return rc;
    if (!bwd) {
        LDETR_CHECK(p.w1 && p.b1 && p.w2 && p.h && p.ypart, "%s: null pointer", what);
        LDETR_CHECK(p.p_drop >= 0.f && p.p_drop < 1.f, "%s: p_drop out of range", what);
    } else {
        LDETR_CHECK(p.dy && p.h && p.w1 && p.w2 && p.dxpart, "%s: null pointer", what);
        LDETR_CHECK((((uintptr_t)p.dy) & 15) == 0, "%s: dy must be 16-byte aligned", what);
    }
    return LDETR_OK;
}

static int ffn_launch(const ldetr_ffn_args* a, int n, bool bwd, void* stream) {
    const char* what = bwd ? "ffn_bwd" : "ffn_fwd";
    LDETR_CHECK(a && (n == 1 || n == 2), "%s: 1 or 2 problems", what);
    FfnParams p[2]; p[0] = a[0]; p[1] = n == 2 ? a[1] : a[0];
    int nb[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        if (int rc = ffn_check_args(what, p[i], bwd)) return rc;
        nb[i] = cdiv(p[i].M, 32) * (p[i].F / FHS);
    }
    if (nb[0] + nb[1] == 0) return LDETR_OK;
    if (bwd) hipLaunchKernelGGL(ffn_bwd_kernel, dim3(nb[0] + nb[1]), dim3(256), 0, (hipStream_t)stream, p[0], p[1], nb[0]);
    else hipLaunchKernelGGL(ffn_fwd_kernel, dim3(nb[0] + nb[1]), dim3(256), 0, (hipStream_t)stream, p[0], p[1], nb[0]);
    note_engine_launch(false);
    return check_launch(what);
}

extern "C" int ldetr_ffn_fwd_group_f32(const ldetr_ffn_args* a, int n, void* stream) { return ffn_launch(a, n, false, stream); }
extern "C" int ldetr_ffn_bwd_group_f32(const ldetr_ffn_args* a, int n, void* stream) { return ffn_launch(a, n, true, stream); }

extern "C" int ldetr_ffn_fwd_f32(const float* x, int64_t ldx, const float* w1, const float* b1, const float* w2, float* h, float* ypart,
                                 int64_t M, int D, int F, float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream) {
    if (int rc = ffn_check("ffn_fwd", M, D, F, x, ldx)) return rc;
    FfnParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = ldx; p.w1 = w1; p.b1 = b1; p.w2 = w2; p.h = h; p.ypart = ypart; p.M = (int)M; p.F = F;
    p.p_drop = p_drop; p.seed = seed; p.seed_ptr = seed_ptr;
    return ffn_launch(&p, 1, false, stream);
}

extern "C" int ldetr_ffn_bwd_f32(const float* dy, const float* x, int64_t ldx, const float* h, const float* w1, const float* w2,
                                 float* dxpart, float* dh, int64_t M, int D, int F, float p_drop, void* stream) {
    if (int rc = ffn_check("ffn_bwd", M, D, F, x, ldx)) return rc;
    FfnParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = ldx; p.w1 = w1; p.w2 = w2; p.h = const_cast<float*>(h); p.M = (int)M; p.F = F; p.p_drop = p_drop;
    p.dy = dy; p.dxpart = dxpart; p.dh = dh;
    return ffn_launch(&p, 1, true, stream);
}
