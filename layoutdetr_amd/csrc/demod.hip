// Demodulation coefficients of the StyleGAN2 modulated convolution, forward and backward, fused.
//   dcoefs[b, o] = rsqrt( sum_{i, kh, kw} (w[o, i, kh, kw] * s[b, i])^2 + eps )            training/networks_stylegan2.py:57-61
// The reference materialises w * s as a [B, O, I, kh, kw] tensor; evaluated here as  rsqrt( sum_i s[b,i]^2 * W2[o,i] + eps )  with
// W2[o, i] = sum_taps w^2  (the squared-operand identity of SURVEY §7).  As torch ops that is ~6 launches forward and ~10 backward
// per layer (square, sum, square, GEMM, add, rsqrt and their autograd nodes) x 14 layers; here: one launch forward (which also
// leaves W2 for the backward) and two backward:
//   t[b, o]      = -1/2 * g[b, o] * dcoefs[b, o]^3
//   dw[o, i, .]  = 2 w[o, i, .] * sum_b t[b, o] s[b, i]^2         (+= into the caller's gradient buffer, same strides as w)
//   ds[b, i]     = 2 s[b, i]    * sum_o t[b, o] W2[o, i]
// The weight is addressed through its strides, so both the OIHW and the channels_last (OHWI in memory) parameter layouts work.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct DemodParams {
    const float* w; long so, si, sh, sw;       // weight [O][I][KH][KW] element strides
    const float* s;                            // styles [B][I]
    float* d;                                  // dcoefs [B][O]
    float* w2;                                 // [O][I]
    const float* g;                            // upstream gradient [B][O]
    float* dw; int dw_accumulate;              // same strides as w
    float* ds;                                 // [B][I]
    int B, O, I, KH, KW;
    float eps;
};

constexpr int DEMOD_MAXB = 64;

// grid O: W2 row of this output channel, then one dot product per sample
__global__ __launch_bounds__(256) void demod_fwd_kernel(DemodParams p) {
    extern __shared__ float w2row[];           // [I]
    const int o = blockIdx.x;
    for (int i = threadIdx.x; i < p.I; i += 256) {
        const float* wp = p.w + (long)o * p.so + (long)i * p.si;
        float acc = 0.f;
        for (int kh = 0; kh < p.KH; kh++)
            for (int kw = 0; kw < p.KW; kw++) { const float v = wp[kh * p.sh + kw * p.sw]; acc += v * v; }
        w2row[i] = acc;
        p.w2[(long)o * p.I + i] = acc;
    }
    __syncthreads();
    // 16 samples at a time: one pass over the row with 16 accumulators, then ONE block reduction for all of them
    __shared__ float red16[4][16];
    for (int b0 = 0; b0 < p.B; b0 += 16) {
        float part[16];
#pragma unroll
        for (int u = 0; u < 16; u++) part[u] = 0.f;
        for (int i = threadIdx.x; i < p.I; i += 256) {
            const float wv = w2row[i];
#pragma unroll
            for (int u = 0; u < 16; u++)
                if (b0 + u < p.B) { const float sv = p.s[(long)(b0 + u) * p.I + i]; part[u] += sv * sv * wv; }
        }
#pragma unroll
        for (int u = 0; u < 16; u++) part[u] = wave_sum(part[u]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int u = 0; u < 16; u++) red16[threadIdx.x >> 6][u] = part[u];
        }
        __syncthreads();
        if (threadIdx.x < 16 && b0 + threadIdx.x < p.B) {
            const float tot = red16[0][threadIdx.x] + red16[1][threadIdx.x] + red16[2][threadIdx.x] + red16[3][threadIdx.x];
            p.d[(long)(b0 + threadIdx.x) * p.O + o] = rsqrtf(tot + p.eps);
        }
    }
}

// grid O: weight gradient of this output channel
__global__ __launch_bounds__(256) void demod_bwd_w_kernel(DemodParams p) {
    __shared__ float t[DEMOD_MAXB];
    const int o = blockIdx.x;
    if (threadIdx.x < p.B) {
        const float dv = p.d[(long)threadIdx.x * p.O + o];
        t[threadIdx.x] = -0.5f * p.g[(long)threadIdx.x * p.O + o] * dv * dv * dv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < p.I; i += 256) {
        float dw2 = 0.f;
        for (int b = 0; b < p.B; b++) { const float sv = p.s[(long)b * p.I + i]; dw2 += t[b] * sv * sv; }
        dw2 *= 2.f;
        const long base = (long)o * p.so + (long)i * p.si;
        for (int kh = 0; kh < p.KH; kh++)
            for (int kw = 0; kw < p.KW; kw++) {
                const long off = base + kh * p.sh + kw * p.sw;
                const float v = p.w[off] * dw2;
                if (p.dw_accumulate) p.dw[off] += v; else p.dw[off] = v;
            }
    }
}

// grid (ceil(I / 64), B): style gradient; block = 64 input channels x 4 lanes over the output channels
__global__ __launch_bounds__(256) void demod_bwd_s_kernel(DemodParams p) {
    extern __shared__ float tb[];              // [O] then [4][64] partial sums
    const int b = blockIdx.y, i = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    for (int o = threadIdx.x; o < p.O; o += 256) {
        const float dv = p.d[(long)b * p.O + o];
        tb[o] = -0.5f * p.g[(long)b * p.O + o] * dv * dv * dv;
    }
    __syncthreads();
    float acc = 0.f;
    if (i < p.I) {
#pragma unroll 8
        for (int o = lane; o < p.O; o += 4) acc += tb[o] * p.w2[(long)o * p.I + i];
    }
    float* part = tb + p.O;
    part[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0 && i < p.I) {
        acc += part[threadIdx.x + 64] + part[threadIdx.x + 128] + part[threadIdx.x + 192];
        p.ds[(long)b * p.I + i] = 2.f * p.s[(long)b * p.I + i] * acc;
    }
}

}  // namespace ldetr

using namespace ldetr;

static int demod_check(const DemodParams& p, const char* what) {
    LDETR_CHECK(p.w && p.s && p.d && p.w2, "%s: null pointer", what);
    LDETR_CHECK(p.B > 0 && p.B <= DEMOD_MAXB && p.O > 0 && p.I > 0 && p.KH > 0 && p.KW > 0, "%s: bad shape (batch <= %d)", what, DEMOD_MAXB);
    LDETR_CHECK((size_t)(p.I > p.O ? p.I : p.O) * sizeof(float) <= 48 * 1024, "%s: more than 12288 channels", what);
    return LDETR_OK;
}

extern "C" int ldetr_demod_fwd_f32(const float* weight, int64_t so, int64_t si, int64_t sh, int64_t sw, const float* styles, float* dcoefs,
                                   float* w2, int B, int O, int I, int KH, int KW, float eps, void* stream) {
    DemodParams p; memset(&p, 0, sizeof(p));
    p.w = weight; p.so = so; p.si = si; p.sh = sh; p.sw = sw; p.s = styles; p.d = dcoefs; p.w2 = w2;
    p.B = B; p.O = O; p.I = I; p.KH = KH; p.KW = KW; p.eps = eps;
    int rc = demod_check(p, "demod_fwd"); if (rc) return rc;
    hipLaunchKernelGGL(demod_fwd_kernel, dim3(O), 256, (size_t)I * sizeof(float), (hipStream_t)stream, p);
    return check_launch("demod_fwd");
}

extern "C" int ldetr_demod_bwd_f32(const float* weight, int64_t so, int64_t si, int64_t sh, int64_t sw, const float* styles, const float* dcoefs,
                                   const float* w2, const float* grad_dcoefs, float* dweight, int accumulate_dweight, float* dstyles, int B,
                                   int O, int I, int KH, int KW, void* stream) {
    DemodParams p; memset(&p, 0, sizeof(p));
    p.w = weight; p.so = so; p.si = si; p.sh = sh; p.sw = sw; p.s = styles; p.d = const_cast<float*>(dcoefs); p.w2 = const_cast<float*>(w2);
    p.g = grad_dcoefs; p.dw = dweight; p.dw_accumulate = accumulate_dweight; p.ds = dstyles;
    p.B = B; p.O = O; p.I = I; p.KH = KH; p.KW = KW;
    int rc = demod_check(p, "demod_bwd"); if (rc) return rc;
    LDETR_CHECK(grad_dcoefs && (dweight || dstyles), "demod_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dweight) {
        hipLaunchKernelGGL(demod_bwd_w_kernel, dim3(O), 256, 0, st, p);
        rc = check_launch("demod_bwd_w"); if (rc) return rc;
    }
    if (dstyles) {
        hipLaunchKernelGGL(demod_bwd_s_kernel, dim3(cdiv(I, 64), B), 256, (size_t)(O + 256) * sizeof(float), st, p);
        rc = check_launch("demod_bwd_s");
    }
    return rc;
}
