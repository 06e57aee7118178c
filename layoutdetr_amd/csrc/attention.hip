// Fused multi-head attention (scaled QK^T + key-padding mask + softmax + dropout + PV), forward
// and backward, for the DETR shapes of LayoutDETR: head_dim 32, Lk <= 256 in registers (longer key sequences: chunked online softmax).
// Replaces the attention core of nn.MultiheadAttention as called at
// training/detr_transformer.py:208-209 (encoder self), :273-274 (decoder self), :277-280 (decoder cross)
// and by nn.TransformerEncoderLayer in training/util.py:21-26 / networks_detr.py:242-243.
//
// Design: one 64-lane wave per (batch, head, 16-row tile).  Scores are produced *transposed*
// (S^T = K Q^T) with v_mfma_f32_16x16x4_f32, so a lane holds 4 keys x 1 query per key tile and the
// whole score row lives in registers.  Because the reduction order over keys is free, the C-layout
// registers of S^T feed the P.V product directly as the MFMA B operand (lane group g, register t
// <-> key 16j + 4g + t): no LDS round trip, no transposes, no score matrix in HBM.
// The backward pass recomputes probabilities from the saved log-sum-exp.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct AttnParams {
    const float* q; const float* k; const float* v;
    long ldq, ldk, ldv;
    const unsigned char* kpm;  // [B][Lk], nonzero = masked key
    float* o; long ldo;
    float* lse;  // [B][H][Lq]
    const float* dout; long lddo;
    float* dq; long lddq;
    float* dk; long lddk;
    float* dv; long lddv;
    int B, H, Lq, Lk;
    float scale, p_drop;
    unsigned long long seed;
    const unsigned long long* seed_ptr;
    int causal;   // self-attention of a decoder: key j visible to query i only if j <= i (BertSelfAttention with is_decoder, med.py:704-739)
};

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Guarded element load WITHOUT control flow: an `ok ? *p : 0` form makes hipcc wrap every load in its own exec-masked region, and the
// loads of a key tile then issue one by one, each waited for before its MFMA (encoder self-attention: 16 us for 0.13 GFLOP).  Selecting
// the ADDRESS (any valid one when out of range) keeps the loads unconditional, so the unrolled tile issues them all at once.
__device__ __forceinline__ float ldz(const float* p, bool ok, const float* safe) {
    const float v = *(ok ? p : safe);
    return ok ? v : 0.f;
}

// Dropout element index: ((b*H + h)*Lq + q)*Lk + key.
__device__ __forceinline__ float attn_drop(const AttnParams& p, int bh, int q, int key, float inv_keep) {
    return drop_scale(p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull), ((uint64_t)bh * p.Lq + q) * p.Lk + key, p.p_drop, inv_keep);
}

// DHC = head_dim / 32 (1 for the DETR blocks; 2..6 for the BERT text encoder's 64..192-wide heads, forward only).
template <int NKT, int DHC>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnParams p) {
    constexpr int DH = 32 * DHC;
    const int nqt = (p.Lq + 15) >> 4;
    const int qt = blockIdx.x % nqt;
    const int bh = blockIdx.x / nqt;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int q0 = qt << 4;
    const int qrow = q0 + li;
    const bool qok = qrow < p.Lq;
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH;
    const float* kb = p.k + (long)b * p.Lk * p.ldk + h * DH;
    const float* vb = p.v + (long)b * p.Lk * p.ldv + h * DH;
    const unsigned char* kpm = p.kpm ? p.kpm + (long)b * p.Lk : nullptr;

    float qf[8 * DHC];
#pragma unroll
    for (int kk = 0; kk < 8 * DHC; kk++) qf[kk] = ldz(qp + 4 * kk + g, qok, p.q) * p.scale;

    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NKT; j++) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int krow = 16 * j + li;
        const bool kok = krow < p.Lk;
        const float* kp = kb + (long)krow * p.ldk;
#pragma unroll
        for (int kk = 0; kk < 8 * DHC; kk++) acc = MFMA16(ldz(kp + 4 * kk + g, kok, p.q), qf[kk], acc);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int key = 16 * j + 4 * g + r;
            bool masked = (key >= p.Lk) || (kpm && kpm[key]) || (p.causal && key > qrow);
            acc[r] = masked ? -INFINITY : acc[r];
            mx = fmaxf(mx, acc[r]);
        }
        s[j] = acc;
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) { float e = expf(s[j][r] - mx); s[j][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (p.lse && g == 0 && qok) p.lse[(long)bh * p.Lq + qrow] = mx + logf(sum);
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;

    f32x4 o[2 * DHC];
#pragma unroll
    for (int c = 0; c < 2 * DHC; c++) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NKT; j++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int key = 16 * j + 4 * g + t;
            const bool kok = key < p.Lk;
            float pv = s[j][t] * inv;
            if (p.p_drop > 0.f) pv *= attn_drop(p, bh, qrow, key, inv_keep);
            const float* vp = vb + (long)key * p.ldv;
#pragma unroll
            for (int c = 0; c < 2 * DHC; c++) o[c] = MFMA16(ldz(vp + 16 * c + li, kok, p.q), pv, o[c]);
        }
    }
    if (qok) {
        float* op = p.o + ((long)b * p.Lq + qrow) * p.ldo + h * DH + 4 * g;
#pragma unroll
        for (int c = 0; c < 2 * DHC; c++) *reinterpret_cast<float4*>(op + 16 * c) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
    }
}

// Long-key forward (Lk > 256: background_size above 512, i.e. more than 16x16 trunk positions).  Same wave layout as
// attn_fwd_kernel, but the keys are walked in chunks of 256 with a running maximum / denominator (online softmax): the output
// accumulators are rescaled by exp(m_old - m_new) when the maximum moves and normalised once at the end, so registers stay at
// the 16-tile footprint whatever Lk is.  The saved log-sum-exp is the global one, which is all the backward needs.
template <int DHC>
__global__ __launch_bounds__(64) void attn_fwd_long_kernel(AttnParams p) {
    constexpr int DH = 32 * DHC, NKT = 16;
    const int nqt = (p.Lq + 15) >> 4;
    const int qt = blockIdx.x % nqt;
    const int bh = blockIdx.x / nqt;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int qrow = (qt << 4) + li;
    const bool qok = qrow < p.Lq;
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH;
    const float* kb = p.k + (long)b * p.Lk * p.ldk + h * DH;
    const float* vb = p.v + (long)b * p.Lk * p.ldv + h * DH;
    const unsigned char* kpm = p.kpm ? p.kpm + (long)b * p.Lk : nullptr;
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;

    float qf[8 * DHC];
#pragma unroll
    for (int kk = 0; kk < 8 * DHC; kk++) qf[kk] = ldz(qp + 4 * kk + g, qok, p.q) * p.scale;

    f32x4 o[2 * DHC];
#pragma unroll
    for (int c = 0; c < 2 * DHC; c++) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float run_max = -INFINITY, run_sum = 0.f;
    const int nkt = (p.Lk + 15) >> 4;
    for (int j0 = 0; j0 < nkt; j0 += NKT) {
        f32x4 s[NKT];
        float mx = run_max;
#pragma unroll
        for (int j = 0; j < NKT; j++) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int krow = 16 * (j0 + j) + li;
            const bool kok = krow < p.Lk;
            const float* kp = kb + (long)krow * p.ldk;
#pragma unroll
            for (int kk = 0; kk < 8 * DHC; kk++) acc = MFMA16(ldz(kp + 4 * kk + g, kok, p.q), qf[kk], acc);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * (j0 + j) + 4 * g + r;
                const bool masked = (key >= p.Lk) || (kpm && kpm[key]) || (p.causal && key > qrow);
                acc[r] = masked ? -INFINITY : acc[r];
                mx = fmaxf(mx, acc[r]);
            }
            s[j] = acc;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        // every key so far masked: keep the accumulators (all zero) untouched and avoid (-inf) - (-inf)
        const float ref = (mx == -INFINITY) ? 0.f : mx;
        const float alpha = expf(run_max - ref);          // run_max = -inf -> 0
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < NKT; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) { const float e = expf(s[j][r] - ref); s[j][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        run_sum = run_sum * alpha + sum;
        run_max = mx;
#pragma unroll
        for (int c = 0; c < 2 * DHC; c++) { o[c][0] *= alpha; o[c][1] *= alpha; o[c][2] *= alpha; o[c][3] *= alpha; }
#pragma unroll
        for (int j = 0; j < NKT; j++) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int key = 16 * (j0 + j) + 4 * g + t;
                const bool kok = key < p.Lk;
                float pv = s[j][t];
                if (p.p_drop > 0.f) pv *= attn_drop(p, bh, qrow, key, inv_keep);
                const float* vp = vb + (long)key * p.ldv;
#pragma unroll
                for (int c = 0; c < 2 * DHC; c++) o[c] = MFMA16(ldz(vp + 16 * c + li, kok, p.q), pv, o[c]);
            }
        }
    }
    const float inv = 1.f / run_sum;
    if (p.lse && g == 0 && qok) p.lse[(long)bh * p.Lq + qrow] = run_max + logf(run_sum);
    if (qok) {
        float* op = p.o + ((long)b * p.Lq + qrow) * p.ldo + h * DH + 4 * g;
#pragma unroll
        for (int c = 0; c < 2 * DHC; c++)
            *reinterpret_cast<float4*>(op + 16 * c) = make_float4(o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv);
    }
}

// Wide-head forward (BERT text encoder / LM decoder: head_dim 64..192, up to 256 tokens).  Same per-wave algorithm as
// attn_fwd_kernel, but a block is 4 waves = 64 queries of one (batch, head) and the key / value rows go through LDS in chunks of
// 64 keys: loaded once per block with coalesced 16-byte loads and shared by the four query tiles, instead of every wave fetching
// every K and V element from global memory with 4-byte strided loads (which at 144 x 4 heads x 256 tokens ran at 9.6 TFLOP/s).
// Row pitch DH + 4 floats: the MFMA operand reads Ks[key li][4 kk + g] and Vs[key 4 g + t][16 c + li] then touch every bank
// exactly twice per 64 lanes, the minimum.  NCH = number of 64-key chunks (Lk <= 64 NCH).
template <int NCH, int DHC>
__global__ __launch_bounds__(256) void attn_fwd_wide_kernel(AttnParams p) {
    constexpr int DH = 32 * DHC, DHP = DH + 4, NKT = 4 * NCH;
    extern __shared__ __attribute__((aligned(16))) float kv_tile[];      // [64][DHP]
    const int nq64 = (p.Lq + 63) >> 6;
    const int qb = blockIdx.x % nq64;
    const int bh = blockIdx.x / nq64;
    const int b = bh / p.H, h = bh - b * p.H;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int qrow = (qb << 6) + (wave << 4) + li;
    const bool qok = qrow < p.Lq;
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH;
    const float* kb = p.k + (long)b * p.Lk * p.ldk + h * DH;
    const float* vb = p.v + (long)b * p.Lk * p.ldv + h * DH;
    const unsigned char* kpm = p.kpm ? p.kpm + (long)b * p.Lk : nullptr;

    auto stage = [&](const float* base, long ld, int key0) {             // rows key0 .. key0 + 63 of K or V -> LDS (zero past Lk)
        constexpr int F4 = DH / 4;
        for (int i = threadIdx.x; i < 64 * F4; i += 256) {
            const int row = i / F4, c4 = i - row * F4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (key0 + row < p.Lk) v = *reinterpret_cast<const float4*>(base + (long)(key0 + row) * ld + 4 * c4);
            *reinterpret_cast<float4*>(kv_tile + row * DHP + 4 * c4) = v;
        }
    };

    float qf[8 * DHC];
#pragma unroll
    for (int kk = 0; kk < 8 * DHC; kk++) qf[kk] = ldz(qp + 4 * kk + g, qok, p.q) * p.scale;

    f32x4 s[NKT];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (c > 0) __syncthreads();
        stage(kb, p.ldk, 64 * c);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = 4 * c + jj;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* kp = kv_tile + (16 * jj + li) * DHP + g;
#pragma unroll
            for (int kk = 0; kk < 8 * DHC; kk++) acc = MFMA16(kp[4 * kk], qf[kk], acc);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * j + 4 * g + r;
                const bool masked = (key >= p.Lk) || (kpm && kpm[key]) || (p.causal && key > qrow);
                acc[r] = masked ? -INFINITY : acc[r];
                mx = fmaxf(mx, acc[r]);
            }
            s[j] = acc;
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NKT; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) { float e = expf(s[j][r] - mx); s[j][r] = e; sum += e; }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (p.lse && g == 0 && qok) p.lse[(long)bh * p.Lq + qrow] = mx + logf(sum);
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;

    f32x4 o[2 * DHC];
#pragma unroll
    for (int cc = 0; cc < 2 * DHC; cc++) o[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        __syncthreads();
        stage(vb, p.ldv, 64 * c);
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 4; jj++) {
            const int j = 4 * c + jj;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int key = 16 * j + 4 * g + t;
                float pv = s[j][t] * inv;
                if (p.p_drop > 0.f) pv *= attn_drop(p, bh, qrow, key, inv_keep);
                const float* vp = kv_tile + (16 * jj + 4 * g + t) * DHP + li;
#pragma unroll
                for (int cc = 0; cc < 2 * DHC; cc++) o[cc] = MFMA16(vp[16 * cc], pv, o[cc]);
            }
        }
    }
    if (qok) {
        float* op = p.o + ((long)b * p.Lq + qrow) * p.ldo + h * DH + 4 * g;
#pragma unroll
        for (int cc = 0; cc < 2 * DHC; cc++) *reinterpret_cast<float4*>(op + 16 * cc) = make_float4(o[cc][0], o[cc][1], o[cc][2], o[cc][3]);
    }
}

// Backward, query-major half: dQ for one 16-query tile.
template <int NKT, int DHC>
__device__ __forceinline__ void attn_bwd_dq(const AttnParams& p, int bh, int qt) {
    constexpr int DH = 32 * DHC;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int qrow = (qt << 4) + li;
    const bool qok = qrow < p.Lq;
    const float* qp = p.q + ((long)b * p.Lq + qrow) * p.ldq + h * DH;
    const float* dop = p.dout + ((long)b * p.Lq + qrow) * p.lddo + h * DH;
    const float* op = p.o + ((long)b * p.Lq + qrow) * p.ldo + h * DH;
    const float* kb = p.k + (long)b * p.Lk * p.ldk + h * DH;
    const float* vb = p.v + (long)b * p.Lk * p.ldv + h * DH;
    const unsigned char* kpm = p.kpm ? p.kpm + (long)b * p.Lk : nullptr;
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;

    float qf[8 * DHC], dof[8 * DHC];
    float delta = 0.f;
#pragma unroll
    for (int kk = 0; kk < 8 * DHC; kk++) {
        qf[kk] = ldz(qp + 4 * kk + g, qok, p.q) * p.scale;
        dof[kk] = ldz(dop + 4 * kk + g, qok, p.q);
        delta += dof[kk] * ldz(op + 4 * kk + g, qok, p.q);
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lse = qok ? p.lse[(long)bh * p.Lq + qrow] : 0.f;

    f32x4 dq[2 * DHC];
#pragma unroll
    for (int c = 0; c < 2 * DHC; c++) dq[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Lk <= 16 NKT: one pass (the loop bound is a launch constant).  Longer key sequences walk chunks of NKT tiles: every tile
    // is independent given the saved log-sum-exp, so nothing is rescaled.
    const int nkt_all = (p.Lk + 15) >> 4;
    for (int j0 = 0; j0 < nkt_all; j0 += NKT)
#pragma unroll
    for (int jj = 0; jj < NKT; jj++) {
        const int j = j0 + jj;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
        const int krow = 16 * j + li;
        const bool kok = krow < p.Lk;
        const float* kp = kb + (long)krow * p.ldk;
        const float* vp = vb + (long)krow * p.ldv;
#pragma unroll
        for (int kk = 0; kk < 8 * DHC; kk++) {
            sacc = MFMA16(ldz(kp + 4 * kk + g, kok, p.q), qf[kk], sacc);
            dpacc = MFMA16(ldz(vp + 4 * kk + g, kok, p.q), dof[kk], dpacc);
        }
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int key = 16 * j + 4 * g + r;
            bool masked = (key >= p.Lk) || (kpm && kpm[key]) || (p.causal && key > qrow);
            float pr = masked ? 0.f : expf(sacc[r] - lse);
            float dpv = dpacc[r];
            if (p.p_drop > 0.f) dpv *= attn_drop(p, bh, qrow, key, inv_keep);
            ds[r] = pr * (dpv - delta) * p.scale;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int key = 16 * j + 4 * g + t;
            const bool k2 = key < p.Lk;
            const float* kp2 = kb + (long)key * p.ldk;
#pragma unroll
            for (int c = 0; c < 2 * DHC; c++) dq[c] = MFMA16(ldz(kp2 + 16 * c + li, k2, p.q), ds[t], dq[c]);
        }
    }
    if (qok) {
        float* dqp = p.dq + ((long)b * p.Lq + qrow) * p.lddq + h * DH + 4 * g;
#pragma unroll
        for (int c = 0; c < 2 * DHC; c++) *reinterpret_cast<float4*>(dqp + 16 * c) = make_float4(dq[c][0], dq[c][1], dq[c][2], dq[c][3]);
    }
}

// Backward, key-major half: dK and dV for one 16-key tile; loops over query tiles.  Wide heads are processed in 32-column
// slabs of the head dimension for the OUTPUT (dK/dV accumulators), with the score recomputation over the full width: the
// accumulators of a 192-wide head would not fit the register file next to the K/V fragments.
template <int DHC>
__device__ __forceinline__ void attn_bwd_dkv(const AttnParams& p, int bh, int kt) {
    constexpr int DH = 32 * DHC;
    const int b = bh / p.H, h = bh - b * p.H;
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    const int krow = (kt << 4) + li;
    const bool kok = krow < p.Lk;
    const float* kp = p.k + ((long)b * p.Lk + krow) * p.ldk + h * DH;
    const float* vp = p.v + ((long)b * p.Lk + krow) * p.ldv + h * DH;
    const unsigned char* kpm = p.kpm ? p.kpm + (long)b * p.Lk : nullptr;
    const bool kmasked = !kok || (kpm && kpm[krow]);
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    // B operands (k = d, j = key): K^T and V^T fragments, loaded once.
    float kf[8 * DHC], vf[8 * DHC];
#pragma unroll
    for (int kk = 0; kk < 8 * DHC; kk++) { kf[kk] = ldz(kp + 4 * kk + g, kok, p.q); vf[kk] = ldz(vp + 4 * kk + g, kok, p.q); }

    f32x4 dk[2 * DHC], dv[2 * DHC];
#pragma unroll
    for (int c = 0; c < 2 * DHC; c++) { dk[c] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[c] = dk[c]; }
    const int nqt = (p.Lq + 15) >> 4;
    for (int jq = 0; jq < nqt; jq++) {
        // A operands (i = query, k = d)
        const int qa = 16 * jq + li;
        const bool qaok = qa < p.Lq;
        const float* qp = p.q + ((long)b * p.Lq + qa) * p.ldq + h * DH;
        const float* dop = p.dout + ((long)b * p.Lq + qa) * p.lddo + h * DH;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8 * DHC; kk++) {
            sacc = MFMA16(ldz(qp + 4 * kk + g, qaok, p.q) * p.scale, kf[kk], sacc);
            dpacc = MFMA16(ldz(dop + 4 * kk + g, qaok, p.q), vf[kk], dpacc);
        }
        // C layout: row = query 16jq + 4g + r, col = key li.
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int qr = 16 * jq + 4 * g + r;
            const bool qok = qr < p.Lq;
            // delta[qr] = sum_d O*dO over DH dims: the 16 lanes of the group each take 2*DHC dims.
            const float* o_r = p.o + ((long)b * p.Lq + qr) * p.ldo + h * DH;
            const float* do_r = p.dout + ((long)b * p.Lq + qr) * p.lddo + h * DH;
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < 2 * DHC; c++) part += ldz(o_r + 16 * c + li, qok, p.q) * ldz(do_r + 16 * c + li, qok, p.q);
            part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64); part += __shfl_xor(part, 8, 64);
            const float lse = qok ? p.lse[(long)bh * p.Lq + qr] : 0.f;
            float pr = (kmasked || !qok || (p.causal && krow > qr)) ? 0.f : expf(sacc[r] - lse);
            float dm = 1.f;
            if (p.p_drop > 0.f) dm = attn_drop(p, bh, qr, krow, inv_keep);
            pd[r] = pr * dm;
            ds[r] = pr * (dpacc[r] * dm - part) * p.scale;
        }
        // dV^T[d][key] += sum_q dO^T[d][q] * Pdrop[q][key];  dK^T[d][key] += sum_q Q^T[d][q] * dS[q][key]
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int qr = 16 * jq + 4 * g + t;
            const bool qok = qr < p.Lq;
            const float* q_r = p.q + ((long)b * p.Lq + qr) * p.ldq + h * DH;
            const float* do_r = p.dout + ((long)b * p.Lq + qr) * p.lddo + h * DH;
#pragma unroll
            for (int c = 0; c < 2 * DHC; c++) {
                dv[c] = MFMA16(ldz(do_r + 16 * c + li, qok, p.q), pd[t], dv[c]);
                dk[c] = MFMA16(ldz(q_r + 16 * c + li, qok, p.q), ds[t], dk[c]);
            }
        }
    }
    if (kok) {
        float* dkp = p.dk + ((long)b * p.Lk + krow) * p.lddk + h * DH + 4 * g;
        float* dvp = p.dv + ((long)b * p.Lk + krow) * p.lddv + h * DH + 4 * g;
#pragma unroll
        for (int c = 0; c < 2 * DHC; c++) {
            *reinterpret_cast<float4*>(dkp + 16 * c) = make_float4(dk[c][0], dk[c][1], dk[c][2], dk[c][3]);
            *reinterpret_cast<float4*>(dvp + 16 * c) = make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]);
        }
    }
}

template <int NKT, int DHC>
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnParams p) {
    const int nqt = (p.Lq + 15) >> 4, nkt = (p.Lk + 15) >> 4;
    const int per = nqt + nkt;
    const int bh = blockIdx.x / per;
    const int w = blockIdx.x - bh * per;
    if (w < nqt) attn_bwd_dq<NKT, DHC>(p, bh, w);
    else attn_bwd_dkv<DHC>(p, bh, w - nqt);
}

// Backward for head_dim 32 with Lq, Lk <= 64 (the DETR encoder's 64 x 64 self-attention and the decoders' cross-attention onto 64 memory
// tokens): ONE block of 8 waves per (batch, head) instead of nqt + nkt single-wave blocks.  Q, K, V and dO of the head are staged in
// LDS once (coalesced 16-byte loads; rows past the sequence are zero), delta = rowsum(O * dO) and the log-sum-exp once per query;
// waves 0..3 then run attn_bwd_dq's tile algorithm and waves 4..7 attn_bwd_dkv's with every MFMA operand read from LDS -- the
// single-wave kernel fetched each operand element from global memory with its own 4-byte load, per tile, and recomputed delta per key
// tile (encoder self-attention: 24.5 us for 0.13 GFLOP).  Same arithmetic, same dropout element index, same results.
__global__ __launch_bounds__(512) void attn_bwd_lds_kernel(AttnParams p) {
    constexpr int DH = 32, DHP = DH + 4;
    __shared__ __attribute__((aligned(16))) float Qs[64 * DHP], Ks[64 * DHP], Vs[64 * DHP], Ds[64 * DHP];
    __shared__ float delta_s[64], lse_s[64];
    __shared__ unsigned char km_s[64];
    const int bh = blockIdx.x, b = bh / p.H, h = bh - b * p.H;
    const int tid = threadIdx.x;
    {   // one float4 of each array per thread: row = tid / 8, columns 4 (tid % 8) .. + 3
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const bool qok = row < p.Lq, kok = row < p.Lk;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const long qr = (long)b * p.Lq + (qok ? row : 0), kr = (long)b * p.Lk + (kok ? row : 0);
        float4 q4 = *reinterpret_cast<const float4*>(p.q + qr * p.ldq + h * DH + c4);
        float4 d4 = *reinterpret_cast<const float4*>(p.dout + qr * p.lddo + h * DH + c4);
        float4 o4 = *reinterpret_cast<const float4*>(p.o + qr * p.ldo + h * DH + c4);
        float4 k4 = *reinterpret_cast<const float4*>(p.k + kr * p.ldk + h * DH + c4);
        float4 v4 = *reinterpret_cast<const float4*>(p.v + kr * p.ldv + h * DH + c4);
        if (!qok) { q4 = z4; d4 = z4; o4 = z4; }
        if (!kok) { k4 = z4; v4 = z4; }
        *reinterpret_cast<float4*>(Qs + row * DHP + c4) = q4;
        *reinterpret_cast<float4*>(Ds + row * DHP + c4) = d4;
        *reinterpret_cast<float4*>(Ks + row * DHP + c4) = k4;
        *reinterpret_cast<float4*>(Vs + row * DHP + c4) = v4;
        float part = o4.x * d4.x + o4.y * d4.y + o4.z * d4.z + o4.w * d4.w;
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
        if ((tid & 7) == 0) { delta_s[row] = part; lse_s[row] = qok ? p.lse[(long)bh * p.Lq + row] : 0.f; }
        if (tid < 64) km_s[tid] = (tid >= p.Lk || (p.kpm && p.kpm[(long)b * p.Lk + tid])) ? 1 : 0;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    const int nqt = (p.Lq + 15) >> 4, nkt = (p.Lk + 15) >> 4;
    if (wave < 4) {
        // ---- dQ of query tile `wave`
        if (wave >= nqt) return;
        const int qrow = (wave << 4) + li;
        const bool qok = qrow < p.Lq;
        float qf[8], dof[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { qf[kk] = Qs[qrow * DHP + 4 * kk + g] * p.scale; dof[kk] = Ds[qrow * DHP + 4 * kk + g]; }
        const float delta = delta_s[qrow], lse = lse_s[qrow];
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int j = 0; j < nkt; j++) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * j + li) * DHP + g;
            const float* vp = Vs + (16 * j + li) * DHP + g;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                sacc = MFMA16(kp[4 * kk], qf[kk], sacc);
                dpacc = MFMA16(vp[4 * kk], dof[kk], dpacc);
            }
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * j + 4 * g + r;
                const bool masked = km_s[key] || (p.causal && key > qrow);
                const float pr = masked ? 0.f : expf(sacc[r] - lse);
                float dpv = dpacc[r];
                if (p.p_drop > 0.f) dpv *= attn_drop(p, bh, qrow, key, inv_keep);
                ds[r] = pr * (dpv - delta) * p.scale;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float* kp2 = Ks + (16 * j + 4 * g + t) * DHP + li;
#pragma unroll
                for (int c = 0; c < 2; c++) dq[c] = MFMA16(kp2[16 * c], ds[t], dq[c]);
            }
        }
        if (qok) {
            float* dqp = p.dq + ((long)b * p.Lq + qrow) * p.lddq + h * DH + 4 * g;
#pragma unroll
            for (int c = 0; c < 2; c++) *reinterpret_cast<float4*>(dqp + 16 * c) = make_float4(dq[c][0], dq[c][1], dq[c][2], dq[c][3]);
        }
    } else {
        // ---- dK, dV of key tile `wave - 4`
        const int kt = wave - 4;
        if (kt >= nkt) return;
        const int krow = (kt << 4) + li;
        const bool kok = krow < p.Lk;
        const bool kmasked = km_s[krow] != 0;
        float kf[8], vf[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { kf[kk] = Ks[krow * DHP + 4 * kk + g]; vf[kk] = Vs[krow * DHP + 4 * kk + g]; }
        f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int jq = 0; jq < nqt; jq++) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
            const float* qp = Qs + (16 * jq + li) * DHP + g;
            const float* dop = Ds + (16 * jq + li) * DHP + g;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                sacc = MFMA16(qp[4 * kk] * p.scale, kf[kk], sacc);
                dpacc = MFMA16(dop[4 * kk], vf[kk], dpacc);
            }
            float pd[4], ds[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int qr = 16 * jq + 4 * g + r;
                const bool qok = qr < p.Lq;
                const float pr = (kmasked || !qok || (p.causal && krow > qr)) ? 0.f : expf(sacc[r] - lse_s[qr]);
                float dm = 1.f;
                if (p.p_drop > 0.f) dm = attn_drop(p, bh, qr, krow, inv_keep);
                pd[r] = pr * dm;
                ds[r] = pr * (dpacc[r] * dm - delta_s[qr]) * p.scale;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int qr = 16 * jq + 4 * g + t;
#pragma unroll
                for (int c = 0; c < 2; c++) {
                    dv[c] = MFMA16(Ds[qr * DHP + 16 * c + li], pd[t], dv[c]);
                    dk[c] = MFMA16(Qs[qr * DHP + 16 * c + li], ds[t], dk[c]);
                }
            }
        }
        if (kok) {
            float* dkp = p.dk + ((long)b * p.Lk + krow) * p.lddk + h * DH + 4 * g;
            float* dvp = p.dv + ((long)b * p.Lk + krow) * p.lddv + h * DH + 4 * g;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                *reinterpret_cast<float4*>(dkp + 16 * c) = make_float4(dk[c][0], dk[c][1], dk[c][2], dk[c][3]);
                *reinterpret_cast<float4*>(dvp + 16 * c) = make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]);
            }
        }
    }
}

static int check_attn(const AttnParams& p, const char* what) {
    LDETR_CHECK(p.q && p.k && p.v && p.o, "%s: null pointer", what);
    LDETR_CHECK(p.B > 0 && p.H > 0 && p.Lq > 0 && p.Lk > 0, "%s: empty problem", what);
    LDETR_CHECK(p.Lk <= 16384, "%s: Lk > 16384 is unsupported", what);
    LDETR_CHECK(p.p_drop >= 0.f && p.p_drop < 1.f, "%s: dropout must be in [0,1)", what);
    LDETR_CHECK((p.ldo % 4) == 0 && ((uintptr_t)p.o & 15) == 0, "%s: output must be 16-byte aligned", what);
    return LDETR_OK;
}

}  // namespace ldetr

using namespace ldetr;

extern "C" int ldetr_attention_fwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       const unsigned char* key_padding_mask, float* out, int64_t ldo, float* lse,
                                       int B, int H, int Lq, int Lk, int head_dim, float scale,
                                       float p_drop, uint64_t seed, const uint64_t* seed_ptr, int causal, void* stream) {
    LDETR_CHECK(head_dim >= 32 && head_dim <= 192 && head_dim % 32 == 0, "attention_fwd: head_dim must be a multiple of 32 up to 192 (got %d)", head_dim);
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.kpm = key_padding_mask;
    p.o = out; p.ldo = ldo; p.lse = lse; p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk;
    p.scale = scale; p.p_drop = p_drop; p.seed = seed; p.seed_ptr = (const unsigned long long*)seed_ptr; p.causal = causal;
    LDETR_CHECK(!causal || Lq == Lk, "attention_fwd: the causal mask is defined for self-attention (Lq == Lk)");
    int rc = check_attn(p, "attention_fwd");
    if (rc) return rc;
    const int nqt = (Lq + 15) / 16, nkt = (Lk + 15) / 16;
    const int grid = B * H * nqt;
    hipStream_t st = (hipStream_t)stream;
#define LDETR_ATTN_FWD(DHC)                                                                     \
    do {                                                                                         \
        if (nkt <= 1) hipLaunchKernelGGL((attn_fwd_kernel<1, DHC>), grid, 64, 0, st, p);         \
        else if (nkt <= 2) hipLaunchKernelGGL((attn_fwd_kernel<2, DHC>), grid, 64, 0, st, p);    \
        else if (nkt <= 4) hipLaunchKernelGGL((attn_fwd_kernel<4, DHC>), grid, 64, 0, st, p);    \
        else if (nkt <= 8) hipLaunchKernelGGL((attn_fwd_kernel<8, DHC>), grid, 64, 0, st, p);    \
        else hipLaunchKernelGGL((attn_fwd_kernel<16, DHC>), grid, 64, 0, st, p);                 \
    } while (0)
    if (nkt > 16) {     // long key sequences: chunked online softmax
        switch (head_dim / 32) {
            case 1: hipLaunchKernelGGL((attn_fwd_long_kernel<1>), grid, 64, 0, st, p); break;
            case 2: hipLaunchKernelGGL((attn_fwd_long_kernel<2>), grid, 64, 0, st, p); break;
            case 3: hipLaunchKernelGGL((attn_fwd_long_kernel<3>), grid, 64, 0, st, p); break;
            case 4: hipLaunchKernelGGL((attn_fwd_long_kernel<4>), grid, 64, 0, st, p); break;
            case 5: hipLaunchKernelGGL((attn_fwd_long_kernel<5>), grid, 64, 0, st, p); break;
            default: hipLaunchKernelGGL((attn_fwd_long_kernel<6>), grid, 64, 0, st, p); break;
        }
        return check_launch("attention_fwd_long");
    }
    // wide heads (the BERT text models): LDS-staged K / V shared by four query tiles per block
    // (32-wide heads too when a block's four query tiles are mostly real: the DETR encoder's 64 x 64 self-attention; bit 1 of the switch)
    if ((head_dim >= 64 || Lq >= 48) && (ldk % 4) == 0 && (ldv % 4) == 0 && ((((uintptr_t)k) | ((uintptr_t)v)) & 15) == 0) {
        const int nch = (Lk + 63) / 64, wgrid = B * H * ((Lq + 63) / 64);
#define LDETR_ATTN_WIDE(NCH, DHC)                                                                                         \
    do {                                                                                                                   \
        constexpr size_t lds = (size_t)64 * (32 * DHC + 4) * sizeof(float);                                                \
        hipLaunchKernelGGL((attn_fwd_wide_kernel<NCH, DHC>), wgrid, 256, lds, st, p);                                      \
    } while (0)
#define LDETR_ATTN_WIDE_D(DHC)                                                                                            \
    do {                                                                                                                   \
        if (nch <= 1) LDETR_ATTN_WIDE(1, DHC); else if (nch <= 2) LDETR_ATTN_WIDE(2, DHC);                                 \
        else if (nch <= 3) LDETR_ATTN_WIDE(3, DHC); else LDETR_ATTN_WIDE(4, DHC);                                          \
    } while (0)
        switch (head_dim / 32) {
            case 1: LDETR_ATTN_WIDE_D(1); break;
            case 2: LDETR_ATTN_WIDE_D(2); break;
            case 3: LDETR_ATTN_WIDE_D(3); break;
            case 4: LDETR_ATTN_WIDE_D(4); break;
            case 5: LDETR_ATTN_WIDE_D(5); break;
            default: LDETR_ATTN_WIDE_D(6); break;
        }
#undef LDETR_ATTN_WIDE_D
#undef LDETR_ATTN_WIDE
        return check_launch("attention_fwd_wide");
    }
    switch (head_dim / 32) {
        case 1: LDETR_ATTN_FWD(1); break;
        case 2: LDETR_ATTN_FWD(2); break;
        case 3: LDETR_ATTN_FWD(3); break;
        case 4: LDETR_ATTN_FWD(4); break;
        case 5: LDETR_ATTN_FWD(5); break;
        default: LDETR_ATTN_FWD(6); break;
    }
#undef LDETR_ATTN_FWD
    return check_launch("attention_fwd");
}

extern "C" int ldetr_attention_bwd_f32(const float* q, int64_t ldq, const float* k, int64_t ldk, const float* v, int64_t ldv,
                                       const unsigned char* key_padding_mask, const float* out, int64_t ldo, const float* lse,
                                       const float* dout, int64_t lddo,
                                       float* dq, int64_t lddq, float* dk, int64_t lddk, float* dv, int64_t lddv,
                                       int B, int H, int Lq, int Lk, int head_dim, float scale,
                                       float p_drop, uint64_t seed, const uint64_t* seed_ptr, int causal, void* stream) {
    LDETR_CHECK(head_dim >= 32 && head_dim <= 192 && head_dim % 32 == 0, "attention_bwd: head_dim must be a multiple of 32 up to 192 (got %d)", head_dim);
    LDETR_CHECK(lse && dout && dq && dk && dv, "attention_bwd: null pointer");
    LDETR_CHECK(!causal || Lq == Lk, "attention_bwd: the causal mask is defined for self-attention (Lq == Lk)");
    AttnParams p; memset(&p, 0, sizeof(p));
    p.q = q; p.k = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.kpm = key_padding_mask;
    p.o = const_cast<float*>(out); p.ldo = ldo; p.lse = const_cast<float*>(lse);
    p.dout = dout; p.lddo = lddo; p.dq = dq; p.lddq = lddq; p.dk = dk; p.lddk = lddk; p.dv = dv; p.lddv = lddv;
    p.B = B; p.H = H; p.Lq = Lq; p.Lk = Lk; p.scale = scale; p.p_drop = p_drop; p.seed = seed; p.seed_ptr = (const unsigned long long*)seed_ptr;
    p.causal = causal;
    int rc = check_attn(p, "attention_bwd");
    if (rc) return rc;
    LDETR_CHECK((lddq % 4) == 0 && (lddk % 4) == 0 && (lddv % 4) == 0 &&
                (((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) == 0, "attention_bwd: gradients must be 16-byte aligned");
    const int nqt = (Lq + 15) / 16, nkt = (Lk + 15) / 16;
    const int grid = B * H * (nqt + nkt);
    hipStream_t st = (hipStream_t)stream;
    // 32-wide heads, 17..64 keys, <= 64 queries: one LDS-staged block per (batch, head) (attn_bwd_lds_kernel)
    if (head_dim == 32 && Lq <= 64 && Lk <= 64 && nkt >= 2 && (ldq % 4) == 0 && (ldk % 4) == 0 && (ldv % 4) == 0 && (lddo % 4) == 0 && (ldo % 4) == 0 &&
        ((((uintptr_t)q) | ((uintptr_t)k) | ((uintptr_t)v) | ((uintptr_t)dout) | ((uintptr_t)out)) & 15) == 0) {   // (dq / dk / dv: checked above for every path)
        hipLaunchKernelGGL(attn_bwd_lds_kernel, B * H, 512, 0, st, p);
        return check_launch("attention_bwd_lds");
    }
#define LDETR_ATTN_BWD(DHC)                                                                     \
    do {                                                                                         \
        if (nkt <= 1) hipLaunchKernelGGL((attn_bwd_kernel<1, DHC>), grid, 64, 0, st, p);         \
        else if (nkt <= 2) hipLaunchKernelGGL((attn_bwd_kernel<2, DHC>), grid, 64, 0, st, p);    \
        else if (nkt <= 4) hipLaunchKernelGGL((attn_bwd_kernel<4, DHC>), grid, 64, 0, st, p);    \
        else if (nkt <= 8) hipLaunchKernelGGL((attn_bwd_kernel<8, DHC>), grid, 64, 0, st, p);    \
        else hipLaunchKernelGGL((attn_bwd_kernel<16, DHC>), grid, 64, 0, st, p);                 \
    } while (0)
    switch (head_dim / 32) {
        case 1: LDETR_ATTN_BWD(1); break;
        case 2: LDETR_ATTN_BWD(2); break;
        case 3: LDETR_ATTN_BWD(3); break;
        case 4: LDETR_ATTN_BWD(4); break;
        case 5: LDETR_ATTN_BWD(5); break;
        default: LDETR_ATTN_BWD(6); break;
    }
#undef LDETR_ATTN_BWD
    return check_launch("attention_bwd");
}
