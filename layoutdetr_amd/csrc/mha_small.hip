// Self-attention block of the short-sequence DETR stacks as ONE forward launch: packed q|k|v projection, scaled QK^T + key-padding
// mask + softmax + dropout + PV, and the output projection.
// Reference: nn.MultiheadAttention(256, 8, dropout=0.1) as called at training/detr_transformer.py:273-274 (decoder self-attention,
// q = k = v = tgt) and through nn.TransformerEncoderLayer at training/util.py:21-26 / networks_detr.py:243, 269, 275 (the layout
// stacks of D: 9 or 10 tokens per sample).
//
// On those stacks a sample has L <= 16 tokens: the three launches of the unfused path (packed projection [B*L, 256] x [768, 256]^T:
// 8 us; fused attention: 6 us; output projection: 6-8 us) are each a fraction of a wave of work per CU.  Here a block owns one
// (sample, head) pair -- the unit inside which the whole sub-block is independent up to the output projection's sum over heads:
//   1. q, k, v of the head: X_b [16 x 256] . W_h^T [256 x 96], v_mfma_f32_16x16x4_f32, the reduction split four ways over the block's
//      waves (operands stream global -> registers in MFMA operand order, every load in flight before the first MFMA), partial tiles
//      summed through LDS in wave order (deterministic); + bias; written once to the packed qkv buffer the backward reads.
//   2. attention of the 16 x 16 score tile by one wave, exactly the register scheme of attn_fwd_kernel (S^T = K Q^T, its C layout is
//      the B operand of P V), same dropout element index -- so ldetr_attention_bwd_f32 regenerates the mask from the same seed.
//   3. the head's contribution to the output projection, O_h [16 x 32] . W_out[:, 32h : 32h+32]^T -> ypart[h][B*L][256]; the 8
//      contributions + bias + residual are added, in head order, by the LayerNorm launch that follows
//      (ldetr_layernorm_fwd_parts_f32) -- the same hand-off the fused feed-forward block uses.
// MFMA-bound per block (384 + 16 + 128 16x16x4 MFMAs = 4.4 k cycles per SIMD), 37-44 % of the rows are padding (9-10 of 16): the
// point is the two launches and two round trips through HBM that are gone, not the rate.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

void note_engine_launch(bool bf16_split_pipe);   // gemm_conv.hip (ldetr_engine_launch_counts): these kernels contract on the f32 MFMA pipe

// One problem = the public argument block (include/ldetr_hip.h: ldetr_mha_small_args): x [B*L, 256], w_in [768, 256], b_in [768], w_out [256, 256],
// kpm [B][L] (nonzero = masked key) or null, qkv [B*L, 768] (projection incl. bias, q unscaled), o [B*L, 256] (attention output before the output
// projection), lse [B][8][L], ypart [8][B*L][256]; backward: dr [B*L, 256] in, dqkv [B*L, 768] and dxpart [8][B*L][256] out.
// A launch carries one or two problems; the second one's (sample, head) blocks follow the first's in the grid.
typedef ldetr_mha_small_args MhaSmallParams;

constexpr int MS_D = 256, MS_H = 8, MS_DH = 32;
constexpr int MS_RP = 97, MS_QP = 100, MS_OP = 36;   // LDS pitches (floats): partial tiles, the head's q|k|v, the head's output

#define MS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void mha_small_fwd_kernel(MhaSmallParams pa, MhaSmallParams pb, int nb0) {
    __shared__ float red[4 * 16 * MS_RP];
    __shared__ __attribute__((aligned(16))) float qs[16 * MS_QP];
    __shared__ __attribute__((aligned(16))) float os[16 * MS_OP];
    const bool second = (int)blockIdx.x >= nb0;
    const MhaSmallParams& p = second ? pb : pa;
    const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = bid / MS_H, h = bid - b * MS_H;
    const int L = p.L;
    const long row0 = (long)b * L;

    // ---- operands of step 3 first (they depend on nothing): this wave's four column tiles of W_out, the head's 32 k
    f32x4 wo[4][2];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int gg = 0; gg < 2; gg++)
            wo[t][gg] = *reinterpret_cast<const f32x4*>(p.w_out + (long)(16 * (4 * w + t) + li) * MS_D + h * MS_DH + 16 * gg + 4 * g);

    // ---- 1. projection: this wave's quarter of the reduction (k in [64 w, 64 w + 64)); lane (li, g) holds k = 16 gg + 4 g + j of its row
    f32x4 xa[4], wb[6][4];
    {
        const bool rok = li < L;
        const float* xr = p.x + (row0 + (rok ? li : 0)) * p.ldx + 64 * w + 4 * g;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 16 * gg);
            xa[gg] = rok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ct = 0; ct < 6; ct++) {
            const float* wr = p.w_in + (long)((ct >> 1) * MS_D + h * MS_DH + (ct & 1) * 16 + li) * MS_D + 64 * w + 4 * g;
#pragma unroll
            for (int gg = 0; gg < 4; gg++) wb[ct][gg] = *reinterpret_cast<const f32x4*>(wr + 16 * gg);
        }
    }
    // the small operands of the later steps as well: nothing after this point waits for a first-touch global load
    float bias6[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int col = (tid + 256 * i) % 96;
        bias6[i] = p.b_in[(col >> 5) * MS_D + h * MS_DH + (col & 31)];
    }
    unsigned km = 0;                                   // bit r: key 4 g + r is masked
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int key = 4 * g + r;
        const bool masked = key >= L || (p.kpm && p.kpm[row0 + (key < L ? key : 0)]);
        km |= masked ? (1u << r) : 0u;
    }
    const uint64_t seed = p.seed + ((p.p_drop > 0.f && p.seed_ptr) ? *p.seed_ptr : 0ull);
    __builtin_amdgcn_sched_barrier(0);                 // (every load above is issued before the first MFMA)
    f32x4 acc[6];
#pragma unroll
    for (int ct = 0; ct < 6; ct++) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gg = 0; gg < 4; gg++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ct = 0; ct < 6; ct++) acc[ct] = MS_MFMA16(xa[gg][j], wb[ct][gg][j], acc[ct]);
    // acc[ct][r] = (token 4 g + r, column 16 ct + li) of this wave's partial sum
#pragma unroll
    for (int ct = 0; ct < 6; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[(w * 16 + 4 * g + r) * MS_RP + 16 * ct + li] = acc[ct][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const int e = tid + 256 * i, row = e / 96, col = e - row * 96, part = col >> 5, c = col & 31;
        float v = bias6[i];
#pragma unroll
        for (int ww = 0; ww < 4; ww++) v += red[(ww * 16 + row) * MS_RP + col];
        qs[row * MS_QP + col] = v;
        if (row < L) p.qkv[(row0 + row) * (3 * MS_D) + part * MS_D + h * MS_DH + c] = v;
    }
    __syncthreads();

    // ---- 2. attention of the (sample, head): wave 0; query li, keys 4 g + r (one key tile)
    if (w == 0) {
        const int bh = b * MS_H + h;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; kk++) s = MS_MFMA16(qs[li * MS_QP + 32 + 4 * kk + g], qs[li * MS_QP + 4 * kk + g] * p.scale, s);
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            s[r] = ((km >> r) & 1u) ? -INFINITY : s[r];
            mx = fmaxf(mx, s[r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; r++) { const float e = expf(s[r] - mx); s[r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        const bool qok = li < L;
        if (g == 0 && qok) p.lse[(long)bh * L + li] = mx + logf(sum);
        const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
        f32x4 oc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int key = 4 * g + t;
            float pv = s[t] * inv;
            if (p.p_drop > 0.f) pv *= drop_scale(seed, ((uint64_t)bh * L + li) * L + key, p.p_drop, inv_keep);   // attn_drop's element index
#pragma unroll
            for (int c = 0; c < 2; c++) oc[c] = MS_MFMA16(qs[key * MS_QP + 64 + 16 * c + li], pv, oc[c]);
        }
        // oc[c][r] = (query li, head column 16 c + 4 g + r)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            *reinterpret_cast<f32x4*>(os + li * MS_OP + 16 * c + 4 * g) = oc[c];
            if (qok) *reinterpret_cast<f32x4*>(p.o + (row0 + li) * MS_D + h * MS_DH + 16 * c + 4 * g) = oc[c];
        }
    }
    __syncthreads();

    // ---- 3. the head's contribution to the output projection: this wave's 64 of the 256 output columns
    f32x4 oa[2];
#pragma unroll
    for (int gg = 0; gg < 2; gg++) oa[gg] = *reinterpret_cast<const f32x4*>(os + li * MS_OP + 16 * gg + 4 * g);
    float* yp = p.ypart + ((long)h * p.B * L + row0) * MS_D;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < 2; gg++)
#pragma unroll
            for (int j = 0; j < 4; j++) y = MS_MFMA16(oa[gg][j], wo[t][gg][j], y);
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (4 * g + r < L) yp[(long)(4 * g + r) * MS_D + 16 * (4 * w + t) + li] = y[r];
    }
}

// Cross-attention sub-block of a decoder layer with <= 16 queries per sample onto <= 64 memory tokens whose K / V projections already
// exist (hip.attention.grouped_kv makes them for all layers at once): query projection + attention + per-head output projection in
// one launch (training/detr_transformer.py:277-280).  Same block = (sample, head) layout as above; K_h and V_h ([Lk x 32]) are staged
// in LDS with 16-byte loads while the query projection's MFMAs run.
// ldetr_mha_cross_args: x [B*Lq, 256] queries (tgt after norm1); w_q / b_q = rows 0..255 of in_proj_weight / in_proj_bias; k, v [B*Lk, >= 256]
// projected memory; kpm [B][Lk] or null; q [B*Lq, 256] projected queries incl. bias (unscaled), o, lse, ypart as above.
typedef ldetr_mha_cross_args MhaCrossParams;

__global__ __launch_bounds__(256) void mha_cross_fwd_kernel(MhaCrossParams p) {
    constexpr int KP = MS_DH + 4;
    __shared__ float red[4 * 16 * 33];
    __shared__ __attribute__((aligned(16))) float qs[16 * KP], os[16 * MS_OP], Ks[64 * KP], Vs[64 * KP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / MS_H, h = blockIdx.x - b * MS_H;
    const int Lq = p.Lq, Lk = p.Lk;
    const long row0 = (long)b * Lq, krow0 = (long)b * Lk;

    f32x4 wo[4][2];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int gg = 0; gg < 2; gg++)
            wo[t][gg] = *reinterpret_cast<const f32x4*>(p.w_out + (long)(16 * (4 * w + t) + li) * MS_D + h * MS_DH + 16 * gg + 4 * g);
    // query projection operands: this wave's quarter of the reduction
    f32x4 xa[4], wb[2][4];
    {
        const bool rok = li < Lq;
        const float* xr = p.x + (row0 + (rok ? li : 0)) * p.ldx + 64 * w + 4 * g;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 16 * gg);
            xa[gg] = rok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ct = 0; ct < 2; ct++) {
            const float* wr = p.w_q + (long)(h * MS_DH + ct * 16 + li) * MS_D + 64 * w + 4 * g;
#pragma unroll
            for (int gg = 0; gg < 4; gg++) wb[ct][gg] = *reinterpret_cast<const f32x4*>(wr + 16 * gg);
        }
    }
    // K_h, V_h -> LDS: 64 rows x 8 float4 each = one float4 of each per thread pair; rows past Lk are zero
    f32x4 kst[2], vst[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
        const bool ok = row < Lk;
        const long r = krow0 + (ok ? row : 0);
        const f32x4 kv = *reinterpret_cast<const f32x4*>(p.k + r * p.ldk + h * MS_DH + c4);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(p.v + r * p.ldv + h * MS_DH + c4);
        kst[i] = ok ? kv : f32x4{0.f, 0.f, 0.f, 0.f}; vst[i] = ok ? vv : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float qbias = (tid < 16 * 32) ? p.b_q[h * MS_DH + (tid & 31)] : 0.f;
    unsigned km = 0;                                   // bit 4 j + r: key 16 j + 4 g + r is masked
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int key = 16 * j + 4 * g + r;
            const bool masked = key >= Lk || (p.kpm && p.kpm[krow0 + (key < Lk ? key : 0)]);
            km |= masked ? (1u << (4 * j + r)) : 0u;
        }
    const uint64_t seed = p.seed + ((p.p_drop > 0.f && p.seed_ptr) ? *p.seed_ptr : 0ull);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i, row = e >> 3, c4 = (e & 7) * 4;
        *reinterpret_cast<f32x4*>(Ks + row * KP + c4) = kst[i];
        *reinterpret_cast<f32x4*>(Vs + row * KP + c4) = vst[i];
    }
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int gg = 0; gg < 4; gg++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++) acc[ct] = MS_MFMA16(xa[gg][j], wb[ct][gg][j], acc[ct]);
#pragma unroll
    for (int ct = 0; ct < 2; ct++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[(w * 16 + 4 * g + r) * 33 + 16 * ct + li] = acc[ct][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i, row = e >> 5, c = e & 31;
        float v = qbias;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) v += red[(ww * 16 + row) * 33 + c];
        qs[row * KP + c] = v;
        if (row < Lq) p.q[(row0 + row) * MS_D + h * MS_DH + c] = v;
    }
    __syncthreads();

    if (w == 0) {
        const int bh = b * MS_H + h;
        float qf[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) qf[kk] = qs[li * KP + 4 * kk + g] * p.scale;
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * j + li) * KP + g;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) a = MS_MFMA16(kp[4 * kk], qf[kk], a);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                a[r] = ((km >> (4 * j + r)) & 1u) ? -INFINITY : a[r];
                mx = fmaxf(mx, a[r]);
            }
            s[j] = a;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) { const float e = expf(s[j][r] - mx); s[j][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.f / sum;
        const bool qok = li < Lq;
        if (g == 0 && qok) p.lse[(long)bh * Lq + li] = mx + logf(sum);
        const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
        f32x4 oc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int key = 16 * j + 4 * g + t;
                float pv = s[j][t] * inv;
                if (p.p_drop > 0.f) pv *= drop_scale(seed, ((uint64_t)bh * Lq + li) * Lk + key, p.p_drop, inv_keep);   // attn_drop's element index
                const float* vp = Vs + key * KP + li;
#pragma unroll
                for (int c = 0; c < 2; c++) oc[c] = MS_MFMA16(vp[16 * c], pv, oc[c]);
            }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            *reinterpret_cast<f32x4*>(os + li * MS_OP + 16 * c + 4 * g) = oc[c];
            if (qok) *reinterpret_cast<f32x4*>(p.o + (row0 + li) * MS_D + h * MS_DH + 16 * c + 4 * g) = oc[c];
        }
    }
    __syncthreads();

    f32x4 oa[2];
#pragma unroll
    for (int gg = 0; gg < 2; gg++) oa[gg] = *reinterpret_cast<const f32x4*>(os + li * MS_OP + 16 * gg + 4 * g);
    float* yp = p.ypart + ((long)h * p.B * Lq + row0) * MS_D;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        f32x4 y = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int gg = 0; gg < 2; gg++)
#pragma unroll
            for (int j = 0; j < 4; j++) y = MS_MFMA16(oa[gg][j], wo[t][gg][j], y);
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (4 * g + r < Lq) yp[(long)(4 * g + r) * MS_D + 16 * (4 * w + t) + li] = y[r];
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Backward of the self-attention sub-block as ONE launch (the unfused path: out_proj data gradient, ldetr_attention_bwd_f32, in_proj data
// gradient = three dependent launches of 8-15 us each).  Same (sample, head) blocks as the forward:
//   1. dO_h [16 x 32] = dr_b [16 x 256] . W_out[:, 32h : 32h+32], the reduction split over the four waves, partial tiles summed through LDS;
//   2. the head's attention backward with every operand in LDS (the register scheme of attn_bwd_lds_kernel): wave 0 -> dQ, wave 1 -> dK, dV; the
//      probabilities are recomputed from the saved log-sum-exp, the dropout mask from the seed and the forward's element index;
//      dqkv (the operand of dW_in += dqkv^T x) is written once;
//   3. the head's contribution to the input gradient, [dq | dk | dv]_h [16 x 96] . W_in,h [96 x 256] -> dxpart[h][B*L][256]: the consumer (the
//      LayerNorm backward in front, ldetr_layernorm_bwd_group_f32's dy_parts, or ldetr_sum_parts_f32) adds the eight contributions in head order.
// Every global operand is requested before the first MFMA; the W_in slab is read as float4 rows (lane li takes output columns 4 li .. 4 li + 3 of
// its wave's 64: MFMA t of a k-step covers columns {4 li + t}, so a row's accumulators store as one float4).
__global__ __launch_bounds__(256) void mha_small_bwd_kernel(MhaSmallParams pa, MhaSmallParams pb, int nb0) {
    constexpr int KP = MS_DH + 4;
    __shared__ float red[4 * 16 * 33];
    __shared__ __attribute__((aligned(16))) float Qs[16 * KP], Ks[16 * KP], Vs[16 * KP], Ds[16 * KP];
    __shared__ __attribute__((aligned(16))) float Gs[16 * MS_QP];
    __shared__ float delta_s[16], lse_s[16];
    const bool second = (int)blockIdx.x >= nb0;
    const MhaSmallParams& p = second ? pb : pa;
    const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = bid / MS_H, h = bid - b * MS_H, bh = bid;
    const int L = p.L;
    const long row0 = (long)b * L;

    // ---- every global operand of the block, step 3's weights first (they depend on nothing)
    f32x4 wi[24];
#pragma unroll
    for (int kk = 0; kk < 24; kk++) {
        const int c = 4 * kk + g;
        wi[kk] = *reinterpret_cast<const f32x4*>(p.w_in + (long)((c >> 5) * MS_D + h * MS_DH + (c & 31)) * MS_D + 64 * w + 4 * li);
    }
    float wo[2][4][4];
#pragma unroll
    for (int gg = 0; gg < 4; gg++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
                wo[ct][gg][j] = p.w_out[(long)(64 * w + 16 * gg + 4 * g + j) * MS_D + h * MS_DH + 16 * ct + li];
    f32x4 xa[4];
    {
        const bool rok = li < L;
        const float* xr = p.dr + (row0 + (rok ? li : 0)) * MS_D + 64 * w + 4 * g;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 16 * gg);
            xa[gg] = rok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 st[2];        // the head's q | k | v rows: 16 x 3 x 8 float4
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i, row = e / 24, rem = e - row * 24, part = rem >> 3, c4 = (rem & 7) * 4;
        const bool ok = e < 384 && row < L;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.qkv + (row0 + (ok ? row : 0)) * (3 * MS_D) + (ok ? part : 0) * MS_D + h * MS_DH + c4);
        st[i] = ok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 o4 = {0.f, 0.f, 0.f, 0.f};
    if (tid < 128) {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.o + (row0 + (row < L ? row : 0)) * MS_D + h * MS_DH + c4);
        o4 = row < L ? v : o4;
    }
    if (tid < 16) lse_s[tid] = tid < L ? p.lse[(long)bh * L + tid] : 0.f;
    unsigned km = 0;                                   // bit r: key 4 g + r is masked (the dQ wave); bit 4: key li is masked (the dK / dV wave)
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int key = 4 * g + r;
        km |= (key >= L || (p.kpm && p.kpm[row0 + (key < L ? key : 0)])) ? (1u << r) : 0u;
    }
    km |= (li >= L || (p.kpm && p.kpm[row0 + (li < L ? li : 0)])) ? 16u : 0u;
    const uint64_t seed = p.seed + ((p.p_drop > 0.f && p.seed_ptr) ? *p.seed_ptr : 0ull);
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    __builtin_amdgcn_sched_barrier(0);

    // ---- 1. dO_h: this wave's quarter of the reduction over the 256 output features
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int gg = 0; gg < 4; gg++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) acc[ct] = MS_MFMA16(xa[gg][j], wo[ct][gg][j], acc[ct]);
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(w * 16 + 4 * g + r) * 33 + 16 * ct + li] = acc[ct][r];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i, row = e / 24, rem = e - row * 24, part = rem >> 3, c4 = (rem & 7) * 4;
        if (e < 384) *reinterpret_cast<f32x4*>((part == 0 ? Qs : (part == 1 ? Ks : Vs)) + row * KP + c4) = st[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 256 * i, row = e >> 5, c = e & 31;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) v += red[(ww * 16 + row) * 33 + c];
        Ds[row * KP + c] = v;
    }
    __syncthreads();
    if (tid < 128) {       // delta = rowsum(O . dO) of the head
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + row * KP + c4);
        float part = o4[0] * d4[0] + o4[1] * d4[1] + o4[2] * d4[2] + o4[3] * d4[3];
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
        if ((tid & 7) == 0) delta_s[row] = part;
    }
    __syncthreads();

    // ---- 2. attention backward of the (sample, head): one 16 x 16 tile
    if (w == 0) {
        float qf[8], dof[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { qf[kk] = Qs[li * KP + 4 * kk + g] * p.scale; dof[kk] = Ds[li * KP + 4 * kk + g]; }
        const float delta = delta_s[li], lse = lse_s[li];
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            sacc = MS_MFMA16(Ks[li * KP + 4 * kk + g], qf[kk], sacc);
            dpacc = MS_MFMA16(Vs[li * KP + 4 * kk + g], dof[kk], dpacc);
        }
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int key = 4 * g + r;
            const float pr = ((km >> r) & 1u) ? 0.f : expf(sacc[r] - lse);
            float dpv = dpacc[r];
            if (p.p_drop > 0.f) dpv *= drop_scale(seed, ((uint64_t)bh * L + li) * L + key, p.p_drop, inv_keep);
            ds[r] = pr * (dpv - delta) * p.scale;
        }
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int c = 0; c < 2; c++) dq[c] = MS_MFMA16(Ks[(4 * g + t) * KP + 16 * c + li], ds[t], dq[c]);
#pragma unroll
        for (int c = 0; c < 2; c++) {
            *reinterpret_cast<f32x4*>(Gs + li * MS_QP + 16 * c + 4 * g) = dq[c];
            if (li < L && p.dqkv) *reinterpret_cast<f32x4*>(p.dqkv + (row0 + li) * (3 * MS_D) + h * MS_DH + 16 * c + 4 * g) = dq[c];
        }
    } else if (w == 1) {
        const bool kmasked = (km & 16u) != 0;
        float kf[8], vf[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { kf[kk] = Ks[li * KP + 4 * kk + g]; vf[kk] = Vs[li * KP + 4 * kk + g]; }
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            sacc = MS_MFMA16(Qs[li * KP + 4 * kk + g] * p.scale, kf[kk], sacc);
            dpacc = MS_MFMA16(Ds[li * KP + 4 * kk + g], vf[kk], dpacc);
        }
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int qr = 4 * g + r;
            const float pr = (kmasked || qr >= L) ? 0.f : expf(sacc[r] - lse_s[qr]);
            float dm = 1.f;
            if (p.p_drop > 0.f) dm = drop_scale(seed, ((uint64_t)bh * L + qr) * L + li, p.p_drop, inv_keep);
            pd[r] = pr * dm;
            ds[r] = pr * (dpacc[r] * dm - delta_s[qr]) * p.scale;
        }
        f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int qr = 4 * g + t;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                dv[c] = MS_MFMA16(Ds[qr * KP + 16 * c + li], pd[t], dv[c]);
                dk[c] = MS_MFMA16(Qs[qr * KP + 16 * c + li], ds[t], dk[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            *reinterpret_cast<f32x4*>(Gs + li * MS_QP + 32 + 16 * c + 4 * g) = dk[c];
            *reinterpret_cast<f32x4*>(Gs + li * MS_QP + 64 + 16 * c + 4 * g) = dv[c];
            if (li < L && p.dqkv) {
                float* gq = p.dqkv + (row0 + li) * (3 * MS_D) + h * MS_DH + 16 * c + 4 * g;
                *reinterpret_cast<f32x4*>(gq + MS_D) = dk[c];
                *reinterpret_cast<f32x4*>(gq + 2 * MS_D) = dv[c];
            }
        }
    }
    __syncthreads();

    // ---- 3. the head's contribution to the input gradient: this wave's 64 of the 256 input features
    f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < 24; kk++) {
        const float a = Gs[li * MS_QP + 4 * kk + g];
#pragma unroll
        for (int t = 0; t < 4; t++) acc[t] = MS_MFMA16(a, wi[kk][t], acc[t]);
    }
    float* xp = p.dxpart + ((long)h * p.B * L + row0) * MS_D + 64 * w + 4 * li;
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (4 * g + r < L) *reinterpret_cast<f32x4*>(xp + (long)(4 * g + r) * MS_D) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
}

// Backward of the cross-attention sub-block (forward: mha_cross_fwd_kernel) as ONE launch, five waves per (sample, head):
//   1. dO_h = dr_b . W_out[:, head] (waves 0..3, as above) while K_h, V_h ([Lk x 32]) and q_h are staged in LDS;
//   2. wave 4 -> dQ over the key tiles; wave kt < ceil(Lk / 16) -> dK, dV of key tile kt, written straight into the caller's buffers (the grouped
//      memory projection's gradient, pitches lddk / lddv);
//   3. dq_h [16 x 32] . W_q,h [32 x 256] -> dxpart[h][B*Lq][256] (waves 0..3).
__global__ __launch_bounds__(320) void mha_cross_bwd_kernel(MhaCrossParams p) {
    constexpr int KP = MS_DH + 4;
    __shared__ float red[4 * 16 * 33];
    __shared__ __attribute__((aligned(16))) float Qs[16 * KP], Ds[16 * KP], Gq[16 * KP], Ks[64 * KP], Vs[64 * KP];
    __shared__ float delta_s[16], lse_s[16];
    __shared__ unsigned char km_s[64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / MS_H, h = blockIdx.x - b * MS_H, bh = blockIdx.x;
    const int Lq = p.Lq, Lk = p.Lk;
    const long row0 = (long)b * Lq, krow0 = (long)b * Lk;
    const int nkt = (Lk + 15) >> 4;

    f32x4 wq[8];
    float wo[2][4][4];
    f32x4 xa[4];
    if (w < 4) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) wq[kk] = *reinterpret_cast<const f32x4*>(p.w_q + (long)(h * MS_DH + 4 * kk + g) * MS_D + 64 * w + 4 * li);
#pragma unroll
        for (int gg = 0; gg < 4; gg++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++)
                    wo[ct][gg][j] = p.w_out[(long)(64 * w + 16 * gg + 4 * g + j) * MS_D + h * MS_DH + 16 * ct + li];
        const bool rok = li < Lq;
        const float* xr = p.dr + (row0 + (rok ? li : 0)) * MS_D + 64 * w + 4 * g;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xr + 16 * gg);
            xa[gg] = rok ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 kst[2], vst[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 320 * i, row = (e >> 3) & 63, c4 = (e & 7) * 4;
        const bool ok = e < 512 && row < Lk;
        const long r = krow0 + (ok ? row : 0);
        const f32x4 kv = *reinterpret_cast<const f32x4*>(p.k + r * p.ldk + h * MS_DH + c4);
        const f32x4 vv = *reinterpret_cast<const f32x4*>(p.v + r * p.ldv + h * MS_DH + c4);
        kst[i] = ok ? kv : f32x4{0.f, 0.f, 0.f, 0.f}; vst[i] = ok ? vv : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 q4 = {0.f, 0.f, 0.f, 0.f}, o4 = {0.f, 0.f, 0.f, 0.f};
    if (tid < 128) {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const long r = row0 + (row < Lq ? row : 0);
        const f32x4 qv = *reinterpret_cast<const f32x4*>(p.q + r * MS_D + h * MS_DH + c4);
        const f32x4 ov = *reinterpret_cast<const f32x4*>(p.o + r * MS_D + h * MS_DH + c4);
        if (row < Lq) { q4 = qv; o4 = ov; }
    }
    if (tid < 16) lse_s[tid] = tid < Lq ? p.lse[(long)bh * Lq + tid] : 0.f;
    if (tid < 64) km_s[tid] = (tid >= Lk || (p.kpm && p.kpm[krow0 + (tid < Lk ? tid : 0)])) ? 1 : 0;
    const uint64_t seed = p.seed + ((p.p_drop > 0.f && p.seed_ptr) ? *p.seed_ptr : 0ull);
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    __builtin_amdgcn_sched_barrier(0);

    if (w < 4) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int gg = 0; gg < 4; gg++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int ct = 0; ct < 2; ct++) acc[ct] = MS_MFMA16(xa[gg][j], wo[ct][gg][j], acc[ct]);
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[(w * 16 + 4 * g + r) * 33 + 16 * ct + li] = acc[ct][r];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int e = tid + 320 * i, row = (e >> 3) & 63, c4 = (e & 7) * 4;
        if (e < 512) {
            *reinterpret_cast<f32x4*>(Ks + row * KP + c4) = kst[i];
            *reinterpret_cast<f32x4*>(Vs + row * KP + c4) = vst[i];
        }
    }
    if (tid < 128) *reinterpret_cast<f32x4*>(Qs + (tid >> 3) * KP + (tid & 7) * 4) = q4;
    __syncthreads();
    for (int e = tid; e < 512; e += 320) {
        const int row = e >> 5, c = e & 31;
        float v = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ww++) v += red[(ww * 16 + row) * 33 + c];
        Ds[row * KP + c] = v;
    }
    __syncthreads();
    if (tid < 128) {
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + row * KP + c4);
        float part = o4[0] * d4[0] + o4[1] * d4[1] + o4[2] * d4[2] + o4[3] * d4[3];
        part += __shfl_xor(part, 1, 64); part += __shfl_xor(part, 2, 64); part += __shfl_xor(part, 4, 64);
        if ((tid & 7) == 0) delta_s[row] = part;
    }
    __syncthreads();

    if (w == 4) {
        // ---- dQ of the 16 queries over every key tile
        float qf[8], dof[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { qf[kk] = Qs[li * KP + 4 * kk + g] * p.scale; dof[kk] = Ds[li * KP + 4 * kk + g]; }
        const float delta = delta_s[li], lse = lse_s[li];
        f32x4 dq[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        for (int j = 0; j < nkt; j++) {
            f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
            const float* kp = Ks + (16 * j + li) * KP + g;
            const float* vp = Vs + (16 * j + li) * KP + g;
#pragma unroll
            for (int kk = 0; kk < 8; kk++) {
                sacc = MS_MFMA16(kp[4 * kk], qf[kk], sacc);
                dpacc = MS_MFMA16(vp[4 * kk], dof[kk], dpacc);
            }
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int key = 16 * j + 4 * g + r;
                const float pr = km_s[key] ? 0.f : expf(sacc[r] - lse);
                float dpv = dpacc[r];
                if (p.p_drop > 0.f) dpv *= drop_scale(seed, ((uint64_t)bh * Lq + li) * Lk + key, p.p_drop, inv_keep);
                ds[r] = pr * (dpv - delta) * p.scale;
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float* kp2 = Ks + (16 * j + 4 * g + t) * KP + li;
#pragma unroll
                for (int c = 0; c < 2; c++) dq[c] = MS_MFMA16(kp2[16 * c], ds[t], dq[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; c++) {
            *reinterpret_cast<f32x4*>(Gq + li * KP + 16 * c + 4 * g) = dq[c];
            if (li < Lq && p.dq) *reinterpret_cast<f32x4*>(p.dq + (row0 + li) * MS_D + h * MS_DH + 16 * c + 4 * g) = dq[c];
        }
    } else if (w < nkt) {
        // ---- dK, dV of key tile w
        const int krow = 16 * w + li;
        const bool kmasked = km_s[krow] != 0;
        float kf[8], vf[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) { kf[kk] = Ks[krow * KP + 4 * kk + g]; vf[kk] = Vs[krow * KP + 4 * kk + g]; }
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dpacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            sacc = MS_MFMA16(Qs[li * KP + 4 * kk + g] * p.scale, kf[kk], sacc);
            dpacc = MS_MFMA16(Ds[li * KP + 4 * kk + g], vf[kk], dpacc);
        }
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int qr = 4 * g + r;
            const float pr = (kmasked || qr >= Lq) ? 0.f : expf(sacc[r] - lse_s[qr]);
            float dm = 1.f;
            if (p.p_drop > 0.f) dm = drop_scale(seed, ((uint64_t)bh * Lq + qr) * Lk + krow, p.p_drop, inv_keep);
            pd[r] = pr * dm;
            ds[r] = pr * (dpacc[r] * dm - delta_s[qr]) * p.scale;
        }
        f32x4 dk[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, dv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int qr = 4 * g + t;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                dv[c] = MS_MFMA16(Ds[qr * KP + 16 * c + li], pd[t], dv[c]);
                dk[c] = MS_MFMA16(Qs[qr * KP + 16 * c + li], ds[t], dk[c]);
            }
        }
        if (krow < Lk) {
            float* dkp = p.dk + (krow0 + krow) * p.lddk + h * MS_DH + 4 * g;
            float* dvp = p.dv + (krow0 + krow) * p.lddv + h * MS_DH + 4 * g;
#pragma unroll
            for (int c = 0; c < 2; c++) {
                *reinterpret_cast<f32x4*>(dkp + 16 * c) = dk[c];
                *reinterpret_cast<f32x4*>(dvp + 16 * c) = dv[c];
            }
        }
    }
    __syncthreads();

    if (w < 4) {
        f32x4 acc[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            const float a = Gq[li * KP + 4 * kk + g];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = MS_MFMA16(a, wq[kk][t], acc[t]);
        }
        float* xp = p.dxpart + ((long)h * p.B * Lq + row0) * MS_D + 64 * w + 4 * li;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (4 * g + r < Lq) *reinterpret_cast<f32x4*>(xp + (long)(4 * g + r) * MS_D) = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    }
}

// Weight gradients of a transformer layer as ONE launch: up to 8 independent contractions dW [rows x cols] += A^T B over the tokens (K = M tokens:
// 144..320 on the short stacks), each a grid of 32 x 32 tiles; a block's four waves split the tokens, stream their operand fragments global ->
// registers (both operands are token-major: the lane's 16 values of a chunk are 16 strided dwords; out-of-range tokens / features read as zero through
// the buffer descriptor's bound), and sum their partial tiles through LDS in wave order.  One writer per output element: deterministic.
struct WgradMulti { ldetr_wgrad_desc d[8]; int start[9]; int n; };

__global__ __launch_bounds__(256) void wgrad_multi_kernel(WgradMulti m) {
    using acc_t = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    __shared__ float red[4][32][33];
    __shared__ float rsum[4][32];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) pi += (i < m.n && (int)blockIdx.x >= m.start[i]) ? 1 : 0;
    const ldetr_wgrad_desc& d = m.d[pi];
    const int bid = (int)blockIdx.x - m.start[pi];
    const int gx = (d.cols + 31) >> 5;
    const int by = bid / gx, bx = bid - by * gx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kl = lane >> 5;
    const int m0 = by * 32, n0 = bx * 32;
    // tokens of this wave: units of 16, dealt evenly
    const int units = (d.M + 15) >> 4, ubeg = wave * units / 4, uend = (wave + 1) * units / 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.A), 0, (int)((long)d.M * d.lda * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.B), 0, (int)((long)d.M * d.ldb * 4), 0x00020000);
    const int OOB = (int)0x80000000;
    const int lda4 = (int)d.lda * 4, ldb4 = (int)d.ldb * 4;
    const int voffA = (m0 + cl < d.rows) ? (8 * kl * (int)d.lda + m0 + cl) * 4 : OOB;
    const int voffB = (n0 + cl < d.cols) ? (8 * kl * (int)d.ldb + n0 + cl) * 4 : OOB;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    const bool want_rs = d.db != nullptr && bx == 0;
    float rs = 0.f;
    for (int u0 = ubeg; u0 < uend; u0 += 4) {
        float a[4][8], bq[4][8];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const bool on = u0 + q < uend;
            const int soff = (u0 + q) * 16;
#pragma unroll
            for (int t = 0; t < 8; t++) {
                a[q][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsA, on ? voffA : OOB, (soff + t) * lda4, 0));
                bq[q][t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsB, on ? voffB : OOB, (soff + t) * ldb4, 0));
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int t = 0; t < 8; t++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][t], bq[q][t], acc, 0, 0, 0);
                rs += a[q][t];
            }
    }
    if (want_rs) {
        rs += __shfl_xor(rs, 32);
        if (kl == 0) rsum[wave][cl] = rs;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kl][cl] = acc[r];
    __syncthreads();
    if (want_rs && tid < 32 && m0 + tid < d.rows) d.db[m0 + tid] += rsum[0][tid] + rsum[1][tid] + rsum[2][tid] + rsum[3][tid];
    const bool vec = (d.ldw & 3) == 0 && ((((uintptr_t)d.dW) & 15) == 0);
    for (int q = tid; q < 256; q += 256) {
        const int row = q >> 3, c4 = (q & 7) * 4;
        if (m0 + row >= d.rows) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = red[0][row][c4 + e] + red[1][row][c4 + e] + red[2][row][c4 + e] + red[3][row][c4 + e];
        float* dst = d.dW + (long)(m0 + row) * d.ldw + n0 + c4;
        if (vec && n0 + c4 + 3 < d.cols) {
            float4 c = *reinterpret_cast<float4*>(dst);
            c.x += v[0]; c.y += v[1]; c.z += v[2]; c.w += v[3];
            *reinterpret_cast<float4*>(dst) = c;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (n0 + c4 + e < d.cols) dst[e] += v[e];
        }
    }
}

}  // namespace ldetr

using namespace ldetr;

static int mha_small_check(const char* what, const MhaSmallParams& p, bool bwd) {
    LDETR_CHECK(p.L >= 1 && p.L <= 16 && p.B >= 0, "%s: 1 <= L <= 16 tokens per sample", what);
    LDETR_CHECK(p.w_in && p.w_out && p.qkv && p.o && p.lse, "%s: null pointer", what);
    LDETR_CHECK(p.p_drop >= 0.f && p.p_drop < 1.f, "%s: p_drop out of range", what);
    if (!bwd) {
        LDETR_CHECK(p.x && p.b_in && p.ypart, "%s: null pointer", what);
        LDETR_CHECK(p.ldx >= MS_D && p.ldx % 4 == 0, "%s: row pitch of x must be a multiple of 4 floats, >= 256", what);
        LDETR_CHECK((((uintptr_t)p.x | (uintptr_t)p.w_in | (uintptr_t)p.w_out | (uintptr_t)p.o | (uintptr_t)p.qkv | (uintptr_t)p.ypart) & 15) == 0,
                    "%s: buffers must be 16-byte aligned", what);
    } else {
        LDETR_CHECK(p.dr && p.dxpart, "%s: null pointer", what);
        LDETR_CHECK((((uintptr_t)p.dr | (uintptr_t)p.w_in | (uintptr_t)p.w_out | (uintptr_t)p.o | (uintptr_t)p.qkv | (uintptr_t)p.dqkv | (uintptr_t)p.dxpart) & 15) == 0,
                    "%s: buffers must be 16-byte aligned", what);
    }
    return LDETR_OK;
}

static int mha_small_launch(const ldetr_mha_small_args* a, int n, bool bwd, void* stream) {
    const char* what = bwd ? "mha_small_bwd" : "mha_small_fwd";
    LDETR_CHECK(a && (n == 1 || n == 2), "%s: 1 or 2 problems", what);
    MhaSmallParams p[2]; p[0] = a[0]; p[1] = n == 2 ? a[1] : a[0];
    int nb[2] = {0, 0};
    for (int i = 0; i < n; i++) {
        if (int rc = mha_small_check(what, p[i], bwd)) return rc;
        nb[i] = p[i].B * MS_H;
    }
    if (nb[0] + nb[1] == 0) return LDETR_OK;
    if (bwd) hipLaunchKernelGGL(mha_small_bwd_kernel, dim3((unsigned)(nb[0] + nb[1])), dim3(256), 0, (hipStream_t)stream, p[0], p[1], nb[0]);
    else hipLaunchKernelGGL(mha_small_fwd_kernel, dim3((unsigned)(nb[0] + nb[1])), dim3(256), 0, (hipStream_t)stream, p[0], p[1], nb[0]);
    note_engine_launch(false);
    return check_launch(what);
}

extern "C" int ldetr_mha_small_fwd_group_f32(const ldetr_mha_small_args* a, int n, void* stream) { return mha_small_launch(a, n, false, stream); }
extern "C" int ldetr_mha_small_bwd_group_f32(const ldetr_mha_small_args* a, int n, void* stream) { return mha_small_launch(a, n, true, stream); }

extern "C" int ldetr_mha_small_fwd_f32(const float* x, int64_t ldx, const float* w_in, const float* b_in, const float* w_out,
                                       const uint8_t* kpm, float* qkv, float* o, float* lse, float* ypart,
                                       int B, int L, int D, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                       void* stream) {
    LDETR_CHECK(D == MS_D && H == MS_H, "mha_small_fwd: d_model must be 256 with 8 heads");
    MhaSmallParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = ldx; p.w_in = w_in; p.b_in = b_in; p.w_out = w_out; p.kpm = kpm; p.qkv = qkv; p.o = o; p.lse = lse; p.ypart = ypart;
    p.B = B; p.L = L; p.scale = scale; p.p_drop = p_drop; p.seed = seed; p.seed_ptr = seed_ptr;
    return mha_small_launch(&p, 1, false, stream);
}

extern "C" int ldetr_mha_cross_bwd_f32(const ldetr_mha_cross_args* a, void* stream) {
    LDETR_CHECK(a != nullptr, "mha_cross_bwd: null argument block");
    const MhaCrossParams& p = *a;
    LDETR_CHECK(p.Lq >= 1 && p.Lq <= 16 && p.Lk >= 1 && p.Lk <= 64 && p.B >= 0, "mha_cross_bwd: 1 <= Lq <= 16 queries and 1 <= Lk <= 64 keys per sample");
    LDETR_CHECK(p.dr && p.w_q && p.k && p.v && p.w_out && p.q && p.o && p.lse && p.dk && p.dv && p.dxpart, "mha_cross_bwd: null pointer");
    LDETR_CHECK(p.ldk % 4 == 0 && p.ldv % 4 == 0 && p.lddk % 4 == 0 && p.lddv % 4 == 0, "mha_cross_bwd: row pitches must be multiples of 4 floats");
    LDETR_CHECK((((uintptr_t)p.dr | (uintptr_t)p.w_q | (uintptr_t)p.w_out | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.o | (uintptr_t)p.q | (uintptr_t)p.dq |
                  (uintptr_t)p.dk | (uintptr_t)p.dv | (uintptr_t)p.dxpart) & 15) == 0, "mha_cross_bwd: buffers must be 16-byte aligned");
    LDETR_CHECK(p.p_drop >= 0.f && p.p_drop < 1.f, "mha_cross_bwd: p_drop out of range");
    if (p.B == 0) return LDETR_OK;
    hipLaunchKernelGGL(mha_cross_bwd_kernel, dim3((unsigned)(p.B * MS_H)), dim3(320), 0, (hipStream_t)stream, p);
    note_engine_launch(false);
    return check_launch("mha_cross_bwd");
}

extern "C" int ldetr_wgrad_multi_f32(const ldetr_wgrad_desc* d, int n, void* stream) {
    LDETR_CHECK(d && n >= 1 && n <= 8, "wgrad_multi: 1..8 problems");
    WgradMulti m; memset(&m, 0, sizeof(m));
    int total = 0;
    for (int i = 0; i < n; i++) {
        const ldetr_wgrad_desc& q = d[i];
        LDETR_CHECK(q.A && q.B && q.dW && q.M >= 0 && q.rows > 0 && q.cols > 0 && q.lda >= q.rows && q.ldb >= q.cols && q.ldw >= q.cols, "wgrad_multi: bad problem %d", i);
        LDETR_CHECK((long)q.M * q.lda * 4 < 0x7fffffffL && (long)q.M * q.ldb * 4 < 0x7fffffffL, "wgrad_multi: operand above 2 GiB");
        m.d[i] = q; m.start[i] = total;
        total += q.M > 0 ? cdiv(q.rows, 32) * cdiv(q.cols, 32) : 0;
    }
    for (int i = n; i <= 8; i++) m.start[i] = total;
    m.n = n;
    if (total == 0) return LDETR_OK;
    hipLaunchKernelGGL(wgrad_multi_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, m);
    note_engine_launch(false);
    return check_launch("wgrad_multi");
}

extern "C" int ldetr_mha_cross_fwd_f32(const float* x, int64_t ldx, const float* w_q, const float* b_q,
                                       const float* k, int64_t ldk, const float* v, int64_t ldv, const float* w_out,
                                       const uint8_t* kpm, float* q, float* o, float* lse, float* ypart,
                                       int B, int Lq, int Lk, int D, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                       void* stream) {
    LDETR_CHECK(D == MS_D && H == MS_H, "mha_cross_fwd: d_model must be 256 with 8 heads");
    LDETR_CHECK(Lq >= 1 && Lq <= 16 && Lk >= 1 && Lk <= 64 && B >= 0, "mha_cross_fwd: 1 <= Lq <= 16 queries and 1 <= Lk <= 64 keys per sample");
    LDETR_CHECK(x && w_q && b_q && k && v && w_out && q && o && lse && ypart, "mha_cross_fwd: null pointer");
    LDETR_CHECK(ldx >= MS_D && ldx % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, "mha_cross_fwd: row pitches must be multiples of 4 floats");
    LDETR_CHECK((((uintptr_t)x | (uintptr_t)w_q | (uintptr_t)w_out | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)q | (uintptr_t)ypart) & 15) == 0,
                "mha_cross_fwd: buffers must be 16-byte aligned");
    LDETR_CHECK(p_drop >= 0.f && p_drop < 1.f, "mha_cross_fwd: p_drop out of range");
    if (B == 0) return LDETR_OK;
    MhaCrossParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = ldx; p.w_q = w_q; p.b_q = b_q; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.w_out = w_out; p.kpm = kpm;
    p.q = q; p.o = o; p.lse = lse; p.ypart = ypart; p.B = B; p.Lq = Lq; p.Lk = Lk; p.scale = scale; p.p_drop = p_drop;
    p.seed = seed; p.seed_ptr = seed_ptr;
    hipLaunchKernelGGL(mha_cross_fwd_kernel, dim3((unsigned)(B * MS_H)), dim3(256), 0, (hipStream_t)stream, p);
    note_engine_launch(false);
    return check_launch("mha_cross_fwd");
}
