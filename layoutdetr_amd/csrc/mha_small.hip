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

struct MhaSmallParams {
    const float* x; long ldx;                       // [B*L, 256]
    const float* w_in; const float* b_in;           // [768, 256], [768]
    const float* w_out;                             // [256, 256]
    const unsigned char* kpm;                       // [B][L], nonzero = masked key, or null
    float* qkv;                                     // [B*L, 768] projection incl. bias (q unscaled), saved for the backward
    float* o;                                       // [B*L, 256] attention output before the output projection, saved for the backward
    float* lse;                                     // [B][8][L]
    float* ypart;                                   // [8][B*L][256]
    int B, L;
    float scale, p_drop;
    unsigned long long seed; const unsigned long long* seed_ptr;
};

constexpr int MS_D = 256, MS_H = 8, MS_DH = 32;
constexpr int MS_RP = 97, MS_QP = 100, MS_OP = 36;   // LDS pitches (floats): partial tiles, the head's q|k|v, the head's output

#define MS_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(256) void mha_small_fwd_kernel(MhaSmallParams p) {
    __shared__ float red[4 * 16 * MS_RP];
    __shared__ __attribute__((aligned(16))) float qs[16 * MS_QP];
    __shared__ __attribute__((aligned(16))) float os[16 * MS_OP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / MS_H, h = blockIdx.x - b * MS_H;
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
    const unsigned long long seed = p.seed + ((p.p_drop > 0.f && p.seed_ptr) ? *p.seed_ptr : 0ull);
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
struct MhaCrossParams {
    const float* x; long ldx;                       // [B*Lq, 256] queries (tgt after norm1)
    const float* w_q; const float* b_q;             // rows 0..255 of in_proj_weight / in_proj_bias
    const float* k; long ldk; const float* v; long ldv;   // [B*Lk, >= 256] projected memory
    const float* w_out;
    const unsigned char* kpm;                       // [B][Lk] or null
    float* q;                                       // [B*Lq, 256] projected queries incl. bias (unscaled), saved for the backward
    float* o; float* lse; float* ypart;             // as in MhaSmallParams ([B*Lq, 256], [B][8][Lq], [8][B*Lq][256])
    int B, Lq, Lk;
    float scale, p_drop;
    unsigned long long seed; const unsigned long long* seed_ptr;
};

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
    const unsigned long long seed = p.seed + ((p.p_drop > 0.f && p.seed_ptr) ? *p.seed_ptr : 0ull);
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

}  // namespace ldetr

using namespace ldetr;

extern "C" int ldetr_mha_small_fwd_f32(const float* x, int64_t ldx, const float* w_in, const float* b_in, const float* w_out,
                                       const uint8_t* kpm, float* qkv, float* o, float* lse, float* ypart,
                                       int B, int L, int D, int H, float scale, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                       void* stream) {
    LDETR_CHECK(D == MS_D && H == MS_H, "mha_small_fwd: d_model must be 256 with 8 heads");
    LDETR_CHECK(L >= 1 && L <= 16 && B >= 0, "mha_small_fwd: 1 <= L <= 16 tokens per sample");
    LDETR_CHECK(x && w_in && b_in && w_out && qkv && o && lse && ypart, "mha_small_fwd: null pointer");
    LDETR_CHECK(ldx >= MS_D && ldx % 4 == 0, "mha_small_fwd: row pitch of x must be a multiple of 4 floats, >= 256");
    LDETR_CHECK((((uintptr_t)x | (uintptr_t)w_in | (uintptr_t)w_out | (uintptr_t)o | (uintptr_t)qkv | (uintptr_t)ypart) & 15) == 0,
                "mha_small_fwd: buffers must be 16-byte aligned");
    LDETR_CHECK(p_drop >= 0.f && p_drop < 1.f, "mha_small_fwd: p_drop out of range");
    if (B == 0) return LDETR_OK;
    MhaSmallParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = ldx; p.w_in = w_in; p.b_in = b_in; p.w_out = w_out; p.kpm = kpm; p.qkv = qkv; p.o = o; p.lse = lse; p.ypart = ypart;
    p.B = B; p.L = L; p.scale = scale; p.p_drop = p_drop; p.seed = seed; p.seed_ptr = (const unsigned long long*)seed_ptr;
    hipLaunchKernelGGL(mha_small_fwd_kernel, dim3((unsigned)(B * MS_H)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("mha_small_fwd");
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
    p.seed = seed; p.seed_ptr = (const unsigned long long*)seed_ptr;
    hipLaunchKernelGGL(mha_cross_fwd_kernel, dim3((unsigned)(B * MS_H)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("mha_cross_fwd");
}
