// Layout losses of the generator phase, fused: on one [B, N, 4] set of generated boxes (xc, yc, w, h) the
// reference evaluates, each as a chain of ~40 elementwise / reduction launches plus as many in the backward,
//     F.mse_loss(bbox_fake[valid], bbox_real[valid])                       training/loss.py:94
//     generalized_iou_loss(bbox_fake[valid], bbox_real[valid])             metrics/metric_layoutnet.py:245-275
//     compute_overlap(bbox_fake, valid)                                    metrics/metric_layoutnet.py:153-179
//     compute_alignment(bbox_fake, valid)                                  metrics/metric_layoutnet.py:182-201
// Here one wave per sample computes the four values AND their gradients with respect to bbox_fake in the same pass (N <= 64
// boxes, one lane each: everything lives in registers / 2 KiB of LDS); the backward is a 4-term weighted sum of the saved gradients.
// Subgradient conventions are autograd's: maximum / minimum split a tie half-half, where() passes the gradient of the taken
// branch only, abs' = sign, min(dim) routes to the first arg-min, nan_to_num blocks the gradient of the entries it replaced, and
// compute_alignment keeps its quirk of comparing valid boxes against padded ones too (the mask is applied to rows only).
//   losses [4][B]: per-sample shares, so that  mse = losses[0].sum(), gIoU = losses[1].sum(), overlap = losses[2] ([B]),
//                  alignment = losses[3] ([B]).   grads [4][B][N][4] = d losses[t][b] / d bbox[b].
#include <algorithm>
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct LayoutLossParams {
    const float* box; const float* ref; const unsigned char* valid;
    float* losses; float* grads;
    const float* gout; float* dbox;
    int B, N;
};

struct Ltrb { float l, t, r, b; };
__device__ __forceinline__ Ltrb to_ltrb(const float (&x)[4]) { return {x[0] - x[2] / 2, x[1] - x[3] / 2, x[0] + x[2] / 2, x[1] + x[3] / 2}; }
// d max(a, b) / da and d min(a, b) / da with autograd's tie rule
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }
// gradient on (l, t, r, b) -> gradient on (xc, yc, w, h)
__device__ __forceinline__ void ltrb_to_xywh(float dl, float dt, float dr, float db, float s, float (&g)[4]) {
    g[0] += s * (dl + dr); g[1] += s * (dt + db); g[2] += s * (dr - dl) * 0.5f; g[3] += s * (db - dt) * 0.5f;
}

__global__ __launch_bounds__(64) void layout_losses_kernel(LayoutLossParams p) {
    __shared__ float sb[64][4];
    __shared__ int sv[64];
    __shared__ int aj[64], ac[64];
    __shared__ float as_[64];
    const int b = blockIdx.x, k = threadIdx.x, N = p.N;
    // number of valid boxes in the whole batch (mse / gIoU are means over all valid boxes) and in this sample
    int tot = 0;
    for (int i = k; i < p.B * N; i += 64) tot += p.valid[i] ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
    const float cnt = tot > 0 ? (float)tot : 1.f;
    float a[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f};
    bool vk = false;
    if (k < N) {
        vk = p.valid[b * N + k] != 0;
#pragma unroll
        for (int c = 0; c < 4; c++) { a[c] = p.box[((long)b * N + k) * 4 + c]; r[c] = p.ref ? p.ref[((long)b * N + k) * 4 + c] : 0.f; sb[k][c] = a[c]; }
        sv[k] = vk;
        aj[k] = -1; ac[k] = 0; as_[k] = 0.f;
    }
    __syncthreads();
    int nb = 0;
    for (int j = 0; j < N; j++) nb += sv[j];
    const float inv_nb = 1.f / (float)nb;     // a sample without valid boxes gives 0 / 0 in the reference as well

    float l_mse = 0.f, l_giou = 0.f, l_ovl = 0.f, l_aln = 0.f;
    float g_mse[4] = {0.f, 0.f, 0.f, 0.f}, g_giou[4] = {0.f, 0.f, 0.f, 0.f}, g_ovl[4] = {0.f, 0.f, 0.f, 0.f}, g_aln[4] = {0.f, 0.f, 0.f, 0.f};
    if (k < N && vk) {
        const Ltrb A = to_ltrb(a);
        const float wA = A.r - A.l, hA = A.b - A.t, a1 = wA * hA;
        if (p.ref) {
            // ---- MSE: sum_c (a - r)^2 / (4 cnt)
#pragma unroll
            for (int c = 0; c < 4; c++) { const float d = a[c] - r[c]; l_mse += d * d; g_mse[c] = 2.f * d / (4.f * cnt); }
            l_mse /= 4.f * cnt;
            // ---- gIoU: per = 2 - ai / au - au / ah, mean over valid boxes
            const Ltrb R = to_ltrb(r);
            const float a2 = (R.r - R.l) * (R.b - R.t);
            const float lmx = fmaxf(A.l, R.l), rmn = fminf(A.r, R.r), tmx = fmaxf(A.t, R.t), bmn = fminf(A.b, R.b);
            const bool cond = (lmx < rmn) && (tmx < bmn);
            const float W = rmn - lmx, H = bmn - tmx, ai = cond ? W * H : 0.f;
            const float au = a1 + a2 - ai;
            const float lmn = fminf(A.l, R.l), rmx = fmaxf(A.r, R.r), tmn = fminf(A.t, R.t), bmx = fmaxf(A.b, R.b);
            const float Wh = rmx - lmn, Hh = bmx - tmn, ah = Wh * Hh;
            l_giou = (1.f - (ai / au - (ah - au) / ah)) / cnt;
            // derivatives on (l, t, r, b) of the generated box
            float dai[4] = {0.f, 0.f, 0.f, 0.f};
            if (cond) { dai[0] = -dmax_a(A.l, R.l) * H; dai[1] = -dmax_a(A.t, R.t) * W; dai[2] = dmin_a(A.r, R.r) * H; dai[3] = dmin_a(A.b, R.b) * W; }
            const float da1[4] = {-hA, -wA, hA, wA};
            const float dah[4] = {-dmin_a(A.l, R.l) * Hh, -dmin_a(A.t, R.t) * Wh, dmax_a(A.r, R.r) * Hh, dmax_a(A.b, R.b) * Wh};
            float dper[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float dau = da1[c] - dai[c];
                const float diou = (dai[c] * au - ai * dau) / (au * au);
                const float dq = (dau * ah - au * dah[c]) / (ah * ah);     // d (au / ah)
                dper[c] = -diou - dq;
            }
            ltrb_to_xywh(dper[0], dper[1], dper[2], dper[3], 1.f / cnt, g_giou);
        }
        // ---- overlap: sum_{j != k} ai_kj / a1_k / n_b  (as box i) and the terms ai_jk / a1_j where k is box j
        float dl = 0.f, dt = 0.f, dr = 0.f, db = 0.f, sum_ai = 0.f;
        for (int j = 0; j < N; j++) {
            if (j == k || !sv[j]) continue;
            const float x[4] = {sb[j][0], sb[j][1], sb[j][2], sb[j][3]};
            const Ltrb J = to_ltrb(x);
            const float a1j = (J.r - J.l) * (J.b - J.t);
            const float lmx = fmaxf(A.l, J.l), rmn = fminf(A.r, J.r), tmx = fmaxf(A.t, J.t), bmn = fminf(A.b, J.b);
            if (!((lmx < rmn) && (tmx < bmn))) continue;
            const float W = rmn - lmx, H = bmn - tmx, ai = W * H;
            // nan_to_num(ai / a1): a zero-area box contributes nothing and passes no gradient
            const float wk = a1 != 0.f ? 1.f / a1 : 0.f, wj = a1j != 0.f ? 1.f / a1j : 0.f;
            if (a1 != 0.f) sum_ai += ai;
            const float ws = wk + wj;
            dl += -dmax_a(A.l, J.l) * H * ws; dt += -dmax_a(A.t, J.t) * W * ws;
            dr += dmin_a(A.r, J.r) * H * ws; db += dmin_a(A.b, J.b) * W * ws;
        }
        if (a1 != 0.f) {
            l_ovl = sum_ai / a1 * inv_nb;
            const float q = -sum_ai / (a1 * a1);
            dl += q * -hA; dt += q * -wA; dr += q * hA; db += q * wA;
        }
        ltrb_to_xywh(dl, dt, dr, db, inv_nb, g_ovl);
        // ---- alignment: min over the six edge / centre coordinates and over the OTHER boxes (padded ones included) of |delta|
        const float X[6] = {A.l, a[0], A.r, A.t, a[1], A.b};
        float best = 0.f; int bj = -1, bc = 0; float bsign = 0.f; bool have = false;
        for (int c = 0; c < 6; c++) {
            float m = 1.f; int mj = k; float ms = 0.f;                      // the diagonal entry is 1 (index j = k)
            bool first = true;
            for (int j = 0; j < N; j++) {
                float v, sg;
                if (j == k) { v = 1.f; sg = 0.f; }
                else {
                    const float x[4] = {sb[j][0], sb[j][1], sb[j][2], sb[j][3]};
                    const Ltrb J = to_ltrb(x);
                    const float Xj = c == 0 ? J.l : c == 1 ? x[0] : c == 2 ? J.r : c == 3 ? J.t : c == 4 ? x[1] : J.b;
                    const float d = X[c] - Xj;
                    v = fabsf(d); sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                }
                if (first || v < m) { m = v; mj = j; ms = sg; first = false; }
            }
            if (!have || m < best) { best = m; bj = mj; bc = c; bsign = ms; have = true; }
        }
        if (best != 1.f) {
            l_aln = -logf(1.f - best) * inv_nb;
            const float s = bsign / (1.f - best) * inv_nb;
            if (bj != k) { aj[k] = bj; ac[k] = bc; as_[k] = s; }
            // own coordinate bc: xl, xc, xr, yt, yc, yb
            if (bc == 0) { g_aln[0] += s; g_aln[2] -= 0.5f * s; } else if (bc == 1) g_aln[0] += s; else if (bc == 2) { g_aln[0] += s; g_aln[2] += 0.5f * s; }
            else if (bc == 3) { g_aln[1] += s; g_aln[3] -= 0.5f * s; } else if (bc == 4) g_aln[1] += s; else { g_aln[1] += s; g_aln[3] += 0.5f * s; }
        }
    }
    __syncthreads();
    if (k < N) {
        // alignment: boxes that were some other box's nearest neighbour receive the opposite gradient (padded boxes too)
        for (int i = 0; i < N; i++) {
            if (aj[i] != k) continue;
            const float s = -as_[i]; const int bc = ac[i];
            if (bc == 0) { g_aln[0] += s; g_aln[2] -= 0.5f * s; } else if (bc == 1) g_aln[0] += s; else if (bc == 2) { g_aln[0] += s; g_aln[2] += 0.5f * s; }
            else if (bc == 3) { g_aln[1] += s; g_aln[3] -= 0.5f * s; } else if (bc == 4) g_aln[1] += s; else { g_aln[1] += s; g_aln[3] += 0.5f * s; }
        }
        const long per = (long)p.B * N * 4, o = ((long)b * N + k) * 4;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            p.grads[o + c] = g_mse[c]; p.grads[per + o + c] = g_giou[c]; p.grads[2 * per + o + c] = g_ovl[c]; p.grads[3 * per + o + c] = g_aln[c];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        l_mse += __shfl_xor(l_mse, o, 64); l_giou += __shfl_xor(l_giou, o, 64);
        l_ovl += __shfl_xor(l_ovl, o, 64); l_aln += __shfl_xor(l_aln, o, 64);
    }
    if (k == 0) { p.losses[b] = l_mse; p.losses[p.B + b] = l_giou; p.losses[2 * p.B + b] = l_ovl; p.losses[3 * p.B + b] = l_aln; }
}

__global__ __launch_bounds__(256) void layout_losses_bwd_kernel(LayoutLossParams p) {
    const long per = (long)p.B * p.N * 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per; i += (long)gridDim.x * 256) {
        const int b = (int)(i / (p.N * 4));
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++) s += p.gout[t * p.B + b] * p.grads[t * per + i];
        p.dbox[i] = s;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// The tail of a loss phase as ONE launch per direction.  StyleGAN2Loss.accumulate_gradients (training/loss.py:84-116, 146-218) forms each phase's loss
// as sum_k w_k * term_k over ~10 terms -- softplus(+-logits) [B], per-sample layout terms [B], scalar reconstruction / cross-entropy terms -- then
// .mean().mul(gain).backward(): ~30 scalar-sized ATen launches forward and ~40 backward per phase.  Here: total = sum_k w_k * c_k * sum_i f_k(x_k[i])
// (c_k = 1 / n_k for a per-sample term that the reference averages, 1 for a term given as per-sample contributions to a sum, f_k in {x, softplus(x),
// softplus(-x)}; a `ratio` term is x[0] / x[1]: the cross-entropy kernel's (loss sum, count) pair), the weighted per-element values for the statistics
// the reference reports (training_stats.report), and in the backward every term's gradient in one pass.
struct LossCombineParams {
    const float* x[16]; float w[16]; int n[16]; int fn[16]; int red[16];
    int K, ld;
    float* vals; float* sums; float* total;     // [K][ld] weighted values, [K] their sums, [1]
    const float* g; float* grads;               // backward: d total, [K][ld]
};

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }        // F.softplus(beta 1, threshold 20)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void loss_combine_fwd_kernel(LossCombineParams p) {
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float total = 0.f;
    for (int k = 0; k < p.K; k++) {
        const float w = p.w[k];
        if (p.fn[k] == 3) {
            const float v = p.x[k][0] / p.x[k][1] * w;
            if (tid == 0) { p.vals[(long)k * p.ld] = v; p.sums[k] = v; }
            total += v;
            continue;
        }
        float part = 0.f;
        for (int i = tid; i < p.n[k]; i += 256) {
            const float x = p.x[k][i];
            const float v = (p.fn[k] == 1 ? softplus_f(x) : (p.fn[k] == 2 ? softplus_f(-x) : x)) * w;
            p.vals[(long)k * p.ld + i] = v;
            part += v;
        }
        part = wave_sum(part);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = part;
        __syncthreads();
        const float s = red[0] + red[1] + red[2] + red[3];
        if (tid == 0) p.sums[k] = s;
        total += s * (p.red[k] ? 1.f : 1.f / (float)p.n[k]);
    }
    if (tid == 0) p.total[0] = total;
}

__global__ __launch_bounds__(256) void loss_combine_bwd_kernel(LossCombineParams p) {
    const float g = p.g[0];
    for (int k = blockIdx.x; k < p.K; k += gridDim.x) {
        const float w = p.w[k];
        if (p.fn[k] == 3) {       // the cross-entropy backward divides by its count itself
            if (threadIdx.x == 0) { p.grads[(long)k * p.ld] = g * w; p.grads[(long)k * p.ld + 1] = 0.f; }
            continue;
        }
        const float c = g * w * (p.red[k] ? 1.f : 1.f / (float)p.n[k]);
        for (int i = threadIdx.x; i < p.n[k]; i += 256) {
            const float x = p.x[k][i];
            float d = 1.f;
            if (p.fn[k] == 1) d = x > 20.f ? 1.f : sigmoid_f(x);
            else if (p.fn[k] == 2) d = -x > 20.f ? -1.f : -sigmoid_f(-x);
            p.grads[(long)k * p.ld + i] = c * d;
        }
    }
}

// F.mse_loss(a[valid], b[valid]) without the gather (the static-shape heads: networks_detr.py:314 / loss.py:240, 245): a [rows][D], b [rows / bdiv][D]
// (bdiv > 1: one reference row per group of bdiv rows, the per-sample z of loss_z), valid [rows] -> out[0] = sum over valid rows of |a - b|^2 /
// (max(count, 1) * D), out[1] = count.  One block (rows are slots of a batch: a few hundred).
struct MaskedMseParams { const float* a; const float* b; const unsigned char* valid; long rows; int D, bdiv; float* out; const float* g; float* da; };

__global__ __launch_bounds__(256) void masked_mse_fwd_kernel(MaskedMseParams p) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x;
    float s = 0.f, c = 0.f;
    const long n = p.rows * p.D;
    for (long i = tid; i < n; i += 256) {
        const long row = i / p.D; const int d = (int)(i - row * p.D);
        if (p.valid[row]) {
            const float e = p.a[i] - p.b[(row / p.bdiv) * p.D + d];
            s += e * e;
            c += d == 0 ? 1.f : 0.f;
        }
    }
    s = wave_sum(s); c = wave_sum(c);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s; red[1][tid >> 6] = c; }
    __syncthreads();
    if (tid == 0) {
        const float cnt = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        p.out[0] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (fmaxf(cnt, 1.f) * (float)p.D);
        p.out[1] = cnt;
    }
}

__global__ __launch_bounds__(256) void masked_mse_bwd_kernel(MaskedMseParams p) {
    const long n = p.rows * p.D;
    const float c = 2.f * p.g[0] / (fmaxf(p.out[1], 1.f) * (float)p.D);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / p.D; const int d = (int)(i - row * p.D);
        p.da[i] = p.valid[row] ? c * (p.a[i] - p.b[(row / p.bdiv) * p.D + d]) : 0.f;
    }
}

}  // namespace ldetr

using namespace ldetr;

extern "C" int ldetr_layout_losses_f32(const float* bbox, const float* bbox_ref, const uint8_t* valid, int B, int N, float* losses,
                                       float* grads, void* stream) {
    LDETR_CHECK(bbox && valid && losses && grads, "layout_losses: null pointer");
    LDETR_CHECK(B > 0 && N > 0 && N <= 64, "layout_losses: needs 1 <= N <= 64 boxes per sample (got %d)", N);
    LayoutLossParams p; memset(&p, 0, sizeof(p));
    p.box = bbox; p.ref = bbox_ref; p.valid = valid; p.losses = losses; p.grads = grads; p.B = B; p.N = N;
    hipLaunchKernelGGL(layout_losses_kernel, dim3(B), 64, 0, (hipStream_t)stream, p);
    return check_launch("layout_losses");
}

extern "C" int ldetr_layout_losses_bwd_f32(const float* grads, const float* grad_losses, int B, int N, float* dbbox, void* stream) {
    LDETR_CHECK(grads && grad_losses && dbbox, "layout_losses_bwd: null pointer");
    LDETR_CHECK(B > 0 && N > 0 && N <= 64, "layout_losses_bwd: needs 1 <= N <= 64 boxes per sample (got %d)", N);
    LayoutLossParams p; memset(&p, 0, sizeof(p));
    p.grads = const_cast<float*>(grads); p.gout = grad_losses; p.dbox = dbbox; p.B = B; p.N = N;
    const long per = (long)B * N * 4;
    hipLaunchKernelGGL(layout_losses_bwd_kernel, dim3((unsigned)((per + 255) / 256 > 1024 ? 1024 : (per + 255) / 256)), 256, 0, (hipStream_t)stream, p);
    return check_launch("layout_losses_bwd");
}

static int loss_combine_fill(LossCombineParams& p, const float* const* x, const float* w, const int* n, const int* fn, const int* red, int K, int ld) {
    LDETR_CHECK(x && w && n && fn && red && K >= 1 && K <= 16 && ld >= 2, "loss_combine: 1..16 terms");
    memset(&p, 0, sizeof(p));
    for (int k = 0; k < K; k++) {
        LDETR_CHECK(x[k] && n[k] >= 1 && n[k] <= ld && fn[k] >= 0 && fn[k] <= 3 && (fn[k] != 3 || n[k] == 2), "loss_combine: bad term %d", k);
        p.x[k] = x[k]; p.w[k] = w[k]; p.n[k] = n[k]; p.fn[k] = fn[k]; p.red[k] = red[k];
    }
    p.K = K; p.ld = ld;
    return LDETR_OK;
}

extern "C" int ldetr_loss_combine_fwd_f32(const float* const* x, const float* w, const int* n, const int* fn, const int* red, int K, int ld,
                                          float* vals, float* sums, float* total, void* stream) {
    LossCombineParams p;
    if (int rc = loss_combine_fill(p, x, w, n, fn, red, K, ld)) return rc;
    LDETR_CHECK(vals && sums && total, "loss_combine_fwd: null output");
    p.vals = vals; p.sums = sums; p.total = total;
    hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), 256, 0, (hipStream_t)stream, p);
    return check_launch("loss_combine_fwd");
}

extern "C" int ldetr_loss_combine_bwd_f32(const float* const* x, const float* w, const int* n, const int* fn, const int* red, int K, int ld,
                                          const float* g, float* grads, void* stream) {
    LossCombineParams p;
    if (int rc = loss_combine_fill(p, x, w, n, fn, red, K, ld)) return rc;
    LDETR_CHECK(g && grads, "loss_combine_bwd: null pointer");
    p.g = g; p.grads = grads;
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3((unsigned)K), 256, 0, (hipStream_t)stream, p);
    return check_launch("loss_combine_bwd");
}

extern "C" int ldetr_masked_mse_fwd_f32(const float* a, const float* b, const uint8_t* valid, int64_t rows, int D, int bdiv, float* out2, void* stream) {
    LDETR_CHECK(a && b && valid && out2 && rows >= 0 && D >= 1 && bdiv >= 1, "masked_mse_fwd: bad arguments");
    MaskedMseParams p; memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.valid = valid; p.rows = rows; p.D = D; p.bdiv = bdiv; p.out = out2;
    hipLaunchKernelGGL(masked_mse_fwd_kernel, dim3(1), 256, 0, (hipStream_t)stream, p);
    return check_launch("masked_mse_fwd");
}

extern "C" int ldetr_masked_mse_bwd_f32(const float* a, const float* b, const uint8_t* valid, int64_t rows, int D, int bdiv, const float* out2,
                                        const float* g, float* da, void* stream) {
    LDETR_CHECK(a && b && valid && out2 && g && da && rows >= 0 && D >= 1 && bdiv >= 1, "masked_mse_bwd: bad arguments");
    if (rows == 0) return LDETR_OK;
    MaskedMseParams p; memset(&p, 0, sizeof(p));
    p.a = a; p.b = b; p.valid = valid; p.rows = rows; p.D = D; p.bdiv = bdiv; p.out = const_cast<float*>(out2); p.g = g; p.da = da;
    const long n = rows * D;
    hipLaunchKernelGGL(masked_mse_bwd_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 256)), 256, 0, (hipStream_t)stream, p);
    return check_launch("masked_mse_bwd");
}
