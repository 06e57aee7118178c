// Fused residual-add + dropout + LayerNorm, forward and backward.
// Replaces the `x = norm(x + dropout(sublayer(x)))` chains of the post-norm DETR layers
// (training/detr_transformer.py:210-214, 275-285; nn.LayerNorm(256), eps 1e-5) which the reference
// executes as 3-4 separate elementwise/reduction kernels.
// HBM-bound: forward reads x, r and writes z (pre-norm sum, kept for backward) and y: 16 B/element;
// one 64-lane wave per row, float4 per lane, row statistics by wave shuffles only.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

// One problem of a launch = the public argument block (include/ldetr_hip.h: ldetr_ln_args); a launch carries one or two of them (the second
// problem's row blocks follow the first's in the grid: two independent stacks' LayerNorms as ONE launch).
typedef ldetr_ln_args LnParams;

// NV = float4 vectors per lane (D = 256*NV at most; lanes past D/4 idle)
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnParams pa, LnParams pb, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const LnParams& p = second ? pb : pa;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long row = (long)(second ? (int)blockIdx.x - nb0 : (int)blockIdx.x) * 4 + wave;
    if (row >= p.rows) return;
    const int D4 = p.D >> 2;
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    const uint64_t seed = p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull);
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        int c4 = lane + i * 64;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < D4) {
            v[i] = reinterpret_cast<const float4*>(p.x + row * p.D)[c4];
            if (p.r) {
                float4 rr = reinterpret_cast<const float4*>(p.r + row * p.D)[c4];
                if (p.r_parts > 0) {
                    if (p.r_bias) { const float4 bb = reinterpret_cast<const float4*>(p.r_bias)[c4]; rr.x += bb.x; rr.y += bb.y; rr.z += bb.z; rr.w += bb.w; }
                    // sixteen loads in flight per round, the ragged tail as a predicated round (uniform predicate), added in slice order.  A load per
                    // iteration serialised 31 round trips (12 us); rounds of eight with a scalar tail still serialised the 7 leftover slices of the
                    // 8-head and 32-slice residuals: 7-10 round trips per launch instead of 1-2.
                    constexpr int PW = NV == 1 ? 16 : 8;      // (wider rows keep their register budget: their launches are bandwidth-, not latency-bound)
                    for (int s = 1; s < p.r_parts; s += PW) {
                        float4 q[PW];
#pragma unroll
                        for (int j = 0; j < PW; j++)
                            q[j] = (s + j < p.r_parts) ? reinterpret_cast<const float4*>(p.r + (s + j) * p.r_part_stride + row * p.D)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int j = 0; j < PW; j++)
                            if (s + j < p.r_parts) { rr.x += q[j].x; rr.y += q[j].y; rr.z += q[j].z; rr.w += q[j].w; }
                    }
                }
                if (p.p_drop > 0.f) {
                    uint64_t e = (uint64_t)row * p.D + (c4 << 2);
                    rr.x *= drop_scale(seed, e, p.p_drop, inv_keep);
                    rr.y *= drop_scale(seed, e + 1, p.p_drop, inv_keep);
                    rr.z *= drop_scale(seed, e + 2, p.p_drop, inv_keep);
                    rr.w *= drop_scale(seed, e + 3, p.p_drop, inv_keep);
                }
                v[i].x += rr.x; v[i].y += rr.y; v[i].z += rr.z; v[i].w += rr.w;
            }
            if (p.z) reinterpret_cast<float4*>(p.z + row * p.D)[c4] = v[i];
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    const float mean = wave_sum(s) / p.D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        int c4 = lane + i * 64;
        if (c4 < D4) {
            float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / p.D + p.eps);
    if (lane == 0) { if (p.mean) p.mean[row] = mean; if (p.rstd) p.rstd[row] = rstd; }
#pragma unroll
    for (int i = 0; i < NV; i++) {
        int c4 = lane + i * 64;
        if (c4 < D4) {
            float4 g = reinterpret_cast<const float4*>(p.gamma)[c4];
            float4 b = reinterpret_cast<const float4*>(p.beta)[c4];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
            reinterpret_cast<float4*>(p.y + row * p.D)[c4] = o;
            if (p.ypos) {
                const float4 pp = reinterpret_cast<const float4*>(p.pos + (row % p.pos_rows) * p.D)[c4];
                o.x += pp.x; o.y += pp.y; o.z += pp.z; o.w += pp.w;
                reinterpret_cast<float4*>(p.ypos + row * p.D)[c4] = o;
            }
        }
    }
}

// Backward: dz = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma.
// dx = dz; dr = dz * dropmask (regenerated).  dgamma/dbeta: per-wave register partials over a
// grid-stride row loop, LDS combine across the 4 waves, one atomicAdd per block per column.
template <int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnParams pa, LnParams pb, int nb0) {
    __shared__ float red[2][4][256 * NV];
    const bool second = (int)blockIdx.x >= nb0;
    const LnParams& p = second ? pb : pa;
    const int bx = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, gx = second ? (int)gridDim.x - nb0 : nb0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D4 = p.D >> 2;
    const float inv_keep = p.p_drop > 0.f ? 1.f / (1.f - p.p_drop) : 1.f;
    const uint64_t seed = p.seed + (p.seed_ptr ? *p.seed_ptr : 0ull);
    float4 ag[NV], ab[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) { ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; }
    for (long row = (long)bx * 4 + wave; row < p.rows; row += (long)gx * 4) {
        const float mean = p.mean[row], rstd = p.rstd[row];
        float4 xh[NV], g[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            int c4 = lane + i * 64;
            xh[i] = make_float4(0.f, 0.f, 0.f, 0.f); g[i] = xh[i];
            if (c4 < D4) {
                float4 zz = reinterpret_cast<const float4*>(p.z + row * p.D)[c4];
                float4 dy = reinterpret_cast<const float4*>(p.dy + row * p.D)[c4];
                if (p.dy2) {
                    const float4 d2 = reinterpret_cast<const float4*>(p.dy2 + row * p.D)[c4];
                    dy.x += d2.x; dy.y += d2.y; dy.z += d2.z; dy.w += d2.w;
                }
                constexpr int PW = NV == 1 ? 16 : 8;
                for (int s = 0; s < p.dy_nparts; s += PW) {      // PW loads in flight, ragged tail predicated (see ln_fwd_kernel)
                    float4 q[PW];
#pragma unroll
                    for (int j = 0; j < PW; j++)
                        q[j] = (s + j < p.dy_nparts) ? reinterpret_cast<const float4*>(p.dy_parts + (s + j) * p.dy_part_stride + row * p.D)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < PW; j++)
                        if (s + j < p.dy_nparts) { dy.x += q[j].x; dy.y += q[j].y; dy.z += q[j].z; dy.w += q[j].w; }
                }
                float4 gm = reinterpret_cast<const float4*>(p.gamma)[c4];
                xh[i].x = (zz.x - mean) * rstd; xh[i].y = (zz.y - mean) * rstd;
                xh[i].z = (zz.z - mean) * rstd; xh[i].w = (zz.w - mean) * rstd;
                g[i].x = dy.x * gm.x; g[i].y = dy.y * gm.y; g[i].z = dy.z * gm.z; g[i].w = dy.w * gm.w;
                s1 += g[i].x + g[i].y + g[i].z + g[i].w;
                s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
                ag[i].x += dy.x * xh[i].x; ag[i].y += dy.y * xh[i].y; ag[i].z += dy.z * xh[i].z; ag[i].w += dy.w * xh[i].w;
                ab[i].x += dy.x; ab[i].y += dy.y; ab[i].z += dy.z; ab[i].w += dy.w;
            }
        }
        const float c1 = wave_sum(s1) / p.D, c2 = wave_sum(s2) / p.D;
#pragma unroll
        for (int i = 0; i < NV; i++) {
            int c4 = lane + i * 64;
            if (c4 < D4) {
                float4 dz;
                dz.x = rstd * (g[i].x - c1 - xh[i].x * c2); dz.y = rstd * (g[i].y - c1 - xh[i].y * c2);
                dz.z = rstd * (g[i].z - c1 - xh[i].z * c2); dz.w = rstd * (g[i].w - c1 - xh[i].w * c2);
                if (p.dx) reinterpret_cast<float4*>(p.dx + row * p.D)[c4] = dz;
                if (p.dr) {
                    if (p.p_drop > 0.f) {
                        uint64_t e = (uint64_t)row * p.D + (c4 << 2);
                        dz.x *= drop_scale(seed, e, p.p_drop, inv_keep);
                        dz.y *= drop_scale(seed, e + 1, p.p_drop, inv_keep);
                        dz.z *= drop_scale(seed, e + 2, p.p_drop, inv_keep);
                        dz.w *= drop_scale(seed, e + 3, p.p_drop, inv_keep);
                    }
                    reinterpret_cast<float4*>(p.dr + row * p.D)[c4] = dz;
                }
            }
        }
    }
    if (!p.dgamma) return;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        int c = (lane + i * 64) << 2;
        red[0][wave][c + 0] = ag[i].x; red[0][wave][c + 1] = ag[i].y; red[0][wave][c + 2] = ag[i].z; red[0][wave][c + 3] = ag[i].w;
        red[1][wave][c + 0] = ab[i].x; red[1][wave][c + 1] = ab[i].y; red[1][wave][c + 2] = ab[i].z; red[1][wave][c + 3] = ab[i].w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.D; c += 256) {
        float g = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        float b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        atomicAdd(p.dgamma + c, g);
        atomicAdd(p.dbeta + c, b);
    }
}

}  // namespace ldetr

using namespace ldetr;

extern "C" int ldetr_layernorm_fwd_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                       float* y, float* z, float* mean, float* rstd, int64_t rows, int D, float eps,
                                       float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream) {
    return ldetr_layernorm_fwd_pos_f32(x, residual, gamma, beta, y, z, mean, rstd, rows, D, eps, p_drop, seed, seed_ptr, nullptr, 0, nullptr, stream);
}

extern "C" int ldetr_layernorm_fwd_pos_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                           float* y, float* z, float* mean, float* rstd, int64_t rows, int D, float eps,
                                           float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                           const float* pos, int64_t pos_rows, float* ypos, void* stream) {
    return ldetr_layernorm_fwd_parts_f32(x, residual, 0, 0, nullptr, gamma, beta, y, z, mean, rstd, rows, D, eps, p_drop, seed, seed_ptr, pos, pos_rows, ypos, stream);
}

static int ln_check_fwd(const LnParams& p) {
    LDETR_CHECK(p.r_parts >= 0 && (p.r_parts == 0 || (p.r && p.r_part_stride >= p.rows * p.D)), "layernorm_fwd: bad partial-sum arguments");
    LDETR_CHECK(p.x && p.gamma && p.beta && p.y, "layernorm_fwd: null pointer");
    LDETR_CHECK((p.pos == nullptr) == (p.ypos == nullptr) && (!p.pos || p.pos_rows > 0), "layernorm_fwd: pos, pos_rows and ypos go together");
    LDETR_CHECK(p.D % 4 == 0 && p.D >= 4 && p.D <= 1024, "layernorm_fwd: D must be a multiple of 4 in [4, 1024]");
    LDETR_CHECK(p.rows >= 0 && p.rows < (1L << 31) - 8, "layernorm_fwd: bad row count");
    return LDETR_OK;
}

// n = 1 or 2 problems (same D) as ONE launch
extern "C" int ldetr_layernorm_fwd_group_f32(const ldetr_ln_args* a, int n, void* stream) {
    LDETR_CHECK(a && (n == 1 || n == 2), "layernorm_fwd_group: 1 or 2 problems");
    LnParams p[2]; p[0] = a[0]; p[1] = n == 2 ? a[1] : a[0];
    for (int i = 0; i < n; i++) {
        if (int rc = ln_check_fwd(p[i])) return rc;
        if (p[i].r_parts == 0) p[i].r_bias = nullptr;
    }
    LDETR_CHECK(n == 1 || p[0].D == p[1].D, "layernorm_fwd_group: both problems must have the same D");
    const int nb0 = (int)((p[0].rows + 3) / 4), nb1 = n == 2 ? (int)((p[1].rows + 3) / 4) : 0;
    if (nb0 + nb1 == 0) return LDETR_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nv = (p[0].D + 255) / 256, grid = nb0 + nb1;
    if (nv == 1) hipLaunchKernelGGL(ln_fwd_kernel<1>, grid, 256, 0, st, p[0], p[1], nb0);
    else if (nv == 2) hipLaunchKernelGGL(ln_fwd_kernel<2>, grid, 256, 0, st, p[0], p[1], nb0);
    else if (nv == 3) hipLaunchKernelGGL(ln_fwd_kernel<3>, grid, 256, 0, st, p[0], p[1], nb0);
    else hipLaunchKernelGGL(ln_fwd_kernel<4>, grid, 256, 0, st, p[0], p[1], nb0);
    return check_launch("layernorm_fwd");
}

extern "C" int ldetr_layernorm_fwd_parts_f32(const float* x, const float* parts, int n_parts, int64_t part_stride, const float* part_bias,
                                             const float* gamma, const float* beta, float* y, float* z, float* mean, float* rstd,
                                             int64_t rows, int D, float eps, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                             const float* pos, int64_t pos_rows, float* ypos, void* stream) {
    LnParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.r = parts; p.gamma = gamma; p.beta = beta; p.y = y; p.z = z; p.mean = mean; p.rstd = rstd;
    p.rows = rows; p.D = D; p.eps = eps; p.p_drop = p_drop; p.seed = seed; p.seed_ptr = seed_ptr;
    p.pos = pos; p.pos_rows = pos_rows; p.ypos = ypos;
    p.r_parts = n_parts; p.r_part_stride = part_stride; p.r_bias = n_parts > 0 ? part_bias : nullptr;
    return ldetr_layernorm_fwd_group_f32(&p, 1, stream);
}

// dgamma/dbeta are accumulated with atomics: the caller zeroes them (or passes running gradients).
extern "C" int ldetr_layernorm_bwd_f32(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                       float* dx, float* dresidual, float* dgamma, float* dbeta, int64_t rows, int D,
                                       float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream) {
    return ldetr_layernorm_bwd2_f32(dy, nullptr, z, mean, rstd, gamma, dx, dresidual, dgamma, dbeta, rows, D, p_drop, seed, seed_ptr, stream);
}

extern "C" int ldetr_layernorm_bwd2_f32(const float* dy, const float* dy2, const float* z, const float* mean, const float* rstd, const float* gamma,
                                        float* dx, float* dresidual, float* dgamma, float* dbeta, int64_t rows, int D,
                                        float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream) {
    return ldetr_layernorm_bwd_parts_f32(dy, dy2, nullptr, 0, 0, z, mean, rstd, gamma, dx, dresidual, dgamma, dbeta, rows, D, p_drop, seed, seed_ptr, stream);
}

static int ln_check_bwd(const LnParams& p) {
    LDETR_CHECK(p.dy_nparts >= 0 && (p.dy_nparts == 0 || (p.dy_parts && p.dy_part_stride >= p.rows * p.D)), "layernorm_bwd: bad partial-sum arguments");
    LDETR_CHECK(p.dy && p.z && p.mean && p.rstd && p.gamma, "layernorm_bwd: null pointer");
    LDETR_CHECK(p.D % 4 == 0 && p.D >= 4 && p.D <= 1024, "layernorm_bwd: D must be a multiple of 4 in [4, 1024]");
    LDETR_CHECK((p.dgamma == nullptr) == (p.dbeta == nullptr), "layernorm_bwd: dgamma and dbeta go together");
    return LDETR_OK;
}

extern "C" int ldetr_layernorm_bwd_group_f32(const ldetr_ln_args* a, int n, void* stream) {
    LDETR_CHECK(a && (n == 1 || n == 2), "layernorm_bwd_group: 1 or 2 problems");
    LnParams p[2]; p[0] = a[0]; p[1] = n == 2 ? a[1] : a[0];
    for (int i = 0; i < n; i++)
        if (int rc = ln_check_bwd(p[i])) return rc;
    LDETR_CHECK(n == 1 || p[0].D == p[1].D, "layernorm_bwd_group: both problems must have the same D");
    auto blocks = [](long rows) { long g = (rows + 3) / 4; return (int)(g > 512 ? 512 : g); };
    const int nb0 = blocks(p[0].rows), nb1 = n == 2 ? blocks(p[1].rows) : 0;
    if (nb0 + nb1 == 0) return LDETR_OK;
    hipStream_t st = (hipStream_t)stream;
    const int nv = (p[0].D + 255) / 256, grid = nb0 + nb1;
    if (nv == 1) hipLaunchKernelGGL(ln_bwd_kernel<1>, grid, 256, 0, st, p[0], p[1], nb0);
    else if (nv == 2) hipLaunchKernelGGL(ln_bwd_kernel<2>, grid, 256, 0, st, p[0], p[1], nb0);
    else if (nv == 3) hipLaunchKernelGGL(ln_bwd_kernel<3>, grid, 256, 0, st, p[0], p[1], nb0);
    else hipLaunchKernelGGL(ln_bwd_kernel<4>, grid, 256, 0, st, p[0], p[1], nb0);
    return check_launch("layernorm_bwd");
}

extern "C" int ldetr_layernorm_bwd_parts_f32(const float* dy, const float* dy2, const float* dy_parts, int n_parts, int64_t part_stride,
                                             const float* z, const float* mean, const float* rstd, const float* gamma,
                                             float* dx, float* dresidual, float* dgamma, float* dbeta, int64_t rows, int D,
                                             float p_drop, uint64_t seed, const uint64_t* seed_ptr, void* stream) {
    LnParams p; memset(&p, 0, sizeof(p));
    p.dy = dy; p.z = const_cast<float*>(z); p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd);
    p.gamma = gamma; p.dx = dx; p.dr = dresidual; p.dgamma = dgamma; p.dbeta = dbeta; p.dy2 = dy2;
    p.dy_parts = dy_parts; p.dy_nparts = n_parts; p.dy_part_stride = part_stride;
    p.rows = rows; p.D = D; p.p_drop = p_drop; p.seed = seed; p.seed_ptr = seed_ptr;
    return ldetr_layernorm_bwd_group_f32(&p, 1, stream);
}

// out[rows][D] = base[rows][D] (or 0) + sum_s parts[s][rows][D], in slice order: where a gradient that travelled between sub-blocks as partial sums
// (the per-head input gradients of ldetr_mha_small_bwd_group_f32) leaves the token stacks as ONE tensor.
__global__ __launch_bounds__(256) void sum_parts_kernel(const float4* __restrict__ base, const float4* __restrict__ parts, int n_parts, long part_stride4, float4* __restrict__ out, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 v = base ? base[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < n_parts; s += 8) {
            float4 q[8];
#pragma unroll
            for (int j = 0; j < 8; j++) q[j] = (s + j < n_parts) ? parts[(long)(s + j) * part_stride4 + i] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (s + j < n_parts) { v.x += q[j].x; v.y += q[j].y; v.z += q[j].z; v.w += q[j].w; }
        }
        out[i] = v;
    }
}

extern "C" int ldetr_sum_parts_f32(const float* base, const float* parts, int n_parts, int64_t part_stride, float* out, int64_t n, void* stream) {
    LDETR_CHECK(out && n >= 0 && n % 4 == 0 && part_stride % 4 == 0 && n_parts >= 0 && (n_parts == 0 || parts), "sum_parts: bad arguments");
    LDETR_CHECK(((((uintptr_t)base) | ((uintptr_t)parts) | ((uintptr_t)out)) & 15) == 0, "sum_parts: buffers must be 16-byte aligned");
    if (n == 0) return LDETR_OK;
    const long n4 = n / 4;
    const int grid = (int)std::min<long>((n4 + 255) / 256, 1024);
    hipLaunchKernelGGL(sum_parts_kernel, grid, 256, 0, (hipStream_t)stream, (const float4*)base, (const float4*)parts, n_parts, (long)(part_stride / 4), (float4*)out, n4);
    return check_launch("sum_parts");
}
