// Data-parallel step kernels over *flat* parameter / gradient buffers.
// Replaces the per-phase sequence in training/training_loop.py:303-313 (torch.cat of every grad,
// all_reduce, /num_gpus, nan_to_num, split + per-tensor Adam) and the G_ema lerp at :320-328.
// With all parameters of a module laid out in one contiguous fp32 buffer the gradient exchange
// needs no pack/unpack copies and the optimiser is a single streaming launch:
//   Adam: reads p, g, m, v and writes p, m, v = 28 B/parameter;  EMA: 12 B/parameter.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

__device__ __forceinline__ float sanitize(float g, float scale, float nanv, float posinf, float neginf) {
    g *= scale;
    if (g != g) return nanv;
    if (g == INFINITY) return posinf;
    if (g == -INFINITY) return neginf;
    return g;
}

__global__ __launch_bounds__(256) void grad_sanitize_kernel(float* g, long n, float scale, float nanv, float posinf, float neginf) {
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = reinterpret_cast<float4*>(g)[i];
        v.x = sanitize(v.x, scale, nanv, posinf, neginf); v.y = sanitize(v.y, scale, nanv, posinf, neginf);
        v.z = sanitize(v.z, scale, nanv, posinf, neginf); v.w = sanitize(v.w, scale, nanv, posinf, neginf);
        reinterpret_cast<float4*>(g)[i] = v;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        g[i] = sanitize(g[i], scale, nanv, posinf, neginf);
}

struct AdamParams {
    float* p; const float* g; float* m; float* v; long n;
    float lr, b1, b2, eps, bc1, bc2_sqrt;
    int fuse_sanitize; float gscale, nanv, posinf, neginf;
    float* pe; float ema_beta;      // optional: p_ema = lerp(p_new, p_ema, beta) in the same pass (G_ema, training_loop.py:320-328)
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamParams& a) {
    if (a.fuse_sanitize) g = sanitize(g, a.gscale, a.nanv, a.posinf, a.neginf);
    m = a.b1 * m + (1.f - a.b1) * g;
    v = a.b2 * v + (1.f - a.b2) * g * g;
    float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p -= (a.lr / a.bc1) * (m / denom);
}

// Block -> address mapping (round 6, tools/probes/stream_probe.hip at D's 90 M parameters = 1.44 GB of p / g / m / v, far beyond the 256 MB Infinity Cache): a block owns
// 16 KB-contiguous chunks of every array (4 rounds of its 256 lanes x 16 B) and strides over chunks -- 5.54 TB/s against 5.11 for the plain grid-stride loop, 4.65 for
// XCD-contiguous eighths (each XCD streaming its own region: worse, not better), 4.9 .. 5.4 for 64 KB .. 1 MB chunks; a plain copy on the same box reaches 5.64 TB/s.
constexpr long STREAM_CHUNK4 = 1024;     // float4 per chunk

template <bool EMA>      // (as a run-time branch on a.pe the plain step lost 7 % of its rate: 5.19 vs 5.58 TB/s)
__global__ __launch_bounds__(256) void adam_kernel(AdamParams a) {
    const long n4 = a.n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long c = (long)blockIdx.x * STREAM_CHUNK4; c < n4; c += (long)gridDim.x * STREAM_CHUNK4) {
      const long ce = c + STREAM_CHUNK4 < n4 ? c + STREAM_CHUNK4 : n4;
      for (long i = c + threadIdx.x; i < ce; i += 256) {
        // g, m, v are touched once per step (2.5 GB for D: nothing of it survives in the 256 MB Infinity Cache anyway): non-temporal, so the
        // stream does not evict the parameters, which the next phase's first kernels read
        f32x4 pv = reinterpret_cast<f32x4*>(a.p)[i];
        f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.g) + i);
        f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.m) + i);
        f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.v) + i);
        float4 p = make_float4(pv[0], pv[1], pv[2], pv[3]), g = make_float4(gv[0], gv[1], gv[2], gv[3]);
        float4 m = make_float4(mv[0], mv[1], mv[2], mv[3]), v = make_float4(vv[0], vv[1], vv[2], vv[3]);
        adam_elem(p.x, g.x, m.x, v.x, a); adam_elem(p.y, g.y, m.y, v.y, a);
        adam_elem(p.z, g.z, m.z, v.z, a); adam_elem(p.w, g.w, m.w, v.w, a);
        reinterpret_cast<float4*>(a.p)[i] = p;
        __builtin_nontemporal_store(f32x4{m.x, m.y, m.z, m.w}, reinterpret_cast<f32x4*>(a.m) + i);
        __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(a.v) + i);
        if constexpr (EMA) {   // the G_ema lerp rides along: the updated parameters are in registers, p is not read a second time
            const f32x4 ev = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(a.pe) + i);
            __builtin_nontemporal_store(f32x4{p.x + a.ema_beta * (ev[0] - p.x), p.y + a.ema_beta * (ev[1] - p.y),
                                              p.z + a.ema_beta * (ev[2] - p.z), p.w + a.ema_beta * (ev[3] - p.w)}, reinterpret_cast<f32x4*>(a.pe) + i);
        }
      }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += stride) {
        adam_elem(a.p[i], a.g[i], a.m[i], a.v[i], a);
        if constexpr (EMA) a.pe[i] = a.p[i] + a.ema_beta * (a.pe[i] - a.p[i]);
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* pe, const float* p, long n, float beta) {
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long c = (long)blockIdx.x * STREAM_CHUNK4; c < n4; c += (long)gridDim.x * STREAM_CHUNK4) {     // block-owned 16 KB chunks, see adam_kernel
      const long ce = c + STREAM_CHUNK4 < n4 ? c + STREAM_CHUNK4 : n4;
      for (long i = c + threadIdx.x; i < ce; i += 256) {
        float4 a = reinterpret_cast<const float4*>(p)[i];
        const f32x4 ev = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(pe) + i);     // p_ema is touched once per iteration: streaming
        float4 e = make_float4(ev[0], ev[1], ev[2], ev[3]);
        e.x = a.x + beta * (e.x - a.x); e.y = a.y + beta * (e.y - a.y);
        e.z = a.z + beta * (e.z - a.z); e.w = a.w + beta * (e.w - a.w);
        __builtin_nontemporal_store(f32x4{e.x, e.y, e.z, e.w}, reinterpret_cast<f32x4*>(pe) + i);
      }
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        pe[i] = p[i] + beta * (pe[i] - p[i]);
}

static inline int stream_grid(long n) {
    long g = (n / 4 + STREAM_CHUNK4 - 1) / STREAM_CHUNK4;      // one block per 16 KB chunk, at most 16 blocks per CU (adam / ema; the sanitise loop strides by the grid either way)
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace ldetr

using namespace ldetr;

// g = nan_to_num(g * scale, nan, posinf, neginf) in place  (training_loop.py:306-309).
extern "C" int ldetr_grad_sanitize_f32(float* g, int64_t n, float scale, float nan_value, float posinf, float neginf, void* stream) {
    LDETR_CHECK(g || n == 0, "grad_sanitize: null pointer");
    LDETR_CHECK(((uintptr_t)g & 15) == 0, "grad_sanitize: buffer must be 16-byte aligned");
    if (n == 0) return LDETR_OK;
    hipLaunchKernelGGL(grad_sanitize_kernel, stream_grid(n), 256, 0, (hipStream_t)stream, g, (long)n, scale, nan_value, posinf, neginf);
    return check_launch("grad_sanitize");
}

// torch.optim.Adam step (no weight decay, no amsgrad) over a flat buffer; `step` is the 1-based step count.
extern "C" int ldetr_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, int64_t step,
                                   float lr, float beta1, float beta2, float eps,
                                   int fuse_sanitize, float gscale, float nan_value, float posinf, float neginf, void* stream) {
    return ldetr_adam_ema_step_f32(p, g, m, v, n, step, lr, beta1, beta2, eps, fuse_sanitize, gscale, nan_value, posinf, neginf, nullptr, 0.f, stream);
}

// ... with the exponential moving average of the parameters updated in the same pass: p_ema = p_new + ema_beta * (p_ema - p_new)
// (training_loop.py:320-328 runs it after the last phase of the iteration; G's parameters do not change between G's optimiser step and
// that point, so the value is the same).  p_ema == NULL: plain Adam.
extern "C" int ldetr_adam_ema_step_f32(float* p, const float* g, float* m, float* v, int64_t n, int64_t step,
                                       float lr, float beta1, float beta2, float eps,
                                       int fuse_sanitize, float gscale, float nan_value, float posinf, float neginf,
                                       float* p_ema, float ema_beta, void* stream) {
    LDETR_CHECK(!p_ema || (((uintptr_t)p_ema) & 15) == 0, "adam_step: p_ema must be 16-byte aligned");
    LDETR_CHECK((p && g && m && v) || n == 0, "adam_step: null pointer");
    LDETR_CHECK((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
    LDETR_CHECK(step >= 1, "adam_step: step must be >= 1");
    if (n == 0) return LDETR_OK;
    AdamParams a; memset(&a, 0, sizeof(a));
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps;
    double bc1 = 1.0 - pow((double)beta1, (double)step);
    double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.bc1 = (float)bc1; a.bc2_sqrt = (float)sqrt(bc2);
    a.fuse_sanitize = fuse_sanitize; a.gscale = gscale; a.nanv = nan_value; a.posinf = posinf; a.neginf = neginf;
    a.pe = p_ema; a.ema_beta = ema_beta;
    if (p_ema) hipLaunchKernelGGL(adam_kernel<true>, stream_grid(n), 256, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(adam_kernel<false>, stream_grid(n), 256, 0, (hipStream_t)stream, a);
    return check_launch("adam_step");
}

// p_ema = p + beta * (p_ema - p)   == p.lerp(p_ema, beta)  (training_loop.py:325-326)
extern "C" int ldetr_ema_lerp_f32(float* p_ema, const float* p, int64_t n, float beta, void* stream) {
    LDETR_CHECK((p_ema && p) || n == 0, "ema_lerp: null pointer");
    LDETR_CHECK((((uintptr_t)p | (uintptr_t)p_ema) & 15) == 0, "ema_lerp: buffers must be 16-byte aligned");
    if (n == 0) return LDETR_OK;
    hipLaunchKernelGGL(ema_kernel, stream_grid(n), 256, 0, (hipStream_t)stream, p_ema, p, (long)n, beta);
    return check_launch("ema_lerp");
}
