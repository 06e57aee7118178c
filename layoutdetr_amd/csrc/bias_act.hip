// bias_act: fused bias + activation + gain (+clamp), forward / 1st / 2nd derivative.
// Replaces the reference plugin entry `bias_act_plugin.bias_act`
// (torch_utils/ops/bias_act.cpp:33-92, kernel torch_utils/ops/bias_act.cu:25-148).
// HBM-bound streaming op: 8 B/element forward (read x, write y), 12 B/element backward
// (read dy, y; write dx).  float4 per lane, grid-stride, 256-thread blocks.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct BiasActParams {
    const float* x;
    const float* b;
    const float* xref;
    const float* yref;
    const float* dy;
    float* y;
    long sizeX;
    int sizeB;
    long stepB;
    float alpha, gain, clamp;
};

template <int A, int G>
__device__ __forceinline__ float bias_act_elem(float x, float b, float xref, float yref, float dy,
                                               float alpha, float gain, float clamp) {
    const float one = 1.f, two = 2.f, expRange = 80.f, halfExpRange = 40.f;
    const float seluScale = 1.0507009873554804934193349852946f;
    const float seluAlpha = 1.6732632423543772848170429916717f;
    float yy = (gain != 0.f) ? yref / gain : 0.f;
    float y = 0.f;
    if (G == 0) x += b; else xref += b;
    if (A == 1) { y = x; }
    if (A == 2) { if (G == 0) y = (x > 0.f) ? x : 0.f; if (G == 1) y = (yy > 0.f) ? x : 0.f; }
    if (A == 3) { if (G == 0) y = (x > 0.f) ? x : x * alpha; if (G == 1) y = (yy > 0.f) ? x : x * alpha; }
    if (A == 4) {
        if (G == 0) { float c = expf(x); float d = one / c; y = (x < -expRange) ? -one : (x > expRange) ? one : (c - d) / (c + d); }
        if (G == 1) y = x * (one - yy * yy);
        if (G == 2) y = x * (one - yy * yy) * (-two * yy);
    }
    if (A == 5) {
        if (G == 0) y = (x < -expRange) ? 0.f : one / (expf(-x) + one);
        if (G == 1) y = x * yy * (one - yy);
        if (G == 2) y = x * yy * (one - yy) * (one - two * yy);
    }
    if (A == 6) {
        if (G == 0) y = (x >= 0.f) ? x : expf(x) - one;
        if (G == 1) y = (yy >= 0.f) ? x : x * (yy + one);
        if (G == 2) y = (yy >= 0.f) ? 0.f : x * (yy + one);
    }
    if (A == 7) {
        if (G == 0) y = (x >= 0.f) ? seluScale * x : (seluScale * seluAlpha) * (expf(x) - one);
        if (G == 1) y = (yy >= 0.f) ? x * seluScale : x * (yy + seluScale * seluAlpha);
        if (G == 2) y = (yy >= 0.f) ? 0.f : x * (yy + seluScale * seluAlpha);
    }
    if (A == 8) {
        if (G == 0) y = (x > expRange) ? x : logf(expf(x) + one);
        if (G == 1) y = x * (one - expf(-yy));
        if (G == 2) { float c = expf(-yy); y = x * c * (one - c); }
    }
    if (A == 9) {
        if (G == 0) {
            y = (x < -expRange) ? 0.f : x / (expf(-x) + one);
        } else {
            float c = expf(xref);
            float d = c + one;
            if (G == 1) y = (xref > halfExpRange) ? x : x * c * (xref + d) / (d * d);
            else y = (xref > halfExpRange) ? 0.f : x * c * (xref * (two - d) + two * d) / (d * d * d);
            yref = (xref < -expRange) ? 0.f : xref / (expf(-xref) + one) * gain;
        }
    }
    if (A == 10) {   // erf-GELU (BERT intermediate / LM-head transform, training/med.py:296-307, 504-518); grad 1: x = dy, xref = pre-activation
        if (G == 0) y = 0.5f * x * (one + erff(x * 0.70710678118654752f));
        if (G == 1) y = x * (0.5f * (one + erff(xref * 0.70710678118654752f)) + xref * 0.3989422804014327f * expf(-0.5f * xref * xref));
    }
    y *= gain * dy;
    if (clamp >= 0.f) {
        if (G == 0) y = (y > -clamp && y < clamp) ? y : (y >= 0.f) ? clamp : -clamp;
        else y = (yref > -clamp && yref < clamp) ? y : 0.f;
    }
    return y;
}

// BM: 0 no bias, 1 one bias value per float4 (stepB % 4 == 0), 2 four consecutive bias
// values (stepB == 1, sizeB % 4 == 0), 3 scalar generic.
template <int A, int G, int BM>
__global__ __launch_bounds__(256) void bias_act_vec4_kernel(BiasActParams p) {
    const long n4 = p.sizeX >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    // operands beyond the Infinity Cache (>= 128 MB per tensor: the 256 x 256 StyleGAN2 layers) are streamed: every element is touched
    // once, non-temporal requests keep them from evicting what the neighbouring kernels still need (5.2 -> TB/s class of the Adam pass)
    const bool nt = G > 0 && p.sizeX >= (1L << 25);      // (the forward's output is read back by the next layer: streaming it measured slower, 6.6 -> 6.0 TB/s)
    auto ld4 = [&](const float* q, long i) {
        if (nt) { const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q) + i); return make_float4(v[0], v[1], v[2], v[3]); }
        return reinterpret_cast<const float4*>(q)[i];
    };
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 x = ld4(p.x, i);
        float4 xr = make_float4(0, 0, 0, 0), yr = xr, dy = make_float4(1, 1, 1, 1), b = xr;
        if (p.xref) xr = ld4(p.xref, i);
        if (p.yref) yr = ld4(p.yref, i);
        if (p.dy) dy = ld4(p.dy, i);
        if (BM == 1) { float v = p.b[((i << 2) / p.stepB) % p.sizeB]; b = make_float4(v, v, v, v); }
        if (BM == 2) { b = *reinterpret_cast<const float4*>(p.b + ((i << 2) % p.sizeB)); }
        float4 y;
        y.x = bias_act_elem<A, G>(x.x, b.x, xr.x, yr.x, dy.x, p.alpha, p.gain, p.clamp);
        y.y = bias_act_elem<A, G>(x.y, b.y, xr.y, yr.y, dy.y, p.alpha, p.gain, p.clamp);
        y.z = bias_act_elem<A, G>(x.z, b.z, xr.z, yr.z, dy.z, p.alpha, p.gain, p.clamp);
        y.w = bias_act_elem<A, G>(x.w, b.w, xr.w, yr.w, dy.w, p.alpha, p.gain, p.clamp);
        if (nt) __builtin_nontemporal_store(f32x4{y.x, y.y, y.z, y.w}, reinterpret_cast<f32x4*>(p.y) + i);
        else reinterpret_cast<float4*>(p.y)[i] = y;
    }
}

template <int A, int G>
__global__ __launch_bounds__(256) void bias_act_scalar_kernel(BiasActParams p) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < p.sizeX; i += stride) {
        float b = p.b ? p.b[(i / p.stepB) % p.sizeB] : 0.f;
        float xr = p.xref ? p.xref[i] : 0.f;
        float yr = p.yref ? p.yref[i] : 0.f;
        float dy = p.dy ? p.dy[i] : 1.f;
        p.y[i] = bias_act_elem<A, G>(p.x[i], b, xr, yr, dy, p.alpha, p.gain, p.clamp);
    }
}

template <int A, int G>
static int launch_bias_act(const BiasActParams& p, hipStream_t st) {
    bool aligned = ((((uintptr_t)p.x | (uintptr_t)p.y | (uintptr_t)p.xref | (uintptr_t)p.yref | (uintptr_t)p.dy) & 15) == 0) &&
                   (p.sizeX % 4 == 0);
    int bm = -1;
    if (aligned) {
        if (!p.b) bm = 0;
        else if (p.stepB % 4 == 0) bm = 1;
        else if (p.stepB == 1 && p.sizeB % 4 == 0 && (((uintptr_t)p.b) & 15) == 0) bm = 2;
    }
    if (bm >= 0) {
        long n4 = p.sizeX >> 2;
        int grid = (int)((n4 + 255) / 256);
        if (grid > 256 * 16) grid = 256 * 16;
        if (grid < 1) grid = 1;
        if (bm == 0) hipLaunchKernelGGL((bias_act_vec4_kernel<A, G, 0>), grid, 256, 0, st, p);
        if (bm == 1) hipLaunchKernelGGL((bias_act_vec4_kernel<A, G, 1>), grid, 256, 0, st, p);
        if (bm == 2) hipLaunchKernelGGL((bias_act_vec4_kernel<A, G, 2>), grid, 256, 0, st, p);
    } else {
        int grid = (int)((p.sizeX + 255) / 256);
        if (grid > 256 * 16) grid = 256 * 16;
        if (grid < 1) grid = 1;
        hipLaunchKernelGGL((bias_act_scalar_kernel<A, G>), grid, 256, 0, st, p);
    }
    return check_launch("bias_act");
}

template <int A>
static int dispatch_grad(const BiasActParams& p, int grad, hipStream_t st) {
    if (grad == 0) return launch_bias_act<A, 0>(p, st);
    if (grad == 1) return launch_bias_act<A, 1>(p, st);
    return launch_bias_act<A, 2>(p, st);
}

}  // namespace ldetr

extern "C" int ldetr_bias_act_f32(const float* x, const float* b, const float* xref, const float* yref,
                                  const float* dy, float* y, int64_t sizeX, int sizeB, int64_t stepB,
                                  int grad, int act, float alpha, float gain, float clamp, void* stream) {
    using namespace ldetr;
    if (sizeX == 0) return LDETR_OK;
    LDETR_CHECK(x && y, "bias_act: x and y must be non-null");
    LDETR_CHECK(sizeX >= 0 && sizeX <= 2147483647LL, "bias_act: x is too large");
    LDETR_CHECK(grad >= 0 && grad <= 2, "bias_act: grad must be 0, 1 or 2");
    LDETR_CHECK(act >= 1 && act <= 10, "bias_act: no kernel found for the specified activation func");
    LDETR_CHECK(act != 10 || grad <= 1, "bias_act: gelu (10) has no second-order gradient");
    LDETR_CHECK(!b || (sizeB > 0 && stepB > 0), "bias_act: b has wrong number of elements");
    if (sizeX == 0) return LDETR_OK;
    BiasActParams p{x, b, xref, yref, dy, y, (long)sizeX, b ? sizeB : 1, b ? (long)stepB : 1, alpha, gain, clamp};
    hipStream_t st = (hipStream_t)stream;
    switch (act) {
        case 1: return dispatch_grad<1>(p, grad, st);
        case 2: return dispatch_grad<2>(p, grad, st);
        case 3: return dispatch_grad<3>(p, grad, st);
        case 4: return dispatch_grad<4>(p, grad, st);
        case 5: return dispatch_grad<5>(p, grad, st);
        case 6: return dispatch_grad<6>(p, grad, st);
        case 7: return dispatch_grad<7>(p, grad, st);
        case 8: return dispatch_grad<8>(p, grad, st);
        case 9: return dispatch_grad<9>(p, grad, st);
        default: return grad == 0 ? launch_bias_act<10, 0>(p, st) : launch_bias_act<10, 1>(p, st);
    }
}
