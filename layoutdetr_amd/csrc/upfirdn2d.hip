// upfirdn2d: zero-insert upsample -> pad/crop -> 2-D FIR -> decimate, per (n, c) plane.
// Replaces the reference plugin entry `upfirdn2d_plugin.upfirdn2d`
// (torch_utils/ops/upfirdn2d.cpp:17-99, kernels torch_utils/ops/upfirdn2d.cu:30-201).
// Stride-aware in x and y (contiguous NCHW and channels_last both run without a copy).
// HBM-bound: every input element is read once from HBM (neighbour re-reads hit L1/LDS) and
// every output element written once.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct UpfirdnParams {
    const float* x;
    const float* f;
    float* y;
    // optional fused epilogue (used by the StyleGAN2 up-layers): y = lrelu(v + bias[c]) * gain2
    const float* bias;
    float act_alpha, act_gain;
    int has_act;
    int upx, upy, downx, downy, padx0, pady0, flip;
    float gain;
    int inW, inH, C, N;
    long xs_w, xs_h, xs_c, xs_n;
    int fw, fh;
    long fs_w, fs_h;
    int outW, outH;
    long ys_w, ys_h, ys_c, ys_n;
};

// floor(a / b) for b > 0 (C division truncates toward zero).
__device__ __forceinline__ int floor_div(int a, int b) {
    int q = a / b;
    return q - ((a - q * b) < 0 ? 1 : 0);
}

#define LDETR_MAX_TAPS 1024

// Filter staged to LDS with flip resolved: fl[ky*fw + kx] is the tap applied to the input
// sample that sits ky rows / kx columns *in filter index order of the reference inner loop*.
__device__ __forceinline__ void stage_filter(const UpfirdnParams& p, float* fl) {
    for (int i = threadIdx.x; i < p.fw * p.fh; i += blockDim.x) {
        int ky = i / p.fw, kx = i - ky * p.fw;
        fl[i] = p.f[kx * p.fs_w + ky * p.fs_h];
    }
    __syncthreads();
}

__device__ __forceinline__ float epilogue(const UpfirdnParams& p, float v, int c) {
    v *= p.gain;
    if (p.has_act) {
        if (p.bias) v += p.bias[c];
        v = (v > 0.f ? v : v * p.act_alpha) * p.act_gain;
    }
    return v;
}

// One output element per thread; threads run along outX (contiguous for NCHW outputs).
__global__ __launch_bounds__(256) void upfirdn2d_planar_kernel(UpfirdnParams p) {
    __shared__ float fl[LDETR_MAX_TAPS];
    stage_filter(p, fl);
    const long total = (long)p.N * p.C * p.outH * p.outW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int outX = (int)(idx % p.outW);
        long t = idx / p.outW;
        int outY = (int)(t % p.outH);
        t /= p.outH;
        int c = (int)(t % p.C);
        int n = (int)(t / p.C);

        int midY = outY * p.downy + p.upy - 1 - p.pady0;
        int inY = min(max(floor_div(midY, p.upy), 0), p.inH);
        int h = min(max(floor_div(midY + p.fh, p.upy), 0), p.inH) - inY;
        int filterY = midY + p.fh - (inY + 1) * p.upy;
        if (p.flip) filterY = p.fh - 1 - filterY;
        int midX = outX * p.downx + p.upx - 1 - p.padx0;
        int inX = min(max(floor_div(midX, p.upx), 0), p.inW);
        int w = min(max(floor_div(midX + p.fw, p.upx), 0), p.inW) - inX;
        int filterX = midX + p.fw - (inX + 1) * p.upx;
        if (p.flip) filterX = p.fw - 1 - filterX;
        int stepX = p.flip ? p.upx : -p.upx;
        int stepY = p.flip ? p.upy : -p.upy;

        const float* xp = p.x + inX * p.xs_w + inY * p.xs_h + c * p.xs_c + n * p.xs_n;
        float v = 0.f;
        for (int yy = 0; yy < h; yy++) {
            int fy = filterY + yy * stepY;
            for (int xx = 0; xx < w; xx++) {
                int fx = filterX + xx * stepX;
                v += xp[xx * p.xs_w + yy * p.xs_h] * fl[fy * p.fw + fx];
            }
        }
        p.y[outX * p.ys_w + outY * p.ys_h + c * p.ys_c + n * p.ys_n] = epilogue(p, v, c);
    }
}

// channels_last: one float4 of channels per thread; threads run along C then outX.
__global__ __launch_bounds__(256) void upfirdn2d_nhwc4_kernel(UpfirdnParams p) {
    __shared__ float fl[LDETR_MAX_TAPS];
    stage_filter(p, fl);
    const int C4 = p.C >> 2;
    const long total = (long)p.N * p.outH * p.outW * C4;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int c = (int)(idx % C4) << 2;
        long t = idx / C4;
        int outX = (int)(t % p.outW);
        t /= p.outW;
        int outY = (int)(t % p.outH);
        int n = (int)(t / p.outH);

        int midY = outY * p.downy + p.upy - 1 - p.pady0;
        int inY = min(max(floor_div(midY, p.upy), 0), p.inH);
        int h = min(max(floor_div(midY + p.fh, p.upy), 0), p.inH) - inY;
        int filterY = midY + p.fh - (inY + 1) * p.upy;
        if (p.flip) filterY = p.fh - 1 - filterY;
        int midX = outX * p.downx + p.upx - 1 - p.padx0;
        int inX = min(max(floor_div(midX, p.upx), 0), p.inW);
        int w = min(max(floor_div(midX + p.fw, p.upx), 0), p.inW) - inX;
        int filterX = midX + p.fw - (inX + 1) * p.upx;
        if (p.flip) filterX = p.fw - 1 - filterX;
        int stepX = p.flip ? p.upx : -p.upx;
        int stepY = p.flip ? p.upy : -p.upy;

        const float* xp = p.x + inX * p.xs_w + inY * p.xs_h + c + n * p.xs_n;
        float4 v = make_float4(0, 0, 0, 0);
        for (int yy = 0; yy < h; yy++) {
            int fy = filterY + yy * stepY;
            for (int xx = 0; xx < w; xx++) {
                float fv = fl[fy * p.fw + filterX + xx * stepX];
                float4 xv = *reinterpret_cast<const float4*>(xp + xx * p.xs_w + yy * p.xs_h);
                v.x += xv.x * fv; v.y += xv.y * fv; v.z += xv.z * fv; v.w += xv.w * fv;
            }
        }
        float4 o;
        o.x = epilogue(p, v.x, c); o.y = epilogue(p, v.y, c + 1);
        o.z = epilogue(p, v.z, c + 2); o.w = epilogue(p, v.w, c + 3);
        *reinterpret_cast<float4*>(p.y + outX * p.ys_w + outY * p.ys_h + c + n * p.ys_n) = o;
    }
}

// The StyleGAN2 up-layer FIR (upfirdn2d.py:191-198 with up = down = 1, 4x4 taps; forward pad [1,1,1,1] on the (2r+1)^2 transposed-conv
// output, backward pad [2,2,2,2] on the (2r)^2 gradient), channels_last.  HBM-bound: 8 bytes per output element.
// One thread = 4 channels x TC = 2 output columns x TR output rows, INPUT-stationary: the strip's TR + 3 input rows are walked once, each row
// ((TC+3) float4, double-buffered in registers) is added into the up to four output rows it touches and is dead after that; the four live output
// rows are the only state.  An input float4 is requested ~3x per output from the vector L1 instead of 16x (the generic kernel's one-output-per-
// thread form saturates it at 1.7 TB/s); lanes run along channels, then column blocks: every request is a fully used 64..128-byte segment.
// Per output the fma chain is (jy, jx)-ordered from 0 -- the order of the generic kernel and of the output-stationary window form of rounds 2-4
// (177 VGPRs, 2 waves per SIMD), so results are bit-identical to both.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void fir4x4_nhwc4_kernel(UpfirdnParams p, int TR, int xcd_blocks) {
    constexpr int TC = 2, NB = 2;
    // taps as wave-uniform scalars: tap applied to the input sample at window offset (jy, jx) = f[3-jy][3-jx], or f[jy][jx] when flipped
    float ft[4][4];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int ky = i >> 2, kx = i & 3;
        const int sy = p.flip ? ky : 3 - ky, sx = p.flip ? kx : 3 - kx;
        ft[ky][kx] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.f[sx * p.fs_w + sy * p.fs_h] * p.gain)));
    }
    const int C4 = p.C >> 2;
    const int XB = (p.outW + TC - 1) / TC, YB = (p.outH + TR - 1) / TR;
    const long total = (long)p.N * YB * XB * C4;
    // blocks are dealt to the 8 XCDs round-robin: XCD k takes the k-th contiguous eighth of the strips, so the 3 halo rows two vertically
    // adjacent strips share are fetched into ONE L2 (xcd_blocks = blocks per XCD, 0 = identity)
    const long blk = xcd_blocks ? (long)(blockIdx.x & 7) * xcd_blocks + (blockIdx.x >> 3) : blockIdx.x;
    long idx = blk * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C4) << 2;
    long t = idx / C4;
    const int x0 = (int)(t % XB) * TC;
    t /= XB;
    const int y0 = (int)(t % YB) * TR;
    const int n = (int)(t / YB);
    const float* xb = p.x + c + n * p.xs_n;
    float* yb = p.y + c + n * p.ys_n;
    const int ix0 = x0 - p.padx0, iy0 = y0 - p.pady0;
    bool colok[TC + 3];
#pragma unroll
    for (int j = 0; j < TC + 3; j++) colok[j] = (unsigned)(ix0 + j) < (unsigned)p.inW;
    float4 row[NB][TC + 3];      // NB - 1 rows of loads in flight ahead of the row being consumed
    auto load_row = [&](int slot, int iy) {
        const bool rowok = (unsigned)iy < (unsigned)p.inH;
        const float* rp = xb + (long)iy * p.xs_h + (long)ix0 * p.xs_w;
#pragma unroll
        for (int j = 0; j < TC + 3; j++)
            row[slot][j] = (rowok && colok[j]) ? *reinterpret_cast<const float4*>(rp + j * p.xs_w) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.has_act && p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + c);
    float4 acc[4][TC];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int ox = 0; ox < TC; ox++) acc[i][ox] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int nrows = TR + 3;
#pragma unroll
    for (int i = 0; i < NB - 1; i++) load_row(i, iy0 + i);
    for (int r0 = 0; r0 < nrows; r0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = r0 + k;
            if (r >= nrows) break;
            load_row((k + NB - 1) % NB, r + NB - 1 < nrows ? iy0 + r + NB - 1 : -1);
            // input row r is tap row jy of output row oy = r - jy (accumulator slot (k - jy) & 3); for a fixed output row the rows arrive in the order
            // jy = 0, 1, 2, 3.  Slots of rows above / below the strip collect sums that are never stored.
#pragma unroll
            for (int jy = 3; jy >= 0; jy--) {
#pragma unroll
                for (int ox = 0; ox < TC; ox++) {
                    float4 v = acc[(k - jy) & 3][ox];
                    f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};      // packed fp32 FMAs (v_pk_fma_f32): the same two roundings per element pair
#pragma unroll
                    for (int jx = 0; jx < 4; jx++) {
                        const float4 xv = row[k % NB][ox + jx];
                        const f32x2 fv = {ft[jy][jx], ft[jy][jx]};
                        lo = __builtin_elementwise_fma(f32x2{xv.x, xv.y}, fv, lo); hi = __builtin_elementwise_fma(f32x2{xv.z, xv.w}, fv, hi);
                    }
                    acc[(k - jy) & 3][ox] = make_float4(lo.x, lo.y, hi.x, hi.y);
                }
            }
            const int oy = r - 3;      // complete with this row (slot (k + 1) & 3)
            if (oy >= 0 && y0 + oy < p.outH) {
#pragma unroll
                for (int ox = 0; ox < TC; ox++) {
                    float4 v = acc[(k + 1) & 3][ox];
                    if (p.has_act) {
                        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                        v.x = (v.x > 0.f ? v.x : v.x * p.act_alpha) * p.act_gain; v.y = (v.y > 0.f ? v.y : v.y * p.act_alpha) * p.act_gain;
                        v.z = (v.z > 0.f ? v.z : v.z * p.act_alpha) * p.act_gain; v.w = (v.w > 0.f ? v.w : v.w * p.act_alpha) * p.act_gain;
                    }
                    if (x0 + ox < p.outW)
                        *reinterpret_cast<float4*>(yb + (long)(y0 + oy) * p.ys_h + (long)(x0 + ox) * p.ys_w) = v;
                }
            }
#pragma unroll
            for (int ox = 0; ox < TC; ox++) acc[(k + 1) & 3][ox] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

}  // namespace ldetr

extern "C" int ldetr_upfirdn2d_f32(const float* x, const float* f, float* y,
                                   int N, int C, int inH, int inW, const int64_t* x_strides_nchw,
                                   int fh, int fw, int64_t f_stride_h, int64_t f_stride_w,
                                   int upx, int upy, int downx, int downy,
                                   int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                                   int outH, int outW, const int64_t* y_strides_nchw,
                                   const float* act_bias, int has_act, float act_alpha, float act_gain,
                                   void* stream) {
    using namespace ldetr;
    LDETR_CHECK(x && f && y, "upfirdn2d: null pointer");
    LDETR_CHECK(N > 0 && C > 0 && inH > 0 && inW > 0, "upfirdn2d: x has zero size");
    LDETR_CHECK(fh >= 1 && fw >= 1, "upfirdn2d: f must be at least 1x1");
    LDETR_CHECK(fh * fw <= LDETR_MAX_TAPS, "upfirdn2d: filter larger than %d taps is unsupported", LDETR_MAX_TAPS);
    LDETR_CHECK(upx >= 1 && upy >= 1, "upfirdn2d: upsampling factor must be at least 1");
    LDETR_CHECK(downx >= 1 && downy >= 1, "upfirdn2d: downsampling factor must be at least 1");
    int eW = (inW * upx + padx0 + padx1 - fw + downx) / downx;
    int eH = (inH * upy + pady0 + pady1 - fh + downy) / downy;
    LDETR_CHECK(eW >= 1 && eH >= 1, "upfirdn2d: output must be at least 1x1");
    LDETR_CHECK(eW == outW && eH == outH, "upfirdn2d: output size mismatch (expected %dx%d)", eH, eW);
    UpfirdnParams p;
    p.x = x; p.f = f; p.y = y;
    p.bias = act_bias; p.has_act = has_act; p.act_alpha = act_alpha; p.act_gain = act_gain;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip ? 1 : 0; p.gain = gain;
    p.inW = inW; p.inH = inH; p.C = C; p.N = N;
    p.xs_n = x_strides_nchw[0]; p.xs_c = x_strides_nchw[1]; p.xs_h = x_strides_nchw[2]; p.xs_w = x_strides_nchw[3];
    p.fw = fw; p.fh = fh; p.fs_w = f_stride_w; p.fs_h = f_stride_h;
    p.outW = outW; p.outH = outH;
    p.ys_n = y_strides_nchw[0]; p.ys_c = y_strides_nchw[1]; p.ys_h = y_strides_nchw[2]; p.ys_w = y_strides_nchw[3];
    hipStream_t st = (hipStream_t)stream;
    bool nhwc4 = (p.xs_c == 1 && p.ys_c == 1 && (C % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 &&
                  p.xs_w % 4 == 0 && p.xs_h % 4 == 0 && p.xs_n % 4 == 0 && p.ys_w % 4 == 0 && p.ys_h % 4 == 0 &&
                  p.ys_n % 4 == 0);
    long total = nhwc4 ? (long)N * outH * outW * (C / 4) : (long)N * C * outH * outW;
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 32) grid = 256 * 32;
    if (nhwc4 && upx == 1 && upy == 1 && downx == 1 && downy == 1 && fw == 4 && fh == 4 && (long)outH * outW >= 64) {
        // input-stationary register form: 2 columns x TR rows per thread, TR = 8 (a strip re-reads 3 rows of its neighbour), 4 where threads are
        // scarce; strips dealt to the XCDs in contiguous eighths.  Round-5 sweep (profiles/r05_fir_sweep.txt): columns per thread 2 / 4, strip
        // height 4..64, 1 or 3 rows of loads in flight, 2..4 waves per SIMD, packed or scalar FMAs all land at 4.9-5.5 TB/s on 16 x 32 x 256^2;
        // only the XCD mapping moved it (+3..8 %): what bounds this kernel is none of occupancy, halo traffic or VALU issue
        const bool big = (long)N * outH * outW * (C / 4) >= (1L << 20);
        const int TC = 2, TR = big ? 8 : 4;
        const long threads = (long)N * ((outH + TR - 1) / TR) * ((outW + TC - 1) / TC) * (C / 4);
        int g = (int)((threads + 255) / 256), xcd_blocks = 0;
        if (g >= 64) { g = (g + 7) / 8 * 8; xcd_blocks = g / 8; }
        hipLaunchKernelGGL(fir4x4_nhwc4_kernel, g, 256, 0, st, p, TR, xcd_blocks);
    } else if (nhwc4) hipLaunchKernelGGL(upfirdn2d_nhwc4_kernel, grid, 256, 0, st, p);
    else hipLaunchKernelGGL(upfirdn2d_planar_kernel, grid, 256, 0, st, p);
    return check_launch("upfirdn2d");
}
