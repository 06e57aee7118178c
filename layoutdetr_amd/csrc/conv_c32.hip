// 3x3 / stride-1 / pad-1 convolution with 32 input and 32 output channels on a large pixel grid, forward and data gradient: the
// 256x256 layers of the StyleGAN2 Decoder (networks_stylegan2.py:482-497: b256.conv1 and the input gradient of b256's layers;
// modulated_conv2d :30-86).  north_star names this layer ("StyleGAN2 modulated conv at 256x256"): on the LDS-tiled engine its 256x32
// narrow tile ran at 61-68 TFLOP/s (303 us for 16 x 256 x 256 pixels), a third of that spent multiplying the style factors into the
// operand while staging it.
//
//   y[p][o] = sum over taps g and channels k of  x[p + g - (1,1)][k] * wf[o][g][k]            (zero outside the image)
//
// Here the WEIGHTS carry the per-sample factors (wf = w * style[b][k] (* demod[b][o])): 9216 products per wave, once, instead of a
// multiply per staged activation; the block's filter bank (9 taps x 32 x 32 = 36 KB) is written to LDS once, in MFMA operand order,
// and the main loop has no barrier and no LDS stores at all:
//   A (32 pixels of an image row x 2 k): lane l reads 64 contiguous bytes of pixel (x0 + l%32 + gx - 1) -- 4 buffer_load_b128 per tap,
//                                        image borders = out-of-range vector offsets, which the buffer unit answers with zeros;
//   B (2 k x 32 output channels)       : 4 ds_read_b128 per tap from the resident bank (a first version held the bank in 144 VGPRs:
//                                        two waves per SIMD and one tap of prefetch, 180-229 us);
//   9 taps x 16 v_mfma_f32_32x32x2_f32 into ONE 32 x 32 accumulator per 32-pixel segment, the A loads of the next two taps in flight.
// Epilogue per segment: (demod) scale, bias, leaky relu x gain, 128-byte row stores.  The data gradient is the same kernel with the
// filter bank read transposed and flipped (wf[k][2-gy][2-gx][o] * demod[b][k]).
// MFMA-bound by construction: 2*9*32*32 flop per pixel -> 123 us at the f32 matrix peak for 16 x 256 x 256 pixels; HBM: x once (the
// nine taps and three rows of a pixel meet in L1 / L2), y once = 268 MB -> 34 us at 8 TB/s.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct ConvC32Params {
    const float* x; float* y; const float* w;                  // x, y: [N, H, W, 32] packed; w: [32 out][3][3][32 in] (OHWI)
    const float* k_scale; long k_scale_ld;                      // per sample, per input channel of THIS contraction (style / demod), or null
    const float* o_scale; long o_scale_ld;                      // per sample, per output channel (forward: demod), or null
    const float* bias;                                          // per output channel, or null
    int act; float act_alpha, act_gain;                         // 0 none, 2 leaky relu
    int N, H, W, tiles_per_wave, waves_per_sample, transposed;
};

typedef __attribute__((__vector_size__(16 * sizeof(float)))) float c32_acc_t;

constexpr int C32_PITCH = 36;      // floats per (tap, output channel) row of the LDS filter bank: 16-byte aligned, rows 144 B apart

__global__ __launch_bounds__(256) void conv3x3_c32_kernel(ConvC32Params p) {
    __shared__ __attribute__((aligned(16))) float wl[9 * 32 * C32_PITCH];      // wf[g][o][k], 41.5 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kl = lane >> 5;
    const long gw = (long)blockIdx.x * 4 + wave;                 // global wave index; a block never crosses a sample
    const int n = (int)(((long)blockIdx.x * 4) / p.waves_per_sample), wv = (int)(gw - (long)n * p.waves_per_sample);
    if (n >= p.N) return;
    // ---- the block's filter bank with the sample's factors folded in
    {
        const float* ks = p.k_scale ? p.k_scale + (long)n * p.k_scale_ld : nullptr;
        for (int e = tid; e < 9216; e += 256) {
            const int o = e / 288, rem = e - o * 288, g = rem >> 5, k = rem & 31;          // w[o][g][k], e runs contiguously over it
            if (!p.transposed) wl[(g * 32 + o) * C32_PITCH + k] = p.w[e] * (ks ? ks[k] : 1.f);
            else wl[((8 - g) * 32 + k) * C32_PITCH + o] = p.w[e] * (ks ? ks[o] : 1.f);    // data gradient: reduce over the forward's output channels, taps flipped
        }
    }
    __syncthreads();
    const int segs = p.W >> 5, tiles = p.H * segs;
    const int t0 = wv * p.tiles_per_wave, t1 = min(tiles, t0 + p.tiles_per_wave);
    if (t0 >= t1) return;
    const float osc = p.o_scale ? p.o_scale[(long)n * p.o_scale_ld + cl] : 1.f;
    const float bs = p.bias ? p.bias[cl] : 0.f;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long)n * p.H * p.W * 32, 0, 0x7fffffff, 0x00020000);
    float* const yb = p.y + (long)n * p.H * p.W * 32;
    const int OOB = (int)0x80000000;
    auto voff = [&](int tile, int g) {      // byte offset of this lane's 64 bytes of tap g, or out of range
        const int y = tile / segs, x0 = (tile - y * segs) << 5;
        const int yy = y + g / 3 - 1, xx = x0 + cl + g % 3 - 1;
        return (tile < t1 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) ? ((yy * p.W + xx) * 32 + 16 * kl) * 4 : OOB;
    };
    auto loadA = [&](int vo, float (&a)[16]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 16 * j, 0);
            a[4 * j] = __int_as_float(v[0]); a[4 * j + 1] = __int_as_float(v[1]); a[4 * j + 2] = __int_as_float(v[2]); a[4 * j + 3] = __int_as_float(v[3]);
        }
    };
    int woff = cl * C32_PITCH + 16 * kl;
    // three-deep ring of A fragments, two taps ahead of the MFMAs (a tap is 16 MFMAs = 0.43 us: one tap ahead did not cover the memory
    // latency); nine taps per segment = three turns of the ring, so the slot of a tap is a compile-time constant
    float a[3][16];
    loadA(voff(t0, 0), a[0]);
    loadA(voff(t0, 1), a[1]);
    for (int tile = t0; tile < t1; tile++) {
        c32_acc_t acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        asm volatile("" : "+v"(woff));      // the bank is loop-invariant: without this the compiler hoists all 144 operand registers out of the loop (2 waves per SIMD)
        const float* const wrow = wl + woff;
#pragma unroll
        for (int g = 0; g < 9; g++) {
            if (g < 7) loadA(voff(tile, g + 2), a[(g + 2) % 3]);
            else loadA(voff(tile + 1, g - 7), a[(g + 2) % 3]);              // (past the wave's last segment: out of range, reads zeros)
            float b[16];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float4 v = *reinterpret_cast<const float4*>(wrow + g * 32 * C32_PITCH + 4 * j);
                b[4 * j] = v.x; b[4 * j + 1] = v.y; b[4 * j + 2] = v.z; b[4 * j + 3] = v.w;
            }
#pragma unroll
            for (int k = 0; k < 16; k++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g % 3][k], b[k], acc, 0, 0, 0);
        }
        const int y = tile / segs, x0 = (tile - y * segs) << 5;
        float* dst = yb + ((long)y * p.W + x0) * 32 + cl;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = acc[r] * osc + bs;
            if (p.act == 2) v = (v > 0.f ? v : v * p.act_alpha) * p.act_gain;
            dst[((r & 3) + 8 * (r >> 2) + 4 * kl) * 32] = v;
        }
    }
}

// -> 1 if the launch was taken, 0 if the shape / epilogue does not fit (the caller continues on the tiled engine), < 0 on error
int try_launch_conv_c32(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW, int stride, int pad,
                        float* y, long ldy, int OH, int OW, const float* k_scale, long k_scale_ld, const ldetr_epilogue* ep,
                        int transposed, hipStream_t st) {
    static const int on = getenv("LDETR_CONV_C32") ? atoi(getenv("LDETR_CONV_C32")) : 1;
    const int N = xt->N, H = xt->H, W = xt->W;
    if (!on || KH != 3 || KW != 3 || stride != 1 || pad != 1 || xt->C != 32 || Cout != 32 || OH != H || OW != W || ldy != 32 || (W & 31)) return 0;
    if (xt->sc != 1 || xt->sw != 32 || xt->sh != (long)W * 32 || xt->sn != (long)H * W * 32) return 0;
    if ((long)H * W * 128 >= 0x7fffffffL || (long)N * H * W < (1L << 18)) return 0;       // 32-bit offsets inside a sample; small grids stay on the engine
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15) != 0) return 0;
    ConvC32Params p; memset(&p, 0, sizeof(p));
    if (ep) {
        if (ep->col_scale || ep->residual || ep->mask_mode || ep->p_drop > 0.f || ep->accumulate || ep->out_scale != 1.f || ep->alpha != 1.f ||
            !(ep->act == 0 || ep->act == 2))
            return 0;
        p.o_scale = ep->samp_scale; p.o_scale_ld = ep->samp_ld; p.bias = ep->col_bias;
        p.act = ep->act; p.act_alpha = ep->act_alpha; p.act_gain = ep->act_gain;
    }
    p.x = x; p.y = y; p.w = w; p.k_scale = k_scale; p.k_scale_ld = k_scale_ld;
    p.N = N; p.H = H; p.W = W; p.transposed = transposed;
    // about four waves per SIMD of the chip, a whole number of 4-wave blocks per sample
    const int tiles = H * (W >> 5);
    int wps = (4096 + N - 1) / N;
    wps = (wps + 3) & ~3;
    if (wps > tiles) wps = (tiles + 3) & ~3;
    p.waves_per_sample = wps;
    p.tiles_per_wave = (tiles + wps - 1) / wps;
    hipLaunchKernelGGL(conv3x3_c32_kernel, dim3((unsigned)((long)N * wps / 4)), dim3(256), 0, st, p);
    return check_launch("conv3x3_c32") == 0 ? 1 : -1;
}

}  // namespace ldetr
