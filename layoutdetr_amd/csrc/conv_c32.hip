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

bool engine_split_enabled();      // gemm_conv.hip: ldetr_set_split_bf16 / LDETR_DEBUG="SPLIT_BF16=0"

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

// The same convolution on the bf16 matrix pipe with the exact three-way operand split (x = hi + mid + lo, six products per k16 slab:
// fp32-equivalent results, see gemm_conv.hip "SPLIT").  The filter bank is split ONCE per block while it is folded into LDS -- the main
// loop's B operand is three 16-byte LDS reads per k16 slab with no VALU -- and only the activations are split per tap (72 VALU per
// lane against 16 v_mfma_f32_32x32x2_f32 = 1024 matrix-pipe cycles saved: 12 v_mfma_f32_32x32x16_bf16 = 384 cycles take their place).
//   A (32 pixels x 16 k, bf16): lane (pixel cl, k-group kl) splits channels [16 m + 8 kl, + 8), m = 0, 1, of its pixel;
//   B (16 k x 32 output channels): bank[tap][m][part][slot = 32 kl + o] = 8 consecutive reduction channels as one 16-byte slot,
//                                  so lane l reads slot l: conflict-free ds_read_b128.
// Non-finite values: Inf splits into Inf + NaN + NaN.  A segment whose accumulator comes out non-finite is recomputed on the f32 pipe
// (v_mfma_f32_32x32x2_f32, operands straight from global memory: cold path), so the kernel returns what conv3x3_c32_kernel returns.
__global__ __launch_bounds__(256) void conv3x3_c32_split_kernel(ConvC32Params p) {
    __shared__ __attribute__((aligned(16))) unsigned wl[9 * 2 * 3 * 64 * 4];      // 55296 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kl = lane >> 5;
    const long gw = (long)blockIdx.x * 4 + wave;
    const int n = (int)(((long)blockIdx.x * 4) / p.waves_per_sample), wv = (int)(gw - (long)n * p.waves_per_sample);
    if (n >= p.N) return;
    const float* ks = p.k_scale ? p.k_scale + (long)n * p.k_scale_ld : nullptr;
    // ---- the block's filter bank, factors folded in, split into its three bf16 parts: 1152 slots of 8 reduction channels
    for (int sidx = tid; sidx < 1152; sidx += 256) {
        const int o = sidx & 31, g = (sidx >> 5) & 1, m = (sidx >> 6) & 1, t = sidx >> 7;
        const int k0 = 16 * m + 8 * g;
        float v[8];
        if (!p.transposed) {
            const float4 w0 = *reinterpret_cast<const float4*>(p.w + (o * 9 + t) * 32 + k0), w1 = *reinterpret_cast<const float4*>(p.w + (o * 9 + t) * 32 + k0 + 4);
            v[0] = w0.x; v[1] = w0.y; v[2] = w0.z; v[3] = w0.w; v[4] = w1.x; v[5] = w1.y; v[6] = w1.z; v[7] = w1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = p.w[((k0 + j) * 9 + (8 - t)) * 32 + o];      // reduce over the forward's output channels, taps flipped
        }
        if (ks) {
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] *= ks[k0 + j];
        }
        unsigned h[4], md[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) split2_bf16(v[2 * j], v[2 * j + 1], h[j], md[j], lo[j]);
        unsigned* dst = wl + (((t * 2 + m) * 3) * 64 + g * 32 + o) * 4;
        *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(dst + 64 * 4) = make_uint4(md[0], md[1], md[2], md[3]);
        *reinterpret_cast<uint4*>(dst + 2 * 64 * 4) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
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
    auto voff = [&](int tile, int g) {      // byte offset of channel 8 kl of this lane's pixel for tap g, or out of range
        const int y = tile / segs, x0 = (tile - y * segs) << 5;
        const int yy = y + g / 3 - 1, xx = x0 + cl + g % 3 - 1;
        return (tile < t1 && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) ? ((yy * p.W + xx) * 32 + 8 * kl) * 4 : OOB;
    };
    auto loadA = [&](int vo, float (&a)[16]) {      // a[0..7] = channels 8 kl .. + 7, a[8..15] = channels 16 + 8 kl .. + 7
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, (j >> 1) * 64 + (j & 1) * 16, 0);
            a[4 * j] = __int_as_float(v[0]); a[4 * j + 1] = __int_as_float(v[1]); a[4 * j + 2] = __int_as_float(v[2]); a[4 * j + 3] = __int_as_float(v[3]);
        }
    };
    int woff = lane * 4;
    float a[3][16];
    loadA(voff(t0, 0), a[0]);
    loadA(voff(t0, 1), a[1]);
    for (int tile = t0; tile < t1; tile++) {
        c32_acc_t acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        asm volatile("" : "+v"(woff));      // keep the bank reads inside the loop (see conv3x3_c32_kernel)
        const unsigned* const wrow = wl + woff;
#pragma unroll
        for (int g = 0; g < 9; g++) {
            if (g < 7) loadA(voff(tile, g + 2), a[(g + 2) % 3]);
            else loadA(voff(tile + 1, g - 7), a[(g + 2) % 3]);
#pragma unroll
            for (int m = 0; m < 2; m++) {
                unsigned ah[4], am[4], al[4];
#pragma unroll
                for (int j = 0; j < 4; j++) split2_bf16(a[g % 3][8 * m + 2 * j], a[g % 3][8 * m + 2 * j + 1], ah[j], am[j], al[j]);
                const uint4 A0 = make_uint4(ah[0], ah[1], ah[2], ah[3]), A1 = make_uint4(am[0], am[1], am[2], am[3]), A2 = make_uint4(al[0], al[1], al[2], al[3]);
                const uint4 B0 = *reinterpret_cast<const uint4*>(wrow + ((g * 2 + m) * 3 + 0) * 256);
                const uint4 B1 = *reinterpret_cast<const uint4*>(wrow + ((g * 2 + m) * 3 + 1) * 256);
                const uint4 B2 = *reinterpret_cast<const uint4*>(wrow + ((g * 2 + m) * 3 + 2) * 256);
#define C32_MF(X, Y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(&X), *reinterpret_cast<const bf16x8_t*>(&Y), acc, 0, 0, 0)
                C32_MF(A1, B1); C32_MF(A0, B2); C32_MF(A2, B0); C32_MF(A0, B1); C32_MF(A1, B0); C32_MF(A0, B0);      // smallest terms first
#undef C32_MF
            }
        }
        float chk = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) chk = __builtin_fmaf(acc[r], 0.f, chk);
        if (__builtin_expect(__ballot(chk != chk) != 0ull, 0)) {
            // cold path: this segment again on the f32 pipe, operands from global memory (step s of a tap: lane k-group kl takes reduction
            // channel c = (s < 8 ? 8 kl + s : 16 + 8 kl + s - 8) on both sides)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            for (int g = 0; g < 9; g++) {
                float af[16];
                loadA(voff(tile, g), af);
#pragma unroll 1
                for (int s2 = 0; s2 < 16; s2++) {
                    const int c = (s2 < 8 ? 8 * kl + s2 : 16 + 8 * kl + s2 - 8);
                    float wv_ = !p.transposed ? p.w[(cl * 9 + g) * 32 + c] : p.w[(c * 9 + (8 - g)) * 32 + cl];
                    if (ks) wv_ *= ks[c];
                    float av = 0.f;
#pragma unroll
                    for (int q = 0; q < 16; q++) av = (q == s2) ? af[q] : av;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wv_, acc, 0, 0, 0);
                }
            }
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
    const int N = xt->N, H = xt->H, W = xt->W;
    if (KH != 3 || KW != 3 || stride != 1 || pad != 1 || xt->C != 32 || Cout != 32 || OH != H || OW != W || ldy != 32 || (W & 31)) return 0;
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
    // bf16 pipe with the exact operand split unless the engine's switch puts everything on the f32 MFMAs
    if (engine_split_enabled()) hipLaunchKernelGGL(conv3x3_c32_split_kernel, dim3((unsigned)((long)N * wps / 4)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv3x3_c32_kernel, dim3((unsigned)((long)N * wps / 4)), dim3(256), 0, st, p);
    note_engine_launch(4, 32, 32, 32, 4, 1, engine_split_enabled() ? 1 : 0, 1, (long)N * wps / 4, transposed ? 1 : 0);
    return check_launch("conv3x3_c32") == 0 ? 1 : -1;
}

}  // namespace ldetr
