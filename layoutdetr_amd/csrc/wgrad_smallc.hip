// Weight gradient of a 3x3 / stride-1 / pad-1 convolution with 32 input and 32 output channels on a large pixel grid — the
// 256x256 StyleGAN2 Decoder layer (networks_stylegan2.py:482-497: b256.conv1), whose gradient the per-tap implicit-GEMM path runs at
// 23 TFLOP/s: a 64x64 tile holds a 32x32 output, 3/4 of every MFMA is padding.
//
//   dw[co][ty][tx][ci] = sum over pixels p of dy[p][co] * x[p + (ty-1, tx-1)][ci]        (zero outside the image)
//
// NHWC makes both MFMA operands of v_mfma_f32_32x32x2_f32 plain coalesced loads with NO LDS staging and no transposition:
//   A (32 rows = co, 2 k = two neighbouring pixels): lane l holds dy[p + l/32][l%32]   = dyflat[32 p + l]
//   B (2 k, 32 cols = ci)                          : lane l holds x[p' + l/32][l%32]   = xflat[32 p' + l],  p' = p shifted by the tap
// so a wave walks along image rows two pixels at a time, issues one 256-byte load for A and nine for B (the nine taps; neighbouring
// taps hit the same cache lines) and nine MFMAs into nine 32x32 accumulators.  No barriers in the main loop; the four waves of a
// block merge through 36 KB of LDS at the end and add the block's 9216 partial sums to dw with fp32 atomics.  Per-sample operand
// scales (style modulation of x, demodulation of dy: ldetr_conv2d_bwd_weight_f32's x_scale / dy_scale) multiply the partial sums
// once per block: a block never crosses a sample.
// MFMA-bound by construction: 2*9*32*32 flop per pixel -> 123 us at the f32 matrix peak for 16 x 256 x 256 pixels; HBM: x and dy once.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct WgradSmallParams {
    const float* x; const float* dy; float* dw;
    const float* x_scale; long x_scale_ld; const float* dy_scale; long dy_scale_ld;
    int N, H, W, rows_per_wave;
};

__global__ __launch_bounds__(256, 2) void wgrad_c32_3x3_kernel(WgradSmallParams p) {
    __shared__ float red[9 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kl = lane >> 5;
    const long unit = (long)blockIdx.x * 4 + wave;                  // this wave's strip of rows_per_wave image rows
    const long row0 = unit * p.rows_per_wave;                       // global row index (n * H + y)
    const int n = (int)(row0 / p.H), y0 = (int)(row0 - (long)n * p.H);
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    // x's descriptor starts (W + 1) pixels BEFORE the sample: every in-image tap offset is then non-negative as a vector offset (the
    // range check looks at the vector offset alone: a negative one plus a positive scalar offset would read as out of range)
    const int shiftx = (p.W + 1) * 128;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + (long)n * p.H * p.W * 32 - (p.W + 1) * 32, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy) + (long)n * p.H * p.W * 32, 0, 0x7fffffff, 0x00020000);
    const int OOB = (int)0x80000000;
    if (row0 < (long)p.N * p.H) {
        for (int ry_i = 0; ry_i < p.rows_per_wave; ry_i++) {
            const int y = y0 + ry_i;
            const bool up = y > 0, dn = y + 1 < p.H;
            const int rowoff = y * p.W;                             // pixel index of (y, 0) inside the sample
            // Addressing without per-load VALU work: each lane keeps ONE byte offset per tap for the row (its pixel column 0 / 1 of a
            // pair, shifted by the tap), the pair position goes into the buffer load's SCALAR offset.  A tap row outside the image, and
            // the one column that falls off the left / right edge in the first / last pair, are an out-of-range vector offset (reads 0).
            const int voA = (rowoff * 32 + lane) * 4;
            int vo[9], vo_first[9], vo_last[9];
#pragma unroll
            for (int t = 0; t < 9; t++) {
                const int ty = t / 3, tx = t % 3;
                const bool rowok = (ty == 0) ? up : ((ty == 2) ? dn : true);
                const int base = ((rowoff + (ty - 1) * p.W + tx - 1) * 32 + lane) * 4 + shiftx;
                vo[t] = rowok ? base : OOB;
                vo_first[t] = (rowok && !(tx == 0 && kl == 0)) ? base : OOB;      // pair at x = 0: column -1 does not exist
                vo_last[t] = (rowok && !(tx == 2 && kl == 1)) ? base : OOB;       // pair at x = W-2: column W does not exist
            }
            auto loadA = [&](int xx) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(ry, voA, xx * 128, 0)); };
            auto loadB = [&](int xx, int t) {
                const int v = (xx == 0) ? vo_first[t] : ((xx == p.W - 2) ? vo_last[t] : vo[t]);      // block-uniform selects
                return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, v, xx * 128, 0));
            };
            // two pixel pairs in flight ahead of the one being multiplied (software pipeline of depth PD over the row)
            constexpr int PD = 2;
            float a_q[PD], b_q[PD][9];
#pragma unroll
            for (int d = 0; d < PD; d++) {
                a_q[d] = loadA(2 * d);
#pragma unroll
                for (int t = 0; t < 9; t++) b_q[d][t] = loadB(2 * d, t);
            }
            for (int xx = 0; xx < p.W; xx += 2 * PD) {
#pragma unroll
                for (int d = 0; d < PD; d++) {
                    const float a = a_q[d];
                    float b[9];
#pragma unroll
                    for (int t = 0; t < 9; t++) b[t] = b_q[d][t];
                    const int nx = xx + 2 * d + 2 * PD;              // refill this slot with the pair PD steps ahead
                    if (nx < p.W) {
                        a_q[d] = loadA(nx);
#pragma unroll
                        for (int t = 0; t < 9; t++) b_q[d][t] = loadB(nx, t);
                    }
                    if (xx + 2 * d < p.W) {
#pragma unroll
                        for (int t = 0; t < 9; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // per-sample operand scales, once per wave: acc[t][r] is element (co = (r&3) + 8 (r>>2) + 4 kl, ci = lane & 31)
    const int ci = lane & 31;
    const float sc = (p.x_scale && row0 < (long)p.N * p.H) ? p.x_scale[(long)n * p.x_scale_ld + ci] : 1.f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int co = (r & 3) + 8 * (r >> 2) + 4 * kl;
        const float s = sc * ((p.dy_scale && row0 < (long)p.N * p.H) ? p.dy_scale[(long)n * p.dy_scale_ld + co] : 1.f);
#pragma unroll
        for (int t = 0; t < 9; t++) acc[t][r] *= s;
    }
    // merge the four waves in LDS (wave after wave: each lane owns its own 144 slots), then one atomic per output element and block
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9; t++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int idx = (t * 16 + r) * 64 + lane;
                    red[idx] = (w == 0 ? 0.f : red[idx]) + acc[t][r];
                }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 9 * 1024; i += 256) {
        const int l = i & 63, tr = i >> 6, r = tr & 15, t = tr >> 4;
        const int co = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), c = l & 31;
        atomicAdd(p.dw + (long)co * 288 + t * 32 + c, red[i]);
    }
}

// -> 1 if the launch was taken, 0 if the shape does not fit (the caller falls back to the per-tap implicit GEMM), < 0 on error
int try_launch_wgrad_smallc(const float* x, const ldetr_tensor4* xt, const float* dy, const ldetr_tensor4* dyt, float* dw, int KH, int KW, int stride, int pad,
                            const float* x_scale, int64_t x_scale_ld, const float* dy_scale, int64_t dy_scale_ld, hipStream_t st) {
    const int N = xt->N, H = xt->H, W = xt->W;
    if (KH != 3 || KW != 3 || stride != 1 || pad != 1 || xt->C != 32 || dyt->C != 32 || dyt->H != H || dyt->W != W || (W & 1)) return 0;
    if (xt->sc != 1 || xt->sw != 32 || xt->sh != (long)W * 32 || xt->sn != (long)H * W * 32) return 0;
    if (dyt->sc != 1 || dyt->sw != 32 || dyt->sh != (long)W * 32 || dyt->sn != (long)H * W * 32) return 0;
    // The kernel's border handling needs a first AND a distinct last column pair (W >= 4); its vector offsets reach (W + 1) pixels past either
    // end of a sample's rows, so (H*W + 2*(W+1)) * 128 bytes must stay below 2^31; small grids stay on the GEMM path.  (fp32 atomics merge the
    // blocks' partial sums: the result is summation-order dependent from run to run, like the engine's atomic split-K path.)
    if (W < 4 || ((long)H * W + 2L * (W + 1)) * 128 >= 0x7fffffffL || (long)N * H * W < (1L << 18)) return 0;
    // strips of whole rows, never crossing a sample; about 4 waves per SIMD-pair of the chip
    static const int target = 2048;   // strips: 2 waves per SIMD
    int rpw = (int)(((long)N * H + target - 1) / target);
    if (rpw < 1) rpw = 1;
    while (rpw > 1 && H % rpw != 0) rpw--;
    WgradSmallParams p;
    p.x = x; p.dy = dy; p.dw = dw; p.x_scale = x_scale; p.x_scale_ld = x_scale_ld; p.dy_scale = dy_scale; p.dy_scale_ld = dy_scale_ld;
    p.N = N; p.H = H; p.W = W; p.rows_per_wave = rpw;
    const long units = ((long)N * H) / rpw;
    hipLaunchKernelGGL(wgrad_c32_3x3_kernel, (int)((units + 3) / 4), 256, 0, st, p);
    note_engine_launch(5, 32, 32, 0, 4, 1, 0, 1, (units + 3) / 4, rpw);
    return check_launch("wgrad_smallc") == 0 ? 1 : -1;
}

}  // namespace ldetr
