// ResNet stem: 7x7 / stride 2 / pad 3 convolution of a 3-channel NCHW image into 64 channels, FrozenBN affine and ReLU fused
// (torchvision resnet50.conv1 + bn1 + relu as the reference's Backbone runs them, training/detr_backbone.py:55-114).
//
// As an implicit GEMM this layer is M = N*OH*OW pixels x 64 channels over K = 147 (3 x 7 x 7): 4.9 GFLOP at 16 x 256 x 256, which
// the engine's scalar-gather operand view ran in 155 us (31 TFLOP/s) — every A element is its own 4-byte load with its own tap
// decode.  Here one block owns 8 x 32 output pixels: the 21 x 70 x 3 input patch they read is loaded ONCE into LDS with row-contiguous
// global loads, the 147 x 64 weights once in MFMA operand order, and the contraction runs out of LDS with compile-time offsets:
//   k = (c, ky, kx) with kx padded to 8 (kx = 7 carries zero weights); one v_mfma_f32_32x32x2_f32 takes (kx, kx + 1) of one (c, ky),
//   so the upper 32 lanes read the same patch row one column to the right — even / odd LDS banks, conflict-free, no VALU addressing.
// A wave owns 2 output rows x 32 columns x 64 channels (2 x 2 accumulators): per k-pair 2 + 2 LDS reads feed 4 MFMAs.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct StemParams {
    const float* x; long sn, sc, sh, sw; int N, H, W;
    const float* w;               // [64][7][7][3]
    const float* scale; const float* shift; int relu;
    float* y; long ldy; int OH, OW;
};

constexpr int ST_ROWS = 8, ST_COLS = 32;                 // output tile
constexpr int ST_PH = 2 * ST_ROWS + 5, ST_PW = 2 * ST_COLS + 6;   // 21 x 70 input patch (one spare column: kx = 7 reads it, weight 0)
constexpr int ST_KR = 3 * 7 * 4;                         // (c, ky, kx pair) rows of the weight image

__global__ __launch_bounds__(256) void stem_conv7x7_kernel(StemParams p) {
    __shared__ float patch[3][ST_PH][ST_PW];
    __shared__ float wimg[ST_KR][2][2][32];              // [(c, ky, kx / 2)][channel half j][kx & 1][channel % 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cl = lane & 31, kl = lane >> 5;
    const int tiles_x = (p.OW + ST_COLS - 1) / ST_COLS, tiles_y = (p.OH + ST_ROWS - 1) / ST_ROWS;
    const int total = p.N * tiles_y * tiles_x;
    // weights in operand order, once per block (a block walks several tiles)
    for (int i = tid; i < ST_KR * 128; i += 256) {
        const int row = i >> 7, e = i & 127, j = e >> 6, h = (e >> 5) & 1, ch = j * 32 + (e & 31);
        const int c = row / 28, ky = (row / 4) % 7, kx = 2 * (row & 3) + h;
        wimg[row][j][h][e & 31] = kx < 7 ? p.w[((long)(ch * 7 + ky) * 7 + kx) * 3 + c] : 0.f;
    }
    float sc[2], sh[2];
#pragma unroll
    for (int j = 0; j < 2; j++) { sc[j] = p.scale ? p.scale[j * 32 + cl] : 1.f; sh[j] = p.shift ? p.shift[j * 32 + cl] : 0.f; }
    const float* pa0 = &patch[0][2 * (2 * wave)][2 * cl + kl];        // output row 2 wave, column cl, tap (0, kl)
    const float* pa1 = pa0 + 2 * ST_PW;                                // output row 2 wave + 1
    const float* pw = &wimg[0][0][kl][cl];
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        int b = tile;
        const int tx = b % tiles_x; b /= tiles_x;
        const int ty = b % tiles_y; const int n = b / tiles_y;
        const int oy0 = ty * ST_ROWS, ox0 = tx * ST_COLS;
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        // input patch: rows of 70 contiguous floats per channel (zero outside the image = the padding)
        const float* xn = p.x + (long)n * p.sn;
        __syncthreads();                                               // the previous tile's reads of the patch are done
        for (int i = tid; i < 3 * ST_PH * ST_PW; i += 256) {
            const int c = i / (ST_PH * ST_PW), r = (i / ST_PW) % ST_PH, q = i % ST_PW;
            const int gy = iy0 + r, gx = ix0 + q;
            float v = 0.f;
            if ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) v = xn[(long)c * p.sc + (long)gy * p.sh + (long)gx * p.sw];
            patch[c][r][q] = v;
        }
        __syncthreads();

        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[m][j][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int ky = 0; ky < 7; ky++)
#pragma unroll
                for (int kp = 0; kp < 4; kp++) {
                    const int off = (c * ST_PH + ky) * ST_PW + 2 * kp, row = (c * 7 + ky) * 4 + kp;
                    const float a0 = pa0[off], a1 = pa1[off];
                    const float b0 = pw[row * 128], b1 = pw[row * 128 + 64];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
        // epilogue: accumulator rows are the 32 output columns of one output row, accumulator columns the channels
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const int oy = oy0 + 2 * wave + m;
            if (oy >= p.OH) continue;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int ch = j * 32 + cl;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * kl;
                    if (ox >= p.OW) continue;
                    float v = acc[m][j][r] * sc[j] + sh[j];
                    if (p.relu) v = v > 0.f ? v : 0.f;
                    p.y[((long)(n * p.OH + oy) * p.OW + ox) * p.ldy + ch] = v;
                }
            }
        }
    }
}

// Called by ldetr_conv2d_fwd_f32: returns -1 if the problem is not the stem's, else the launch status.
int try_launch_stem_conv(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW, int stride, int pad,
                         float* y, long ldy, int OH, int OW, const float* in_scale, const ldetr_epilogue* ep, hipStream_t st) {
    if (xt->C != 3 || Cout != 64 || KH != 7 || KW != 7 || stride != 2 || pad != 3 || in_scale || ldy < 64) return -1;
    if (ep && (ep->samp_scale || ep->residual || ep->mask_mode || ep->p_drop > 0.f || ep->accumulate || ep->a_rowsum || ep->alpha != 1.f ||
               ep->out_scale != 1.f || (ep->act != 0 && ep->act != 1)))
        return -1;
    StemParams p;
    p.x = x; p.sn = xt->sn; p.sc = xt->sc; p.sh = xt->sh; p.sw = xt->sw; p.N = xt->N; p.H = xt->H; p.W = xt->W;
    p.w = w; p.scale = ep ? ep->col_scale : nullptr; p.shift = ep ? ep->col_bias : nullptr; p.relu = ep ? ep->act == 1 : 0;
    p.y = y; p.ldy = ldy; p.OH = OH; p.OW = OW;
    const long blocks = (long)xt->N * ((OH + ST_ROWS - 1) / ST_ROWS) * ((OW + ST_COLS - 1) / ST_COLS);
    if (blocks <= 0 || blocks > 0x7fffffffL) return -1;
    note_engine_launch(6, 0, 64, 0, 4, 1, 0, 1, blocks < 512 ? blocks : 512, 0);
    hipLaunchKernelGGL(stem_conv7x7_kernel, (int)(blocks < 512 ? blocks : 512), 256, 0, st, p);   // two blocks per CU, each walks its tiles
    return check_launch("stem_conv7x7");
}

}  // namespace ldetr
