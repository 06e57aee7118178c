// Exact-f32 MFMA contraction engine for gfx950: dense GEMMs and implicit-GEMM convolutions.
//
// One kernel template covers every dense contraction on the hot path:
//   * nn.Linear fwd / dX / dW                      (training/detr_transformer.py:187-189, networks_detr.py:50-62)
//   * ResNet-50 conv fwd / bwd-data / bwd-weight    (training/detr_backbone.py:98-114; ATen conv2d in the reference)
//   * StyleGAN2 modulated conv, transposed conv     (training/networks_stylegan2.py:30-75, ops/conv2d_resample.py:113-135)
// C[m, n] = sum_k A(m, k) * B(k, n), where A and B are *operand views*: dense row-major / col-major
// matrices, or NHWC gathers (im2col is never materialised).  All activations are NHWC
// (channels_last) so the reduction index (tap, channel) is contiguous along channels.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 products and accumulation (the reference step is
// fp32 end-to-end with TF32 disabled, training/training_loop.py:104-105).  Peak 157 TFLOP/s.
// Tiling: 256 threads = 4 waves (2x2); block tile 128x128 (BK 16) or 64x64 (BK 32); LDS tiles are
// k-major ([k][row], row stride R+2 -> conflict-free MFMA operand reads and transposing writes),
// double-buffered with register staging (global loads for tile t+1 are in flight during the
// MFMAs of tile t; one barrier per k-tile).
#include <cstdlib>
#include <type_traits>
#include <atomic>
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {
int try_launch_stem_conv(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW, int stride, int pad,
                         float* y, long ldy, int OH, int OW, const float* in_scale, const ldetr_epilogue* ep, hipStream_t st);
int try_launch_conv_c32(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW, int stride, int pad,
                        float* y, long ldy, int OH, int OW, const float* k_scale, long k_scale_ld, const ldetr_epilogue* ep,
                        int transposed, hipStream_t st);
static thread_local int64_t t_launches_f32 = 0, t_launches_split = 0;   // ldetr_engine_launch_counts: contraction kernels issued by this thread, by matrix pipe
void note_engine_launch(bool bf16_split_pipe) { (bf16_split_pipe ? t_launches_split : t_launches_f32)++; }   // contraction kernels of other translation units (p3_engine.hip)
int try_launch_wgrad_smallc(const float* x, const ldetr_tensor4* xt, const float* dy, const ldetr_tensor4* dyt, float* dw, int KH, int KW, int stride, int pad,
                            const float* x_scale, int64_t x_scale_ld, const float* dy_scale, int64_t dy_scale_ld, hipStream_t st);

enum {
    OP_KC_DENSE = 0,  // (row r, k): src[r*ld + k]
    OP_KC_CONV = 1,   // row = dst pixel, k = (tap, c): src pixel = dst*stride - pad + tap
    OP_KC_CONVT = 2,  // row = dst pixel of a parity class, k = (tap', c): src pixel = (dst + pad - tap)/stride
    OP_KC_WTAP = 3,   // row = out channel, k = (tap', c) with tap remap: src[r*ld + tap*C + c]
    OP_RC_DENSE = 4,  // (k, row r): src[k*ld + r]
    OP_RC_PIX = 5,    // k = pixel gathered with one fixed tap, r = channel
    OP_RC_WT = 6,     // k = (tap', co): src[co*ld + tap*Cr + r]
    OP_RC_CONVK = 7,  // k = dst pixel, r = (tap, c) scalar gather (3-channel stem weight gradient)
};

struct Operand {
    const float* p;
    long ld;
    int vec;  // 1: float4 loads are legal (alignment + divisibility checked on the host)
    // gather geometry
    int SH, SW;              // source spatial dims
    long sn, sh, sw, sc;     // source strides (elements) for sample / y / x / channel
    int DH, DW;              // destination pixel grid (full grid; parity classes subdivide it)
    int C;                   // channels enumerated by k (KC modes) or the k-channel count (RC_WT)
    int Cr;                  // RC_WT: inner channel count (tap stride in the weight row)
    int stride, pad;
    int KH, KW;
    const float* scale;      // optional per-sample per-channel scale [nsamp][scale_ld]
    long scale_ld;
    int tapped;              // RC_PIX: 1 if this operand takes the per-z tap offset, else centre (dense)
};

struct GemmEpilogue {
    float alpha;
    const float* col_scale;
    const float* col_bias;
    const float* samp_scale;
    long samp_ld;
    const float* residual;
    long ldr;
    int act;  // 0 none, 1 relu, 2 lrelu
    float act_alpha, act_gain;
    const float* mask_src;  // backward: multiply by d(act)/d(pre) reconstructed from the saved output
    long ldm;
    int mask_mode;  // 0 none, 1 relu (src > 0), 2 lrelu (src > 0 ? gain : gain*alpha)
    float out_scale;
    float p_drop;
    unsigned long long seed;
    const unsigned long long* seed_ptr;
    int accumulate;
    const float* row_scale;  // internal (weight gradients): per-output-row factor, linear like alpha (FrozenBN scale of the dy operand)
    float* a_rowsum;         // ta == 1 dense GEMM: a_rowsum[m] += sum_k A[k][m]
};

struct GemmParams {
    Operand A, B;
    int M, N, K;  // for zmode 1 these are recomputed per parity class
    float* C;
    long ldc;
    int zmode;   // 0: z = split-K slice; 1: z = parity class; 2: z = tap + ntaps * split-K slice
    int splitk;  // number of K slices (atomicAdd output when > 1)
    int pstep;   // parity step (= conv stride) for zmode 1
    int nsamp;
    int pix_per_sample;  // rows per sample in the *full* destination grid (epilogue + scale lookups)
    int ntaps;           // zmode 2
    long c_tap_stride;   // zmode 2: column offset per tap in C
    // weight gradients with per-sample operand scales (modulated convs): K slices aligned to samples, so the two scales
    // leave the k-loop and become one factor per partial tile:  C += acc * srow[samp][m] * scol[samp][n]
    int samp_pix, samp_q;             // pixels (k indices) per sample, slices per sample; samp_pix == 0: ordinary split-K
    const float* srow; long srow_ld;  // per-sample factor of output row m
    const float* scol; long scol_ld;  // per-sample factor of output column n
    int ep_vec;          // host-checked: every epilogue operand is float4-addressable -> LDS-staged row-major epilogue
    float* ws;           // split-K fix-up: per-(tile, slice) partial tiles; null = fp32 atomics into C
    int* ws_count;       //   per-tile arrival counters (zero between launches)
    long long* trace;    // development aid (tools/trace_tiles.py): 4 wall-clock stamps per block, or null
    int narrow;          // host: the 256x32 tile was chosen (lets one-k-tile-per-tap channel counts take the scalar-addressed loads)
    GemmEpilogue ep;
};

struct TapMap {
    int kh0, kw0, tstep, nty, ntx;
};

struct ZCtx {
    int M, K, kbeg, kend;
    int py, px, DH2, DW2;
    TapMap tm;
    int fkh, fkw;
    long c_off;
    int samp;   // sample a sample-aligned K slice belongs to
};

// Incremental decode of the reduction index: k = (tap', c) for conv operands, k = pixel (n, y, x) for
// weight-gradient operands.  Decoded once per thread with integer divisions, then advanced by BK per k-tile
// with carries only (the old per-tile divisions cost more VALU issue slots than the loads themselves).
struct KDec {
    int c, ty, tx;   // KC gathers / RC_WT: channel, tap row/col index (in the tap map)
    int n, y, x;     // RC_PIX / RC_CONVK: pixel
};

__device__ __forceinline__ void kdec_init_tap(KDec& d, const TapMap& tm, int k, int C) {
    int t = k / C; d.c = k - t * C;
    d.ty = t / tm.ntx; d.tx = t - d.ty * tm.ntx;
}
__device__ __forceinline__ void kdec_step_tap(KDec& d, const TapMap& tm, int step, int C) {
    d.c += step;
    while (d.c >= C) { d.c -= C; if (++d.tx == tm.ntx) { d.tx = 0; ++d.ty; } }
}
__device__ __forceinline__ void kdec_init_pix(KDec& d, int k, int DH, int DW) {
    int per = DH * DW;
    d.n = k / per; int rem = k - d.n * per;
    d.y = rem / DW; d.x = rem - d.y * DW;
}
__device__ __forceinline__ void kdec_step_pix(KDec& d, int step, int DH, int DW) {
    d.x += step;
    while (d.x >= DW) { d.x -= DW; if (++d.y == DH) { d.y = 0; ++d.n; } }
}

struct RowCtx {  // per-thread cached decode of a KC gather row (a destination pixel)
    long base;
    int y, x, samp, valid;
};

template <int MODE>
__device__ __forceinline__ RowCtx make_row(const Operand& o, const GemmParams& p, const ZCtx& z, int r, int R) {
    RowCtx rc;
    rc.valid = r < R;
    rc.base = 0; rc.y = 0; rc.x = 0; rc.samp = 0;
    if (MODE == OP_KC_CONV) {
        int per = o.DH * o.DW;
        int n = r / per; int rem = r - n * per;
        int y = rem / o.DW; int x = rem - y * o.DW;
        rc.samp = n; rc.base = (long)n * o.sn;
        rc.y = y * o.stride - o.pad; rc.x = x * o.stride - o.pad;
    } else if (MODE == OP_KC_CONVT) {
        int per = z.DH2 * z.DW2;
        int n = per > 0 ? r / per : 0; int rem = r - n * per;
        int y2 = z.DW2 > 0 ? rem / z.DW2 : 0; int x2 = rem - y2 * z.DW2;
        rc.samp = n; rc.base = (long)n * o.sn;
        rc.y = y2 * p.pstep + z.py + o.pad; rc.x = x2 * p.pstep + z.px + o.pad;
    } else {
        rc.base = (long)r * o.ld;
    }
    return rc;
}

// Load 4 consecutive k values (k .. k+3) of one row for a k-contiguous operand.
template <int MODE>
__device__ __forceinline__ float4 load_kc(const Operand& o, const ZCtx& z, const RowCtx& rc, int k, const KDec& d) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!rc.valid || k >= z.kend) return v;
    if (MODE == OP_KC_DENSE) {
        const float* s = o.p + rc.base + k;
        if (o.vec) return *reinterpret_cast<const float4*>(s);
        v.x = s[0];
        if (k + 1 < z.kend) v.y = s[1];
        if (k + 2 < z.kend) v.z = s[2];
        if (k + 3 < z.kend) v.w = s[3];
        return v;
    }
    if (MODE == OP_KC_WTAP) {
        int kh = z.tm.kh0 + z.tm.tstep * d.ty, kw = z.tm.kw0 + z.tm.tstep * d.tx;
        return *reinterpret_cast<const float4*>(o.p + rc.base + (long)(kh * o.KW + kw) * o.C + d.c);
    }
    // gather modes
    if (o.vec) {
        int kh = z.tm.kh0 + z.tm.tstep * d.ty, kw = z.tm.kw0 + z.tm.tstep * d.tx;
        int sy, sx; bool ok;
        if (MODE == OP_KC_CONV) {
            sy = rc.y + kh; sx = rc.x + kw;
            ok = (sy >= 0) & (sy < o.SH) & (sx >= 0) & (sx < o.SW);
        } else {
            int ny = rc.y - kh, nx = rc.x - kw;
            sy = ny / o.stride; sx = nx / o.stride;
            ok = (ny >= 0) & (nx >= 0) & (sy < o.SH) & (sx < o.SW);
        }
        if (!ok) return v;
        v = *reinterpret_cast<const float4*>(o.p + rc.base + (long)sy * o.sh + (long)sx * o.sw + d.c);
        if (o.scale) {
            float4 s = *reinterpret_cast<const float4*>(o.scale + (long)rc.samp * o.scale_ld + d.c);
            v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
        }
        return v;
    }
    float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int kk = k + j;
        if (kk >= z.kend) break;
        int t = kk / o.C; int c = kk - t * o.C;
        int ty = t / z.tm.ntx, tx = t - ty * z.tm.ntx;
        int kh = z.tm.kh0 + z.tm.tstep * ty, kw = z.tm.kw0 + z.tm.tstep * tx;
        int sy, sx; bool ok;
        if (MODE == OP_KC_CONV) {
            sy = rc.y + kh; sx = rc.x + kw;
            ok = (sy >= 0) & (sy < o.SH) & (sx >= 0) & (sx < o.SW);
        } else {
            int ny = rc.y - kh, nx = rc.x - kw;
            sy = ny / o.stride; sx = nx / o.stride;
            ok = (ny >= 0) & (nx >= 0) & (sy < o.SH) & (sx < o.SW);
        }
        if (ok) {
            float val = o.p[rc.base + (long)sy * o.sh + (long)sx * o.sw + (long)c * o.sc];
            if (o.scale) val *= o.scale[(long)rc.samp * o.scale_ld + c];
            e[j] = val;
        }
    }
    return make_float4(e[0], e[1], e[2], e[3]);
}

// Load 4 consecutive row values (r .. r+3) at reduction index k for a row-contiguous operand.
template <int MODE>
__device__ __forceinline__ float4 load_rc(const Operand& o, const GemmParams& p, const ZCtx& z, int k, int r, int R, const KDec& d) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= z.kend || r >= R) return v;
    if (MODE == OP_RC_DENSE) {
        const float* s = o.p + (long)k * o.ld + r;
        if (o.vec) return *reinterpret_cast<const float4*>(s);
        v.x = s[0];
        if (r + 1 < R) v.y = s[1];
        if (r + 2 < R) v.z = s[2];
        if (r + 3 < R) v.w = s[3];
        return v;
    }
    if (MODE == OP_RC_WT) {
        int kh = z.tm.kh0 + z.tm.tstep * d.ty, kw = z.tm.kw0 + z.tm.tstep * d.tx;
        const float* s = o.p + (long)d.c * o.ld + (long)(kh * o.KW + kw) * o.Cr + r;
        if (o.vec) return *reinterpret_cast<const float4*>(s);
        v.x = s[0];
        if (r + 1 < R) v.y = s[1];
        if (r + 2 < R) v.z = s[2];
        if (r + 3 < R) v.w = s[3];
        return v;
    }
    if (MODE == OP_RC_PIX) {
        int sy = d.y, sx = d.x;
        if (o.tapped) { sy = d.y * o.stride - o.pad + z.fkh; sx = d.x * o.stride - o.pad + z.fkw; }
        if ((sy < 0) | (sy >= o.SH) | (sx < 0) | (sx >= o.SW)) return v;
        const float* s = o.p + (long)d.n * o.sn + (long)sy * o.sh + (long)sx * o.sw + r;
        if (o.vec) {
            v = *reinterpret_cast<const float4*>(s);
        } else {
            v.x = s[0];
            if (r + 1 < R) v.y = s[1];
            if (r + 2 < R) v.z = s[2];
            if (r + 3 < R) v.w = s[3];
        }
        if (o.scale) {
            const float* sc = o.scale + (long)d.n * o.scale_ld + r;
            v.x *= sc[0];
            if (r + 1 < R) v.y *= sc[1];
            if (r + 2 < R) v.z *= sc[2];
            if (r + 3 < R) v.w *= sc[3];
        }
        return v;
    }
    // OP_RC_CONVK: k = destination pixel, r = (tap, c) natural order, scalar gather
    {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int rr = r + j;
            if (rr >= R) break;
            int t = rr / o.C; int c = rr - t * o.C;
            int kh = t / o.KW; int kw = t - kh * o.KW;
            int sy = d.y * o.stride - o.pad + kh, sx = d.x * o.stride - o.pad + kw;
            if ((sy >= 0) & (sy < o.SH) & (sx >= 0) & (sx < o.SW))
                e[j] = o.p[(long)d.n * o.sn + (long)sy * o.sh + (long)sx * o.sw + (long)c * o.sc];
        }
        return make_float4(e[0], e[1], e[2], e[3]);
    }
}

template <int BKT>
__device__ __forceinline__ ZCtx make_zctx(const GemmParams& p) {
    ZCtx z;
    z.M = p.M; z.K = p.K; z.py = 0; z.px = 0; z.DH2 = p.A.DH; z.DW2 = p.A.DW;
    z.tm.kh0 = 0; z.tm.kw0 = 0; z.tm.tstep = 1; z.tm.nty = p.A.KH; z.tm.ntx = p.A.KW > 0 ? p.A.KW : 1;
    z.fkh = 0; z.fkw = 0; z.c_off = 0;
    int ks = blockIdx.z;
    if (p.zmode == 1) {
        int s = p.pstep;
        int ncls = s * s;
        int cls = ncls - 1 - blockIdx.z % ncls;   // heaviest parity class (most taps) first: blocks are dispatched in z order and the light classes fill the tail
        ks = blockIdx.z / ncls;
        z.py = cls / s; z.px = cls - z.py * s;
        z.DH2 = (p.A.DH - z.py + s - 1) / s; z.DW2 = (p.A.DW - z.px + s - 1) / s;
        if (z.DH2 < 0) z.DH2 = 0;
        if (z.DW2 < 0) z.DW2 = 0;
        z.M = p.nsamp * z.DH2 * z.DW2;
        z.tm.tstep = s;
        z.tm.kh0 = (z.py + p.A.pad) % s; z.tm.kw0 = (z.px + p.A.pad) % s;
        z.tm.nty = z.tm.kh0 < p.A.KH ? (p.A.KH - z.tm.kh0 + s - 1) / s : 0;
        z.tm.ntx = z.tm.kw0 < p.A.KW ? (p.A.KW - z.tm.kw0 + s - 1) / s : 0;
        z.K = z.tm.nty * z.tm.ntx * p.A.C;
        if (z.tm.ntx == 0) z.tm.ntx = 1;
    } else if (p.zmode == 2) {
        int tap = blockIdx.z % p.ntaps;
        ks = blockIdx.z / p.ntaps;
        int KWt = p.A.tapped ? p.A.KW : p.B.KW;
        z.fkh = tap / KWt; z.fkw = tap - z.fkh * KWt;
        z.c_off = (long)tap * p.c_tap_stride;
    }
    z.samp = 0;
    if (p.samp_pix > 0) {
        const int n = ks / p.samp_q, j = ks - n * p.samp_q;
        const int len = ((p.samp_pix + p.samp_q - 1) / p.samp_q + BKT - 1) / BKT * BKT;
        z.samp = n;
        z.kbeg = n * p.samp_pix + j * len;
        z.kend = min(n * p.samp_pix + (j + 1) * len, (n + 1) * p.samp_pix);
        if (z.kend < z.kbeg) z.kend = z.kbeg;
    } else if (p.splitk > 1) {
        int ktiles = (z.K + BKT - 1) / BKT;
        int per = (ktiles + p.splitk - 1) / p.splitk;
        z.kbeg = ks * per * BKT;
        z.kend = min(z.K, (ks + 1) * per * BKT);
    } else {
        z.kbeg = 0; z.kend = z.K;
    }
    return z;
}

// One output element through the epilogue chain documented in include/ldetr_hip.h.
// cs / cb: the column's scale and bias, fetched once per column by the caller (not once per element).
__device__ __forceinline__ float apply_epilogue(const GemmEpilogue& ep, float v, long orow, int n, int samp, long ldc, float inv_keep,
                                                float cs, float cb, float res, float mval) {
    v *= ep.alpha;
    if (ep.row_scale) v *= ep.row_scale[orow];
    v *= cs;
    if (ep.samp_scale) v *= ep.samp_scale[(long)samp * ep.samp_ld + n];
    v += cb;
    v += res;
    if (ep.act == 1) v = fmaxf(v, 0.f);
    else if (ep.act == 2) v = (v > 0.f ? v : v * ep.act_alpha) * ep.act_gain;
    if (ep.mask_mode) {   // mval = mask_src[orow][n], fetched by the caller (float4 in the row-major epilogue)
        if (ep.mask_mode == 1) v = mval > 0.f ? v : 0.f;
        else v *= (mval > 0.f ? ep.act_gain : ep.act_gain * ep.act_alpha);
    }
    if (ep.p_drop > 0.f) v *= drop_scale(ep.seed + (ep.seed_ptr ? *ep.seed_ptr : 0ull), (uint64_t)(orow * ldc + n), ep.p_drop, inv_keep);
    return v * ep.out_scale;
}

// NWV waves per block: 4 (2x2 wave grid) or 8 (2x4: same tile, half the accumulators per wave, twice the waves per SIMD to
// cover each other's barrier / staging phases).
// SPLIT: the contraction runs on the bf16 matrix pipe with fp32-equivalent results.  Every operand element is written to LDS as
// three bf16 values x = hi + mid + lo (8 + 8 + 8 significant bits: together the whole fp32 significand, the split is exact) and
// each k16 slab issues the six products of weight >= 2^-18 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) into the same fp32
// accumulators.  bf16 x bf16 products are exact in fp32, the dropped terms are <= 2^-26 relative: the result is at least as close
// to the exact contraction as the f32 MFMA path (tools/proto_bf16x6: rms error 1.6e-7 vs 2.0e-7 against fp64 at K = 4096), while
// the matrix pipe spends 6 x 8 passes (v_mfma_f32_32x32x16_bf16) where the f32 path spends 8 x 16 (v_mfma_f32_32x32x2_f32).
// Same 32x32 accumulator layout, so loaders, split-K fix-up and epilogues are shared with the f32 path.  Non-finite inputs
// (Inf splits into Inf + NaN; so does |x| within one bf16 ulp of FLT_MAX) are handled after the loop: a tile with a non-finite
// accumulator is recomputed on the f32 pipe inside the same launch, so such launches return what the f32 instantiation returns.
// (pk_bf16 / split2_bf16 / bf16x8_t: ldetr_common.hpp -- shared with conv_c32.hip)
#ifndef LDETR_KC_PAD
#define LDETR_KC_PAD 64   // bytes between the (part, k-block) planes of a k-contiguous operand's LDS image (see gemm_f32_kernel)
#endif

template <int BM, int BN, int BKT, int AMODE, int BMODE, int NWV, bool FAST, bool SPLIT = false>
__global__ __launch_bounds__(NWV * 64) void gemm_f32_kernel(GemmParams p) {
    constexpr int NT = NWV * 64;
    constexpr int WGN = BN == 32 ? 1 : ((NWV == 8 && BN >= 128) ? 4 : 2), WGM = NWV / WGN;   // BN == 32: all waves along M (narrow-N tile)
    constexpr bool A_KC = AMODE <= OP_KC_WTAP;
    constexpr bool B_KC = BMODE <= OP_KC_WTAP;
    constexpr int LDA = BM + 2, LDB = BN + 2;
    constexpr int QK = BKT / 4;                                  // float4 quads along k per row
    constexpr int NUA = BM * BKT / 4 / NT, NUB = BN * BKT / 4 / NT;  // float4 units per thread per k-tile
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    static_assert(NUA >= 1 && NUB >= 1 && TM >= 1 && TN >= 1, "tile too small for this many waves");
    // dynamic LDS (the 128x128xBK32 image is 65 KiB: above the 64 KiB static limit, within the CU's 160 KiB)
    extern __shared__ __attribute__((aligned(16))) float ldetr_smem[];
    float (*As)[BKT][LDA] = reinterpret_cast<float (*)[BKT][LDA]>(ldetr_smem);
    float (*Bs)[BKT][LDB] = reinterpret_cast<float (*)[BKT][LDB]>(ldetr_smem + 2 * BKT * LDA);
    // SPLIT: one buffer of [part][k-block of 8][slot][8 bf16]: a lane's MFMA operand (8 consecutive k of one row) is one 16-byte
    // read.  k-contiguous operands keep slot = row (the 32 lanes of a k-block read 512 contiguous bytes; writes are 8 bytes per
    // (row, half k-block), and 64 bytes of padding per plane keep the four k-blocks a wave writes for one row off each other's
    // banks).  Row-contiguous operands (a lane loads 4 adjacent rows of one k) use slot = (row % 4) * rows/4 + (row / 4 + rotation(row % 4))
    // mod rows/4: the lanes of a store hit consecutive slots and a pass of an operand read hits every bank group once; with
    // slot = row those stores were 16-way bank conflicts.
    char* const sbase = reinterpret_cast<char*>(ldetr_smem);
    constexpr int PLA = A_KC ? BM * 16 + LDETR_KC_PAD : BM * 16, PLB = B_KC ? BN * 16 + LDETR_KC_PAD : BN * 16;   // bytes per (part, k-block) plane
    constexpr int SB_OFF = 3 * (BKT / 8) * PLA;
    // (rotation 0 / 8 / 4 / 12 slots for rows 4q, 4q+1, 4q+2, 4q+3: the 16 lanes of one pass of an operand read then hit 16 different
    //  16-byte bank groups; with 0 / 8 / 0 / 8 they hit 8, two lanes each)
    auto slotA = [](int row) { return A_KC ? row : ((row & 3) * (BM / 4) + (((row >> 2) + 8 * (row & 1) + 4 * ((row >> 1) & 1)) & (BM / 4 - 1))); };
    auto slotB = [](int row) { return B_KC ? row : ((row & 3) * (BN / 4) + (((row >> 2) + 8 * (row & 1) + 4 * ((row >> 1) & 1)) & (BN / 4 - 1))); };
    auto adrA = [&](int part, int kb, int row, int koff) { return sbase + (part * (BKT / 8) + kb) * PLA + slotA(row) * 16 + koff * 2; };
    auto adrB = [&](int part, int kb, int row, int koff) { return sbase + SB_OFF + (part * (BKT / 8) + kb) * PLB + slotB(row) * 16 + koff * 2; };
    // row-contiguous operands under SPLIT: a thread's units are grouped G consecutive k for the same four rows, so the bf16 values
    // it writes to one LDS row are adjacent (one 8 / 4 / 2-byte store per row and part instead of G 2-byte stores)
    constexpr int GA = (SPLIT && !A_KC) ? (NUA % 4 == 0 ? 4 : (NUA % 2 == 0 ? 2 : 1)) : 1;
    constexpr int GB = (SPLIT && !B_KC) ? (NUB % 4 == 0 ? 4 : (NUB % 2 == 0 ? 2 : 1)) : 1;

#ifndef LDETR_TILE_TRACE
#define LDETR_TILE_TRACE 1
#endif
    const long long tr0 = (LDETR_TILE_TRACE && p.trace) ? wall_clock64() : 0;
    const ZCtx z = make_zctx<BKT>(p);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= z.M) return;  // uniform per block (parity classes may be smaller than the launch grid)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // unit i of thread t -> (first row, first k) of the float4 it stages (shared by the assignment below and the split path's f32 redo)
    auto unitA = [](int t, int i, int& r, int& k) {
        if constexpr (A_KC) { const int u = t + i * NT; r = u / QK; k = (u - r * QK) << 2; }
        else {
            // SPLIT: adjacent lanes take the two k-groups that share a 16-byte LDS slot (its low / high 8 bytes), so the 32 lanes of a
            // store pass fill 256 contiguous bytes; lanes 2q and 2q+1 then load from two rows of the operand (two coalesced runs)
            const int slot = t + (i / GA) * NT;
            const int kg = SPLIT ? 2 * ((slot >> 1) / (BM / 4)) + (slot & 1) : slot / (BM / 4);
            const int rq = SPLIT ? (slot >> 1) % (BM / 4) : slot - kg * (BM / 4);
            k = kg * GA + (i % GA); r = rq << 2;
        }
    };
    auto unitB = [](int t, int i, int& r, int& k) {
        if constexpr (B_KC) { const int u = t + i * NT; r = u / QK; k = (u - r * QK) << 2; }
        else {
            const int slot = t + (i / GB) * NT;
            const int kg = SPLIT ? 2 * ((slot >> 1) / (BN / 4)) + (slot & 1) : slot / (BN / 4);
            const int rq = SPLIT ? (slot >> 1) % (BN / 4) : slot - kg * (BN / 4);
            k = kg * GB + (i % GB); r = rq << 2;
        }
    };
    // Per-thread unit assignment + incremental k decode state.
    int a_r[NUA], a_k[NUA], b_r[NUB], b_k[NUB];
    RowCtx a_rc[NUA], b_rc[NUB];
    KDec a_d[NUA], b_d[NUB];
#pragma unroll
    for (int i = 0; i < NUA; i++) {
        a_d[i].c = a_d[i].ty = a_d[i].tx = a_d[i].n = a_d[i].y = a_d[i].x = 0;
        unitA(tid, i, a_r[i], a_k[i]);
        if constexpr (A_KC) {
            a_rc[i] = make_row<AMODE>(p.A, p, z, m0 + a_r[i], z.M);
            if constexpr (AMODE != OP_KC_DENSE) kdec_init_tap(a_d[i], z.tm, z.kbeg + a_k[i], p.A.C);
        } else {
            if constexpr (AMODE == OP_RC_WT) kdec_init_tap(a_d[i], z.tm, z.kbeg + a_k[i], p.A.C);
            if constexpr (AMODE == OP_RC_PIX || AMODE == OP_RC_CONVK) kdec_init_pix(a_d[i], z.kbeg + a_k[i], p.A.DH, p.A.DW);
        }
    }
#pragma unroll
    for (int i = 0; i < NUB; i++) {
        b_d[i].c = b_d[i].ty = b_d[i].tx = b_d[i].n = b_d[i].y = b_d[i].x = 0;
        unitB(tid, i, b_r[i], b_k[i]);
        if constexpr (B_KC) {
            b_rc[i] = make_row<BMODE>(p.B, p, z, n0 + b_r[i], p.N);
            if constexpr (BMODE != OP_KC_DENSE) kdec_init_tap(b_d[i], z.tm, z.kbeg + b_k[i], p.B.C);
        } else {
            if constexpr (BMODE == OP_RC_WT) kdec_init_tap(b_d[i], z.tm, z.kbeg + b_k[i], p.B.C);
            if constexpr (BMODE == OP_RC_PIX || BMODE == OP_RC_CONVK) kdec_init_pix(b_d[i], z.kbeg + b_k[i], p.B.DH, p.B.DW);
        }
    }

    // Scalar-addressed operand loads.  PMC on the ResNet 3x3 shapes: ~100 VALU instructions per wave per k-tile went into the
    // gather addressing (tap decode, bounds tests, 64-bit address arithmetic per float4 unit) next to 16 MFMAs, and VALU issue
    // does not overlap the matrix pipe on this part: the loop ran at 64 % MFMA utilisation, 93 % with the loads removed.  When the
    // channel count is a multiple of the k-tile, a whole k-tile lies inside ONE filter tap, so tap and channel offset are
    // block-uniform: they live in SGPRs and go into the buffer load's scalar offset; each unit keeps a byte offset of its pixel
    // (computed once) and a bit mask of the taps that fall inside the image.  Per unit and k-tile: one bit test, one select, one
    // buffer_load_dwordx4 (an out-of-range vector offset returns zeros: that is the padding).
    // FAST is chosen by the host (fast_operands_ok in launch_gemm checks every condition below); the generic instantiation carries
    // none of this code (as runtime branches it slowed the generic path of the 32-channel 256^2 layer from 420 to 510 us).
    constexpr bool A_FAST = FAST && (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT || AMODE == OP_KC_DENSE || AMODE == OP_RC_DENSE || AMODE == OP_RC_PIX);
    constexpr bool B_FAST = FAST && (BMODE == OP_KC_DENSE || BMODE == OP_RC_WT || BMODE == OP_RC_DENSE || BMODE == OP_RC_PIX || BMODE == OP_KC_WTAP);
    constexpr bool TAP_STATE = FAST && (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT || BMODE == OP_RC_WT || BMODE == OP_KC_WTAP);
    constexpr bool fastA = A_FAST, fastB = B_FAST;
    int a_voff[NUA]; unsigned a_msk[NUA]; int b_voff[NUB];
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p), 0, 0, 0x00020000), rsB = rsA;
    int f_c0 = 0, f_tx = 0, f_ty = 0, f_tap = 0;   // tap / channel of the next k-tile to load (block-uniform)
    const int f_C = A_KC ? p.A.C : p.B.C;           // channels per tap of the reduction index
    constexpr bool k_tiles_in_taps = TAP_STATE;
    if constexpr (FAST && AMODE == OP_KC_CONV) {
        const long padoff = (long)p.A.pad * p.A.sh + (long)p.A.pad * p.A.sw;
        if (fastA) {
            rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p) - padoff, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < NUA; i++) {
                const RowCtx& rc = a_rc[i];
                a_voff[i] = (int)((rc.base + (long)(rc.y + p.A.pad) * p.A.sh + (long)(rc.x + p.A.pad) * p.A.sw + a_k[i]) * 4);
                // taps inside the image: ty in [ylo, yhi), tx in [xlo, xhi) -- closed form, then one OR per valid tap row
                unsigned m = 0;
                if (rc.valid) {
                    const int ylo = max(0, -rc.y), yhi = min(z.tm.nty, p.A.SH - rc.y);
                    const int xlo = max(0, -rc.x), xhi = min(z.tm.ntx, p.A.SW - rc.x);
                    if (xhi > xlo) {
                        const unsigned xm = ((1u << xhi) - 1u) & ~((1u << xlo) - 1u);
                        for (int ty = ylo; ty < yhi; ty++) m |= xm << (ty * z.tm.ntx);
                    }
                }
                a_msk[i] = m;
            }
        }
    }
    if constexpr (FAST && AMODE == OP_KC_CONVT) {
        // data gradient of a strided conv, one parity class per blockIdx.z: source pixel of tap (ty, tx) is (sy0 - ty, sx0 - tx) with
        // sy0 = (y - kh0) / stride exact; the descriptor base is moved back by the largest tap offset so the scalar offset
        // ((nty-1-ty) sh + (ntx-1-tx) sw + c0) stays non-negative
        const long shift = (long)(z.tm.nty - 1) * p.A.sh + (long)(z.tm.ntx - 1) * p.A.sw;
        if (fastA) {
            rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p) - shift, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < NUA; i++) {
                const RowCtx& rc = a_rc[i];
                unsigned m = 0; int vo = 0;
                if (rc.valid) {
                    const int sy0 = (rc.y - z.tm.kh0) / p.A.stride, sx0 = (rc.x - z.tm.kw0) / p.A.stride;
                    vo = (int)((rc.base + (long)sy0 * p.A.sh + (long)sx0 * p.A.sw + a_k[i]) * 4);
                    const int ylo = max(0, sy0 - p.A.SH + 1), yhi = min(z.tm.nty, sy0 + 1);
                    const int xlo = max(0, sx0 - p.A.SW + 1), xhi = min(z.tm.ntx, sx0 + 1);
                    if (xhi > xlo) {
                        const unsigned xm = ((1u << xhi) - 1u) & ~((1u << xlo) - 1u);
                        for (int ty = ylo; ty < yhi; ty++) m |= xm << (ty * z.tm.ntx);
                    }
                }
                a_voff[i] = vo; a_msk[i] = m;
            }
        }
    }
    if constexpr (FAST && AMODE == OP_KC_DENSE) {
        if (fastA) {
            rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < NUA; i++) a_voff[i] = a_rc[i].valid ? (int)((a_rc[i].base + a_k[i]) * 4) : (int)0x80000000;
        }
    }
    if constexpr (FAST && AMODE == OP_RC_DENSE) {   // A[k][m], m contiguous: row k0 + a_k of the tile, columns m0 + a_r .. + 3
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NUA; i++) a_voff[i] = (m0 + a_r[i] < z.M) ? (int)(((long)a_k[i] * p.A.ld + m0 + a_r[i]) * 4) : (int)0x80000000;
    }
    if constexpr (FAST && BMODE == OP_RC_DENSE) {
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B.p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NUB; i++) b_voff[i] = (n0 + b_r[i] < p.N) ? (int)(((long)b_k[i] * p.B.ld + n0 + b_r[i]) * 4) : (int)0x80000000;
    }
    // B[k = pixel][r = channel] (the input operand of a weight gradient).  With the row width a multiple (or a divisor) of the
    // k-tile, the 32 pixels of a k-tile start at a block-uniform (sample, row, column): those live in SGPRs, each unit keeps its
    // fixed offset inside the tile's pixel patch; per unit and k-tile: two adds, two range tests, one select, one buffer load.
    int g_n = 0, g_y0 = 0, g_x0 = 0;          // first pixel of the next k-tile
    int b_sy[NUB], b_sx[NUB];
    const int pix_st = p.B.tapped ? p.B.stride : 1, pix_pad = p.B.tapped ? p.B.pad : 0;
    const int pix_kh = p.B.tapped ? z.fkh : 0, pix_kw = p.B.tapped ? z.fkw : 0;
    constexpr bool PIX_STATE = FAST && (AMODE == OP_RC_PIX || BMODE == OP_RC_PIX);
    const int pixDH = AMODE == OP_RC_PIX ? p.A.DH : p.B.DH, pixDW = AMODE == OP_RC_PIX ? p.A.DW : p.B.DW;   // the pixel grid k runs over
    int a_sy[NUA], a_sx[NUA];
    const int apix_st = p.A.tapped ? p.A.stride : 1, apix_pad = p.A.tapped ? p.A.pad : 0;
    const int apix_kh = p.A.tapped ? z.fkh : 0, apix_kw = p.A.tapped ? z.fkw : 0;
    if constexpr (FAST && AMODE == OP_RC_PIX) {   // the same view as the A operand (transposed-conv weight gradient)
        const long padoff = (long)apix_pad * p.A.sh + (long)apix_pad * p.A.sw;
        rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p) - padoff, 0, 0x7fffffff, 0x00020000);
        const int wrow = pixDW >= BKT ? BKT : pixDW;
#pragma unroll
        for (int i = 0; i < NUA; i++) {
            const int uy = a_k[i] / wrow, ux = a_k[i] - uy * wrow;
            a_sy[i] = uy * apix_st - apix_pad + apix_kh; a_sx[i] = ux * apix_st - apix_pad + apix_kw;
            a_voff[i] = (m0 + a_r[i] < z.M) ? (int)(((long)uy * apix_st * p.A.sh + (long)ux * apix_st * p.A.sw + m0 + a_r[i]) * 4) : (int)0x80000000;
        }
    }
    auto reset_pix = [&]() {
        if constexpr (PIX_STATE) {
            const int per = pixDH * pixDW;
            g_n = z.kbeg / per; const int rem = z.kbeg - g_n * per;
            g_y0 = rem / pixDW; g_x0 = rem - g_y0 * pixDW;
        }
    };
    reset_pix();
    if constexpr (FAST && BMODE == OP_RC_PIX) {
        const long padoff = (long)pix_pad * p.B.sh + (long)pix_pad * p.B.sw;
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B.p) - padoff, 0, 0x7fffffff, 0x00020000);
        const int wrow = p.B.DW >= BKT ? BKT : p.B.DW;     // pixels of the tile per image row
#pragma unroll
        for (int i = 0; i < NUB; i++) {
            const int uy = b_k[i] / wrow, ux = b_k[i] - uy * wrow;
            b_sy[i] = uy * pix_st - pix_pad + pix_kh; b_sx[i] = ux * pix_st - pix_pad + pix_kw;
            b_voff[i] = (n0 + b_r[i] < p.N) ? (int)(((long)uy * pix_st * p.B.sh + (long)ux * pix_st * p.B.sw + n0 + b_r[i]) * 4) : (int)0x80000000;
        }
    }
    if constexpr (FAST && BMODE == OP_KC_WTAP) {   // weights [n][tap][c]: row base + (tap * C + c0) scalar
        rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B.p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < NUB; i++) b_voff[i] = b_rc[i].valid ? (int)((b_rc[i].base + b_k[i]) * 4) : (int)0x80000000;
    }
    if constexpr (FAST && BMODE == OP_KC_DENSE) {
        if (fastB) {
            rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B.p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < NUB; i++) b_voff[i] = b_rc[i].valid ? (int)((b_rc[i].base + b_k[i]) * 4) : (int)0x80000000;
        }
    }
    if constexpr (FAST && BMODE == OP_RC_WT) {
        // weights read row-contiguous: element (k = (tap, c), r) at c * ld + tap * Cr + r; tap and c0 are block-uniform per k-tile
        if (fastB) {
            rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B.p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int i = 0; i < NUB; i++) b_voff[i] = (n0 + b_r[i] < p.N) ? (int)(((long)b_k[i] * p.B.ld + n0 + b_r[i]) * 4) : (int)0x80000000;
        }
    }
    auto reset_kstate = [&]() {   // block-uniform position of the first k-tile (the loaders advance it tile by tile)
        reset_pix();
        if constexpr (TAP_STATE) {
            if (k_tiles_in_taps) {
                f_tap = z.kbeg / f_C; f_c0 = z.kbeg - f_tap * f_C;
                f_ty = f_tap / z.tm.ntx; f_tx = f_tap - f_ty * z.tm.ntx;
            }
        }
    };
    reset_kstate();
    auto as_float4 = [](auto v) { return make_float4(__int_as_float(v[0]), __int_as_float(v[1]), __int_as_float(v[2]), __int_as_float(v[3])); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // register staging of the next k-tile (global loads run one k-tile ahead of the MFMAs; two ahead measured slower)
    float4 ra0[NUA], rb0[NUB];
    // Per-sample operand scale of the tap-addressed conv operands (style modulation of x / demodulation of dy): the factors are
    // fetched next to the data but applied when the tile is written to LDS.  Multiplying right after the load made every k-tile
    // wait for its own global loads before the MFMAs of the tile in flight could start.
    constexpr bool A_DEFER = A_FAST && (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT);
    constexpr int NSA = A_DEFER ? NUA : 1;
    float4 sa0[NSA];
    const bool a_defer = A_DEFER && fastA && p.A.scale != nullptr;
    // loads the k-tile starting at k0 and advances the decode state to the following tile
    auto gload = [&](int k0, float4 (&ra)[NUA], float4 (&rb)[NUB], float4 (&sa)[NSA]) {
#pragma unroll
        for (int i = 0; i < NUA; i++) {
            if constexpr (A_FAST) {
                if (fastA) {
                    if constexpr (AMODE == OP_KC_DENSE) {
                        ra[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsA, a_voff[i], k0 * 4, 0));
                    } else if constexpr (AMODE == OP_RC_PIX) {
                        const int sy = a_sy[i] + g_y0 * apix_st, sx = a_sx[i] + g_x0 * apix_st;
                        const bool ok = ((unsigned)sy < (unsigned)p.A.SH) & ((unsigned)sx < (unsigned)p.A.SW);
                        const int soff = (int)(((long)g_n * p.A.sn + (long)(g_y0 * apix_st + apix_kh) * p.A.sh + (long)(g_x0 * apix_st + apix_kw) * p.A.sw) * 4);
                        ra[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? a_voff[i] : (int)0x80000000, soff, 0));
                        if (p.A.scale && ok && m0 + a_r[i] < z.M) {
                            const float4 sc = *reinterpret_cast<const float4*>(p.A.scale + (long)g_n * p.A.scale_ld + m0 + a_r[i]);
                            ra[i].x *= sc.x; ra[i].y *= sc.y; ra[i].z *= sc.z; ra[i].w *= sc.w;
                        }
                    } else if constexpr (AMODE == OP_RC_DENSE) {   // rows of the tile past the end of the reduction read as zeros
                        ra[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsA, a_k[i] < z.kend - k0 ? a_voff[i] : (int)0x80000000, k0 * (int)p.A.ld * 4, 0));
                    } else {
                        const int soff = AMODE == OP_KC_CONV ? (f_ty * (int)p.A.sh + f_tx * (int)p.A.sw + f_c0) * 4
                                                             : ((z.tm.nty - 1 - f_ty) * (int)p.A.sh + (z.tm.ntx - 1 - f_tx) * (int)p.A.sw + f_c0) * 4;
                        const bool tap_ok = (a_msk[i] >> f_tap) & 1u;
                        const int vo = tap_ok ? a_voff[i] : (int)0x80000000;
                        ra[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsA, vo, soff, 0));
                        if constexpr (A_DEFER) {
                            if (p.A.scale)   // (rows past the end carry no sample index: they read the table's first entry and hold zeros anyway)
                                sa[i] = *reinterpret_cast<const float4*>(tap_ok ? p.A.scale + (long)a_rc[i].samp * p.A.scale_ld + f_c0 + a_k[i] : p.A.scale);
                        }
                    }
                    continue;
                }
            }
            if constexpr (A_KC) {
                ra[i] = load_kc<AMODE>(p.A, z, a_rc[i], k0 + a_k[i], a_d[i]);
                if constexpr (AMODE != OP_KC_DENSE) kdec_step_tap(a_d[i], z.tm, BKT, p.A.C);
            } else {
                ra[i] = load_rc<AMODE>(p.A, p, z, k0 + a_k[i], m0 + a_r[i], z.M, a_d[i]);
                if constexpr (AMODE == OP_RC_WT) kdec_step_tap(a_d[i], z.tm, BKT, p.A.C);
                if constexpr (AMODE == OP_RC_PIX || AMODE == OP_RC_CONVK) kdec_step_pix(a_d[i], BKT, p.A.DH, p.A.DW);
            }
        }
#pragma unroll
        for (int i = 0; i < NUB; i++) {
            if constexpr (B_FAST) {
                if (fastB) {
                    if constexpr (BMODE == OP_KC_DENSE) {
                        rb[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsB, b_voff[i], k0 * 4, 0));
                    } else if constexpr (BMODE == OP_KC_WTAP) {
                        const int kh = z.tm.kh0 + z.tm.tstep * f_ty, kw = z.tm.kw0 + z.tm.tstep * f_tx;
                        rb[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsB, b_voff[i], ((kh * p.B.KW + kw) * p.B.C + f_c0) * 4, 0));
                    } else if constexpr (BMODE == OP_RC_DENSE) {
                        rb[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsB, b_k[i] < z.kend - k0 ? b_voff[i] : (int)0x80000000, k0 * (int)p.B.ld * 4, 0));
                    } else if constexpr (BMODE == OP_RC_PIX) {
                        const int sy = b_sy[i] + g_y0 * pix_st, sx = b_sx[i] + g_x0 * pix_st;
                        const bool ok = ((unsigned)sy < (unsigned)p.B.SH) & ((unsigned)sx < (unsigned)p.B.SW);
                        const int soff = (int)(((long)g_n * p.B.sn + (long)(g_y0 * pix_st + pix_kh) * p.B.sh + (long)(g_x0 * pix_st + pix_kw) * p.B.sw) * 4);
                        rb[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsB, ok ? b_voff[i] : (int)0x80000000, soff, 0));
                        if (p.B.scale && ok && n0 + b_r[i] < p.N) {
                            const float4 sc = *reinterpret_cast<const float4*>(p.B.scale + (long)g_n * p.B.scale_ld + n0 + b_r[i]);
                            rb[i].x *= sc.x; rb[i].y *= sc.y; rb[i].z *= sc.z; rb[i].w *= sc.w;
                        }
                    } else {
                        const int kh = z.tm.kh0 + z.tm.tstep * f_ty, kw = z.tm.kw0 + z.tm.tstep * f_tx;
                        rb[i] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rsB, b_voff[i], (f_c0 * (int)p.B.ld + (kh * p.B.KW + kw) * p.B.Cr) * 4, 0));
                    }
                    continue;
                }
            }
            if constexpr (B_KC) {
                rb[i] = load_kc<BMODE>(p.B, z, b_rc[i], k0 + b_k[i], b_d[i]);
                if constexpr (BMODE != OP_KC_DENSE) kdec_step_tap(b_d[i], z.tm, BKT, p.B.C);
            } else {
                rb[i] = load_rc<BMODE>(p.B, p, z, k0 + b_k[i], n0 + b_r[i], p.N, b_d[i]);
                if constexpr (BMODE == OP_RC_WT) kdec_step_tap(b_d[i], z.tm, BKT, p.B.C);
                if constexpr (BMODE == OP_RC_PIX || BMODE == OP_RC_CONVK) kdec_step_pix(b_d[i], BKT, p.B.DH, p.B.DW);
            }
        }
        if constexpr (PIX_STATE) {   // next k-tile: 32 pixels further along the (sample, row, column) order
            if (pixDW >= BKT) { g_x0 += BKT; if (g_x0 >= pixDW) { g_x0 = 0; g_y0++; } }
            else g_y0 += BKT / pixDW;
            if (g_y0 >= pixDH) { g_y0 = 0; g_n++; }
        }
        if constexpr (TAP_STATE) {
            if (k_tiles_in_taps) {   // next k-tile: one k-tile of channels further, or the next tap
                f_c0 += BKT;
                if (f_c0 >= f_C) { f_c0 = 0; f_tap++; f_tx++; if (f_tx == z.tm.ntx) { f_tx = 0; f_ty++; } }
            }
        }
    };
    auto comp = [](const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); };
    auto lstore = [&](auto split_tag, int buf, float4 (&ra)[NUA], const float4 (&rb)[NUB], const float4 (&sa)[NSA]) {
        constexpr bool SPL = decltype(split_tag)::value;   // the bf16 image, or (SPLIT kernels' non-finite fallback) the f32 one
        if constexpr (A_DEFER) {
            if (a_defer) {
#pragma unroll
                for (int i = 0; i < NUA; i++) { ra[i].x *= sa[i].x; ra[i].y *= sa[i].y; ra[i].z *= sa[i].z; ra[i].w *= sa[i].w; }
            }
        }
        if constexpr (SPL) {
            unsigned h0, m0_, l0, h1, m1, l1;
            if constexpr (A_KC) {
#pragma unroll
                for (int i = 0; i < NUA; i++) {
                    split2_bf16(ra[i].x, ra[i].y, h0, m0_, l0); split2_bf16(ra[i].z, ra[i].w, h1, m1, l1);
                    *reinterpret_cast<uint2*>(adrA(0, a_k[i] >> 3, a_r[i], a_k[i] & 7)) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(adrA(1, a_k[i] >> 3, a_r[i], a_k[i] & 7)) = make_uint2(m0_, m1);
                    *reinterpret_cast<uint2*>(adrA(2, a_k[i] >> 3, a_r[i], a_k[i] & 7)) = make_uint2(l0, l1);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NUA; i += GA)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        unsigned short* dh = reinterpret_cast<unsigned short*>(adrA(0, a_k[i] >> 3, a_r[i] + e, a_k[i] & 7));
                        unsigned short* dm = reinterpret_cast<unsigned short*>(adrA(1, a_k[i] >> 3, a_r[i] + e, a_k[i] & 7));
                        unsigned short* dl = reinterpret_cast<unsigned short*>(adrA(2, a_k[i] >> 3, a_r[i] + e, a_k[i] & 7));
                        if constexpr (GA == 4) {
                            split2_bf16(comp(ra[i], e), comp(ra[i + 1], e), h0, m0_, l0); split2_bf16(comp(ra[i + 2], e), comp(ra[i + 3], e), h1, m1, l1);
                            *reinterpret_cast<uint2*>(dh) = make_uint2(h0, h1); *reinterpret_cast<uint2*>(dm) = make_uint2(m0_, m1); *reinterpret_cast<uint2*>(dl) = make_uint2(l0, l1);
                        } else if constexpr (GA == 2) {
                            split2_bf16(comp(ra[i], e), comp(ra[i + 1], e), h0, m0_, l0);
                            *reinterpret_cast<unsigned*>(dh) = h0; *reinterpret_cast<unsigned*>(dm) = m0_; *reinterpret_cast<unsigned*>(dl) = l0;
                        } else {
                            split2_bf16(comp(ra[i], e), 0.f, h0, m0_, l0);
                            *dh = (unsigned short)h0; *dm = (unsigned short)m0_; *dl = (unsigned short)l0;
                        }
                    }
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int i = 0; i < NUB; i++) {
                    split2_bf16(rb[i].x, rb[i].y, h0, m0_, l0); split2_bf16(rb[i].z, rb[i].w, h1, m1, l1);
                    *reinterpret_cast<uint2*>(adrB(0, b_k[i] >> 3, b_r[i], b_k[i] & 7)) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(adrB(1, b_k[i] >> 3, b_r[i], b_k[i] & 7)) = make_uint2(m0_, m1);
                    *reinterpret_cast<uint2*>(adrB(2, b_k[i] >> 3, b_r[i], b_k[i] & 7)) = make_uint2(l0, l1);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NUB; i += GB)
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        unsigned short* dh = reinterpret_cast<unsigned short*>(adrB(0, b_k[i] >> 3, b_r[i] + e, b_k[i] & 7));
                        unsigned short* dm = reinterpret_cast<unsigned short*>(adrB(1, b_k[i] >> 3, b_r[i] + e, b_k[i] & 7));
                        unsigned short* dl = reinterpret_cast<unsigned short*>(adrB(2, b_k[i] >> 3, b_r[i] + e, b_k[i] & 7));
                        if constexpr (GB == 4) {
                            split2_bf16(comp(rb[i], e), comp(rb[i + 1], e), h0, m0_, l0); split2_bf16(comp(rb[i + 2], e), comp(rb[i + 3], e), h1, m1, l1);
                            *reinterpret_cast<uint2*>(dh) = make_uint2(h0, h1); *reinterpret_cast<uint2*>(dm) = make_uint2(m0_, m1); *reinterpret_cast<uint2*>(dl) = make_uint2(l0, l1);
                        } else if constexpr (GB == 2) {
                            split2_bf16(comp(rb[i], e), comp(rb[i + 1], e), h0, m0_, l0);
                            *reinterpret_cast<unsigned*>(dh) = h0; *reinterpret_cast<unsigned*>(dm) = m0_; *reinterpret_cast<unsigned*>(dl) = l0;
                        } else {
                            split2_bf16(comp(rb[i], e), 0.f, h0, m0_, l0);
                            *dh = (unsigned short)h0; *dm = (unsigned short)m0_; *dl = (unsigned short)l0;
                        }
                    }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NUA; i++) {
            if constexpr (A_KC) {
                As[buf][a_k[i] + 0][a_r[i]] = ra[i].x; As[buf][a_k[i] + 1][a_r[i]] = ra[i].y;
                As[buf][a_k[i] + 2][a_r[i]] = ra[i].z; As[buf][a_k[i] + 3][a_r[i]] = ra[i].w;
            } else {
                As[buf][a_k[i]][a_r[i] + 0] = ra[i].x; As[buf][a_k[i]][a_r[i] + 1] = ra[i].y;
                As[buf][a_k[i]][a_r[i] + 2] = ra[i].z; As[buf][a_k[i]][a_r[i] + 3] = ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < NUB; i++) {
            if constexpr (B_KC) {
                Bs[buf][b_k[i] + 0][b_r[i]] = rb[i].x; Bs[buf][b_k[i] + 1][b_r[i]] = rb[i].y;
                Bs[buf][b_k[i] + 2][b_r[i]] = rb[i].z; Bs[buf][b_k[i] + 3][b_r[i]] = rb[i].w;
            } else {
                Bs[buf][b_k[i]][b_r[i] + 0] = rb[i].x; Bs[buf][b_k[i]][b_r[i] + 1] = rb[i].y;
                Bs[buf][b_k[i]][b_r[i] + 2] = rb[i].z; Bs[buf][b_k[i]][b_r[i] + 3] = rb[i].w;
            }
        }
    };

    const int nk = (z.kend > z.kbeg) ? (z.kend - z.kbeg + BKT - 1) / BKT : 0;
    const int kl = lane >> 5, cl = lane & 31;
    // Operand fetch from LDS runs one k-pair ahead of the MFMAs in a second register set: without it hipcc emits
    // {ds_read, s_waitcnt lgkmcnt(0), 4 MFMA} per k-pair into ONE register set and the matrix pipe idles for the LDS latency
    // every 256 cycles.
    auto compute = [&](int buf) {
        float a[2][TM], b[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) a[0][i] = As[buf][kl][wm * WM + i * 32 + cl];
#pragma unroll
        for (int j = 0; j < TN; j++) b[0][j] = Bs[buf][kl][wn * WN + j * 32 + cl];
#pragma unroll
        for (int kk = 0; kk < BKT / 2; kk++) {
            const int c = kk & 1, nx = c ^ 1;
            if (kk + 1 < BKT / 2) {
#pragma unroll
                for (int i = 0; i < TM; i++) a[nx][i] = As[buf][(kk + 1) * 2 + kl][wm * WM + i * 32 + cl];
#pragma unroll
                for (int j = 0; j < TN; j++) b[nx][j] = Bs[buf][(kk + 1) * 2 + kl][wn * WN + j * 32 + cl];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (the scheduler otherwise sinks it below them)
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto compute_split = [&]() {
#pragma unroll
        for (int ks = 0; ks < BKT / 16; ks++) {
            bf16x8_t a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int q = 0; q < 3; q++) a[i][q] = *reinterpret_cast<const bf16x8_t*>(adrA(q, 2 * ks + kl, wm * WM + i * 32 + cl, 0));
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int q = 0; q < 3; q++) b[j][q] = *reinterpret_cast<const bf16x8_t*>(adrB(q, 2 * ks + kl, wn * WN + j * 32 + cl, 0));
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {   // smallest terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
                }
        }
    };
    if (nk > 0) {
        gload(z.kbeg, ra0, rb0, sa0);
        lstore(std::bool_constant<SPLIT>{}, 0, ra0, rb0, sa0);
    }
    long long tr1 = 0, tr2 = 0;
    if constexpr (SPLIT) {
        // One LDS buffer (the bf16 image of a tile is 1.5x the f32 one); registers hold the next k-tile meanwhile.  (A second register
        // set with tiles kt+1 and kt+2 both in flight measured the same to slower: the loop is bound by the operand split's VALU
        // work and the barriers, not by load latency, and the extra 24-32 VGPRs cost the 256x32 tile an occupancy step.)
        __syncthreads();
        for (int kt = 0; kt < nk; kt++) {
            if (kt + 1 < nk) gload(z.kbeg + (kt + 1) * BKT, ra0, rb0, sa0);
            compute_split();
            __syncthreads();
            if (kt + 1 < nk) { lstore(std::bool_constant<SPLIT>{}, 0, ra0, rb0, sa0); __syncthreads(); }
        }
        // Non-finite operands.  The three-way split of +-Inf (and of a finite value within one bf16 ulp of FLT_MAX) is Inf + NaN + NaN, so
        // where the f32 pipe would produce +-Inf this loop produces NaN -- and the step's gradient sanitiser (training_loop.py:308:
        // nan -> 0, +-inf -> +-1e5) tells the two apart.  A tile that comes out of the loop with any non-finite accumulator is therefore
        // recomputed on the f32 MFMA pipe, here, before the epilogue (block-uniform branch; one LDS buffer of the f32 image fits inside
        // the bf16 image; never taken on finite data: the check is one fma per accumulator register).
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) chk = __builtin_fmaf(acc[i][j][r], 0.f, chk);   // NaN iff some accumulator is NaN / Inf
        if (__builtin_expect(__syncthreads_or(chk != chk), 0)) {
            // cold path: everything it needs beyond the loaders' state is recomputed from an opaque copy of the thread index, so that
            // none of its addressing is hoisted above the hot loop (it cost the 128x64 instantiations an occupancy step otherwise)
            int t2 = tid;
            asm volatile("" : "+v"(t2));
            // ONE buffer of the f32 image, B right behind A (the double-buffered As / Bs pointers of the f32 instantiation would reach past this
            // instantiation's LDS allocation, which is sized for the bf16 image)
            float (*A1)[LDA] = reinterpret_cast<float (*)[LDA]>(ldetr_smem);
            float (*B1)[LDB] = reinterpret_cast<float (*)[LDB]>(ldetr_smem + BKT * LDA);
            static_assert((size_t)BKT * (LDA + LDB) * sizeof(float) <= (size_t)3 * (BKT / 8) * (PLA + PLB), "the f32 image of one k-tile must fit inside the bf16 image");
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
            reset_kstate();
            const int kl2 = (t2 & 63) >> 5, cl2 = t2 & 31, w2 = t2 >> 6, wm2 = w2 / WGN, wn2 = w2 % WGN;
            for (int kt = 0; kt < nk; kt++) {
                gload(z.kbeg + kt * BKT, ra0, rb0, sa0);
                if constexpr (A_DEFER) {
                    if (a_defer) {
#pragma unroll
                        for (int i = 0; i < NUA; i++) { ra0[i].x *= sa0[i].x; ra0[i].y *= sa0[i].y; ra0[i].z *= sa0[i].z; ra0[i].w *= sa0[i].w; }
                    }
                }
#pragma unroll
                for (int i = 0; i < NUA; i++) {
                    int r, k; unitA(t2, i, r, k);
                    if constexpr (A_KC) { A1[k + 0][r] = ra0[i].x; A1[k + 1][r] = ra0[i].y; A1[k + 2][r] = ra0[i].z; A1[k + 3][r] = ra0[i].w; }
                    else { A1[k][r + 0] = ra0[i].x; A1[k][r + 1] = ra0[i].y; A1[k][r + 2] = ra0[i].z; A1[k][r + 3] = ra0[i].w; }
                }
#pragma unroll
                for (int i = 0; i < NUB; i++) {
                    int r, k; unitB(t2, i, r, k);
                    if constexpr (B_KC) { B1[k + 0][r] = rb0[i].x; B1[k + 1][r] = rb0[i].y; B1[k + 2][r] = rb0[i].z; B1[k + 3][r] = rb0[i].w; }
                    else { B1[k][r + 0] = rb0[i].x; B1[k][r + 1] = rb0[i].y; B1[k][r + 2] = rb0[i].z; B1[k][r + 3] = rb0[i].w; }
                }
                __syncthreads();
#pragma unroll 1
                for (int kk = 0; kk < BKT / 2; kk++) {
#pragma unroll
                    for (int i = 0; i < TM; i++) {
                        const float a = A1[kk * 2 + kl2][wm2 * WM + i * 32 + cl2];
#pragma unroll
                        for (int j = 0; j < TN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, B1[kk * 2 + kl2][wn2 * WN + j * 32 + cl2], acc[i][j], 0, 0, 0);
                    }
                }
                __syncthreads();
            }
        }
    } else
    {
        __syncthreads();
        if (LDETR_TILE_TRACE && p.trace) tr1 = wall_clock64();
        if (!LDETR_TILE_TRACE) __builtin_amdgcn_sched_barrier(0);   // (the stamp sites double as scheduling fences: without either, hipcc's schedule of this kernel is 7 % slower end to end)
        for (int kt = 0; kt < nk; kt++) {
            const int buf = kt & 1;
            if (kt + 1 < nk) gload(z.kbeg + (kt + 1) * BKT, ra0, rb0, sa0);
            compute(buf);
            if (kt + 1 < nk) lstore(std::false_type{}, buf ^ 1, ra0, rb0, sa0);   // (writing these between the MFMAs of the last k-pairs measured slower:
            __syncthreads();                               //  the vmcnt wait then sits in the middle of the MFMA stream)
        }
    }

    if (LDETR_TILE_TRACE && p.trace) tr2 = wall_clock64();
    if (!LDETR_TILE_TRACE) __builtin_amdgcn_sched_barrier(0);
    auto trace_out = [&]() {
        if (LDETR_TILE_TRACE && p.trace && threadIdx.x == 0) {
            long long* t = p.trace + 5 * (((long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
            t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = wall_clock64();
            t[4] = ((long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));   // XCC_ID, HW_ID
        }
    };
    // Split-K fix-up: every slice parks its raw tile in the workspace (register order, so all traffic is coalesced), the slice
    // that arrives last at the tile's counter sums the slices in slice order (deterministic, unlike the atomic path) and
    // carries on into the ordinary fused epilogue.  No zero-fill of C and no second epilogue launch.
    bool direct = p.splitk <= 1;
    if (p.splitk > 1 && p.ws) {
        const int zb = p.zmode == 1 ? p.pstep * p.pstep : (p.zmode == 2 ? p.ntaps : 1);
        const int ks = blockIdx.z / zb;
        const long tile = ((long)(blockIdx.z - ks * zb) * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        float* slot0 = p.ws + tile * p.splitk * (long)(BM * BN);
        float* mine = slot0 + (long)ks * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    __hip_atomic_store(mine + ((i * TN + j) * 16 + r) * NT + tid, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Agent-scope (sc1) stores/loads go past the per-XCD L2s, so draining this wave's stores is all the ordering the counter
        // needs; a full __threadfence() here writes back + invalidates the whole 4 MiB L2 and measured 100-150 us per launch.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int s_last;
        __syncthreads();
        if (tid == 0) {
            int old = atomicAdd(p.ws_count + tile, 1);
            s_last = (old == p.splitk - 1);
            // re-armed for the next launch; agent-scope store like the atomics that read it (a plain store would sit in this XCD's L2)
            if (s_last) __hip_atomic_store(p.ws_count + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) { trace_out(); return; }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        for (int sl = 0; sl < p.splitk; sl++) {
            const float* src = slot0 + (long)sl * (BM * BN);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        acc[i][j][r] += __hip_atomic_load(src + ((i * TN + j) * 16 + r) * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        direct = true;
    }

    // Epilogue.  acc[i][j][r]: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col = lane&31 of the 32x32 tile.
    const GemmEpilogue& ep = p.ep;
    const float inv_keep = ep.p_drop > 0.f ? 1.f / (1.f - ep.p_drop) : 1.f;
    if (direct && p.ep_vec) {
        // Row-major epilogue: the accumulator tile goes through LDS (the operand buffers are free now) so that every thread
        // handles float4 pieces of rows: residual / mask / scale reads and the store are 16 B per lane and 4x fewer instructions
        // than the MFMA C-layout's 4 B per lane.
        constexpr int CP = BN + 4;                                                    // row pitch (16 B aligned, conflict-free)
        constexpr int LDSF = 2 * BKT * (LDA + LDB);                                   // floats available
        constexpr int PASSES = (BM * CP + LDSF - 1) / LDSF, RP = BM / PASSES;         // rows staged per pass
        static_assert(BM % PASSES == 0 && RP * CP <= LDSF && (RP % 32) == 0, "epilogue staging does not fit");
        static_assert(!SPLIT || RP * CP * 4 <= 3 * (BKT / 8) * (PLA + PLB), "epilogue staging does not fit the split path's LDS image");
        float* Cs = ldetr_smem;
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            if (ps > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int rb = wm * WM + i * 32 - ps * RP;                            // wave-uniform
                if (rb < 0 || rb >= RP) continue;
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        Cs[(rb + (r & 3) + 8 * (r >> 2) + 4 * kl) * CP + wn * WN + j * 32 + cl] = acc[i][j][r];
            }
            __syncthreads();
            for (int u = tid; u < RP * (BN / 4); u += NT) {
                const int rl = u / (BN / 4), c4 = (u - rl * (BN / 4)) * 4;
                const int m = m0 + ps * RP + rl, n = n0 + c4;
                if (m >= z.M || n >= p.N) continue;
                long orow = m;
                if (p.zmode == 1 && p.pstep > 1) {
                    int per = z.DH2 * z.DW2;
                    int nn = m / per; int rem = m - nn * per;
                    int y2 = rem / z.DW2; int x2 = rem - y2 * z.DW2;
                    orow = ((long)nn * p.A.DH + (y2 * p.pstep + z.py)) * p.A.DW + (x2 * p.pstep + z.px);
                }
                const int samp = (ep.samp_scale && p.pix_per_sample > 0) ? (int)((unsigned)orow / (unsigned)p.pix_per_sample) : 0;
                const float4 a4 = *reinterpret_cast<const float4*>(Cs + rl * CP + c4);
                // (plain ifs: a `cond ? *ptr : constant` on float4 makes hipcc select between ADDRESSES and park the constant in scratch)
                float4 cs4 = make_float4(1.f, 1.f, 1.f, 1.f), cb4 = make_float4(0.f, 0.f, 0.f, 0.f), rs4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ep.col_scale) cs4 = *reinterpret_cast<const float4*>(ep.col_scale + n);
                if (ep.col_bias) cb4 = *reinterpret_cast<const float4*>(ep.col_bias + n);
                if (ep.residual) rs4 = *reinterpret_cast<const float4*>(ep.residual + orow * ep.ldr + n);
                float4 mk4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ep.mask_mode) mk4 = *reinterpret_cast<const float4*>(ep.mask_src + orow * ep.ldm + n);
                float* dst = p.C + z.c_off + orow * p.ldc + n;
                float4 o;
                o.x = apply_epilogue(ep, a4.x, orow, n + 0, samp, p.ldc, inv_keep, cs4.x, cb4.x, rs4.x, mk4.x);
                o.y = apply_epilogue(ep, a4.y, orow, n + 1, samp, p.ldc, inv_keep, cs4.y, cb4.y, rs4.y, mk4.y);
                o.z = apply_epilogue(ep, a4.z, orow, n + 2, samp, p.ldc, inv_keep, cs4.z, cb4.z, rs4.z, mk4.z);
                o.w = apply_epilogue(ep, a4.w, orow, n + 3, samp, p.ldc, inv_keep, cs4.w, cb4.w, rs4.w, mk4.w);
                if (ep.accumulate) { const float4 c = *reinterpret_cast<const float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
                *reinterpret_cast<float4*>(dst) = o;
            }
        }
        trace_out(); return;
    }
    float cs[TN], cb[TN], sfc[TN];
#pragma unroll
    for (int j = 0; j < TN; j++) {
        int n = n0 + wn * WN + j * 32 + cl;
        cs[j] = (ep.col_scale && n < p.N) ? ep.col_scale[n] : 1.f;
        cb[j] = (ep.col_bias && n < p.N) ? ep.col_bias[n] : 0.f;
        sfc[j] = (p.scol && n < p.N) ? p.scol[(long)z.samp * p.scol_ld + n] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            if (m >= z.M) continue;
            long orow = m;
            if (p.zmode == 1 && p.pstep > 1) {
                int per = z.DH2 * z.DW2;
                int n = m / per; int rem = m - n * per;
                int y2 = rem / z.DW2; int x2 = rem - y2 * z.DW2;
                orow = ((long)n * p.A.DH + (y2 * p.pstep + z.py)) * p.A.DW + (x2 * p.pstep + z.px);
            }
            // only the per-sample scale needs the sample index; the division was costing more issue slots than a K=64 main loop
            int samp = (ep.samp_scale && p.pix_per_sample > 0) ? (int)((unsigned)orow / (unsigned)p.pix_per_sample) : 0;
            const float sfr = p.srow ? p.srow[(long)z.samp * p.srow_ld + m] : 1.f;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                int n = n0 + wn * WN + j * 32 + cl;
                if (n >= p.N) continue;
                float* dst = p.C + z.c_off + orow * p.ldc + n;
                const float a = acc[i][j][r] * sfr * sfc[j];
                if (!direct) {
                    atomicAdd(dst, a * (ep.row_scale ? ep.alpha * ep.row_scale[orow] : ep.alpha));   // raw partial sums; the rest of the epilogue runs in epilogue_kernel
                } else {
                    float v = apply_epilogue(ep, a, orow, n, samp, p.ldc, inv_keep, cs[j], cb[j], ep.residual ? ep.residual[orow * ep.ldr + n] : 0.f,
                                           ep.mask_mode ? ep.mask_src[orow * ep.ldm + n] : 0.f);
                    if (ep.accumulate) *dst += v; else *dst = v;
                }
            }
        }
    }
    trace_out();
}

// ---------------------------------------------------------------------------------------------
// Latency-bound dense GEMMs (the transformer's 256-wide projections on a few hundred tokens: fewer 64x64 tiles than CUs).
// One block owns a 32x32 tile of C; its NW waves each take a contiguous 1/NW of K and stream their own A/B fragments
// global -> registers in MFMA operand order (no LDS staging, no barrier in the k-loop, every load of a wave in flight at once),
// the NW partial tiles are summed through LDS in wave order (deterministic) and the fused epilogue writes float4 rows.
// Replaces {zero-fill, split-K atomics, epilogue} = 3 graph nodes by one launch.
//   TA/TB: 0 = operand stored [rows, K] (k-contiguous), 1 = stored [K, rows] (row-contiguous).
template <int T>
__device__ __forceinline__ void small_load(const Operand& o, int row, int R, int k0, int kl, int kend, float (&f)[16]) {
    // f[t] holds reduction index k0 + 16*kl + t of `row` (the two half-waves interleave as the x2 MFMA expects)
    const int kb = k0 + 16 * kl;
    if (T == 0) {
        const float* src = o.p + (long)row * o.ld + kb;
        if (row < R && o.vec) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (kb + 4 * j < kend) v = *reinterpret_cast<const float4*>(src + 4 * j);
                f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; t++) f[t] = (row < R && kb + t < kend) ? src[t] : 0.f;
        }
    } else {
        const float* src = o.p + (long)kb * o.ld + row;
#pragma unroll
        for (int t = 0; t < 16; t++) f[t] = (row < R && kb + t < kend) ? src[(long)t * o.ld] : 0.f;
    }
}

// Scalar-addressed variant (K a multiple of 32, operand below 2 GiB): the lane's byte offset is computed once, the k position goes
// into the buffer load's scalar offset -- no per-round 64-bit address arithmetic or bounds tests (11 VALU instructions per MFMA
// were going into them; VALU issue is not hidden behind the matrix pipe here either).
template <int T>
__device__ __forceinline__ void small_load_fast(const __amdgpu_buffer_rsrc_t& rs, int voff, int k0, int ld, float (&f)[16]) {
    if (T == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * j, k0 * 4, 0);
            f[4 * j] = __int_as_float(v[0]); f[4 * j + 1] = __int_as_float(v[1]); f[4 * j + 2] = __int_as_float(v[2]); f[4 * j + 3] = __int_as_float(v[3]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < 16; t++) f[t] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, (k0 + t) * ld * 4, 0));
    }
}

// Body of one 32x32 tile (bx, by) and K slice bz of `sk`; shared by the single-problem kernel and the paired launch below.
template <int TA, int TB, int NW, bool FAST = false>
__device__ __forceinline__ void gemm_small_body(const GemmParams& p, const int bx, const int by, const int bz, const int sk, const int gx) {
    using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
    __shared__ float red[NW][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cl = lane & 31, kl = lane >> 5;
    const int m0 = by * 32, n0 = bx * 32;
    // reduction range of this block (split-K over blockIdx.z when the launch asked for it), then of this wave
    const int chunks_all = (p.K + 31) / 32;
    const int cper = (chunks_all + sk - 1) / sk;
    const int kblk0 = bz * cper * 32, kblk1 = min(p.K, kblk0 + cper * 32);
    const int chunks = kblk1 > kblk0 ? (kblk1 - kblk0 + 31) / 32 : 0, per = (chunks + NW - 1) / NW;
    const int kbeg = kblk0 + wave * per * 32, kend = min(kblk1, kbeg + per * 32);
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A.p), 0, FAST ? 0x7fffffff : 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.B.p), 0, FAST ? 0x7fffffff : 0, 0x00020000);
    const int voffA = (m0 + cl < p.M) ? (TA == 0 ? ((m0 + cl) * (int)p.A.ld + 16 * kl) * 4 : (16 * kl * (int)p.A.ld + m0 + cl) * 4) : (int)0x80000000;
    const int voffB = (n0 + cl < p.N) ? (TB == 0 ? ((n0 + cl) * (int)p.B.ld + 16 * kl) * 4 : (16 * kl * (int)p.B.ld + n0 + cl) * 4) : (int)0x80000000;
    auto loadA = [&](int k0, float (&f)[16]) {
        if constexpr (FAST) small_load_fast<TA>(rsA, voffA, k0, (int)p.A.ld, f); else small_load<TA>(p.A, m0 + cl, p.M, k0, kl, kend, f);
        if constexpr (TA == 0) {
            if (p.A.scale) {   // one factor per k for all rows (FrozenBN scale of the incoming gradient: the 1x1 data gradients routed here)
                const int kb = k0 + 16 * kl;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (kb + 4 * j < kend) {
                        const float4 s4 = *reinterpret_cast<const float4*>(p.A.scale + kb + 4 * j);
                        f[4 * j] *= s4.x; f[4 * j + 1] *= s4.y; f[4 * j + 2] *= s4.z; f[4 * j + 3] *= s4.w;
                    }
                }
            }
        }
    };
    auto loadB = [&](int k0, float (&f)[16]) {
        if constexpr (FAST) small_load_fast<TB>(rsB, voffB, k0, (int)p.B.ld, f); else small_load<TB>(p.B, n0 + cl, p.N, k0, kl, kend, f);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    float a0[16], b0[16], a1[16], b1[16];
    const bool want_rs = TA == 1 && p.ep.a_rowsum != nullptr && bx == 0;   // the A panel is the same for every column block
    float rs = 0.f;
    if (kbeg < kend) {
        loadA(kbeg, a0);
        loadB(kbeg, b0);
    }
    for (int k = kbeg; k < kend; k += 64) {
        if (k + 32 < kend) {
            loadA(k + 32, a1);
            loadB(k + 32, b1);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc, 0, 0, 0);
        if (want_rs) {
#pragma unroll
            for (int t = 0; t < 16; t++) rs += a0[t];
        }
        if (k + 32 >= kend) break;
        if (k + 64 < kend) {
            loadA(k + 64, a0);
            loadB(k + 64, b0);
        }
#pragma unroll
        for (int t = 0; t < 16; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc, 0, 0, 0);
        if (want_rs) {
#pragma unroll
            for (int t = 0; t < 16; t++) rs += a1[t];
        }
    }
    __shared__ float rsum[NW][32];
    if (want_rs) {
        rs += __shfl_xor(rs, 32);
        if (kl == 0) rsum[wave][cl] = rs;
    }
#pragma unroll
    for (int r = 0; r < 16; r++) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kl][cl] = acc[r];
    __syncthreads();
    if (want_rs && tid < 32 && m0 + tid < p.M) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NW; w++) tot += rsum[w][tid];
        if (sk > 1) atomicAdd(p.ep.a_rowsum + m0 + tid, tot);   // one writer per K slice
        else p.ep.a_rowsum[m0 + tid] += tot;                    // unique writer per m (column block 0), launches on one stream are ordered
    }
    const GemmEpilogue& ep = p.ep;
    const float inv_keep = ep.p_drop > 0.f ? 1.f / (1.f - ep.p_drop) : 1.f;
    // Split-K: every slice parks its 32x32 partial in the workspace, the slice arriving last at the tile's counter adds them up
    // in slice order and runs the epilogue (same protocol as gemm_f32_kernel's fix-up: agent-scope stores / loads + vmcnt(0)).
    float* slot0 = nullptr;
    if (sk > 1) {
        const long tile = (long)by * gx + bx;
        slot0 = p.ws + tile * sk * 1024;
        float* mine = slot0 + (long)bz * 1024;
        for (int q = tid; q < 256; q += NW * 64) {
            const int row = q >> 3, c4 = (q & 7) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) sum += red[w][row][c4 + e];
                __hip_atomic_store(mine + q * 4 + e, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __shared__ int s_last_small;
        __syncthreads();
        if (tid == 0) {
            const int old = atomicAdd(p.ws_count + tile, 1);
            s_last_small = (old == sk - 1);
            if (s_last_small) __hip_atomic_store(p.ws_count + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last_small) return;
    }
    for (int q = tid; q < 256; q += NW * 64) {   // 256 float4 groups: row = q / 8, cols 4*(q % 8) .. +3
        const int row = q >> 3, c4 = (q & 7) * 4;
        const int m = m0 + row;
        if (m >= p.M) continue;
        float v[4];
        if (sk > 1) {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = 0.f;
            float part[8][4];                       // sk <= 8: every slice's loads are issued before the first add (one latency, not sk)
#pragma unroll
            for (int z = 0; z < 8; z++)
#pragma unroll
                for (int e = 0; e < 4; e++)
                    part[z][e] = z < sk ? __hip_atomic_load(slot0 + (long)z * 1024 + q * 4 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
            for (int z = 0; z < 8; z++)
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] += part[z][e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float sum = 0.f;
#pragma unroll
                for (int w = 0; w < NW; w++) sum += red[w][row][c4 + e];
                v[e] = sum;
            }
        }
        const int samp = (ep.samp_scale && p.pix_per_sample > 0) ? m / p.pix_per_sample : 0;
        float* dst = p.C + (long)m * p.ldc + n0 + c4;
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (n0 + c4 + e < p.N) v[e] = apply_epilogue(ep, v[e], m, n0 + c4 + e, samp, p.ldc, inv_keep, ep.col_scale ? ep.col_scale[n0 + c4 + e] : 1.f, ep.col_bias ? ep.col_bias[n0 + c4 + e] : 0.f,
                                                             ep.residual ? ep.residual[(long)m * ep.ldr + n0 + c4 + e] : 0.f,
                                                             ep.mask_mode ? ep.mask_src[(long)m * ep.ldm + n0 + c4 + e] : 0.f);
        if (n0 + c4 + 3 < p.N && (p.ldc & 3) == 0 && ((((uintptr_t)p.C) & 15) == 0)) {
            float4 o = make_float4(v[0], v[1], v[2], v[3]);
            if (ep.accumulate) { float4 c = *reinterpret_cast<float4*>(dst); o.x += c.x; o.y += c.y; o.z += c.z; o.w += c.w; }
            *reinterpret_cast<float4*>(dst) = o;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (n0 + c4 + e < p.N) { if (ep.accumulate) dst[e] += v[e]; else dst[e] = v[e]; }
        }
    }
}

template <int TA, int TB, int NW, bool FAST = false>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(GemmParams p) {
    gemm_small_body<TA, TB, NW, FAST>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, gridDim.x);
}

// Two independent small problems in ONE launch (the data and the weight gradient of a linear layer: both read dY).  These
// GEMMs sit at the launch-latency floor (~6 us for 20 MFLOP), so a launch saved is their whole cost saved.  Tiles of problem 0
// come first in the linear block order (K slice major), then problem 1's; each problem may carry its own in-kernel split-K.
template <int TA0, int TB0, int TA1, int TB1, int NW, bool FAST = false>
__global__ __launch_bounds__(NW * 64) void gemm_small_pair_kernel(GemmParams p0, GemmParams p1, int gx0, int nt0, int sk0, int gx1, int nt1, int sk1) {
    const int b = blockIdx.x;
    if (b < nt0 * sk0) { const int t = b % nt0; gemm_small_body<TA0, TB0, NW, FAST>(p0, t % gx0, t / gx0, b / nt0, sk0, gx0); }
    else { const int c = b - nt0 * sk0, t = c % nt1; gemm_small_body<TA1, TB1, NW, FAST>(p1, t % gx1, t / gx1, c / nt1, sk1, gx1); }
}

static inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }


// Caller-provided scratch for the split-K fix-up (ldetr_set_workspace): WS_COUNTERS ints of arrival counters, then partial tiles.
// Both areas are handed out as RINGS, one fresh slice per launch: launches of one stream are ordered anyway, but a captured
// hipGraph may run independent branches concurrently (autograd hops streams for AccumulateGrad nodes),, and two split-K
// kernels must never share a counter / partial area.
// Counters are zero at rest (the last-arriving block re-arms its own), so a counter slice can be reused without clearing;
// partial tiles need no clearing at all.  A launch that needs more than the whole ring falls back to fp32 atomics.
constexpr long WS_COUNTERS = 262144;
struct Workspace { void* ptr; size_t bytes; size_t counter_cursor; size_t partial_cursor; };
static Workspace g_workspace[64];
static Workspace& workspace_for_current_device() {
    static Workspace none = {nullptr, 0, 0, 0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return none;
    return g_workspace[dev];
}

// Split-K plan of a small-tile problem: few tiles + a long reduction (the decoders' 2048-wide FFN on 144 rows) would have 40 blocks
// walk K one after the other; split K over up to 8 blocks per tile (>= 256 of K each) and reduce through the workspace in-kernel.
static int plan_small_split(GemmParams& p, long blocks) {
    int sk = 1;
    if (blocks <= 128 && p.K >= 1024) {
        sk = p.K / 256; if (sk > 8) sk = 8;
        while (sk > 1 && blocks * sk > 640) sk--;
        if (sk > 1) {
            Workspace& w = workspace_for_current_device();
            const size_t need = ((size_t)blocks * sk * 1024 * sizeof(float) + 255) & ~(size_t)255;
            const size_t pbytes = w.bytes > WS_COUNTERS * sizeof(int) ? w.bytes - WS_COUNTERS * sizeof(int) : 0;
            if (w.ptr && need <= pbytes) {
                if (w.counter_cursor + blocks > (size_t)WS_COUNTERS) w.counter_cursor = 0;
                if (w.partial_cursor + need > pbytes) w.partial_cursor = 0;
                p.ws_count = reinterpret_cast<int*>(w.ptr) + w.counter_cursor;
                p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(w.ptr) + WS_COUNTERS * sizeof(int) + w.partial_cursor);
                w.counter_cursor += blocks; w.partial_cursor += need;
            } else sk = 1;
        }
    }
    return sk;
}

static bool small_fast_ok(const GemmParams& p, int ta, int tb) {
    static const int on = (int)knob("SMALL_FAST", 1);
    const long a_bytes = (ta ? (long)p.K * p.A.ld : (long)p.M * p.A.ld) * 4, b_bytes = (tb ? (long)p.K * p.B.ld : (long)p.N * p.B.ld) * 4;
    return on && (p.K % 32) == 0 && a_bytes < 0x7fffffffL && b_bytes < 0x7fffffffL && (ta || (p.A.ld % 4) == 0) && (tb || (p.B.ld % 4) == 0) &&
           ((((uintptr_t)p.A.p) | ((uintptr_t)p.B.p)) & 15) == 0;
}

template <int TA, int TB>
static int launch_small(GemmParams& p, hipStream_t st) {
    dim3 grid(cdiv(p.N, 32), cdiv(p.M, 32), 1);
    const long blocks = (long)grid.x * grid.y;
    const int sk = plan_small_split(p, blocks);
    grid.z = sk;
    const int kblock = (p.K + sk - 1) / sk;
    const bool fast = small_fast_ok(p, TA, TB);
    const bool w8 = blocks * sk <= 256 && kblock >= 512;
    if (fast) {
        if (w8) hipLaunchKernelGGL((gemm_small_kernel<TA, TB, 8, true>), grid, 512, 0, st, p);
        else hipLaunchKernelGGL((gemm_small_kernel<TA, TB, 4, true>), grid, 256, 0, st, p);
    } else {
        if (w8) hipLaunchKernelGGL((gemm_small_kernel<TA, TB, 8>), grid, 512, 0, st, p);
        else hipLaunchKernelGGL((gemm_small_kernel<TA, TB, 4>), grid, 256, 0, st, p);
    }
    t_launches_f32++;
    note_engine_launch(2, 32, 32, 0, w8 ? 8 : 4, fast ? 1 : 0, 0, sk, blocks * sk, TA * 2 + TB);
    return check_launch("gemm_small");
}

// Second phase of a split-K contraction: C holds sum_k (already scaled by alpha); apply the rest of the epilogue in place.
struct EpiParams { GemmEpilogue ep; float* C; long ldc; long rows; int N; int pix_per_sample; };

__global__ __launch_bounds__(256) void gemm_epilogue_kernel(EpiParams q) {
    const float inv_keep = q.ep.p_drop > 0.f ? 1.f / (1.f - q.ep.p_drop) : 1.f;
    const long total = q.rows * q.N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long row = i / q.N; int n = (int)(i - row * q.N);
        int samp = (q.ep.samp_scale && q.pix_per_sample > 0) ? (int)(row / q.pix_per_sample) : 0;
        float* dst = q.C + row * q.ldc + n;
        *dst = apply_epilogue(q.ep, *dst, row, n, samp, q.ldc, inv_keep, q.ep.col_scale ? q.ep.col_scale[n] : 1.f, q.ep.col_bias ? q.ep.col_bias[n] : 0.f,
                              q.ep.residual ? q.ep.residual[row * q.ep.ldr + n] : 0.f,
                              q.ep.mask_mode ? q.ep.mask_src[row * q.ep.ldm + n] : 0.f);
    }
}

static bool epilogue_is_linear(const GemmEpilogue& ep) {
    return !ep.col_scale && !ep.samp_scale && !ep.col_bias && !ep.residual && !ep.act && !ep.mask_mode && ep.p_drop == 0.f &&
           ep.out_scale == 1.f;
}


static void init_operand(Operand& o) { memset(&o, 0, sizeof(o)); o.KW = 1; o.KH = 1; o.stride = 1; o.C = 1; o.Cr = 1; }

// Launch policy.  Tile 128x128 (BK 16) when it alone fills the chip; otherwise 64x64 (BK 32), and when even those
// tiles leave most of the 256 CUs idle the reduction is split across grid.z (fp32 atomics into a zeroed C) with
// the non-linear part of the epilogue applied by a second streaming pass.
//   out_rows: rows of C touched (for the memset / epilogue pass); zbase: grid.z multiplicity before split-K;
//   auto_split: the caller allows the policy to split K (explicit p.splitk > 1 is always honoured);
//   caller_zeroed: C was already zeroed by the caller (weight-gradient entry points).
#ifndef T128_BK
#define T128_BK 32
#endif
#ifndef T128_WAVES
#define T128_WAVES 8
#endif
#ifndef LDETR_EPILOGUE_VEC
#define LDETR_EPILOGUE_VEC 1
#endif
#ifndef T12864_SPLIT_MIN_TILES
#define T12864_SPLIT_MIN_TILES 64
#endif
#ifndef T12864_WAVES
#define T12864_WAVES 8
#define T12864_BK 32
#endif

// Dense GEMMs with fewer 64x64 tiles than this (and at most this much work) take the register-streaming 32x32 kernel.
static const long SMALL_GEMM_TILES = 256;   // 64x64-tile count below which the small-tile kernels run
constexpr long SMALL_GEMM_MNK = 1l << 30;
#ifndef WGRAD_MIN_K
#define WGRAD_MIN_K 512
#endif

// Transient scratch for other kernels of the library (partial sums of the row reductions): a slice of the same ring, valid for
// the launches the caller enqueues next on its stream.  nullptr when no workspace is registered or the request does not fit.
// Split-K scratch for the other contraction kernels of the library (p3_engine.hip): `tiles` arrival counters + `partial_bytes` of partial tiles
// from the same rings; false if no workspace is registered or it is too small.
bool splitk_ws_alloc(long tiles, size_t partial_bytes, float** ws, int** counters) {
    Workspace& w = workspace_for_current_device();
    const size_t need = (partial_bytes + 255) & ~(size_t)255;
    const size_t pbytes = w.bytes > WS_COUNTERS * sizeof(int) ? w.bytes - WS_COUNTERS * sizeof(int) : 0;
    if (!w.ptr || need > pbytes || tiles > WS_COUNTERS) return false;
    if (w.counter_cursor + tiles > (size_t)WS_COUNTERS) w.counter_cursor = 0;
    if (w.partial_cursor + need > pbytes) w.partial_cursor = 0;
    *counters = reinterpret_cast<int*>(w.ptr) + w.counter_cursor;
    *ws = reinterpret_cast<float*>(reinterpret_cast<char*>(w.ptr) + WS_COUNTERS * sizeof(int) + w.partial_cursor);
    w.counter_cursor += tiles; w.partial_cursor += need;
    return true;
}

float* scratch_alloc(size_t bytes) {
    Workspace& w = workspace_for_current_device();
    const size_t pbytes = w.bytes > WS_COUNTERS * sizeof(int) ? w.bytes - WS_COUNTERS * sizeof(int) : 0;
    bytes = (bytes + 255) & ~(size_t)255;
    if (!w.ptr || bytes > pbytes) return nullptr;
    if (w.partial_cursor + bytes > pbytes) w.partial_cursor = 0;
    float* r = reinterpret_cast<float*>(reinterpret_cast<char*>(w.ptr) + WS_COUNTERS * sizeof(int) + w.partial_cursor);
    w.partial_cursor += bytes;
    return r;
}

// Zero-fill of a pitched fp32 matrix for the atomic split-K path.  A kernel rather than hipMemset2DAsync: memset nodes captured into
// a hipGraph were observed to replay out of order with their neighbouring kernel nodes on ROCm 7.2 (garbage gradients on replay).
__global__ void __launch_bounds__(256) zero_fill_kernel(float* __restrict__ dst, long pitch, long width, long rows) {
    const long total = rows * width;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / width;
        dst[r * pitch + (i - r * width)] = 0.f;
    }
}
static int zero_fill(float* dst, long pitch, long width, long rows, hipStream_t st) {
    if (rows <= 0 || width <= 0) return LDETR_OK;
    if (pitch == width) { width *= rows; pitch = width; rows = 1; }   // dense: one long row
    long total = rows * width;
    int g = (int)((total + 255) / 256); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(zero_fill_kernel, g, 256, 0, st, dst, pitch, width, rows);
    return check_launch("zero_fill");
}

static long long* g_trace_buffer = nullptr;   // ldetr_debug_trace_tiles
static std::atomic<int> g_split_bf16_override{-1};   // ldetr_set_split_bf16: -1 = environment / default, else the tile mask (process-wide; atomic: tests toggle it while other host threads may launch)

// Host-side conditions of the kernel's scalar-addressed loads (FAST instantiation): every operand whose view supports them must
// qualify, otherwise the generic instantiation runs.  BKT is 32 for every tile shape.
template <int AMODE, int BMODE>
static bool fast_operands_ok(const GemmParams& p, int Mmax) {
    static const int fast_loads = (int)knob("FAST_LOADS", 63);   // 1 conv, 2 dense B, 4 transposed conv, 8 dense A, 16 row-contiguous dense, 32 pixel-major (weight gradient input)
    constexpr int BKT = 32;
    constexpr bool a_cap = (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT || AMODE == OP_KC_DENSE || AMODE == OP_RC_DENSE || AMODE == OP_RC_PIX);
    constexpr bool b_cap = (BMODE == OP_KC_DENSE || BMODE == OP_RC_WT || BMODE == OP_RC_DENSE || BMODE == OP_RC_PIX || BMODE == OP_KC_WTAP);
    if (!a_cap && !b_cap) return false;
    const long lim = 0x7fffffffL;
    auto taps_ok = [&](int C, int KH, int KW) {   // a k-tile never straddles a tap (C == k-tile measured slower: 483 vs 420 us on the 32-channel 256^2 layer)
        return C > 0 && (C % BKT) == 0 && C >= (p.narrow ? BKT : 2 * BKT) && (long)KH * KW <= 32 && p.samp_pix == 0;
    };
    if (AMODE == OP_KC_CONV) {
        const long padoff = (long)p.A.pad * p.A.sh + (long)p.A.pad * p.A.sw;
        if (!((fast_loads & 1) && p.A.vec && p.zmode == 0 && taps_ok(p.A.C, p.A.KH, p.A.KW) && ((long)p.nsamp * p.A.sn + padoff) * 4 < lim)) return false;
    }
    if (AMODE == OP_KC_CONVT) {
        const long shift = (long)(p.A.KH - 1) * p.A.sh + (long)(p.A.KW - 1) * p.A.sw;
        const int tstep = p.zmode == 1 ? p.pstep : 1;
        if (!((fast_loads & 4) && p.A.vec && taps_ok(p.A.C, p.A.KH, p.A.KW) && tstep == p.A.stride &&
              ((long)p.nsamp * p.A.sn + 2 * shift + p.A.sh + p.A.sw) * 4 < lim)) return false;
    }
    if (AMODE == OP_KC_DENSE) {
        if (!((fast_loads & 8) && p.A.vec && (p.K % BKT) == 0 && (long)Mmax * p.A.ld * 4 < lim && p.samp_pix == 0 && p.zmode == 0)) return false;
    }
    if (BMODE == OP_KC_DENSE) {
        if (!((fast_loads & 2) && p.B.vec && (p.K % BKT) == 0 && (long)p.N * p.B.ld * 4 < lim && p.samp_pix == 0 && (p.zmode == 0 || a_cap))) return false;
    }
    if (AMODE == OP_RC_DENSE) {   // (k0 * ld) * 4 must fit the 32-bit scalar offset
        if (!((fast_loads & 16) && p.A.vec && (long)p.K * p.A.ld * 4 < lim)) return false;
    }
    if (BMODE == OP_RC_DENSE) {
        if (!((fast_loads & 16) && p.B.vec && (long)p.K * p.B.ld * 4 < lim)) return false;
    }
    if (BMODE == OP_KC_WTAP) {
        if (!((fast_loads & 4) && p.B.vec && taps_ok(p.B.C, p.B.KH, p.B.KW) && p.B.C == p.A.C && (long)p.N * p.B.ld * 4 < lim)) return false;
    }
    if (AMODE == OP_RC_PIX) {
        const int DH = p.A.DH, DW = p.A.DW, pad = p.A.tapped ? p.A.pad : 0;
        const bool rows_ok = DH > 0 && DW > 0 && ((DW % BKT) == 0 || ((BKT % DW) == 0 && ((long)DH * DW) % BKT == 0));
        if (!((fast_loads & 32) && p.A.vec && rows_ok && (p.K % BKT) == 0 && (BMODE != OP_RC_PIX || (p.B.DH == DH && p.B.DW == DW)) &&
              ((long)p.nsamp * p.A.sn + (long)pad * (p.A.sh + p.A.sw) + (long)BKT * (p.A.sh + p.A.sw)) * 4 < lim)) return false;
    }
    if (BMODE == OP_RC_PIX) {   // k-tiles of 32 pixels aligned to the image rows
        const int DH = p.B.DH, DW = p.B.DW, pad = p.B.tapped ? p.B.pad : 0;
        const bool rows_ok = DH > 0 && DW > 0 && ((DW % BKT) == 0 || ((BKT % DW) == 0 && ((long)DH * DW) % BKT == 0));
        if (!((fast_loads & 32) && p.B.vec && rows_ok && (p.K % BKT) == 0 &&
              ((long)p.nsamp * p.B.sn + (long)pad * (p.B.sh + p.B.sw) + (long)BKT * (p.B.sh + p.B.sw)) * 4 < lim)) return false;
    }
    if (BMODE == OP_RC_WT) {
        const int C = (AMODE <= OP_KC_WTAP) ? p.A.C : p.B.C;
        if (!((fast_loads & 4) && p.B.vec && taps_ok(p.B.C, p.B.KH, p.B.KW) && p.B.C == C &&
              ((long)p.B.C * p.B.ld + (long)p.B.KH * p.B.KW * p.B.Cr) * 4 < lim)) return false;
    }
    return true;
}

template <int BM, int BN, int BKT, int AMODE, int BMODE, int NWV, bool FAST, bool SPLIT = false>
static int launch_tile_impl(GemmParams& p, dim3 grid, hipStream_t st) {
    p.trace = g_trace_buffer;
    constexpr bool a_kc = AMODE <= OP_KC_WTAP, b_kc = BMODE <= OP_KC_WTAP;
    constexpr size_t lds = SPLIT ? (size_t)3 * (BKT / 8) * ((BM * 16 + (a_kc ? LDETR_KC_PAD : 0)) + (BN * 16 + (b_kc ? LDETR_KC_PAD : 0)))
                                 : (size_t)2 * BKT * ((BM + 2) + (BN + 2)) * sizeof(float);
    auto kern = gemm_f32_kernel<BM, BN, BKT, AMODE, BMODE, NWV, FAST, SPLIT>;
    if (lds > 64 * 1024) {
        // one-time opt-in to > 64 KiB of dynamic LDS for this instantiation, PER DEVICE (a function attribute belongs to the device's code object: a
        // process that drives a second GPU -- tests, a future multi-device host -- must raise it there as well)
        static std::atomic<bool> raised[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
        if (dev < 0 || !raised[dev].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
                set_error("gemm: cannot raise the dynamic LDS limit to %zu bytes", lds);
                return LDETR_ERR_LAUNCH;
            }
            if (dev >= 0) raised[dev].store(true, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(kern, grid, NWV * 64, lds, st, p);
    (SPLIT ? t_launches_split : t_launches_f32)++;
    note_engine_launch(1, BM, BN, BKT, NWV, FAST ? 1 : 0, SPLIT ? 1 : 0, p.splitk > 1 ? p.splitk : 1, (long)grid.x * grid.y * grid.z, AMODE * 16 + BMODE);
    return check_launch("gemm_f32");
}

template <int BM, int BN, int BKT, int AMODE, int BMODE, int NWV = 4>
static int launch_tile(GemmParams& p, dim3 grid, int Mmax, hipStream_t st) {
    constexpr bool cap = (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT || AMODE == OP_KC_DENSE || AMODE == OP_RC_DENSE || AMODE == OP_RC_PIX || BMODE == OP_KC_DENSE || BMODE == OP_RC_WT || BMODE == OP_RC_DENSE || BMODE == OP_RC_PIX || BMODE == OP_KC_WTAP);
    if constexpr (cap) {
        if (fast_operands_ok<AMODE, BMODE>(p, Mmax)) return launch_tile_impl<BM, BN, BKT, AMODE, BMODE, NWV, true>(p, grid, st);
    }
    return launch_tile_impl<BM, BN, BKT, AMODE, BMODE, NWV, false>(p, grid, st);
}

template <int AMODE, int BMODE>
static int launch_gemm(GemmParams& p, int Mmax, long out_rows, int zbase, bool auto_split, bool caller_zeroed, hipStream_t st) {
    const int sk0 = p.splitk > 1 ? p.splitk : 1;
    long t128 = (long)cdiv(Mmax, 128) * cdiv(p.N, 128) * zbase * sk0;
    bool use128 = (t128 >= 384 && Mmax >= 128 && p.N >= 128);   // measured: the 128-tile only wins with >= ~1.5 blocks per CU
    long t64 = (long)cdiv(Mmax, 64) * cdiv(p.N, 64) * zbase;
    // 128x64 tile: every wave owns a 64x32 sub-tile = TWO independent MFMA accumulator chains (the 64x64 tile has one per wave,
    // so any LDS/barrier hiccup idles its SIMD's matrix pipe) and needs 1.5 instead of 2 LDS operand reads per MFMA.
    long t12864 = (long)cdiv(Mmax, 128) * cdiv(p.N, 64) * zbase * sk0;
    bool use12864 = !use128 && t12864 >= 512 && Mmax >= 128;
    if (!use128 && !use12864 && auto_split && p.splitk <= 1 && zbase == 1 && Mmax >= 128 && t12864 >= T12864_SPLIT_MIN_TILES &&
        !(p.ep.accumulate && !epilogue_is_linear(p.ep))) {
        // mid-size problems (ResNet layer2-4 at batch 16): the 8-wave 128x64 tile with the reduction split until its grid fills the
        // chip beats the 4-wave 64x64 tile, as long as every slice keeps >= 512 of K
        const int sk = (int)((512 + t12864 - 1) / t12864);
        if (sk >= 2 && sk <= 8 && p.K / sk >= 512) { use12864 = true; p.splitk = sk; }
    }
    if (!use128 && !use12864 && auto_split && p.splitk <= 1 && t64 < 768 && !(p.ep.accumulate && !epilogue_is_linear(p.ep))) {
        // PMC: with <= 2 resident blocks per CU the single-accumulator waves leave the MFMA pipe ~55% idle; more, shorter blocks fill it
        int want = (int)(((t64 < 160 ? 384 : 1024) + t64 - 1) / t64);
        int maxs = p.K / (t64 < 160 ? 128 : 256);   // keep >= 4 (8) k-tiles of 32 per slice
        int sk = want < maxs ? want : maxs;
        if (sk >= 2) p.splitk = sk;
    }
    {   // development override for policy sweeps (tools/sweep_policy.py): LDETR_DEBUG="FORCE_TILE=t,FORCE_SK=n", t: 1 = 64x64, 2 = 128x64, 3 = 128x128
        static const int ft = (int)knob("FORCE_TILE", 0);
        static const int fs = (int)knob("FORCE_SK", 0);
        if (ft) {
            use128 = ft == 3 && Mmax >= 128 && p.N >= 128; use12864 = ft == 2 && Mmax >= 128;
            if (fs) { int s2 = fs; while (s2 > 1 && p.K / s2 < 64) s2--; p.splitk = (p.ep.accumulate && !epilogue_is_linear(p.ep)) ? 1 : s2; }
        }
    }
    // Narrow outputs (N <= 32: the 32-channel 256^2 StyleGAN2 layers): a 64-wide tile leaves half of every MFMA's columns empty.
    // 256 x 32 tile, four waves stacked along M (each 64 x 32: two accumulator chains, 1.5 LDS operand reads per MFMA).
    constexpr bool narrow_cap = (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT) && (BMODE == OP_KC_DENSE || BMODE == OP_RC_WT || BMODE == OP_KC_WTAP);
    bool use_narrow = false;
    if constexpr (narrow_cap) {
        if (p.N <= 32 && p.N % 4 == 0 && p.splitk <= 1 && (long)cdiv(Mmax, 256) * zbase >= 512) {
            use_narrow = true; use128 = use12864 = false; p.narrow = 1;
        }
    }
    GemmEpilogue full = p.ep;
    const bool split = p.splitk > 1;
    bool fixup = false;
    if (split && !(caller_zeroed && epilogue_is_linear(p.ep)) && !(p.ep.accumulate && epilogue_is_linear(p.ep))) {
        // in-kernel fix-up when the caller registered a workspace that can hold this launch's partial tiles
        Workspace& w = workspace_for_current_device();
        const int bm = (use128 || use12864) ? 128 : 64, bn = use128 ? 128 : 64;
        const long tiles = (long)cdiv(p.N, bn) * cdiv(Mmax, bm) * zbase;
        const size_t need = ((size_t)tiles * p.splitk * bm * bn * sizeof(float) + 255) & ~(size_t)255;
        const size_t pbytes = w.bytes > WS_COUNTERS * sizeof(int) ? w.bytes - WS_COUNTERS * sizeof(int) : 0;
        if (w.ptr && tiles <= WS_COUNTERS && need <= pbytes) {
            if (w.counter_cursor + tiles > (size_t)WS_COUNTERS) w.counter_cursor = 0;
            if (w.partial_cursor + need > pbytes) w.partial_cursor = 0;
            p.ws_count = reinterpret_cast<int*>(w.ptr) + w.counter_cursor;
            p.ws = reinterpret_cast<float*>(reinterpret_cast<char*>(w.ptr) + WS_COUNTERS * sizeof(int) + w.partial_cursor);
            w.counter_cursor += tiles;
            w.partial_cursor += need;
            fixup = true;
        }
    }
    if (split && !fixup) {
        if (!p.ep.accumulate && !caller_zeroed) {
            if (int zrc = zero_fill(p.C, p.ldc, p.N, out_rows, st)) return zrc;
        }
    }
    {
        const GemmEpilogue& e = p.ep;
        auto ok4 = [](const void* q, long ld) { return !q || (al16(q) && (ld % 4) == 0); };
        p.ep_vec = (p.N % 4 == 0) && al16(p.C) && (p.ldc % 4 == 0) && (p.c_tap_stride % 4 == 0) && ok4(e.col_scale, 0) && ok4(e.col_bias, 0) &&
                   ok4(e.residual, e.ldr) && ok4(e.samp_scale, e.samp_ld) && ok4(e.mask_mode ? e.mask_src : nullptr, e.ldm) && !p.srow && !p.scol && !e.row_scale && LDETR_EPILOGUE_VEC;
    }
    dim3 grid(cdiv(p.N, use128 ? 128 : 64), cdiv(Mmax, (use128 || use12864) ? 128 : 64), zbase * (split ? p.splitk : 1));
    int rc;
    // bf16 split path (see gemm_f32_kernel): the 128-row tiles and the narrow tile, scalar-addressed operands only; four waves per
    // block so that every wave owns >= two 32x32 accumulators (LDS operand bytes per MFMA halve with each doubling of the wave tile)
    constexpr bool split_cap = (AMODE == OP_KC_CONV || AMODE == OP_KC_CONVT || AMODE == OP_KC_DENSE || AMODE == OP_RC_DENSE || AMODE == OP_RC_PIX || BMODE == OP_KC_DENSE || BMODE == OP_RC_WT || BMODE == OP_RC_DENSE || BMODE == OP_RC_PIX || BMODE == OP_KC_WTAP);
    static const int split_tiles_env = (int)knob("SPLIT_BF16", 15);   // bit 0: 128x128, bit 1: 128x64, bit 2: 256x32, bit 3: 64x64
    constexpr int split_min_kk = 128;
    const int split_ovr = g_split_bf16_override.load(std::memory_order_relaxed);
    const int split_tiles = split_ovr >= 0 ? split_ovr : split_tiles_env;
    const int split_on = split_tiles;
    bool sp = false;
    if constexpr (split_cap) sp = split_on && (p.K / (split ? p.splitk : 1)) >= split_min_kk && fast_operands_ok<AMODE, BMODE>(p, Mmax);
    if constexpr (narrow_cap) {
        if (use_narrow) {
            if constexpr (split_cap) { if (sp && (split_on & 4)) return launch_tile_impl<256, 32, 32, AMODE, BMODE, 4, true, true>(p, dim3(1, cdiv(Mmax, 256), zbase), st); }
            return launch_tile<256, 32, 32, AMODE, BMODE, 4>(p, dim3(1, cdiv(Mmax, 256), zbase), Mmax, st);
        }
    }
    bool sp_done = false;
    rc = 0;
    if constexpr (split_cap) {
        if (sp && use128 && (split_on & 1)) { rc = launch_tile_impl<128, 128, 32, AMODE, BMODE, 4, true, true>(p, grid, st); sp_done = true; }
        else if (sp && use12864 && (split_on & 2)) { rc = launch_tile_impl<128, 64, 32, AMODE, BMODE, 4, true, true>(p, grid, st); sp_done = true; }
        else if (sp && !use128 && !use12864 && (split_on & 8)) { rc = launch_tile_impl<64, 64, 32, AMODE, BMODE, 4, true, true>(p, grid, st); sp_done = true; }
    }
    if (sp_done) {}
    else if (use128) rc = launch_tile<128, 128, T128_BK, AMODE, BMODE, T128_WAVES>(p, grid, Mmax, st);
    else if (use12864) rc = launch_tile<128, 64, T12864_BK, AMODE, BMODE, T12864_WAVES>(p, grid, Mmax, st);
    else rc = launch_tile<64, 64, 32, AMODE, BMODE>(p, grid, Mmax, st);
    if (rc) return rc;
    if (split && !fixup && !epilogue_is_linear(full)) {
        EpiParams q;
        q.ep = full; q.ep.alpha = 1.f; q.ep.row_scale = nullptr; q.ep.accumulate = 0;
        q.C = p.C; q.ldc = p.ldc; q.rows = out_rows; q.N = p.N; q.pix_per_sample = p.pix_per_sample;
        long total = out_rows * p.N;
        int g = (int)((total + 255) / 256); if (g > 4096) g = 4096; if (g < 1) g = 1;
        hipLaunchKernelGGL(gemm_epilogue_kernel, g, 256, 0, st, q);
        rc = check_launch("gemm_epilogue");
    }
    return rc;
}

// Weight gradients reduce over pixels (K = N*OH*OW, up to 2^20) into a small [Cout, taps, Cin] output: pick the tile
// first (128-wide tiles when the channel counts allow them: two accumulator chains per wave, fewer LDS reads per MFMA),
// then split K until that tile's grid fills the chip.  Mirrors the tile thresholds of launch_gemm.
// (conv_c32.hip follows the same switch: value_f32_mfma_only of the bench line and the f32-pipe halves of the parity tests cover it too)
bool engine_split_enabled() {
    static const int split_tiles_env = (int)knob("SPLIT_BF16", 15);
    const int ovr = g_split_bf16_override.load(std::memory_order_relaxed);
    return (ovr >= 0 ? ovr : split_tiles_env) != 0;
}

static int wgrad_auto_split(int M, int N, int K, int zbase) {
    // few pixels (small per-GPU batches): shorter slices keep the chip busy; the floor of a weight-gradient launch is its serial k-loop
    // (measured: 128-pixel slices help at 2 samples per GPU, 25.3 -> 24.8 ms per step, and cost 4 % at 16 per GPU: keyed on the launch's work)
    const int WMINK = ((double)M * N * K * zbase < 6e8) ? 128 : WGRAD_MIN_K;
    auto pick = [&](long tiles, long target, int min_k) {
        long s = (target + tiles - 1) / tiles, maxs = K / min_k;
        if (s > maxs) s = maxs;
        return (int)(s < 1 ? 1 : s);
    };
    if (M >= 128 && N >= 128) {
        long t = (long)cdiv(M, 128) * cdiv(N, 128) * zbase; int s = pick(t, 512, WMINK);
        if (t * s >= 384) return s;
    }
    if (M >= 128) {
        long t = (long)cdiv(M, 128) * cdiv(N, 64) * zbase; int s = pick(t, 768, WMINK);
        if (t * s >= 512) return s;
    }
    long t = (long)cdiv(M, 64) * cdiv(N, 64) * zbase;
    return pick(t, 768, WMINK);
}

// Per-sample operand scales of a weight gradient -> sample-aligned K slices (see GemmParams::samp_pix).  `want` = split the
// policy asked for; the slice count becomes nsamp * q with q slices per sample.
static void set_sample_slices(GemmParams& p, int nsamp, int pix_per_samp, int want, const float* srow, long srow_ld,
                              const float* scol, long scol_ld) {
    int q = (want + nsamp / 2) / nsamp;
    if (q < 1) q = 1;
    while (q > 1 && pix_per_samp / q < 64) q--;
    p.samp_pix = pix_per_samp; p.samp_q = q;
    p.srow = srow; p.srow_ld = srow_ld; p.scol = scol; p.scol_ld = scol_ld;
    p.splitk = nsamp * q;
    if (p.splitk > 1) p.ep.accumulate = 0;   // atomics onto the (zeroed or existing) buffer
}

static void fill_epilogue(GemmEpilogue& ep, const ldetr_epilogue* e) {
    memset(&ep, 0, sizeof(ep));
    ep.out_scale = 1.f; ep.act_gain = 1.f; ep.alpha = 1.f;
    if (!e) return;
    ep.alpha = e->alpha;
    ep.col_scale = e->col_scale; ep.col_bias = e->col_bias;
    ep.samp_scale = e->samp_scale; ep.samp_ld = e->samp_ld;
    ep.residual = e->residual; ep.ldr = e->ldr;
    ep.act = e->act; ep.act_alpha = e->act_alpha; ep.act_gain = e->act_gain;
    ep.mask_src = e->mask_src; ep.ldm = e->ldm; ep.mask_mode = e->mask_mode;
    ep.out_scale = e->out_scale; ep.p_drop = e->p_drop; ep.seed = e->seed; ep.seed_ptr = (const unsigned long long*)e->seed_ptr; ep.accumulate = e->accumulate;
    ep.a_rowsum = e->a_rowsum;
}

}  // namespace ldetr

using namespace ldetr;

// Scratch memory for the in-kernel split-K reduction on the calling thread's current device.  `ptr` must be zero-filled
// device memory that stays alive (and is used by one stream at a time); null/0 unregisters (atomic split-K path).
extern "C" int ldetr_debug_trace_tiles(int64_t* buffer) {
    g_trace_buffer = reinterpret_cast<long long*>(buffer);
    return LDETR_OK;
}

extern "C" int ldetr_engine_launch_counts(int64_t* f32_pipe, int64_t* bf16_split_pipe) {
    if (f32_pipe) *f32_pipe = t_launches_f32;
    if (bf16_split_pipe) *bf16_split_pipe = t_launches_split;
    return LDETR_OK;
}

extern "C" int ldetr_set_split_bf16(int tiles) {
    const int prev = g_split_bf16_override.exchange(tiles);
    return prev;
}

extern "C" int ldetr_set_workspace(void* ptr, int64_t bytes) {
    int dev = 0;
    LDETR_CHECK(hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64, "set_workspace: no current device");
    LDETR_CHECK(bytes >= 0, "set_workspace: negative size");
    LDETR_CHECK(!ptr || (((uintptr_t)ptr) & 15) == 0, "set_workspace: pointer must be 16-byte aligned");
    LDETR_CHECK(!ptr || bytes > (int64_t)(WS_COUNTERS * sizeof(int)), "set_workspace: too small");
    g_workspace[dev].ptr = ptr; g_workspace[dev].bytes = ptr ? (size_t)bytes : 0;
    g_workspace[dev].counter_cursor = 0; g_workspace[dev].partial_cursor = 0;
    return LDETR_OK;
}

// ---------------------------------------------------------------------------------------------
// Dense GEMM.  C[M,N] = op(A) * op(B), row-major C with leading dimension ldc.
//   ta == 0: A is [M,K] row-major (lda);  ta == 1: A is stored [K,M] (lda)
//   tb == 0: B is stored [N,K] row-major (ldb)  — i.e. nn.Linear weight layout [out,in];
//   tb == 1: B is stored [K,N] (ldb)
extern "C" int ldetr_gemm_f32(const float* A, int64_t lda, int ta, const float* B, int64_t ldb, int tb,
                              float* C, int64_t ldc, int M, int N, int K, int splitk,
                              const ldetr_epilogue* ep, int pix_per_sample, void* stream) {
    LDETR_CHECK(A && B && C, "gemm: null pointer");
    LDETR_CHECK(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
    if (M == 0 || N == 0) return LDETR_OK;
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    p.A.p = A; p.A.ld = lda; p.B.p = B; p.B.ld = ldb;
    p.A.vec = al16(A) && (lda % 4 == 0) && (ta ? (M % 4 == 0) : (K % 4 == 0));
    p.B.vec = al16(B) && (ldb % 4 == 0) && (tb ? (N % 4 == 0) : (K % 4 == 0));
    p.M = M; p.N = N; p.K = K; p.C = C; p.ldc = ldc;
    p.zmode = 0; p.splitk = splitk < 1 ? 1 : splitk; p.pstep = 1; p.nsamp = 1;
    p.pix_per_sample = pix_per_sample;
    fill_epilogue(p.ep, ep);
    LDETR_CHECK(!(p.splitk > 1 && p.ep.accumulate && !epilogue_is_linear(p.ep)), "gemm: split-K + accumulate needs a linear epilogue");
    hipStream_t st = (hipStream_t)stream;
    const bool auto_split = (splitk == 0);   // splitk: 0 = let the launch policy decide, 1 = never split, >1 = explicit
    if (p.ep.a_rowsum) {
        LDETR_CHECK(ta == 1 && lda == M, "gemm: a_rowsum needs ta == 1 and a packed A (lda == M)");
        const bool small = auto_split && (long)cdiv(M, 64) * cdiv(N, 64) < SMALL_GEMM_TILES && (long)M * N * K <= SMALL_GEMM_MNK && tb;
        if (!small) {   // not the kernel that folds the row sums in: one column-sum pass over A = [K, M]
            int rc = ldetr_colsum_f32(A, p.ep.a_rowsum, 1, K, M, stream);
            if (rc) return rc;
            p.ep.a_rowsum = nullptr;
        }
    }
    if (auto_split && (long)cdiv(M, 64) * cdiv(N, 64) < SMALL_GEMM_TILES && (long)M * N * K <= SMALL_GEMM_MNK) {
        if (!ta && !tb) return launch_small<0, 0>(p, st);
        if (!ta && tb) return launch_small<0, 1>(p, st);
        if (ta && tb) return launch_small<1, 1>(p, st);
    }
    if (!ta && !tb) return launch_gemm<OP_KC_DENSE, OP_KC_DENSE>(p, M, M, 1, auto_split, false, st);
    if (!ta && tb) return launch_gemm<OP_KC_DENSE, OP_RC_DENSE>(p, M, M, 1, auto_split, false, st);
    if (ta && tb) return launch_gemm<OP_RC_DENSE, OP_RC_DENSE>(p, M, M, 1, auto_split, false, st);
    set_error("gemm: (ta=1, tb=0) is not instantiated");
    return LDETR_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// Two dense GEMMs in one call (the data gradient dX = dY W and the weight gradient dW += dY^T X of a linear layer).  When both are
// plain small-tile problems (the usual case on the transformers' token counts) they run as ONE launch of gemm_small_pair_kernel;
// anything else falls back to two ldetr_gemm_f32 calls in order.  Same semantics either way.
static bool small_class(const ldetr_gemm_desc& g) {
    if (g.splitk != 0 || g.M <= 0 || g.N <= 0 || g.K <= 0) return false;
    return (long)cdiv(g.M, 64) * cdiv(g.N, 64) < SMALL_GEMM_TILES && (long)g.M * g.N * g.K <= SMALL_GEMM_MNK;
}

static void fill_dense(GemmParams& p, const ldetr_gemm_desc& g) {
    memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    p.A.p = g.A; p.A.ld = g.lda; p.B.p = g.B; p.B.ld = g.ldb;
    p.A.vec = al16(g.A) && (g.lda % 4 == 0) && (g.ta ? (g.M % 4 == 0) : (g.K % 4 == 0));
    p.B.vec = al16(g.B) && (g.ldb % 4 == 0) && (g.tb ? (g.N % 4 == 0) : (g.K % 4 == 0));
    p.M = g.M; p.N = g.N; p.K = g.K; p.C = g.C; p.ldc = g.ldc;
    p.zmode = 0; p.splitk = 1; p.pstep = 1; p.nsamp = 1; p.pix_per_sample = g.pix_per_sample;
    fill_epilogue(p.ep, g.ep);
}

static bool pair_single_launch(const ldetr_gemm_desc* g0, const ldetr_gemm_desc* g1) {
    static const int pair_on = (int)knob("GEMM_PAIR", 1);
    // the instantiated pairings: NN + TN (a linear layer's data + weight gradient) and TN + TN (the two weight gradients of the
    // feed-forward block, hip/ffn.py); a row-sum output needs a transposed, packed A
    const bool nn_tn = g0->ta == 0 && g0->tb == 1 && g1->ta == 1 && g1->tb == 1, tn_tn = g0->ta == 1 && g0->tb == 1 && g1->ta == 1 && g1->tb == 1;
    return pair_on && (nn_tn || tn_tn) && g0->A && g0->B && g0->C && g1->A && g1->B && g1->C && small_class(*g0) && small_class(*g1) &&
           (!g1->ep || !g1->ep->a_rowsum || g1->lda == g1->M) && (!(g0->ep && g0->ep->a_rowsum) || (tn_tn && g0->lda == g0->M));
}

extern "C" int ldetr_gemm_pair_is_single_launch(const ldetr_gemm_desc* g0, const ldetr_gemm_desc* g1) {
    return (g0 && g1 && pair_single_launch(g0, g1)) ? 1 : 0;
}

extern "C" int ldetr_gemm_pair_f32(const ldetr_gemm_desc* g0, const ldetr_gemm_desc* g1, void* stream) {
    LDETR_CHECK(g0 && g1, "gemm_pair: null descriptor");
    if (pair_single_launch(g0, g1)) {
        GemmParams p0, p1;
        fill_dense(p0, *g0); fill_dense(p1, *g1);
        const int gx0 = cdiv(p0.N, 32), nt0 = gx0 * cdiv(p0.M, 32), gx1 = cdiv(p1.N, 32), nt1 = gx1 * cdiv(p1.M, 32);
        // one block size for both: 8 waves only when both problems would take them on their own
        auto wants8 = [](long blocks, int sk, int K) { return blocks * sk <= 256 && (K + sk - 1) / sk >= 512; };
        if (g0->ta == 1) {     // TN + TN
            const int sk0 = plan_small_split(p0, nt0), sk1 = plan_small_split(p1, nt1);
            const bool w8 = wants8(nt0, sk0, p0.K) && wants8(nt1, sk1, p1.K);
            hipStream_t st = (hipStream_t)stream;
            const dim3 grid((unsigned)(nt0 * sk0 + nt1 * sk1));
            if (small_fast_ok(p0, 1, 1) && small_fast_ok(p1, 1, 1)) {
                if (w8) hipLaunchKernelGGL((gemm_small_pair_kernel<1, 1, 1, 1, 8, true>), grid, 512, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
                else hipLaunchKernelGGL((gemm_small_pair_kernel<1, 1, 1, 1, 4, true>), grid, 256, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
            } else if (w8) hipLaunchKernelGGL((gemm_small_pair_kernel<1, 1, 1, 1, 8>), grid, 512, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
            else hipLaunchKernelGGL((gemm_small_pair_kernel<1, 1, 1, 1, 4>), grid, 256, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
            t_launches_f32++;
            note_engine_launch(3, 32, 32, 0, w8 ? 8 : 4, (small_fast_ok(p0, 1, 1) && small_fast_ok(p1, 1, 1)) ? 1 : 0, 0, sk0 * 256 + sk1, (long)grid.x, 3 * 4 + 3);
            return check_launch("gemm_small_pair");
        }
        if (!p0.ep.a_rowsum) {
            const int sk0 = plan_small_split(p0, nt0), sk1 = plan_small_split(p1, nt1);
            const bool w8 = wants8(nt0, sk0, p0.K) && wants8(nt1, sk1, p1.K);
            hipStream_t st = (hipStream_t)stream;
            const dim3 grid((unsigned)(nt0 * sk0 + nt1 * sk1));
            if (small_fast_ok(p0, 0, 1) && small_fast_ok(p1, 1, 1)) {
                if (w8) hipLaunchKernelGGL((gemm_small_pair_kernel<0, 1, 1, 1, 8, true>), grid, 512, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
                else hipLaunchKernelGGL((gemm_small_pair_kernel<0, 1, 1, 1, 4, true>), grid, 256, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
            } else if (w8) hipLaunchKernelGGL((gemm_small_pair_kernel<0, 1, 1, 1, 8>), grid, 512, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
            else hipLaunchKernelGGL((gemm_small_pair_kernel<0, 1, 1, 1, 4>), grid, 256, 0, st, p0, p1, gx0, nt0, sk0, gx1, nt1, sk1);
            t_launches_f32++;
            note_engine_launch(3, 32, 32, 0, w8 ? 8 : 4, (small_fast_ok(p0, 0, 1) && small_fast_ok(p1, 1, 1)) ? 1 : 0, 0, sk0 * 256 + sk1, (long)grid.x, 1 * 4 + 3);
            return check_launch("gemm_small_pair");
        }
    }
    int rc = ldetr_gemm_f32(g0->A, g0->lda, g0->ta, g0->B, g0->ldb, g0->tb, g0->C, g0->ldc, g0->M, g0->N, g0->K, g0->splitk, g0->ep, g0->pix_per_sample, stream);
    if (rc) return rc;
    return ldetr_gemm_f32(g1->A, g1->lda, g1->ta, g1->B, g1->ldb, g1->tb, g1->C, g1->ldc, g1->M, g1->N, g1->K, g1->splitk, g1->ep, g1->pix_per_sample, stream);
}

static void set_conv_src(Operand& o, const float* x, const ldetr_tensor4* t) {
    o.p = x; o.SH = t->H; o.SW = t->W; o.sn = t->sn; o.sh = t->sh; o.sw = t->sw; o.sc = t->sc;
}

// ---------------------------------------------------------------------------------------------
// conv2d forward (correlation, as F.conv2d):  y[n,oh,ow,co] = sum x[n, oh*s-p+kh, ow*s-p+kw, ci] * w[co,kh,kw,ci]
// x: any strides (NHWC fast path when sc == 1 and C % 4 == 0); w: OHWI contiguous; y: NHWC rows of ldy.
extern "C" int ldetr_conv2d_fwd_f32(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW,
                                    int stride, int pad, float* y, int64_t ldy, int OH, int OW,
                                    const float* in_scale, int64_t in_scale_ld,
                                    const ldetr_epilogue* ep, void* stream) {
    LDETR_CHECK(x && w && y && xt, "conv2d_fwd: null pointer");
    int eOH = (xt->H + 2 * pad - KH) / stride + 1, eOW = (xt->W + 2 * pad - KW) / stride + 1;
    LDETR_CHECK(eOH == OH && eOW == OW, "conv2d_fwd: output size mismatch (expected %dx%d)", eOH, eOW);
    {   // the ResNet stem (3 -> 64 channels, 7x7 / 2) has its own LDS-resident kernel (csrc/stem_conv.hip)
        const int took = try_launch_stem_conv(x, xt, w, Cout, KH, KW, stride, pad, y, ldy, OH, OW, in_scale, ep, (hipStream_t)stream);
        if (took >= 0) return took;
    }
    {   // 32 -> 32 channels, 3x3, on a large grid (the 256x256 StyleGAN2 layers): filter bank in registers, no LDS (csrc/conv_c32.hip)
        const int took = try_launch_conv_c32(x, xt, w, Cout, KH, KW, stride, pad, y, ldy, OH, OW, in_scale, in_scale_ld, ep, 0, (hipStream_t)stream);
        if (took > 0) t_launches_f32++;
        if (took != 0) return took > 0 ? LDETR_OK : LDETR_ERR_LAUNCH;
    }
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    set_conv_src(p.A, x, xt);
    p.A.DH = OH; p.A.DW = OW; p.A.C = xt->C; p.A.stride = stride; p.A.pad = pad; p.A.KH = KH; p.A.KW = KW;
    p.A.scale = in_scale; p.A.scale_ld = in_scale_ld;
    p.A.vec = (xt->sc == 1) && (xt->C % 4 == 0) && al16(x) && (xt->sn % 4 == 0) && (xt->sh % 4 == 0) && (xt->sw % 4 == 0) &&
              (!in_scale || (al16(in_scale) && in_scale_ld % 4 == 0));
    long K = (long)KH * KW * xt->C;
    p.B.p = w; p.B.ld = K; p.B.vec = al16(w) && (K % 4 == 0);
    p.M = xt->N * OH * OW; p.N = Cout; p.K = (int)K; p.C = y; p.ldc = ldy;
    p.zmode = 0; p.splitk = 1; p.pstep = 1; p.nsamp = xt->N; p.pix_per_sample = OH * OW;
    fill_epilogue(p.ep, ep);
    hipStream_t st = (hipStream_t)stream;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && p.A.vec && !in_scale && xt->sw == xt->C && xt->sh == (long)xt->W * xt->C &&
        xt->sn == (long)xt->H * xt->W * xt->C) {
        p.A.ld = xt->C;  // pure GEMM view of a packed NHWC tensor
        // few output tiles (the trunk's 1x1 convolutions at 2-4 samples per GPU: 32..128 tiles of 64x64 on 256 CUs): the latency-bound
        // small-tile kernel, like every other dense contraction of that class (ldetr_gemm_f32); same epilogue
        if (!p.ep.samp_scale && (long)cdiv(p.M, 64) * cdiv(p.N, 64) < SMALL_GEMM_TILES && (long)p.M * p.N * p.K <= SMALL_GEMM_MNK)
            return launch_small<0, 0>(p, st);
        return launch_gemm<OP_KC_DENSE, OP_KC_DENSE>(p, p.M, p.M, 1, true, false, st);
    }
    return launch_gemm<OP_KC_CONV, OP_KC_DENSE>(p, p.M, p.M, 1, true, false, st);
}

// conv2d backward-data: dx[n,ih,iw,ci] = sum_{co,kh,kw} dy[n,oh,ow,co] * w[co,kh,kw,ci],  ih = oh*s - p + kh.
// Stride-s problems are decomposed into s*s output-parity classes so no multiply-by-zero work is issued.
extern "C" int ldetr_conv2d_bwd_data_f32(const float* dy, const ldetr_tensor4* dyt, const float* w, int Cin, int KH, int KW,
                                         int stride, int pad, float* dx, int64_t lddx, int IH, int IW,
                                         const float* dy_scale, int64_t dy_scale_ld,
                                         const ldetr_epilogue* ep, void* stream) {
    LDETR_CHECK(dy && w && dx && dyt, "conv2d_bwd_data: null pointer");
    LDETR_CHECK(dyt->sc == 1 && dyt->C % 4 == 0 && Cin % 4 == 0, "conv2d_bwd_data: channels must be NHWC-contiguous multiples of 4");
    {   // the same register-resident kernel with the filter bank read transposed and flipped
        const int took = try_launch_conv_c32(dy, dyt, w, Cin, KH, KW, stride, pad, dx, lddx, IH, IW, dy_scale, dy_scale_ld, ep, 1, (hipStream_t)stream);
        if (took > 0) t_launches_f32++;
        if (took != 0) return took > 0 ? LDETR_OK : LDETR_ERR_LAUNCH;
    }
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    set_conv_src(p.A, dy, dyt);
    p.A.DH = IH; p.A.DW = IW; p.A.C = dyt->C; p.A.stride = stride; p.A.pad = pad; p.A.KH = KH; p.A.KW = KW;
    p.A.scale = dy_scale; p.A.scale_ld = dy_scale_ld;
    p.A.vec = al16(dy) && (dyt->sn % 4 == 0) && (dyt->sh % 4 == 0) && (dyt->sw % 4 == 0) &&
              (!dy_scale || (al16(dy_scale) && dy_scale_ld % 4 == 0));
    LDETR_CHECK(p.A.vec, "conv2d_bwd_data: dy must be 16-byte aligned NHWC");
    p.B.p = w; p.B.ld = (long)KH * KW * Cin; p.B.C = dyt->C; p.B.Cr = Cin; p.B.KH = KH; p.B.KW = KW;
    p.B.vec = al16(w);
    p.N = Cin; p.C = dx; p.ldc = lddx;
    p.zmode = 1; p.splitk = 1; p.pstep = stride; p.nsamp = dyt->N; p.pix_per_sample = IH * IW;
    p.M = dyt->N * IH * IW; p.K = KH * KW * dyt->C;
    fill_epilogue(p.ep, ep);
    {   // 1x1 / stride 1 on few output tiles (the trunk at 2-4 samples per GPU): dx[M, Cin] = (dy * scale)[M, Cout] . w[Cout, Cin] as a
        // dense small-tile contraction over Cout, like the forward (ldetr_conv2d_fwd_f32)
        if (KH == 1 && KW == 1 && stride == 1 && pad == 0 && dyt->sw == dyt->C && dyt->sh == (long)dyt->W * dyt->C &&
            dyt->sn == (long)dyt->H * dyt->W * dyt->C && (!dy_scale || dy_scale_ld == 0) && dyt->C % 4 == 0 && !p.ep.samp_scale &&
            (long)cdiv(p.M, 64) * cdiv(p.N, 64) < SMALL_GEMM_TILES && (long)p.M * p.N * p.K <= SMALL_GEMM_MNK) {
            GemmParams g = p;
            g.zmode = 0; g.pstep = 1; g.A.ld = dyt->C; g.B.ld = Cin; g.B.vec = al16(w) && (Cin % 4 == 0);
            return launch_small<0, 1>(g, (hipStream_t)stream);
        }
    }
    int Mmax = dyt->N * cdiv(IH, stride) * cdiv(IW, stride);
    return launch_gemm<OP_KC_CONVT, OP_RC_WT>(p, Mmax, (long)p.M, stride * stride, true, false, (hipStream_t)stream);
}

// conv2d backward-weight: dw[co,kh,kw,ci] = sum_{n,oh,ow} dy[n,oh,ow,co] * x[n, oh*s-p+kh, ow*s-p+kw, ci].
// One GEMM per tap (grid.z = taps x split-K), fp32 atomic accumulation into a zeroed dw.
extern "C" int ldetr_conv2d_bwd_weight_f32(const float* x, const ldetr_tensor4* xt, const float* dy, const ldetr_tensor4* dyt,
                                           float* dw, int KH, int KW, int stride, int pad, int splitk,
                                           const float* x_scale, int64_t x_scale_ld,
                                           const float* dy_scale, int64_t dy_scale_ld, int accumulate, void* stream) {
    LDETR_CHECK(x && dy && dw && xt && dyt, "conv2d_bwd_weight: null pointer");
    LDETR_CHECK(dyt->sc == 1, "conv2d_bwd_weight: dy must be NHWC");
    int Cin = xt->C, Cout = dyt->C, OH = dyt->H, OW = dyt->W;
    hipStream_t st = (hipStream_t)stream;
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    int Kpix = dyt->N * OH * OW;
    const bool gather = (xt->sc != 1 || Cin % 4 != 0);
    if (splitk < 1)   // 0 = automatic
        splitk = gather ? wgrad_auto_split(Cout, KH * KW * Cin, Kpix, 1) : wgrad_auto_split(Cout, Cin, Kpix, KH * KW);
    fill_epilogue(p.ep, nullptr);
    p.splitk = splitk; p.pstep = 1; p.nsamp = xt->N; p.pix_per_sample = 0;
    long wsz = (long)Cout * KH * KW * Cin;
    if (!accumulate) { if (int zrc = zero_fill(dw, wsz, wsz, 1, st)) return zrc; }
    if (!gather) {   // 32 -> 32 channels on a large grid: the LDS-free operand-streaming kernel (wgrad_smallc.hip)
        const int took = try_launch_wgrad_smallc(x, xt, dy, dyt, dw, KH, KW, stride, pad, x_scale, x_scale_ld, dy_scale, dy_scale_ld, st);
        if (took > 0) t_launches_f32++;
        if (took != 0) return took > 0 ? LDETR_OK : LDETR_ERR_LAUNCH;
    }
    if (splitk == 1 && accumulate) p.ep.accumulate = 1;
    if (gather) {
        // 3-channel (or strided-channel) input, e.g. the ResNet stem on an NCHW image: a single GEMM with
        // A = dy viewed [k = pixel][m = co] and B = scalar im2col gather [k = pixel][n = (tap, c)].
        LDETR_CHECK(dyt->sw == Cout && dyt->sh == (long)OW * Cout && dyt->sn == (long)OH * OW * Cout, "conv2d_bwd_weight: dy must be packed NHWC");
        LDETR_CHECK(!x_scale && !dy_scale, "conv2d_bwd_weight: scales unsupported on the scalar-gather path");
        p.A.p = dy; p.A.ld = Cout; p.A.vec = al16(dy) && (Cout % 4 == 0);
        set_conv_src(p.B, x, xt);
        p.B.DH = OH; p.B.DW = OW; p.B.C = Cin; p.B.stride = stride; p.B.pad = pad; p.B.KH = KH; p.B.KW = KW;
        p.M = Cout; p.N = KH * KW * Cin; p.K = Kpix; p.C = dw; p.ldc = (long)KH * KW * Cin; p.zmode = 0;
        return launch_gemm<OP_RC_DENSE, OP_RC_CONVK>(p, p.M, p.M, 1, false, true, st);
    }
    // A = dy viewed as [k = pixel][m = co]; B = x gathered per tap [k = pixel][n = ci]
    if (dy_scale && dy_scale_ld == 0) { p.ep.row_scale = dy_scale; dy_scale = nullptr; }   // one scale per co for all samples: out of the k-loop
    p.M = Cout; p.N = Cin; p.K = Kpix; p.C = dw; p.ldc = (long)KH * KW * Cin;
    p.zmode = 2; p.ntaps = KH * KW; p.c_tap_stride = Cin;
    if ((dy_scale || x_scale) && (!dy_scale || dy_scale_ld != 0) && (!x_scale || x_scale_ld != 0) && OH * OW >= 64) {
        // per-sample scales (style modulation of x, demodulation of dy): factor them out of the k-loop
        set_sample_slices(p, dyt->N, OH * OW, splitk, dy_scale, dy_scale_ld, x_scale, x_scale_ld);
        if (p.splitk == 1 && accumulate) p.ep.accumulate = 1;
        dy_scale = nullptr; x_scale = nullptr;
    }
    const bool dy_packed = !dy_scale && dyt->sw == Cout && dyt->sh == (long)OW * Cout && dyt->sn == (long)OH * OW * Cout;
    if (dy_packed) {
        // packed NHWC dy without per-sample scale is a plain [pixels, Cout] matrix: no pixel decode for that operand;
        // a 1x1 / stride 1 conv over packed x is a plain transposed GEMM altogether
        p.A.p = dy; p.A.ld = Cout; p.A.vec = al16(dy) && (Cout % 4 == 0);
        const bool x_packed = !x_scale && KH == 1 && KW == 1 && stride == 1 && pad == 0 && xt->sw == Cin && xt->sh == (long)xt->W * Cin &&
                              xt->sn == (long)xt->H * xt->W * Cin;
        if (x_packed) {
            p.B.p = x; p.B.ld = Cin; p.B.vec = al16(x);
            // few output tiles and a short reduction (the trunk's 1x1 weight gradients at 2-4 samples per GPU): the small-tile kernel
            // (dw is zero or holds the running gradient at this point: accumulate; its own in-kernel split-K when the reduction is long)
            if ((long)cdiv(p.M, 64) * cdiv(p.N, 64) < SMALL_GEMM_TILES && (long)p.M * p.N * p.K <= SMALL_GEMM_MNK && Kpix <= 8192) {
                p.zmode = 0; p.ntaps = 1; p.c_tap_stride = 0; p.splitk = 1; p.ep.accumulate = 1;
                return launch_small<1, 1>(p, st);
            }
            return launch_gemm<OP_RC_DENSE, OP_RC_DENSE>(p, p.M, p.M, 1, false, true, st);
        }
        set_conv_src(p.B, x, xt);
        p.B.DH = OH; p.B.DW = OW; p.B.stride = stride; p.B.pad = pad; p.B.KH = KH; p.B.KW = KW; p.B.tapped = 1;
        p.B.scale = x_scale; p.B.scale_ld = x_scale_ld;
        p.B.vec = al16(x) && (xt->sn % 4 == 0) && (xt->sh % 4 == 0) && (xt->sw % 4 == 0) &&
                  (!x_scale || (al16(x_scale) && x_scale_ld % 4 == 0));
        return launch_gemm<OP_RC_DENSE, OP_RC_PIX>(p, p.M, p.M, p.ntaps, false, true, st);
    }
    set_conv_src(p.A, dy, dyt);
    p.A.DH = OH; p.A.DW = OW; p.A.stride = 1; p.A.pad = 0; p.A.tapped = 0;
    p.A.scale = dy_scale; p.A.scale_ld = dy_scale_ld;
    p.A.vec = al16(dy) && (Cout % 4 == 0) && (dyt->sn % 4 == 0) && (dyt->sh % 4 == 0) && (dyt->sw % 4 == 0) &&
              (!dy_scale || (al16(dy_scale) && dy_scale_ld % 4 == 0));
    set_conv_src(p.B, x, xt);
    p.B.DH = OH; p.B.DW = OW; p.B.stride = stride; p.B.pad = pad; p.B.KH = KH; p.B.KW = KW; p.B.tapped = 1;
    p.B.scale = x_scale; p.B.scale_ld = x_scale_ld;
    p.B.vec = al16(x) && (xt->sn % 4 == 0) && (xt->sh % 4 == 0) && (xt->sw % 4 == 0) &&
              (!x_scale || (al16(x_scale) && x_scale_ld % 4 == 0));
    p.M = Cout; p.N = Cin; p.K = Kpix; p.C = dw; p.ldc = (long)KH * KW * Cin;
    p.zmode = 2; p.ntaps = KH * KW; p.c_tap_stride = Cin;
    return launch_gemm<OP_RC_PIX, OP_RC_PIX>(p, p.M, p.M, p.ntaps, false, true, st);
}

// conv_transpose2d forward (as F.conv_transpose2d with weight given as the *un-transposed* OHWI tensor
// w[co,kh,kw,ci]):  y[n,oh,ow,co] = sum_{ci,kh,kw} x[n,ih,iw,ci] * w[co,kh,kw,ci],  oh = ih*s + kh - p.
extern "C" int ldetr_conv_transpose2d_fwd_f32(const float* x, const ldetr_tensor4* xt, const float* w, int Cout, int KH, int KW,
                                              int stride, int pad, float* y, int64_t ldy, int OH, int OW,
                                              const float* in_scale, int64_t in_scale_ld,
                                              const ldetr_epilogue* ep, void* stream) {
    LDETR_CHECK(x && w && y && xt, "conv_transpose2d_fwd: null pointer");
    LDETR_CHECK(xt->sc == 1 && xt->C % 4 == 0, "conv_transpose2d_fwd: x must be NHWC with C % 4 == 0");
    int eOH = (xt->H - 1) * stride - 2 * pad + KH, eOW = (xt->W - 1) * stride - 2 * pad + KW;
    LDETR_CHECK(eOH == OH && eOW == OW, "conv_transpose2d_fwd: output size mismatch (expected %dx%d)", eOH, eOW);
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    set_conv_src(p.A, x, xt);
    p.A.DH = OH; p.A.DW = OW; p.A.C = xt->C; p.A.stride = stride; p.A.pad = pad; p.A.KH = KH; p.A.KW = KW;
    p.A.scale = in_scale; p.A.scale_ld = in_scale_ld;
    p.A.vec = al16(x) && (xt->sn % 4 == 0) && (xt->sh % 4 == 0) && (xt->sw % 4 == 0) &&
              (!in_scale || (al16(in_scale) && in_scale_ld % 4 == 0));
    LDETR_CHECK(p.A.vec, "conv_transpose2d_fwd: x must be 16-byte aligned NHWC");
    p.B.p = w; p.B.ld = (long)KH * KW * xt->C; p.B.C = xt->C; p.B.KH = KH; p.B.KW = KW; p.B.vec = al16(w);
    LDETR_CHECK(p.B.vec, "conv_transpose2d_fwd: w must be 16-byte aligned");
    p.N = Cout; p.C = y; p.ldc = ldy;
    p.zmode = 1; p.splitk = 1; p.pstep = stride; p.nsamp = xt->N; p.pix_per_sample = OH * OW;
    p.M = xt->N * OH * OW; p.K = KH * KW * xt->C;
    fill_epilogue(p.ep, ep);
    int Mmax = xt->N * cdiv(OH, stride) * cdiv(OW, stride);
    return launch_gemm<OP_KC_CONVT, OP_KC_WTAP>(p, Mmax, (long)p.M, stride * stride, false, false, (hipStream_t)stream);
}

// conv_transpose2d backward-data: dx[n,ih,iw,ci] = sum_{co,kh,kw} dy[n, ih*s+kh-p, iw*s+kw-p, co] * w[co,kh,kw,ci]
extern "C" int ldetr_conv_transpose2d_bwd_data_f32(const float* dy, const ldetr_tensor4* dyt, const float* w, int Cin, int KH, int KW,
                                                   int stride, int pad, float* dx, int64_t lddx, int IH, int IW,
                                                   const float* dy_scale, int64_t dy_scale_ld,
                                                   const ldetr_epilogue* ep, void* stream) {
    LDETR_CHECK(dy && w && dx && dyt, "conv_transpose2d_bwd_data: null pointer");
    LDETR_CHECK(dyt->sc == 1 && dyt->C % 4 == 0 && Cin % 4 == 0, "conv_transpose2d_bwd_data: channels must be NHWC multiples of 4");
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    set_conv_src(p.A, dy, dyt);
    p.A.DH = IH; p.A.DW = IW; p.A.C = dyt->C; p.A.stride = stride; p.A.pad = pad; p.A.KH = KH; p.A.KW = KW;
    p.A.scale = dy_scale; p.A.scale_ld = dy_scale_ld;
    p.A.vec = al16(dy) && (dyt->sn % 4 == 0) && (dyt->sh % 4 == 0) && (dyt->sw % 4 == 0) &&
              (!dy_scale || (al16(dy_scale) && dy_scale_ld % 4 == 0));
    LDETR_CHECK(p.A.vec, "conv_transpose2d_bwd_data: dy must be 16-byte aligned NHWC");
    p.B.p = w; p.B.ld = (long)KH * KW * Cin; p.B.C = dyt->C; p.B.Cr = Cin; p.B.KH = KH; p.B.KW = KW; p.B.vec = al16(w);
    p.M = dyt->N * IH * IW; p.N = Cin; p.K = KH * KW * dyt->C; p.C = dx; p.ldc = lddx;
    p.zmode = 0; p.splitk = 1; p.pstep = 1; p.nsamp = dyt->N; p.pix_per_sample = IH * IW;
    fill_epilogue(p.ep, ep);
    return launch_gemm<OP_KC_CONV, OP_RC_WT>(p, p.M, p.M, 1, true, false, (hipStream_t)stream);
}

// conv_transpose2d backward-weight: dw[co,kh,kw,ci] = sum_{n,ih,iw} dy[n, ih*s+kh-p, iw*s+kw-p, co] * x[n,ih,iw,ci]
extern "C" int ldetr_conv_transpose2d_bwd_weight_f32(const float* x, const ldetr_tensor4* xt, const float* dy, const ldetr_tensor4* dyt,
                                                     float* dw, int KH, int KW, int stride, int pad, int splitk,
                                                     const float* x_scale, int64_t x_scale_ld,
                                                     const float* dy_scale, int64_t dy_scale_ld, int accumulate, void* stream) {
    LDETR_CHECK(x && dy && dw && xt && dyt, "conv_transpose2d_bwd_weight: null pointer");
    LDETR_CHECK(dyt->sc == 1 && xt->sc == 1 && xt->C % 4 == 0 && dyt->C % 4 == 0, "conv_transpose2d_bwd_weight: NHWC, C % 4 == 0 required");
    int Cin = xt->C, Cout = dyt->C;
    hipStream_t st = (hipStream_t)stream;
    GemmParams p; memset(&p, 0, sizeof(p));
    init_operand(p.A); init_operand(p.B);
    if (splitk < 1) splitk = wgrad_auto_split(Cout, Cin, xt->N * xt->H * xt->W, KH * KW);   // 0 = automatic
    fill_epilogue(p.ep, nullptr);
    p.splitk = splitk; p.pstep = 1; p.nsamp = xt->N; p.pix_per_sample = 0;
    long wsz = (long)Cout * KH * KW * Cin;
    if (!accumulate) { if (int zrc = zero_fill(dw, wsz, wsz, 1, st)) return zrc; }
    if (splitk == 1 && accumulate) p.ep.accumulate = 1;
    if ((dy_scale || x_scale) && (!dy_scale || dy_scale_ld != 0) && (!x_scale || x_scale_ld != 0) && xt->H * xt->W >= 64) {
        set_sample_slices(p, xt->N, xt->H * xt->W, splitk, dy_scale, dy_scale_ld, x_scale, x_scale_ld);   // scales out of the k-loop
        if (p.splitk == 1 && accumulate) p.ep.accumulate = 1;
        dy_scale = nullptr; x_scale = nullptr;
    }
    // k enumerates *input* pixels (n, ih, iw); A = dy gathered at (ih*s + kh - p, iw*s + kw - p), B = x dense.
    set_conv_src(p.A, dy, dyt);
    p.A.DH = xt->H; p.A.DW = xt->W; p.A.stride = stride; p.A.pad = pad; p.A.KH = KH; p.A.KW = KW; p.A.tapped = 1;
    p.A.scale = dy_scale; p.A.scale_ld = dy_scale_ld;
    p.A.vec = al16(dy) && (dyt->sn % 4 == 0) && (dyt->sh % 4 == 0) && (dyt->sw % 4 == 0) &&
              (!dy_scale || (al16(dy_scale) && dy_scale_ld % 4 == 0));
    set_conv_src(p.B, x, xt);
    p.B.DH = xt->H; p.B.DW = xt->W; p.B.stride = 1; p.B.pad = 0; p.B.tapped = 0;
    p.B.scale = x_scale; p.B.scale_ld = x_scale_ld;
    p.B.vec = al16(x) && (xt->sn % 4 == 0) && (xt->sh % 4 == 0) && (xt->sw % 4 == 0) &&
              (!x_scale || (al16(x_scale) && x_scale_ld % 4 == 0));
    p.M = Cout; p.N = Cin; p.K = xt->N * xt->H * xt->W; p.C = dw; p.ldc = (long)KH * KW * Cin;
    p.zmode = 2; p.ntaps = KH * KW; p.c_tap_stride = Cin;
    if (!x_scale && xt->sw == Cin && xt->sh == (long)xt->W * Cin && xt->sn == (long)xt->H * xt->W * Cin) {
        p.B.p = x; p.B.ld = Cin; p.B.vec = al16(x);   // packed x: a plain [pixels, Cin] matrix
        return launch_gemm<OP_RC_PIX, OP_RC_DENSE>(p, p.M, p.M, p.ntaps, false, true, st);
    }
    return launch_gemm<OP_RC_PIX, OP_RC_PIX>(p, p.M, p.M, p.ntaps, false, true, st);
}
