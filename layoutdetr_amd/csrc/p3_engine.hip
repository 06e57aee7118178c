// Plane-format contraction engine for gfx950: the ResNet-50 trunk's convolutions on the bf16 matrix pipe with fp32-equivalent
// results and NO operand conversion inside the k-loop (training/detr_backbone.py:98-114 via torchvision's resnet50; ATen conv2d +
// its autograd in the reference).
//
// Activation / weight format "P3": an fp32 tensor [rows][C] (C % 8 == 0, channels contiguous = NHWC pixels or OHWI weights) is
// stored as its exact three-way bf16 split x = hi + mid + lo (8 + 8 + 8 significant bits; ldetr_common.hpp::split2_bf16_exact: every normal
// fp32 value, +-0, +-Inf and NaN come back bit for bit -- -0 as +0 -- and subnormals keep the bits bf16's subnormal range holds), in 48-byte
// groups of 8 channels: [rows][C/8][3 planes][8 x bf16].  6 bytes per element; a row's k-range is one contiguous segment holding
// all three planes.  Producers (this engine's epilogues, the max-pool, ldetr_p3_split_f32) write it once; consumers move it
// global -> LDS with buffer_load ... lds (16 bytes per lane, no VGPR round trip, no VALU) and feed v_mfma_f32_32x32x16_bf16 with
// ds_read_b128 fragments.  Per k16 slab the six products of weight >= 2^-18 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) go
// into the same fp32 accumulators: at least as close to the exact contraction as v_mfma_f32_32x32x2_f32 (gemm_conv.hip, SPLIT).
//
// p3_nt_kernel: C[m][n] = sum_k A(m, k) B[n][k]; A = implicit-GEMM gather of P3 pixels (k = (tap, channel), stride 1 or 2, zero
// padding = out-of-range buffer offsets), B = P3 weights, k-contiguous.  Forward convs, and data gradients through transposed /
// tap-flipped weight copies (ldetr_p3_weight_bwd).  3-stage LDS ring of BK = 32 tiles, one barrier per k-tile, counted vmcnt
// (two tiles in flight across the barrier), XOR-swizzled 64-byte LDS rows (conflict-free ds_read_b128), XCD-aware tile order,
// in-kernel split-K (partial tiles parked with agent-scope stores, last arriver reduces in slice order).
// Epilogue through LDS in row-major 8-channel groups: FrozenBN scale/shift, residual (P3 or f32), ReLU, ReLU-gradient mask from a
// P3 tensor's hi plane, outputs as P3 and / or f32.
#include <cstdlib>
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

bool splitk_ws_alloc(long tiles, size_t partial_bytes, float** ws, int** counters);   // gemm_conv.hip (ring of ldetr_set_workspace)
void note_engine_launch(bool bf16_split_pipe);                                          // gemm_conv.hip (ldetr_engine_launch_counts)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct P3Epi {
    float alpha;
    const float* col_scale;
    const float* col_bias;
    const char* res_p3;      // residual [M][N] P3
    const float* res_f32;    // residual [M][N] f32
    const char* mask_p3;     // backward: multiply by (mask > 0), mask [M][N] P3 (hi plane decides)
    int relu;
    char* out_p3;
    float* out_f32;
};

struct P3NtParams {
    const char* A;           // P3 pixels, shifted back by pad_off bytes (all tap offsets non-negative)
    const char* B;           // P3 weights [N][taps * Cin]
    unsigned a_bytes, b_bytes;   // descriptor ranges
    int M, N, Cin;
    int KH, KW, stride, pad; // taps = KH * KW <= 32
    int H, W, OH, OW;        // source grid, destination grid (rows of C enumerate (n, oy, ox))
    int out_H, out_W, out_step, out_py, out_px;   // output row -> pixel of the stored tensor: (oy*out_step + out_py, ox*out_step + out_px) in an out_H x out_W grid
    int tap_mode;            // 0: src = dst*stride - pad + tap (forward);  1: src = (dst + pad - tap) / stride over the taps of one parity class (data gradient, stride 2)
    int kh0, kw0, tstep, nty, ntx;   // tap walk: kh = kh0 + tstep*ty (ty < nty), kw likewise; weights' tap index = ty*ntx + tx in B's k order
    int nkt;                 // k-tiles in total (taps * Cin / 32)
    int splitk;
    float* ws; int* ws_count;
    int mtiles, ntiles, xm, xn;   // tile grid and the XCD array laid over it (xcd_tile)
    int nclass, nimg;        // data gradient with stride > 1: blockIdx.z enumerates the stride^2 parity classes of the destination pixels (one launch)
    P3Epi ep;
    int ngroups;             // grouped forward (2): the same geometry on a second operand set (P3Group2) as the z = 1 half of the grid
};

// Second (activations, weights, epilogue) set of a grouped forward launch: a kernel argument of its own, so that the kernels that never group
// (the paired backward) do not carry it (their register budget is their occupancy)
struct P3Group2 {
    const char* A2; const char* B2;
    P3Epi ep2;
};

// (merging the planes: hi + (mid + lo).  mid + lo is the exact remainder (<= 16 significant bits), so the outer sum is the stored value exactly;
// (hi + mid) + lo overflows for values within 2^-8 of FLT_MAX, whose hi + mid is 2^128)
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

// XCD-aware tile order.  Block b runs on XCD b % 8 and every XCD has its own 4 MiB L2, so the 8 XCDs are laid over the tile grid as an
// xm x xn array (xm * xn = 8): XCD (xi, xj) owns the m-tiles of chunk xi and the n-tiles of chunk xj, n fastest.  Its L2 then holds 1/xm of the
// A panel rows and 1/xn of the weights; through the fabric go A x xn + B x xm bytes per launch, which the host minimises (layer 4: the
// 14 MB weight image used to be fetched by all eight L2s).  Returns false for the padding blocks of a ragged partition.
__device__ __forceinline__ bool xcd_tile(int b, int mtiles, int ntiles, int xm, int xn, int& tm, int& tn) {
    const int x = b & 7, i = b >> 3, xi = x / xn, xj = x - xi * xn;
    const int mlo = xi * mtiles / xm, mhi = (xi + 1) * mtiles / xm, nlo = xj * ntiles / xn, nhi = (xj + 1) * ntiles / xn;
    const int cn = nhi - nlo;
    if (cn <= 0 || i >= (mhi - mlo) * cn) return false;
    tm = mlo + i / cn; tn = nlo + i - (i / cn) * cn;
    return true;
}

// Epilogue of one wave: its WM x WN accumulator tile goes through the wave's own LDS region (Cs, row pitch WN + 4 floats) and comes back as
// 8-channel groups per lane: scale / shift, residual (P3 or fp32), ReLU, ReLU-gradient mask from a P3 tensor's hi plane, P3 and / or fp32 stores.
// rowmap(rl) -> output row (pixel index of the stored tensor) of local row rl of this wave's tile, or -1.
template <int WM, int WN, int TM, int TN, typename RowMap>
__device__ __forceinline__ void p3_wave_epilogue(const P3Epi& ep, float* Cs, const f32x16 (&acc)[TM][TN], int lane, int nbase, int N, RowMap rowmap) {
    constexpr int CP = WN + 4;
    const int cl = lane & 31, kl = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                Cs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl) * CP + j * 32 + cl] = acc[i][j][r];
    // (each wave reads back its own region only: LDS operations of one wave execute in order, no barrier)
    constexpr int CH = WN / 8;                            // 8-channel groups per row of the wave tile
    constexpr int RPI = 64 / CH;                          // rows per iteration
    const int ch = lane % CH, rl0 = lane / CH;
    const int n = nbase + ch * 8;
    float cs[8], cb[8];
#pragma unroll
    for (int e = 0; e < 8; e++) { cs[e] = ep.alpha; cb[e] = 0.f; }
    if (n < N) {
        if (ep.col_scale) {
            const float4 s0 = *reinterpret_cast<const float4*>(ep.col_scale + n), s1 = *reinterpret_cast<const float4*>(ep.col_scale + n + 4);
            cs[0] *= s0.x; cs[1] *= s0.y; cs[2] *= s0.z; cs[3] *= s0.w; cs[4] *= s1.x; cs[5] *= s1.y; cs[6] *= s1.z; cs[7] *= s1.w;
        }
        if (ep.col_bias) {
            const float4 s0 = *reinterpret_cast<const float4*>(ep.col_bias + n), s1 = *reinterpret_cast<const float4*>(ep.col_bias + n + 4);
            cb[0] = s0.x; cb[1] = s0.y; cb[2] = s0.z; cb[3] = s0.w; cb[4] = s1.x; cb[5] = s1.y; cb[6] = s1.z; cb[7] = s1.w;
        }
    }
#pragma unroll 2
    for (int it = 0; it < WM / RPI; it++) {
        const int rl = it * RPI + rl0;
        const long orow = rowmap(rl);
        if (orow < 0 || n >= N) continue;
        const float4 v0 = *reinterpret_cast<const float4*>(Cs + rl * CP + ch * 8), v1 = *reinterpret_cast<const float4*>(Cs + rl * CP + ch * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = v[e] * cs[e] + cb[e];
        const long goff = (orow * N + n) * 6;             // byte offset of this 8-channel group in a P3 [rows][N] tensor
        if (ep.res_p3) {
            const u32x4 h = *reinterpret_cast<const u32x4*>(ep.res_p3 + goff), md = *reinterpret_cast<const u32x4*>(ep.res_p3 + goff + 16),
                        lo = *reinterpret_cast<const u32x4*>(ep.res_p3 + goff + 32);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                v[2 * e] += bf_lo(h[e]) + (bf_lo(md[e]) + bf_lo(lo[e]));
                v[2 * e + 1] += bf_hi(h[e]) + (bf_hi(md[e]) + bf_hi(lo[e]));
            }
        }
        if (ep.res_f32) {
            const float4 r0 = *reinterpret_cast<const float4*>(ep.res_f32 + orow * N + n), r1 = *reinterpret_cast<const float4*>(ep.res_f32 + orow * N + n + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
        if (ep.relu) {   // ATen's threshold: relu(NaN) = NaN (fmaxf would return 0)
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = v[e] < 0.f ? 0.f : v[e];
        }
        if (ep.mask_p3) {   // threshold_backward: the gradient passes where mask > 0, which is false for NaN (hi plane: +0 < bits <= +Inf)
            const u32x4 h = *reinterpret_cast<const u32x4*>(ep.mask_p3 + goff);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if ((h[e] << 16) - 1u >= 0x7f800000u) v[2 * e] = 0.f;
                if ((h[e] & 0xffff0000u) - 1u >= 0x7f800000u) v[2 * e + 1] = 0.f;
            }
        }
        if (ep.out_f32) {
            *reinterpret_cast<float4*>(ep.out_f32 + orow * N + n) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(ep.out_f32 + orow * N + n + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        if (ep.out_p3) {
            u32x4 h, md, lo;
#pragma unroll
            for (int e = 0; e < 4; e++) { unsigned a, b, c; split2_bf16_exact(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; md[e] = b; lo[e] = c; }
            *reinterpret_cast<u32x4*>(ep.out_p3 + goff) = h;
            *reinterpret_cast<u32x4*>(ep.out_p3 + goff + 16) = md;
            *reinterpret_cast<u32x4*>(ep.out_p3 + goff + 32) = lo;
        }
    }
}

// Split-K hand-off shared by the contraction kernels: every slice parks its raw accumulators in the workspace (register order: coalesced,
// agent-scope stores), the slice that arrives last at the tile's counter sums all slices in slice order (deterministic) and returns true.
template <int TM, int TN, int NT>
__device__ __forceinline__ bool p3_splitk_reduce(f32x16 (&acc)[TM][TN], float* ws, int* ws_count, long tile, int ks, int splitk, int tile_elems, int tid, int* flag) {
    float* slot0 = ws + tile * splitk * (long)tile_elems;
    float* mine = slot0 + (long)ks * tile_elems;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                __hip_atomic_store(mine + ((i * TN + j) * 16 + r) * NT + tid, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const int old = atomicAdd(ws_count + tile, 1);
        const int last = old == splitk - 1;
        if (last) __hip_atomic_store(ws_count + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = last;
    }
    __syncthreads();
    const int last = *flag;
    __syncthreads();   // the flag word is part of the epilogue's staging area
    if (!last) return false;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    for (int sl = 0; sl < splitk; sl++) {
        const float* src = slot0 + (long)sl * tile_elems;
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    acc[i][j][r] += __hip_atomic_load(src + ((i * TN + j) * 16 + r) * NT + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return true;
}

// ---- Non-finite operands.  The three-way split is exact for finite values only; +-Inf and NaN are stored as (x, 0, 0) (split2_bf16_exact), and a
// product like Inf.hi x 1.0.mid = Inf x 0 then puts NaN where an fp32 convolution gives Inf -- a different class for the step's gradient
// sanitiser (training_loop.py:308: NaN -> 0, +-Inf -> +-1e5).  Every contraction kernel therefore looks at its accumulators after the k-loop
// (any non-finite operand or overflow leaves a non-finite sum: one fma per register, one ballot), and a wave that finds one recomputes its tile
// of this block's reduction slice with fp32 FMAs on the merged planes: the result of an fp32 evaluation, whatever the operands.  Cold path:
// plain loops over (row, column), results handed back to the accumulator registers through a wave-private LDS image; a clean launch pays the
// check only.
template <int TM, int TN>
__device__ __forceinline__ bool p3_acc_non_finite(const f32x16 (&acc)[TM][TN]) {
    float chk = 0.f;
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) chk = fmaf(acc[i][j][r], 0.f, chk);
    return __builtin_amdgcn_ballot_w64(chk != chk) != 0ull;
}

typedef unsigned int ld128_t __attribute__((__vector_size__(16)));

// the eight fp32 values of one 48-byte P3 group (out-of-range vector offset -> zeros)
__device__ __forceinline__ void p3_load_group(const __amdgpu_buffer_rsrc_t rs, unsigned vo, int so, float (&v)[8]) {
    const ld128_t h = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so, 0), md = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so + 16, 0),
                  lo = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so + 32, 0);
#pragma unroll
    for (int e = 0; e < 4; e++) {
        v[2 * e] = bf_lo(h[e]) + (bf_lo(md[e]) + bf_lo(lo[e]));
        v[2 * e + 1] = bf_hi(h[e]) + (bf_hi(md[e]) + bf_hi(lo[e]));
    }
}

// one element: channel c of the P3 row at byte offset `row` (out of range -> 0)
__device__ __forceinline__ float p3_load_elem(const __amdgpu_buffer_rsrc_t rs, unsigned row, int c) {
    const unsigned vo = row == 0x80000000u ? row : row + (unsigned)((c >> 3) * 48 + (c & 7) * 2);
    const unsigned h = __builtin_amdgcn_raw_buffer_load_b16(rs, vo, 0, 0), md = __builtin_amdgcn_raw_buffer_load_b16(rs, vo, 16, 0),
                   lo = __builtin_amdgcn_raw_buffer_load_b16(rs, vo, 32, 0);
    return bf_lo(h) + (bf_lo(md) + bf_lo(lo));
}

// dot(row, col) for every element of the wave's WM x WN tile -> the wave's accumulator registers (img: wave-private LDS, WM * WN floats;
// acc[i][j][r] of lane (kl, cl) is row i*32 + (r&3) + 8*(r>>2) + 4*kl, column j*32 + cl)
template <int WM, int WN, int TM, int TN, typename Dot>
__device__ __forceinline__ void p3_cold_tile(f32x16 (&acc)[TM][TN], float* img, int lane, Dot dot) {
    for (int e = lane; e < WM * WN; e += 64) {
        const int row = e / WN, col = e - row * WN;
        const float v = dot(row, col);
        const int i = row >> 5, rr = row & 31, j = col >> 5, cl = col & 31;
        img[((i * TN + j) * 16 + (rr & 3) + 4 * (rr >> 3)) * 64 + ((rr >> 2) & 1) * 32 + cl] = v;
    }
    // (LDS operations of one wave execute in order: no barrier)
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = img[((i * TN + j) * 16 + r) * 64 + lane];
}

// Output row m of the gather kernel: byte offset of its reference pixel and the taps that fall inside the image (see p3_load_tile).
__device__ __forceinline__ void p3_nt_row(const P3NtParams& p, int m, unsigned& off, unsigned& vm) {
    vm = 0; off = 0;
    if (m >= p.M) return;
    const int hw = p.OH * p.OW;
    const int n = m / hw, rem = m - n * hw, oy = rem / p.OW, ox = rem - oy * p.OW;
    if (p.tap_mode == 0) {
        const int iy = oy * p.stride, ix = ox * p.stride;   // tap (pad, pad): in range for every stride / pad used
        off = (unsigned)(((n * p.H + iy) * p.W + ix) * p.Cin) * 6u;
        if (p.nty * p.ntx == 1) vm = 1u;                    // 1x1, pad 0
        else
            for (int ty = 0; ty < p.nty; ty++)
                for (int tx = 0; tx < p.ntx; tx++) {
                    const int sy = iy - p.pad + p.kh0 + ty, sx = ix - p.pad + p.kw0 + tx;
                    if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) vm |= 1u << (ty * p.ntx + tx);
                }
    } else {
        // destination (input-gradient) pixel of this parity class; source (output-gradient) pixel = (dst + pad - kh) / stride
        const int dy = oy * p.out_step + p.out_py + p.pad, dx = ox * p.out_step + p.out_px + p.pad;
        const int by = (dy - p.kh0) / p.stride, bx = (dx - p.kw0) / p.stride;   // tap (kh0, kw0): the largest source index
        off = (unsigned)(((n * p.H + by) * p.W + bx) * p.Cin) * 6u;
        for (int ty = 0; ty < p.nty; ty++)
            for (int tx = 0; tx < p.ntx; tx++) {
                const int sy = by - ty, sx = bx - tx;
                if (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) vm |= 1u << (ty * p.ntx + tx);
            }
    }
}

// Scalar offsets of k-tile (tap, cc) in A (relative to a row's reference pixel; p.A is shifted back so that they are never negative) and in B.
__device__ __forceinline__ void p3_nt_koff(const P3NtParams& p, int tap, int cc, int& sA, int& sB) {
    const int ty = tap / p.ntx, tx = tap - ty * p.ntx;
    int dpix;
    if (p.tap_mode == 0) dpix = (p.kh0 + ty) * p.W + (p.kw0 + tx);             // reference = tap (pad, pad); p.A shifted by (pad*W + pad) pixels
    else dpix = (p.nty - 1 - ty) * p.W + (p.ntx - 1 - tx);                      // reference = source of tap (kh0, kw0); p.A shifted by ((nty-1)*W + ntx-1) pixels
    sA = dpix * p.Cin * 6 + cc * 192;
    sB = ((p.kh0 + p.tstep * ty) * p.KW + p.kw0 + p.tstep * tx) * p.Cin * 6 + cc * 192;
}

// One k-tile of this wave's operand rows, global -> registers: its 16-row groups of A (gathered pixels) and B (weight rows), three
// 16-byte pieces per lane and group (piece = (lane % 4) ^ swizzle of its 64-byte instruction slice).  Register staging, not LDS-DMA:
// measured on MI355X (tools/p3_dev.py dma / reg, profiles/r04_p3_staging_probe.txt) `buffer_load ... lds` tops out at 42-55 GB/s per CU
// from L2-resident operands whatever the access shape, plain buffer_load_dwordx4 + ds_write_b128 moves 105-220 GB/s per CU, and every
// DMA-fed version of this kernel sat exactly on the first figure.  (Free functions: hipcc drops the host stub of a kernel template whose
// lambda captures mutable locals next to buffer builtins.)
template <int BM, int BN, int NW, bool ONE>
__device__ __forceinline__ void p3_load_tile(const P3NtParams& p, const __amdgpu_buffer_rsrc_t rsA, const __amdgpu_buffer_rsrc_t rsB,
                                             const unsigned* offA, const unsigned* vmA, const unsigned* offB, int& tap, int& cc, int ktpt, bool live,
                                             ld128_t* ra, ld128_t* rb) {
    constexpr int RGA = BM / 16 / NW, RGB = BN / 16 / NW;
    if constexpr (ONE) {
        // 1x1 convolution = plain GEMM over contiguous P3 rows: no tap walk, no per-row validity masks (rows past M carry an out-of-range offset
        // from the start); k-tile `cc` of either operand sits cc * 192 bytes into its row (one s_mul per k-tile).  A dead tile is an out-of-range
        // VECTOR offset (the descriptor's range check does not cover the scalar offset): one v_cndmask per row group
        const int so = cc * 192;
        ++cc;      // (unconditionally: a dead tile's offset is irrelevant, and a `live`-dependent update would put the counter into a VGPR -- the loads' scalar
        //           offset then needs a readfirstlane loop each)
#pragma unroll
        for (int i = 0; i < RGA; i++) {
            const unsigned vo = live ? offA[i] : 0x80000000u;
#pragma unroll
            for (int j = 0; j < 3; j++) ra[i * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, so + j * 64, 0);
        }
#pragma unroll
        for (int i = 0; i < RGB; i++) {
            const unsigned vo = live ? offB[i] : 0x80000000u;
#pragma unroll
            for (int j = 0; j < 3; j++) rb[i * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, so + j * 64, 0);
        }
        return;
    }
    // live = false: a k-tile past the end of the slice.  Its loads are still issued (out-of-range offsets: zeros, no memory traffic) so that every
    // iteration of the k-loop issues the same number of loads and the compiler's vmcnt waits stay exact (a conditional load would make the
    // wait before each LDS store cover the whole prefetch queue).
    int sA, sB;   // uniform
    p3_nt_koff(p, tap, cc, sA, sB);
#pragma unroll
    for (int i = 0; i < RGA; i++) {
        const unsigned vo = (live && ((vmA[i] >> tap) & 1u)) ? offA[i] : 0x80000000u;   // out of the descriptor's range: the load returns zeros (padding)
#pragma unroll
        for (int j = 0; j < 3; j++) ra[i * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, sA + j * 64, 0);
    }
#pragma unroll
    for (int i = 0; i < RGB; i++) {
        const unsigned vo = live ? offB[i] : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 3; j++) rb[i * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, vo, sB + j * 64, 0);
    }
    if (live && ++cc == ktpt) { cc = 0; ++tap; }
}

template <int BM, int BN, int NW>
__device__ __forceinline__ void p3_store_tile(char* sb, int w, int lane, const ld128_t* ra, const ld128_t* rb) {
    constexpr int RGA = BM / 16 / NW, RGB = BN / 16 / NW, A_BYTES = BM * 192;
#pragma unroll
    for (int i = 0; i < RGA; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) *reinterpret_cast<ld128_t*>(sb + ((w * RGA + i) * 3 + j) * 1024 + lane * 16) = ra[i * 3 + j];
#pragma unroll
    for (int i = 0; i < RGB; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) *reinterpret_cast<ld128_t*>(sb + A_BYTES + ((w * RGB + i) * 3 + j) * 1024 + lane * 16) = rb[i * 3 + j];
}

// BM x BN block tile, NW waves as a WGM x WGN grid (WGN = 4 for eight waves on a 128-wide tile), two LDS stages of BK = 32, PF k-tiles in
// flight in registers: during the MFMAs of k-tile t the loads of tiles t+1 .. t+PF are outstanding; tile t+1 is written to the other stage before
// the iteration's one barrier and its registers take tile t+1+PF.  What bounds these launches is bytes in flight per CU over the load latency
// (profiles/r04_pmc_sq.txt: matrix pipe 22-26 % busy on the 1x1 shapes with ONE tile in flight; 12 waves x 6 KiB / ~1.5 us = the ~50 GB/s per
// CU every earlier variant sat on), so the depth is what the register file allows at the tile's occupancy.
template <int BM, int BN, int NW, int PF, int NST, bool GROUPS, bool ONE = false>
__device__ __forceinline__ void p3_nt_body(P3NtParams p, const P3Group2& g2, const int bx, const int by, const int bz) {   // (bx, by, bz): the block's place in the (tiles, k-slices, parity classes) grid
    constexpr int NT = NW * 64;   // NST = LDS stages: 2 = one barrier per k-tile; 1 = two barriers, half the LDS (more blocks per CU)
    constexpr int WGN = (NW == 8 && BN >= 128) ? 4 : 2, WGM = NW / WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;   // wave tile, 32x32 accumulators per wave
    constexpr int A_BYTES = BM * 192, B_BYTES = BN * 192, ST_BYTES = A_BYTES + B_BYTES;
    constexpr int RGA = BM / 16 / NW, RGB = BN / 16 / NW;   // 16-row groups per wave and operand
    static_assert(TM >= 1 && TN >= 1 && RGA >= 1 && RGB >= 1 && BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "tile / wave-count mismatch");
    extern __shared__ __attribute__((aligned(1024))) char p3_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w / WGN, wn = w % WGN;   // w in an SGPR: LDS-DMA bases and scalar offsets stay scalar
    const int cl = lane & 31, kl = lane >> 5;

    int ws_tile0 = 0;
    if (p.nclass > 1) {
        // this block's parity class (heaviest = most taps first: blocks are dispatched in z order, the light classes fill the tail)
        const int sd = p.stride, cls = p.nclass - 1 - bz, py = cls / sd, px = cls - py * sd;
        ws_tile0 = cls * p.mtiles * p.ntiles;
        p.kh0 = (py + p.pad) % sd; p.kw0 = (px + p.pad) % sd;
        p.nty = p.kh0 < p.KH ? (p.KH - p.kh0 + sd - 1) / sd : 0; p.ntx = p.kw0 < p.KW ? (p.KW - p.kw0 + sd - 1) / sd : 0;
        const int ntaps = p.nty * p.ntx;
        if (ntaps == 0) { p.nty = 0; p.ntx = 1; }
        const unsigned shift = (unsigned)(((p.nty > 0 ? p.nty - 1 : 0) * p.W + (p.ntx - 1)) * p.Cin) * 6u;
        p.A -= shift; p.a_bytes += shift;
        p.OH = (p.out_H - py + sd - 1) / sd; p.OW = (p.out_W - px + sd - 1) / sd;
        p.M = p.nimg * p.OH * p.OW; p.out_py = py; p.out_px = px;
        p.nkt = ntaps * p.Cin / 32;
        p.mtiles = (p.M + BM - 1) / BM;
    } else if (GROUPS && p.ngroups > 1 && bz > 0) {
        p.A = g2.A2; p.B = g2.B2; p.ep = g2.ep2;
        ws_tile0 = p.mtiles * p.ntiles;
    }
    int tm, tn;
    if (!xcd_tile(bx, p.mtiles, p.ntiles, p.xm, p.xn, tm, tn)) return;
    const int t = tm * p.ntiles + tn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int ks = by;
    const int per = (p.nkt + p.splitk - 1) / p.splitk;
    const int kt0 = ks * per, kt1 = min(p.nkt, kt0 + per);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.B), 0, p.b_bytes, 0x00020000);

    // ---- per-lane source rows (row-in-group = lane / 4, 16-byte piece = (lane % 4) ^ swizzle)
    const int rr = lane >> 2, pq = (lane & 3) ^ ((rr >> 2) & 3);
    const int ktpt = p.Cin >> 5;                          // k-tiles per tap
    unsigned offA[RGA], vmA[RGA], offB[RGB];
#pragma unroll
    for (int i = 0; i < RGA; i++) {
        unsigned vm, off;
        p3_nt_row(p, m0 + (w * RGA + i) * 16 + rr, off, vm);
        offA[i] = (ONE && vm == 0u) ? 0x80000000u : off + pq * 16; vmA[i] = vm;
    }
#pragma unroll
    for (int i = 0; i < RGB; i++) {
        const int n = n0 + (w * RGB + i) * 16 + rr;
        offB[i] = n < p.N ? (unsigned)n * (unsigned)(p.KH * p.KW * p.Cin) * 6u + pq * 16 : 0x80000000u;
    }

    // uniform walk over (tap, channel tile)
    int tap = ONE ? 0 : kt0 / ktpt, cc = kt0 - tap * ktpt;          // next k-tile to issue (ONE: cc counts k-tiles from the row start)

    // ---- fragment read offsets: piece e = 3*(2s + kl) + plane -> 64-byte slice e / 4, slot (e % 4) ^ swizzle(row)
    const int frr = cl & 15, frg = cl >> 4;               // row within its 16-row group, group within the 32-row block
    unsigned fo[2][3];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const int e = 3 * (2 * s + kl) + pl;
            fo[s][pl] = (unsigned)((frg * 3 + (e >> 2)) * 1024 + frr * 64 + (((e & 3) ^ ((frr >> 2) & 3)) << 4));
        }
    const unsigned foA = (unsigned)(wm * (WM / 16) * 3 * 1024), foB = (unsigned)(A_BYTES + wn * (WN / 16) * 3 * 1024);

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nloc = kt1 - kt0;
    ld128_t ra[PF][RGA * 3], rb[PF][RGB * 3];
#pragma unroll
    for (int d = 0; d < PF; d++) p3_load_tile<BM, BN, NW, ONE>(p, rsA, rsB, offA, vmA, offB, tap, cc, ktpt, d < nloc, ra[d], rb[d]);
    if (nloc > 0) p3_store_tile<BM, BN, NW>(p3_smem, w, lane, ra[0], rb[0]);
    __syncthreads();
    for (int it0 = 0; it0 < nloc; it0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; d++) {
            const int it = it0 + d;
            // tile `it` went from ra[d] to LDS before the last barrier: its registers take tile it + PF
            p3_load_tile<BM, BN, NW, ONE>(p, rsA, rsB, offA, vmA, offB, tap, cc, ktpt, it + PF < nloc, ra[d], rb[d]);
            __builtin_amdgcn_sched_barrier(0);      // pin the next tile's loads ahead of this tile's MFMAs: left alone, the scheduler sinks them behind the fragment reads (their registers are
            //                                         then shared with the fragments) and every k-tile waits out its own load latency
            if (it < nloc) {
                const char* sb = p3_smem + (it % NST) * ST_BYTES;
#pragma unroll
                for (int s = 0; s < 2; s++) {
                    bf16x8_t a[TM][3], b[TN][3];
#pragma unroll
                    for (int i = 0; i < TM; i++)
#pragma unroll
                        for (int pl = 0; pl < 3; pl++)
                            a[i][pl] = *reinterpret_cast<const bf16x8_t*>(sb + foA + i * (2 * 3 * 1024) + fo[s][pl]);
#pragma unroll
                    for (int j = 0; j < TN; j++)
#pragma unroll
                        for (int pl = 0; pl < 3; pl++)
                            b[j][pl] = *reinterpret_cast<const bf16x8_t*>(sb + foB + j * (2 * 3 * 1024) + fo[s][pl]);
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
                if (NST == 1) __syncthreads();   // every wave is done reading the one stage
                if (it + 1 < nloc) p3_store_tile<BM, BN, NW>(p3_smem + ((it + 1) % NST) * ST_BYTES, w, lane, ra[(d + 1) % PF], rb[(d + 1) % PF]);
                __syncthreads();
            }
        }
    }

    // ---- non-finite operands: this wave's tile of the slice again, in fp32 (the k-loop ended on a barrier: LDS is free; a wave's image
    // lives in the region its epilogue stages through)
    constexpr int CP = WN + 4;
    // (the launch sizes the dynamic LDS as max(NST * ST_BYTES, NW * WM * CP * 4): the epilogue's staging area)
    if (p3_acc_non_finite<TM, TN>(acc)) {
        p3_cold_tile<WM, WN, TM, TN>(acc, reinterpret_cast<float*>(p3_smem) + w * (WM * CP), lane, [&](int row, int col) -> float {
            unsigned off, vm;
            p3_nt_row(p, m0 + wm * WM + row, off, vm);
            const int n = n0 + wn * WN + col;
            const unsigned ob = n < p.N ? (unsigned)n * (unsigned)(p.KH * p.KW * p.Cin) * 6u : 0x80000000u;
            float s = 0.f;
            for (int kt = kt0; kt < kt1; kt++) {
                const int tp = kt / ktpt;
                int sA, sB;
                p3_nt_koff(p, tp, kt - tp * ktpt, sA, sB);
                const unsigned vo = ((vm >> tp) & 1u) ? off : 0x80000000u;   // padding: zeros, multiplied like the matrix path's
                for (int g = 0; g < 4; g++) {
                    float a[8], b[8];
                    p3_load_group(rsA, vo, sA + g * 48, a); p3_load_group(rsB, ob, sB + g * 48, b);
#pragma unroll
                    for (int e = 0; e < 8; e++) s = fmaf(a[e], b[e], s);
                }
            }
            return s;
        });
    }
    // ---- split-K, then the epilogue
    if (p.splitk > 1 && !p3_splitk_reduce<TM, TN, NT>(acc, p.ws, p.ws_count, ws_tile0 + t, ks, p.splitk, BM * BN, tid, reinterpret_cast<int*>(p3_smem))) return;
    const int ohw = p.OH * p.OW;
    p3_wave_epilogue<WM, WN, TM, TN>(p.ep, reinterpret_cast<float*>(p3_smem) + w * (WM * CP), acc, lane, n0 + wn * WN, p.N, [&](int rl) -> long {
        const int m = m0 + wm * WM + rl;
        if (m >= p.M) return -1;
        if (p.out_step == 1 && p.out_H == p.OH && p.out_W == p.OW) return m;
        const int nn = m / ohw, rem = m - nn * ohw, oy = rem / p.OW, ox = rem - oy * p.OW;
        return ((long)nn * p.out_H + (oy * p.out_step + p.out_py)) * p.out_W + (ox * p.out_step + p.out_px);
    });
}

template <int BM, int BN, int NW, int PF, int NST, bool ONE = false>
__global__ __launch_bounds__(NW * 64) void p3_nt_kernel(P3NtParams p, P3Group2 g2) { p3_nt_body<BM, BN, NW, PF, NST, true, ONE>(p, g2, blockIdx.x, blockIdx.y, blockIdx.z); }

// ---------------------------------------------------------------------------------------------
// p3_c3_kernel: 3x3 / stride 1 / pad 1 convolutions (forward, and the data gradient as the same conv with mirrored taps) on 2-D pixel
// patches.  What bounds the gather kernel above is the CU's global-load path (~50 GB/s per CU of L1 misses, whatever the access shape:
// profiles/r04_p3_staging_probe.txt), and a per-tap gather fetches every input pixel nine times.  Here a block's 128 output pixels are
// 8 x 16 patches (16 x 8 / two 8 x 8 images / ... for narrower images), the (PH+2) x (PW+2) halo of the current 32-channel slice is loaded
// ONCE into LDS (out-of-image pixels = out-of-range offsets = zeros) and all nine taps read their A fragments from it at shifted
// addresses; only the 64 x 32 weight tile of a tap streams per k-step.  Bytes per MFMA drop 2.3x (36 KiB + 9 x 12 KiB per channel slice
// instead of 9 x 36 KiB).  The next slice's halo is prefetched into registers across the nine taps and written between two barriers.
// 4 waves (2 x 2), wave tile 64 pixels x 32 channels; LDS: halo 39 KiB + 3 weight stages x 12 KiB = 75 KiB -> two blocks per CU.
struct P3C3Params {
    const char* X; const char* Wt;
    unsigned x_bytes, w_bytes;
    int N_img, H, W, Cin, Nout;
    int lgPW, lgPH, NP;          // patch width / height (powers of two), patches per 128-row tile
    int HWp, HPIX, NH, NG;       // halo row width, halo pixels per patch, halo pixels per tile, 16-pixel groups per tile (<= 13)
    int ppr, ppi;                // patches per image row, per image
    int flip;                    // data gradient: tap (ty, tx) reads the halo at (2 - ty, 2 - tx)
    int ncc;                     // 32-channel slices
    int splitk; float* ws; int* ws_count;
    int mtiles, ntiles, xm, xn;
    P3Epi ep;
    int ngroups;                 // grouped forward: see P3NtParams / P3Group2
};

constexpr int C3_NGMAX = 13, C3_A_BYTES = C3_NGMAX * 3072, C3_B_BYTES = 4 * 3072, C3_LDS = C3_A_BYTES + 3 * C3_B_BYTES;

__device__ __forceinline__ void p3_c3_decode_tile_row(const P3C3Params& p, int tm, int r, int& n, int& y0, int& x0, int& py, int& px) {
    const int lgpp = p.lgPW + p.lgPH;
    const int sp = r >> lgpp, q = r & ((1 << lgpp) - 1);
    py = q >> p.lgPW; px = q & ((1 << p.lgPW) - 1);
    const int pid = tm * p.NP + sp;
    n = pid / p.ppi; const int prem = pid - n * p.ppi;
    const int pr = prem / p.ppr, pc = prem - pr * p.ppr;
    y0 = pr << p.lgPH; x0 = pc << p.lgPW;
}

template <int TAP>
__device__ __forceinline__ void p3_c3_tap(const P3C3Params& p, const __amdgpu_buffer_rsrc_t rsA, const __amdgpu_buffer_rsrc_t rsB, char* smem, int w, int lane,
                                          const unsigned (&offA)[4], unsigned offB, int cc, bool more_cc, bool more_tile,
                                          ld128_t (&ar)[12], ld128_t (&br)[3], const unsigned (&h0)[2], const unsigned (&ce)[2][3], const unsigned (&fo)[2][3],
                                          unsigned foB, f32x16 (&acc)[2][1]) {
    // ---- global loads first: the next weight tile (tap + 1, or tap 0 of the next slice), and this tap's share of the next halo
    if (more_tile) {
        const int ntap = TAP == 8 ? 0 : TAP + 1, ncc = TAP == 8 ? cc + 1 : cc;
        const int sB = ntap * p.Cin * 6 + ncc * 192;
#pragma unroll
        for (int j = 0; j < 3; j++) br[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB, sB + j * 64, 0);
    }
    if (more_cc) {
#pragma unroll
        for (int q = TAP * 12 / 9; q < (TAP + 1) * 12 / 9; q++)
            ar[q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[q / 3], (cc + 1) * 192 + (q % 3) * 64, 0);
    }
    // ---- this tap's MFMAs: A fragments from the halo at the tap's shift, B fragments from stage TAP % 3
    constexpr int ty = TAP / 3, tx = TAP % 3;
    const int dt = p.flip ? (2 - ty) * p.HWp + (2 - tx) : ty * p.HWp + tx;
    unsigned hb[2], sw[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const unsigned h = h0[i] + dt;
        hb[i] = (h >> 4) * 3072 + ((h & 15) << 6);
        sw[i] = ((h >> 2) & 3) << 4;
    }
    const char* sb = smem + C3_A_BYTES + (TAP % 3) * C3_B_BYTES + foB;
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) {
        bf16x8_t a[2][3], b[3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int pl = 0; pl < 3; pl++) a[i][pl] = *reinterpret_cast<const bf16x8_t*>(smem + hb[i] + (ce[s2][pl] ^ sw[i]));
#pragma unroll
        for (int pl = 0; pl < 3; pl++) b[pl] = *reinterpret_cast<const bf16x8_t*>(sb + fo[s2][pl]);
#pragma unroll
        for (int i = 0; i < 2; i++) {   // smallest terms first
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[1], acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[2], acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[0], acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[1], acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[0], acc[i][0], 0, 0, 0);
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], acc[i][0], 0, 0, 0);
        }
    }
    // ---- the next weight tile goes to its stage; one barrier per tap
    if (more_tile) {
        char* dst = smem + C3_A_BYTES + ((TAP + 1) % 3) * C3_B_BYTES + w * 3072 + lane * 16;
#pragma unroll
        for (int j = 0; j < 3; j++) *reinterpret_cast<ld128_t*>(dst + j * 1024) = br[j];
    }
    __syncthreads();
}

template <bool GROUPS>
__device__ __forceinline__ void p3_c3_body(P3C3Params p, const P3Group2& g2, const int bx, const int by, const int bz) {
    constexpr int BM = 128, BN = 64, NT = 256, WM = 64, WN = 32;
    if (GROUPS && p.ngroups > 1 && bz > 0) { p.X = g2.A2; p.Wt = g2.B2; p.ep = g2.ep2; }
    extern __shared__ __attribute__((aligned(1024))) char p3_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int cl = lane & 31, kl = lane >> 5;
    int tm, tn;
    if (!xcd_tile(bx, p.mtiles, p.ntiles, p.xm, p.xn, tm, tn)) return;
    const int t = tm * p.ntiles + tn;
    const int n0 = tn * BN;
    const int ks = by;
    const int per = (p.ncc + p.splitk - 1) / p.splitk;
    const int cc0 = ks * per, cc1 = min(p.ncc, cc0 + per);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.Wt), 0, p.w_bytes, 0x00020000);

    // ---- loader rows: halo groups w, w + 4, w + 8, w + 12 (16 pixels each) and weight group w
    const int rr = lane >> 2, pq = (lane & 3) ^ ((rr >> 2) & 3);
    unsigned offA[4], offB;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int h = (w + 4 * i) * 16 + rr;
        unsigned off = 0x80000000u;
        if (h < p.NH) {
            const int sp = h / p.HPIX, rem = h - sp * p.HPIX, hy = rem / p.HWp, hx = rem - hy * p.HWp;
            const int pid = tm * p.NP + sp;
            const int n = pid / p.ppi, prem = pid - n * p.ppi, pr = prem / p.ppr, pc = prem - pr * p.ppr;
            const int y = (pr << p.lgPH) + hy - 1, x = (pc << p.lgPW) + hx - 1;
            if (n < p.N_img && y >= 0 && y < p.H && x >= 0 && x < p.W) off = (unsigned)(((n * p.H + y) * p.W + x) * p.Cin) * 6u + pq * 16;
        }
        offA[i] = off;
    }
    {
        const int n = n0 + w * 16 + rr;
        offB = n < p.Nout ? (unsigned)n * (unsigned)(9 * p.Cin) * 6u + pq * 16 : 0x80000000u;
    }
    // ---- fragment addressing
    unsigned h0[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = wm * WM + i * 32 + cl;
        const int lgpp = p.lgPW + p.lgPH;
        const int sp = r >> lgpp, q = r & ((1 << lgpp) - 1);
        h0[i] = (unsigned)(sp * p.HPIX + (q >> p.lgPW) * p.HWp + (q & ((1 << p.lgPW) - 1)));
    }
    const int frr = cl & 15, frg = cl >> 4;
    unsigned ce[2][3], fo[2][3];
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const int e = 3 * (2 * s2 + kl) + pl;
            ce[s2][pl] = (unsigned)(((e >> 2) * 1024) | ((e & 3) << 4));
            fo[s2][pl] = (unsigned)((frg * 3 + (e >> 2)) * 1024 + frr * 64 + (((e & 3) ^ ((frr >> 2) & 3)) << 4));
        }
    const unsigned foB = (unsigned)(wn * 2 * 3072);

    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][0][r] = 0.f;

    ld128_t ar[12], br[3];
    if (cc1 > cc0) {
        // first halo and first weight tile
#pragma unroll
        for (int q = 0; q < 12; q++) ar[q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA[q / 3], cc0 * 192 + (q % 3) * 64, 0);
#pragma unroll
        for (int j = 0; j < 3; j++) br[j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB, cc0 * 192 + j * 64, 0);
#pragma unroll
        for (int q = 0; q < 12; q++)
            if (w + 4 * (q / 3) < C3_NGMAX) *reinterpret_cast<ld128_t*>(p3_smem + ((w + 4 * (q / 3)) * 3 + (q % 3)) * 1024 + lane * 16) = ar[q];
#pragma unroll
        for (int j = 0; j < 3; j++) *reinterpret_cast<ld128_t*>(p3_smem + C3_A_BYTES + w * 3072 + j * 1024 + lane * 16) = br[j];
    }
    __syncthreads();
    for (int cc = cc0; cc < cc1; cc++) {
        const bool more_cc = cc + 1 < cc1;
#define C3_TAP(T_) p3_c3_tap<T_>(p, rsA, rsB, p3_smem, w, lane, offA, offB, cc, more_cc, (T_) < 8 || more_cc, ar, br, h0, ce, fo, foB, acc)
        C3_TAP(0); C3_TAP(1); C3_TAP(2); C3_TAP(3); C3_TAP(4); C3_TAP(5); C3_TAP(6); C3_TAP(7); C3_TAP(8);
#undef C3_TAP
        if (more_cc) {   // every wave passed tap 8's barrier: the halo can be replaced
#pragma unroll
            for (int q = 0; q < 12; q++)
                if (w + 4 * (q / 3) < C3_NGMAX) *reinterpret_cast<ld128_t*>(p3_smem + ((w + 4 * (q / 3)) * 3 + (q % 3)) * 1024 + lane * 16) = ar[q];
            __syncthreads();
        }
    }

    constexpr int CP = WN + 4;
    static_assert(4 * WM * CP * 4 <= C3_LDS, "epilogue staging does not fit");
    if (p3_acc_non_finite<2, 1>(acc)) {   // non-finite operands: this wave's tile of the slice again, in fp32 (see p3_cold_tile)
        p3_cold_tile<WM, WN, 2, 1>(acc, reinterpret_cast<float*>(p3_smem) + w * (WM * CP), lane, [&](int row, int col) -> float {
            int n, y0, x0, py, px;
            p3_c3_decode_tile_row(p, tm, wm * WM + row, n, y0, x0, py, px);
            if (n >= p.N_img) return 0.f;
            const int y = y0 + py, x = x0 + px, no = n0 + wn * WN + col;
            const unsigned ob = no < p.Nout ? (unsigned)no * (unsigned)(9 * p.Cin) * 6u : 0x80000000u;
            float s = 0.f;
            for (int cc = cc0; cc < cc1; cc++)
                for (int tap = 0; tap < 9; tap++) {
                    const int ty = tap / 3, tx = tap - ty * 3;
                    const int sy = y + (p.flip ? 1 - ty : ty - 1), sx = x + (p.flip ? 1 - tx : tx - 1);
                    const unsigned vo = (sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) ? (unsigned)(((n * p.H + sy) * p.W + sx) * p.Cin) * 6u : 0x80000000u;
                    for (int g = 0; g < 4; g++) {
                        float a[8], b[8];
                        p3_load_group(rsA, vo, cc * 192 + g * 48, a); p3_load_group(rsB, ob, tap * p.Cin * 6 + cc * 192 + g * 48, b);
#pragma unroll
                        for (int e = 0; e < 8; e++) s = fmaf(a[e], b[e], s);
                    }
                }
            return s;
        });
    }
    if (p.splitk > 1 && !p3_splitk_reduce<2, 1, NT>(acc, p.ws, p.ws_count, t + bz * p.mtiles * p.ntiles, ks, p.splitk, BM * BN, tid, reinterpret_cast<int*>(p3_smem))) return;
    p3_wave_epilogue<WM, WN, 2, 1>(p.ep, reinterpret_cast<float*>(p3_smem) + w * (WM * CP), acc, lane, n0 + wn * WN, p.Nout, [&](int rl) -> long {
        int n, y0, x0, py, px;
        p3_c3_decode_tile_row(p, tm, wm * WM + rl, n, y0, x0, py, px);
        if (n >= p.N_img) return -1;
        return ((long)n * p.H + (y0 + py)) * p.W + (x0 + px);
    });
}

__global__ __launch_bounds__(256, 2) void p3_c3_kernel(P3C3Params p, P3Group2 g2) { p3_c3_body<true>(p, g2, blockIdx.x, blockIdx.y, blockIdx.z); }

// ---------------------------------------------------------------------------------------------
// p3_tn_kernel: weight gradients.  dW[co][tap][ci] += row_scale[co] * sum_pix dY[pix][co] * X[src(pix, tap)][ci]: per tap a GEMM whose
// reduction index is the output pixel, so BOTH operands are stored reduction-major (a pixel's channels are contiguous).  The LDS image
// of a [32 pixels x 32 channels] P3 block is the same one the NT kernel builds (16-pixel groups x three 1-KiB DMA instructions, XOR
// swizzle); the MFMA fragments (8 consecutive pixels of one channel per lane) come out of it with ds_read_b64_tr_b16, the gfx950
// transposing LDS read: a 16-lane group reads a [4 pixels][16 channels] block, lane i supplying the address of (pixel i/4, channel
// quad i%4) and receiving the 4 pixels of channel i.  Conflict-free for this image (32 distinct 8-byte units per 32-lane pass).
// Grid: (co tiles x ci tiles x taps, pixel slices); fp32 atomics into dW (the flat .grad buffer), like gemm_conv.hip's weight gradients.
struct P3TnParams {
    const char* dY; const char* X;   // X shifted back by pad_off bytes
    unsigned dy_bytes, x_bytes;
    int Cout, Cin, KH, KW, stride, pad;
    int H, W, OH, OW;                // X grid, dY grid
    int npix;                        // N * OH * OW
    float inv_ohw, inv_ow;
    int mtiles, ntiles, splitk;
    const float* row_scale;          // [Cout] or null
    float alpha;
    float* dW;                       // [Cout][KH*KW][Cin]
};

__device__ __forceinline__ void fastdivmod(int x, int d, float invd, int& q, int& r) {   // exact for 0 <= x < 2^24
    q = (int)((float)x * invd);
    r = x - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r < 0) { --q; r += d; }
}

// Transposing LDS reads as inline asm: hipcc's waitcnt pass cannot tell the builtin's LDS access from the LDS-DMA writes in flight and
// drains the whole DMA queue (vmcnt(0)) before every one of them, i.e. it serialises the ring.  The asm is invisible to that pass, so the
// lgkmcnt waits are written by hand (tr_wait) before the fragments are used.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2_t tr_read(unsigned addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// (sched_barrier: the asm results look ready to the scheduler, which would otherwise hoist the consuming MFMAs above the wait)
__device__ __forceinline__ void tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
__device__ __forceinline__ bf16x8_t tr_join(u32x2_t lo, u32x2_t hi) {
    const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int BM, int BN>
__device__ __forceinline__ void p3_tn_load(const P3TnParams& p, const __amdgpu_buffer_rsrc_t rsA, const __amdgpu_buffer_rsrc_t rsB, int w, int lane,
                                           int k0, int kend, int m0, int n0, int kh, int kw, ld128_t* ra, ld128_t* rb) {
    constexpr int UA = BM / 64, UB = BN / 64;   // [16 pixels x 32 channels] units per wave and operand
    const int rg = w & 1, rr = lane >> 2, pq = (lane & 3) ^ ((rr >> 2) & 3);
    const int pix = k0 + rg * 16 + rr;
    const bool inr = pix < kend;
    // A = dY: dense pixel rows
    const unsigned offA = inr ? (unsigned)pix * (unsigned)p.Cout * 6u + pq * 16 : 0x80000000u;
#pragma unroll
    for (int i = 0; i < UA; i++) {
        const int cb = (w >> 1) * UA + i;
        const int sA = (m0 + cb * 32) * 6;
#pragma unroll
        for (int j = 0; j < 3; j++) ra[i * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, offA, sA + j * 64, 0);
    }
    // B = X gathered at the tap
    unsigned offB = 0x80000000u;
    {
        int n, rem, oy, ox;
        fastdivmod(pix, p.OH * p.OW, p.inv_ohw, n, rem);
        fastdivmod(rem, p.OW, p.inv_ow, oy, ox);
        const int sy = oy * p.stride - p.pad + kh, sx = ox * p.stride - p.pad + kw;
        if (inr && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W)
            offB = (unsigned)(((n * p.H + oy * p.stride) * p.W + ox * p.stride) * p.Cin) * 6u + pq * 16;   // reference = tap (pad, pad); the tap's shift is uniform
    }
    const int sBt = (kh * p.W + kw) * p.Cin * 6;
#pragma unroll
    for (int i = 0; i < UB; i++) {
        const int cb = (w >> 1) * UB + i;
        const int sB = sBt + (n0 + cb * 32) * 6;
#pragma unroll
        for (int j = 0; j < 3; j++) rb[i * 3 + j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, offB, sB + j * 64, 0);
    }
}

template <int BM, int BN>
__device__ __forceinline__ void p3_tn_store(char* sb, int w, int lane, const ld128_t* ra, const ld128_t* rb) {
    constexpr int UA = BM / 64, UB = BN / 64, A_BYTES = BM * 192;
    const int rg = w & 1;
#pragma unroll
    for (int i = 0; i < UA; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) *reinterpret_cast<ld128_t*>(sb + ((((w >> 1) * UA + i) * 2 + rg) * 3 + j) * 1024 + lane * 16) = ra[i * 3 + j];
#pragma unroll
    for (int i = 0; i < UB; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) *reinterpret_cast<ld128_t*>(sb + A_BYTES + ((((w >> 1) * UB + i) * 2 + rg) * 3 + j) * 1024 + lane * 16) = rb[i * 3 + j];
}

template <int BM, int BN, int PF, int NST>
__device__ __forceinline__ void p3_tn_body(const P3TnParams& p, const int bx, const int by) {   // bx: pixel slice, by: (tile, tap)
    constexpr int A_BYTES = BM * 192, B_BYTES = BN * 192, ST_BYTES = A_BYTES + B_BYTES;   // NST LDS stages (see p3_nt_body)
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(1024))) char p3_smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w >> 1, wn = w & 1;
    const int cl = lane & 31, kl = lane >> 5;
    const int taps = p.KH * p.KW;
    // grid = (pixel slices, tiles x taps): blocks are dispatched x-fastest and block i runs on XCD i % 8, so with a multiple of 8 slices every
    // block that reads a given pixel slice of dY / X shares one XCD's L2 (the other order made all 8 L2s fetch every slice: 155 MB of fabric
    // traffic per launch, profiles/r04a_pmc_traffic.json)
    int b = by;
    const int tap = b % taps; b /= taps;
    const int tn = b % p.ntiles, tm = b / p.ntiles;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkt = (p.npix + 31) >> 5;
    const int per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = bx * per, kt1 = min(nkt, kt0 + per);
    const int nloc = kt1 - kt0;
    if (nloc <= 0) return;
    const int kend = min(kt1 * 32, p.npix);   // pixels past the slice load as zeros (out-of-range offsets): the dead tiles of the prefetch queue

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.dY), 0, p.dy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.X), 0, p.x_bytes, 0x00020000);

    // transposing fragment reads: lane -> (channel group g of 16, lane-in-group i16, pixel octet kl); read r covers pixels 8 kl + 4 r .. + 3
    const int g = cl >> 4, i16 = cl & 15, chq = i16 & 3;
    const int c = 2 * g + (chq >> 1), half = chq & 1;
    unsigned fo[2][3];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int pl = 0; pl < 3; pl++) {
            const int rowin = 8 * kl + 4 * r + (i16 >> 2);
            const int e = 3 * c + pl;
            fo[r][pl] = (unsigned)((e >> 2) * 1024 + rowin * 64 + (((e & 3) ^ ((rowin >> 2) & 3)) << 4) + half * 8);
        }
    const unsigned foA = (unsigned)(wm * TM * 2 * 3072), foB = (unsigned)(A_BYTES + wn * TN * 2 * 3072);
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)p3_smem;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    // PF pixel tiles in flight in registers (see p3_nt_kernel); every iteration issues the same loads, dead tiles as out-of-range offsets
    ld128_t ga[PF][(BM / 64) * 3], gb[PF][(BN / 64) * 3];
#pragma unroll
    for (int d = 0; d < PF; d++) p3_tn_load<BM, BN>(p, rsA, rsB, w, lane, (kt0 + d) * 32, kend, m0, n0, kh, kw, ga[d], gb[d]);
    p3_tn_store<BM, BN>(p3_smem, w, lane, ga[0], gb[0]);
    __syncthreads();
    for (int it0 = 0; it0 < nloc; it0 += PF) {
#pragma unroll
      for (int d = 0; d < PF; d++) {
        const int it = it0 + d;
        p3_tn_load<BM, BN>(p, rsA, rsB, w, lane, (kt0 + it + PF) * 32, kend, m0, n0, kh, kw, ga[d], gb[d]);
        if (it < nloc) {
        const unsigned sbase = lds_base + (unsigned)((it % NST) * ST_BYTES);
        u32x2_t ra[2][TM][3][2], rb[2][TN][3][2];   // [k16 step][block][plane][pixel half]
#pragma unroll
        for (int s = 0; s < 2; s++) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    ra[s][i][pl][0] = tr_read(sbase + foA + (i * 2 + s) * 3072 + fo[0][pl]);
                    ra[s][i][pl][1] = tr_read(sbase + foA + (i * 2 + s) * 3072 + fo[1][pl]);
                }
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    rb[s][j][pl][0] = tr_read(sbase + foB + (j * 2 + s) * 3072 + fo[0][pl]);
                    rb[s][j][pl][1] = tr_read(sbase + foB + (j * 2 + s) * 3072 + fo[1][pl]);
                }
            if (s == 0) tr_wait();   // step 1's reads stay in flight behind step 0's MFMAs
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s == 1) tr_wait();
            bf16x8_t a[TM][3], bb[TN][3];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) a[i][pl] = tr_join(ra[s][i][pl][0], ra[s][i][pl][1]);
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int pl = 0; pl < 3; pl++) bb[j][pl] = tr_join(rb[s][j][pl][0], rb[s][j][pl][1]);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[j][0], acc[i][j], 0, 0, 0);
                }
        }
        if (NST == 1) __syncthreads();   // every wave is done reading the one stage
        if (it + 1 < nloc) p3_tn_store<BM, BN>(p3_smem + ((it + 1) % NST) * ST_BYTES, w, lane, ga[(d + 1) % PF], gb[(d + 1) % PF]);
        __syncthreads();
        }
      }
    }
    // acc[i][j][r]: row (out channel) = (r&3) + 8*(r>>2) + 4*kl, col (in channel) = cl
    if (p3_acc_non_finite<TM, TN>(acc)) {   // non-finite operands: this wave's tile of the pixel slice again, in fp32 (see p3_cold_tile)
        p3_cold_tile<TM * 32, TN * 32, TM, TN>(acc, reinterpret_cast<float*>(p3_smem) + w * (TM * TN * 1024), lane, [&](int row, int col) -> float {
            const int co = m0 + wm * TM * 32 + row, ci = n0 + wn * TN * 32 + col;
            if (co >= p.Cout || ci >= p.Cin) return 0.f;
            float s = 0.f;
            const int pend = min(kt1 * 32, kend);
            for (int pix = kt0 * 32; pix < pend; pix++) {
                int n, rem, oy, ox;
                fastdivmod(pix, p.OH * p.OW, p.inv_ohw, n, rem);
                fastdivmod(rem, p.OW, p.inv_ow, oy, ox);
                const int sy = oy * p.stride - p.pad + kh, sx = ox * p.stride - p.pad + kw;
                const bool in = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
                const unsigned xo = in ? (unsigned)(((n * p.H + oy * p.stride) * p.W + ox * p.stride) * p.Cin) * 6u + (unsigned)((kh * p.W + kw) * p.Cin * 6) : 0x80000000u;
                s = fmaf(p3_load_elem(rsA, (unsigned)pix * (unsigned)p.Cout * 6u, co), p3_load_elem(rsB, xo, ci), s);   // padding: zero, multiplied like the matrix path's
            }
            return s;
        });
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
            if (m >= p.Cout) continue;
            const float sc = p.row_scale ? p.alpha * p.row_scale[m] : p.alpha;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = n0 + (wn * TN + j) * 32 + cl;
                if (n < p.Cin) atomicAdd(p.dW + ((long)m * taps + tap) * p.Cin + n, acc[i][j][r] * sc);
            }
        }
}

template <int BM, int BN, int PF, int NST>
__global__ __launch_bounds__(256) void p3_tn_kernel(P3TnParams p) { p3_tn_body<BM, BN, PF, NST>(p, blockIdx.x, blockIdx.y); }

// ---------------------------------------------------------------------------------------------
// Data gradient + weight gradient of one convolution as ONE launch.  Each of the two kernels alone is a short launch whose blocks run in
// lock-step (all load first, all store last: the matrix pipe idles through the fill and the chip-wide epilogue burst), and the two are
// independent: on two streams they overlap to 0.65-0.85 of their serial time (tools/p3_dev.py pair), but parallel branches of a captured
// hipGraph do not (0.92-1.0; `pair graph`), and the step is a graph.  So the pair is one grid holding both kernels' blocks; both read dY
// (one L2 fill).
// Which body a block of the paired grid runs, and its index in that body's own grid: the weight gradient's blocks first, padded to a multiple of
// 8 so that both bodies keep their block -> XCD alignment (block b runs on XCD b % 8); the data gradient's blocks move in as those retire.  Measured
// alternatives (profiles/r04_p3_sweeps.txt): data gradient first 438.6 images/s against 443.5; the two kinds interleaved in chunks of 8 in
// proportion to their counts 437.7 (0.98 of the separate launches against 0.85): CUs that hold both kinds at once lose more than the overlap gains.
__device__ __forceinline__ bool p3_pair_split(int b, int n_tn8, int& idx) {
    if (b < n_tn8 * 8) { idx = b; return true; }
    idx = b - n_tn8 * 8;
    return false;
}

template <int BM, int BN, int NST, bool ONE = false>
__global__ __launch_bounds__(256) void p3_bwd_pair_nt_kernel(P3NtParams pn, P3TnParams pt, int n_tn, int n_tn8, int tn_sk, int nt_gx, int nt_sk) {
    int idx;
    if (p3_pair_split(blockIdx.x, n_tn8, idx)) {
        if (idx < n_tn) p3_tn_body<64, 64, 1, 1>(pt, idx % tn_sk, idx / tn_sk);
        return;
    }
    P3Group2 none;   // (never read: GROUPS = false)
    p3_nt_body<BM, BN, 4, 1, NST, false, ONE>(pn, none, idx % nt_gx, (idx / nt_gx) % nt_sk, idx / (nt_gx * nt_sk));
}

__global__ __launch_bounds__(256, 2) void p3_bwd_pair_c3_kernel(P3C3Params pc, P3TnParams pt, int n_tn, int n_tn8, int tn_sk, int c3_gx) {
    int idx;
    if (p3_pair_split(blockIdx.x, n_tn8, idx)) {
        if (idx < n_tn) p3_tn_body<64, 64, 1, 1>(pt, idx % tn_sk, idx / tn_sk);
        return;
    }
    P3Group2 none;   // (never read: GROUPS = false)
    p3_c3_body<false>(pc, none, idx % c3_gx, idx / c3_gx, 0);
}

// ---- fp32 <-> P3 streaming conversions.  One thread = one 8-channel group.
__global__ __launch_bounds__(256) void p3_split_kernel(const float* __restrict__ src, long ld, char* __restrict__ dst, long rows, int C) {
    const int cg = C >> 3;
    const long total = rows * cg;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const long r = u / cg; const int g = (int)(u - r * cg);
        const float4 v0 = *reinterpret_cast<const float4*>(src + r * ld + g * 8), v1 = *reinterpret_cast<const float4*>(src + r * ld + g * 8 + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        u32x4 h, md, lo;
#pragma unroll
        for (int e = 0; e < 4; e++) { unsigned a, b, c; split2_bf16_exact(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; md[e] = b; lo[e] = c; }
        char* d = dst + u * 48;
        *reinterpret_cast<u32x4*>(d) = h; *reinterpret_cast<u32x4*>(d + 16) = md; *reinterpret_cast<u32x4*>(d + 32) = lo;
    }
}

__global__ __launch_bounds__(256) void p3_merge_kernel(const char* __restrict__ src, float* __restrict__ dst, long ld, long rows, int C) {
    const int cg = C >> 3;
    const long total = rows * cg;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const long r = u / cg; const int g = (int)(u - r * cg);
        const char* s = src + u * 48;
        const u32x4 h = *reinterpret_cast<const u32x4*>(s), md = *reinterpret_cast<const u32x4*>(s + 16), lo = *reinterpret_cast<const u32x4*>(s + 32);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            v[2 * e] = bf_lo(h[e]) + (bf_lo(md[e]) + bf_lo(lo[e]));
            v[2 * e + 1] = bf_hi(h[e]) + (bf_hi(md[e]) + bf_hi(lo[e]));
        }
        *reinterpret_cast<float4*>(dst + r * ld + g * 8) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + r * ld + g * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// Weights for the data gradient: w [O][KH][KW][I] fp32 -> P3 [I][KH*KW][O] (k = (tap, out channel) contiguous, taps in the
// original order; the kernel's tap walk does the flipping).  One thread = 8 out channels of one (i, tap).
__global__ __launch_bounds__(256) void p3_weight_bwd_kernel(const float* __restrict__ w, const float* __restrict__ o_scale, char* __restrict__ dst, int O, int T, int I) {
    const int og = O >> 3;
    const long total = (long)I * T * og;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const int g = (int)(u % og); const long it = u / og; const int tp = (int)(it % T); const int i = (int)(it / T);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = w[((long)(g * 8 + e) * T + tp) * I + i] * (o_scale ? o_scale[g * 8 + e] : 1.f);
        u32x4 h, md, lo;
#pragma unroll
        for (int e = 0; e < 4; e++) { unsigned a, b, c; split2_bf16_exact(v[2 * e], v[2 * e + 1], a, b, c); h[e] = a; md[e] = b; lo[e] = c; }
        char* d = dst + u * 48;
        *reinterpret_cast<u32x4*>(d) = h; *reinterpret_cast<u32x4*>(d + 16) = md; *reinterpret_cast<u32x4*>(d + 32) = lo;
    }
}

// Both P3 images of every convolution weight of a module in ONE launch (after each optimiser step): table row c =
// {w, o_scale or 0, dst_fwd or 0, dst_bwd or 0, O, T, I, first block}; a block covers 256 of the conv's O*T*I/8 groups.
//   dst_fwd: the split of w as stored, [O][T][I]  (forward B operand);  dst_bwd: [I][T][O] of w * o_scale[o]  (data-gradient B operand).
__global__ __launch_bounds__(256) void p3_weight_prep_kernel(const long long* __restrict__ table, int nconv) {
    int c = 0;
    while (c + 1 < nconv && (long long)blockIdx.x >= table[(c + 1) * 8 + 7]) ++c;
    const long long* row = table + c * 8;
    const float* w = reinterpret_cast<const float*>(row[0]);
    const float* sc = reinterpret_cast<const float*>(row[1]);
    char* df = reinterpret_cast<char*>(row[2]);
    char* db = reinterpret_cast<char*>(row[3]);
    const int O = (int)row[4], T = (int)row[5], I = (int)row[6];
    const long u = ((long)blockIdx.x - row[7]) * 256 + threadIdx.x;
    const long units = (long)O * T * I / 8;
    if (u >= units) return;
    if (df) {
        const float4 v0 = *reinterpret_cast<const float4*>(w + u * 8), v1 = *reinterpret_cast<const float4*>(w + u * 8 + 4);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        u32x4 h, md, lo;
#pragma unroll
        for (int e = 0; e < 4; e++) { unsigned a, b, cc; split2_bf16_exact(v[2 * e], v[2 * e + 1], a, b, cc); h[e] = a; md[e] = b; lo[e] = cc; }
        char* d = df + u * 48;
        *reinterpret_cast<u32x4*>(d) = h; *reinterpret_cast<u32x4*>(d + 16) = md; *reinterpret_cast<u32x4*>(d + 32) = lo;
    }
    if (db) {
        // u -> (out-channel group g, tap tp, in channel i), i fastest: the eight reads of a wave are eight coalesced rows
        const int i = (int)(u % I); const long r = u / I; const int tp = (int)(r % T); const int g = (int)(r / T);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = w[((long)(g * 8 + e) * T + tp) * I + i] * (sc ? sc[g * 8 + e] : 1.f);
        u32x4 h, md, lo;
#pragma unroll
        for (int e = 0; e < 4; e++) { unsigned a, b, cc; split2_bf16_exact(v[2 * e], v[2 * e + 1], a, b, cc); h[e] = a; md[e] = b; lo[e] = cc; }
        char* d = db + (((long)i * T + tp) * (O / 8) + g) * 48;
        *reinterpret_cast<u32x4*>(d) = h; *reinterpret_cast<u32x4*>(d + 16) = md; *reinterpret_cast<u32x4*>(d + 32) = lo;
    }
}

// XCD array (xm x xn = 8) over an mtiles x ntiles grid minimising the bytes every L2 has to pull through the fabric: a_bytes * xn + b_bytes * xm.
static void choose_xcd_array(int mtiles, int ntiles, double a_bytes, double b_bytes, int& xm, int& xn, long& grid_x) {
    double best = -1.0; xm = 8; xn = 1;
    for (int n = 1; n <= 8; n *= 2) {
        const int m = 8 / n;
        if ((n > ntiles && n > 1) || (m > mtiles && m > 1)) continue;
        const double cost = a_bytes * n + b_bytes * m;
        if (best < 0 || cost < best) { best = cost; xm = m; xn = n; }
    }
    grid_x = 8L * cdiv(mtiles, xm) * cdiv(ntiles, xn);
}


// The dynamic-LDS opt-in is a per-device function attribute: `raised` is one flag per device ordinal (a process that launches on a second GPU
// must opt in there too).
struct RaisedPerDevice { bool on[64] = {}; };
static bool raise_lds(const void* kern, size_t lds, const char* what, RaisedPerDevice& raised) {
    if (lds <= 64 * 1024) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && raised.on[dev]) return true;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        set_error("%s: cannot raise the dynamic LDS limit to %zu bytes", what, lds);
        return false;
    }
    if (dev >= 0) raised.on[dev] = true;
    return true;
}

// What the launch policy chose for this thread's most recent plane-format launch (ldetr_p3_last_launch): tests assert the template a bench
// shape reaches, so that a policy change cannot silently leave a kernel without a parity case.
//   kind: 1 gather (p3_nt), 2 patch (p3_c3), 3 weight gradient (p3_tn), 4 paired gather + weight gradient, 5 paired patch + weight gradient
struct P3LastLaunch { int kind, bm, bn, nw, splitk, xm, xn, ncls, tn_splitk; long grid; };
static thread_local P3LastLaunch t_last = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// ---- gather kernel: tile grid, XCD array, split-K scratch of one configuration -> grid dimensions
struct NtGrid { long gx; int sk, ncls; size_t lds; };

template <int BM, int BN, int NW, int NST>
static NtGrid plan_nt(P3NtParams& p, int sk) {
    p.mtiles = cdiv(p.M, BM); p.ntiles = cdiv(p.N, BN);
    NtGrid g; g.ncls = p.nclass > 1 ? p.nclass : (p.ngroups > 1 ? p.ngroups : 1);   // grid z: parity classes (data gradient) or groups (grouped forward)
    const long nt = (long)p.mtiles * p.ntiles;
    choose_xcd_array(p.mtiles, p.ntiles, (double)p.M * p.Cin * 6.0, (double)p.N * p.KH * p.KW * p.Cin * 6.0, p.xm, p.xn, g.gx);
    if (sk > p.nkt) sk = p.nkt;
    if (sk < 1) sk = 1;
    p.splitk = sk; p.ws = nullptr; p.ws_count = nullptr;
    if (sk > 1 && !splitk_ws_alloc(nt * g.ncls, (size_t)nt * g.ncls * sk * BM * BN * sizeof(float), &p.ws, &p.ws_count)) p.splitk = sk = 1;
    g.sk = sk;
    constexpr int WGN_ = (NW == 8 && BN >= 128) ? 4 : 2, WM_ = BM / (NW / WGN_), WN_ = BN / WGN_;
    constexpr size_t lds_loop = (size_t)NST * (BM + BN) * 192, lds_epi = (size_t)NW * WM_ * (WN_ + 4) * 4;   // the epilogue stages the tile through LDS
    g.lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    return g;
}

// Block-slot target of the split-K / pixel-slice policies by the launch's algorithmic work: below 1.5 GFLOP (the trunk at a
// few samples per GPU) a launch is a latency chain and every extra reduction slice adds a partial-tile round trip to it -> 256 slots; above, 512
// slots fill the chip.  2 samples per GPU 19.14 -> 18.77 ms, 4 per GPU 21.46 -> 21.02, 16 per GPU unchanged (profiles/r04_p3_sweeps.txt).
static long slot_target(double flops, long dflt) {
    static const double small = 1.5 * 1e9;
    return (flops < small && dflt > 256) ? 256 : dflt;
}
static double nt_flops(const P3NtParams& p) { return 2.0 * p.M * p.N * p.nkt * 32.0 * (p.nclass > 1 ? p.nclass : 1); }
static double tn_flops(const P3TnParams& p) { return 2.0 * p.npix * p.Cout * (double)p.Cin * p.KH * p.KW; }

// split-K factor of a tile configuration: fill `slots` block slots, at least four k-tiles per slice
static int nt_splitk(const P3NtParams& p, int bm, int bn, long slots) {
    const long nt = (long)cdiv(p.M, bm) * cdiv(p.N, bn);
    int sk = 1;
    if (nt < slots * 3 / 4) { sk = (int)(slots / nt); if (sk > p.nkt / 4) sk = p.nkt / 4; if (sk > 16) sk = 16; if (sk < 1) sk = 1; }
    return sk;
}

// 1x1 convolution without parity classes (forward incl. the strided downsample convs, stride-1 data gradients): the gather kernel's plain-GEMM loop
static bool nt_is_plain_gemm(const P3NtParams& p) { return p.KH * p.KW == 1 && p.pad == 0 && p.nclass <= 1 && (p.tap_mode == 0 || p.stride == 1); }

template <int BM, int BN, int NW, int PF, int NST>
static int launch_nt_cfg(P3NtParams& p, const P3Group2& g2, int sk, hipStream_t st) {
    const NtGrid g = plan_nt<BM, BN, NW, NST>(p, sk);
    if constexpr (PF == 1 && NST == 1 && (BM == BN) && (NW == 4 ? BM == 64 : BM == 128)) {
        if (nt_is_plain_gemm(p)) {
            auto kern1 = p3_nt_kernel<BM, BN, NW, PF, NST, true>;
            static RaisedPerDevice raised1;
            if (!raise_lds(reinterpret_cast<const void*>(kern1), g.lds, "p3_nt", raised1)) return LDETR_ERR_LAUNCH;
            hipLaunchKernelGGL(kern1, dim3((unsigned)g.gx, g.sk, g.ncls), NW * 64, g.lds, st, p, g2);
            t_last = {1, BM, BN, NW, g.sk, p.xm, p.xn, g.ncls, 0, g.gx};
            note_engine_launch(true);
            return check_launch("p3_nt");
        }
    }
    auto kern = p3_nt_kernel<BM, BN, NW, PF, NST>;
    static RaisedPerDevice raised;
    if (!raise_lds(reinterpret_cast<const void*>(kern), g.lds, "p3_nt", raised)) return LDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)g.gx, g.sk, g.ncls), NW * 64, g.lds, st, p, g2);
    t_last = {1, BM, BN, NW, g.sk, p.xm, p.xn, g.ncls, 0, g.gx};
    note_engine_launch(true);
    return check_launch("p3_nt");
}

static int launch_nt(P3NtParams& p, const P3Group2& g2, bool forward, hipStream_t st) {
    // Tile configurations: 1 = 128x128, 8 waves; 2 = 128x64, 4 waves; 3 = 64x64, 4 waves; 4 = 128x128, 4 waves.  One LDS stage (two barriers
    // per k-tile, half the LDS: more blocks per CU) and one k-tile of register prefetch are the defaults: the sweeps over the trunk shapes at
    // 16 x 256^2 (profiles/r04_p3_sweeps.txt: tile x prefetch depth x stage count) put 64x64 / one stage first on all but the large-grid
    // forward shapes, where the 128x128 tile's halved operand traffic wins; deeper prefetch never paid (the loads are not what these short
    // launches wait for: blocks run in lock-step through fill, loop and a chip-wide epilogue burst).
    // (two LDS stages / two k-tiles of prefetch were instantiated through round 5: profiles/r04_p3_sweeps.txt, profiles/r05_p3_pf.txt.)
    static const int force_tile = (int)knob("P3_TILE", 0);
    int cfg = 3;
    if (forward && p.nclass <= 1 && ((p.M >= 65536 && p.N >= 128) || (p.M >= 16384 && p.N >= 256 && p.nkt <= 16))) cfg = 1;
    if (force_tile) cfg = force_tile;
    const int bm = cfg == 3 ? 64 : 128, bn = (cfg == 1 || cfg == 4) ? 128 : 64;
    const int sk = nt_splitk(p, bm, bn, slot_target(nt_flops(p), (cfg == 2 || cfg == 3) ? 512 : 256));
    switch (cfg) {
        case 1: return launch_nt_cfg<128, 128, 8, 1, 1>(p, g2, sk, st);
        case 2: return launch_nt_cfg<128, 64, 4, 1, 1>(p, g2, sk, st);
        case 3: return launch_nt_cfg<64, 64, 4, 1, 1>(p, g2, sk, st);
        default: return launch_nt_cfg<128, 128, 4, 1, 1>(p, g2, sk, st);
    }
}

// 3x3 / stride 1 / pad 1 on pixel patches: returns false when the geometry does not fit (the caller then takes the gather kernel).
static bool c3_geometry(int N, int H, int W, P3C3Params& p) {
    auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
    int PW = W < 16 ? W : 16;
    if (lg(PW) < 0 || W % PW) return false;
    int PH = 128 / PW; if (PH > H) PH = H;
    while (PH > 1 && (H % PH || lg(PH) < 0)) PH >>= 1;
    if (lg(PH) < 0 || H % PH || 128 % (PW * PH)) return false;
    p.lgPW = lg(PW); p.lgPH = lg(PH); p.NP = 128 / (PW * PH);
    p.HWp = PW + 2; p.HPIX = (PH + 2) * (PW + 2); p.NH = p.NP * p.HPIX; p.NG = (p.NH + 15) / 16;
    if (p.NG > C3_NGMAX) return false;
    p.ppr = W / PW; p.ppi = (H / PH) * (W / PW);
    p.N_img = N; p.H = H; p.W = W;
    p.mtiles = (int)(((long)N * p.ppi + p.NP - 1) / p.NP);
    return true;
}

// tile grid, split-K scratch, XCD array of the patch kernel -> grid_x (grid = (grid_x, splitk))
static long plan_c3(P3C3Params& p) {
    p.ntiles = cdiv(p.Nout, 64);
    const long nt = (long)p.mtiles * p.ntiles;
    int sk = 1;
    const long slots = slot_target(2.0 * p.N_img * p.H * p.W * (double)p.Nout * 9.0 * p.Cin, 512);
    if (nt < slots * 3 / 4) { sk = (int)(slots / nt); if (sk > p.ncc / 2) sk = p.ncc / 2; if (sk > 16) sk = 16; if (sk < 1) sk = 1; }
    if (sk > p.ncc) sk = p.ncc;
    p.splitk = sk; p.ws = nullptr; p.ws_count = nullptr;
    const int ng = p.ngroups > 1 ? p.ngroups : 1;
    if (sk > 1 && !splitk_ws_alloc(nt * ng, (size_t)nt * ng * sk * 128 * 64 * sizeof(float), &p.ws, &p.ws_count)) p.splitk = sk = 1;
    long grid_x;
    choose_xcd_array(p.mtiles, p.ntiles, (double)p.N_img * p.H * p.W * p.Cin * 6.0, (double)p.Nout * 9 * p.Cin * 6.0, p.xm, p.xn, grid_x);
    return grid_x;
}

static int launch_c3(P3C3Params& p, const P3Group2& g2, hipStream_t st) {
    const long grid_x = plan_c3(p);
    static RaisedPerDevice raised;
    if (!raise_lds(reinterpret_cast<const void*>(&p3_c3_kernel), C3_LDS, "p3_c3", raised)) return LDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(p3_c3_kernel, dim3((unsigned)grid_x, p.splitk, p.ngroups > 1 ? p.ngroups : 1), 256, C3_LDS, st, p, g2);
    t_last = {2, 128, 64, 4, p.splitk, p.xm, p.xn, p.ngroups > 1 ? p.ngroups : 1, 0, grid_x};
    note_engine_launch(true);
    return check_launch("p3_c3");
}

static void fill_epi(P3Epi& e, const ldetr_p3_epilogue* s) {
    memset(&e, 0, sizeof(e));
    e.alpha = 1.f;
    if (!s) return;
    e.alpha = s->alpha; e.col_scale = s->col_scale; e.col_bias = s->col_bias;
    e.res_p3 = (const char*)s->residual_p3; e.res_f32 = s->residual_f32; e.mask_p3 = (const char*)s->relu_mask_p3; e.relu = s->relu;
}

// ---- weight gradient: pixel-slice count of a tile configuration
template <int BM, int BN>
static void plan_tn(P3TnParams& p, int target_blocks) {
    p.mtiles = cdiv(p.Cout, BM); p.ntiles = cdiv(p.Cin, BN);
    const long nt = (long)p.mtiles * p.ntiles * p.KH * p.KW;
    const int nkt = (p.npix + 31) / 32;
    int sk = (int)((target_blocks + nt - 1) / nt);
    if (sk > nkt / 4) sk = nkt / 4;
    if (sk >= 8) sk = (sk + 4) / 8 * 8;   // a multiple of the XCD count: slice s -> XCD s % 8
    if (sk < 1) sk = 1;
    p.splitk = sk;
}

template <int BM, int BN, int PF, int NST>
static int launch_tn_cfg(P3TnParams& p, int target_blocks, hipStream_t st) {
    plan_tn<BM, BN>(p, target_blocks);
    const long nt = (long)p.mtiles * p.ntiles * p.KH * p.KW;
    constexpr size_t lds_loop = (size_t)NST * (BM + BN) * 192, lds_cold = (size_t)(BM / 64) * (BN / 64) * 4096 * 4;   // (the cold path's accumulator image)
    constexpr size_t lds = lds_loop > lds_cold ? lds_loop : lds_cold;
    auto kern = p3_tn_kernel<BM, BN, PF, NST>;
    static RaisedPerDevice raised;
    if (!raise_lds(reinterpret_cast<const void*>(kern), lds, "p3_tn", raised)) return LDETR_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3(p.splitk, (unsigned)nt, 1), 256, lds, st, p);
    t_last = {3, BM, BN, 4, 0, 0, 0, 1, p.splitk, nt};
    note_engine_launch(true);
    return check_launch("p3_tn");
}

static int launch_tn(P3TnParams& p, hipStream_t st) {
    // 64 x 64, one LDS stage, one pixel tile in flight: the most resident waves per CU (sweep over the trunk shapes: 1179 us against 1580-1940 for
    // the wider tiles; two stages / two tiles in flight 1222-1243 us against 1193: profiles/r04_p3_sweeps.txt)
    static const int force_tile = (int)knob("P3_WTILE", 0);
    switch (force_tile) {
        case 1: return launch_tn_cfg<128, 128, 1, 1>(p, (int)slot_target(tn_flops(p), 256), st);
        case 2: return launch_tn_cfg<128, 64, 1, 1>(p, (int)slot_target(tn_flops(p), 256), st);
        default: return launch_tn_cfg<64, 64, 1, 1>(p, (int)slot_target(tn_flops(p), 512), st);
    }
}

// ---- operand set-up shared by the single and the paired entry points
static int setup_bwd_data(const void* dy, int N, int OH, int OW, int Cout, const void* wb, int Cin, int KH, int KW, int stride, int pad, int IH, int IW,
                          const ldetr_p3_epilogue* ep, void* out_p3, float* out_f32, bool& use_c3, P3C3Params& c, P3NtParams& p) {
    LDETR_CHECK(dy && wb && (out_p3 || out_f32), "p3_conv2d_bwd_data: null operand");
    LDETR_CHECK(Cout % 32 == 0 && Cin % 8 == 0 && KH * KW <= 32 && pad < KH && pad < KW && (stride == 1 || stride == 2),
                "p3_conv2d_bwd_data: unsupported geometry (Cin=%d Cout=%d k=%dx%d stride=%d pad=%d)", Cin, Cout, KH, KW, stride, pad);
    const long dybytes = (long)N * OH * OW * Cout * 6, wbytes = (long)Cin * KH * KW * Cout * 6;
    use_c3 = false;
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && OH == IH && OW == IW && dybytes < 0x7fffffffL && wbytes < 0x7fffffffL) {
        memset(&c, 0, sizeof(c));
        if (c3_geometry(N, IH, IW, c)) {   // dx = conv(dy, wb) with mirrored taps: source pixel = dst + 1 - k
            c.X = (const char*)dy; c.x_bytes = (unsigned)dybytes; c.Wt = (const char*)wb; c.w_bytes = (unsigned)wbytes;
            c.Cin = Cout; c.Nout = Cin; c.flip = 1; c.ncc = Cout / 32;
            fill_epi(c.ep, ep); c.ep.out_p3 = (char*)out_p3; c.ep.out_f32 = out_f32;
            use_c3 = true;
            return LDETR_OK;
        }
    }
    memset(&p, 0, sizeof(p));
    p.B = (const char*)wb; p.b_bytes = (unsigned)wbytes;
    p.N = Cin; p.Cin = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.H = OH; p.W = OW;
    p.out_H = IH; p.out_W = IW; p.out_step = stride; p.tap_mode = 1; p.tstep = stride; p.nimg = N;
    fill_epi(p.ep, ep); p.ep.out_p3 = (char*)out_p3; p.ep.out_f32 = out_f32;
    const long max_shift = ((long)(KH - 1) * OW + (KW - 1)) * Cout * 6;
    LDETR_CHECK(dybytes + max_shift < 0x7fffffffL && wbytes < 0x7fffffffL, "p3_conv2d_bwd_data: tensor too large for 31-bit buffer offsets");
    if (stride == 1) {
        p.kh0 = 0; p.kw0 = 0; p.nty = KH; p.ntx = KW;
        p.A = (const char*)dy - max_shift; p.a_bytes = (unsigned)(dybytes + max_shift);
        p.OH = IH; p.OW = IW; p.M = N * IH * IW; p.nkt = KH * KW * Cout / 32; p.nclass = 1;
    } else {
        // sized for the heaviest class (py, px) = (stride - 1, ...): the kernel derives every class's own taps, grid and shift from its class index
        p.A = (const char*)dy; p.a_bytes = (unsigned)dybytes;
        p.OH = (IH + stride - 1) / stride; p.OW = (IW + stride - 1) / stride; p.M = N * p.OH * p.OW;
        p.nkt = ((KH + stride - 1) / stride) * ((KW + stride - 1) / stride) * Cout / 32; p.nclass = stride * stride;
        if (p.nkt == 0) p.nkt = 1;
    }
    return LDETR_OK;
}

static int setup_bwd_weight(const void* x, int N, int H, int W, int Cin, const void* dy, int Cout, int KH, int KW, int stride, int pad,
                            const float* dy_scale, float* dw, P3TnParams& p) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    LDETR_CHECK(x && dy && dw, "p3_conv2d_bwd_weight: null operand");
    LDETR_CHECK(Cin % 32 == 0 && Cout % 32 == 0 && pad < KH && pad < KW && (stride == 1 || stride == 2),
                "p3_conv2d_bwd_weight: unsupported geometry (Cin=%d Cout=%d k=%dx%d stride=%d pad=%d)", Cin, Cout, KH, KW, stride, pad);
    const long xbytes = (long)N * H * W * Cin * 6, dybytes = (long)N * OH * OW * Cout * 6, pad_off = ((long)pad * W + pad) * Cin * 6;
    LDETR_CHECK(xbytes + pad_off < 0x7fffffffL && dybytes < 0x7fffffffL && (long)N * OH * OW < (1L << 24), "p3_conv2d_bwd_weight: tensor too large for 31-bit buffer offsets");
    memset(&p, 0, sizeof(p));
    p.dY = (const char*)dy; p.dy_bytes = (unsigned)dybytes; p.X = (const char*)x - pad_off; p.x_bytes = (unsigned)(xbytes + pad_off);
    p.Cout = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
    p.npix = N * OH * OW; p.inv_ohw = 1.0f / (float)(OH * OW); p.inv_ow = 1.0f / (float)OW;
    p.row_scale = dy_scale; p.alpha = 1.f; p.dW = dw;
    return LDETR_OK;
}

}  // namespace ldetr

using namespace ldetr;

extern "C" int ldetr_p3_last_launch(int32_t* info10) {
    LDETR_CHECK(info10 != nullptr, "p3_last_launch: null output");
    const P3LastLaunch& l = t_last;
    const int32_t v[10] = {l.kind, l.bm, l.bn, l.nw, l.splitk, l.xm, l.xn, l.ncls, l.tn_splitk, (int32_t)(l.grid > 0x7fffffffL ? 0x7fffffff : l.grid)};
    for (int i = 0; i < 10; i++) info10[i] = v[i];
    return LDETR_OK;
}

extern "C" int ldetr_p3_split_f32(const float* src, int64_t ld, void* dst, int64_t rows, int C, void* stream) {
    LDETR_CHECK(src && dst && rows >= 0 && C > 0 && C % 8 == 0 && ld % 4 == 0, "p3_split: C must be a multiple of 8 and rows 16-byte aligned (C=%d)", C);
    if (rows == 0) return LDETR_OK;
    const long units = rows * (C / 8);
    const int blocks = (int)std::min<long>((units + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(p3_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (long)ld, (char*)dst, (long)rows, C);
    return check_launch("p3_split");
}

extern "C" int ldetr_p3_merge_f32(const void* src, float* dst, int64_t ld, int64_t rows, int C, void* stream) {
    LDETR_CHECK(src && dst && rows >= 0 && C > 0 && C % 8 == 0 && ld % 4 == 0, "p3_merge: C must be a multiple of 8 (C=%d)", C);
    if (rows == 0) return LDETR_OK;
    const long units = rows * (C / 8);
    const int blocks = (int)std::min<long>((units + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(p3_merge_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const char*)src, dst, (long)ld, (long)rows, C);
    return check_launch("p3_merge");
}

extern "C" int ldetr_p3_weight_bwd(const float* w_ohwi, const float* o_scale, void* dst, int O, int KH, int KW, int I, void* stream) {
    LDETR_CHECK(w_ohwi && dst && O % 8 == 0 && I > 0 && KH > 0 && KW > 0, "p3_weight_bwd: O must be a multiple of 8 (O=%d)", O);
    const long units = (long)I * KH * KW * (O / 8);
    const int blocks = (int)std::min<long>((units + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(p3_weight_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w_ohwi, o_scale, (char*)dst, O, KH * KW, I);
    return check_launch("p3_weight_bwd");
}

extern "C" int ldetr_p3_weight_prep(const int64_t* table_dev, int nconv, int total_blocks, void* stream) {
    LDETR_CHECK(table_dev && nconv > 0 && total_blocks > 0, "p3_weight_prep: empty table");
    hipLaunchKernelGGL(p3_weight_prep_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const long long*>(table_dev), nconv);
    return check_launch("p3_weight_prep");
}

// x2 != NULL: the grouped form (ldetr_p3_conv2d_fwd_dual) -- the same geometry on a second operand / epilogue / output set in the same launch
static int p3_conv2d_fwd_impl(const void* x, int N, int H, int W, int Cin, const void* w, int Cout, int KH, int KW, int stride, int pad,
                              const ldetr_p3_epilogue* ep, void* out_p3, float* out_f32,
                              const void* x2, const void* w2, const ldetr_p3_epilogue* ep2, void* out2_p3, float* out2_f32, void* stream) {
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    LDETR_CHECK(x && w && (out_p3 || out_f32), "p3_conv2d_fwd: null operand");
    LDETR_CHECK(!x2 || (w2 && (out2_p3 || out2_f32)), "p3_conv2d_fwd_dual: null operand in the second set");
    LDETR_CHECK(Cin % 32 == 0 && Cout % 8 == 0 && KH * KW <= 32 && pad < KH && pad < KW && (stride == 1 || stride == 2),
                "p3_conv2d_fwd: unsupported geometry (Cin=%d Cout=%d k=%dx%d stride=%d pad=%d)", Cin, Cout, KH, KW, stride, pad);
    const long xbytes = (long)N * H * W * Cin * 6, wbytes = (long)Cout * KH * KW * Cin * 6, pad_off = ((long)pad * W + pad) * Cin * 6;
    LDETR_CHECK(xbytes + pad_off < 0x7fffffffL && wbytes < 0x7fffffffL && (long)N * OH * OW * Cout * 6 < (1L << 40), "p3_conv2d_fwd: tensor too large for 31-bit buffer offsets");
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1) {
        P3C3Params c; memset(&c, 0, sizeof(c));
        if (c3_geometry(N, H, W, c)) {
            c.X = (const char*)x; c.x_bytes = (unsigned)xbytes; c.Wt = (const char*)w; c.w_bytes = (unsigned)wbytes;
            c.Cin = Cin; c.Nout = Cout; c.flip = 0; c.ncc = Cin / 32;
            fill_epi(c.ep, ep); c.ep.out_p3 = (char*)out_p3; c.ep.out_f32 = out_f32;
            P3Group2 g2; memset(&g2, 0, sizeof(g2));
            if (x2) {
                c.ngroups = 2; g2.A2 = (const char*)x2; g2.B2 = (const char*)w2;
                fill_epi(g2.ep2, ep2); g2.ep2.out_p3 = (char*)out2_p3; g2.ep2.out_f32 = out2_f32;
            }
            return launch_c3(c, g2, (hipStream_t)stream);
        }
    }
    P3NtParams p; memset(&p, 0, sizeof(p));
    p.A = (const char*)x - pad_off; p.a_bytes = (unsigned)(xbytes + pad_off); p.B = (const char*)w; p.b_bytes = (unsigned)wbytes;
    p.M = N * OH * OW; p.N = Cout; p.Cin = Cin; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.H = H; p.W = W; p.OH = OH; p.OW = OW;
    p.out_H = OH; p.out_W = OW; p.out_step = 1; p.tap_mode = 0; p.kh0 = 0; p.kw0 = 0; p.tstep = 1; p.nty = KH; p.ntx = KW;
    p.nkt = KH * KW * Cin / 32;
    fill_epi(p.ep, ep); p.ep.out_p3 = (char*)out_p3; p.ep.out_f32 = out_f32;
    P3Group2 g2; memset(&g2, 0, sizeof(g2));
    if (x2) {
        p.ngroups = 2; g2.A2 = (const char*)x2 - pad_off; g2.B2 = (const char*)w2;
        fill_epi(g2.ep2, ep2); g2.ep2.out_p3 = (char*)out2_p3; g2.ep2.out_f32 = out2_f32;
    }
    if (p.M == 0) return LDETR_OK;
    return launch_nt(p, g2, true, (hipStream_t)stream);
}

extern "C" int ldetr_p3_conv2d_fwd(const void* x, int N, int H, int W, int Cin, const void* w, int Cout, int KH, int KW, int stride, int pad,
                                   const ldetr_p3_epilogue* ep, void* out_p3, float* out_f32, void* stream) {
    return p3_conv2d_fwd_impl(x, N, H, W, Cin, w, Cout, KH, KW, stride, pad, ep, out_p3, out_f32, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

// The same convolution geometry on TWO (activations, weights, epilogue, output) sets as ONE launch: G's and D's ResNet trunks see the same
// backgrounds through the same architecture with different weights, and two short launches in one grid overlap each other's fill and store
// burst (the second set's blocks are dispatched behind the first's; see ldetr_p3_conv2d_bwd_pair).
extern "C" int ldetr_p3_conv2d_fwd_dual(const void* x1, const void* x2, int N, int H, int W, int Cin, const void* w1, const void* w2, int Cout, int KH, int KW,
                                        int stride, int pad, const ldetr_p3_epilogue* ep1, const ldetr_p3_epilogue* ep2, void* out1_p3, float* out1_f32,
                                        void* out2_p3, float* out2_f32, void* stream) {
    LDETR_CHECK(x2 != nullptr, "p3_conv2d_fwd_dual: null operand in the second set");
    return p3_conv2d_fwd_impl(x1, N, H, W, Cin, w1, Cout, KH, KW, stride, pad, ep1, out1_p3, out1_f32, x2, w2, ep2, out2_p3, out2_f32, stream);
}

// dx[n][iy][ix][ci] = sum dy[n][(iy + pad - kh) / stride][(ix + pad - kw) / stride][co] * wb[ci][kh][kw][co] over the taps that divide:
// the stride^2 parity classes of (iy, ix) are one launch (a class without taps still runs its epilogue, e.g. writes the residual).
extern "C" int ldetr_p3_conv2d_bwd_data(const void* dy, int N, int OH, int OW, int Cout, const void* wb, int Cin, int KH, int KW, int stride, int pad,
                                        int IH, int IW, const ldetr_p3_epilogue* ep, void* out_p3, float* out_f32, void* stream) {
    bool use_c3; P3C3Params c; P3NtParams p;
    const int rc = setup_bwd_data(dy, N, OH, OW, Cout, wb, Cin, KH, KW, stride, pad, IH, IW, ep, out_p3, out_f32, use_c3, c, p);
    if (rc != LDETR_OK) return rc;
    P3Group2 none; memset(&none, 0, sizeof(none));
    if (use_c3) return launch_c3(c, none, (hipStream_t)stream);
    if (p.M == 0) return LDETR_OK;
    return launch_nt(p, none, false, (hipStream_t)stream);
}

extern "C" int ldetr_p3_conv2d_bwd_weight(const void* x, int N, int H, int W, int Cin, const void* dy, int Cout, int KH, int KW, int stride, int pad,
                                          const float* dy_scale, float* dw, void* stream) {
    P3TnParams p;
    const int rc = setup_bwd_weight(x, N, H, W, Cin, dy, Cout, KH, KW, stride, pad, dy_scale, dw, p);
    if (rc != LDETR_OK) return rc;
    if (p.npix == 0) return LDETR_OK;
    return launch_tn(p, (hipStream_t)stream);
}

// Both gradients of one convolution (ldetr_p3_conv2d_bwd_data + ldetr_p3_conv2d_bwd_weight, same arguments) as ONE launch when the two
// kernels can share a grid (p3_bwd_pair_*_kernel), else as the two launches.  *launches (optional) receives the number of launches made.
extern "C" int ldetr_p3_conv2d_bwd_pair(const void* dy, int N, int OH, int OW, int Cout, const void* wb, const void* x, int Cin, int KH, int KW, int stride, int pad,
                                        int IH, int IW, const ldetr_p3_epilogue* ep, void* dx_p3, float* dx_f32, const float* dy_scale, float* dw,
                                        int* launches, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    bool use_c3; P3C3Params c; P3NtParams pn; P3TnParams pt;
    int rc = setup_bwd_data(dy, N, OH, OW, Cout, wb, Cin, KH, KW, stride, pad, IH, IW, ep, dx_p3, dx_f32, use_c3, c, pn);
    if (rc != LDETR_OK) return rc;
    rc = setup_bwd_weight(x, N, IH, IW, Cin, dy, Cout, KH, KW, stride, pad, dy_scale, dw, pt);
    if (rc != LDETR_OK) return rc;
    LDETR_CHECK(pt.OH == OH && pt.OW == OW, "p3_conv2d_bwd_pair: dy is %dx%d but the geometry gives %dx%d", OH, OW, pt.OH, pt.OW);
    if (launches) *launches = 0;
    if (pt.npix == 0) return LDETR_OK;
    static const int pair_on = (int)knob("P3_PAIR", 3);   // bit 0: gather kernel + weight gradient, bit 1: patch kernel + weight gradient
    static const bool forced = knob("P3_TILE", 0) || knob("P3_WTILE", 0);
    if (!forced && ((use_c3 && (pair_on & 2)) || (!use_c3 && (pair_on & 1)))) {
        plan_tn<64, 64>(pt, (int)slot_target(tn_flops(pt), 512));
        const long n_tn = (long)pt.splitk * pt.mtiles * pt.ntiles * pt.KH * pt.KW, n_tn_pad = (n_tn + 7) / 8 * 8;
        constexpr size_t lds_tn = (size_t)1 * (64 + 64) * 192;   // one LDS stage for the weight gradient's blocks: more blocks of either kind per CU
        if (use_c3) {
            const long gx = plan_c3(c);
            const size_t lds = C3_LDS > lds_tn ? (size_t)C3_LDS : lds_tn;
            static RaisedPerDevice raised;
            if (!raise_lds(reinterpret_cast<const void*>(&p3_bwd_pair_c3_kernel), lds, "p3_bwd_pair_c3", raised)) return LDETR_ERR_LAUNCH;
            hipLaunchKernelGGL(p3_bwd_pair_c3_kernel, dim3((unsigned)(n_tn_pad + gx * c.splitk)), 256, lds, st, c, pt, (int)n_tn, (int)(n_tn_pad / 8), pt.splitk, (int)gx);
            t_last = {5, 128, 64, 4, c.splitk, c.xm, c.xn, 1, pt.splitk, n_tn_pad + gx * c.splitk};
        } else {
            const NtGrid g = plan_nt<64, 64, 4, 1>(pn, nt_splitk(pn, 64, 64, slot_target(nt_flops(pn), 512)));
            const size_t lds = g.lds > lds_tn ? g.lds : lds_tn;
            if (nt_is_plain_gemm(pn))
                hipLaunchKernelGGL((p3_bwd_pair_nt_kernel<64, 64, 1, true>), dim3((unsigned)(n_tn_pad + g.gx * g.sk * g.ncls)), 256, lds, st, pn, pt, (int)n_tn, (int)(n_tn_pad / 8), pt.splitk,
                                   (int)g.gx, g.sk);
            else
                hipLaunchKernelGGL((p3_bwd_pair_nt_kernel<64, 64, 1>), dim3((unsigned)(n_tn_pad + g.gx * g.sk * g.ncls)), 256, lds, st, pn, pt, (int)n_tn, (int)(n_tn_pad / 8), pt.splitk,
                                   (int)g.gx, g.sk);
            t_last = {4, 64, 64, 4, g.sk, pn.xm, pn.xn, g.ncls, pt.splitk, n_tn_pad + g.gx * g.sk * g.ncls};
        }
        note_engine_launch(true);
        if (launches) *launches = 1;
        return check_launch("p3_bwd_pair");
    }
    rc = launch_tn(pt, st);
    if (rc != LDETR_OK) return rc;
    P3Group2 none; memset(&none, 0, sizeof(none));
    rc = use_c3 ? launch_c3(c, none, st) : (pn.M == 0 ? LDETR_OK : launch_nt(pn, none, false, st));
    if (launches) *launches = 2;
    return rc;
}
