// HBM-bound helper kernels around the contraction engine: fused elementwise + per-(sample, channel)
// reductions for the modulated-conv / bias_act backward, NHWC max-pool, toRGB backward.
// All tensors are row-major [rows][C] with C contiguous (NHWC pixels are rows).
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

enum { RR_COLSUM = 0, RR_ACTGRAD = 1, RR_MULRED = 2 };

struct RRParams {
    const float* a; const float* b; float* out;
    const float* colv;   // [C]
    const float* sampv;  // [B][C]
    float* red1; long red1_bs;
    float* red2; long red2_bs;
    float* part;         // [2][blocks][C] partial sums instead of atomics (finished by rr_finish_kernel), or NULL
    long P; int B, C, mode, act, rows_pb;
    float alpha, gain;
};

// Block = 256 threads = TQ channel-quads x RL row lanes.  grid = (row chunks, B, channel chunks).
template <int MODE>
__global__ __launch_bounds__(256) void rowreduce_kernel(RRParams p) {
    __shared__ float4 sh1[256];
    __shared__ float4 sh2[256];
    const int C4 = p.C >> 2;
    const int TQ = C4 < 256 ? C4 : 256;
    const int RL = 256 / TQ;
    const int tx = threadIdx.x % TQ, ty = threadIdx.x / TQ;
    const int cq = blockIdx.z * TQ + tx;
    const int b = blockIdx.y;
    const bool active = (ty < RL) && (cq < C4);
    float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1;
    if (active) {
        const int c = cq << 2;
        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f), sv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.colv) cv = *reinterpret_cast<const float4*>(p.colv + c);
        if (p.sampv) sv = *reinterpret_cast<const float4*>(p.sampv + (long)b * p.C + c);
        const long r0 = (long)blockIdx.x * p.rows_pb;
        const long r1e = min(p.P, r0 + p.rows_pb);
        auto process = [&](const float4& a, const float4& x2, long off) {
            if (MODE == RR_COLSUM) {
                r1.x += a.x; r1.y += a.y; r1.z += a.z; r1.w += a.w;
            } else if (MODE == RR_ACTGRAD) {
                // a = dy, x2 = y (saved output).  dv = dy * gain * (y > 0 ? 1 : alpha)   [act 2: lrelu, 1: relu, 0: linear]
                const float4& y = x2;
                float4 dv;
                float gp = p.gain, gn = (p.act == 2) ? p.gain * p.alpha : (p.act == 1 ? 0.f : p.gain);
                dv.x = a.x * (y.x > 0.f ? gp : gn); dv.y = a.y * (y.y > 0.f ? gp : gn);
                dv.z = a.z * (y.z > 0.f ? gp : gn); dv.w = a.w * (y.w > 0.f ? gp : gn);
                if (p.out) *reinterpret_cast<float4*>(p.out + off) = dv;
                r1.x += dv.x; r1.y += dv.y; r1.z += dv.z; r1.w += dv.w;
                if (p.red2) {
                    // pre-activation v = y/gain (y > 0) or y/(gain*alpha); u*d = v - bias; red2 += dv * (v - bias) / d
                    float ip = 1.f / gp, in = (gn != 0.f) ? 1.f / gn : 0.f;
                    float vx = y.x * (y.x > 0.f ? ip : in) - cv.x, vy = y.y * (y.y > 0.f ? ip : in) - cv.y;
                    float vz = y.z * (y.z > 0.f ? ip : in) - cv.z, vw = y.w * (y.w > 0.f ? ip : in) - cv.w;
                    r2.x += dv.x * vx / sv.x; r2.y += dv.y * vy / sv.y; r2.z += dv.z * vz / sv.z; r2.w += dv.w * vw / sv.w;
                }
            } else {  // RR_MULRED: out = a * sampv (optional), red1 += a * b
                if (p.out) {
                    float4 o = make_float4(a.x * sv.x, a.y * sv.y, a.z * sv.z, a.w * sv.w);
                    *reinterpret_cast<float4*>(p.out + off) = o;
                }
                r1.x += a.x * x2.x; r1.y += a.y * x2.y; r1.z += a.z * x2.z; r1.w += a.w * x2.w;
            }
        };
        // four rows per trip, all loads issued before the first store (the output may alias an input, so the compiler
        // cannot hoist them itself): 8 float4 loads in flight per lane
        constexpr int UN = 4;
        for (long r = r0 + ty; r < r1e; r += (long)RL * UN) {
            float4 av[UN], bv[UN];
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const long rr = r + (long)u * RL;
                if (rr < r1e) {
                    const long off = ((long)b * p.P + rr) * p.C + c;
                    av[u] = *reinterpret_cast<const float4*>(p.a + off);
                    if (MODE != RR_COLSUM) bv[u] = *reinterpret_cast<const float4*>(p.b + off);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; u++) {
                const long rr = r + (long)u * RL;
                if (rr < r1e) process(av[u], bv[u], ((long)b * p.P + rr) * p.C + c);
            }
        }
    }
    sh1[threadIdx.x] = r1; sh2[threadIdx.x] = r2;
    __syncthreads();
    if (active && ty == 0) {
        for (int j = 1; j < RL; j++) {
            float4 t = sh1[j * TQ + tx]; r1.x += t.x; r1.y += t.y; r1.z += t.z; r1.w += t.w;
            float4 u = sh2[j * TQ + tx]; r2.x += u.x; r2.y += u.y; r2.z += u.z; r2.w += u.w;
        }
        const int c = cq << 2;
        if (p.part) {
            const long nblk = (long)gridDim.x * gridDim.y, blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
            *reinterpret_cast<float4*>(p.part + blk * p.C + c) = r1;
            if (p.red2) *reinterpret_cast<float4*>(p.part + (nblk + blk) * p.C + c) = r2;
            return;
        }
        if (p.red1) {
            float* d = p.red1 + (long)b * p.red1_bs + c;
            atomicAdd(d, r1.x); atomicAdd(d + 1, r1.y); atomicAdd(d + 2, r1.z); atomicAdd(d + 3, r1.w);
        }
        if (p.red2) {
            float* d = p.red2 + (long)b * p.red2_bs + c;
            atomicAdd(d, r2.x); atomicAdd(d + 1, r2.y); atomicAdd(d + 2, r2.z); atomicAdd(d + 3, r2.w);
        }
    }
}

// Second stage of the row reductions: red[b][c] += sum over the row chunks' partial sums (all samples' chunks when the
// reduction is shared by the batch, bs == 0).  No long same-address atomic queues: with one atomic per block and channel the
// first stage's time was (#blocks x ~0.1 us) whatever the bandwidth; here at most `segs` (<= 16) atomics meet per address.
// grid (ceil(C / 64), B or 1, 2 * segs): z / segs = 0 -> red1, 1 -> red2; z % segs = slice of the partial rows.
// Block = 64 channels x 4 partial-row lanes.
__global__ __launch_bounds__(256) void rr_finish_kernel(RRParams p, int chunks, int segs) {
    __shared__ float sh[256];
    const int which = blockIdx.z / segs, seg = blockIdx.z % segs;
    float* red = which ? p.red2 : p.red1;
    const long bs = which ? p.red2_bs : p.red1_bs;
    if (!red) return;
    const bool shared_by_batch = bs == 0;
    if (shared_by_batch && blockIdx.y > 0) return;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    const long nblk = (long)chunks * p.B;
    const float* part = p.part + (which ? nblk * p.C : 0);
    const long first = shared_by_batch ? 0 : (long)blockIdx.y * chunks, count = shared_by_batch ? nblk : chunks;
    const long per = (count + segs - 1) / segs, i0 = seg * per, i1 = min(count, i0 + per);
    float acc = 0.f;
    if (c < p.C) {
#pragma unroll 4
        for (long i = i0 + lane; i < i1; i += 4) acc += part[(first + i) * p.C + c];
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (lane == 0 && c < p.C && i0 < i1) {
        acc += sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192];
        float* d = red + (shared_by_batch ? 0 : (long)blockIdx.y * bs) + c;
        if (segs > 1) atomicAdd(d, acc); else *d += acc;
    }
}

static int launch_rr(RRParams& p, hipStream_t st) {
    const int C4 = p.C / 4;
    const int TQ = C4 < 256 ? C4 : 256;
    const int RL = 256 / TQ;
    // a few blocks per CU in total, never straddling a sample; their partial sums go through the workspace to rr_finish_kernel
    static const long target = 2048;
    const long groups = (long)p.B * cdiv(C4, TQ);
    long chunks = target / groups; if (chunks < 1) chunks = 1;
    long rows_pb = (p.P + chunks - 1) / chunks;
    const long min_rows = (long)RL * 4;
    if (rows_pb < min_rows) rows_pb = min_rows;
    rows_pb = (rows_pb + RL - 1) / RL * RL;
    p.rows_pb = (int)rows_pb;
    dim3 grid((unsigned)((p.P + rows_pb - 1) / rows_pb), p.B, cdiv(C4, TQ));
    const long nblk = (long)grid.x * grid.y;
    // atomics queue up per output cache line: 32 per block and line at ~5 ns each, i.e. ~0.16 us per sharing block -- cheaper than
    // the ~4.5 us of a second launch up to ~30 sharers (measured: 48 us vs 10 us at 256 sharers); beyond that the partial sums go
    // through the workspace to rr_finish_kernel
    const long sharers = p.red1 && p.red1_bs == 0 ? nblk : (long)grid.x;     // blocks adding into one output element
    p.part = (p.red1 || p.red2) && sharers > 32 ? scratch_alloc((size_t)nblk * p.C * sizeof(float) * (p.red2 ? 2 : 1)) : nullptr;
    if (p.mode == RR_COLSUM) hipLaunchKernelGGL(rowreduce_kernel<RR_COLSUM>, grid, 256, 0, st, p);
    else if (p.mode == RR_ACTGRAD) hipLaunchKernelGGL(rowreduce_kernel<RR_ACTGRAD>, grid, 256, 0, st, p);
    else hipLaunchKernelGGL(rowreduce_kernel<RR_MULRED>, grid, 256, 0, st, p);
    int rc = check_launch("rowreduce"); if (rc || !p.part) return rc;
    int segs = (int)(sharers / 32); if (segs < 1) segs = 1; if (segs > 16) segs = 16;
    hipLaunchKernelGGL(rr_finish_kernel, dim3(cdiv(p.C, 64), p.B, (p.red2 ? 2 : 1) * segs), 256, 0, st, p, (int)grid.x, segs);
    return check_launch("rowreduce_finish");
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 2 / pad 1 max-pool on NHWC (torchvision ResNet stem; reference: detr_backbone.py:105).
struct PoolParams {
    const float* x; float* y; unsigned char* idx; const float* dy; float* dx;
    int N, H, W, C, OH, OW;
};

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(PoolParams p) {
    const int C4 = p.C >> 2;
    const long total = (long)p.N * p.OH * p.OW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) << 2; long t = i / C4;
        int ox = (int)(t % p.OW); t /= p.OW;
        int oy = (int)(t % p.OH); int n = (int)(t / p.OH);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        uchar4 mi = make_uchar4(0, 0, 0, 0);
#pragma unroll
        for (int kh = 0; kh < 3; kh++) {
            int y = oy * 2 - 1 + kh;
            if (y < 0 || y >= p.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                int x = ox * 2 - 1 + kw;
                if (x < 0 || x >= p.W) continue;
                float4 v = *reinterpret_cast<const float4*>(p.x + (((long)n * p.H + y) * p.W + x) * p.C + c);
                unsigned char k = (unsigned char)(kh * 3 + kw);
                if (v.x > m.x || v.x != v.x) { m.x = v.x; mi.x = k; }
                if (v.y > m.y || v.y != v.y) { m.y = v.y; mi.y = k; }
                if (v.z > m.z || v.z != v.z) { m.z = v.z; mi.z = k; }
                if (v.w > m.w || v.w != v.w) { m.w = v.w; mi.w = k; }
            }
        }
        long o = (((long)n * p.OH + oy) * p.OW + ox) * p.C + c;
        *reinterpret_cast<float4*>(p.y + o) = m;
        if (p.idx) *reinterpret_cast<uchar4*>(p.idx + o) = mi;
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(PoolParams p) {
    const int C4 = p.C >> 2;
    const long total = (long)p.N * p.H * p.W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) << 2; long t = i / C4;
        int x = (int)(t % p.W); t /= p.W;
        int y = (int)(t % p.H); int n = (int)(t / p.H);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        int oy0 = y >> 1, oy1 = (y + 1) >> 1;  // windows with oy*2-1 <= y <= oy*2+1
        int ox0 = x >> 1, ox1 = (x + 1) >> 1;
        for (int oy = oy0; oy <= oy1; oy++) {
            if (oy >= p.OH) continue;
            for (int ox = ox0; ox <= ox1; ox++) {
                if (ox >= p.OW) continue;
                unsigned char k = (unsigned char)((y - (oy * 2 - 1)) * 3 + (x - (ox * 2 - 1)));
                long o = (((long)n * p.OH + oy) * p.OW + ox) * p.C + c;
                uchar4 mi = *reinterpret_cast<const uchar4*>(p.idx + o);
                float4 d = *reinterpret_cast<const float4*>(p.dy + o);
                if (mi.x == k) g.x += d.x;
                if (mi.y == k) g.y += d.y;
                if (mi.z == k) g.z += d.z;
                if (mi.w == k) g.w += d.w;
            }
        }
        *reinterpret_cast<float4*>(p.dx + (((long)n * p.H + y) * p.W + x) * p.C + c) = g;
    }
}

// ---------------------------------------------------------------------------------------------
// toRGB (1x1 modulated conv to 3 channels, no demodulation) backward:
//   dx[b,p,c]     = s[b,c] * sum_co dy[b,p,co] * w[co,c]
//   dws[b,co,c]  += sum_p dy[b,p,co] * x[b,p,c]          (per-sample; host folds in styles / weights)
//   dbias[co]    += sum_{b,p} dy[b,p,co]
struct RgbParams {
    const float* x; const float* dy; const float* w; const float* s;
    float* dx; float* dws; float* dbias;
    long P; int B, C, rows_pb;
};

__global__ __launch_bounds__(256) void torgb_bwd_kernel(RgbParams p) {
    __shared__ float4 sh[3][256];
    const int C4 = p.C >> 2;
    const int TQ = C4 < 256 ? C4 : 256;
    const int RL = 256 / TQ;
    const int tx = threadIdx.x % TQ, ty = threadIdx.x / TQ;
    const int cq = blockIdx.z * TQ + tx;
    const int b = blockIdx.y;
    const bool active = (ty < RL) && (cq < C4);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    float db0 = 0.f, db1 = 0.f, db2 = 0.f;
    if (active) {
        const int c = cq << 2;
        const float4 w0 = *reinterpret_cast<const float4*>(p.w + c);
        const float4 w1 = *reinterpret_cast<const float4*>(p.w + p.C + c);
        const float4 w2 = *reinterpret_cast<const float4*>(p.w + 2 * p.C + c);
        const float4 sv = *reinterpret_cast<const float4*>(p.s + (long)b * p.C + c);
        const long r0 = (long)blockIdx.x * p.rows_pb;
        const long r1e = min(p.P, r0 + p.rows_pb);
        for (long r = r0 + ty; r < r1e; r += RL) {
            const long row = (long)b * p.P + r;
            const float d0 = p.dy[row * 3], d1 = p.dy[row * 3 + 1], d2 = p.dy[row * 3 + 2];
            const float4 xv = *reinterpret_cast<const float4*>(p.x + row * p.C + c);
            float4 o;
            o.x = sv.x * (d0 * w0.x + d1 * w1.x + d2 * w2.x); o.y = sv.y * (d0 * w0.y + d1 * w1.y + d2 * w2.y);
            o.z = sv.z * (d0 * w0.z + d1 * w1.z + d2 * w2.z); o.w = sv.w * (d0 * w0.w + d1 * w1.w + d2 * w2.w);
            *reinterpret_cast<float4*>(p.dx + row * p.C + c) = o;
            a0.x += d0 * xv.x; a0.y += d0 * xv.y; a0.z += d0 * xv.z; a0.w += d0 * xv.w;
            a1.x += d1 * xv.x; a1.y += d1 * xv.y; a1.z += d1 * xv.z; a1.w += d1 * xv.w;
            a2.x += d2 * xv.x; a2.y += d2 * xv.y; a2.z += d2 * xv.z; a2.w += d2 * xv.w;
            if (cq == 0) { db0 += d0; db1 += d1; db2 += d2; }
        }
    }
    sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = a2;
    __syncthreads();
    if (active && ty == 0) {
        for (int j = 1; j < RL; j++) {
            float4 t0 = sh[0][j * TQ + tx], t1 = sh[1][j * TQ + tx], t2 = sh[2][j * TQ + tx];
            a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
            a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
            a2.x += t2.x; a2.y += t2.y; a2.z += t2.z; a2.w += t2.w;
        }
        const int c = cq << 2;
        float* d = p.dws + (long)b * 3 * p.C + c;
        atomicAdd(d, a0.x); atomicAdd(d + 1, a0.y); atomicAdd(d + 2, a0.z); atomicAdd(d + 3, a0.w);
        d += p.C;
        atomicAdd(d, a1.x); atomicAdd(d + 1, a1.y); atomicAdd(d + 2, a1.z); atomicAdd(d + 3, a1.w);
        d += p.C;
        atomicAdd(d, a2.x); atomicAdd(d + 1, a2.y); atomicAdd(d + 2, a2.z); atomicAdd(d + 3, a2.w);
    }
    // bias gradient: the block's row lanes are summed through LDS first, so one block issues 3 atomics instead of 3 x RL onto the
    // same cache line (at 256^2 x 16 samples that queue of 65536 same-line atomics WAS the kernel: 340 us)
    if (p.dbias && blockIdx.z == 0) {
        __syncthreads();
        float* f = reinterpret_cast<float*>(&sh[0][0]);
        if (tx == 0 && ty < RL) { f[ty * 3] = db0; f[ty * 3 + 1] = db1; f[ty * 3 + 2] = db2; }
        __syncthreads();
        if (threadIdx.x < 3) {
            float t = 0.f;
            for (int j = 0; j < RL; j++) t += f[j * 3 + threadIdx.x];
            atomicAdd(p.dbias + threadIdx.x, t);
        }
    }
}

// Second stage of the toRGB backward: dws [B][3][C] = per-sample sums of dy x (x unmodulated) ->
//   dw[o][c] += sum_b dws[b][o][c] * s[b][c]   (accumulated: dw is the flat .grad view or a zero-filled temporary)
//   ds[b][c]  = sum_o dws[b][o][c] * w[o][c]
// (the reference's autograd forms both with broadcast multiplies and reductions: 4 launches + 2 accumulations per layer)
__global__ __launch_bounds__(256) void torgb_finish_kernel(const float* __restrict__ dws, const float* __restrict__ s, const float* __restrict__ w,
                                                           float* __restrict__ dw, float* __restrict__ ds, int B, int C) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3 * C) {
        const int c = i % C;
        float a = 0.f;
        for (int b = 0; b < B; b++) a += dws[(long)b * 3 * C + i] * s[(long)b * C + c];
        dw[i] += a;
    }
    if (i < B * C) {
        const int b = i / C, c = i - b * C;
        const float* d = dws + (long)b * 3 * C + c;
        ds[i] = d[0] * w[c] + d[C] * w[C + c] + d[2 * C] * w[2 * C + c];
    }
}

}  // namespace ldetr

using namespace ldetr;

static int rr_common(RRParams& p, const char* what) {
    LDETR_CHECK(p.a, "%s: null pointer", what);
    LDETR_CHECK(p.C % 4 == 0 && p.C > 0, "%s: C must be a positive multiple of 4", what);
    LDETR_CHECK(p.B > 0 && p.P >= 0, "%s: bad shape", what);
    return LDETR_OK;
}

// red[b][c] += sum_p a[b,p,c]   (B = 1 gives a plain column sum).  red must be pre-zeroed by the caller.
extern "C" int ldetr_colsum_f32(const float* a, float* red, int B, int64_t P, int C, void* stream) {
    RRParams p; memset(&p, 0, sizeof(p));
    p.a = a; p.red1 = red; p.red1_bs = C; p.B = B; p.P = P; p.C = C; p.mode = RR_COLSUM;
    int rc = rr_common(p, "colsum"); if (rc) return rc;
    if (P == 0) return LDETR_OK;
    return launch_rr(p, (hipStream_t)stream);
}

// bias_act first-derivative fused with its reductions (reference: bias_act.py:166-171 does dx then dx.sum()):
//   dv = dy * gain * (y > 0 ? 1 : alpha)      act: 0 linear, 1 relu, 2 lrelu
//   dbias[c]     += sum_{b,p} dv
//   ddemod[b][c] += sum_p dv * (pre(y) - bias[c]) / demod[b][c]      (only when ddemod != NULL)
extern "C" int ldetr_act_bwd_reduce_f32(const float* dy, const float* y, float* dv, const float* bias, const float* demod,
                                        float* dbias, float* ddemod, int B, int64_t P, int C, int act, float alpha, float gain,
                                        void* stream) {
    RRParams p; memset(&p, 0, sizeof(p));
    p.a = dy; p.b = y; p.out = dv; p.colv = bias; p.sampv = demod; p.red1 = dbias; p.red1_bs = 0;
    p.red2 = ddemod; p.red2_bs = C; p.B = B; p.P = P; p.C = C; p.mode = RR_ACTGRAD; p.act = act; p.alpha = alpha; p.gain = gain;
    int rc = rr_common(p, "act_bwd_reduce"); if (rc) return rc;
    LDETR_CHECK(y, "act_bwd_reduce: y is required");
    LDETR_CHECK(!ddemod || demod, "act_bwd_reduce: ddemod needs demod");
    if (P == 0) return LDETR_OK;
    return launch_rr(p, (hipStream_t)stream);
}

//   out[b,p,c] = a[b,p,c] * scale[b][c]   (skipped when out == NULL)
//   red[b][c] += sum_p a[b,p,c] * x[b,p,c]
extern "C" int ldetr_mul_reduce_f32(const float* a, const float* x, const float* scale, float* out, float* red,
                                    int B, int64_t P, int C, void* stream) {
    RRParams p; memset(&p, 0, sizeof(p));
    p.a = a; p.b = x; p.out = out; p.sampv = scale; p.red1 = red; p.red1_bs = C; p.B = B; p.P = P; p.C = C; p.mode = RR_MULRED;
    int rc = rr_common(p, "mul_reduce"); if (rc) return rc;
    LDETR_CHECK(x, "mul_reduce: x is required");
    if (P == 0) return LDETR_OK;
    return launch_rr(p, (hipStream_t)stream);
}

extern "C" int ldetr_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C, void* stream) {
    LDETR_CHECK(x && y, "maxpool_fwd: null pointer");
    LDETR_CHECK(C % 4 == 0, "maxpool_fwd: C must be a multiple of 4");
    PoolParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.y = y; p.idx = idx; p.N = N; p.H = H; p.W = W; p.C = C; p.OH = (H + 2 - 3) / 2 + 1; p.OW = (W + 2 - 3) / 2 + 1;
    long total = (long)N * p.OH * p.OW * (C / 4);
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(maxpool_fwd_kernel, grid, 256, 0, (hipStream_t)stream, p);
    return check_launch("maxpool_fwd");
}

extern "C" int ldetr_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C, void* stream) {
    LDETR_CHECK(dy && idx && dx, "maxpool_bwd: null pointer");
    LDETR_CHECK(C % 4 == 0, "maxpool_bwd: C must be a multiple of 4");
    PoolParams p; memset(&p, 0, sizeof(p));
    p.dy = dy; p.idx = const_cast<unsigned char*>(idx); p.dx = dx; p.N = N; p.H = H; p.W = W; p.C = C;
    p.OH = (H + 2 - 3) / 2 + 1; p.OW = (W + 2 - 3) / 2 + 1;
    long total = (long)N * H * W * (C / 4);
    int grid = (int)((total + 255) / 256); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(maxpool_bwd_kernel, grid, 256, 0, (hipStream_t)stream, p);
    return check_launch("maxpool_bwd");
}

namespace ldetr {
// ToRGBLayer forward (networks_stylegan2.py:349-353: modulated 1x1 convolution to 3 colour channels, no demodulation, + bias):
//   y[b][p][o] = bias[o] + sum_c x[b][p][c] * w[o][c] * s[b][c]
// One read of x, 12 bytes written per pixel: HBM-bound.  LPP = min(C / 4, 64) lanes share a pixel (a float4 of channels each, C / 4 / LPP
// trips), their three partial dot products meet through xor shuffles.  On the contraction engine this launch pads 3 output channels to a
// 64-wide tile: 144 us at 16 x 256 x 256 x 32 for 134 MB.
struct RgbFwdParams { const float* x; const float* w; const float* s; const float* bias; float* y; long P; int B, C; };

template <int LPP>
__global__ __launch_bounds__(256) void torgb_fwd_kernel(RgbFwdParams p) {
    const int lane_in = threadIdx.x % LPP;
    const long pix_per_block = 256 / LPP;
    const int C4 = p.C >> 2;
    const int b = blockIdx.y;
    const float* sb = p.s + (long)b * p.C;
    // the (sample-scaled) weights this lane needs, for up to two trips (C <= 512)
    float4 w0[2], w1[2], w2[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int cq = lane_in + t * LPP;
        if (cq < C4) {
            const float4 sv = *reinterpret_cast<const float4*>(sb + 4 * cq);
            float4 a = *reinterpret_cast<const float4*>(p.w + 4 * cq), bq = *reinterpret_cast<const float4*>(p.w + p.C + 4 * cq), c = *reinterpret_cast<const float4*>(p.w + 2 * p.C + 4 * cq);
            w0[t] = make_float4(a.x * sv.x, a.y * sv.y, a.z * sv.z, a.w * sv.w);
            w1[t] = make_float4(bq.x * sv.x, bq.y * sv.y, bq.z * sv.z, bq.w * sv.w);
            w2[t] = make_float4(c.x * sv.x, c.y * sv.y, c.z * sv.z, c.w * sv.w);
        } else { w0[t] = w1[t] = w2[t] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    const float b0 = p.bias ? p.bias[0] : 0.f, b1 = p.bias ? p.bias[1] : 0.f, b2 = p.bias ? p.bias[2] : 0.f;
    const long stride = (long)gridDim.x * pix_per_block;
    for (long r = (long)blockIdx.x * pix_per_block + threadIdx.x / LPP; r < p.P; r += stride) {
        const float* xr = p.x + ((long)b * p.P + r) * p.C;
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int cq = lane_in + t * LPP;
            if (cq < C4) {
                const float4 xv = *reinterpret_cast<const float4*>(xr + 4 * cq);
                d0 += xv.x * w0[t].x + xv.y * w0[t].y + xv.z * w0[t].z + xv.w * w0[t].w;
                d1 += xv.x * w1[t].x + xv.y * w1[t].y + xv.z * w1[t].z + xv.w * w1[t].w;
                d2 += xv.x * w2[t].x + xv.y * w2[t].y + xv.z * w2[t].z + xv.w * w2[t].w;
            }
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) { d0 += __shfl_xor(d0, o, 64); d1 += __shfl_xor(d1, o, 64); d2 += __shfl_xor(d2, o, 64); }
        if (lane_in == 0) {
            float* yr = p.y + ((long)b * p.P + r) * 3;
            yr[0] = d0 + b0; yr[1] = d1 + b1; yr[2] = d2 + b2;
        }
    }
}
}  // namespace ldetr

// x [B][P][C] (NHWC pixels), w [3][C], styles [B][C], bias [3] or NULL -> y [B][P][3].  C a multiple of 4 up to 512, a power of two in quads.
extern "C" int ldetr_torgb_fwd_f32(const float* x, const float* w, const float* styles, const float* bias, float* y,
                                   int B, int64_t P, int C, void* stream) {
    LDETR_CHECK(x && w && styles && y, "torgb_fwd: null pointer");
    const int C4 = C / 4;
    LDETR_CHECK(C % 4 == 0 && C4 >= 1 && C <= 512 && (C4 & (C4 - 1)) == 0, "torgb_fwd: C must be 4 x a power of two, at most 512 (got %d)", C);
    LDETR_CHECK((((uintptr_t)x | (uintptr_t)w | (uintptr_t)styles) & 15) == 0, "torgb_fwd: buffers must be 16-byte aligned");
    if (B == 0 || P == 0) return LDETR_OK;
    RgbFwdParams p; p.x = x; p.w = w; p.s = styles; p.bias = bias; p.y = y; p.P = P; p.B = B; p.C = C;
    const int lpp = C4 < 64 ? C4 : 64;
    const long ppb = 256 / lpp;
    long gx = (P + ppb - 1) / ppb; if (gx > 4096) gx = 4096;
    const dim3 grid((unsigned)gx, (unsigned)B);
    hipStream_t st = (hipStream_t)stream;
    switch (lpp) {
        case 1: hipLaunchKernelGGL(torgb_fwd_kernel<1>, grid, 256, 0, st, p); break;
        case 2: hipLaunchKernelGGL(torgb_fwd_kernel<2>, grid, 256, 0, st, p); break;
        case 4: hipLaunchKernelGGL(torgb_fwd_kernel<4>, grid, 256, 0, st, p); break;
        case 8: hipLaunchKernelGGL(torgb_fwd_kernel<8>, grid, 256, 0, st, p); break;
        case 16: hipLaunchKernelGGL(torgb_fwd_kernel<16>, grid, 256, 0, st, p); break;
        case 32: hipLaunchKernelGGL(torgb_fwd_kernel<32>, grid, 256, 0, st, p); break;
        default: hipLaunchKernelGGL(torgb_fwd_kernel<64>, grid, 256, 0, st, p); break;
    }
    return check_launch("torgb_fwd");
}

// dws [B][3][C] and dbias [3] must be zeroed by the caller.
extern "C" int ldetr_torgb_bwd_f32(const float* x, const float* dy, const float* w, const float* styles,
                                   float* dx, float* dws, float* dbias, int B, int64_t P, int C, void* stream) {
    LDETR_CHECK(x && dy && w && styles && dx && dws, "torgb_bwd: null pointer");
    LDETR_CHECK(C % 4 == 0, "torgb_bwd: C must be a multiple of 4");
    RgbParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.dy = dy; p.w = w; p.s = styles; p.dx = dx; p.dws = dws; p.dbias = dbias; p.B = B; p.P = P; p.C = C;
    const int C4 = C / 4; const int TQ = C4 < 256 ? C4 : 256; const int RL = 256 / TQ;
    long rows_pb = (long)RL * 16;
    long nb = ((P + rows_pb - 1) / rows_pb) * B * cdiv(C4, TQ);
    while (nb > 2048) { rows_pb *= 2; nb = ((P + rows_pb - 1) / rows_pb) * B * cdiv(C4, TQ); }
    p.rows_pb = (int)rows_pb;
    dim3 grid((unsigned)((P + rows_pb - 1) / rows_pb), B, cdiv(C4, TQ));
    hipLaunchKernelGGL(torgb_bwd_kernel, grid, 256, 0, (hipStream_t)stream, p);
    return check_launch("torgb_bwd");
}

extern "C" int ldetr_torgb_bwd_finish_f32(const float* dws, const float* s, const float* w, int B, int C, float* dw, float* ds, void* stream) {
    LDETR_CHECK(dws && s && w && dw && ds && B > 0 && C > 0, "torgb_bwd_finish: bad arguments");
    const int n = (B > 3 ? B : 3) * C;
    hipLaunchKernelGGL(torgb_finish_kernel, dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream, dws, s, w, dw, ds, B, C);
    return check_launch("torgb_bwd_finish");
}
