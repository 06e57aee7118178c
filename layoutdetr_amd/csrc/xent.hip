// Label-smoothed softmax cross entropy over a vocabulary-sized last dimension (the LM text decoder's loss:
// CrossEntropyLoss(reduction='mean', label_smoothing=0.1) on the shifted prediction scores, training/med.py:911-916;
// ignore_index = -100 for padded tokens).  HBM-bound: the forward reads every logit once (online max / sum-exp), the backward
// reads it once more and writes the gradient (in place if the caller wishes); nothing of vocabulary size is kept in between
// beyond the logits themselves (row log-sum-exp only).
//   loss_i = lse_i - (1 - eps) * x_i[t_i] - eps * mean_c x_i[c]                         (rows with t_i != ignore_index)
//   loss   = sum_i loss_i / count,   d loss / d x_i[c] = (softmax_i[c] - (1 - eps) [c == t_i] - eps / V) * g / count
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct XentParams {
    const float* x; long ld;
    const long long* tgt;
    float* row_lse;
    float* loss_sum; float* count;        // forward: accumulated with atomics (caller zeroes them)
    const float* gscale;                  // backward: upstream gradient (device scalar)
    float* dx; long ldd;
    long rows; int V;
    long long ignore_index;
    float eps;
};

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
}

// Rows are dealt round-robin to the blocks and each block adds its share of the loss once: one atomic per ROW made 2 x rows
// same-address device-scope atomics (~0.1 us each on this multi-die part) the longest thing in the kernel.
__global__ __launch_bounds__(256) void xent_fwd_kernel(XentParams p) {
  float block_loss = 0.f, block_count = 0.f;
  for (long row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const float* x = p.x + row * p.ld;
    const long long t = p.tgt[row];
    const bool valid = t != p.ignore_index && t >= 0 && t < p.V;   // out-of-range labels never become addresses
    float m = -INFINITY, s = 0.f, sx = 0.f;
    if (valid) {
        const int V4 = p.V >> 2;
        for (int i = threadIdx.x; i < V4; i += 256) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            const float mv = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
            const float mn = fmaxf(m, mv);
            s = s * __expf(m - mn) + __expf(v.x - mn) + __expf(v.y - mn) + __expf(v.z - mn) + __expf(v.w - mn);
            m = mn;
            sx += (v.x + v.y) + (v.z + v.w);
        }
        for (int i = (V4 << 2) + threadIdx.x; i < p.V; i += 256) {
            const float v = x[i];
            const float mn = fmaxf(m, v);
            s = s * __expf(m - mn) + __expf(v - mn);
            m = mn; sx += v;
        }
    }
    // block reduction of (m, s) and sx
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        online_merge(m, s, m2, s2);
        sx += __shfl_xor(sx, o, 64);
    }
    __shared__ float sm[4], ss[4], sxs[4];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sm[wave] = m; ss[wave] = s; sxs[wave] = sx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = sm[0]; s = ss[0]; sx = sxs[0];
#pragma unroll
        for (int w = 1; w < 4; w++) { online_merge(m, s, sm[w], ss[w]); sx += sxs[w]; }
        if (valid) {
            const float lse = m + logf(s);
            p.row_lse[row] = lse;
            const float li = lse - (1.f - p.eps) * x[t] - p.eps * (sx / (float)p.V);
            block_loss += li; block_count += 1.f;
        } else {
            p.row_lse[row] = 0.f;
        }
    }
    __syncthreads();   // sm / ss / sxs are rewritten by the next row
  }
  if (threadIdx.x == 0 && block_count > 0.f) { atomicAdd(p.loss_sum, block_loss); atomicAdd(p.count, block_count); }
}

__global__ __launch_bounds__(256) void xent_bwd_kernel(XentParams p) {
    const long row = blockIdx.x;
    const float* x = p.x + row * p.ld;
    float* dx = p.dx + row * p.ldd;
    const long long t = p.tgt[row];
    const bool valid = t != p.ignore_index && t >= 0 && t < p.V;   // out-of-range labels never become addresses
    const float cnt = *p.count;
    const float g = (valid && cnt > 0.f) ? (*p.gscale) / cnt : 0.f;
    const float lse = p.row_lse[row];
    const float sm = p.eps / (float)p.V;
    const int V4 = p.V >> 2;
    for (int i = threadIdx.x; i < V4; i += 256) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            const float4 v = reinterpret_cast<const float4*>(x)[i];
            const int c = i << 2;
            o.x = (__expf(v.x - lse) - sm - (c + 0 == t ? 1.f - p.eps : 0.f)) * g;
            o.y = (__expf(v.y - lse) - sm - (c + 1 == t ? 1.f - p.eps : 0.f)) * g;
            o.z = (__expf(v.z - lse) - sm - (c + 2 == t ? 1.f - p.eps : 0.f)) * g;
            o.w = (__expf(v.w - lse) - sm - (c + 3 == t ? 1.f - p.eps : 0.f)) * g;
        }
        reinterpret_cast<float4*>(dx)[i] = o;
    }
    for (int i = (V4 << 2) + threadIdx.x; i < p.V; i += 256)
        dx[i] = valid ? (__expf(x[i] - lse) - sm - (i == t ? 1.f - p.eps : 0.f)) * g : 0.f;
}


// Token embedding of the LM decoder (nn.Embedding(vocab, hidden, padding_idx), training/med.py:60,88 of the reference).
// fwd: out[i, :] = W[ids[i], :] (+ pos[i % T, :] when pos is given: the position rows the BertEmbeddings forward adds right after).
// bwd: dW[ids[i], :] += dy[i, :] with fp32 atomics, rows whose id is padding_idx or out of range are skipped.  (aten's
// embedding_dense_backward sorts the ids with rocPRIM above 3072 tokens; that sort did not survive hipGraph replay on ROCm 7.2.)
struct EmbParams {
    const float* W; const float* pos; const long long* ids; float* out; const float* dy; float* dW;
    long n; int d; int V; int T; long long padding_idx;
};

__global__ __launch_bounds__(256) void embedding_fwd_kernel(EmbParams p) {
    const int d4 = p.d >> 2;
    const long total = p.n * d4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / d4; const int c = (int)(i - row * d4);
        const long long id = p.ids[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0 && id < p.V) v = reinterpret_cast<const float4*>(p.W + id * p.d)[c];
        if (p.pos) {
            const float4 q = reinterpret_cast<const float4*>(p.pos + (row % p.T) * p.d)[c];
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
        }
        reinterpret_cast<float4*>(p.out + row * p.d)[c] = v;
    }
}

__global__ __launch_bounds__(256) void embedding_bwd_kernel(EmbParams p) {
    const long total = p.n * p.d;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / p.d; const int c = (int)(i - row * p.d);
        const long long id = p.ids[row];
        if (id < 0 || id >= p.V || id == p.padding_idx) continue;
        atomicAdd(p.dW + id * p.d + c, p.dy[i]);
    }
}

}  // namespace ldetr

using namespace ldetr;

static int xent_check(const XentParams& p, const char* what) {
    LDETR_CHECK(p.x && p.tgt && p.row_lse, "%s: null pointer", what);
    LDETR_CHECK(p.rows >= 0 && p.V > 0, "%s: bad shape", what);
    LDETR_CHECK((p.ld % 4) == 0 && (((uintptr_t)p.x) & 15) == 0, "%s: logits rows must be 16-byte aligned", what);
    LDETR_CHECK(p.eps >= 0.f && p.eps < 1.f, "%s: label smoothing must be in [0, 1)", what);
    return LDETR_OK;
}

extern "C" int ldetr_softmax_xent_fwd_f32(const float* logits, int64_t ld, const int64_t* targets, float* row_lse, float* loss_sum,
                                          float* count, int64_t rows, int V, int64_t ignore_index, float label_smoothing, void* stream) {
    XentParams p; memset(&p, 0, sizeof(p));
    p.x = logits; p.ld = ld; p.tgt = (const long long*)targets; p.row_lse = row_lse; p.loss_sum = loss_sum; p.count = count;
    p.rows = rows; p.V = V; p.ignore_index = ignore_index; p.eps = label_smoothing;
    int rc = xent_check(p, "softmax_xent_fwd"); if (rc) return rc;
    LDETR_CHECK(loss_sum && count, "softmax_xent_fwd: null accumulator");
    if (rows == 0) return LDETR_OK;
    hipLaunchKernelGGL(xent_fwd_kernel, dim3((unsigned)(rows < 1024 ? rows : 1024)), 256, 0, (hipStream_t)stream, p);
    return check_launch("softmax_xent_fwd");
}

extern "C" int ldetr_softmax_xent_bwd_f32(const float* logits, int64_t ld, const int64_t* targets, const float* row_lse,
                                          const float* count, const float* grad_out, float* dlogits, int64_t ldd, int64_t rows, int V,
                                          int64_t ignore_index, float label_smoothing, void* stream) {
    XentParams p; memset(&p, 0, sizeof(p));
    p.x = logits; p.ld = ld; p.tgt = (const long long*)targets; p.row_lse = const_cast<float*>(row_lse); p.count = const_cast<float*>(count);
    p.gscale = grad_out; p.dx = dlogits; p.ldd = ldd; p.rows = rows; p.V = V; p.ignore_index = ignore_index; p.eps = label_smoothing;
    int rc = xent_check(p, "softmax_xent_bwd"); if (rc) return rc;
    LDETR_CHECK(count && grad_out && dlogits, "softmax_xent_bwd: null pointer");
    LDETR_CHECK((ldd % 4) == 0 && (((uintptr_t)dlogits) & 15) == 0, "softmax_xent_bwd: gradient rows must be 16-byte aligned");
    if (rows == 0) return LDETR_OK;
    hipLaunchKernelGGL(xent_bwd_kernel, dim3((unsigned)rows), 256, 0, (hipStream_t)stream, p);
    return check_launch("softmax_xent_bwd");
}

extern "C" int ldetr_embedding_fwd_f32(const float* weight, const float* pos, const int64_t* ids, float* out, int64_t n, int d, int V,
                                       int T, void* stream) {
    LDETR_CHECK(weight && ids && out, "embedding_fwd: null pointer");
    LDETR_CHECK(n >= 0 && d > 0 && (d % 4) == 0 && V > 0 && (!pos || T > 0), "embedding_fwd: bad shape (hidden size must be a multiple of 4)");
    LDETR_CHECK(((((uintptr_t)weight) | ((uintptr_t)out) | ((uintptr_t)pos)) & 15) == 0, "embedding_fwd: rows must be 16-byte aligned");
    if (n == 0) return LDETR_OK;
    EmbParams p; memset(&p, 0, sizeof(p));
    p.W = weight; p.pos = pos; p.ids = (const long long*)ids; p.out = out; p.n = n; p.d = d; p.V = V; p.T = T > 0 ? T : 1;
    long blocks = (n * (d / 4) + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3((unsigned)blocks), 256, 0, (hipStream_t)stream, p);
    return check_launch("embedding_fwd");
}

extern "C" int ldetr_embedding_bwd_f32(const float* dy, const int64_t* ids, float* dweight, int64_t n, int d, int V, int64_t padding_idx,
                                       void* stream) {
    LDETR_CHECK(dy && ids && dweight, "embedding_bwd: null pointer");
    LDETR_CHECK(n >= 0 && d > 0 && V > 0, "embedding_bwd: bad shape");
    if (n == 0) return LDETR_OK;
    EmbParams p; memset(&p, 0, sizeof(p));
    p.dy = dy; p.ids = (const long long*)ids; p.dW = dweight; p.n = n; p.d = d; p.V = V; p.padding_idx = padding_idx;
    long blocks = (n * (long)d + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(embedding_bwd_kernel, dim3((unsigned)blocks), 256, 0, (hipStream_t)stream, p);
    return check_launch("embedding_bwd");
}
