// Batched small matrix product over strided operand views: C[b1, b2, m, n] = alpha * sum_k A[b1, b2, m, k] * B[b1, b2, k, n].
//
// Used by hip/composite.py only -- the regulariser phases of training/loss.py (R1: loss.py:207-215, path length: loss.py:119-142), which
// differentiate the transformer stacks twice (`create_graph=True`).  There the per-(sample, head) products of the attention
// (Q K^T, P V and the products of their first derivatives) are autograd nodes whose backward is this same product on transposed views,
// so the operands are addressed through four strides each -- (sample, head, row, column) of a [B*L, heads*dh] activation -- and a
// transpose costs nothing.  Sizes on that path: M, N <= 64 (256 at 512^2 backgrounds), K = 32 .. 256, B*heads <= 256 problems:
// latency-bound; one wave per 16 x 16 output tile, the K range walked in 16-wide steps through registers (A row / B column values are
// shared across the tile by LDS), fp32 FMAs in k order (deterministic).  The hot path's attention runs in csrc/attention.hip /
// csrc/mha_small.hip (MFMA); this kernel is never on it.
#include "ldetr_common.hpp"
#include "../../include/ldetr_hip.h"

namespace ldetr {

struct BmmParams {
    const float* A; const float* B; float* C;
    long sa[4], sb[4], sc[4];     // (b1, b2, row, col) element strides
    int nb1, nb2, M, N, K;
    float alpha;
};

__global__ __launch_bounds__(256) void bmm_strided_kernel(BmmParams p) {
    __shared__ float As[16][17], Bs[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_n = (p.N + 15) >> 4;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int b2 = blockIdx.y, b1 = blockIdx.z;
    const float* A = p.A + (long)b1 * p.sa[0] + (long)b2 * p.sa[1];
    const float* B = p.B + (long)b1 * p.sb[0] + (long)b2 * p.sb[1];
    const int m = tm * 16 + ty, n = tn * 16 + tx;
    float acc = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        // A tile [16 rows m][16 k]: thread (ty, tx) loads A[m = tm*16 + ty][k0 + tx]; B tile [16 k][16 n]: B[k0 + ty][n = tn*16 + tx]
        const int ka = k0 + tx, kb = k0 + ty;
        As[ty][tx] = (m < p.M && ka < p.K) ? A[(long)m * p.sa[2] + (long)ka * p.sa[3]] : 0.f;
        Bs[ty][tx] = (kb < p.K && n < p.N) ? B[(long)kb * p.sb[2] + (long)n * p.sb[3]] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = fmaf(As[ty][k], Bs[k][tx], acc);
        __syncthreads();
    }
    if (m < p.M && n < p.N) p.C[(long)b1 * p.sc[0] + (long)b2 * p.sc[1] + (long)m * p.sc[2] + (long)n * p.sc[3]] = p.alpha * acc;
}

}  // namespace ldetr

extern "C" int ldetr_bmm_strided_f32(const float* A, const int64_t* sa, const float* B, const int64_t* sb, float* C, const int64_t* sc,
                                     int nb1, int nb2, int M, int N, int K, float alpha, void* stream) {
    using namespace ldetr;
    LDETR_CHECK(A && B && C && sa && sb && sc, "bmm_strided: operands and stride arrays must be non-null");
    LDETR_CHECK(nb1 >= 0 && nb2 >= 0 && M >= 0 && N >= 0 && K >= 0, "bmm_strided: negative size");
    if ((long)nb1 * nb2 * M * N == 0) return LDETR_OK;
    LDETR_CHECK(nb1 <= 65535 && nb2 <= 65535, "bmm_strided: at most 65535 problems per batch axis");
    BmmParams p;
    p.A = A; p.B = B; p.C = C;
    for (int i = 0; i < 4; ++i) { p.sa[i] = sa[i]; p.sb[i] = sb[i]; p.sc[i] = sc[i]; }
    p.nb1 = nb1; p.nb2 = nb2; p.M = M; p.N = N; p.K = K; p.alpha = alpha;
    const long tiles = (long)cdiv(M, 16) * cdiv(N, 16);
    LDETR_CHECK(tiles <= 0x7fffffffL, "bmm_strided: too many tiles");
    hipLaunchKernelGGL(bmm_strided_kernel, dim3((unsigned)tiles, nb2, nb1), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("bmm_strided");
}
