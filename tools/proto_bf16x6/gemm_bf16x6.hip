// Prototype (development aid, not part of the product library): fp32 GEMM C[M,N] = A[M,K] . B[N,K]^T computed on the bf16 matrix
// pipe with a three-way operand split x = hi + mid + lo (8 + 8 + 8 significant bits: the three bf16 values carry the whole fp32
// significand) and the six products whose weight is >= 2^-18: hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid.  bf16 x bf16 products
// are exact in the fp32 accumulator, the dropped terms are <= 2^-26 relative, so the result is fp32-equivalent while the matrix
// pipe runs 6 bf16 MFMAs (v_mfma_f32_32x32x16_bf16, 8 passes) where the f32 path runs 8 f32 MFMAs (v_mfma_f32_32x32x2_f32,
// 16 passes) for the same k16 slab: 2.67x less matrix-pipe time.  The question this prototype answers is whether the operand
// split (VALU) and the 1.5x LDS traffic leave any of that.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 16, NT = 256;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2 r = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<unsigned*>(&r);
}

// 4 floats -> 4 bf16 of each part (element 0 in the low half of .x)
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    hi.x = pk_bf16(v.x, v.y); hi.y = pk_bf16(v.z, v.w);
    const float r0 = v.x - __uint_as_float(hi.x << 16), r1 = v.y - __uint_as_float(hi.x & 0xffff0000u);
    const float r2 = v.z - __uint_as_float(hi.y << 16), r3 = v.w - __uint_as_float(hi.y & 0xffff0000u);
    mid.x = pk_bf16(r0, r1); mid.y = pk_bf16(r2, r3);
    const float s0 = r0 - __uint_as_float(mid.x << 16), s1 = r1 - __uint_as_float(mid.x & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(mid.y << 16), s3 = r3 - __uint_as_float(mid.y & 0xffff0000u);
    lo.x = pk_bf16(s0, s1); lo.y = pk_bf16(s2, s3);
}

template <int NPROD>
__global__ __launch_bounds__(NT) void gemm_bf16_split_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B, long ldb,
                                                             float* __restrict__ C, long ldc, int M, int N, int K) {
    // [buffer][part][k-block of 8][row][8 bf16]: a lane's MFMA operand (8 consecutive k of one row) is one 16-byte LDS read and
    // the 32 lanes of a k-block read 512 contiguous bytes
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][3][BK / 8][BM][8];
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][3][BK / 8][BN][8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int lrow = tid >> 2, lk4 = tid & 3;     // loader: rows lrow and lrow + 64, floats 4 lk4 .. 4 lk4 + 3 of the k-tile
    float4 ra[2], rb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = lrow + 64 * i, k = k0 + 4 * lk4;
            ra[i] = (m0 + r < M && k < K) ? *reinterpret_cast<const float4*>(A + (long)(m0 + r) * lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = (n0 + r < N && k < K) ? *reinterpret_cast<const float4*>(B + (long)(n0 + r) * ldb + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = lrow + 64 * i;
            uint2 h, m, l;
            split4(ra[i], h, m, l);
            *reinterpret_cast<uint2*>(&sA[buf][0][lk4 >> 1][r][(lk4 & 1) * 4]) = h;
            *reinterpret_cast<uint2*>(&sA[buf][1][lk4 >> 1][r][(lk4 & 1) * 4]) = m;
            *reinterpret_cast<uint2*>(&sA[buf][2][lk4 >> 1][r][(lk4 & 1) * 4]) = l;
            split4(rb[i], h, m, l);
            *reinterpret_cast<uint2*>(&sB[buf][0][lk4 >> 1][r][(lk4 & 1) * 4]) = h;
            *reinterpret_cast<uint2*>(&sB[buf][1][lk4 >> 1][r][(lk4 & 1) * 4]) = m;
            *reinterpret_cast<uint2*>(&sB[buf][2][lk4 >> 1][r][(lk4 & 1) * 4]) = l;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    const int kb = lane >> 5, cl = lane & 31;
    auto compute = [&](int buf) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int p = 0; p < 3; p++) {
                a[i][p] = *reinterpret_cast<const bf16x8*>(&sA[buf][p][kb][wm * 64 + i * 32 + cl][0]);
                b[i][p] = *reinterpret_cast<const bf16x8*>(&sB[buf][p][kb][wn * 64 + i * 32 + cl][0]);
            }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                // smallest terms first
                if (NPROD >= 6) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
                }
                if (NPROD >= 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
            }
    };
    const int nk = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        compute(buf);
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    // C layout of the 32x32 MFMA: lane -> column cl, rows 8 (r / 4) + 4 kb + r % 4
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int col = n0 + wn * 64 + j * 32 + cl;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = m0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * kb + (r & 3);
                if (row < M && col < N) C[(long)row * ldc + col] = acc[i][j][r];
            }
        }
}

extern "C" int proto_gemm_bf16_split(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K, int nprod, void* stream) {
    if (K % 4 || lda % 4 || ldb % 4) return 1;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    hipStream_t st = (hipStream_t)stream;
    if (nprod >= 6) hipLaunchKernelGGL((gemm_bf16_split_kernel<6>), grid, NT, 0, st, A, lda, B, ldb, C, ldc, M, N, K);
    else if (nprod >= 3) hipLaunchKernelGGL((gemm_bf16_split_kernel<3>), grid, NT, 0, st, A, lda, B, ldb, C, ldc, M, N, K);
    else hipLaunchKernelGGL((gemm_bf16_split_kernel<1>), grid, NT, 0, st, A, lda, B, ldb, C, ldc, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
