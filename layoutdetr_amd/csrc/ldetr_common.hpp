// Shared helpers for the layoutdetr_amd gfx950 kernels.
// CDNA4 only: 64-lane wavefronts, f32 MFMA, no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define LDETR_OK 0
#define LDETR_ERR_ARG 1
#define LDETR_ERR_LAUNCH 2
#define LDETR_ERR_UNSUPPORTED 3

namespace ldetr {

void set_error(const char* fmt, ...);

// Slice of the caller-registered workspace (ldetr_set_workspace) for the launches enqueued next; nullptr if unavailable.
// Defined in gemm_conv.hip next to the split-K ring it shares.
float* scratch_alloc(size_t bytes);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LDETR_ERR_LAUNCH;
    }
    return LDETR_OK;
}

#define LDETR_CHECK(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            ldetr::set_error(__VA_ARGS__);        \
            return LDETR_ERR_ARG;                 \
        }                                         \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Counter-based RNG over (seed, element index): the same pair is evaluated in forward and backward, so dropout masks are
// regenerated rather than stored.  32-bit mixing (murmur3 finaliser over the folded index, keyed by both seed halves): two
// quarter-rate 32-bit multiplies per element instead of the 64-bit multiplies of a splitmix64 finaliser, which cost ~200 cycles
// per element and were half of the VALU work of the attention kernels (VALU issue is not hidden behind the matrix pipe).
__device__ __forceinline__ uint32_t rng_bits(uint64_t seed, uint64_t idx) {
    uint32_t x = (uint32_t)idx ^ (uint32_t)seed;
    x *= 0xcc9e2d51u;
    x = (x << 15) | (x >> 17);
    x ^= (uint32_t)(seed >> 32) + (uint32_t)(idx >> 32) * 0x1b873593u;
    x ^= x >> 16; x *= 0x85ebca6bu;
    x ^= x >> 13; x *= 0xc2b2ae35u;
    x ^= x >> 16;
    return x;
}
// keep-scale for dropout: 0 if dropped else 1/(1-p). p_drop in [0,1).
__device__ __forceinline__ float drop_scale(uint64_t seed, uint64_t idx, float p_drop, float inv_keep) {
    // 24-bit uniform in [0,1)
    float u = (float)(rng_bits(seed, idx) >> 8) * (1.0f / 16777216.0f);
    return (u >= p_drop) ? inv_keep : 0.0f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace ldetr
