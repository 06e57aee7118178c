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

// Development knobs (tile / split-K / prefetch forcing of the sweeps, A/B switches of single kernels): ONE environment variable,
// LDETR_DEBUG="KEY=value,KEY=value", parsed once per process (ldetr_core.cpp); a key that is not set yields the default, which is the measured
// best.  Keys are listed in DESIGN.md ("Diagnostic switches").
long knob(const char* key, long dflt);
double knob_f(const char* key, double dflt);

// What the launch policy of the fp32-operand engine chose for the calling thread's most recent contraction launch (ldetr_engine_last_launch; defined in
// ldetr_core.cpp): kind 1 gemm_f32_kernel<BM, BN, BK, AMODE, BMODE, waves, FAST, SPLIT>, 2 gemm_small_kernel, 3 gemm_small_pair_kernel,
// 4 conv3x3_c32(_split)_kernel, 5 wgrad_c32_3x3_kernel, 6 stem_conv7x7_kernel.  The parity tests assert the template a bench shape reaches and bench.py
// labels its per-launch records with it.
void note_engine_launch(int kind, int bm, int bn, int bk, int waves, int fast, int split, int splitk, long blocks, int modes);

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

// ---- exact three-way bf16 split of fp32 operands (gemm_conv.hip's SPLIT path, conv_c32.hip): x = hi + mid + lo
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t r = {(__bf16)a, (__bf16)b};
    return *reinterpret_cast<unsigned*>(&r);
}
// 9 VALU instructions per pair: 3 packed conversions (round to nearest even), 2 x (shift, mask) to widen a part back to fp32, 2 packed
// subtractions (exact: a part is the leading bits of what it is subtracted from).  The shift is inline asm because hipcc otherwise
// re-converts the low element on its own instead of shifting the packed word (one more instruction per part).
__device__ __forceinline__ void split2_bf16(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    auto widen = [](unsigned w) {
        unsigned lo16;
        asm("v_lshlrev_b32 %0, 16, %1" : "=v"(lo16) : "v"(w));
        f32x2_t r = {__uint_as_float(lo16), __uint_as_float(w & 0xffff0000u)};
        return r;
    };
    const f32x2_t x = {x0, x1};
    h = pk_bf16(x.x, x.y);
    const f32x2_t r = x - widen(h);
    m = pk_bf16(r.x, r.y);
    const f32x2_t t = r - widen(m);
    l = pk_bf16(t.x, t.y);
}
// The split of values that are STORED (plane-format producers, p3_engine.hip): exact for EVERY fp32 value, so that hi + mid + lo gives the
// value back bit for bit.  split2_bf16 is exact for finite values whose leading part does not round up to Inf; the rest take the rare branch:
// finite values within half a bf16 ulp of the overflow threshold get a truncated hi (the remainder then has <= 16 significant bits: still
// exact in two more parts), and +-Inf / NaN are stored as (x, 0, 0) -- the plain split would store Inf - Inf = NaN in the lower planes and
// turn every Inf into NaN, a different class for training_loop.py:308's nan_to_num(nan=0, posinf=1e5, neginf=-1e5).
__device__ __forceinline__ void split2_bf16_exact(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    split2_bf16(x0, x1, h, m, l);
    if (__builtin_expect((((h & 0x7f80u) == 0x7f80u) | ((h & 0x7f800000u) == 0x7f800000u)), 0)) {
        auto fix = [](float x, unsigned& he, unsigned& me, unsigned& le) {   // 16-bit parts of one element
            if ((he & 0x7f80u) != 0x7f80u) return;
            const unsigned xb = __float_as_uint(x);
            if ((xb & 0x7f800000u) == 0x7f800000u) {
                he = (xb & 0x007fffffu) ? ((xb >> 16) | 0x0040u) : (xb >> 16);   // a NaN stays a (quiet) NaN even if its payload sits in the low bits
                me = le = 0u;
                return;
            }
            he = xb >> 16;
            const float r = x - __uint_as_float(he << 16);
            me = pk_bf16(r, 0.f) & 0xffffu;
            const float t = r - __uint_as_float(me << 16);
            le = pk_bf16(t, 0.f) & 0xffffu;
        };
        unsigned h0 = h & 0xffffu, h1 = h >> 16, m0 = m & 0xffffu, m1 = m >> 16, l0 = l & 0xffffu, l1 = l >> 16;
        fix(x0, h0, m0, l0); fix(x1, h1, m1, l1);
        h = h0 | (h1 << 16); m = m0 | (m1 << 16); l = l0 | (l1 << 16);
    }
}
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

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
