// Probe for VERDICT r5 item 7: what bounds the optimiser's streaming rate at D's 90 M parameters (0.73 of 8 TB/s vs 0.84 at 45 M)?
// Adam-shaped traffic (read p, g, m, v; write p, m, v = 28 B / parameter) under different block -> address mappings, and the EMA shape
// (read p, pe; write pe = 12 B).  hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe [n]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void adam4(f32x4& p, f32x4 g, f32x4& m, f32x4& v) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m[i] = 0.0f * m[i] + g[i];
        v[i] = 0.99f * v[i] + 0.01f * g[i] * g[i];
        p[i] -= 1e-5f * (m[i] / (sqrtf(v[i]) * 10.f + 1e-8f));
    }
}
template <int MAP, bool NT>
__global__ __launch_bounds__(256) void adam_probe(float* p, const float* g, float* m, float* v, long n4, long chunk4) {
    // MAP 0: grid-stride.  MAP 1: XCD-contiguous (block b runs on XCD b % 8: each XCD streams its own eighth).  MAP 2: each block owns contiguous chunks of chunk4 float4.
    long start, stride, end = n4;
    if (MAP == 0) { start = (long)blockIdx.x * 256 + threadIdx.x; stride = (long)gridDim.x * 256; }
    else if (MAP == 1) {
        const long per = (n4 + 7) / 8; const int xcd = blockIdx.x & 7; const long lb = blockIdx.x >> 3, nb = gridDim.x >> 3;
        start = xcd * per + lb * 256 + threadIdx.x; stride = nb * 256; end = (xcd + 1) * per < n4 ? (xcd + 1) * per : n4;
    } else { start = 0; stride = 0; }
    if (MAP != 2) {
        for (long i = start; i < end; i += stride) {
            f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
            f32x4 gv = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i) : reinterpret_cast<const f32x4*>(g)[i];
            f32x4 mv = NT ? __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i) : reinterpret_cast<f32x4*>(m)[i];
            f32x4 vv = NT ? __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i) : reinterpret_cast<f32x4*>(v)[i];
            adam4(pv, gv, mv, vv);
            reinterpret_cast<f32x4*>(p)[i] = pv;
            if (NT) { __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + i); __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i); }
            else { reinterpret_cast<f32x4*>(m)[i] = mv; reinterpret_cast<f32x4*>(v)[i] = vv; }
        }
    } else {
        for (long c = (long)blockIdx.x * chunk4; c < n4; c += (long)gridDim.x * chunk4) {
            const long ce = c + chunk4 < n4 ? c + chunk4 : n4;
            for (long i = c + threadIdx.x; i < ce; i += 256) {
                f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
                f32x4 gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
                f32x4 mv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i);
                f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i);
                adam4(pv, gv, mv, vv);
                reinterpret_cast<f32x4*>(p)[i] = pv;
                __builtin_nontemporal_store(mv, reinterpret_cast<f32x4*>(m) + i); __builtin_nontemporal_store(vv, reinterpret_cast<f32x4*>(v) + i);
            }
        }
    }
}
template <int U>   // U float4 per thread in flight before the first use
__global__ __launch_bounds__(256) void adam_unrolled(float* p, const float* g, float* m, float* v, long n4) {
    const long stride = (long)gridDim.x * 256;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += stride * U) {
        f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const long i = i0 + u * stride; if (i < n4) { pv[u] = reinterpret_cast<f32x4*>(p)[i]; gv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
            mv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(m) + i); vv[u] = __builtin_nontemporal_load(reinterpret_cast<f32x4*>(v) + i); } }
#pragma unroll
        for (int u = 0; u < U; u++) { const long i = i0 + u * stride; if (i < n4) { adam4(pv[u], gv[u], mv[u], vv[u]); reinterpret_cast<f32x4*>(p)[i] = pv[u];
            __builtin_nontemporal_store(mv[u], reinterpret_cast<f32x4*>(m) + i); __builtin_nontemporal_store(vv[u], reinterpret_cast<f32x4*>(v) + i); } }
    }
}
__global__ __launch_bounds__(256) void copy_probe(const float* a, float* b, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a) + i), reinterpret_cast<f32x4*>(b) + i);
}
template <typename F> static float timeit(F f, int reps = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; i++) f();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 89928156L; const long n4 = n / 4;
    float *p, *g, *m, *v; size_t by = (size_t)n * 4;
    CK(hipMalloc(&p, by)); CK(hipMalloc(&g, by)); CK(hipMalloc(&m, by)); CK(hipMalloc(&v, by));
    CK(hipMemset(p, 0, by)); CK(hipMemset(g, 0, by)); CK(hipMemset(m, 0, by)); CK(hipMemset(v, 0, by));
    auto rep = [&](const char* what, float ms, double bytes_per) { printf("%-58s %8.1f us  %6.3f TB/s (%.3f of 8)\n", what, ms * 1e3, bytes_per * n / ms / 1e9, bytes_per * n / ms / 1e9 / 8.0); };
    printf("n = %ld parameters (%.0f MB per array)\n", n, by / 1e6);
    rep("copy a -> b (8 B/elem), 4096 blocks", timeit([&] { hipLaunchKernelGGL(copy_probe, 4096, 256, 0, 0, g, m, n4); }), 8);
    for (int gsz : {1024, 2048, 4096, 8192, 16384}) {
        char s[128];
        snprintf(s, sizeof s, "adam grid-stride NT, %d blocks", gsz); rep(s, timeit([&] { hipLaunchKernelGGL((adam_probe<0, true>), gsz, 256, 0, 0, p, g, m, v, n4, 0L); }), 28);
    }
    rep("adam grid-stride, plain loads/stores, 4096 blocks", timeit([&] { hipLaunchKernelGGL((adam_probe<0, false>), 4096, 256, 0, 0, p, g, m, v, n4, 0L); }), 28);
    for (int gsz : {2048, 4096, 8192}) {
        char s[128];
        snprintf(s, sizeof s, "adam XCD-contiguous eighths NT, %d blocks", gsz); rep(s, timeit([&] { hipLaunchKernelGGL((adam_probe<1, true>), gsz, 256, 0, 0, p, g, m, v, n4, 0L); }), 28);
    }
    for (long ch : {1024L, 4096L, 16384L, 65536L}) {
        char s[128];
        snprintf(s, sizeof s, "adam block-owned chunks of %ld KB NT, 4096 blocks", ch * 16 / 1024); rep(s, timeit([&] { hipLaunchKernelGGL((adam_probe<2, true>), 4096, 256, 0, 0, p, g, m, v, n4, ch); }), 28);
    }
    rep("adam 2 float4 in flight per array, 2048 blocks", timeit([&] { hipLaunchKernelGGL((adam_unrolled<2>), 2048, 256, 0, 0, p, g, m, v, n4); }), 28);
    rep("adam 4 float4 in flight per array, 1024 blocks", timeit([&] { hipLaunchKernelGGL((adam_unrolled<4>), 1024, 256, 0, 0, p, g, m, v, n4); }), 28);
    // two half-size launches back to back (does the footprint matter, or the launch?)
    rep("adam as two launches over halves, 4096 blocks each", timeit([&] { hipLaunchKernelGGL((adam_probe<0, true>), 4096, 256, 0, 0, p, g, m, v, n4 / 2, 0L);
        hipLaunchKernelGGL((adam_probe<0, true>), 4096, 256, 0, 0, p + n / 2, g + n / 2, m + n / 2, v + n / 2, n4 / 2, 0L); }), 28);
    return 0;
}
