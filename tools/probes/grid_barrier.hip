// Development probe: what does a grid-wide barrier cost on MI355X?  (Next-round question: would persistent kernels with grid barriers between the
// sub-blocks of the <= 16-token transformer stacks beat one launch per sub-block at ~3 us of step time per launch?)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier.hip -o tools/probes/grid_barrier && tools/probes/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void barrier_kernel(unsigned* counter, int nbar, int work, float* sink) {
    const unsigned nb = gridDim.x;
    float acc = threadIdx.x;
    for (int b = 0; b < nbar; b++) {
        for (int i = 0; i < work; i++) acc = acc * 1.0001f + 0.5f;      // stand-in for a phase of work
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned target = (unsigned)(b + 1) * nb;
            atomicAdd(counter, 1u);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (acc == 12345.f) sink[0] = acc;
}

__global__ void tiny_kernel(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0 && x[0] == 12345.f) x[1] = 1.f; }

int main() {
    unsigned* counter; float* sink;
    hipMalloc(&counter, 4); hipMalloc(&sink, 64);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {64, 128, 256, 512, 1024}) {
        for (int threads : {64, 256}) {
            for (int work : {0, 2000}) {
                float ms[2];
                int idx = 0;
                for (int nbar : {10, 210}) {
                    hipMemsetAsync(counter, 0, 4, st);
                    hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(threads), 0, st, counter, nbar, work, sink);   // warm-up
                    hipMemsetAsync(counter, 0, 4, st);
                    hipEventRecord(e0, st);
                    hipLaunchKernelGGL(barrier_kernel, dim3(blocks), dim3(threads), 0, st, counter, nbar, work, sink);
                    hipEventRecord(e1, st);
                    hipStreamSynchronize(st);
                    hipEventElapsedTime(&ms[idx++], e0, e1);
                }
                printf("blocks %4d x %3d threads, work %4d: %.2f us per barrier (+work)\n", blocks, threads, work, (ms[1] - ms[0]) * 1e3 / 200.0);
            }
        }
    }
    // the launch-side comparison: N dependent tiny kernels in a captured graph
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(tiny_kernel, dim3(144), dim3(256), 0, st, sink);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(e0, st); hipGraphLaunch(ge, st); hipEventRecord(e1, st); hipStreamSynchronize(st);
    float t; hipEventElapsedTime(&t, e0, e1);
    printf("200 dependent tiny kernels in a hipGraph: %.2f us per kernel\n", t * 1e3 / 200.0);
    return 0;
}
