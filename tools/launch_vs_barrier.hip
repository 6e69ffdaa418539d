// dev: what does a dependent kernel boundary cost against a grid-wide barrier inside one persistent launch?
// 1280 workgroups x 256 threads (5 per CU, 16 KB of LDS each: the layer kernels' shape).
//   hipcc --offload-arch=gfx950 -O3 tools/launch_vs_barrier.hip -o tools/launch_vs_barrier.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void k_empty(float* out) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (sm[(threadIdx.x + 1) & 255] == -1.f) out[blockIdx.x] = 1.f;
}
// sense-reversing counter barrier: arrivals at agent scope, waiters poll with agent-scope loads; bounded spin
__global__ __launch_bounds__(256) void k_barrier(unsigned* ctr, int iters, float* out) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = (float)threadIdx.x;
    const unsigned nwg = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = nwg * (unsigned)(it + 1);
            unsigned spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (sm[(threadIdx.x + 1) & 255] == -1.f) out[blockIdx.x] = 1.f;
}
int main() {
    const int nwg = 1280, iters = 2000;
    float* out; unsigned* ctr;
    hipMalloc(&out, nwg * sizeof(float)); hipMalloc(&ctr, 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_empty, dim3(nwg), dim3(256), 16384, 0, out);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_empty, dim3(nwg), dim3(256), 16384, 0, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d dependent launches of an empty 1280 x 256 kernel: %.2f us each\n", iters, ms * 1e3 / iters);
    hipMemset(ctr, 0, 256);
    hipLaunchKernelGGL(k_barrier, dim3(nwg), dim3(256), 16384, 0, ctr, 10, out);
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 256);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_barrier, dim3(nwg), dim3(256), 16384, 0, ctr, iters, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    unsigned h = 0; hipMemcpy(&h, ctr, 4, hipMemcpyDeviceToHost);
    printf("%d grid barriers among 1280 resident workgroups: %.2f us each (counter %u, expected %u)\n", iters, ms * 1e3 / iters, h, (unsigned)nwg * iters);
    return 0;
}
