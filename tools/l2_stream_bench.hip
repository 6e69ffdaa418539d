// Dev tool: how fast can ONE workgroup (one CU) re-stream an L2-resident weight block?
// This is the floor of the per-sample time of the one-CU-per-utterance decode kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/l2_stream_bench.hip -o /tmp/l2bench && /tmp/l2bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream(const float4* __restrict__ w, int n4_per_pass, int passes, float* out) {
    const int tid = threadIdx.x;
    float acc = 0.f;
    // each thread reads element tid + k*1024 (wave-contiguous 1 KiB per instruction)
    for (int p = 0; p < passes; ++p) {
        for (int i = tid; i < n4_per_pass; i += 1024 * DEPTH) {
            float4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) { int j = i + d * 1024; v[d] = j < n4_per_pass ? w[j] : make_float4(0, 0, 0, 0); }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc += v[d].x + v[d].y + v[d].z + v[d].w;
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main() {
    const size_t bytes = 1700 * 1024;          // paper-size QPNet streams ~1.7 MB per sample
    const int n4 = bytes / 16;
    float4* w; float* out;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(w, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int passes = 2000;
    for (int grid : {1, 20, 256}) {
        for (int depth : {4, 8, 16}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (depth == 4) hipLaunchKernelGGL(stream<4>, dim3(grid), dim3(1024), 0, 0, w, n4, passes, out);
                if (depth == 8) hipLaunchKernelGGL(stream<8>, dim3(grid), dim3(1024), 0, 0, w, n4, passes, out);
                if (depth == 16) hipLaunchKernelGGL(stream<16>, dim3(grid), dim3(1024), 0, 0, w, n4, passes, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep) printf("grid %3d depth %2d: %.2f us per 1.7MB pass -> %.1f GB/s per CU\n", grid, depth, ms * 1e3 / passes, bytes / (ms * 1e-3 / passes) / 1e9);
            }
        }
    }
    return 0;
}
