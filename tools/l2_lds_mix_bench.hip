// Dev tool: does LDS->VGPR read traffic slow down L2->VGPR streaming on one CU?  (decode kernel: every 4 KiB weight
// tile comes with 4 ds_read_b128 of the input vector, i.e. as many LDS bytes into VGPRs as weight bytes)
//   hipcc -O3 --offload-arch=gfx950 tools/l2_lds_mix_bench.hip -o /tmp/l2mix && /tmp/l2mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int LDSR, int FMAS>
__global__ __launch_bounds__(1024) void stream(const float4* __restrict__ w, int n4_per_pass, int passes, float* out) {
    __shared__ float4 xs[1024];
    const int tid = threadIdx.x, lane = tid & 63;
    xs[tid] = make_float4(tid, 1, 2, 3);
    __syncthreads();
    float acc = 0.f;
    for (int p = 0; p < passes; ++p) {
        for (int i = tid; i < n4_per_pass; i += 1024 * 4) {
            float4 v[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) { int j = i + d * 1024; v[d] = j < n4_per_pass ? w[j] : make_float4(0, 0, 0, 0); }
            float4 x[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) x[d] = make_float4(1, 1, 1, 1);
#pragma unroll
            for (int d = 0; d < LDSR; ++d) { x[d] = xs[((lane & 3) * 4 + d + (p & 1) * 16) & 1023]; }
            if (FMAS) {
#pragma unroll
                for (int d = 0; d < 4; ++d) { acc = fmaf(v[d].x, x[d].x, acc); acc = fmaf(v[d].y, x[d].y, acc); acc = fmaf(v[d].z, x[d].z, acc); acc = fmaf(v[d].w, x[d].w, acc); }
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) acc += (v[d].x + x[d].x) + (v[d].y + x[d].y);
            }
        }
        asm volatile("" ::: "memory");
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

template <int LDSR, int FMAS>
void run(const float4* w, int n4, float* out, size_t bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int passes = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((stream<LDSR, FMAS>), dim3(1), dim3(1024), 0, 0, w, n4, passes, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("ds_read_b128 per 4 loads: %d  fma16: %d : %.2f us per 1.7MB pass -> %.1f GB/s\n", LDSR, FMAS, ms * 1e3 / passes, bytes / (ms * 1e-3 / passes) / 1e9);
    }
}

int main() {
    const size_t bytes = 1700 * 1024;
    const int n4 = bytes / 16;
    float4* w; float* out;
    CK(hipMalloc(&w, bytes)); CK(hipMalloc(&out, 4096));
    CK(hipMemset(w, 0, bytes));
    run<0, 0>(w, n4, out, bytes); run<1, 0>(w, n4, out, bytes); run<2, 0>(w, n4, out, bytes); run<4, 0>(w, n4, out, bytes);
    run<0, 1>(w, n4, out, bytes); run<4, 1>(w, n4, out, bytes);
    return 0;
}
