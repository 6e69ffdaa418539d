// dev: the fp32 matrix-core rate this card sustains (the roofline's 157 TFLOP/s assumes 2.4 GHz on all 256 CUs).
// Every wave issues independent v_mfma_f32_32x32x2_f32 back to back; wall time from HIP events, shader clock from s_memtime
// against the 100 MHz s_memrealtime.   hipcc --offload-arch=gfx950 -O3 tools/mfma_clock.hip -o /tmp/mfma_clock && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_mfma(float* out, unsigned long long* clk, int iters) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = (float)threadIdx.x * 1e-9f, y = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int wpc = 1; wpc <= 2; ++wpc) {                 // workgroups (4 waves each) per CU
        const int nwg = 256 * wpc;
        float* out; unsigned long long* clk;
        hipMalloc(&out, nwg * 256 * sizeof(float)); hipMalloc(&clk, nwg * 2 * sizeof(unsigned long long));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mfma, dim3(nwg), dim3(256), 0, 0, out, clk, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma, dim3(nwg), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[4]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
        const double flop = (double)nwg * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
        printf("%d workgroup(s)/CU: %.3f ms, %.1f TFLOP/s; s_memtime ticks %llu vs 100 MHz ticks %llu -> counter %.3f GHz; "
               "MFMA cycles needed at 64/instr: %.0f -> implied shader clock >= %.3f GHz\n",
               wpc, ms, flop / (ms * 1e-3) / 1e12, h[0], h[1], (double)h[0] / ((double)h[1] / 100e6) / 1e9,
               (double)iters * 4 * 64 * wpc, (double)iters * 4 * 64 * wpc / (ms * 1e-3) / 1e9);
        hipFree(out); hipFree(clk);
    }
    return 0;
}
