// dev microbenchmark: latency of a 4-stage ring of cross-CU hand-offs with data-tagged 8-byte granules
// (the transport of decode_coop.hip), to price a layer-pipelined decode:  S0 -(64)-> S1 -(64)-> K -(256)-> P -(1)-> S0
//   hipcc --offload-arch=gfx950 -O3 -o tools/hopbench tools/hop_bench.hip && tools/hopbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
__device__ __forceinline__ void gst(u64* g, unsigned tag, float v) { __hip_atomic_store(g, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 gld(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// role r waits for n_in[r] granules from its predecessor, does `work` dependent FMA chains + barriers, publishes n_out[r]
__global__ __launch_bounds__(1024) void k_ring(u64* X, int iters, int work, int stride8, int* xcc, long long* cyc) {
    __shared__ float sm[1024];
    const int role = stride8 ? (blockIdx.x / 8) : (blockIdx.x % 4);   // stride8: roles of a group are blocks g, g+8, g+16, g+24 (same XCD under round-robin dispatch)
    const int grp = stride8 ? (blockIdx.x % 8) : (blockIdx.x / 4);
    if (grp != 0) return;                                              // one active group; the others exit
    const int n_in[4] = {1, 64, 64, 256}, n_out[4] = {64, 64, 256, 1};
    u64* in = X + ((role + 3) & 3) * 512;       // written by the predecessor
    u64* out = X + role * 512;
    const int tid = threadIdx.x;
    if (tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[role] = (int)(id & 0xf); }
    long long t0 = 0;
    for (int it = 1; it <= iters; ++it) {
        if (!(role == 0 && it == 1)) {          // S0 starts the first round without waiting
            const unsigned tag = role == 0 ? (unsigned)(it - 1) : (unsigned)it;
            if (tid < n_in[role]) {
                u64 v = gld(in + tid); unsigned spins = 0;
                while ((unsigned)(v >> 32) != tag && ++spins < (1u << 22)) v = gld(in + tid);
                sm[tid] = __uint_as_float((unsigned)v);
            }
        }
        if (role == 0 && it == 2 && tid == 0) t0 = clock64();
        __syncthreads();
        float a = sm[tid & 255];
        for (int w = 0; w < work; ++w) {        // `work` dependent stages of a 16-FMA chain + barrier (a matvec stage)
#pragma unroll
            for (int k = 0; k < 16; ++k) a = __builtin_fmaf(a, 1.0001f, 0.5f);
            sm[tid] = a; __syncthreads(); a = sm[(tid + 1) & 1023];
        }
        if (tid < n_out[role]) gst(out + tid, (unsigned)it, a);
    }
    if (role == 0 && tid == 0) {
        // wait for the last round to come back
        u64 v = gld(in); unsigned spins = 0;
        while ((unsigned)(v >> 32) != (unsigned)iters && ++spins < (1u << 24)) v = gld(in);
        cyc[0] = clock64() - t0;
    }
}
int main() {
    u64* X; int* xcc; long long* cyc;
    hipMalloc(&X, 4 * 512 * 8); hipMalloc(&xcc, 64); hipMalloc(&cyc, 64);
    const int iters = 20000;
    for (int stride8 = 0; stride8 < 2; ++stride8)
        for (int work = 0; work <= 8; work += 4) {
            hipMemset(X, 0, 4 * 512 * 8);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_ring, dim3(32), dim3(1024), 0, 0, X, iters, work, stride8, xcc, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int hx[4]; hipMemcpy(hx, xcc, 16, hipMemcpyDeviceToHost);
            printf("%s work/role=%d: %.3f us per round of 4 hops (%.3f us/hop incl. work); XCC ids %d %d %d %d\n", stride8 ? "blocks g,g+8,g+16,g+24" : "blocks 0,1,2,3         ",
                   work, ms * 1e3 / iters, ms * 1e3 / iters / 4, hx[0], hx[1], hx[2], hx[3]);
        }
    return 0;
}
