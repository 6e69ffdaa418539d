// dev: does a narrower cache scope shorten a same-XCD hand-off?  The 4-stage ring of tools/hop_bench.hip with the granule store /
// poll load issued as inline asm with sc1 (agent scope, what the decode kernels use), sc0 (workgroup scope) or no scope bits.
// Bounded spins: a scope under which the data never becomes visible shows up as time-outs, not as a hang.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hop_scope.bin tools/hop_scope.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned long long u64;
template <int SC> __device__ __forceinline__ void gst(u64* g, u64 v) {
    if (SC == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(g), "v"(v) : "memory");
    else if (SC == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(g), "v"(v) : "memory");
    else if (SC == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(g), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(g), "v"(v) : "memory");
}
template <int SC> __device__ __forceinline__ u64 gld(const u64* g) {
    u64 v;
    if (SC == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
    else if (SC == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
    else if (SC == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(g) : "memory");
    return v;
}
template <int SC>
__global__ __launch_bounds__(256) void k_ring(u64* X, int iters, int* xcc, long long* cyc, int* fails) {
    __shared__ float sm[256];
    const int role = blockIdx.x / 8, grp = blockIdx.x % 8;            // roles of a group: blocks g, g+8, g+16, g+24 (one XCD under round-robin dispatch)
    if (grp != 0) return;
    u64* in = X + ((role + 3) & 3) * 64;
    u64* out = X + role * 64;
    const int tid = threadIdx.x;
    if (tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[role] = (int)(id & 0xf); }
    long long t0 = 0; int nf = 0;
    for (int it = 1; it <= iters; ++it) {
        if (!(role == 0 && it == 1) && tid < 64) {
            const unsigned tag = role == 0 ? (unsigned)(it - 1) : (unsigned)it;
            u64 v = gld<SC>(in + tid); unsigned spins = 0;
            while ((unsigned)(v >> 32) != tag && ++spins < (1u << 14)) v = gld<SC>(in + tid);
            if ((unsigned)(v >> 32) != tag) ++nf;
            sm[tid] = __uint_as_float((unsigned)v);
        }
        if (role == 0 && it == 2 && tid == 0) t0 = clock64();
        __syncthreads();
        if (tid < 64) gst<SC>(out + tid, ((u64)(unsigned)it << 32) | __float_as_uint(sm[tid] + 1.0f));
        __syncthreads();
    }
    if (role == 0 && tid == 0) cyc[0] = clock64() - t0;
    if (nf) atomicAdd(fails, nf);
}
template <int SC> static void run(const char* name, u64* X, int* xcc, long long* cyc, int* fails, int iters) {
    hipMemset(X, 0, 4 * 64 * 8); hipMemset(fails, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_ring<SC>, dim3(32), dim3(256), 0, 0, X, iters, xcc, cyc, fails);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int hx[4], hf; hipMemcpy(hx, xcc, 16, hipMemcpyDeviceToHost); hipMemcpy(&hf, fails, 4, hipMemcpyDeviceToHost);
    printf("%-22s %.3f us per hop, %d time-outs, XCC ids %d %d %d %d\n", name, ms * 1e3 / iters / 4, hf, hx[0], hx[1], hx[2], hx[3]);
}
int main() {
    u64* X; int* xcc; long long* cyc; int* fails;
    hipMalloc(&X, 4 * 64 * 8); hipMalloc(&xcc, 64); hipMalloc(&cyc, 64); hipMalloc(&fails, 4);
    run<1>("sc1 (agent scope)", X, xcc, cyc, fails, 20000);
    run<2>("sc0 (workgroup scope)", X, xcc, cyc, fails, 200);
    run<0>("no scope bits", X, xcc, cyc, fails, 200);
    run<3>("sc0 sc1 (system)", X, xcc, cyc, fails, 2000);
    return 0;
}
