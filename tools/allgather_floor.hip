// dev microbenchmark: the transport floor of the cooperative decode kernel (csrc/decode_coop.hip) for the repo-default geometry
// (reference src/utils/param_model.py:58-64: n_resch 512, 12 fixed + 4 adaptive layers): per generated sample 2 * 16 + 3 = 35 dependent
// stages, each ending in an all-gather of a C- or S-float vector among the G workgroups of the utterance, as 8-byte {tag, value}
// granules gathered by the kernel's own sweep (gather_vec: four granules per lane per sweep, the first ceil(n / 256) waves).
// NO arithmetic and NO weight stream: what remains is the serial chain of publish -> visible -> gathered -> barrier edges.
//   hipcc --offload-arch=gfx950 -O3 -o tools/allgather_floor.bin tools/allgather_floor.hip && tools/allgather_floor.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;
#define NT 768
__device__ __forceinline__ void gr_store(u64* g, unsigned tag, float v) { __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 gr_load(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool gather_vec(const u64* src, int n, unsigned tag, float* dst, int tid) {
    const int base = (tid >> 6) * 256 + (tid & 63);
    if ((tid >> 6) * 256 >= n) return true;
    u64 v[4]; unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int i = base + 64 * k; v[k] = i < n ? gr_load(src + i) : ((u64)tag << 32); ok &= (unsigned)(v[k] >> 32) == tag; }
        if (__all(ok)) break;
        if (++spins > (1u << 22)) return false;
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + 64 * k; if (i < n) dst[i] = __uint_as_float((unsigned)v[k]); }
    return true;
}

// stage s of a sample: vector length n_s (512 for the 2 x 16 layer stages, 256 for the three post-net stages); buffer s is reused once per sample
__global__ __launch_bounds__(NT) void k_chain(u64* X, int G, int samples, int L, int C, int S, int* fail) {
    __shared__ float sm[1024];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nst = 2 * L + 3;
    float acc = (float)g;
    for (int t = 1; t <= samples; ++t) {
        for (int s = 0; s < nst; ++s) {
            const int n = s < 2 * L ? C : S, slice = n / G;
            u64* buf = X + (size_t)s * 512;
            if (tid < slice) gr_store(buf + g * slice + tid, (unsigned)t, acc + (float)tid);      // this workgroup's rows of the stage's output
            if (!gather_vec(buf, n, (unsigned)t, sm, tid)) { if (tid == 0) *fail = 1; return; }
            __syncthreads();
            acc = sm[(g * 7 + s) % n] * 0.5f + 1.0f;                                              // depend on the gathered vector
            __syncthreads();
        }
    }
    if (tid == 0 && acc == 12345.678f) *fail = 2;
}

int main() {
    u64* X; int* fail;
    (void)hipMalloc(&X, 64 * 512 * sizeof(u64)); (void)hipMalloc(&fail, 4);
    for (int G : {64, 32, 8}) {
        (void)hipMemset(X, 0, 64 * 512 * sizeof(u64)); (void)hipMemset(fail, 0, 4);
        const int samples = 4000;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_chain, dim3(G), dim3(NT), 0, 0, X, G, 200, 16, 512, 256, fail);         // warm-up (tags 1..200)
        (void)hipDeviceSynchronize(); (void)hipMemset(X, 0, 64 * 512 * sizeof(u64));
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain, dim3(G), dim3(NT), 0, 0, X, G, samples, 16, 512, 256, fail);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
        int f = 0; (void)hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
        printf("G = %2d workgroups: %7.2f us per sample = %5.2f us per all-gather edge (35 edges: 32 of 512 floats, 3 of 256)%s\n", G, ms * 1e3 / samples, ms * 1e3 / samples / 35.0, f ? "  [WAIT TIMED OUT]" : "");
    }
    return 0;
}
