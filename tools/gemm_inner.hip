// dev: which part of k_gemm_nn's chunk loop keeps the matrix cores from their peak?  The same 128x128xGK tile loop with the
// pieces switched on one by one: (0) LDS-fed MFMAs only, (1) + one barrier per chunk, (2) + the register->LDS staging stores,
// (3) + the global loads of the next chunk.   hipcc --offload-arch=gfx950 -O3 tools/gemm_inner.hip -o tools/gemm_inner.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define GM 128
#define GN 128
#ifndef GK
#define GK 16
#endif
#define LDA (GK + 1)
template <int MODE>
__global__ __launch_bounds__(256) void k_inner(const float* __restrict__ Ag, const float* __restrict__ Bg, float* out, int nk, int lda, int ldb) {
    extern __shared__ float sm[];
    float* As = sm; float* Bs = sm + 2 * GM * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int TA = GK / 4, RA = 256 / TA, PA = GM / RA, PB = GK / 8;
    const int ar = tid / TA, ak4 = (tid % TA) * 4, bk = tid >> 5, bn4 = (tid & 31) * 4;
    if (nk > 1000) for (int i = tid; i < 2 * GM * LDA + 2 * GK * GN; i += 256) sm[i] = 1e-3f * (float)((i * 2654435761u >> 20) & 1023) - 0.5f;
    float4 ra[PA], rb[PB];
    for (int p = 0; p < PA; ++p) ra[p] = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int p = 0; p < PB; ++p) rb[p] = make_float4(1.f, 2.f, 3.f, 4.f);
    const float* ap = Ag + (size_t)(blockIdx.x % 160) * GM * lda + (size_t)ar * lda + ak4;
    const float* bp = Bg + (size_t)bk * ldb + (blockIdx.x % 8) * GN + bn4;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        if (MODE >= 3) {
#pragma unroll
            for (int p = 0; p < PA; ++p) ra[p] = *(const float4*)(ap + (size_t)p * RA * lda + (size_t)(kc % (1024 / GK)) * GK);
#pragma unroll
            for (int p = 0; p < PB; ++p) rb[p] = *(const float4*)(bp + (size_t)((kc % (1024 / GK)) * GK + 8 * p) * ldb);
        }
        const float* A = As + (kc & 1) * (GM * LDA) + (32 * wave + (lane & 31)) * LDA + (lane >> 5);
        const float* Bq = Bs + (kc & 1) * (GK * GN) + (lane >> 5) * GN + (lane & 31);
        float an = A[0], bn0 = Bq[0], bn1 = Bq[32], bn2 = Bq[64], bn3 = Bq[96];
#pragma unroll
        for (int ks = 0; ks < GK / 2; ++ks) {
            const float a = an, b0 = bn0, b1 = bn1, b2 = bn2, b3 = bn3;
            if (ks + 1 < GK / 2) {
                an = A[2 * (ks + 1)];
                bn0 = Bq[2 * (ks + 1) * GN]; bn1 = Bq[2 * (ks + 1) * GN + 32]; bn2 = Bq[2 * (ks + 1) * GN + 64]; bn3 = Bq[2 * (ks + 1) * GN + 96];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= 2) {
            float* ad = As + ((kc + 1) & 1) * (GM * LDA) + ar * LDA + ak4;
            float* bd = Bs + ((kc + 1) & 1) * (GK * GN) + bk * GN + bn4;
#pragma unroll
            for (int p = 0; p < PA; ++p) { float* d = ad + RA * p * LDA; d[0] = ra[p].x; d[1] = ra[p].y; d[2] = ra[p].z; d[3] = ra[p].w; }
#pragma unroll
            for (int p = 0; p < PB; ++p) *(float4*)(bd + 8 * p * GN) = rb[p];
        }
        if (MODE >= 1) __syncthreads();
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int MODE>
static void run(int wpc, int nk, int nwg_override, const float* A, const float* B, float* out, int lda, int ldb) {
    const size_t lds = (size_t)(2 * GM * LDA + 2 * GK * GN) * sizeof(float);
    hipFuncSetAttribute((const void*)k_inner<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int nwg = nwg_override ? nwg_override : 256 * wpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_inner<MODE>, dim3(nwg), dim3(256), lds, 0, A, B, out, 64, lda, ldb);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_inner<MODE>, dim3(nwg), dim3(256), lds, 0, A, B, out, nk, lda, ldb);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)nwg * nk * 2.0 * GM * GN * GK;
    if (nwg_override) printf("%d workgroups x %d chunks: ", nwg, nk);
    printf("mode %d, %d workgroup(s)/CU, GK %d: %.3f ms, %.1f TFLOP/s\n", MODE, wpc, GK, ms, flop / (ms * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const int nk = argc > 1 ? atoi(argv[1]) : 20000;
    const int lda = 1088, ldb = 1024;
    float *A, *B, *out;
    hipMalloc(&A, (size_t)160 * GM * lda * sizeof(float)); hipMalloc(&B, (size_t)lda * ldb * sizeof(float)); hipMalloc(&out, (size_t)256 * 4 * 256 * sizeof(float));
    {   // realistic operand values (zeros would understate the matrix cores' power draw)
        const size_t na = (size_t)160 * GM * lda, nb = (size_t)lda * ldb;
        float* h = (float*)malloc((na > nb ? na : nb) * sizeof(float));
        unsigned st = 12345u;
        for (size_t i = 0; i < na; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((int)(st >> 8) % 2001 - 1000) * 1e-3f; }
        hipMemcpy(A, h, na * sizeof(float), hipMemcpyHostToDevice);
        for (size_t i = 0; i < nb; ++i) { st = st * 1664525u + 1013904223u; h[i] = ((int)(st >> 8) % 2001 - 1000) * 1e-4f; }
        hipMemcpy(B, h, nb * sizeof(float), hipMemcpyHostToDevice);
        free(h);
    }
    if (argc > 2) {                      // tile-sized work items: `nwg` workgroups of nk chunks each (1280 x 68 = one C=512 gate GEMM)
        const int nwg = atoi(argv[2]);
        hipFree(out); hipMalloc(&out, (size_t)nwg * 256 * sizeof(float));
        run<3>(3, nk, nwg, A, B, out, lda, ldb);
        return 0;
    }
    for (int wpc = 1; wpc <= 3; ++wpc) {
        run<0>(wpc, nk, 0, A, B, out, lda, ldb); run<1>(wpc, nk, 0, A, B, out, lda, ldb);
        run<2>(wpc, nk, 0, A, B, out, lda, ldb); run<3>(wpc, nk, 0, A, B, out, lda, ldb);
    }
    return 0;
}
