// dev (round 6): can ONE wave run a whole 16-row tile of the residual-stack backward by itself -- every weight fragment read from LDS (ds_read_b128 per four
// MFMAs), no workgroup barrier, no other wave's help -- and keep the matrix cores busy?  The stack kernels split a tile over four waves with three barriers;
// their MFMA density is 0.28-0.41 at ~1.25 waves per SIMD (the dependency window of a batch-1 chunk admits no more).  This is the inner structure of the
// alternative: per tile 64 MFMAs (dg = dXout . Wr, K 64 x N 64), a gate-like VALU block, the dZ transpose through a wave-private LDS scratch, 256 MFMAs
// (dZ . W1, K 128 x N 128), an output transpose; WAVES waves per workgroup, one workgroup per CU.  Prints cycles per tile and the MFMA issue fraction.
//   hipcc --offload-arch=gfx950 -O3 tools/wave_tile_bench.hip -o tools/wave_tile_bench.bin && tools/wave_tile_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int WAVES, int MODE>      // MODE 0: MFMAs + LDS fragment reads only; 1: + transposes through the scratch; 2: + gate-like VALU; 3: + global row loads / stores
__global__ __launch_bounds__(64 * WAVES, 1) void k_tile(const float* __restrict__ gin, float* __restrict__ gout, long long* cyc, int ntiles) {
    extern __shared__ float sm[];
    float* W1 = sm;                         // [8 ks4][8 nt][64 lanes] float4  = 64 KB
    float* Wr = sm + 8 * 8 * 64 * 4;        // [4 ks4][4 nt][64] float4        = 16 KB
    float* scr = Wr + 4 * 4 * 64 * 4 + (threadIdx.x >> 6) * (16 * 132);     // wave-private scratch [16][132]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < (8 * 8 + 4 * 4) * 64 * 4; i += 64 * WAVES) sm[i] = 1e-3f * (float)((i * 2654435761u >> 20) & 1023) - 0.5f;
    __syncthreads();
    const float4* W1f = (const float4*)W1 + lane; const float4* Wrf = (const float4*)Wr + lane;
    float xa[16], za[32];
    for (int k = 0; k < 16; ++k) xa[k] = 0.01f * (lane + k);
    f32x4 keep = {0, 0, 0, 0};
    const size_t rowbase = ((size_t)blockIdx.x * WAVES + wave) * 16 * 128;
    long long t0 = clock64();
    for (int t = 0; t < ntiles; ++t) {
        if (MODE >= 3) {
            const float* src = gin + rowbase + (size_t)(t & 7) * 2048 * 128;
#pragma unroll
            for (int k = 0; k < 16; ++k) xa[k] += src[(lane & 15) * 128 + 4 * k + (lane >> 4)];
        }
        // ---- dg = dXout . Wr : 4 n-tiles, K = 64
        f32x4 dg[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) dg[n] = (f32x4){0, 0, 0, 0};
        float4 bw[2][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) bw[0][n] = Wrf[(0 * 4 + n) * 64];
#pragma unroll
        for (int ks4 = 0; ks4 < 4; ++ks4) {
            if (ks4 + 1 < 4) {
#pragma unroll
                for (int n = 0; n < 4; ++n) bw[(ks4 + 1) & 1][n] = Wrf[((ks4 + 1) * 4 + n) * 64];
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) dg[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * ks4 + 0], bw[ks4 & 1][n].x, dg[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) dg[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * ks4 + 1], bw[ks4 & 1][n].y, dg[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) dg[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * ks4 + 2], bw[ks4 & 1][n].z, dg[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) dg[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[4 * ks4 + 3], bw[ks4 & 1][n].w, dg[n], 0, 0, 0);
        }
        // ---- gate-like block: 16 values per lane -> 32 dz values
        float dz[32];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = dg[n][i];
                if (MODE >= 2) { const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-v)); const float th = 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * v)) - 1.0f; dz[8 * n + 2 * i] = v * th * s * (1.f - s); dz[8 * n + 2 * i + 1] = v * s * (1.f - th * th); }
                else { dz[8 * n + 2 * i] = v; dz[8 * n + 2 * i + 1] = -v; }
            }
        // ---- dZ (C layout: lane = column, 4 rows) -> A layout (lane = row, k) through the scratch
        if (MODE >= 1) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    scr[(4 * (lane >> 4) + i) * 132 + 16 * n + (lane & 15)] = dz[8 * n + 2 * i];
                    scr[(4 * (lane >> 4) + i) * 132 + 64 + 16 * n + (lane & 15)] = dz[8 * n + 2 * i + 1];
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
#pragma unroll
            for (int k = 0; k < 32; ++k) za[k] = scr[(lane & 15) * 132 + 4 * k + (lane >> 4)];
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) za[k] = dz[k];
        }
        // ---- d[x_cur | x_past] = dZ . W1 : 8 n-tiles, K = 128
        f32x4 acc[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[n] = (f32x4){0, 0, 0, 0};
        float4 b[2][8];
#pragma unroll
        for (int n = 0; n < 8; ++n) b[0][n] = W1f[(0 * 8 + n) * 64];
#pragma unroll
        for (int ks4 = 0; ks4 < 8; ++ks4) {
            if (ks4 + 1 < 8) {
#pragma unroll
                for (int n = 0; n < 8; ++n) b[(ks4 + 1) & 1][n] = W1f[((ks4 + 1) * 8 + n) * 64];
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4 * ks4 + 0], b[ks4 & 1][n].x, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4 * ks4 + 1], b[ks4 & 1][n].y, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4 * ks4 + 2], b[ks4 & 1][n].z, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[4 * ks4 + 3], b[ks4 & 1][n].w, acc[n], 0, 0, 0);
        }
        // ---- outputs: through the scratch as whole rows
        if (MODE >= 1) {
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) scr[(4 * (lane >> 4) + i) * 132 + 16 * n + (lane & 15)] = acc[n][i];
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float4 v = *(const float4*)(scr + (2 * r + (lane >> 5)) * 132 + 4 * (lane & 31));
                if (MODE >= 3) *(float4*)(gout + rowbase + (size_t)(t & 7) * 2048 * 128 + (2 * r + (lane >> 5)) * 128 + 4 * (lane & 31)) = v;
                else { keep[0] += v.x; keep[1] += v.y; keep[2] += v.z; keep[3] += v.w; }
            }
        } else {
#pragma unroll
            for (int n = 0; n < 8; ++n) keep += acc[n];
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) xa[k] = xa[k] * 0.5f + keep[k & 3] * 1e-6f;
    }
    long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * WAVES + wave] = t1 - t0;
    if (keep[0] == 123.456f) gout[0] = keep[1] + keep[2] + keep[3];
}

template <int WAVES, int MODE>
static void run(const float* gin, float* gout, long long* cyc, int ntiles) {
    const size_t lds = (size_t)((8 * 8 + 4 * 4) * 64 * 4 + WAVES * 16 * 132) * sizeof(float);
    CHECK(hipFuncSetAttribute((const void*)k_tile<WAVES, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_tile<WAVES, MODE>), dim3(256), dim3(64 * WAVES), lds, 0, gin, gout, cyc, 4);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_tile<WAVES, MODE>), dim3(256), dim3(64 * WAVES), lds, 0, gin, gout, cyc, ntiles);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long h[256 * 8]; CHECK(hipMemcpy(h, cyc, sizeof(long long) * 256 * WAVES, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 256 * WAVES; ++i) avg += (double)h[i]; avg /= 256.0 * WAVES * ntiles;
    const double tiles = 256.0 * WAVES * ntiles, flop = tiles * 320 * 2048;      // 320 MFMAs of 16x16x4 (1024 MACs) per tile
    printf("waves/CU %d (%.1f per SIMD) mode %d: %.0f clock64 ticks per tile per wave; kernel %.3f ms for %.0f tiles -> %.2f us per 1000 tiles on the chip, %.1f TFLOP/s (%.2f of 157.3)\n",
           WAVES, WAVES / 4.0, MODE, avg, ms, tiles, ms * 1e3 / tiles * 1000, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
}

int main() {
    float *gin, *gout; long long* cyc;
    const size_t n = (size_t)8 * 2048 * 128 * 16;
    CHECK(hipMalloc(&gin, n * sizeof(float))); CHECK(hipMalloc(&gout, n * sizeof(float))); CHECK(hipMalloc(&cyc, sizeof(long long) * 256 * 8));
    CHECK(hipMemset(gin, 0, n * sizeof(float)));
    const int nt = 200;
    run<4, 0>(gin, gout, cyc, nt); run<4, 1>(gin, gout, cyc, nt); run<4, 2>(gin, gout, cyc, nt); run<4, 3>(gin, gout, cyc, nt);
    run<8, 0>(gin, gout, cyc, nt); run<8, 1>(gin, gout, cyc, nt); run<8, 2>(gin, gout, cyc, nt); run<8, 3>(gin, gout, cyc, nt);
    return 0;
}
