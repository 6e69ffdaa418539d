// train_bwd.hip -- hand-written backward of QPNet.forward on gfx950 (what torch autograd does for the
// reference at src/bin/qpnet_train.py:529-530), fp32 MFMA.
//
//   k_post_bwd   : dlogits -> dY0 -> dS0 (skip-sum grad) -> per-layer gate grads DGS = dS0 . Ws_l
//   k_layer_bwd  : dXout, DGS -> dg -> (dzs, dzt) -> d[x_cur | x_past | aux] = dZ . W1
//                  x_cur part (+ residual) is stored at its own row; the x_past part is the backward of
//                  the pitch-dependent gather = scatter-add to row tap[n] (float atomics only for the
//                  adaptive blocks; fixed blocks have a unique writer per row)
//   k_wgrad2     : every weight gradient is a time-contraction dW[m][n] = sum_t A[t][m] B[t][n]; one launch per
//                  weight family over (time chunk, layer), partial slabs (deterministic), bias grads = colsum(A)
//   k_causal_bwd : the one-hot causal conv's weight grad is a histogram over sample classes -> LDS table
//   k_up_bwd     : gradient of the (1,1,1,U) upsampling kernel and its bias
//   k_reduce_grad: slabs -> flat gradient in state_dict order;  k_adam: torch.optim.Adam update.
#include "train_common.h"
#include "train_post.h"
#include "qpn_handle.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>

// ------------------------------------------------------------------------------------------ post-net backward
// dynamic LDS: P[TM][lda(max(Q,S))] | R[TM][lda(S)]
template <int MT>
__global__ __launch_bounds__(512) void k_post_bwd(TrainParams p, TrainBwd bw) {
    constexpr int TM = 16 * MT;
    extern __shared__ float sm[];
    const int S = p.S, Q = p.Q, LC = p.LC;
    const int ldp = tr_lda(Q > S ? Q : S), ldr = tr_lda(S);
    float* P = sm; float* R = sm + TM * ldp;
    const int b = blockIdx.y, t0 = blockIdx.x * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTS = S / 16;
    // The ReLU masks of the two epilogues (saved pre-activations Y0, S0 at this wave's output positions) are requested AHEAD of the
    // contraction they follow -- Y0 before the staging, S0 before the first contraction -- and pinned with an empty asm in front of
    // it: as plain epilogue loads each of the 8 sat behind its own s_waitcnt (a per-row `if` around load and store), 16 exposed L2
    // round trips per workgroup.  Rows past the chunk end read the arena's padding rows and are masked at the store.
    const int npairs = (NTS + 1) / 2;
    const int np0 = wave < npairs ? wave : 0;
    const int pnt[2] = {2 * np0, (2 * np0 + 1 < NTS) ? 2 * np0 + 1 : 2 * np0};
    float my[2][MT][4], ms[2][MT][4];
    auto mask_load = [&](const float* src, float (&m)[2][MT][4]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = t0 + 16 * mt + 4 * (lane >> 4) + i;      // rows past the chunk end: readable padding (see the arena), masked below
                    m[j][mt][i] = src[((size_t)b * p.BL + r) * S + 16 * pnt[j] + (lane & 15)];
                }
    };
#define PIN8(m) asm volatile("" :: "v"(m[0][0][0]), "v"(m[0][0][1]), "v"(m[0][0][2]), "v"(m[0][0][3]), "v"(m[1][0][0]), "v"(m[1][0][1]), "v"(m[1][0][2]), "v"(m[1][0][3]))
    mask_load(p.Y0, my);
    // stage dlogits rows
    for (int idx = tid; idx < TM * (Q / 2); idx += 512) {
        const int r = idx / (Q / 2), k = (idx - r * (Q / 2)) * 2;
        float2 v = make_float2(0.f, 0.f);
        if (t0 + r < p.BL) v = *(const float2*)(bw.dlogits + ((size_t)b * p.BL + t0 + r) * Q + k);
        *(float2*)(P + (size_t)r * ldp + k) = v;
    }
    __syncthreads();
    mask_load(p.S0, ms);
    PIN8(my);
    if (MT > 1) {
#pragma unroll
        for (int mt = 1; mt < MT; ++mt) asm volatile("" :: "v"(my[0][mt][0]), "v"(my[0][mt][1]), "v"(my[0][mt][2]), "v"(my[0][mt][3]), "v"(my[1][mt][0]), "v"(my[1][mt][1]), "v"(my[1][mt][2]), "v"(my[1][mt][3]));
    }
    // dY0 = (dlogits . W2) * (Y0 > 0)
    for (int pb = 0; pb < npairs; pb += 8) {
        const int np = pb + wave;
        if (np < npairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
            f32x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<MT, 2>(acc, P, ldp, p.wp + p.p2t_f4, NTS, nts, Q, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j && nt1 == nt0) break;
                const int c = 16 * (j ? nt1 : nt0) + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        const bool in = t0 + r < p.BL;
                        const size_t o = ((size_t)b * p.BL + t0 + r) * S + c;
                        const float pre = pb == 0 ? my[j][mt][i] : (in ? p.Y0[o] : 0.f);
                        const float v = (in && pre > 0.f) ? acc[mt][j][i] : 0.f;
                        if (in) bw.DY0[o] = v;
                        R[(size_t)r * ldr + c] = v;
                    }
            }
        }
    }
    __syncthreads();
    PIN8(ms);
    if (MT > 1) {
#pragma unroll
        for (int mt = 1; mt < MT; ++mt) asm volatile("" :: "v"(ms[0][mt][0]), "v"(ms[0][mt][1]), "v"(ms[0][mt][2]), "v"(ms[0][mt][3]), "v"(ms[1][mt][0]), "v"(ms[1][mt][1]), "v"(ms[1][mt][2]), "v"(ms[1][mt][3]));
    }
#undef PIN8
    // dS0 = (dY0 . W1post) * (S0 > 0)   -> P (dlogits are dead)
    for (int pb = 0; pb < npairs; pb += 8) {
        const int np = pb + wave;
        if (np < npairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
            f32x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<MT, 2>(acc, R, ldr, p.wp + p.p1t_f4, NTS, nts, S, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j && nt1 == nt0) break;
                const int c = 16 * (j ? nt1 : nt0) + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        const bool in = t0 + r < p.BL;
                        const size_t o = ((size_t)b * p.BL + t0 + r) * S + c;
                        const float pre = pb == 0 ? ms[j][mt][i] : (in ? p.S0[o] : 0.f);
                        const float v = (in && pre > 0.f) ? acc[mt][j][i] : 0.f;
                        if (in) bw.DS0[o] = v;
                        P[(size_t)r * ldp + c] = v;
                    }
            }
        }
    }
    __syncthreads();
    // DGS[t][l*C + c] = sum_s dS0[t][s] Ws_l[s][c]
    const int NTL = LC / 16, lpairs = (NTL + 1) / 2;
    for (int pb = 0; pb < lpairs; pb += 8) {
        const int np = pb + wave;
        if (np < lpairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTL) ? 2 * np + 1 : 2 * np;
            f32x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<MT, 2>(acc, P, ldp, p.wp + p.wst_f4, NTL, nts, S, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j && nt1 == nt0) break;
                const int c = 16 * (j ? nt1 : nt0) + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        if (t0 + r < p.BL) bw.DGS[((size_t)b * p.BL + t0 + r) * LC + c] = acc[mt][j][i];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ post-net backward, wide tiles (S = Q = 256, n_resch 64)
// Counterpart of k_post_fwd_w (train_fwd.hip): 16 * MT rows per workgroup (MT = 5: a 20 000-row chunk is one round of 250
// workgroups), the transposed post-net weights stream from L2 once per 80 rows instead of once per 16, one [16 MT][256] LDS
// tile reused in place by the stages, every output array written as whole rows.  The ReLU masks are the signs of the rectified
// activations the forward stored (p.Y0, p.S0), requested ahead of the contraction they follow.
__device__ __forceinline__ void zero_dx_slice(const TrainParams& p, const TrainBwd& bw, int j, int b, size_t part, size_t nparts, int t, int nthr);
// zero_dx: this launch also does k_zero_dx's zeroing (fire-and-forget stores ahead of the first contraction; nothing here touches those
// arrays, the layer backward that does starts after this kernel) -- one launch and two stream events fewer per step
template <int MT>
__global__ __launch_bounds__(512) void k_post_bwd_w(TrainParams p, TrainBwd bw, int zero_dx) {
    if (zero_dx) for (int j = 0; j < p.L; ++j) zero_dx_slice(p, bw, j, blockIdx.y, blockIdx.x, gridDim.x, threadIdx.x, 512);
    extern __shared__ float sm[];
    float4 bq[2];
    post_bwd_w_tile<MT, false>(p, bw, sm, 0ull, 0ull, bq);
}
// Forward (skip sum, post-net, cross entropy) AND backward of a row tile in one kernel, for callers that run both anyway (qpn_train_step): dL/dlogits goes from the cross entropy
// to the first backward contraction through the LDS tile it is already in, the two ReLU masks are 40 sign bits a lane instead of 160 scattered loads, one launch, one prologue
// and the gap between two kernels less (the separate backward alone: 116 -> 104 us without its staging and mask loads).
template <int MT>
__global__ __launch_bounds__(512) void k_post_fb_w(TrainParams p, TrainBwd bw, int zero_dx) {
    if (zero_dx) for (int j = 0; j < p.L; ++j) zero_dx_slice(p, bw, j, blockIdx.y, blockIdx.x, gridDim.x, threadIdx.x, 512);
    extern __shared__ float sm[];
    unsigned long long mS = 0ull, mY = 0ull; float4 bq[2];
    post_fwd_w_tile<MT, true>(p, sm, mS, mY, bq);
    post_bwd_w_tile<MT, true>(p, bw, sm, mS, mY, bq);
}

// ------------------------------------------------------------------------------------------ layer backward
// dynamic LDS: Dx[TM][lda(C)] | Dz[TM][lda(2C)] | Sg, Th, Dg [TM][lda(C)] each
// DXin = grads w.r.t. this layer's OUTPUT (A: own-row part, B: scattered part, both zero-filled where unwritten);
// DXout = grads w.r.t. this layer's INPUT (same two-part form), consumed by layer l-1 / the causal backward.
template <int MT>
__global__ __launch_bounds__(256) void k_layer_bwd(TrainParams p, TrainBwd bw, int l, int last, int pp) {
    constexpr int TM = 16 * MT;
    extern __shared__ float sm[];
    const TrLayer ly = p.layers[l];
    const int C = p.C, Ktp = p.Ktp, Ap = p.Ap;
    const int ldx = tr_lda(C), ldz = tr_lda(2 * C);
    float* Dx = sm; float* Dz = sm + TM * ldx;
    const int b = blockIdx.y, n0 = ly.s_out + tr_xcd_tile(blockIdx.x, gridDim.x, pp & 1) * TM;      // pp: bit 0 XCD swizzle
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t rb = (size_t)b * p.N1;
    const size_t nDX = (size_t)p.B * p.N1 * C;     // grads wrt X[j] live at DXA/DXB[0] + j*nDX
    const float* DAin = bw.DXA[0] + (size_t)(l + 1) * nDX + rb * C; const float* DBin = bw.DXB[0] + (size_t)(l + 1) * nDX + rb * C;
    float* DAout = bw.DXA[0] + (size_t)l * nDX + rb * C; float* DBout = bw.DXB[0] + (size_t)l * nDX + rb * C;
    const int NCG = C / 16;
    // ---- stage dXout (= own-row part + scattered part), the saved gate halves and the skip-path gate grads: 16-byte loads
    float* Sg = Dz + TM * ldz; float* Th = Sg + TM * ldx; float* Dg = Th + TM * ldx;
    const float* SG = p.SG + ((size_t)(l * p.B + b) * p.N1) * C;
    const float* TH = p.TH + ((size_t)(l * p.B + b) * p.N1) * C;
    const int win0 = p.N1 - p.BL;
    // the residual-1x1 fragments of this wave's first n-tile do not depend on the tile: requested before the staging loads, consumed
    // behind the barrier (requested there, their L2 round trip was exposed in front of the first contraction)
    float4 bqr[4][1];
    if (!last) { const int ntr[1] = {wave < NCG ? wave : 0}; wave_b_preload<1, 4>(bqr, p.wp + ly.wrt_f4, NCG, ntr, C, lane); }
    {
        const int C4 = C / 4, tpr = C4 < 64 ? C4 : 64, rpp = 256 / tpr;
        const int tr = tid / tpr, tc = tid - tr * tpr;
        if (tr < rpp)
            for (int r = tr; r < TM; r += rpp) {
                const int n = n0 + r;
                for (int c4 = tc; c4 < C4; c4 += tpr) {
                    const int c = c4 * 4;
                    float4 dx = make_float4(0.f, 0.f, 0.f, 0.f), sg = dx, th = dx, dgs = dx;
                    if (n < p.N1) {
                        if (!last) {
                            const float4 a = *(const float4*)(DAin + (size_t)n * C + c), bb = *(const float4*)(DBin + (size_t)n * C + c);
                            dx = make_float4(a.x + bb.x, a.y + bb.y, a.z + bb.z, a.w + bb.w);
                        }
                        sg = *(const float4*)(SG + (size_t)n * C + c); th = *(const float4*)(TH + (size_t)n * C + c);
                        if (n >= win0) dgs = *(const float4*)(bw.DGS + ((size_t)b * p.BL + (n - win0)) * p.LC + (size_t)l * C + c);
                    }
                    float* d0 = Dx + (size_t)r * ldx + c; *(float2*)d0 = make_float2(dx.x, dx.y); *(float2*)(d0 + 2) = make_float2(dx.z, dx.w);
                    float* d1 = Sg + (size_t)r * ldx + c; *(float2*)d1 = make_float2(sg.x, sg.y); *(float2*)(d1 + 2) = make_float2(sg.z, sg.w);
                    float* d2 = Th + (size_t)r * ldx + c; *(float2*)d2 = make_float2(th.x, th.y); *(float2*)(d2 + 2) = make_float2(th.z, th.w);
                    float* d3 = Dg + (size_t)r * ldx + c; *(float2*)d3 = make_float2(dgs.x, dgs.y); *(float2*)(d3 + 2) = make_float2(dgs.z, dgs.w);
                }
            }
    }
    __syncthreads();
    // ---- dg = dXout . Wr + DGS ; dz = dg * gate'
    float* DZg = bw.DZ + ((size_t)l * p.B * p.N1 + rb) * 2 * C;
    for (int nt = wave; nt < NCG; nt += 4) {
        f32x4 acc[MT][1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = (f32x4){0, 0, 0, 0};
        if (!last) {
            const int nts[1] = {nt};
            if (nt == wave) wave_gemm_run<MT, 1, 4>(acc, Dx, ldx, bqr, p.wp + ly.wrt_f4, NCG, nts, C, lane);
            else wave_gemm_deep<MT, 1, 4>(acc, Dx, ldx, p.wp + ly.wrt_f4, NCG, nts, C, lane);
        }
        const int c = 16 * nt + (lane & 15);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i, n = n0 + r;
                const float dg = acc[mt][0][i] + Dg[(size_t)r * ldx + c];
                const float sg = Sg[(size_t)r * ldx + c], th = Th[(size_t)r * ldx + c];
                float dzs, dzt; tr_gate_bwd(dg, sg, th, dzs, dzt);      // (th: the gate product the Th tile holds)
                if (n < p.N1) { DZg[(size_t)n * 2 * C + c] = dzs; DZg[(size_t)n * 2 * C + C + c] = dzt; }
                Dz[(size_t)r * ldz + c] = dzs; Dz[(size_t)r * ldz + C + c] = dzt;
            }
    }
    // ---- d[x_cur | x_past | aux] = dZ . W1.  A wave's n-tiles run back to back; the weight fragments of the NEXT n-tile and the
    //      scatter rows are requested before the barrier / under the previous tile's epilogue (K = 2C <= 128: all 8 steps fit the ring)
    const int NTK = Ktp / 16;
    const int* taps = p.TAP + ly.tap_off + (size_t)b * p.N1;      // a table for every layer (k_train_prep)
    float* DH = bw.DHUP + rb * Ap;
    constexpr int PD3 = 8;
    const float4* W1t = p.wp + ly.w1t_f4;
    float4 bq[PD3][1];
    int tprow[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + 16 * mt + 4 * (lane >> 4) + i;
            tprow[mt][i] = taps[n < p.N1 ? n : p.N1 - 1];
        }
    if (wave < NTK) { const int nts0[1] = {wave}; wave_b_preload<1, PD3>(bq, W1t, NTK, nts0, 2 * C, lane); }
    __syncthreads();
    for (int nt = wave; nt < NTK; nt += 4) {
        f32x4 acc[MT][1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = (f32x4){0, 0, 0, 0};
        const int nts[1] = {nt};
        wave_gemm_run<MT, 1, PD3>(acc, Dz, ldz, bq, W1t, NTK, nts, 2 * C, lane);
        if (2 * C <= 16 * PD3 && nt + 4 < NTK) { const int ntn[1] = {nt + 4}; wave_b_preload<1, PD3>(bq, W1t, NTK, ntn, 2 * C, lane); }
        const int k = 16 * nt + (lane & 15);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i, n = n0 + r;
                if (n >= p.N1) continue;
                const float v = acc[mt][0][i];
                if (k < C) DAout[(size_t)n * C + k] = v + Dx[(size_t)r * ldx + k];              // + residual path
                else if (k < 2 * C) {
                    if (ly.adaptive) atomicAdd(&DBout[(size_t)tprow[mt][i] * C + (k - C)], v);          // gather backward (collisions)
                    else DBout[(size_t)tprow[mt][i] * C + (k - C)] = v;                           // unique writer
                } else if (k < 2 * C + Ap) atomicAdd(&DH[(size_t)n * Ap + (k - 2 * C)], v);      // unique writer per layer, layers in order: fire-and-forget add
            }
        if (!(2 * C <= 16 * PD3) && nt + 4 < NTK) { const int ntn[1] = {nt + 4}; wave_b_preload<1, PD3>(bq, W1t, NTK, ntn, 2 * C, lane); }
    }
}

// ------------------------------------------------------------------------------------------ layer backward, persistent form (n_resch = 64)
// Same arithmetic as k_layer_bwd, organised like k_layer_fwd_p (train_fwd.hip): 2 workgroups per CU walk contiguous ranges of
// 16-row tiles with the layer's TRANSPOSED weights resident in registers -- wave w owns column tile w of dg = dXout . Wr and
// column tiles {w, w + 4, w + 8} of d[x_cur | x_past | aux] = dZ . W1 (4 + 3 x 8 fragment float4s per lane) -- the next tile's
// rows (dXout parts, saved gate halves, skip-path gate grads) and scatter targets in flight under the current tile, a branch-free
// tile loop (clamped loads, out-of-range stores redirected to a scratch row: exact s_waitcnt vmcnt counts), LDS-only barriers,
// all fragment reads ahead of their MFMAs.  Both results leave through LDS as whole rows: dZ (512 B rows), the own-row input
// gradient (+ residual path), the pitch-tap part as one 256-byte row per tap row -- a plain store for the fixed blocks (unique
// writer), ONE full-row float-atomic instruction per row for the adaptive blocks (the accumulator layout gave 64-byte strips
// of four different rows per instruction) -- and the aux columns as row-contiguous atomics.
template <int NTK, bool LAST>      // NTK = Ktp / 16 column tiles of the input gradient (11 for n_resch 64, n_aux 39)
__global__ __launch_bounds__(256, 2) void k_layer_bwd_p(TrainParams p, TrainBwd bw, int l, int flags, int tiles, float* dummy) {     // flags: bit 1 XCD swizzle
    constexpr int C = 64;
    constexpr int ldx = ((C + 29) / 32) * 32 + 2, ldz = ((2 * C + 29) / 32) * 32 + 2, ldo = ((16 * NTK + 29) / 32) * 32 + 2;
    constexpr int NJ = (NTK + 3) / 4;                              // column tiles of the second contraction per wave
    constexpr bool HOIST = NTK == 8;                               // the auxiliary 1x1 at frame rate (TrainParams::hoist): D / G instead of the aux columns (train_common.h, tr_aux_bwd)
    extern __shared__ float sm[];
    // two staging buffers {Dx | Sg | Th | Dg}, [16][ldx] each; Dz [16][ldz]; Os [16][ldo] (the second contraction's outputs); hoist: Gp [4][16]
    float* Dz = sm + 8 * 16 * ldx;
    float* Os = Dz + 16 * ldz;
    float* Gp = Os + 16 * ldo;
    const TrLayer ly = p.layers[l];
    const int Ap = p.Ap, N1 = p.N1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    int t_first, t_count;
    tr_tile_range(blockIdx.x, gridDim.x, tiles, flags & 2, t_first, t_count);
    if (t_count <= 0) return;
    const int t_last = t_first + t_count - 1;
    const size_t rb = (size_t)b * N1;
    const size_t nDX = (size_t)p.B * N1 * C;
    const float* DAin = bw.DXA[0] + (size_t)(l + 1) * nDX + rb * C; const float* DBin = bw.DXB[0] + (size_t)(l + 1) * nDX + rb * C;
    float* DAout = bw.DXA[0] + (size_t)l * nDX + rb * C; float* DBout = bw.DXB[0] + (size_t)l * nDX + rb * C;
    const float* SG = p.SG + ((size_t)(l * p.B + b) * N1) * C;
    const float* TH = p.TH + ((size_t)(l * p.B + b) * N1) * C;
    const int win0 = N1 - p.BL;
    const float* DGS = bw.DGS + (size_t)b * p.BL * p.LC + (size_t)l * C;
    float* DZg = bw.DZ + ((size_t)l * p.B * N1 + rb) * 2 * C;
    float* DH = HOIST ? nullptr : bw.DHUP + rb * Ap;
    float* GWl = HOIST ? bw.GW + (size_t)(l * p.B + b) * N1 : nullptr;
    const int* taps = p.TAP + ly.tap_off + rb;
    // ---- resident weight fragments
    const float4* Wrt = p.wp + ly.wrt_f4; const float4* W1t = p.wp + ly.w1t_f4;
    float4 wr[4], w1[NJ][8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wr[ks] = LAST ? make_float4(0.f, 0.f, 0.f, 0.f) : Wrt[((size_t)ks * 4 + wave) * 64 + lane];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nt = wave + 4 * j < NTK ? wave + 4 * j : NTK - 1;          // (a wave without a j-th tile keeps a copy it never uses)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w1[j][ks] = W1t[((size_t)ks * NTK + nt) * 64 + lane];
    }
    // ---- staging in: thread -> (row srow, 16-byte piece sc4) of the four [16][C] arrays
    const int srow = tid >> 4, sc4 = tid & 15;
    float4 ra, rb2, rsg, rth, rdg;
    auto load_rows = [&](int t) {
        const int n = ly.s_out + t * 16 + srow;
        const int nn = n < N1 ? n : N1 - 1;
        // 32-bit element offsets (one batch item's rows x channels fit easily; the batch part is in the bases): one full-rate multiply and
        // the SAME offset register behind four uniform bases, instead of a 64-bit multiply-add and a 64-bit add per load
        const unsigned o = __umul24((unsigned)nn, (unsigned)C) + 4u * sc4;
        ra = *(const float4*)(DAin + o); rb2 = *(const float4*)(DBin + o);          // (rows of the last layer: finite garbage, not used)
        rsg = *(const float4*)(SG + o); rth = *(const float4*)(TH + o);
        const int nw = nn >= win0 ? nn - win0 : 0;
        rdg = *(const float4*)(DGS + (__umul24((unsigned)nw, (unsigned)p.LC) + 4u * sc4));
    };
    auto store_rows = [&](int t, float* B) {
        const int n = ly.s_out + t * 16 + srow;
        const bool in = n < N1;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 dx = (in && !LAST) ? make_float4(ra.x + rb2.x, ra.y + rb2.y, ra.z + rb2.z, ra.w + rb2.w) : z;
        const float4 sg = in ? rsg : z, th = in ? rth : z, dg = (in && n >= win0) ? rdg : z;
        float* d0 = B + (size_t)srow * ldx + 4 * sc4;
        *(float2*)d0 = make_float2(dx.x, dx.y); *(float2*)(d0 + 2) = make_float2(dx.z, dx.w);
        float* d1 = d0 + 16 * ldx; *(float2*)d1 = make_float2(sg.x, sg.y); *(float2*)(d1 + 2) = make_float2(sg.z, sg.w);
        float* d2 = d1 + 16 * ldx; *(float2*)d2 = make_float2(th.x, th.y); *(float2*)(d2 + 2) = make_float2(th.z, th.w);
        float* d3 = d2 + 16 * ldx; *(float2*)d3 = make_float2(dg.x, dg.y); *(float2*)(d3 + 2) = make_float2(dg.z, dg.w);
    };
    // ---- staging out.  dZ: thread -> 8-byte pieces (rows zrow, zrow + 4, + 8, + 12; column zc2) of the [16][128] tile;
    //      input gradient: wave w owns rows 4w .. 4w+3, lane = channel: one 256-byte row per instruction
    const int zrow = tid >> 6, zc2 = (tid & 63) * 2;
    float* const dmy = dummy + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * 128;       // two 512-byte scratch rows per workgroup
    int tprow[4], tpnext[4];
    auto load_taps = [&](int t, int (&tp)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int n = ly.s_out + t * 16 + 4 * wave + i; tp[i] = taps[n < N1 ? n : N1 - 1]; }
    };
    // aux hoist: the WJ entries of this lane's four dZ rows and its four PA values -- c*: the tile in hand, n*: the next tile's, in flight
    float2 cwj[4], nwj[4]; float cpa[4], npa[4]; int cpoff = 0, npoff = 0;
    auto load_aux = [&](int t, float2 (&wj)[4], float (&pa)[4], int& poff) {
        const int n0 = ly.s_out + t * 16;
        poff = tr_pa_off(p, l, b, n0);
        const float2* w = p.WJ + n0 + 4 * (lane >> 4);             // (16 rows of padding behind row N1 - 1)
        const float* q4 = p.PA + poff + 16 * wave + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) wj[i] = w[i];
        pa[0] = q4[0]; pa[1] = q4[C]; pa[2] = q4[2 * C]; pa[3] = q4[3 * C];
    };
    load_rows(t_first);
    if constexpr (HOIST) load_aux(t_first, cwj, cpa, cpoff);
    load_taps(t_first, tprow);
    store_rows(t_first, sm);
    const int arow = lane & 15, ak = lane >> 4;
    const int c = 16 * wave + (lane & 15);
    for (int ti = 0; ti < t_count; ++ti) {
        const int t = t_first + ti, n0 = ly.s_out + t * 16;
        float* Dx = sm + (ti & 1) * 4 * 16 * ldx; float* Sg = Dx + 16 * ldx; float* Th = Sg + 16 * ldx; float* Dg = Th + 16 * ldx;
        { const int tn = t + 1 < t_last ? t + 1 : t_last; load_rows(tn); load_taps(tn, tpnext); if constexpr (HOIST) load_aux(tn, nwj, npa, npoff); }      // (past the range: a harmless reload)
        TR_LDS_BARRIER();                                          // this tile's staged rows complete; Dz / Os free (readers: previous trip)
        // ---- dg = dXout . Wr + DGS ; dz = dg * gate'
        float xa[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const float* ap = Dx + (size_t)arow * ldx + 16 * ks + ak; xa[ks][0] = ap[0]; xa[ks][1] = ap[4]; xa[ks][2] = ap[8]; xa[ks][3] = ap[12]; }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 a0 = (f32x4){0, 0, 0, 0}, a1 = (f32x4){0, 0, 0, 0};
        if (!LAST) {
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][0], wr[ks].x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][0], wr[ks + 1].x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][1], wr[ks].y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][1], wr[ks + 1].y, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][2], wr[ks].z, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][2], wr[ks + 1].z, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][3], wr[ks].w, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][3], wr[ks + 1].w, a1, 0, 0, 0);
            }
        }
        float dzs[4], dzt[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * (lane >> 4) + i;
            const float dg = (a0[i] + a1[i]) + Dg[(size_t)r * ldx + c];
            const float sg = Sg[(size_t)r * ldx + c], th = Th[(size_t)r * ldx + c];
            tr_gate_bwd(dg, sg, th, dzs[i], dzt[i]);              // (th: the gate product the Th tile holds)
            Dz[(size_t)r * ldz + c] = dzs[i];
            Dz[(size_t)r * ldz + C + c] = dzt[i];
        }
        float dacc[4] = {0.f, 0.f, 0.f, 0.f}, eacc[2] = {0.f, 0.f};
        if constexpr (HOIST) {      // the frame-rate aux term's backward (same arithmetic as k_stack_bwd)
            const TrAuxBwd ab = tr_aux_bwd(cwj, cpa, dzs, dzt, lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) dacc[k] = ab.d[k];
            eacc[0] = ab.e[0]; eacc[1] = ab.e[1];
            if ((lane & 15) == 0) { float* gq = Gp + 16 * wave + 4 * (lane >> 4); gq[0] = ab.gp[0]; gq[1] = ab.gp[1]; gq[2] = ab.gp[2]; gq[3] = ab.gp[3]; }
        }
        TR_LDS_BARRIER();
        // ---- d[x_cur | x_past | aux] = dZ . W1
        float za[8][4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { const float* zp = Dz + (size_t)arow * ldz + 16 * ks + ak; za[ks][0] = zp[0]; za[ks][1] = zp[4]; za[ks][2] = zp[8]; za[ks][3] = zp[12]; }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][0], w1[j][ks].x, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][1], w1[j][ks].y, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][2], w1[j][ks].z, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][3], w1[j][ks].w, acc[j], 0, 0, 0);
        }
        {   // dZ rows to global (512 B each), under the contraction: thread -> rows zrow + 4k, 8-byte piece zc2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = zrow + 4 * k;
                float* d = DZg + (__umul24((unsigned)(n0 + r), 2u * C) + zc2);
                d = n0 + r < N1 ? d : dmy + zc2;
                *(float2*)d = *(const float2*)(Dz + (size_t)r * ldz + zc2);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nt = wave + 4 * j;
            if (nt >= NTK) continue;                               // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) Os[(size_t)(4 * (lane >> 4) + i) * ldo + 16 * nt + (lane & 15)] = acc[j][i];
        }
        TR_LDS_BARRIER();
        // ---- outputs as whole rows: wave w owns rows 4w .. 4w+3, lane = channel
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * wave + i, n = n0 + r;
            const bool in = n < N1;
            const float own = Os[(size_t)r * ldo + lane] + Dx[(size_t)r * ldx + lane];                 // + residual path
            float* da = DAout + (__umul24((unsigned)n, (unsigned)C) + lane);
            *(in ? da : dmy + lane) = own;
            const float past = Os[(size_t)r * ldo + C + lane];
            float* db = DBout + (__umul24((unsigned)tprow[i], (unsigned)C) + lane);
            db = in ? db : dmy + 128 + lane;
            if (ly.adaptive) atomicAdd(db, past);                                                       // gather backward (collisions)
            else *db = past;                                                                            // unique writer
            if constexpr (!HOIST) if (lane < Ap) {                                                      // (Ap <= 64)
                float* dh = DH + (__umul24((unsigned)n, (unsigned)Ap) + lane);
                atomicAdd(in ? dh : dmy + 192 + lane, Os[(size_t)r * ldo + 2 * C + lane]);               // unique writer per layer, layers in order
            }
        }
        if constexpr (HOIST) {      // D_l[frame][gate row] += this tile's share; the tile's 16 G values
            if (lane < 16) {
                float* dp = bw.DPA + cpoff + 16 * wave + lane;
                atomicAdd(dp, dacc[0]); atomicAdd(dp + C, dacc[1]); atomicAdd(dp + 2 * C, dacc[2]); atomicAdd(dp + 3 * C, dacc[3]);
                float* ep = bw.EB + (size_t)(l * TR_EB_SLOTS + (blockIdx.x & (TR_EB_SLOTS - 1))) * 2 * C + 16 * wave + lane;
                atomicAdd(ep, eacc[0]); atomicAdd(ep + C, eacc[1]);
            }
            if (tid < 16 && n0 + tid < N1) GWl[n0 + tid] = (Gp[tid] + Gp[16 + tid]) + (Gp[32 + tid] + Gp[48 + tid]);
        }
        store_rows(t + 1 < t_last ? t + 1 : t_last, sm + ((ti + 1) & 1) * 4 * 16 * ldx);
#pragma unroll
        for (int i = 0; i < 4; ++i) tprow[i] = tpnext[i];
        if constexpr (HOIST) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { cwj[i] = nwj[i]; cpa[i] = npa[i]; }
            cpoff = npoff;
        }
    }
}

// ------------------------------------------------------------------------------------------ weight gradients, v2
// One launch covers every layer (blockIdx.y) and every time chunk (blockIdx.x).  A workgroup (4 waves)
// computes the WHOLE output block dW[M][N] of its layer for its chunk of rows: wave w owns m-tiles
// {w, w+4, ...} x all n-tiles, operands staged once per 32 rows with 16-byte loads (no re-reads across
// output tiles), partial result written to its slab.  dW = A^T B with
//   A[row][m]  : plain array, or the sum of two arrays (grad wrt a layer output = own-row + scattered part)
//   B[row][n]  : plain | relu(array) | product of two arrays (gate = sigma*tanh) | [x_cur | x_past | aux | 0]
//                | one-hot of the row's sample class (k_wgrad3 only)
struct Wg2 {
    const float* A; const float* A2; size_t A_lstride; int lda, M;        // per-layer stride (floats)
    int rowsA;                                                              // rows per batch item in A's array
    int bmode; const float* B1; const float* B2; size_t B_lstride; int ldb, N, Nvalid;
    int rowsB;
    const float* hup; const int* tap; int C, Ap;
    const int* xc;                                                          // bmode 4: B[t][q] = (xc[row] == q), one-hot of the sample class
    int nb, nlayers;
    float* slab; int gstage;
    int row0A[TR_MAXL], row0B[TR_MAXL], R[TR_MAXL];                         // window per layer
    int goff[TR_MAXL], gbias[TR_MAXL], tap_off[TR_MAXL], dil[TR_MAXL];
    int ldc, ncol_groups;                                                   // post: N split into column groups (blockIdx.z)
};

struct Wg2L {            // per-workgroup scalars hoisted out of the kernel-argument arrays
    const float* A; const float* A2; const float* B1; const float* B2; const float* hup; const int* tap;
    int lda, ldb, M, Ng, Nvalid, rowsA, rowsB, row0A, row0B, C, Ap, dil, nb, ncol0, rend;
    unsigned uR;
};

// tap rows of the stage starting at rs (pitch-dependent gather source rows), loaded one stage ahead of their use
template <int NB>
__device__ __forceinline__ void wg_taps(const Wg2L& q, int rs, int (&tp)[NB], bool b_act, int b_row0, int b_rstep) {
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int r = b_row0 + k * b_rstep, rr = rs + r;
        int v = 0;
        if (b_act && r < 32 && rr < q.rend) {
            const unsigned b = q.nb > 1 ? (unsigned)rr / q.uR : 0u; const int i = rr - (int)(b * q.uR);
            const int nloc = q.row0B + i;
            v = q.tap ? q.tap[(size_t)b * q.rowsB + nloc] : nloc - q.dil;
        }
        tp[k] = v;
    }
}

template <int BMODE, int NA, int NB>
__device__ __forceinline__ void wg_fetch(const Wg2L& q, int rs, float4 (&ra)[NA], float4 (&ra2)[NA], float4 (&rb)[NB], float4 (&rb2)[NB], const int (&tpv)[NB],
                                         bool a_act, int a_row0, int a_rstep, int a_col, bool b_act, int b_row0, int b_rstep, int b_col) {
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int r = a_row0 + k * a_rstep, rr = rs + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), v2 = v;
        if (a_act && r < 32 && rr < q.rend && a_col < q.M) {
            const unsigned b = q.nb > 1 ? (unsigned)rr / q.uR : 0u; const int i = rr - (int)(b * q.uR);
            const size_t o = ((size_t)b * q.rowsA + q.row0A + i) * q.lda + a_col;
            v = *(const float4*)(q.A + o);
            if (q.A2) v2 = *(const float4*)(q.A2 + o);
        }
        ra[k] = v; ra2[k] = v2;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int r = b_row0 + k * b_rstep, rr = rs + r;
        const int n = q.ncol0 + b_col;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f), v2 = make_float4(1.f, 1.f, 1.f, 1.f);
        if (b_act && r < 32 && rr < q.rend && b_col < q.Ng && n < q.Nvalid) {
            const unsigned b = q.nb > 1 ? (unsigned)rr / q.uR : 0u; const int i = rr - (int)(b * q.uR);
            const int nloc = q.row0B + i;
            const size_t row = (size_t)b * q.rowsB + nloc;
            if (BMODE <= 1) v = *(const float4*)(q.B1 + row * q.ldb + n);
            else if (BMODE == 2) { v = *(const float4*)(q.B1 + row * q.ldb + n); v2 = *(const float4*)(q.B2 + row * q.ldb + n); }
            else {
                if (n < q.C) v = *(const float4*)(q.B1 + row * q.C + n);
                else if (n < 2 * q.C) v = *(const float4*)(q.B1 + ((size_t)b * q.rowsB + tpv[k]) * q.C + (n - q.C));
                else v = *(const float4*)(q.hup + row * q.Ap + (n - 2 * q.C));
            }
        }
        rb[k] = v; rb2[k] = v2;
    }
}

template <int BMODE, int MPW, int NTMAX>
__global__ __launch_bounds__(256) void k_wgrad2(Wg2 w, int nch) {
    extern __shared__ float sm[];
    constexpr int RS = 32;
    constexpr int NA = 2 * MPW;              // staging passes of A: 32 rows / (256 / (Mp/4)) with Mp <= 64*MPW
    constexpr int NB = (NTMAX + 1) / 2 + 1;  // staging passes of B: ceil(32 / floor(256 / (Np/4))), Np <= 16*NTMAX
    const int y = blockIdx.y, ch = blockIdx.x, zg = blockIdx.z;
    const int Mp = (w.M + 15) & ~15, Ng = w.N / w.ncol_groups, Np = (Ng + 15) & ~15;
    const int ldA = tr_ldt(Mp), ldB = tr_ldt(Np);
    float* As = sm; float* Bs = sm + RS * ldA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MT = Mp / 16, NT = Np / 16;
    const int Rl = w.R[y];
    const int64_t total = (int64_t)Rl * w.nb;
    const int64_t per = ((total + nch - 1) / nch + RS - 1) / RS * RS;
    const int rbeg = (int)(per * ch), rend = (int)(per * ch + per < total ? per * ch + per : total);
    Wg2L q;
    q.A = w.A + (size_t)y * w.A_lstride; q.A2 = w.A2 ? w.A2 + (size_t)y * w.A_lstride : nullptr;
    q.B1 = w.B1 + (size_t)y * w.B_lstride; q.B2 = w.B2 ? w.B2 + (size_t)y * w.B_lstride : nullptr;
    q.hup = w.hup; q.tap = (w.tap && w.tap_off[y] >= 0) ? w.tap + w.tap_off[y] : nullptr;
    q.lda = w.lda; q.ldb = w.ldb; q.M = w.M; q.Ng = Ng; q.Nvalid = w.Nvalid; q.rowsA = w.rowsA; q.rowsB = w.rowsB;
    q.row0A = w.row0A[y]; q.row0B = w.row0B[y]; q.C = w.C; q.Ap = w.Ap; q.dil = w.dil[y]; q.nb = w.nb; q.ncol0 = zg * Ng; q.rend = rend;
    q.uR = (unsigned)(Rl > 0 ? Rl : 1);
    const int gbias = zg == 0 ? w.gbias[y] : -1, goff = w.goff[y];
    const int A4 = Mp / 4, B4 = Np / 4;
    const int a_col = (tid % A4) * 4, a_row0 = tid / A4, a_rstep = 256 / A4;
    const int b_rstep = 256 / B4 > 0 ? 256 / B4 : 1;
    const int b_col = (tid % B4) * 4, b_row0 = tid / B4;
    const bool a_act = tid < a_rstep * A4, b_act = tid < b_rstep * B4;
    f32x4 acc[MPW][NTMAX];
#pragma unroll
    for (int a = 0; a < MPW; ++a)
#pragma unroll
        for (int b = 0; b < NTMAX; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
    float csum = 0.f;
    float4 ra[NA], ra2[NA], rb[NB], rb2[NB];
    int tpv[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) tpv[k] = 0;
    if (BMODE == 3 && rbeg < rend) wg_taps<NB>(q, rbeg, tpv, b_act, b_row0, b_rstep);
    if (rbeg < rend) wg_fetch<BMODE, NA, NB>(q, rbeg, ra, ra2, rb, rb2, tpv, a_act, a_row0, a_rstep, a_col, b_act, b_row0, b_rstep, b_col);
    if (BMODE == 3 && rbeg + RS < rend) wg_taps<NB>(q, rbeg + RS, tpv, b_act, b_row0, b_rstep);
    const int g = lane >> 4, cl = lane & 15;
    for (int rs = rbeg; rs < rend; rs += RS) {
        // registers -> LDS
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int r = a_row0 + k * a_rstep;
            if (a_act && r < RS) {
                float4 v = ra[k];
                v.x += ra2[k].x; v.y += ra2[k].y; v.z += ra2[k].z; v.w += ra2[k].w;
                *(float4*)(As + (size_t)r * ldA + a_col) = v;
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int r = b_row0 + k * b_rstep;
            if (b_act && r < RS) {
                float4 v = rb[k];
                if (BMODE == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (BMODE == 2) { v.x *= rb2[k].x; v.y *= rb2[k].y; v.z *= rb2[k].z; v.w *= rb2[k].w; }
                *(float4*)(Bs + (size_t)r * ldB + b_col) = v;
            }
        }
        __syncthreads();
        // next stage's rows fly while the matrix cores work on this one
        if (rs + RS < rend) wg_fetch<BMODE, NA, NB>(q, rs + RS, ra, ra2, rb, rb2, tpv, a_act, a_row0, a_rstep, a_col, b_act, b_row0, b_rstep, b_col);
        if (BMODE == 3 && rs + 2 * RS < rend) wg_taps<NB>(q, rs + 2 * RS, tpv, b_act, b_row0, b_rstep);   // gather rows of the stage after next
        if (gbias >= 0 && tid < q.M) { float s = 0.f; for (int r = 0; r < RS; ++r) s += As[r * ldA + tid]; csum += s; }
#pragma unroll
        for (int ks = 0; ks < RS / 4; ++ks) {
            float bfr[NTMAX];
#pragma unroll
            for (int nt = 0; nt < NTMAX; ++nt) bfr[nt] = nt < NT ? Bs[(4 * ks + g) * ldB + 16 * nt + cl] : 0.f;
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                const int mt = wave + 4 * mi;
                if (mt < MT) {
                    const float a = As[(4 * ks + g) * ldA + 16 * mt + cl];
#pragma unroll
                    for (int nt = 0; nt < NTMAX; ++nt)
                        if (nt < NT) acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[nt], acc[mi][nt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    float* out = w.slab + (size_t)ch * w.gstage;
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int mt = wave + 4 * mi;
        if (mt >= MT) continue;
#pragma unroll
        for (int nt = 0; nt < NTMAX; ++nt) {
            if (nt >= NT) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * mt + 4 * (lane >> 4) + i, n = 16 * nt + (lane & 15);
                if (m < q.M && n < Ng) out[goff + (size_t)m * w.ldc + q.ncol0 + n] = acc[mi][nt][i];
            }
        }
    }
    if (gbias >= 0 && tid < q.M) out[gbias + tid] = csum;
}

template <int BMODE, int MPW, int NTMAX>
static int launch_wgrad2(const Wg2& w, int nch, hipStream_t stream) {
    const int Mp = (w.M + 15) & ~15, Np = ((w.N / w.ncol_groups) + 15) & ~15;
    if (Mp / 16 > 4 * MPW || Np / 16 > NTMAX) return -1;
    const size_t lds = (size_t)32 * (tr_ldt(Mp) + tr_ldt(Np)) * sizeof(float);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_wgrad2<BMODE, MPW, NTMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_wgrad2<BMODE, MPW, NTMAX>), dim3(nch, w.nlayers, w.ncol_groups), dim3(256), lds, stream, w, nch);
    return 0;
}
template <int BMODE>
static bool wgrad2_mode(const Wg2& w, int nch, hipStream_t stream) {
    if (launch_wgrad2<BMODE, 1, 4>(w, nch, stream) == 0) return true;
    if (launch_wgrad2<BMODE, 4, 4>(w, nch, stream) == 0) return true;
    if (launch_wgrad2<BMODE, 2, 12>(w, nch, stream) == 0) return true;
    if (launch_wgrad2<BMODE, 4, 8>(w, nch, stream) == 0) return true;       // 256 x 128 blocks (n_resch up to 128 with column groups)
    return false;
}
// ------------------------------------------------------------------------------------------ weight gradients, v3
// Same contraction and slab layout as k_wgrad2, for the geometries where the tile counts are known at compile time
// (M = 64*MPW rows of dW -> every wave owns exactly MPW m-tiles, N per column group = 16*NT): no per-tile guards in
// the matrix-core loop, no branches around the staging loads (out-of-range rows are clamped and zeroed, the three
// sources of the [x_cur | x_past | aux] operand become one per-thread base/stride), bias column sums taken from the
// staging registers instead of 32 LDS reads per stage.  k_wgrad2 stays as the generic fallback.
// upsampling kernel grad: dw[j] = sum_{a,f} dH[a, U f + j] h[a,f];  db = sum dH   (qpnet.py:134-158)
// One workgroup per (frame, batch item): thread j < U owns sample U f + j of the frame, reads its dH row (Ap contiguous floats: the
// workgroup's reads are one contiguous U*Ap*4-byte block) and dots it with the frame's feature column held in LDS; one global atomic per
// (j, frame) instead of an LDS atomic per element (the element-wise version was 20 us for 3.8 MB: contended LDS atomics on 110 bins).
struct UpArgs { const float* h; const float* DHUP; float* gflat; int64_t up_w, up_b; float gscale; int U, Ap, A, F, N1, nfr, B; };
static UpArgs up_args(const TrainParams& p, const TrainBwd& bw) {
    UpArgs u; u.h = p.h; u.DHUP = bw.DHUP; u.gflat = bw.gflat; u.gscale = bw.gscale; u.U = p.U; u.Ap = p.Ap; u.A = p.A; u.F = p.F; u.N1 = p.N1;
    u.up_w = p.up_w; u.up_b = p.up_b; u.B = p.B;
    const int64_t q0 = (int64_t)p.F * p.U - p.N1;
    u.nfr = p.U > 0 ? (int)(((int64_t)p.F * p.U - 1) / p.U - q0 / p.U + 1) : 0;       // frames that the N1 rows touch
    return u;
}
// fx: frame (counted from the first one the rows touch), b: batch item; nthr threads (a multiple of 64, <= 256), all of which call this
__device__ __forceinline__ void up_bwd_body(const UpArgs& p, int fx, int b, int tid, int nthr) {
    __shared__ float hcol[64];
    __shared__ float red[4];
    const int U = p.U, Ap = p.Ap, A = p.A;
    const int64_t q0 = (int64_t)p.F * U - p.N1;           // h_up sample index of row 0
    const int f = (int)(q0 / U) + fx;                     // frames touched: f0 .. f0 + nfr - 1
    float dsum = 0.f;
    for (int a0 = 0; a0 < A; a0 += 64) {                  // (one pass for the usual 39 features)
        __syncthreads();
        if (tid < 64) hcol[tid] = (a0 + tid < A && f < p.F) ? p.h[((size_t)b * A + a0 + tid) * p.F + f] : 0.f;
        __syncthreads();
        for (int j = tid; j < U; j += nthr) {
            const int64_t n = (int64_t)f * U + j - q0;    // local row
            if (n < 0 || n >= p.N1 || f >= p.F) continue;
            const float* row = p.DHUP + ((size_t)b * p.N1 + n) * Ap + a0;
            float acc = 0.f;
            const int na = A - a0 < 64 ? A - a0 : 64;
            // the row's (up to 16) float4 words requested TOGETHER, then summed: with a run-time trip count the loop was load -> wait -> add
            // per word, ten memory latencies in a row -- 73 us beside the memory-bound dW1 launch (11 us alone)
            const int nq = na >> 2;
            float4 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = *(const float4*)(row + 4 * (q < nq ? q : 0));
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < nq) { acc += v[q].x * hcol[4 * q] + v[q].y * hcol[4 * q + 1] + v[q].z * hcol[4 * q + 2] + v[q].w * hcol[4 * q + 3]; dsum += (v[q].x + v[q].y) + (v[q].z + v[q].w); }
            for (int a = 4 * nq; a < na; ++a) { const float v1 = row[a]; acc += v1 * hcol[a]; dsum += v1; }
            atomicAdd(&p.gflat[p.up_w + j], acc * p.gscale);
        }
    }
    for (int s = 32; s >= 1; s >>= 1) dsum += __shfl_xor(dsum, s);
    if ((tid & 63) == 0) red[tid >> 6] = dsum;
    __syncthreads();
    if (tid == 0) { float a = 0.f; for (int k = 0; k < nthr / 64; ++k) a += red[k]; atomicAdd(&p.gflat[p.up_b], a * p.gscale); }
}

// (ch, y, zg) = (time chunk, layer, column group) of this workgroup: blockIdx of the one-contraction launches, decoded from a flat index in k_wgrad_tail
template <int BMODE, int MPW, int NT, bool TWO_A>
__device__ __forceinline__ void wgrad3_body(const Wg2& w, const int nch, const int ch, const int y, const int zg) {
    extern __shared__ float sm[];
    constexpr int RS = 32, Mp = 64 * MPW, Np = 16 * NT;
    constexpr int ldA = ((Mp + 15) / 32) * 32 + 16, ldB = ((Np + 15) / 32) * 32 + 16;       // tr_ldt
    constexpr int A4 = Mp / 4, ARS = 256 / A4, NA = RS / ARS;                               // A: every thread active, NA full passes
    constexpr int B4 = Np / 4, BRS = 256 / B4, NB = (RS + BRS - 1) / BRS;                   // B: threads < BRS*B4 active
    static_assert(256 % A4 == 0 && RS % ARS == 0, "A staging must tile the stage's rows exactly");
    constexpr int BUF = RS * (ldA + ldB);                                                   // floats of one stage buffer
    float* As = sm; float* Bs = sm + RS * ldA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Rl = w.R[y];
    const int64_t total = (int64_t)Rl * w.nb;
    const int64_t per = ((total + nch - 1) / nch + RS - 1) / RS * RS;
    const int rbeg = (int)(per * ch), rend = (int)(per * ch + per < total ? per * ch + per : total);
    const float* A = w.A + (size_t)y * w.A_lstride;
    const float* A2 = TWO_A ? w.A2 + (size_t)y * w.A_lstride : nullptr;      // TWO_A: the A operand is the sum of two arrays (a template flag: the second
                                                                               // staging set costs 32 registers the one-array launches need for a third workgroup per CU)
    const float* B1 = BMODE == 4 ? nullptr : w.B1 + (size_t)y * w.B_lstride;
    const float* B2 = w.B2 ? w.B2 + (size_t)y * w.B_lstride : nullptr;
    const int* tap = BMODE == 3 ? w.tap + w.tap_off[y] : nullptr;      // BMODE 3: a tap table for every layer (the launcher checks)
    const int row0A = w.row0A[y], row0B = w.row0B[y], ncol0 = zg * Np;
    const unsigned uR = (unsigned)(Rl > 0 ? Rl : 1);
    const bool multi = w.nb > 1;
    const int gbias = zg == 0 ? w.gbias[y] : -1, goff = w.goff[y];
    const int a_col = (tid % A4) * 4, a_row0 = tid / A4;
    const int b_col = (tid % B4) * 4, b_row0 = tid / B4;
    const int n = ncol0 + b_col;
    const bool b_act = tid < BRS * B4;
    const bool b_pad = n >= w.Nvalid;                           // zero padding columns of the operand (K padded to 16)
    // B operand source of this thread's four columns
    const float* bbase; int bstride; bool use_tap = false;
    if (BMODE == 3) {
        if (n < w.C) { bbase = B1 + n; bstride = w.C; }
        else if (n < 2 * w.C) { bbase = B1 + (n - w.C); bstride = w.C; use_tap = true; }
        else { bbase = w.hup + (b_pad ? 0 : n - 2 * w.C); bstride = w.Ap; }
    } else if (BMODE == 4) { bbase = nullptr; bstride = 0; }
    else { bbase = B1 + n; bstride = w.ldb; }
    const ptrdiff_t b2off = (BMODE == 2) ? (B2 - B1) : 0;

    f32x4 acc[MPW][NT];
#pragma unroll
    for (int a = 0; a < MPW; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = (f32x4){0, 0, 0, 0};
    float4 cs4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra[NA], ra2[NA], rb[NB], rb2[NB];
    int tpv[NB];
    int rbi[NB];                     // BMODE 4: sample class of the staged row (the one-hot is made when the stage goes to LDS)
    unsigned okA = 0, okB = 0;       // bit k: staged row k is a real row.  The loaded registers are NOT touched before the stage goes to
                                     // LDS: a select on a just-loaded value puts an s_waitcnt vmcnt in front of the matrix-core loop
                                     // (measured with in-kernel stamps: 6-8 k cycles per stage in the skip 1x1 launch)

    // Staging addresses.  A stage that lies inside one batch item and ends before rend -- all but the last one or two of a chunk -- takes
    // the FAST form: one uniform base per operand and stage plus per-thread offsets fixed for the whole kernel.  The generic form below
    // (row clamp, per-row batch split, 64-bit multiplies: ~650 instructions = 5.7 k cycles per stage in the dW1 launch, as long as its
    // 176 MFMAs) only runs for the others.
    const unsigned a_voff = (unsigned)a_row0 * (unsigned)w.lda + (unsigned)a_col, a_kstep = (unsigned)ARS * (unsigned)w.lda;
    const unsigned b_voff = (unsigned)b_row0 * (unsigned)w.ldb + (unsigned)n, b_kstep = (unsigned)BRS * (unsigned)w.ldb;

    // rows of a stage: r in [0,32); rows at or past rend are clamped to the last valid row and zeroed
    auto rowsplit = [&](int rr, int& b, int& i) { if (multi) { b = (int)((unsigned)rr / uR); i = rr - b * (int)uR; } else { b = 0; i = rr; } };
    auto stage_fast = [&](int rs, int& b0, int& i0) {       // uniform
        rowsplit(rs, b0, i0);
        return rs + RS <= rend && i0 + RS <= (int)uR;
    };
    // (every thread loads its row's tap entry, needed or not: a per-thread condition around the load becomes a branch with an
    //  s_waitcnt behind each load, which also drains the operand loads issued just before -- the next stage's prefetch no longer
    //  ran under the MFMAs)
    auto taps = [&](int rs) {
        int b0, i0;
        if (stage_fast(rs, b0, i0)) {
            const int* tst = tap + ((size_t)b0 * w.rowsB + row0B + i0);
#pragma unroll
            for (int k = 0; k < NB; ++k) { const int r = b_row0 + k * BRS; tpv[k] = tst[r < RS ? r : b_row0]; }
            return;
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int r = b_row0 + k * BRS;
            int rr = rs + r; rr = rr < rend ? rr : rend - 1;
            int b, i; rowsplit(rr, b, i);
            const int nloc = row0B + i;
            tpv[k] = tap[(size_t)b * w.rowsB + nloc];
        }
    };
    // Both forms only compute ADDRESSES (and the row masks) under the uniform branch; the loads themselves are issued once, behind it.
    // [Loads in both arms made the merged values phi nodes: the copies at the join need the data, i.e. an s_waitcnt vmcnt(0) in front of
    //  the matrix-core loop -- the whole memory latency exposed in every stage.]
    auto fetch = [&](int rs) {
        const float* pa[NA]; const float* pb[NB]; const int* pi[NB];
        int b0, i0;
        if (stage_fast(rs, b0, i0)) {
            const float* Ast = A + ((size_t)b0 * w.rowsA + row0A + i0) * w.lda;
#pragma unroll
            for (int k = 0; k < NA; ++k) pa[k] = Ast + (k * a_kstep + a_voff);
            okA = ~0u;
            const size_t brow = (size_t)b0 * w.rowsB + row0B + i0;
            const float* bsrc = BMODE == 3 ? bbase + (size_t)b0 * w.rowsB * bstride : B1 + brow * w.ldb;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = b_row0 + k * BRS;
                if (BMODE == 4) pi[k] = w.xc + brow + (r < RS ? r : b_row0);
                else if (BMODE == 3) {
                    const unsigned row = use_tap ? (unsigned)tpv[k] : (unsigned)(row0B + i0 + (r < RS ? r : b_row0));
                    pb[k] = bsrc + __umul24(row, (unsigned)bstride);
                } else pb[k] = bsrc + ((r < RS ? k * b_kstep : 0u) + b_voff);
            }
            okB = b_pad ? 0u : ~0u;
        } else {
            okA = 0; okB = 0;
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                const int r = a_row0 + k * ARS;
                int rr = rs + r; const bool ok = rr < rend; rr = ok ? rr : rend - 1;
                int b, i; rowsplit(rr, b, i);
                pa[k] = A + (((size_t)b * w.rowsA + row0A + i) * w.lda + a_col);
                okA |= ok ? 1u << k : 0u;
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = b_row0 + k * BRS;
                int rr = rs + r; const bool ok = r < RS && rr < rend && !b_pad; rr = rr < rend ? rr : rend - 1;
                int b, i; rowsplit(rr, b, i);
                const size_t row = (size_t)b * w.rowsB + ((BMODE == 3 && use_tap) ? tpv[k] : row0B + i);
                if (BMODE == 4) pi[k] = w.xc + row; else pb[k] = bbase + row * bstride;
                okB |= ok ? 1u << k : 0u;
            }
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            ra[k] = *(const float4*)pa[k];
            if (TWO_A) ra2[k] = *(const float4*)(pa[k] + (A2 - A));
        }
        if (b_act) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (BMODE == 4) rbi[k] = *pi[k];
                else {
                    rb[k] = *(const float4*)pb[k];
                    if (BMODE == 2) rb2[k] = *(const float4*)(pb[k] + b2off);
                }
            }
        }
    };
    if (rbeg < rend) {
        if (BMODE == 3) taps(rbeg);
        fetch(rbeg);
        if (BMODE == 3 && rbeg + RS < rend) taps(rbeg + RS);
    }
    const int g = lane >> 4, cl = lane & 15;
    // staging registers -> LDS buffer `buf`
    auto to_lds = [&](int buf) {
        float* Ab = As + buf * BUF; float* Bb = Bs + buf * BUF;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            float4 v = ra[k];
            if (TWO_A) { v.x += ra2[k].x; v.y += ra2[k].y; v.z += ra2[k].z; v.w += ra2[k].w; }
            { const bool ok = (okA >> k) & 1u; v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; }     // (component-wise: a select between float4 objects makes the staging arrays addressable -> scratch)
            cs4.x += v.x; cs4.y += v.y; cs4.z += v.z; cs4.w += v.w;
            *(float4*)(Ab + (size_t)(a_row0 + k * ARS) * ldA + a_col) = v;
        }
        if (b_act) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int r = b_row0 + k * BRS;
                if (r < RS) {
                    float4 v;
                    if (BMODE == 4) { const int dlt = rbi[k] - n; v = make_float4(dlt == 0 ? 1.f : 0.f, dlt == 1 ? 1.f : 0.f, dlt == 2 ? 1.f : 0.f, dlt == 3 ? 1.f : 0.f); }
                    else v = rb[k];
                    if (BMODE == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    else if (BMODE == 2) { v.x *= rb2[k].x; v.y *= rb2[k].y; v.z *= rb2[k].z; v.w *= rb2[k].w; }
                    { const bool ok = (okB >> k) & 1u; v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f; }
                    *(float4*)(Bb + (size_t)r * ldB + b_col) = v;
                }
            }
        }
    };
    auto ksteps = [&](int buf, int k0, int k1) {
        const float* Ab = As + buf * BUF; const float* Bb = Bs + buf * BUF;
#pragma unroll
        for (int ks = k0; ks < k1; ++ks) {
            float bfr[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bfr[nt] = Bb[(4 * ks + g) * ldB + 16 * nt + cl];
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) {
                const float a = Ab[(4 * ks + g) * ldA + 16 * (wave + 4 * mi) + cl];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mi][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bfr[nt], acc[mi][nt], 0, 0, 0);
            }
        }
    };
    for (int rs = rbeg; rs < rend; rs += RS) {
        to_lds(0);
        __syncthreads();
        // next stage's rows fly while the matrix cores work on this one
        if (rs + RS < rend) {
            fetch(rs + RS);
            if (BMODE == 3 && rs + 2 * RS < rend) taps(rs + 2 * RS);       // gather rows of the stage after next
        }
        ksteps(0, 0, RS / 4);
        __syncthreads();
    }
    float* out = w.slab + (size_t)ch * w.gstage;
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) {
        const int mt = wave + 4 * mi;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * mt + 4 * (lane >> 4) + i, nn = 16 * nt + (lane & 15);
                out[goff + (size_t)m * w.ldc + ncol0 + nn] = acc[mi][nt][i];
            }
        }
    }
    if (gbias >= 0) {          // bias grads: column sums of A, reduced over the ARS row groups of the staging layout
        *(float4*)(sm + (size_t)a_row0 * Mp + a_col) = cs4;
        __syncthreads();
        if (tid < Mp) { float s = 0.f; for (int r = 0; r < ARS; ++r) s += sm[r * Mp + tid]; out[gbias + tid] = s; }
    }
}
template <int BMODE, int MPW, int NT, bool TWO_A, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void k_wgrad3(Wg2 w, int nch) {          // MINW: waves per SIMD the register allocation must allow
    wgrad3_body<BMODE, MPW, NT, TWO_A>(w, nch, blockIdx.x, blockIdx.y, blockIdx.z);
}

#ifndef QPN_WGRAD_MINW
#define QPN_WGRAD_MINW 2
#endif
template <int BMODE, int MPW, int NT, bool TWO_A>
static void launch_wgrad3_k(const Wg2& w, int nch, size_t lds, hipStream_t stream) {
    // the 256 x 64 one-array launches (post-net pair, skip 1x1) are 512 workgroups = two per CU: the allocation may use half a SIMD's registers
    // (round 2 asked for a third of it -- three workgroups per CU; with the staging addresses hoisted that cap spills 80-150 B per lane)
    constexpr int MINW = (MPW == 4 && NT == 4 && !TWO_A) ? QPN_WGRAD_MINW : 1;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_wgrad3<BMODE, MPW, NT, TWO_A, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_wgrad3<BMODE, MPW, NT, TWO_A, MINW>), dim3(nch, w.nlayers, w.ncol_groups), dim3(256), lds, stream, w, nch);
}
template <int BMODE, int MPW, int NT>
static bool wgrad3_fits(const Wg2& w) {
    const int Ng = w.N / w.ncol_groups;
    if (w.bmode != BMODE) return false;
    if (w.rowsB >= (1 << 24) || (int64_t)w.rowsB * (w.C > w.ldb ? w.C : w.ldb) >= (1ll << 32)) return false;      // 24 x 24-bit row x stride products of the fast staging path
    if (w.M != 64 * MPW || Ng != 16 * NT || w.N % w.ncol_groups || (w.Nvalid != w.N && (w.ncol_groups != 1 || w.Nvalid % 4))) return false;
    if (BMODE == 3 && (w.ldb != w.C || w.ncol_groups != 1 || (w.C % 4) || (w.Ap % 4) || !w.tap)) return false;
    if (BMODE == 3) for (int l = 0; l < w.nlayers; ++l) if (w.tap_off[l] < 0) return false;      // the kernel loads taps unconditionally
    return true;
}
template <int MPW, int NT> static constexpr size_t wgrad3_lds() { return (size_t)32 * (tr_ldt(64 * MPW) + tr_ldt(16 * NT)) * sizeof(float); }
template <int BMODE, int MPW, int NT>
static bool launch_wgrad3(const Wg2& w, int nch, hipStream_t stream) {
    if (!wgrad3_fits<BMODE, MPW, NT>(w)) return false;
    const size_t lds = wgrad3_lds<MPW, NT>();
    if (w.A2) launch_wgrad3_k<BMODE, MPW, NT, true>(w, nch, lds, stream);
    else launch_wgrad3_k<BMODE, MPW, NT, false>(w, nch, lds, stream);
    return true;
}
static bool wgrad3_any(const Wg2& w, int nch, bool generic, hipStream_t stream) {
    if (w.bmode == 4) return launch_wgrad3<4, 1, 8>(w, nch, stream) || launch_wgrad3<4, 1, 4>(w, nch, stream);      // causal table: C = 64, 128- or 64-class groups
    if (generic) return false;
    switch (w.bmode) {
    case 3: return launch_wgrad3<3, 2, 11>(w, nch, stream) || launch_wgrad3<3, 2, 8>(w, nch, stream) || launch_wgrad3<3, 1, 7>(w, nch, stream);      // C = 64 (K = 176 / 128: aux at sample / frame rate), C = 32, n_aux 33..48
    case 0: return launch_wgrad3<0, 1, 4>(w, nch, stream) || launch_wgrad3<0, 4, 4>(w, nch, stream);      // res / skip 1x1 at C = 64 (B = the gate product)
    case 1: return launch_wgrad3<1, 4, 4>(w, nch, stream);                                                  // post-net (B rectified), 64-column groups
    default: return false;
    }
}

// smallest number of column groups g (N % g == 0, 4-column granularity kept) such that an M x N/g block fits a kernel's tile budget
static int wgrad_col_groups(int M, int N) {
    const int nt_max = M <= 64 ? 4 : M <= 128 ? 12 : 8;          // <1,4>, <2,12>, <4,8> / <4,4>
    for (int g = 1; g <= N / 16; ++g)
        if (N % g == 0 && (N / g) % 4 == 0 && (N / g + 15) / 16 <= nt_max) return g;
    return 1;
}

static bool wgrad2_any(const Wg2& w, int nch, bool generic, hipStream_t stream) {
    if (w.M > 256 && w.bmode != 4) {
        // more output rows than a workgroup's tile budget (n_skipch / n_quantize > 256): blocks of 256 rows of dW, each a launch of its own -- the A operand's
        // columns, the slab rows and the bias entries shift together
        bool ok = true;
        for (int m0 = 0; m0 < w.M && ok; m0 += 256) {
            Wg2 sub = w;
            sub.A = w.A + m0; if (w.A2) sub.A2 = w.A2 + m0;
            sub.M = w.M - m0 < 256 ? w.M - m0 : 256;
            for (int l = 0; l < w.nlayers; ++l) { sub.goff[l] = w.goff[l] + m0 * w.ldc; if (w.gbias[l] >= 0) sub.gbias[l] = w.gbias[l] + m0; }
            sub.ncol_groups = wgrad_col_groups(sub.M, sub.N);
            ok = wgrad2_any(sub, nch, generic, stream);
        }
        return ok;
    }
    if (wgrad3_any(w, nch, generic, stream)) return true;
    switch (w.bmode) {
    case 4: return false;                       // one-hot operand: k_wgrad3 only (the launcher checks the geometry first)
    case 1: return wgrad2_mode<1>(w, nch, stream);
    case 2: return wgrad2_mode<2>(w, nch, stream);
    case 3: return wgrad2_mode<3>(w, nch, stream);
    default: return wgrad2_mode<0>(w, nch, stream);
    }
}

// ------------------------------------------------------------------------------------------ small backward kernels
// Slabs -> flat gradient, walked in SLAB order: a wave reads 64 consecutive floats of each partial slab (the flat order is a
// permutation of it -- transposed blocks, interleaved taps -- so a gather in flat order touches scattered 4-byte words of 64 slabs),
// and the sum goes to the parameter(s) it feeds; entries nothing feeds are zeroed, the trailer appended.
// [s0, s1): the slab elements this launch reduces (the early range runs on the side stream under the layer backward); `tail`: this launch
// also zeroes the unfed entries and appends the trailer
__global__ void k_reduce_grad_s(const float* __restrict__ slab, const int* __restrict__ gdst, const int* __restrict__ gdst_list, const int* __restrict__ gzero, int n_gzero,
                                int nch, int gstage, int64_t n, float* __restrict__ g, float scale, int append_scale, int s0, int s1, int skip0, int skip1, int tail,
                                const int* __restrict__ status, int flag_again) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + s0;
    // trailer word 1 = "this rank's step is flagged" (the sticky status word, as a float): summed by the gradient exchange, so that EVERY rank's Adam kernel
    // skips the update of a step one rank flagged (k_adam).  flag_again: the backward's LAST launch writes it once more -- the trailer itself may have been appended
    // by the early launch on the side stream, before the stack backward could set its own bit -- unless the trailer has left with an early exchange bucket already
    if (flag_again && append_scale && i == s0) g[n + 1] = (status && *status) ? 1.f : 0.f;
    if (i < s1) {
        if (i >= skip0 && i < skip1) return;                 // (already reduced by the early launch)
        const int k0 = gdst[i], k1 = gdst[i + 1];
        if (k0 == k1) return;                                // padding of the slab layout
        // eight slabs' words requested together (the one-at-a-time sum kept ~4 loads in flight per lane: 29 us for 61 MB)
        float a = 0.f;
        const float* sp = slab + i;
        int c = 0;
        for (; c + 8 <= nch; c += 8) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = sp[(size_t)(c + k) * gstage];
            a += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; c < nch; ++c) a += sp[(size_t)c * gstage];
        a *= scale;
        for (int k = k0; k < k1; ++k) g[gdst_list[k]] = a;
        return;
    }
    if (!tail) return;
    const int64_t j = i - s1;
    if (j < n_gzero) g[gzero[j]] = 0.f;
    else if (append_scale && j < n_gzero + 4 && !(flag_again && j == n_gzero + 1)) g[n + (j - n_gzero)] = j == n_gzero ? scale : (j == n_gzero + 1 && status && *status) ? 1.f : 0.f;
}

// causal conv weight grad: dW[c][q][tap] = sum over rows whose sample == q; LDS table per channel block
__global__ __launch_bounds__(256) void k_causal_bwd(TrainParams p, TrainBwd bw, int rows_per_wg, int CB) {
    extern __shared__ float tab[];                 // [2][Q][CB]; CB: channels per pass (64, fewer when the class count is large: the table stays within 128 KB)
    const int C = p.C, Q = p.Q;
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)p.B * p.N1;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < total ? r0 + rows_per_wg : total;
    for (int cb = 0; cb < C; cb += CB) {
        const int cbv = C - cb < CB ? C - cb : CB;          // channels of this block (the last one is partial when C % 64 != 0)
        for (int i = tid; i < 2 * Q * CB; i += 256) tab[i] = 0.f;
        __syncthreads();
        const int cpr = cbv;                         // channels per row handled by consecutive threads
        for (int64_t idx = r0 * cpr + tid; idx < r1 * cpr; idx += 256) {
            const int64_t rr = idx / cpr; const int c = (int)(idx - rr * cpr);
            const int b = (int)(rr / p.N1), n = (int)(rr - (int64_t)b * p.N1);
            const size_t o = ((size_t)b * p.N1 + n) * C + cb + c;
            const float v = bw.DXA[0][o] + bw.DXB[0][o];   // after the layer-0 backward the result sits in parity 0 (see launcher)
            const int64_t xo = (int64_t)p.T - p.N0 + n;
            int64_t s0 = p.x[(size_t)b * p.T + xo] % Q, s1 = p.x[(size_t)b * p.T + xo + 1] % Q;
            if (s0 < 0) s0 += Q;
            if (s1 < 0) s1 += Q;
            atomicAdd(&tab[(0 * Q + s0) * CB + c], v);
            atomicAdd(&tab[(1 * Q + s1) * CB + c], v);
        }
        __syncthreads();
        // flush: flat index ((c*Q + q)*2 + tap); iterate (q,tap) fastest for contiguous atomics
        for (int i = tid; i < 2 * Q * cbv; i += 256) {
            const int c = i / (2 * Q), qt = i - c * 2 * Q, q = qt >> 1, tp = qt & 1;
            const float v = tab[(tp * Q + q) * CB + c];
            if (v != 0.f) atomicAdd(&bw.gflat[p.causal_w + ((size_t)(cb + c) * Q + q) * 2 + tp], v * bw.gscale);
        }
        // bias grad = sum over all rows = sum over q of the tap-0 table
        for (int c = tid; c < cbv; c += 256) {
            float s = 0.f;
            for (int q = 0; q < Q; ++q) s += tab[(0 * Q + q) * CB + c];
            atomicAdd(&bw.gflat[p.causal_b + cb + c], s * bw.gscale);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(128) void k_up_bwd(UpArgs u) { up_bwd_body(u, blockIdx.x, blockIdx.y, threadIdx.x, 128); }

// Aux hoist: the gradients the frame-rate form leaves to a kernel of its own, from what the layer backward accumulated: D (DPA), G (GW) and
// the gate-bias gradient E_l[n] = colsum(dZ_l) (EB).  With h_up[a][U f + j] = h[a][f] w_up[j] + b_up (reference src/nets/qpnet.py:134-158) and
// z_l[t] += Va_l . h_up[:, t] + ba_l (qpnet.py:215-216, 663-664):
//   role A, blocks [0, L A):        dVa_l[n][a] = sum_{b, f} D_l[b][f][n] h[b][a][f] + b_up E_l[n]          (D_l[b][f][n] = sum_{t in f} w_up[j(t)] dZ_l[t][n])
//                                   db_up      += sum_n Va_l[n][a] E_l[n]                                    (one atomic per block onto the entry the reduction zeroed)
//   role B, blocks [L A, L A + U):  dw_up[j]    = sum_{l, b, t: j(t) = j, t >= s_out(l)} G_l[b][t]          (G_l[b][t] = sum_n dZ_l[t][n] (Va_l . h[b][:, f(t)])[n])
// It needs nothing of the weight-gradient launches: it runs on the side stream next to them.
__global__ __launch_bounds__(128) void k_aux_tail(AuxArgs p) {
    __shared__ float hrow[1024];
    __shared__ float red[2];
    const AuxGeom& ag = p.g;
    const int C = p.C, A = p.A, L = p.L, tid = threadIdx.x;
    const int bx = blockIdx.x;
    float v = 0.f;
    const bool role_a = bx < L * A;
    if (role_a) {
        const int l = bx / A, a = bx - l * A, n = tid;    // (128 threads = 2C gate rows)
        const size_t wi = (n < C ? ag.auxS[l] + (size_t)n * A : ag.auxT[l] + (size_t)(n - C) * A) + a;
        const float va = p.flat[wi];
        float e = 0.f;
        {
            float ev[TR_EB_SLOTS];
#pragma unroll
            for (int k = 0; k < TR_EB_SLOTS; ++k) ev[k] = p.EB[(size_t)(l * TR_EB_SLOTS + k) * 2 * C + n];
#pragma unroll
            for (int k = 0; k < TR_EB_SLOTS; ++k) e += ev[k];
        }
        float acc = 0.f;
        for (int b = 0; b < p.B; ++b) {
            const float* hb = p.h + ((size_t)b * A + a) * p.F + p.ffirst;
            const float* D = p.DPA + (size_t)(l * p.B + b) * (p.nfr + 1) * 2 * C + n;
            for (int f0 = 0; f0 < p.nfr; f0 += 1024) {
                const int nf = p.nfr - f0 < 1024 ? p.nfr - f0 : 1024;
                __syncthreads();
                for (int i = tid; i < nf; i += 128) hrow[i] = hb[f0 + i];
                __syncthreads();
                int f = 0;
                for (; f + 64 <= nf; f += 64) {           // 64 rows requested together (the loop is a chain of memory round trips otherwise)
                    float d[64];
#pragma unroll
                    for (int k = 0; k < 64; ++k) d[k] = D[(size_t)(f0 + f + k) * 2 * C];
#pragma unroll
                    for (int k = 0; k < 64; ++k) acc += d[k] * hrow[f + k];
                }
                for (; f + 16 <= nf; f += 16) {
                    float d[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) d[k] = D[(size_t)(f0 + f + k) * 2 * C];
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc += d[k] * hrow[f + k];
                }
                for (; f < nf; ++f) acc += D[(size_t)(f0 + f) * 2 * C] * hrow[f];
            }
        }
        p.gflat[wi] = (acc + p.flat[p.up_b] * e) * p.gscale;
        v = va * e;
    } else {
        const int j = bx - L * A;
        const int q0 = p.F * p.U - p.N1;                 // h_up sample index of row 0
        const int items = L * p.B * p.nfr;
        for (int it = tid; it < items; it += 128) {
            const int fx = it % p.nfr, lb = it / p.nfr, l = lb / p.B;
            const int n = (p.ffirst + fx) * p.U + j - q0;
            if (n >= ag.s_out[l] && n < p.N1) v += p.GW[(size_t)lb * p.N1 + n];
        }
    }
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        if (role_a) atomicAdd(p.gflat + p.up_b, (red[0] + red[1]) * p.gscale);
        else p.gflat[p.up_w + (bx - L * A)] = (red[0] + red[1]) * p.gscale;
    }
}

// torch.optim.Adam (single tensor semantics, fp32)
// rep: the step's reports, written straight into pinned host memory by the step's last kernel (a copy engine launch each -- 3-4 us on the stream -- otherwise):
// the status word and, when asked for, the 64 partial sums of the loss (qpn_train_step)
struct AdamReports { int* h_status; double* h_loss; const double* d_loss; };
__global__ void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                       float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, const float* __restrict__ den, int* __restrict__ status, AdamReports rep) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // The device-side status word (sticky until the host reads it: a tap / target out of range, an abandoned stack launch) flags results that must not
    // reach the parameters: the update of a flagged step -- and of the steps enqueued behind it until the host has collected the word, two steps later
    // at most -- is skipped; weights and both moments stay what the last clean step left (the host raises; the caller may drop the chunk and go on).
    // Data-parallel (den = the exchanged trailer {summed row count, summed flags}): a step ANY rank flagged is skipped by EVERY rank -- the replicas stay
    // identical -- and a rank that was not flagged itself sets bit 8 of its own word, so that its host raises too.
    const bool peer = den && den[1] > 0.f;
    const bool skip = peer || (status && *status);
    if (blockIdx.x == 0) {
        if (rep.h_loss && threadIdx.x < 64) rep.h_loss[threadIdx.x] = rep.d_loss[threadIdx.x];
        if (threadIdx.x == 0 && status) {
            int st = *status;
            if (peer && !st) { st = 8; atomicOr(status, 8); }
            if (rep.h_status) *rep.h_status = st;
            if (!skip) atomicAdd((unsigned long long*)(status + 2), 1ull);      // updates APPLIED on this handle (qpn_train_applied_updates)
        }
    }
    if (i >= n || skip) return;
    float gi = g[i];
    if (den) gi = gi / den[0];                                  // data-parallel: summed row-weighted gradients / summed row count
    if (wd != 0.f) gi += wd * w[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = w[i] - (lr / bc1) * (mi / denom);
}

// Zero exactly what the backward reads without having written it.  Grad wrt X[j] has two parts:
//   DXA[j] (own-row part): layer j writes rows [s_out(j), N1); layer j-1 (or the causal backward for j = 0) reads from
//           row s_in(j) (0 for j = 0) -> rows [s_in(j) or 0, s_out(j)) must read as zero;
//   DXB[j] (pitch-tap scatter part): a fixed layer is the unique writer of rows [s_in(j), N1 - dilation), so only the last
//           `dilation` rows (and, for j = 0, nothing in front) need zeros; an adaptive layer adds with atomics -> all rows.
// X[L]'s gradient is never read (the last block's residual output is unused, qpnet.py:306-309).
// part / nparts: the slice of (layer j, batch item b)'s zeroing this caller does, with nthr threads of which this is thread t
__device__ __forceinline__ void zero_dx_slice(const TrainParams& p, const TrainBwd& bw, int j, int b, size_t part, size_t nparts, int t, int nthr) {
    const int C = p.C;
    const TrLayer ly = p.layers[j];
    const size_t nDX = (size_t)p.B * p.N1 * C;
    float* A = bw.DXA[0] + (size_t)j * nDX + (size_t)b * p.N1 * C;
    float* Bq = bw.DXB[0] + (size_t)j * nDX + (size_t)b * p.N1 * C;
    const int a0 = j == 0 ? 0 : ly.s_in, a1 = ly.s_out;
    const int b0 = ly.adaptive ? 0 : p.N1 - ly.dilation, b1 = p.N1;
    const size_t na = (size_t)(a1 - a0) * C / 4, nb = (size_t)(b1 > b0 ? b1 - b0 : 0) * C / 4;      // C % 16 == 0
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    // layer 0's blocks also clear the aux-feature gradient [N1][Ap] of this batch item (every layer adds to it); aux hoist: every layer's
    // blocks clear the layer's frame-rate accumulators D_l[b] instead
    const size_t nh = p.hoist ? (size_t)(p.nfr + 1) * 2 * C / 4 : (j == 0 ? (size_t)p.N1 * p.Ap / 4 : 0);
    const size_t ne = (p.hoist && b == 0) ? (size_t)TR_EB_SLOTS * 2 * C / 4 : 0;            // ... and (batch item 0's blocks) the layer's gate-bias accumulators E_l
    float* H = p.hoist ? bw.DPA + (size_t)(j * p.B + b) * (p.nfr + 1) * 2 * C : bw.DHUP + (size_t)b * p.N1 * p.Ap;
    for (size_t i = part * nthr + t; i < na + nb + nh + ne; i += nparts * nthr) {
        if (i < na) ((float4*)(A + (size_t)a0 * C))[i] = z;
        else if (i < na + nb) ((float4*)(Bq + (size_t)(b0 > 0 ? b0 : 0) * C))[i - na] = z;
        else if (i < na + nb + nh) ((float4*)H)[i - na - nb] = z;
        else ((float4*)(bw.EB + (size_t)j * TR_EB_SLOTS * 2 * C))[i - na - nb - nh] = z;
    }
}
__global__ void k_zero_dx(TrainParams p, TrainBwd bw) {
    zero_dx_slice(p, bw, blockIdx.y, blockIdx.z, blockIdx.x, gridDim.x, threadIdx.x, blockDim.x);
}

// ------------------------------------------------------------------------------------------ launchers
void qpn_launch_zero_dx(const TrainParams& p, const TrainBwd& bw, hipStream_t stream) {
    hipLaunchKernelGGL(k_zero_dx, dim3(64, p.L, p.B), dim3(256), 0, stream, p, bw);
}

// slabs -> flat gradient (writes every entry), then the histogram-style gradients on top (causal table when no
// contraction owns it, upsampling kernel)
// the skip / post-net part of the slab reduction, launched on the side stream right behind those weight gradients
// tail: this launch also zeroes the entries no slab feeds and appends the trailer (then the final reduction must not: the upsampling
// kernel's gradient is added onto those zeros in between, see qpn_launch_bwd)
static void launch_reduce_early(const TrainBwd& bw, hipStream_t st, int tail, int nch_side) {
    const TrainSlabs& sl = *bw.sl;
    const int64_t n = (int64_t)(sl.g_early1 - sl.g_early0) + (tail ? bw.n_gzero + 4 : 0);
    hipLaunchKernelGGL(k_reduce_grad_s, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bw.slab, bw.gdst, bw.gdst_list, bw.gzero, bw.n_gzero,
                       nch_side, bw.gstage, bw.n_params, bw.gflat, bw.gscale, bw.append_scale, sl.g_early0, sl.g_early1, 0, 0, tail, bw.status, 0);
}
static void launch_up_bwd(const TrainParams& p, const TrainBwd& bw, hipStream_t st) {
    const UpArgs u = up_args(p, bw);
    hipLaunchKernelGGL(k_up_bwd, dim3(u.nfr, p.B), dim3(128), 0, st, u);
}

// early_done: the [g_early0, g_early1) slab range has been reduced already; up_done: so have the zeroing / trailer and, on top of the zeros, the
// upsampling kernel's gradient
static void launch_aux_tail(const TrainParams& p, const TrainBwd& bw, const AuxGeom& ag, hipStream_t st) {
    hipLaunchKernelGGL(k_aux_tail, dim3(p.L * p.A + p.U), dim3(128), 0, st, tr_aux_args(p, &bw, ag));
}

int qpn_launch_grad_tail(const TrainParams& p, const TrainBwd& bw, const AuxGeom* ag, hipStream_t stream, bool early_done, bool up_done) {
    const int C = p.C, Q = p.Q, B = p.B, N1 = p.N1;
    const TrainSlabs& sl = *bw.sl;
    hipLaunchKernelGGL(k_reduce_grad_s, dim3((unsigned)(((int64_t)bw.gstage + (up_done ? 0 : bw.n_gzero + 4) + 255) / 256)), dim3(256), 0, stream, bw.slab, bw.gdst, bw.gdst_list, bw.gzero, bw.n_gzero,
                       bw.nch, bw.gstage, bw.n_params, bw.gflat, bw.gscale, bw.append_scale, 0, bw.gstage, early_done ? sl.g_early0 : 0, early_done ? sl.g_early1 : 0, up_done ? 0 : 1, bw.status, bw.append_scale == 1 ? 1 : 0);
    {
        const int64_t total = (int64_t)B * N1;
        const int nwg = 128, rpw = (int)((total + nwg - 1) / nwg);
        if (sl.g_cw < 0) {
            int CB = C < 64 ? C : 64;
            while (CB > 1 && (size_t)2 * Q * CB * sizeof(float) > 128 * 1024) CB /= 2;
            const size_t lds_c = (size_t)2 * Q * CB * sizeof(float);
            if (lds_c > 160 * 1024) { qpn_set_error("causal-table gradient: n_quantize too large for the LDS histogram"); return QPN_EINVAL; }
            if (lds_c > 48 * 1024) QPN_HIP(hipFuncSetAttribute((const void*)k_causal_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
            hipLaunchKernelGGL(k_causal_bwd, dim3(nwg), dim3(256), lds_c, stream, p, bw, rpw, CB);
        }
        if (p.U > 0 && !up_done && !p.hoist) launch_up_bwd(p, bw, stream);
        if (p.hoist && ag && !up_done) launch_aux_tail(p, bw, *ag, stream);      // (behind the reduction's zeroing: it adds onto the upsampling-bias entry; up_done: it ran on the side stream)
    }
    qpn_prof_mark(PG_GRAD_TAIL, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}

bool qpn_stack_bwd_fits(const TrainParams& p);
int qpn_launch_stack_bwd(const TrainParams& p, const TrainBwd& bw, const StackQ& q, const TrainKnobs& k, hipStream_t stream);
bool qpn_stack_bwd_w_fits(const TrainParams& p);
int qpn_launch_stack_bwd_w(const TrainParams& p, const TrainBwd& bw, const StackQ& q, const TrainKnobs& k, hipStream_t stream);

void qpn_launch_post_fb(const TrainParams& p, const TrainBwd& bw, hipStream_t stream) {
    constexpr int MTW = 5;
    const size_t ldsw = (size_t)16 * MTW * (tr_lda(256) + 2 * tr_lda(64)) * sizeof(float);
    (void)hipFuncSetAttribute((const void*)k_post_fb_w<MTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
    hipLaunchKernelGGL((k_post_fb_w<MTW>), dim3((p.BL + 16 * MTW - 1) / (16 * MTW), p.B), dim3(512), ldsw, stream, p, bw, 1);
}

// post_done: the forward's launch sequence has run the post-net's backward (and the zeroing) already
int qpn_launch_bwd(const TrainParams& p, const TrainBwd& bw, const TrainKnobs& k, const AuxGeom& ag, const StackQ* sq, hipStream_t stream, bool post_done) {
    const int C = p.C, S = p.S, Q = p.Q, L = p.L, B = p.B, N1 = p.N1, BL = p.BL;
    const TrainSlabs& sl = *bw.sl;
    const bool off32 = N1 < (1 << 24) && (int64_t)N1 * (p.LC > 2 * C ? p.LC : 2 * C) < (1ll << 32);       // k_layer_bwd_p's 32-bit element offsets (one batch item)
    const bool persist = C == 64 && (p.Ktp == 176 || p.hoist) && p.Ap <= 64 && off32 && k.persist_bwd;
    // the whole stack's backward as ONE persistent launch over a (layer, tile) work queue (train_stack.hip); QPN_STACK_QUEUE_BWD=0 (or
    // QPN_STACK_QUEUE=0) keeps a launch per layer
    const bool stack_q = persist && k.stack_q_bwd && sq && sq->flags && p.qctl && qpn_stack_bwd_fits(p);
    const bool wr_summed = stack_q;
    const size_t nDX = (size_t)B * N1 * C;
    // 16-row tiles (measured 11-15 % faster than 32 rows: twice the workgroups, a shorter last round)
    const size_t lds_post = (size_t)16 * (tr_lda(Q > S ? Q : S) + tr_lda(S)) * sizeof(float);
    const size_t lds_layer = (size_t)16 * (4 * tr_lda(C) + tr_lda(2 * C)) * sizeof(float);
    if (lds_post > 160 * 1024 || lds_layer > 160 * 1024) { qpn_set_error("backward tiles do not fit LDS"); return QPN_EINVAL; }
    // grads wrt layer outputs: DXA/DXB[l] for l = 0..L (index l = grad wrt X[l]); zero (scatter targets / unwritten rows)
    // only the rows a consumer reads but no producer writes (see k_zero_dx): 4 adaptive scatter targets instead of 2(L+1) full arrays
    hipStream_t side = bw.side; hipEvent_t ev_fork = bw.ev_fork, ev_join = bw.ev_join;   // created with the handle's TrainState, on its device
    const bool overlap = side && !qpn_prof_serial() && !k.serial;
    const bool post_wide = S == 256 && Q == 256 && C == 64 && (p.LC == 256 || p.LC == 512) && k.post_wide;
    const bool zero_in_post = post_wide && k.zero_in_post;
    if (post_done) { }
    else if (!zero_in_post) qpn_launch_zero_dx(p, bw, stream);
    if (post_done) { }
    else if (post_wide) {      // 80 rows per workgroup (k_post_bwd_w)
        constexpr int MTW = 5;
        const size_t ldsw = (size_t)16 * MTW * tr_lda(256) * sizeof(float);
        QPN_HIP(hipFuncSetAttribute((const void*)k_post_bwd_w<MTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
        hipLaunchKernelGGL((k_post_bwd_w<MTW>), dim3((BL + 16 * MTW - 1) / (16 * MTW), B), dim3(512), ldsw, stream, p, bw, zero_in_post ? 1 : 0);
    } else {
        if (lds_post > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_post_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_post);
        hipLaunchKernelGGL((k_post_bwd<1>), dim3((BL + 15) / 16, B), dim3(512), lds_post, stream, p, bw);
    }
    qpn_prof_mark(PG_POST_BWD, stream);
    // ---- the skip 1x1 and post-net weight gradients only need what k_post_bwd has just produced (dS0, dY0, dlogits) and
    // saved activations: they run on a side stream UNDER the layer backward (the layer kernels are latency-bound but their
    // workgroups still occupy every CU, so the overlap is partial).  Per-group profiling (bench roofline) and QPN_TRAIN_SERIAL=1
    // keep everything on the one stream.
    const int nch = bw.nch;
    int nch_side_ = nch;      // time chunks (= partial slabs) of the skip / post-net contractions, which run on the side stream under the layer backward and are reduced by a launch of their own
    const bool gen = k.wgrad_generic;
    Wg2 wbase; memset(&wbase, 0, sizeof(wbase));
    wbase.slab = bw.slab; wbase.gstage = bw.gstage; wbase.nb = B; wbase.C = C; wbase.Ap = p.Ap; wbase.hup = p.HUP; wbase.tap = p.TAP; wbase.ncol_groups = 1;
    bool ok = true;
    auto launch_skip_post = [&](hipStream_t st) {
        Wg2 w = wbase;
        {   // dWs_l = dS0^T g_l over the last BL rows; the shared skip-bias grad = colsum(dS0) (layer 0's block only)
            w.nlayers = L;
            w.A = bw.DS0; w.A2 = nullptr; w.A_lstride = 0; w.lda = S; w.M = S; w.rowsA = BL;
            w.bmode = 0; w.B1 = p.TH; w.B2 = nullptr; w.B_lstride = nDX; w.ldb = C; w.N = C; w.Nvalid = C; w.rowsB = N1; w.ldc = C;      // (p.TH: the gate product, as it is)
            w.ncol_groups = wgrad_col_groups(w.M, w.N);
            for (int l = 0; l < L; ++l) { w.row0A[l] = 0; w.row0B[l] = N1 - BL; w.R[l] = BL; w.goff[l] = sl.g_ws[l]; w.gbias[l] = l == 0 ? sl.g_bs : -1; w.tap_off[l] = -1; }
            ok = ok && wgrad2_any(w, nch_side_, gen, st);
            qpn_prof_mark(PG_WGRAD_SKIP, st);
        }
        {   // post-net: dW2[q][s] = dlogits^T relu(Y0), dW1[o][s] = dY0^T relu(S0); N split into 64-column groups
            w.nlayers = 1; w.A_lstride = w.B_lstride = 0; w.B2 = nullptr; w.bmode = 1; w.rowsA = w.rowsB = BL;
            w.row0A[0] = w.row0B[0] = 0; w.R[0] = BL; w.tap_off[0] = -1;
            w.ncol_groups = S % 64 == 0 ? S / 64 : 1;
            w.A = bw.dlogits; w.lda = Q; w.M = Q; w.B1 = p.Y0; w.ldb = S; w.N = S; w.Nvalid = S; w.ldc = S; w.goff[0] = sl.g_p2; w.gbias[0] = sl.g_bp2;
            if (Q == S && k.post_pair) {
                // both contractions have the same shape: ONE launch with the second as "layer 1" (strides = the distance between the arrays,
                // modulo 2^64), 2 x 256 workgroups = two per CU, so one's staging runs under the other's MFMAs (alone, each launch put one
                // workgroup on a CU: 35 % of every stage with the matrix cores idle, in-kernel stamps)
                w.nlayers = 2;
                w.A_lstride = (size_t)(bw.DY0 - bw.dlogits); w.B_lstride = (size_t)(p.S0 - p.Y0);
                w.row0A[1] = w.row0B[1] = 0; w.R[1] = BL; w.tap_off[1] = -1; w.goff[1] = sl.g_p1; w.gbias[1] = sl.g_bp1;
                ok = ok && wgrad2_any(w, nch_side_, gen, st);
            } else {
                ok = ok && wgrad2_any(w, nch_side_, gen, st);
                w.A = bw.DY0; w.lda = S; w.M = S; w.B1 = p.S0; w.goff[0] = sl.g_p1; w.gbias[0] = sl.g_bp1;
                ok = ok && wgrad2_any(w, nch_side_, gen, st);
            }
            qpn_prof_mark(PG_WGRAD_POST, st);
        }
    };
    // ---- weight gradients that need the layer backward: dW1 (needs dZ_l), the residual 1x1 (needs dX_{l+1}), the causal table
    auto build_w1 = [&]() {     // dW1_l = dZ_l^T [x_cur | x_past | aux],  bias1 grads = colsum(dZ_l)
        Wg2 w = wbase;
        w.A = bw.DZ; w.A2 = nullptr; w.A_lstride = (size_t)B * N1 * 2 * C; w.lda = 2 * C; w.M = 2 * C; w.rowsA = N1;
        w.bmode = 3; w.B1 = p.X; w.B2 = nullptr; w.B_lstride = nDX; w.ldb = C; w.N = p.Ktp; w.Nvalid = p.hoist ? 2 * C : 2 * C + p.Ap; w.rowsB = N1;      // (aux hoist: no aux columns -- k_aux_tail)
        w.nlayers = L; w.ldc = p.Ktp; w.ncol_groups = wgrad_col_groups(w.M, w.N);
        for (int l = 0; l < L; ++l) {
            const TrLayer& ly = p.layers[l];
            w.row0A[l] = w.row0B[l] = ly.s_out; w.R[l] = N1 - ly.s_out; w.goff[l] = sl.g_w1[l]; w.gbias[l] = sl.g_b1[l];
            w.tap_off[l] = ly.tap_off; w.dil[l] = ly.dilation;
        }
        return w;
    };
    auto build_wr = [&]() {     // dWr_l = dXout_l^T g_l (dXout_l = grad wrt X[l+1]); zero rows for the last layer
        Wg2 w = wbase;
        // (the stack queue's tiles have replaced the own-row part by the sum of both parts: store_dx in k_stack_bwd)
        w.A = bw.DXA[0] + nDX; w.A2 = wr_summed ? nullptr : bw.DXB[0] + nDX; w.A_lstride = nDX; w.lda = C; w.M = C; w.rowsA = N1;
        w.bmode = 0; w.B1 = p.TH; w.B2 = nullptr; w.B_lstride = nDX; w.ldb = C; w.N = C; w.Nvalid = C; w.rowsB = N1; w.ldc = C;      // (p.TH: the gate product, as it is)
        w.nlayers = L; w.ncol_groups = wgrad_col_groups(w.M, w.N);
        for (int l = 0; l < L; ++l) {
            w.row0A[l] = w.row0B[l] = p.layers[l].s_out;
            w.R[l] = l == L - 1 ? 0 : N1 - p.layers[l].s_out; w.goff[l] = sl.g_wr[l]; w.gbias[l] = sl.g_br[l]; w.tap_off[l] = -1; w.dil[l] = 0;
        }
        return w;
    };
    // the memory-bound reduction of the skip / post-net slabs (half of k_reduce_grad's 130 MB) runs on the side stream under the
    // matrix-bound layer backward instead of at the end of the step
    const bool early_reduce = overlap && bw.gdst && sl.g_early1 > sl.g_early0 && k.reduce_early;
    // with the early reduction doing the zeroing / trailer too, the upsampling kernel's gradient (atomics onto those zeros, needs dH of every
    // layer) runs on the side stream next to dW1 instead of behind the final reduction
    // (aux hoist: k_aux_tail takes k_up_bwd's place: it needs the layer backward's D / G / E and the zeroed upsampling-bias entry)
    const bool up_side = early_reduce && p.U > 0 && k.up_side;
    // (only with the early reduction: otherwise ONE launch reduces every block with one slab count)
    if (early_reduce) {
        // measured on the overlapped step: 64 -> 48 chunks makes each of these launches slower alone and the step faster
        nch_side_ = k.wgrad_chunks_side > 0 && k.wgrad_chunks_side <= nch ? k.wgrad_chunks_side : (nch <= 48 ? nch : 48);
    }
    if (overlap) {
        QPN_HIP(hipEventRecord(ev_fork, stream));
        QPN_HIP(hipStreamWaitEvent(side, ev_fork, 0));
        qpn_prof_mark(-1, side);
        launch_skip_post(side);
        if (early_reduce) { launch_reduce_early(bw, side, up_side ? 1 : 0, nch_side_); qpn_prof_mark(PG_GRAD_TAIL, side); }
        // the post-net block of the flat gradient (and, with up_side, the zeroing and the row-count trailer behind it) is final from here on:
        // a data-parallel caller exchanges that bucket while the layer backward still runs (qpn_train_early_bucket)
        if (early_reduce && up_side && bw.ev_early && bw.early_recorded) { QPN_HIP(hipEventRecord(bw.ev_early, side)); *bw.early_recorded = 1; }
    }
    const int swz = k.xcd_swizzle ? 1 : 0;
    if (p.hoist && !persist) { qpn_set_error("internal: the frame-rate aux term needs the register-resident layer kernels"); return QPN_EINVAL; }
    if (stack_q) {      // one wave per tile where that form exists (train_stackw.hip), else one workgroup per tile
        const int rcq = (k.stack_wave_bwd && qpn_stack_bwd_w_fits(p)) ? qpn_launch_stack_bwd_w(p, bw, *sq, k, stream) : qpn_launch_stack_bwd(p, bw, *sq, k, stream);
        if (rcq) return rcq;
    }
    for (int l = L - 1; l >= 0 && !stack_q; --l) {
        const TrLayer& ly = p.layers[l];
        const int rows = N1 - ly.s_out;
        const int tiles = (rows + 15) / 16;
        if (persist) {      // register-resident weights, 2 workgroups per CU over contiguous tile ranges (k_layer_bwd_p)
            int G = qpn_num_cus() * 2; if (G > tiles) G = tiles;
            if (G > 1024) G = 1024;                                // (scratch_rows holds a pair of rows for 1024 workgroups per batch item)
            const size_t ldsp = (size_t)(8 * 16 * tr_lda(C) + 16 * tr_lda(2 * C) + 16 * tr_lda(p.Ktp) + 64) * sizeof(float);
            const bool last = l == L - 1;
            const void* kf = p.hoist ? (last ? (const void*)k_layer_bwd_p<8, true> : (const void*)k_layer_bwd_p<8, false>)
                                     : (last ? (const void*)k_layer_bwd_p<11, true> : (const void*)k_layer_bwd_p<11, false>);
            (void)hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
            if (p.hoist) {
                if (last) hipLaunchKernelGGL((k_layer_bwd_p<8, true>), dim3(G, B), dim3(256), ldsp, stream, p, bw, l, swz ? 2 : 0, tiles, p.scratch_rows);
                else hipLaunchKernelGGL((k_layer_bwd_p<8, false>), dim3(G, B), dim3(256), ldsp, stream, p, bw, l, swz ? 2 : 0, tiles, p.scratch_rows);
            } else {
                if (last) hipLaunchKernelGGL((k_layer_bwd_p<11, true>), dim3(G, B), dim3(256), ldsp, stream, p, bw, l, swz ? 2 : 0, tiles, p.scratch_rows);
                else hipLaunchKernelGGL((k_layer_bwd_p<11, false>), dim3(G, B), dim3(256), ldsp, stream, p, bw, l, swz ? 2 : 0, tiles, p.scratch_rows);
            }
        } else {
            if (lds_layer > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_layer_bwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_layer);
            hipLaunchKernelGGL((k_layer_bwd<1>), dim3(tiles, B), dim3(256), lds_layer, stream, p, bw, l, l == L - 1 ? 1 : 0, swz);
        }
    }
    // the residual-1x1 weight gradient is a memory-bound 64 x 64 contraction: on the side stream next to the matrix-heavy dW1 launch
    // (measured 1028 -> 1052 steps/s; the causal table's contraction there as well: 1028 again, the side chain becomes the longer one)
    auto build_causal = [&]() {      // causal conv table: dWc[tap][c][q] = (dX0)^T onehot(class of x[t-1+tap]); bias = colsum(dX0)
        Wg2 w = wbase;
        w.nlayers = 2; w.ncol_groups = Q / (Q % 64 == 0 ? 64 : 128);     /* 64-class groups: 512 workgroups, two per CU (29 -> 25 us) */ w.bmode = 4; w.xc = p.XC; w.B1 = w.B2 = nullptr; w.B_lstride = 0;
        w.A = bw.DXA[0]; w.A2 = bw.DXB[0]; w.A_lstride = 0; w.lda = C; w.M = C; w.rowsA = N1;
        w.N = Q; w.Nvalid = Q; w.rowsB = N1 + 1; w.ldb = 0; w.ldc = Q;
        for (int tp = 0; tp < 2; ++tp) { w.row0A[tp] = 0; w.row0B[tp] = tp; w.R[tp] = N1; w.goff[tp] = sl.g_cw + tp * C * Q; w.gbias[tp] = tp == 0 ? sl.g_cb : -1; w.tap_off[tp] = -1; w.dil[tp] = 0; }
        return w;
    };
    // [One launch for everything that waits for the whole layer backward (dW1 + dWr + causal table + upsampling kernel as roles of one kernel)
    //  was tried: 1146 steps/s against 1205.  dW1 is a 247-register kernel (159 VGPRs + 88 accumulators): two of its workgroups fill a CU's
    //  register file, nothing runs beside them -- which is also why kernels launched next to it on the side stream start when it ends.]
    const bool wr_side = overlap && k.wr_side;
    if (wr_side || up_side) {
        QPN_HIP(hipEventRecord(bw.ev_mid, stream));
        QPN_HIP(hipStreamWaitEvent(side, bw.ev_mid, 0));
        qpn_prof_mark(-1, side);
        if (up_side) { if (p.hoist) launch_aux_tail(p, bw, ag, side); else launch_up_bwd(p, bw, side); qpn_prof_mark(PG_GRAD_TAIL, side); }      // (behind the early reduction's zeroing, on the same stream)
        if (wr_side) { ok = ok && wgrad2_any(build_wr(), nch, gen, side); qpn_prof_mark(PG_WGRAD_WR, side); }
    }
    if (overlap) QPN_HIP(hipEventRecord(ev_join, side));
    qpn_prof_mark(PG_LAYER_BWD, stream);
    ok = ok && wgrad2_any(build_w1(), nch, gen, stream);
    qpn_prof_mark(PG_WGRAD, stream);
    if (!wr_side) { ok = ok && wgrad2_any(build_wr(), nch, gen, stream); qpn_prof_mark(PG_WGRAD_WR, stream); }
    if (!overlap) launch_skip_post(stream);
    if (sl.g_cw >= 0) ok = ok && wgrad2_any(build_causal(), nch, gen, stream);
    qpn_prof_mark(PG_WGRAD_CAUSAL, stream);
    if (overlap) { QPN_HIP(hipStreamWaitEvent(stream, ev_join, 0)); qpn_prof_mark(-1, stream); }      // joined BEFORE any early return: the caller's stream must own everything enqueued here
    if (!ok) { qpn_set_error("weight-gradient tiles: unsupported geometry (n_resch <= 128 on this path; wider stacks take the GEMM path)"); return QPN_EINVAL; }
    return qpn_launch_grad_tail(p, bw, &ag, stream, early_reduce, up_side);
}

int qpn_launch_adam(float* w, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2, float eps, float wd, const float* den, int* status,
                    int* h_status, double* h_loss, const double* d_loss, hipStream_t stream) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    AdamReports rep; rep.h_status = status ? h_status : nullptr; rep.h_loss = d_loss ? h_loss : nullptr; rep.d_loss = d_loss;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2), den, status, rep);
    qpn_prof_mark(PG_ADAM, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
