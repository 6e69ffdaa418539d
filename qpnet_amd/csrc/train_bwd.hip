// train_bwd.hip -- hand-written backward of QPNet.forward on gfx950 (what torch autograd does for the
// reference at src/bin/qpnet_train.py:529-530), fp32 MFMA.
//
//   k_post_bwd   : dlogits -> dY0 -> dS0 (skip-sum grad) -> per-layer gate grads DGS = dS0 . Ws_l
//   k_layer_bwd  : dXout, DGS -> dg -> (dzs, dzt) -> d[x_cur | x_past | aux] = dZ . W1
//                  x_cur part (+ residual) is stored at its own row; the x_past part is the backward of
//                  the pitch-dependent gather = scatter-add to row tap[n] (float atomics only for the
//                  adaptive blocks; fixed blocks have a unique writer per row)
//   k_wgrad      : every weight gradient is a time-contraction dW[m][n] = sum_t A[t][m] B[t][n]; split over
//                  `nch` time chunks into partial slabs (deterministic), bias grads = column sums of A
//   k_causal_bwd : the one-hot causal conv's weight grad is a histogram over sample classes -> LDS table
//   k_up_bwd     : gradient of the (1,1,1,U) upsampling kernel and its bias
//   k_reduce_grad: slabs -> flat gradient in state_dict order;  k_adam: torch.optim.Adam update.
#include "train_common.h"
#include "qpn_handle.h"
#include <string.h>
#include <math.h>

// ------------------------------------------------------------------------------------------ post-net backward
// dynamic LDS: P[64][lda(max(Q,S))] | R[64][lda(S)]
__global__ __launch_bounds__(512) void k_post_bwd(TrainParams p, TrainBwd bw) {
    extern __shared__ float sm[];
    const int S = p.S, Q = p.Q, LC = p.LC;
    const int ldp = tr_lda(Q > S ? Q : S), ldr = tr_lda(S);
    float* P = sm; float* R = sm + 64 * ldp;
    const int b = blockIdx.y, t0 = blockIdx.x * TR_TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTS = S / 16;
    // stage dlogits rows
    for (int idx = tid; idx < TR_TM * (Q / 2); idx += 512) {
        const int r = idx / (Q / 2), k = (idx - r * (Q / 2)) * 2;
        float2 v = make_float2(0.f, 0.f);
        if (t0 + r < p.BL) v = *(const float2*)(bw.dlogits + ((size_t)b * p.BL + t0 + r) * Q + k);
        *(float2*)(P + (size_t)r * ldp + k) = v;
    }
    __syncthreads();
    // dY0 = (dlogits . W2) * (Y0 > 0)
    const int npairs = (NTS + 1) / 2;
    for (int pb = 0; pb < npairs; pb += 8) {
        const int np = pb + wave;
        if (np < npairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
            f32x4 acc[4][2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<4, 2>(acc, P, ldp, p.wp + p.p2t_f4, NTS, nts, Q, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j && nt1 == nt0) break;
                const int c = 16 * (j ? nt1 : nt0) + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        float v = 0.f;
                        if (t0 + r < p.BL) {
                            const size_t o = ((size_t)b * p.BL + t0 + r) * S + c;
                            v = p.Y0[o] > 0.f ? acc[mt][j][i] : 0.f;
                            bw.DY0[o] = v;
                        }
                        R[(size_t)r * ldr + c] = v;
                    }
            }
        }
    }
    __syncthreads();
    // dS0 = (dY0 . W1post) * (S0 > 0)   -> P (dlogits are dead)
    for (int pb = 0; pb < npairs; pb += 8) {
        const int np = pb + wave;
        if (np < npairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
            f32x4 acc[4][2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<4, 2>(acc, R, ldr, p.wp + p.p1t_f4, NTS, nts, S, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j && nt1 == nt0) break;
                const int c = 16 * (j ? nt1 : nt0) + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        float v = 0.f;
                        if (t0 + r < p.BL) {
                            const size_t o = ((size_t)b * p.BL + t0 + r) * S + c;
                            v = p.S0[o] > 0.f ? acc[mt][j][i] : 0.f;
                            bw.DS0[o] = v;
                        }
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
            f32x4 acc[4][2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<4, 2>(acc, P, ldp, p.wp + p.wst_f4, NTL, nts, S, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j && nt1 == nt0) break;
                const int c = 16 * (j ? nt1 : nt0) + (lane & 15);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        if (t0 + r < p.BL) bw.DGS[((size_t)b * p.BL + t0 + r) * LC + c] = acc[mt][j][i];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ layer backward
// dynamic LDS: Dx[64][lda(C)] | Dz[64][lda(2C)]
// DXin = grads w.r.t. this layer's OUTPUT (A: own-row part, B: scattered part, both zero-filled where unwritten);
// DXout = grads w.r.t. this layer's INPUT (same two-part form), consumed by layer l-1 / the causal backward.
__global__ __launch_bounds__(256) void k_layer_bwd(TrainParams p, TrainBwd bw, int l, int last, int pp) {
    extern __shared__ float sm[];
    const TrLayer ly = p.layers[l];
    const int C = p.C, Ktp = p.Ktp, Ap = p.Ap;
    const int ldx = tr_lda(C), ldz = tr_lda(2 * C);
    float* Dx = sm; float* Dz = sm + 64 * ldx;
    const int b = blockIdx.y, n0 = ly.s_out + blockIdx.x * TR_TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t rb = (size_t)b * p.N1;
    const float* DAin = bw.DXA[pp] + rb * C; const float* DBin = bw.DXB[pp] + rb * C;
    float* DAout = bw.DXA[pp ^ 1] + rb * C; float* DBout = bw.DXB[pp ^ 1] + rb * C;
    const int NCG = C / 16;
    // ---- dXout tile
    for (int idx = tid; idx < TR_TM * (C / 2); idx += 256) {
        const int r = idx / (C / 2), k = (idx - r * (C / 2)) * 2, n = n0 + r;
        float2 v = make_float2(0.f, 0.f);
        if (!last && n < p.N1) {
            const float2 a = *(const float2*)(DAin + (size_t)n * C + k), bb = *(const float2*)(DBin + (size_t)n * C + k);
            v = make_float2(a.x + bb.x, a.y + bb.y);
        }
        *(float2*)(Dx + (size_t)r * ldx + k) = v;
    }
    __syncthreads();
    // ---- dg = dXout . Wr + DGS ; dz = dg * gate'
    const float* SG = p.SG + ((size_t)(l * p.B + b) * p.N1) * C;
    const float* TH = p.TH + ((size_t)(l * p.B + b) * p.N1) * C;
    float* DZg = bw.DZ + rb * 2 * C;
    const int win0 = p.N1 - p.BL;
    for (int nt = wave; nt < NCG; nt += 4) {
        f32x4 acc[4][1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][0] = (f32x4){0, 0, 0, 0};
        if (!last) { const int nts[1] = {nt}; wave_gemm<4, 1>(acc, Dx, ldx, p.wp + ly.wrt_f4, NCG, nts, C, lane); }
        const int c = 16 * nt + (lane & 15);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i, n = n0 + r;
                float dzs = 0.f, dzt = 0.f;
                if (n < p.N1) {
                    float dg = acc[mt][0][i];
                    if (n >= win0) dg += bw.DGS[((size_t)b * p.BL + (n - win0)) * p.LC + (size_t)l * C + c];
                    const float sg = SG[(size_t)n * C + c], th = TH[(size_t)n * C + c];
                    dzs = dg * th * sg * (1.0f - sg);
                    dzt = dg * sg * (1.0f - th * th);
                    DZg[(size_t)n * 2 * C + c] = dzs; DZg[(size_t)n * 2 * C + C + c] = dzt;
                }
                Dz[(size_t)r * ldz + c] = dzs; Dz[(size_t)r * ldz + C + c] = dzt;
            }
    }
    __syncthreads();
    // ---- d[x_cur | x_past | aux] = dZ . W1
    const int NTK = Ktp / 16;
    const int* taps = ly.adaptive ? p.TAP + ly.tap_off + (size_t)b * p.N1 : nullptr;
    float* DH = bw.DHUP + rb * Ap;
    for (int nt = wave; nt < NTK; nt += 4) {
        f32x4 acc[4][1];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt][0] = (f32x4){0, 0, 0, 0};
        const int nts[1] = {nt};
        wave_gemm<4, 1>(acc, Dz, ldz, p.wp + ly.w1t_f4, NTK, nts, 2 * C, lane);
        const int k = 16 * nt + (lane & 15);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i, n = n0 + r;
                if (n >= p.N1) continue;
                const float v = acc[mt][0][i];
                if (k < C) DAout[(size_t)n * C + k] = v + Dx[(size_t)r * ldx + k];              // + residual path
                else if (k < 2 * C) {
                    if (taps) atomicAdd(&DBout[(size_t)taps[n] * C + (k - C)], v);               // gather backward (collisions)
                    else DBout[(size_t)(n - ly.dilation) * C + (k - C)] = v;                      // unique writer
                } else if (k < 2 * C + Ap) DH[(size_t)n * Ap + (k - 2 * C)] += v;
            }
    }
}

// ------------------------------------------------------------------------------------------ weight gradients
struct WgDesc {
    // A operand: A[row][m] (+ A2), M columns
    const float* A; const float* A2; int lda, M;
    int rowsA, row0A;            // array rows per batch item / first row of the window
    // B operand
    int bmode;                   // 0 plain, 1 relu, 2 product B1*B2, 3 [X[n] | X[tap[n]] | HUP[n] | 0]
    const float* B1; const float* B2; int ldb, N, Nvalid;
    int rowsB, row0B;
    const int* tap; int tap_rows; int dil; int C, Ap; const float* hup;
    int R, nb;                   // window rows per batch item, batch items
    float* slab; int gstage, goff, ldc, gbias;   // gbias < 0: no column sums
};

__global__ __launch_bounds__(256) void k_wgrad(WgDesc w, int nch) {
    __shared__ float At[64 * 80];
    __shared__ float Bt[64 * 80];
    const int ldt = 80;
    const int ntn = (w.N + 63) / 64;
    const int m0 = (blockIdx.x / ntn) * 64, nn0 = (blockIdx.x % ntn) * 64;
    const int ch = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t total = (int64_t)w.R * w.nb;
    const int64_t per = (total + nch - 1) / nch;
    const int64_t r_begin = per * ch, r_end = r_begin + per < total ? r_begin + per : total;
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0, 0, 0, 0};
    float csum = 0.f;
    for (int64_t rs = r_begin; rs < r_end; rs += 64) {
        // stage 64 rows x 64 columns of A and B
        for (int idx = tid; idx < 64 * 64; idx += 256) {
            const int r = idx >> 6, c = idx & 63;
            const int64_t rr = rs + r;
            float a = 0.f, bv = 0.f;
            if (rr < r_end) {
                const int b = (int)(rr / w.R), i = (int)(rr - (int64_t)b * w.R);
                if (m0 + c < w.M) {
                    const size_t o = ((size_t)b * w.rowsA + w.row0A + i) * w.lda + m0 + c;
                    a = w.A[o]; if (w.A2) a += w.A2[o];
                }
                const int n = nn0 + c;
                if (n < w.Nvalid) {
                    const size_t row = (size_t)b * w.rowsB + w.row0B + i;
                    if (w.bmode == 0) bv = w.B1[row * w.ldb + n];
                    else if (w.bmode == 1) { bv = w.B1[row * w.ldb + n]; bv = bv > 0.f ? bv : 0.f; }
                    else if (w.bmode == 2) bv = w.B1[row * w.ldb + n] * w.B2[row * w.ldb + n];
                    else {
                        const int nloc = w.row0B + i;
                        if (n < w.C) bv = w.B1[row * w.C + n];
                        else if (n < 2 * w.C) {
                            const int tp = w.tap ? w.tap[(size_t)b * w.tap_rows + nloc] : nloc - w.dil;
                            bv = w.B1[((size_t)b * w.rowsB + tp) * w.C + (n - w.C)];
                        } else bv = w.hup[row * w.Ap + (n - 2 * w.C)];
                    }
                }
            }
            At[r * ldt + c] = a; Bt[r * ldt + c] = bv;
        }
        __syncthreads();
        if (w.gbias >= 0 && nn0 == 0 && tid < 64) { float s = 0.f; for (int r = 0; r < 64; ++r) s += At[r * ldt + tid]; csum += s; }
        const int g = lane >> 4, cl = lane & 15;
#pragma unroll 4
        for (int ks = 0; ks < 16; ++ks) {
            const float a = At[(4 * ks + g) * ldt + 16 * wave + cl];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, Bt[(4 * ks + g) * ldt + 16 * j + cl], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    float* out = w.slab + (size_t)ch * w.gstage;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 16 * wave + 4 * (lane >> 4) + i, n = nn0 + 16 * j + (lane & 15);
            if (m < w.M && n < w.N) out[w.goff + (size_t)m * w.ldc + n] = acc[j][i];
        }
    if (w.gbias >= 0 && nn0 == 0 && tid < 64 && m0 + tid < w.M) out[w.gbias + m0 + tid] = csum;
}

// ------------------------------------------------------------------------------------------ small backward kernels
__global__ void k_reduce_grad(const float* __restrict__ slab, const int* __restrict__ gsrc, int nch, int gstage, int64_t n, float* __restrict__ g) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int s = gsrc[i];
    float a = 0.f;
    if (s >= 0) for (int c = 0; c < nch; ++c) a += slab[(size_t)c * gstage + s];
    g[i] = a;
}

// causal conv weight grad: dW[c][q][tap] = sum over rows whose sample == q; LDS table per channel block
__global__ __launch_bounds__(256) void k_causal_bwd(TrainParams p, TrainBwd bw, int rows_per_wg) {
    extern __shared__ float tab[];                 // [2][Q][CB]
    const int C = p.C, Q = p.Q;
    const int CB = C < 64 ? C : 64;
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)p.B * p.N1;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < total ? r0 + rows_per_wg : total;
    for (int cb = 0; cb < C; cb += CB) {
        for (int i = tid; i < 2 * Q * CB; i += 256) tab[i] = 0.f;
        __syncthreads();
        const int cpr = CB;                          // channels per row handled by consecutive threads
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
        for (int i = tid; i < 2 * Q * CB; i += 256) {
            const int c = i / (2 * Q), qt = i - c * 2 * Q, q = qt >> 1, tp = qt & 1;
            const float v = tab[(tp * Q + q) * CB + c];
            if (v != 0.f) atomicAdd(&bw.gflat[p.causal_w + ((size_t)(cb + c) * Q + q) * 2 + tp], v);
            if (tp == 0) { /* bias: sum over q of tap-0 table == sum over rows */ }
        }
        // bias grad = sum over all rows = sum over q of the tap-0 table
        for (int c = tid; c < CB; c += 256) {
            float s = 0.f;
            for (int q = 0; q < Q; ++q) s += tab[(0 * Q + q) * CB + c];
            atomicAdd(&bw.gflat[p.causal_b + cb + c], s);
        }
        __syncthreads();
    }
}

// upsampling kernel grad: dw[j] = sum_{a,f} dH[a, U f + j] h[a,f];  db = sum dH   (qpnet.py:134-158)
__global__ __launch_bounds__(256) void k_up_bwd(TrainParams p, TrainBwd bw, int rows_per_wg) {
    extern __shared__ float accu[];                // [U + 1]
    const int U = p.U, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i <= U; i += 256) accu[i] = 0.f;
    __syncthreads();
    const int64_t total = (int64_t)p.B * p.N1;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg, r1 = r0 + rows_per_wg < total ? r0 + rows_per_wg : total;
    for (int64_t rr = r0 + wave; rr < r1; rr += 4) {
        const int b = (int)(rr / p.N1), n = (int)(rr - (int64_t)b * p.N1);
        const int64_t q = (int64_t)p.F * U - p.N1 + n;
        const int64_t f = q / U; const int j = (int)(q - f * U);
        float dv = 0.f, pr = 0.f;
        if (lane < p.A) { dv = bw.DHUP[((size_t)b * p.N1 + n) * p.Ap + lane]; pr = dv * p.h[((size_t)b * p.A + lane) * p.F + f]; }
        for (int s = 32; s >= 1; s >>= 1) { dv += __shfl_xor(dv, s); pr += __shfl_xor(pr, s); }
        if (lane == 0) { atomicAdd(&accu[j], pr); atomicAdd(&accu[U], dv); }
    }
    __syncthreads();
    for (int i = tid; i <= U; i += 256) atomicAdd(&bw.gflat[i < U ? p.up_w + i : p.up_b], accu[i]);
}

// torch.optim.Adam (single tensor semantics, fp32)
__global__ void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                       float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (wd != 0.f) gi += wd * w[i];
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    w[i] = w[i] - (lr / bc1) * (mi / denom);
}

// ------------------------------------------------------------------------------------------ launchers
static void launch_wgrad(const WgDesc& w, int nch, hipStream_t stream) {
    const int ntm = (w.M + 63) / 64, ntn = (w.N + 63) / 64;
    hipLaunchKernelGGL(k_wgrad, dim3(ntm * ntn, nch), dim3(256), 0, stream, w, nch);
}

int qpn_launch_bwd(const TrainParams& p, const TrainBwd& bw, hipStream_t stream) {
    const int C = p.C, S = p.S, Q = p.Q, L = p.L, B = p.B, N1 = p.N1, BL = p.BL;
    const size_t nDX = (size_t)B * N1 * C;
    const size_t lds_post = (size_t)64 * (tr_lda(Q > S ? Q : S) + tr_lda(S)) * sizeof(float);
    const size_t lds_layer = (size_t)64 * (tr_lda(C) + tr_lda(2 * C)) * sizeof(float);
    if (lds_post > 160 * 1024 || lds_layer > 160 * 1024) { qpn_set_error("backward tiles do not fit LDS"); return QPN_EINVAL; }
    if (lds_post > 48 * 1024) QPN_HIP(hipFuncSetAttribute((const void*)k_post_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_post));
    if (lds_layer > 48 * 1024) QPN_HIP(hipFuncSetAttribute((const void*)k_layer_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_layer));
    QPN_HIP(hipMemsetAsync(bw.DHUP, 0, (size_t)B * N1 * p.Ap * sizeof(float), stream));
    hipLaunchKernelGGL(k_post_bwd, dim3((BL + TR_TM - 1) / TR_TM, B), dim3(512), lds_post, stream, p, bw);
    qpn_prof_mark(PG_POST_BWD, stream);
    WgDesc w; memset(&w, 0, sizeof(w));
    w.slab = bw.slab; w.gstage = bw.gstage; w.nb = B; w.C = C; w.Ap = p.Ap;
    // post 2: dW2[q][s] = sum dlogits[t][q] relu(Y0)[t][s]
    w.A = bw.dlogits; w.A2 = nullptr; w.lda = Q; w.M = Q; w.rowsA = BL; w.row0A = 0;
    w.bmode = 1; w.B1 = p.Y0; w.ldb = S; w.N = S; w.Nvalid = S; w.rowsB = BL; w.row0B = 0; w.R = BL;
    w.goff = bw.g_p2; w.ldc = S; w.gbias = bw.g_bp2; launch_wgrad(w, bw.nch, stream);
    // post 1: dW1[o][s] = sum dY0[t][o] relu(S0)[t][s]
    w.A = bw.DY0; w.lda = S; w.M = S; w.B1 = p.S0; w.goff = bw.g_p1; w.gbias = bw.g_bp1; launch_wgrad(w, bw.nch, stream);
    // skip: dWs_l[s][c] = sum dS0[t][s] g_l[t][c]; skip bias grads (shared by all layers) = colsum(dS0)
    for (int l = 0; l < L; ++l) {
        w.A = bw.DS0; w.lda = S; w.M = S; w.rowsA = BL; w.row0A = 0;
        w.bmode = 2; w.B1 = p.SG + (size_t)l * B * N1 * C; w.B2 = p.TH + (size_t)l * B * N1 * C; w.ldb = C; w.N = C; w.Nvalid = C;
        w.rowsB = N1; w.row0B = N1 - BL; w.R = BL;
        w.goff = bw.g_ws[l]; w.ldc = C; w.gbias = l == 0 ? bw.g_bs : -1; launch_wgrad(w, bw.nch, stream);
    }
    qpn_prof_mark(PG_WGRAD, stream);
    int pp = 0;
    for (int l = L - 1; l >= 0; --l) {
        const TrLayer& ly = p.layers[l];
        const int last = l == L - 1;
        // outputs of this layer's backward go to parity pp^1: zero them (scatter target / rows without a writer)
        QPN_HIP(hipMemsetAsync(bw.DXA[pp ^ 1], 0, nDX * sizeof(float), stream));
        QPN_HIP(hipMemsetAsync(bw.DXB[pp ^ 1], 0, nDX * sizeof(float), stream));
        const int rows = N1 - ly.s_out;
        hipLaunchKernelGGL(k_layer_bwd, dim3((rows + TR_TM - 1) / TR_TM, B), dim3(256), lds_layer, stream, p, bw, l, last, pp);
        qpn_prof_mark(PG_LAYER_BWD, stream);
        // dW1 = dZ^T [x_cur | x_past | aux]
        w.A = bw.DZ; w.A2 = nullptr; w.lda = 2 * C; w.M = 2 * C; w.rowsA = N1; w.row0A = ly.s_out;
        w.bmode = 3; w.B1 = p.X + (size_t)l * B * N1 * C; w.B2 = nullptr; w.hup = p.HUP; w.N = p.Ktp; w.Nvalid = 2 * C + p.Ap;
        w.rowsB = N1; w.row0B = ly.s_out; w.R = rows; w.tap = ly.adaptive ? p.TAP + ly.tap_off : nullptr; w.tap_rows = N1; w.dil = ly.dilation;
        w.goff = bw.g_w1[l]; w.ldc = p.Ktp; w.gbias = bw.g_b1[l]; launch_wgrad(w, bw.nch, stream);
        // dWr = dXout^T g   (zero for the last layer: its residual output is unused)
        w.A = bw.DXA[pp]; w.A2 = bw.DXB[pp]; w.lda = C; w.M = C; w.rowsA = N1; w.row0A = ly.s_out;
        w.bmode = 2; w.B1 = p.SG + (size_t)l * B * N1 * C; w.B2 = p.TH + (size_t)l * B * N1 * C; w.ldb = C; w.N = C; w.Nvalid = C;
        w.rowsB = N1; w.row0B = ly.s_out; w.R = last ? 0 : rows; w.tap = nullptr;
        w.goff = bw.g_wr[l]; w.ldc = C; w.gbias = bw.g_br[l]; launch_wgrad(w, bw.nch, stream);
        qpn_prof_mark(PG_WGRAD, stream);
        pp ^= 1;
    }
    // flat gradient: slabs first (writes every entry), then the histogram-style grads on top
    hipLaunchKernelGGL(k_reduce_grad, dim3((unsigned)((bw.n_params + 255) / 256)), dim3(256), 0, stream, bw.slab, bw.gsrc, bw.nch, bw.gstage, bw.n_params, bw.gflat);
    {
        // grads w.r.t. the causal output sit in parity `pp`; k_causal_bwd reads parity 0 -> pass a shifted view
        TrainBwd b2 = bw; b2.DXA[0] = bw.DXA[pp]; b2.DXB[0] = bw.DXB[pp];
        const int64_t total = (int64_t)B * N1;
        const int nwg = 64, rpw = (int)((total + nwg - 1) / nwg);
        const int CB = C < 64 ? C : 64;
        const size_t lds_c = (size_t)2 * Q * CB * sizeof(float);
        if (lds_c > 48 * 1024) QPN_HIP(hipFuncSetAttribute((const void*)k_causal_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_c));
        hipLaunchKernelGGL(k_causal_bwd, dim3(nwg), dim3(256), lds_c, stream, p, b2, rpw);
        if (p.U > 0) hipLaunchKernelGGL(k_up_bwd, dim3(nwg), dim3(256), (size_t)(p.U + 1) * sizeof(float), stream, p, bw, rpw);
    }
    qpn_prof_mark(PG_GRAD_TAIL, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}

int qpn_launch_adam(float* w, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2, float eps, float wd, hipStream_t stream) {
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, w, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1, (float)sqrt(bc2));
    qpn_prof_mark(PG_ADAM, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
