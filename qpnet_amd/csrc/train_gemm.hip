// train_gemm.hip -- the training step for WIDE residual stacks (n_resch > 128; the repo-default QPNet has 512 channels):
// every contraction of QPNet.forward / its backward (reference src/nets/qpnet.py:239-312,626-670; loss.backward() at
// src/bin/qpnet_train.py:529-530) as an LDS-tiled fp32 MFMA GEMM with the layer-specific work fused into the operand
// loader and the epilogue.
//
// At C = 64 a layer's weights are 0.2 MB and the per-tile kernels of train_fwd/bwd.hip stream them from L2 for every
// 16-row tile.  At C = 512 they are 5.2 MB per layer (more than one XCD's L2) and a step is 2.9 TFLOP: the path is
// compute-bound (~470 FLOP/B), so both operands go through LDS in 128 x 128 x GK tiles (GK = 16) and every weight
// element is reused by 128 time rows.
//
//   k_gemm_nn  C[M,N] = A[M,K] . B[K,N]     A = activations, time-major rows (optionally a row GATHER: the pitch-
//              dependent tap, or several arrays side by side along K), B = weights packed K-major and zero padded
//              (gather kernel, once per step).  Epilogues: gate (sigma*tanh, saves both halves), residual add,
//              bias, ReLU mask, gate backward, and the 3-way input-gradient split (own row | scatter to the tap row |
//              aux features).
//   k_gemm_tn  C[M,N] = sum_t A[t,M] . B[t,N]   weight gradients: both operands are time-major activations, the time
//              axis is the contraction; split over the time axis into `nsplit` deterministic partial slabs.
//
// Tile: 256 threads = 4 waves, wave w owns rows 32w..32w+31 x all 128 columns = four 32x32 accumulators
// (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD = the fp32 peak, one A and one B float per lane per instruction, so a
// 16-deep chunk costs 8 x (1 + 4) ds_read_b32 against 32 MFMAs = 2048 matrix-core cycles).  Operands are staged
// global -> registers -> LDS with the NEXT chunk's loads in flight under the MFMAs (two LDS buffers, one barrier per
// chunk); A tile [128][GK + 1] (odd stride: conflict-free column reads), B tile [GK][128].
#include "train_common.h"
#include "qpn_handle.h"
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GM 128
#define GN 128
#ifndef GK
#define GK 16                // K depth of a k_gemm_nn chunk (16 or 32): 16 -> 33 KB of LDS per workgroup, 3 co-resident per CU (registers)
#endif                       //   instead of 2; measured 25.0 -> 28.3 steps/s on the C=512 geometry (with GKT); 4 per CU (forced 128 registers): no further gain
#ifndef GKT
#define GKT 16               // ... of a k_gemm_tn chunk (16 or 32)
#endif
#define G_LDA (GK + 1)
#ifndef GEMM_WAVES
#define GEMM_WAVES
#endif
#define G_LDB 128

enum { AL_PLAIN = 0, AL_SUM2 = 1, AL_RELU = 2 };
enum { EP_BIAS = 0, EP_GATE, EP_RES, EP_MASK, EP_DZ, EP_DA, EP_STORE };

struct GArgs {
    int M, N, K, nb;                 // rows per batch item, valid columns, K (multiple of 32), batch items (grid.z)
    int row_base;                    // time row of m = 0 (first valid output row of the layer)
    // ---- A operand: up to three arrays side by side along K
    const float* a[3]; int a_ld[3]; int a_kend[3]; int a_kvalid[3]; int a_tap[3]; int a_row0[3]; long a_bs[3];
    const float* a2;                 // AL_SUM2: second addend, same indexing as a[0]
    long a_rep_stride; int a_rep_len;   // a[0] repeats every a_rep_len columns with this stride (the L gate arrays of the skip sum)
    const int* tap; long tap_bs; int dil;   // a_tap rows: tap ? tap[b*tap_bs + n] : n - dil
    // ---- B operand
    const float* b; int ldb;
    // ---- epilogue
    const float* bias;
    float* o[3]; long o_bs; int o_ld; int o_row0;
    const float* e[3]; long e_bs; int e_ld; int e_row0;
    const float* e3; long e3_bs; int e3_ld; int e3_col0; int e3_rowmin;     // EP_DZ: skip-path gate gradients [b][t][L*C], t = n - rowmin
    int C, Ap; int adaptive; int last;
    float* dh; long dh_bs;           // EP_DA: aux-feature gradient [b][n][Ap]
    float* db; long db_bs;           // EP_DA: tap-row gradient [b][n][C]
};

__device__ __forceinline__ float sigmoid_g(float z) { return __frcp_rn(1.0f + __expf(-z)); }
__device__ __forceinline__ float tanh_g(float z) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * z)) - 1.0f; }

template <int AL, int EPI>
__global__ __launch_bounds__(256) GEMM_WAVES void k_gemm_nn(GArgs g) {
    extern __shared__ float sm[];
    float* As = sm;                              // [2][GM * G_LDA]
    float* Bs = sm + 2 * GM * G_LDA;             // [2][GK * G_LDB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * GN, m0 = blockIdx.y * GM, b = blockIdx.z;
    constexpr int TA = GK / 4, RA = 256 / TA, PA = GM / RA, PB = GK / 8;
    const int ar = tid / TA, ak4 = (tid % TA) * 4;            // A staging: rows ar + RA*p, four k
    const int bk = tid >> 5, bn4 = (tid & 31) * 4;           // B staging: k rows bk + 8p, four columns
    int rid[PA], rtap[PA];
    const bool any_tap = g.a_tap[0] | g.a_tap[1] | g.a_tap[2];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        int m = m0 + ar + RA * p; m = m < g.M ? m : g.M - 1;             // rows past the end repeat the last one (never stored)
        rid[p] = m;
        const int n = g.row_base + m;
        rtap[p] = any_tap ? (g.tap ? g.tap[(size_t)b * g.tap_bs + n] : n - g.dil) : 0;
    }
    float4 ra[PA], ra2[PA];       // (ra2: the second array of AL_SUM2)
    bool ra_valid = true;         // the staged columns are real ones.  Sum / ReLU / zeroing happen when the chunk goes to LDS: applied to the
                                  // just-loaded registers they put an s_waitcnt vmcnt in front of the chunk's MFMAs (AL_SUM2, AL_RELU did)
    // per-thread row pointers of the (up to) three side-by-side A arrays, computed once: the chunk loop only adds the chunk's
    // column offset (the row * ld products and the tap / own-row choice used to sit in front of every chunk's loads)
    const float* pa0[PA]; const float* pa1[PA]; const float* pa2[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        pa0[p] = g.a[0] + (size_t)b * g.a_bs[0] + (g.a_tap[0] ? (size_t)rtap[p] : (size_t)(g.a_row0[0] + rid[p])) * g.a_ld[0] + ak4;
        pa1[p] = g.a[1] ? g.a[1] + (size_t)b * g.a_bs[1] + (g.a_tap[1] ? (size_t)rtap[p] : (size_t)(g.a_row0[1] + rid[p])) * g.a_ld[1] + ak4 : pa0[p];
        pa2[p] = g.a[2] ? g.a[2] + (size_t)b * g.a_bs[2] + (g.a_tap[2] ? (size_t)rtap[p] : (size_t)(g.a_row0[2] + rid[p])) * g.a_ld[2] + ak4 : pa0[p];
    }
    const ptrdiff_t d2 = AL == AL_SUM2 ? g.a2 - g.a[0] : 0;
    // (macros with literal row indices, not a lambda or a loop: hipcc keeps arrays it cannot fully scalarise in scratch memory)
#define G_LOADA_ROW(P) { \
            const float* ptr_ = (s_ == 0 ? pa0[P] : s_ == 1 ? pa1[P] : pa2[P]) + off_; \
            const float4 v_ = *(const float4*)ptr_; \
            if (AL == AL_SUM2) ra2[P] = *(const float4*)(ptr_ + d2); \
            ra[P] = (AL == AL_PLAIN && !valid_) ? make_float4(0.f, 0.f, 0.f, 0.f) : v_; }      /* (plain loads: hipcc schedules this select behind the MFMAs by itself) */
#define G_LOADA(kc) { \
        const int k0_ = (kc) * GK; \
        const int s_ = k0_ < g.a_kend[0] ? 0 : k0_ < g.a_kend[1] ? 1 : 2; \
        const int kl_ = k0_ - (s_ ? g.a_kend[s_ - 1] : 0); \
        const bool valid_ = kl_ + ak4 < g.a_kvalid[s_]; \
        ra_valid = valid_; \
        ptrdiff_t off_ = valid_ ? kl_ : -ak4;              /* columns past the valid ones: a safe address (column 0), value zeroed */ \
        if (g.a_rep_len) {                                 /* a[0] repeats every a_rep_len columns (the L gate arrays of the skip sum) */ \
            const int kk_ = kl_ + ak4, l_ = kk_ / g.a_rep_len; \
            off_ = (ptrdiff_t)l_ * g.a_rep_stride + (valid_ ? kk_ - l_ * g.a_rep_len - ak4 : -ak4); \
        } \
        G_LOADA_ROW(0) G_LOADA_ROW(1) \
        if (PA > 2) { G_LOADA_ROW(2 % PA) G_LOADA_ROW(3 % PA) } }
    // (plain scalars instead of an array for the B staging registers: hipcc kept a captured float4[4] in scratch memory,
    //  80 bytes per lane stored and re-loaded every chunk with an s_waitcnt right behind the loads)
    const float* bsrc = g.b + (size_t)bk * g.ldb + n0 + bn4;
    const size_t bstep8 = (size_t)8 * g.ldb, bchunk = (size_t)GK * g.ldb;
    float4 rb0, rb1, rb2, rb3;
#define G_LOADB(kc) { const float* bp_ = bsrc + (size_t)(kc) * bchunk; rb0 = *(const float4*)bp_; rb1 = *(const float4*)(bp_ + bstep8); \
                      if (PB > 2) { rb2 = *(const float4*)(bp_ + 2 * bstep8); rb3 = *(const float4*)(bp_ + 3 * bstep8); } }
    auto put = [&](int buf) {
        float* ad = As + buf * (GM * G_LDA) + ar * G_LDA + ak4;
        float* bd = Bs + buf * (GK * G_LDB) + bk * G_LDB + bn4;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            float* d = ad + RA * p * G_LDA;
            float4 v = ra[p];
            if (AL == AL_SUM2) { v.x += ra2[p].x; v.y += ra2[p].y; v.z += ra2[p].z; v.w += ra2[p].w; }
            if (AL == AL_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            const bool keep = AL == AL_PLAIN || ra_valid;
            d[0] = keep ? v.x : 0.f; d[1] = keep ? v.y : 0.f; d[2] = keep ? v.z : 0.f; d[3] = keep ? v.w : 0.f;
        }
        *(float4*)(bd) = rb0; *(float4*)(bd + 8 * G_LDB) = rb1;
        if (PB > 2) { *(float4*)(bd + 16 * G_LDB) = rb2; *(float4*)(bd + 24 * G_LDB) = rb3; }
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nk = g.K / GK;
    rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nk > 0) { G_LOADA(0); G_LOADB(0); put(0); }
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const bool more = kc + 1 < nk;
        if (more) { G_LOADA(kc + 1); G_LOADB(kc + 1); }         // in flight under this chunk's MFMAs
        const float* A = As + (kc & 1) * (GM * G_LDA) + (32 * wave + (lane & 31)) * G_LDA + (lane >> 5);
        const float* Bq = Bs + (kc & 1) * (GK * G_LDB) + (lane >> 5) * G_LDB + (lane & 31);
        // fragments of k-step ks+1 are read from LDS before the MFMAs of k-step ks issue (one step of lookahead in registers)
        float an = A[0], bn0 = Bq[0], bn1 = Bq[32], bn2 = Bq[64], bn3 = Bq[96];
#pragma unroll
        for (int ks = 0; ks < GK / 2; ++ks) {
            const float a = an, b0 = bn0, b1 = bn1, b2 = bn2, b3 = bn3;
            if (ks + 1 < GK / 2) {
                an = A[2 * (ks + 1)];
                bn0 = Bq[2 * (ks + 1) * G_LDB]; bn1 = Bq[2 * (ks + 1) * G_LDB + 32]; bn2 = Bq[2 * (ks + 1) * G_LDB + 64]; bn3 = Bq[2 * (ks + 1) * G_LDB + 96];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) put((kc + 1) & 1);
        __syncthreads();
    }
    // ---------------- epilogue.  acc[j][r]: row = 32*wave + (r&3) + 8*(r>>2) + 4*(lane>>5), col = 32*j + (lane&31)
    const int cl = lane & 31, rq = 4 * (lane >> 5);
    if (EPI == EP_GATE) {
        // tile columns: [sigma c0..c0+63 | tanh c0..c0+63]; the two halves of a channel meet in one lane (acc[j], acc[j+2])
        const int c0 = 64 * blockIdx.x;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + 32 * j + cl;
            if (c >= g.C) continue;
            const float bs = g.bias[c], bt = g.bias[g.C + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + rq;
                if (m >= g.M) continue;
                const float sg = sigmoid_g(acc[j][r] + bs), th = tanh_g(acc[j + 2][r] + bt);
                const size_t o = (size_t)b * g.o_bs + (size_t)(g.o_row0 + m) * g.o_ld + c;
                g.o[0][o] = sg; g.o[1][o] = th; g.o[2][o] = sg * th;
            }
        }
        return;
    }
    if (EPI == EP_DZ) {
        // dg = dXout.Wr^T + skip-path part (rows of the last batch_length only); dz = dg * gate'.  Branch-free, the 3 x 16 operand
        // loads of a column batch in flight together (a per-element branch around the skip-path load serialised them: 34 TFLOP/s)
        const float* sgp = g.e[0] + (size_t)b * g.e_bs; const float* thp = g.e[1] + (size_t)b * g.e_bs;
        const float* dgp = g.e3 + (size_t)b * g.e3_bs + g.e3_col0;
        float* op = g.o[0] + (size_t)b * g.o_bs;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 32 * j + cl;
            if (n >= g.N) continue;
            // four rows per batch: 12 operand registers in flight (sixteen rows cost 48 and held the kernel at two workgroups per CU)
#pragma unroll
            for (int hb = 0; hb < 4; ++hb) {
                float sg[4], th[4], ds[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * hb + q;
                    int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + rq; m = m < g.M ? m : g.M - 1;
                    const int row = g.e_row0 + m;
                    const size_t eo = (size_t)row * g.e_ld + n;
                    sg[q] = sgp[eo]; th[q] = thp[eo];
                    const int wr = row - g.e3_rowmin;
                    const float t = dgp[(size_t)(wr > 0 ? wr : 0) * g.e3_ld + n];
                    ds[q] = wr >= 0 ? t : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * hb + q;
                    const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + rq;
                    if (m >= g.M) continue;
                    const float dg = acc[j][r] + ds[q];
                    const size_t oo = (size_t)(g.o_row0 + m) * g.o_ld + n;
                    op[oo] = dg * th[q] * sg[q] * (1.0f - sg[q]);
                    op[oo + g.C] = dg * sg[q] * (1.0f - th[q] * th[q]);
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 32 * j + cl;
        if (n >= g.N) continue;
        const float bias = (EPI == EP_BIAS || EPI == EP_RES) ? g.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + rq;
            if (m >= g.M) continue;
            const float v = acc[j][r];
            if (EPI == EP_BIAS || EPI == EP_STORE) {
                g.o[0][(size_t)b * g.o_bs + (size_t)(g.o_row0 + m) * g.o_ld + n] = v + bias;
            } else if (EPI == EP_RES) {            // x_out = (W g + b) + x_in
                g.o[0][(size_t)b * g.o_bs + (size_t)(g.o_row0 + m) * g.o_ld + n] =
                    (v + bias) + g.e[0][(size_t)b * g.e_bs + (size_t)(g.e_row0 + m) * g.e_ld + n];
            } else if (EPI == EP_MASK) {           // backward of a ReLU: pass where the saved pre-activation was positive
                const float pre = g.e[0][(size_t)b * g.e_bs + (size_t)(g.e_row0 + m) * g.e_ld + n];
                g.o[0][(size_t)b * g.o_bs + (size_t)(g.o_row0 + m) * g.o_ld + n] = pre > 0.f ? v : 0.f;
            } else if (EPI == EP_DZ) {             // handled below (all loads of a column batch issued together)
            } else if (EPI == EP_DA) {             // d[x_cur | x_past | aux] = dZ . W1^T
                const int row = g.e_row0 + m;
                if (n < g.C) {                     // own row, plus the residual path's gradient
                    float add = 0.f;
                    if (!g.last) { const size_t eo = (size_t)b * g.e_bs + (size_t)row * g.e_ld + n; add = g.e[0][eo] + g.e[1][eo]; }
                    g.o[0][(size_t)b * g.o_bs + (size_t)row * g.o_ld + n] = v + add;
                } else if (n < 2 * g.C) {          // backward of the gather: scatter to the tap row (collisions only when pitch-adaptive)
                    const int tr = g.tap ? g.tap[(size_t)b * g.tap_bs + row] : row - g.dil;
                    float* d = g.db + (size_t)b * g.db_bs + (size_t)tr * g.C + (n - g.C);
                    if (g.adaptive) atomicAdd(d, v); else *d = v;
                } else {
                    atomicAdd(g.dh + (size_t)b * g.dh_bs + (size_t)row * g.Ap + (n - 2 * g.C), v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ time contraction (weight gradients)
enum { BM_PLAIN = 0, BM_RELU = 1, BM_GATHER = 3 };
struct TArgs {
    int M, N;                        // valid rows / columns of dW
    int nb, nsplit;
    const float* a; const float* a2; long a_ls; int lda; int rowsA;      // A[t][m] (+ a2), per-layer stride, rows per batch item
    const float* b1; long b_ls; int ldb; int rowsB;                      // B[t][n]
    const float* hup; const int* tap; int C, Ap;                         // BM_GATHER: [x_cur | x_past | aux]
    float* slab; long gstage; int ldc;
    int row0A[TR_MAXL], row0B[TR_MAXL], R[TR_MAXL], goff[TR_MAXL], gbias[TR_MAXL], tap_off[TR_MAXL], dil[TR_MAXL];
};

// Staging as in k_wgrad3 (train_bwd.hip; in-kernel stamps there): a 16-row stage that lies inside one batch item and before the chunk's end
// takes ONE uniform base per operand plus per-thread offsets; only ADDRESSES are computed under the uniform branch, the loads are issued once
// behind it and the loaded registers are not touched (row masks, the two-array sum, ReLU) before the stage goes to LDS -- the first form selected
// `row valid ? v : 0` right behind each load, i.e. five s_waitcnt vmcnt(0) and ~650 instructions of 64-bit row arithmetic in front of every
// stage's 32 MFMAs.  TWO_A is a template flag for the same reason (a run-time `if (A2)` around a load makes the loaded value a phi).
template <int BMODE, bool TWO_A>
__global__ __launch_bounds__(256) GEMM_WAVES void k_gemm_tn(TArgs g) {
    extern __shared__ float sm[];
    float* As = sm;                              // [2][GKT * 128]
    float* Bs = sm + 2 * GKT * GM;                // [2][GKT * 128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * GN, m0 = blockIdx.y * GM;
    const int y = blockIdx.z / g.nsplit, sp = blockIdx.z - y * g.nsplit;
    const int Rl = g.R[y];
    const long total = (long)Rl * g.nb;
    const long per = ((total + g.nsplit - 1) / g.nsplit + GKT - 1) / GKT * GKT;
    const long kbeg = per * sp, kend = kbeg + per < total ? kbeg + per : total;
    const int sk = tid >> 5, c4 = (tid & 31) * 4;            // staging: k rows sk + 8p, four columns
    const int am = m0 + c4, bn = n0 + c4;
    const bool a_ok = am < g.M;
    const float* A = g.a + (size_t)y * g.a_ls;
    const float* A2 = g.a2 ? g.a2 + (size_t)y * g.a_ls : nullptr;
    const float* B1 = g.b1 + (size_t)y * g.b_ls;
    const int* tap = (BMODE == BM_GATHER && g.tap && g.tap_off[y] >= 0) ? g.tap + g.tap_off[y] : nullptr;
    // B column source of this thread (fixed for the whole contraction)
    int bkind = 0; bool b_ok = bn < g.N; const float* bbase = B1 + (b_ok ? bn : 0); int bstride = g.ldb;       // (a column past N: loaded from column 0, zeroed at the LDS write)
    if (BMODE == BM_GATHER) {
        if (bn < g.C) { bkind = 0; bbase = B1 + bn; bstride = g.C; }
        else if (bn < 2 * g.C) { bkind = 1; bbase = B1 + (bn - g.C); bstride = g.C; }
        else { bkind = 2; bbase = g.hup + (bn < 2 * g.C + g.Ap ? bn - 2 * g.C : 0); bstride = g.Ap; b_ok = bn < 2 * g.C + g.Ap; }
    }
    const int row0A = g.row0A[y], row0B = g.row0B[y], dil = g.dil[y];
    constexpr int PT = GKT / 8;
    float4 ra[PT], ra2[PT], rb[PT];
    int tpv[PT];                                 // gather rows of the NEXT stage's x_past columns (loaded one stage ahead)
    unsigned okm = 0;                            // bit p: staged row p is a real row
    const bool use_tap = BMODE == BM_GATHER && bkind == 1 && tap;
    const int Rli = Rl > 0 ? Rl : 1;
    auto stage_pos = [&](long k0, int& bi0, int& i0) {          // uniform
        if (g.nb > 1) { bi0 = (int)(k0 / Rli); i0 = (int)(k0 - (long)bi0 * Rli); } else { bi0 = 0; i0 = (int)k0; }
        return k0 + GKT <= kend && i0 + GKT <= Rli;
    };
    auto taps = [&](long k0) {                   // every thread loads (a per-thread condition around the load would put a wait behind it)
        if (BMODE != BM_GATHER || !tap) return;
        int bi0, i0; const bool fast = stage_pos(k0, bi0, i0);
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            long kk = k0 + sk + 8 * p; kk = kk < kend ? kk : kend - 1;
            int bi = bi0, i = i0 + sk + 8 * p;
            if (!fast) { bi = g.nb > 1 ? (int)(kk / Rli) : 0; i = (int)(kk - (long)bi * Rli); }
            tpv[p] = tap[(size_t)bi * g.rowsB + row0B + i];
        }
    };
    auto load = [&](long k0) {
        const float* pa[PT]; const float* pb[PT];
        int bi0, i0;
        if (stage_pos(k0, bi0, i0)) {
            const float* Ast = A + ((size_t)bi0 * g.rowsA + row0A + i0) * g.lda + (a_ok ? am : 0);
            const float* Bst = bbase + (size_t)bi0 * g.rowsB * bstride;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                pa[p] = Ast + (size_t)(sk + 8 * p) * g.lda;
                const int row = use_tap ? tpv[p] : (BMODE == BM_GATHER && bkind == 1) ? row0B + i0 + sk + 8 * p - dil : row0B + i0 + sk + 8 * p;
                pb[p] = Bst + (size_t)row * bstride;
            }
            okm = ~0u;
        } else {
            okm = 0;
#pragma unroll
            for (int p = 0; p < PT; ++p) {
                long kk = k0 + sk + 8 * p;
                const bool ok = kk < kend;
                kk = ok ? kk : kend - 1;
                const int bi = g.nb > 1 ? (int)(kk / Rli) : 0; const int i = (int)(kk - (long)bi * Rli);
                pa[p] = A + ((size_t)bi * g.rowsA + row0A + i) * g.lda + (a_ok ? am : 0);
                const int row = use_tap ? tpv[p] : (BMODE == BM_GATHER && bkind == 1) ? row0B + i - dil : row0B + i;
                pb[p] = bbase + ((size_t)bi * g.rowsB + row) * bstride;
                okm |= ok ? 1u << p : 0u;
            }
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            ra[p] = *(const float4*)pa[p];
            if (TWO_A) ra2[p] = *(const float4*)(pa[p] + (A2 - A));
            rb[p] = *(const float4*)pb[p];
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const bool ok = (okm >> p) & 1u;
            float4 v = ra[p], w = rb[p];
            if (TWO_A) { v.x += ra2[p].x; v.y += ra2[p].y; v.z += ra2[p].z; v.w += ra2[p].w; }
            if (BMODE == BM_RELU) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
            const bool oa = ok && a_ok, ob = ok && b_ok;
            v.x = oa ? v.x : 0.f; v.y = oa ? v.y : 0.f; v.z = oa ? v.z : 0.f; v.w = oa ? v.w : 0.f;      // (component-wise: see k_wgrad3)
            w.x = ob ? w.x : 0.f; w.y = ob ? w.y : 0.f; w.z = ob ? w.z : 0.f; w.w = ob ? w.w : 0.f;
            *(float4*)(As + buf * (GKT * GM) + (sk + 8 * p) * GM + c4) = v;
            *(float4*)(Bs + buf * (GKT * GN) + (sk + 8 * p) * GN + c4) = w;
        }
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int gbias = blockIdx.x == 0 ? g.gbias[y] : -1;
    float csum = 0.f;
    if (kbeg < kend) { taps(kbeg); load(kbeg); if (kbeg + GKT < kend) taps(kbeg + GKT); put(0); }
    __syncthreads();
    int buf = 0;
    for (long k0 = kbeg; k0 < kend; k0 += GKT, buf ^= 1) {
        const bool more = k0 + GKT < kend;
        if (more) { load(k0 + GKT); if (k0 + 2 * GKT < kend) taps(k0 + 2 * GKT); }
        const float* Aq = As + buf * (GKT * GM) + (lane >> 5) * GM + 32 * wave + (lane & 31);
        const float* Bq = Bs + buf * (GKT * GN) + (lane >> 5) * GN + (lane & 31);
        float an = Aq[0], bn0 = Bq[0], bn1 = Bq[32], bn2 = Bq[64], bn3 = Bq[96];       // one k-step of lookahead in registers
#pragma unroll
        for (int ks = 0; ks < GKT / 2; ++ks) {
            const float a = an, b0 = bn0, b1 = bn1, b2 = bn2, b3 = bn3;
            if (ks + 1 < GKT / 2) {
                an = Aq[2 * (ks + 1) * GM];
                bn0 = Bq[2 * (ks + 1) * GN]; bn1 = Bq[2 * (ks + 1) * GN + 32]; bn2 = Bq[2 * (ks + 1) * GN + 64]; bn3 = Bq[2 * (ks + 1) * GN + 96];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b3, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (gbias >= 0 && tid < GM) {              // bias gradient = column sums of A (n-tile 0 only)
            const float* Ac = As + buf * (GKT * GM) + tid;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < GKT; ++k) s += Ac[k * GM];
            csum += s;
        }
        if (more) put(buf ^ 1);
        __syncthreads();
    }
    float* out = g.slab + (size_t)sp * g.gstage;
    const int cl = lane & 31, rq = 4 * (lane >> 5);
    const int goff = g.goff[y];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + 32 * j + cl;
        if (n >= g.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + rq;
            if (m < g.M) out[goff + (size_t)m * g.ldc + n] = acc[j][r];
        }
    }
    if (gbias >= 0 && tid < GM && m0 + tid < g.M) out[gbias + m0 + tid] = csum;
}

// ------------------------------------------------------------------------------------------ host side
static const size_t LDS_NN = (size_t)(2 * GM * G_LDA + 2 * GK * G_LDB) * sizeof(float);
static const size_t LDS_TN = (size_t)(2 * GKT * GM + 2 * GKT * GN) * sizeof(float);

template <int AL, int EPI>
static void launch_nn(const GArgs& g, int ntiles_n, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)k_gemm_nn<AL, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_NN);      // (per device: set on every launch, like launch_tn)
    if (g.M <= 0) return;
    hipLaunchKernelGGL((k_gemm_nn<AL, EPI>), dim3(ntiles_n, (g.M + GM - 1) / GM, g.nb), dim3(256), LDS_NN, stream, g);
}
template <int BMODE>
static void launch_tn(const TArgs& g, int nlayers, hipStream_t stream) {
    (void)hipFuncSetAttribute((const void*)k_gemm_tn<BMODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TN);      // (per device: set on every launch)
    (void)hipFuncSetAttribute((const void*)k_gemm_tn<BMODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_TN);
    const dim3 grid((g.N + GN - 1) / GN, (g.M + GM - 1) / GM, nlayers * g.nsplit);
    if (g.a2) hipLaunchKernelGGL((k_gemm_tn<BMODE, true>), grid, dim3(256), LDS_TN, stream, g);
    else hipLaunchKernelGGL((k_gemm_tn<BMODE, false>), grid, dim3(256), LDS_TN, stream, g);
}

static GArgs gbase(const TrainParams& p) {
    GArgs g; memset(&g, 0, sizeof(g));
    g.nb = p.B; g.C = p.C; g.Ap = p.Ap;
    for (int s = 0; s < 3; ++s) { g.a_kend[s] = 1 << 30; g.a[s] = p.X; }
    return g;
}

void qpn_launch_prep(const TrainParams& p, const AuxGeom& ag, hipStream_t stream);

int qpn_launch_fwd_gemm(const TrainParams& p, const TrainGemm& w, hipStream_t stream) {
    const int C = p.C, S = p.S, Q = p.Q, L = p.L, B = p.B, N1 = p.N1, BL = p.BL;
    const size_t nDX = (size_t)B * N1 * C;
    { AuxGeom ag0; memset(&ag0, 0, sizeof(ag0)); qpn_launch_prep(p, ag0, stream); }
    qpn_prof_mark(PG_PREP, stream);
    for (int l = 0; l < L; ++l) {
        const TrLayer& ly = p.layers[l];
        const int rows = N1 - ly.s_out;
        {   // z = [x_cur | x_past | aux] . W1 -> gate (qpnet.py:626-640 / 657-670)
            GArgs g = gbase(p);
            g.M = rows; g.N = 2 * C; g.K = w.K1; g.row_base = ly.s_out;
            const float* Xin = p.X + (size_t)l * nDX;
            g.a[0] = Xin; g.a_ld[0] = C; g.a_kend[0] = C; g.a_kvalid[0] = C; g.a_tap[0] = 0; g.a_row0[0] = ly.s_out; g.a_bs[0] = (long)N1 * C;
            g.a[1] = Xin; g.a_ld[1] = C; g.a_kend[1] = 2 * C; g.a_kvalid[1] = C; g.a_tap[1] = 1; g.a_row0[1] = 0; g.a_bs[1] = (long)N1 * C;
            g.a[2] = p.HUP; g.a_ld[2] = p.Ap; g.a_kend[2] = w.K1; g.a_kvalid[2] = p.Ap; g.a_tap[2] = 0; g.a_row0[2] = ly.s_out; g.a_bs[2] = (long)N1 * p.Ap;
            g.tap = ly.adaptive ? p.TAP + ly.tap_off : nullptr; g.tap_bs = N1; g.dil = ly.dilation;
            g.b = w.wp + w.w1[l]; g.ldb = w.N1g;
            g.bias = p.bp + ly.bias1;
            g.o[0] = p.SG + (size_t)l * nDX; g.o[1] = p.TH + (size_t)l * nDX; g.o[2] = w.G + (size_t)l * nDX; g.o_bs = (long)N1 * C; g.o_ld = C; g.o_row0 = ly.s_out;
            launch_nn<AL_PLAIN, EP_GATE>(g, w.N1g / GN, stream);
        }
        if (l + 1 < L) {   // x_out = Wr g + br + x_cur   (the last block's residual output is never used, qpnet.py:306-309)
            GArgs g = gbase(p);
            g.M = rows; g.N = C; g.K = C; g.row_base = ly.s_out;
            g.a[0] = w.G + (size_t)l * nDX; g.a_ld[0] = C; g.a_kvalid[0] = C; g.a_row0[0] = ly.s_out; g.a_bs[0] = (long)N1 * C;
            g.b = w.wp + w.wr[l]; g.ldb = w.Cg;
            g.bias = p.bp + ly.biasr;
            g.o[0] = p.X + (size_t)(l + 1) * nDX; g.o_bs = (long)N1 * C; g.o_ld = C; g.o_row0 = ly.s_out;
            g.e[0] = p.X + (size_t)l * nDX; g.e_bs = (long)N1 * C; g.e_ld = C; g.e_row0 = ly.s_out;
            launch_nn<AL_PLAIN, EP_RES>(g, w.Cg / GN, stream);
        }
    }
    qpn_prof_mark(PG_LAYER_FWD, stream);
    {   // skip sum over all layers as ONE K = L*C contraction over the last batch_length rows (qpnet.py:283,306,309)
        GArgs g = gbase(p);
        g.M = BL; g.N = S; g.K = L * C; g.row_base = N1 - BL;
        g.a[0] = w.G; g.a_ld[0] = C; g.a_kvalid[0] = L * C; g.a_row0[0] = N1 - BL; g.a_bs[0] = (long)N1 * C;
        g.a_rep_len = C; g.a_rep_stride = (long)nDX;
        g.b = w.wp + w.ws; g.ldb = w.Sg; g.bias = p.bp + p.bias_s;
        g.o[0] = p.S0; g.o_bs = (long)BL * S; g.o_ld = S; g.o_row0 = 0;
        launch_nn<AL_PLAIN, EP_BIAS>(g, w.Sg / GN, stream);
        // _postprocess (qpnet.py:566-571): relu -> 1x1 -> relu -> 1x1
        g = gbase(p);
        g.M = BL; g.N = S; g.K = S; g.a[0] = p.S0; g.a_ld[0] = S; g.a_kvalid[0] = S; g.a_bs[0] = (long)BL * S;
        g.b = w.wp + w.p1; g.ldb = w.Sg; g.bias = p.bp + p.bias_p1;
        g.o[0] = p.Y0; g.o_bs = (long)BL * S; g.o_ld = S;
        launch_nn<AL_RELU, EP_BIAS>(g, w.Sg / GN, stream);
        g.N = Q; g.a[0] = p.Y0; g.b = w.wp + w.p2; g.ldb = w.Qg; g.bias = p.bp + p.bias_p2;
        g.o[0] = p.logits; g.o_bs = (long)BL * Q; g.o_ld = Q;
        launch_nn<AL_RELU, EP_BIAS>(g, w.Qg / GN, stream);
    }
    qpn_prof_mark(PG_POST_FWD, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}

void qpn_launch_zero_dx(const TrainParams& p, const TrainBwd& bw, hipStream_t stream);
int qpn_launch_grad_tail(const TrainParams& p, const TrainBwd& bw, const AuxGeom* ag, hipStream_t stream, bool early_done, bool up_done);

// post-net weight gradients dW2 = dlogits^T relu(Y0), dW1 = dY0^T relu(S0) (+ bias column sums) into the `nch` partial slabs
static void launch_post_wgrad_gemm(const TrainParams& p, const TrainBwd& bw, hipStream_t stream) {
    TArgs a; memset(&a, 0, sizeof(a));
    a.nb = p.B; a.nsplit = bw.nch; a.slab = bw.slab; a.gstage = bw.gstage; a.C = p.C; a.Ap = p.Ap;
    a.M = p.Q; a.N = p.S; a.a = bw.dlogits; a.lda = p.Q; a.rowsA = p.BL; a.b1 = p.Y0; a.ldb = p.S; a.rowsB = p.BL; a.ldc = p.S;
    a.R[0] = p.BL; a.goff[0] = bw.sl->g_p2; a.gbias[0] = bw.sl->g_bp2; a.tap_off[0] = -1;
    launch_tn<BM_RELU>(a, 1, stream);
    a.M = p.S; a.a = bw.DY0; a.lda = p.S; a.b1 = p.S0; a.goff[0] = bw.sl->g_p1; a.gbias[0] = bw.sl->g_bp1;
    launch_tn<BM_RELU>(a, 1, stream);
}

int qpn_launch_bwd_gemm(const TrainParams& p, const TrainBwd& bw, const TrainGemm& w, hipStream_t stream) {
    const int C = p.C, S = p.S, Q = p.Q, L = p.L, B = p.B, N1 = p.N1, BL = p.BL;
    const size_t nDX = (size_t)B * N1 * C;
    qpn_launch_zero_dx(p, bw, stream);
    {   // post-net backward: dY0 = (dlogits . W2) * (Y0 > 0); dS0 = (dY0 . W1) * (S0 > 0); DGS = dS0 . Ws
        GArgs g = gbase(p);
        g.M = BL; g.N = S; g.K = Q; g.a[0] = bw.dlogits; g.a_ld[0] = Q; g.a_kvalid[0] = Q; g.a_bs[0] = (long)BL * Q;
        g.b = w.wp + w.p2t; g.ldb = w.Sg;
        g.o[0] = bw.DY0; g.o_bs = (long)BL * S; g.o_ld = S; g.e[0] = p.Y0; g.e_bs = (long)BL * S; g.e_ld = S;
        launch_nn<AL_PLAIN, EP_MASK>(g, w.Sg / GN, stream);
        g.K = S; g.a[0] = bw.DY0; g.a_ld[0] = S; g.a_kvalid[0] = S; g.a_bs[0] = (long)BL * S;
        g.b = w.wp + w.p1t; g.o[0] = bw.DS0; g.e[0] = p.S0;
        launch_nn<AL_PLAIN, EP_MASK>(g, w.Sg / GN, stream);
        g = gbase(p);
        g.M = BL; g.N = L * C; g.K = S; g.a[0] = bw.DS0; g.a_ld[0] = S; g.a_kvalid[0] = S; g.a_bs[0] = (long)BL * S;
        g.b = w.wp + w.wst; g.ldb = w.LCg;
        g.o[0] = bw.DGS; g.o_bs = (long)BL * L * C; g.o_ld = L * C;
        launch_nn<AL_PLAIN, EP_STORE>(g, w.LCg / GN, stream);
    }
    qpn_prof_mark(PG_POST_BWD, stream);
    for (int l = L - 1; l >= 0; --l) {
        const TrLayer& ly = p.layers[l];
        const int rows = N1 - ly.s_out;
        const bool last = l == L - 1;
        float* DZl = bw.DZ + (size_t)l * B * N1 * 2 * C;
        {   // dg = dXout . Wr^T + DGS_l ; dz = dg * gate'
            GArgs g = gbase(p);
            g.M = rows; g.N = C; g.K = last ? 0 : C; g.row_base = ly.s_out;
            g.a[0] = bw.DXA[0] + (size_t)(l + 1) * nDX; g.a2 = bw.DXB[0] + (size_t)(l + 1) * nDX;
            g.a_ld[0] = C; g.a_kvalid[0] = C; g.a_row0[0] = ly.s_out; g.a_bs[0] = (long)N1 * C;
            g.b = w.wp + w.wrt[l]; g.ldb = w.Cg;
            g.o[0] = DZl; g.o_bs = (long)N1 * 2 * C; g.o_ld = 2 * C; g.o_row0 = ly.s_out;
            g.e[0] = p.SG + (size_t)l * nDX; g.e[1] = p.TH + (size_t)l * nDX; g.e_bs = (long)N1 * C; g.e_ld = C; g.e_row0 = ly.s_out;
            g.e3 = bw.DGS; g.e3_bs = (long)BL * L * C; g.e3_ld = L * C; g.e3_col0 = l * C; g.e3_rowmin = N1 - BL;
            launch_nn<AL_SUM2, EP_DZ>(g, w.Cg / GN, stream);
        }
        {   // d[x_cur | x_past | aux] = dZ . W1^T
            GArgs g = gbase(p);
            g.M = rows; g.N = 2 * C + p.Ap; g.K = 2 * C; g.row_base = ly.s_out;
            g.a[0] = DZl; g.a_ld[0] = 2 * C; g.a_kvalid[0] = 2 * C; g.a_row0[0] = ly.s_out; g.a_bs[0] = (long)N1 * 2 * C;
            g.b = w.wp + w.w1t[l]; g.ldb = w.Ktg;
            g.tap = ly.adaptive ? p.TAP + ly.tap_off : nullptr; g.tap_bs = N1; g.dil = ly.dilation; g.adaptive = ly.adaptive; g.last = last;
            g.o[0] = bw.DXA[0] + (size_t)l * nDX; g.o_bs = (long)N1 * C; g.o_ld = C;
            g.e[0] = bw.DXA[0] + (size_t)(l + 1) * nDX; g.e[1] = bw.DXB[0] + (size_t)(l + 1) * nDX; g.e_bs = (long)N1 * C; g.e_ld = C; g.e_row0 = ly.s_out;
            g.db = bw.DXB[0] + (size_t)l * nDX; g.db_bs = (long)N1 * C;
            g.dh = bw.DHUP; g.dh_bs = (long)N1 * p.Ap;
            launch_nn<AL_PLAIN, EP_DA>(g, w.Ktg / GN, stream);
        }
    }
    qpn_prof_mark(PG_LAYER_BWD, stream);
    // ---- weight gradients: time contractions into `nch` partial slabs
    TArgs t; memset(&t, 0, sizeof(t));
    t.nb = B; t.nsplit = bw.nch; t.slab = bw.slab; t.gstage = bw.gstage; t.hup = p.HUP; t.tap = p.TAP; t.C = C; t.Ap = p.Ap;
    {   // dW1_l = dZ_l^T [x_cur | x_past | aux]; bias1 grads = colsum(dZ_l)
        TArgs a = t;
        a.M = 2 * C; a.N = 2 * C + p.Ap; a.a = bw.DZ; a.a_ls = (long)B * N1 * 2 * C; a.lda = 2 * C; a.rowsA = N1;
        a.b1 = p.X; a.b_ls = (long)nDX; a.ldb = C; a.rowsB = N1; a.ldc = p.Ktp;
        for (int l = 0; l < L; ++l) {
            const TrLayer& ly = p.layers[l];
            a.row0A[l] = a.row0B[l] = ly.s_out; a.R[l] = N1 - ly.s_out; a.goff[l] = bw.sl->g_w1[l]; a.gbias[l] = bw.sl->g_b1[l];
            a.tap_off[l] = ly.adaptive ? ly.tap_off : -1; a.dil[l] = ly.dilation;
        }
        launch_tn<BM_GATHER>(a, L, stream);
    }
    {   // dWr_l = dXout_l^T g_l
        TArgs a = t;
        a.M = C; a.N = C; a.a = bw.DXA[0] + nDX; a.a2 = bw.DXB[0] + nDX; a.a_ls = (long)nDX; a.lda = C; a.rowsA = N1;
        a.b1 = w.G; a.b_ls = (long)nDX; a.ldb = C; a.rowsB = N1; a.ldc = C;
        for (int l = 0; l < L; ++l) {
            a.row0A[l] = a.row0B[l] = p.layers[l].s_out; a.R[l] = l == L - 1 ? 0 : N1 - p.layers[l].s_out;
            a.goff[l] = bw.sl->g_wr[l]; a.gbias[l] = bw.sl->g_br[l]; a.tap_off[l] = -1;
        }
        launch_tn<BM_PLAIN>(a, L, stream);
    }
    {   // dWs_l = dS0^T g_l over the last BL rows; the shared skip-bias grad = colsum(dS0)
        TArgs a = t;
        a.M = S; a.N = C; a.a = bw.DS0; a.a_ls = 0; a.lda = S; a.rowsA = BL;
        a.b1 = w.G; a.b_ls = (long)nDX; a.ldb = C; a.rowsB = N1; a.ldc = C;
        for (int l = 0; l < L; ++l) { a.row0A[l] = 0; a.row0B[l] = N1 - BL; a.R[l] = BL; a.goff[l] = bw.sl->g_ws[l]; a.gbias[l] = l == 0 ? bw.sl->g_bs : -1; a.tap_off[l] = -1; }
        launch_tn<BM_PLAIN>(a, L, stream);
    }
    launch_post_wgrad_gemm(p, bw, stream);
    qpn_prof_mark(PG_WGRAD, stream);
    return qpn_launch_grad_tail(p, bw, nullptr, stream, false, false);
}
