// train_fwd.hip -- teacher-forced forward of QPNet on gfx950 (MI355X): QPNet.forward
// (reference src/nets/qpnet.py:239-312) as fused fp32-MFMA kernels.
//
//   k_train_prep : OneHot+CausalConv1d as a 2-row table lookup (qpnet.py:60-79,110-132), the
//                  ConvTranspose2d upsampling of h (qpnet.py:134-158) and the pitch-dependent
//                  tap table _dilated_index (qpnet.py:592-604) -- no (B,T,256) one-hot, no
//                  C-fold replicated int64 index tensor.
//   k_layer_fwd  : one gated residual block (qpnet.py:626-640 / 657-670) per launch:
//                  gather(prev, pitch tap) -> [x_cur | x_past | aux] . W1 on MFMA -> sigmoid*tanh
//                  in registers -> res 1x1 on MFMA + residual.  The skip 1x1 is NOT done here:
//   k_post_fwd   : sum_l Ws_l g_l (one K = L*C contraction over the saved gates, last
//                  batch_length rows only, qpnet.py:283,306,309) -> relu -> 1x1 -> relu -> 1x1
//                  (_postprocess, qpnet.py:566-571), all on MFMA, logits written time-major
//                  (B, BL, Q) exactly as the reference returns them.
#include "train_common.h"
#include "train_post.h"

// gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32): ~6 instructions each instead of
// the ~50 of libm's expf/tanhf, which made the epilogue as long as the GEMM; abs error ~1e-7 (tolerance: loss 1e-4)
// (__frcp_rn is a correctly rounded division: div_scale / rcp / 4 fma / div_fmas / div_fixup, 11 instructions; v_rcp_f32 is 1 ulp)
__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float tanhf_(float z) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * z)) - 1.0f; }

// Aux hoist: PA_l[b][fx][n] = sum_a Va_l[n][a] h[b][a][ffirst + fx] for every layer, gate row n (sigma rows, then tanh rows) and frame the
// chunk's rows touch -- the auxiliary 1x1 (reference src/nets/qpnet.py:215-216, 663-664) applied BEFORE the rank-1 upsampling it commutes with
// (qpnet.py:134-158): L x nfr x 2C x n_aux MACs per batch item instead of L x N1 x 2C x 48.  A role of k_train_prep's launch: one workgroup per
// (AUXP_FPB frames, layer, batch item); thread = (gate row n, half of the frames).
#define AUXP_FPB 8                                    // frames per workgroup
__device__ __forceinline__ void aux_proj_block(const TrainParams& p, const AuxGeom& ag, int fxb, int l, int b, float* sm) {
    // sm: Va_l [2C][A] (A odd: conflict-free column reads; even A: two-way), h tile [A][AUXP_FPB]
    const int fx0 = fxb * AUXP_FPB, tid = threadIdx.x, n = tid & 127, fh = tid >> 7, C = p.C, A = p.A, CA = C * A;
    float* Va = sm; float* hs = sm + 2 * CA;
    {   // the sigma rows' and the tanh rows' weights are two contiguous [C][A] blocks of the flat parameters: a linear copy, 16 words in flight per thread
        const float* s0 = p.flat + ag.auxS[l]; const float* s1 = p.flat + ag.auxT[l];
        for (int base = 0; base < 2 * CA; base += 256 * 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int i = base + 256 * k + tid, ic = i < 2 * CA ? i : 0; v[k] = ic < CA ? s0[ic] : s1[ic - CA]; }
#pragma unroll
            for (int k = 0; k < 16; ++k) { const int i = base + 256 * k + tid; if (i < 2 * CA) Va[i] = v[k]; }
        }
    }
    for (int i = tid; i < A * AUXP_FPB; i += 256) {
        const int a = i / AUXP_FPB, f = i - a * AUXP_FPB, fx = fx0 + f;
        hs[i] = fx < p.nfr ? p.h[((size_t)b * A + a) * p.F + p.ffirst + fx] : 0.f;       // (the padding row behind the last frame: zeros)
    }
    __syncthreads();
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, vs = 0.f;
    for (int a = 0; a < A; ++a) {
        const float v = Va[n * A + a];
        const float4 h4 = *(const float4*)(hs + a * AUXP_FPB + 4 * fh);
        acc[0] += v * h4.x; acc[1] += v * h4.y; acc[2] += v * h4.z; acc[3] += v * h4.w;
        vs += v;
    }
    float* out = p.PA + ((size_t)(l * p.B + b) * (p.nfr + 1) + fx0 + 4 * fh) * 2 * C + n;
#pragma unroll
    for (int f = 0; f < 4; ++f) if (fx0 + 4 * fh + f <= p.nfr) out[(size_t)f * 2 * C] = acc[f];
    // the constant part of the upsampled features, b_up * sum_a Va[n][a], joins the packed gate bias (written by k_refresh, earlier on this stream): one writer per (layer, gate row)
    if (fxb == 0 && b == 0 && fh == 0) p.bp[ag.bias1[l] + n] += p.flat[p.up_b] * vs;
}

// Elementwise over (row, 4 channels): every access is a coalesced 16-byte load / store, no per-row serialisation.
//   items [0, N1*C/4)                : X0[n][c..c+3] = tap-0 row of class x[n] + tap-1 row of class x[n+1] + bias  (transposed table p.ct)
//   items [.., + N1*Ap/4)            : upsampled aux features HUP[n][a..a+3]
//   items [.., + N1*L)               : tap row of layer l at row n: pitch-dependent (adaptive) or n - dilation (fixed) -- a table for
//                                      every layer lets the consumers load taps without a branch (+ the class ids XC for the table gradient)
//   aux hoist: the workgroups behind the first `nprep` are aux_proj_block's
__global__ __launch_bounds__(256) void k_train_prep(TrainParams p, AuxGeom ag, int nprep) {
    const int b = blockIdx.y;
    if ((int)blockIdx.x >= nprep) {
        extern __shared__ float sm[];
        const int idx = blockIdx.x - nprep, nfxb = (p.nfr + 1 + AUXP_FPB - 1) / AUXP_FPB;
        aux_proj_block(p, ag, idx % nfxb, idx / nfxb, b, sm);
        return;
    }
    const int C = p.C, Q = p.Q, N1 = p.N1, Ap = p.Ap;
    const int C4 = C / 4, A4 = Ap / 4;
    const int nX = N1 * C4, nH = p.hoist ? 0 : N1 * A4;          // (aux hoist: no sample-rate aux rows; the frame-rate projections are k_aux_proj's)
    const int total = nX + nH + N1 * p.L;
    if (p.hoist && b == 0)         // row -> {w_up[j], j}: the upsampling weight and within-frame offset of every row (and of 16 rows past the end: the pattern continues)
        for (int n = blockIdx.x * 256 + threadIdx.x; n < N1 + 16; n += nprep * 256) {
            const int q = p.F * p.U - N1 + n, j = q - (q / p.U) * p.U;
            p.WJ[n] = make_float2(p.flat[p.up_w + j], __int_as_float(j));
        }
    if (p.qctl && blockIdx.x == 0 && b == 0 && threadIdx.x < 32) p.qctl[threadIdx.x < 16 ? threadIdx.x : 1024 + (threadIdx.x - 16) * TR_QHEAD_STRIDE] = 0u;      // abort word / counters / sub-queue heads of this step's stack queues (train_stack.hip)
    if (p.qtab && b == 0)          // tile table of the one-launch residual stack (train_stack.hip)
        for (int pos = blockIdx.x * 256 + threadIdx.x; pos <= p.qtotal; pos += nprep * 256) {
            int4 ea, eb; tr_queue_entry_fwd(p, pos, ea, eb);
            p.qtab[2 * pos] = ea; p.qtab[2 * pos + 1] = eb;
            if (p.qtab_b) { tr_queue_entry_bwd(p, pos, ea, eb); p.qtab_b[2 * pos] = ea; p.qtab_b[2 * pos + 1] = eb; }
        }
    for (int it = blockIdx.x * 256 + threadIdx.x; it < total; it += nprep * 256) {
        if (it < nX) {
            const int n = it / C4, c = (it - n * C4) * 4;
            const int64_t xo = (int64_t)p.T - p.N0 + n;
            int64_t s0 = p.x[(size_t)b * p.T + xo] % Q, s1 = p.x[(size_t)b * p.T + xo + 1] % Q;
            if (s0 < 0) s0 += Q;
            if (s1 < 0) s1 += Q;
            const float4 r0 = *(const float4*)(p.ct + (size_t)s0 * C + c);               // tap 0 = older sample (qpnet.py:110-132)
            const float4 r1 = *(const float4*)(p.ct + ((size_t)Q + s1) * C + c);
            const float4 bb = *(const float4*)(p.flat + p.causal_b + c);
            float4 v;
            v.x = (r0.x + r1.x) + bb.x; v.y = (r0.y + r1.y) + bb.y; v.z = (r0.z + r1.z) + bb.z; v.w = (r0.w + r1.w) + bb.w;
            *(float4*)(p.X + ((size_t)b * N1 + n) * C + c) = v;
            if (c == 0) { p.XC[(size_t)b * (N1 + 1) + n] = (int)s0; if (n == N1 - 1) p.XC[(size_t)b * (N1 + 1) + N1] = (int)s1; }
        } else if (it < nX + nH) {
            // upsampled aux features, aligned at the END of h_up (negative hindex, qpnet.py:269-276)
            const int i2 = it - nX, n = i2 / A4, a0 = (i2 - n * A4) * 4;
            float v[4];
            int64_t f = 0; int j = 0;
            if (p.U > 0) { const int64_t q = (int64_t)p.F * p.U - N1 + n; f = q / p.U; j = (int)(q - f * p.U); }
            else f = (int64_t)p.F - N1 + n;
            const float wj = p.U > 0 ? p.flat[p.up_w + j] : 1.0f, ub = p.U > 0 ? p.flat[p.up_b] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = a0 + e;
                float t = 0.0f;
                if (a < p.A) { t = p.h[((size_t)b * p.A + a) * p.F + f]; if (p.U > 0) t = t * wj + ub; }
                v[e] = t;
            }
            *(float4*)(p.HUP + ((size_t)b * N1 + n) * Ap + a0) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            // pitch-dependent taps: N1 + rint(float32(-d*dil) + float32(idx)), idx = n - N1 (qpnet.py:595-600)
            const int i3 = it - nX - nH, l = i3 / N1, n = i3 - l * N1;
            const TrLayer ly = p.layers[l];
            int tap;
            if (ly.adaptive) {
                const float dv = p.d[(size_t)b * p.Td + (p.Td - N1 + n)];
                const float dil = -dv * (float)ly.dilation;
                const float s = __fadd_rn(dil, (float)(n - N1));
                tap = N1 + (int)rintf(s);
                if (n >= ly.s_out && (tap < ly.s_in || tap > n)) { atomicOr(p.status, 1); }   // reference assert (qpnet.py:294)
            } else tap = n - ly.dilation;
            tap = tap < 0 ? 0 : (tap >= N1 ? N1 - 1 : tap);
            p.TAP[ly.tap_off + (size_t)b * N1 + n] = tap;
        }
    }
}

// One gated residual block on one tile of TM = 16*MT time rows starting at row n0 of batch item b.
// dynamic LDS: As[TM][lda(Ktp)] | Gs[TM][lda(C)]
template <int MT>
__device__ __forceinline__ void layer_fwd_tile(const TrainParams& p, const int l, const int last, const int b, const int n0, float* sm) {
    constexpr int TM = 16 * MT;
    const TrLayer ly = p.layers[l];
    const int C = p.C, Ktp = p.Ktp, Ap = p.Ap;
    const int lda = tr_lda(Ktp), ldg = tr_lda(C);
    float* As = sm; float* Gs = sm + TM * lda;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* Xin = p.X + ((size_t)(l * p.B + b) * p.N1) * C;
    const float* hup = p.HUP + (size_t)b * p.N1 * Ap;
    const int* taps = p.TAP + ly.tap_off + (size_t)b * p.N1;      // every layer has a table: no branch (and no s_waitcnt) around the load
    // the biases the two epilogues add are requested here and pinned (empty asm) in front of the gate contraction: loaded where
    // they are used, each sits behind its own s_waitcnt right after the MFMA loop -- an exposed L2 round trip per epilogue
    const float* bias1 = p.bp + ly.bias1;
    const float* br = p.bp + ly.biasr;
    const int cpre = 16 * (wave < C / 16 ? wave : 0) + (lane & 15);
    const float bs_pre = bias1[cpre], bt_pre = bias1[C + cpre], bb_pre = br[cpre];
    // ... and so are the residual-1x1 weight fragments of this wave's n-tile (consumed behind the gate epilogue's barrier)
    float4 wrq[4][1];
    if (!last) { const int ntp[1] = {wave < C / 16 ? wave : 0}; wave_b_preload<1, 4>(wrq, p.wp + ly.wr_f4, C / 16, ntp, C, lane); }
    // ---- stage [x_cur | x_past | aux | 0] rows into LDS: 16-byte global loads, 8-byte LDS stores (lda is even, not /4).
    //      All tap rows first, then all row segments, then the LDS stores: two memory round trips per tile, not two per row.
    {
        const int K4 = Ktp / 4;                       // float4 columns per row
        const int tpr = K4 < 64 ? K4 : 64;            // threads per row (one float4 each, extra columns in a second pass)
        const int rpp = 256 / tpr;                    // rows per pass (>= 4)
        const int tr = tid / tpr, tc = tid - tr * tpr;
        constexpr int NR = TM / 4;                    // passes needed at most
        if (tr < rpp) {
            int tp[NR];
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int n = n0 + tr + k * rpp;
                tp[k] = taps[n < p.N1 ? n : p.N1 - 1];
            }
            for (int c4 = tc; c4 < K4; c4 += tpr) {
                const int k0 = c4 * 4;
                const int kind = k0 < C ? 0 : k0 < 2 * C ? 1 : k0 < 2 * C + Ap ? 2 : 3;
                // source array / column / row stride as integer selects (the nested pointer ternary became divergent branches whose
                // joins carry s_waitcnt vmcnt(0): they waited for every load requested above, taps or not)
                const bool from_x = kind <= 1;
                const int col = kind == 0 ? k0 : kind == 1 ? k0 - C : kind == 2 ? k0 - 2 * C : 0;
                const unsigned long long ax = (unsigned long long)Xin, ah = (unsigned long long)hup;
                const float* base = (const float*)(from_x ? ax : ah) + col;
                const int stride = from_x ? C : Ap;
                float4 v[NR];
#pragma unroll
                for (int k = 0; k < NR; ++k) {          // loads only: a select on the loaded value here makes hipcc wait for each load before it issues the next
                    const int r = tr + k * rpp, n = n0 + r;
                    const bool ok = r < TM && n < p.N1 && kind != 3;
                    const int row = ok ? (kind == 1 ? tp[k] : n) : 0;
                    v[k] = *(const float4*)(base + (size_t)row * stride);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    const int r = tr + k * rpp;
                    if (r < TM) {
                        const bool ok = n0 + r < p.N1 && kind != 3;
                        const float4 w = ok ? v[k] : make_float4(0.f, 0.f, 0.f, 0.f);
                        float* dst = As + (size_t)r * lda + k0;
                        *(float2*)dst = make_float2(w.x, w.y); *(float2*)(dst + 2) = make_float2(w.z, w.w);
                    }
                }
            }
        }
    }
    __syncthreads();
    const float4* W1 = p.wp + ly.w1_f4;
    asm volatile("" :: "v"(bs_pre), "v"(bt_pre), "v"(bb_pre));
    float* SG = p.SG + ((size_t)(l * p.B + b) * p.N1) * C;
    float* TH = p.TH + ((size_t)(l * p.B + b) * p.N1) * C;
    const int NCG = C / 16;
    for (int cg = wave; cg < NCG; cg += 4) {
        f32x4 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
        const int nts[2] = {cg, NCG + cg};
        wave_gemm<MT, 2>(acc, As, lda, W1, 2 * NCG, nts, Ktp, lane);
        const int c = 16 * cg + (lane & 15);
        const float bs = cg == wave ? bs_pre : bias1[c], bt = cg == wave ? bt_pre : bias1[C + c];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i;
                const float sg = sigmoidf_(acc[mt][0][i] + bs);
                const float th = tanhf_(acc[mt][1][i] + bt);
                const float gg = sg * th;
                Gs[(size_t)r * ldg + c] = gg;
                if (n0 + r < p.N1) { SG[(size_t)(n0 + r) * C + c] = sg; TH[(size_t)(n0 + r) * C + c] = gg; }      // (p.TH holds the gate PRODUCT sigma * tanh: train_common.h)
            }
    }
    if (last) return;               // the last block's residual output is never used (qpnet.py:306-309)
    __syncthreads();
    const float4* Wr = p.wp + ly.wr_f4;
    float* Xout = p.X + ((size_t)((l + 1) * p.B + b) * p.N1) * C;
    for (int nt = wave; nt < NCG; nt += 4) {
        f32x4 acc[MT][1];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt][0] = (f32x4){0, 0, 0, 0};
        const int nts[1] = {nt};
        if (nt == wave) wave_gemm_run<MT, 1, 4>(acc, Gs, ldg, wrq, Wr, NCG, nts, C, lane);
        else wave_gemm_deep<MT, 1, 4>(acc, Gs, ldg, Wr, NCG, nts, C, lane);
        const int c = 16 * nt + (lane & 15);
        const float bb = nt == wave ? bb_pre : br[c];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i;
                if (n0 + r < p.N1) Xout[(size_t)(n0 + r) * C + c] = (acc[mt][0][i] + bb) + As[(size_t)r * lda + c];
            }
    }
}

// one layer per launch.  [A persistent all-layers kernel with per-tile completion flags was tried: on this multi-XCD part
// an agent-scope release is a whole-L2 write-back (buffer_wbl2 sc1) per tile, 4x slower than the eight launches.]
template <int MT>
__global__ __launch_bounds__(256) void k_layer_fwd(TrainParams p, int l, int last) {      // last: bit 0 last layer, bit 1 XCD swizzle
    extern __shared__ float sm[];
    layer_fwd_tile<MT>(p, l, last & 1, blockIdx.y, p.layers[l].s_out + tr_xcd_tile(blockIdx.x, gridDim.x, last & 2) * 16 * MT, sm);
}

// ------------------------------------------------------------------------------------------------ persistent form, n_resch = 64
// One launch per layer as before, but 2 workgroups per CU that each walk a contiguous range of 16-row tiles, and the layer's
// weights live in REGISTERS for the whole launch: wave w owns gate column tiles {w (sigma), 4 + w (tanh)} and residual column
// tile w, i.e. KS x 2 + 4 fragment float4s per lane (104 registers at KS = 11), read from L2 once per workgroup instead of once
// per tile (115 MB -> 27 MB per launch) -- the contraction loop has no global load in it.  Around it:
//  * the rows of tile t+1 are requested before the matrix cores start on tile t and go to the other LDS buffer behind them
//    (their tap-table entry one tile earlier still): neither round trip of the pitch-dependent gather is exposed after tile 0;
//    every thread's three row pieces have a FIXED source (x_cur | x_past | aux): straight-line address code, no divergence;
//  * sigma, tanh and the block output leave through LDS: a thread stores 8-byte pieces of whole 256-byte rows (full cache
//    lines, one address per array and tile) instead of 64-byte column strips straight from the accumulator layout;
//  * barriers order LDS only (TR_LDS_BARRIER): the prefetch and the stores stay in flight across them.
// (reference: the fixed / adaptive gated block of src/nets/qpnet.py:626-670; same arithmetic as layer_fwd_tile above.)
template <int KS, bool LAST>
__global__ __launch_bounds__(256, 2) void k_layer_fwd_p(TrainParams p, int l, int flags, int tiles, float* dummy) {     // flags: bit 1 XCD swizzle
    constexpr int C = 64, Ktp = 16 * KS;
    constexpr bool HOIST = KS == 8;                              // K = 2C: the auxiliary 1x1 at frame rate (TrainParams::hoist; one extra MFMA step per accumulator)
    constexpr int lda = ((Ktp + 29) / 32) * 32 + 2, ldg = ((C + 29) / 32) * 32 + 2;     // tr_lda: conflict-free fragment reads, 8-byte aligned rows
    extern __shared__ float sm[];
    float* Gs = sm + 32 * lda;                                   // As buffers: sm, sm + 16 * lda
    float* SGs = Gs + 16 * ldg; float* THs = SGs + 16 * ldg; float* Xs = THs + 16 * ldg;
    const TrLayer ly = p.layers[l];
    const int Ap = p.Ap, N1 = p.N1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y;
    int t_first, t_count;
    tr_tile_range(blockIdx.x, gridDim.x, tiles, flags & 2, t_first, t_count);
    if (t_count <= 0) return;
    const int t_last = t_first + t_count - 1;
    const float* Xin = p.X + ((size_t)(l * p.B + b) * N1) * C;
    const float* hup = p.HUP + (size_t)b * N1 * Ap;
    const int* taps = p.TAP + ly.tap_off + (size_t)b * N1;
    float* SG = p.SG + ((size_t)(l * p.B + b) * N1) * C;
    float* TH = p.TH + ((size_t)(l * p.B + b) * N1) * C;
    float* Xout = p.X + ((size_t)((l + 1) * p.B + b) * N1) * C;
    // ---- resident weight fragments and biases
    const float4* W1 = p.wp + ly.w1_f4; const float4* Wr = p.wp + ly.wr_f4;
    float4 w1[KS][2], wr[4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { w1[ks][0] = W1[((size_t)ks * 8 + wave) * 64 + lane]; w1[ks][1] = W1[((size_t)ks * 8 + 4 + wave) * 64 + lane]; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wr[ks] = LAST ? make_float4(0.f, 0.f, 0.f, 0.f) : Wr[((size_t)ks * 4 + wave) * 64 + lane];
    const int c = 16 * wave + (lane & 15);
    const float bs = p.bp[ly.bias1 + c], bt = p.bp[ly.bias1 + C + c], bb = p.bp[ly.biasr + c];
    // ---- staging in: thread -> row srow, 16-byte piece sc4 of x_cur, of x_past (row tap[n]) and -- threads with sc4 < 12 -- of the
    //      aux columns (zero beyond n_aux).  staging out: thread -> 8-byte pieces (row orow, orow + 8; column oc2) of a [16][64] tile.
    //      The tile loop is STRAIGHT-LINE code: every load and store is unconditional (rows past the chunk end are clamped on
    //      the way in and redirected to a scratch row on the way out), so hipcc's s_waitcnt vmcnt(N) before the use of a prefetched
    //      register counts the younger stores exactly instead of draining them (a guarded store is "maybe zero stores" to the pass)
    const int srow = tid >> 4, sc4 = tid & 15;
    const bool aux_thread = sc4 < (Ktp - 2 * C) / 4, aux_real = 4 * sc4 < Ap;
    const int orow = tid >> 5, oc2 = (tid & 31) * 2;
    float* const dmy = dummy + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 * C + oc2;       // two scratch rows per workgroup
    int tp; float4 rc, rp, rx = make_float4(0.f, 0.f, 0.f, 0.f);
    // aux hoist: the extra step's operands of the tile in flight (rwj / rpb*: requested with its rows) and of the tile in hand (xaux_*)
    float2 rwj = make_float2(0.f, 0.f); float rpb0 = 0.f, rpb1 = 0.f, xaux_a = 0.f, xaux_b0 = 0.f, xaux_b1 = 0.f;
    auto load_tap = [&](int t) { const int n = ly.s_out + t * 16 + srow; tp = taps[n < N1 ? n : N1 - 1]; };
    auto load_rows = [&](int t) {
        const int n = ly.s_out + t * 16 + srow, nn = n < N1 ? n : N1 - 1;
        // (32-bit element offsets behind uniform bases: see k_layer_bwd_p)
        rc = *(const float4*)(Xin + (__umul24((unsigned)nn, (unsigned)C) + 4u * sc4));
        rp = *(const float4*)(Xin + (__umul24((unsigned)tp, (unsigned)C) + 4u * sc4));
        if constexpr (HOIST) {
            const int n0 = ly.s_out + t * 16;
            const float* pa = p.PA + tr_pa_off(p, l, b, n0) + ((lane >> 4) & 1) * 2 * C + 16 * wave + (lane & 15);
            rwj = p.WJ[n0 + (lane & 15)];                        // (16 rows of padding behind row N1 - 1)
            rpb0 = pa[0]; rpb1 = pa[C];
        } else rx = *(const float4*)(hup + (__umul24((unsigned)nn, (unsigned)Ap) + (aux_real ? 4u * sc4 : 0u)));
    };
    auto store_rows = [&](int t, float* As) {
        const bool in = ly.s_out + t * 16 + srow < N1;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vc = in ? rc : z, vp = in ? rp : z, vx = (in && aux_real) ? rx : z;
        float* d = As + (size_t)srow * lda + 4 * sc4;
        *(float2*)d = make_float2(vc.x, vc.y); *(float2*)(d + 2) = make_float2(vc.z, vc.w);
        *(float2*)(d + C) = make_float2(vp.x, vp.y); *(float2*)(d + C + 2) = make_float2(vp.z, vp.w);
        if constexpr (HOIST) {      // B operand: lane (k = lane >> 4, column lane & 15) = PA[first frame + k][...] for k < 2, 0 for the unused k-slots
            xaux_a = tr_aux_a(rwj, lane); xaux_b0 = lane < 32 ? rpb0 : 0.f; xaux_b1 = lane < 32 ? rpb1 : 0.f;
        } else if (aux_thread) { *(float2*)(d + 2 * C) = make_float2(vx.x, vx.y); *(float2*)(d + 2 * C + 2) = make_float2(vx.z, vx.w); }
    };
    auto store_out = [&](const float* T, float* dst, int n0, bool live) {      // a [16][64] LDS tile -> rows n0.. of a [N1][64] array, whole rows
        const float2 v0 = *(const float2*)(T + (size_t)orow * ldg + oc2), v1 = *(const float2*)(T + (size_t)(orow + 8) * ldg + oc2);
        float* d0 = dst + (__umul24((unsigned)(n0 + orow), (unsigned)C) + oc2);
        float* d1 = d0 + 8 * C;
        d0 = (live && n0 + orow < N1) ? d0 : dmy;
        d1 = (live && n0 + orow + 8 < N1) ? d1 : dmy + C;
        *(float2*)d0 = v0; *(float2*)d1 = v1;
    };
    // ---- prologue: rows of the first tile, tap of the second
    load_tap(t_first);
    load_rows(t_first);
    store_rows(t_first, sm);
    load_tap(t_first + 1 < t_last ? t_first + 1 : t_last);
    const int arow = lane & 15, ak = lane >> 4;
    for (int ti = 0; ti < t_count; ++ti) {
        const int t = t_first + ti, n0 = ly.s_out + t * 16;
        float* As = sm + (ti & 1) * 16 * lda;
        load_rows(t + 1 < t_last ? t + 1 : t_last);              // in flight under this tile's contractions (past the range: a harmless reload)
        load_tap(t + 2 < t_last ? t + 2 : t_last);
        TR_LDS_BARRIER();                                          // As[ti & 1] complete (written at the end of the previous trip); Xs of the previous tile complete
        if (!LAST) store_out(Xs, Xout, n0 - 16, ti > 0);
        float xa[KS][4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* ap = As + (size_t)arow * lda + 16 * ks + ak;
            xa[ks][0] = ap[0]; xa[ks][1] = ap[4]; xa[ks][2] = ap[8]; xa[ks][3] = ap[12];
        }
        __builtin_amdgcn_sched_barrier(0);                        // all fragment reads of the tile issued before the first MFMA (hipcc otherwise
                                                                  // sinks each read to its use: an LDS round trip in front of every second MFMA pair)
        f32x4 a0 = (f32x4){0, 0, 0, 0}, a1 = (f32x4){0, 0, 0, 0};
        if constexpr (HOIST) {      // w_up[j(row)] * PA[frame(row)][column]: the auxiliary 1x1 of the tile's (at most two) frames
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xaux_a, xaux_b0, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xaux_a, xaux_b1, a1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][0], w1[ks][0].x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][0], w1[ks][1].x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][1], w1[ks][0].y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][1], w1[ks][1].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][2], w1[ks][0].z, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][2], w1[ks][1].z, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][3], w1[ks][0].w, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][3], w1[ks][1].w, a1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (4 * (lane >> 4) + i) * ldg + c;
            const float sg = sigmoidf_(a0[i] + bs), th = tanhf_(a1[i] + bt);
            Gs[o] = sg * th; SGs[o] = sg;
        }
        TR_LDS_BARRIER();
        if (!LAST) {                                              // the last block's residual output is never used (qpnet.py:306-309)
            float ga[4][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float* gp = Gs + (size_t)arow * ldg + 16 * ks + ak;
                ga[ks][0] = gp[0]; ga[ks][1] = gp[4]; ga[ks][2] = gp[8]; ga[ks][3] = gp[12];
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 ar0 = (f32x4){0, 0, 0, 0}, ar1 = (f32x4){0, 0, 0, 0};     // two chains: a dependent 16x16x4 pair is 40 cycles apart, the pipe issues every 32
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][0], wr[ks].x, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][0], wr[ks + 1].x, ar1, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][1], wr[ks].y, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][1], wr[ks + 1].y, ar1, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][2], wr[ks].z, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][2], wr[ks + 1].z, ar1, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][3], wr[ks].w, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][3], wr[ks + 1].w, ar1, 0, 0, 0);
            }
            store_out(SGs, SG, n0, true);                         // (under the residual contraction)
            store_out(Gs, TH, n0, true);                          // the gate product (p.TH: train_common.h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (lane >> 4) + i;
                Xs[r * ldg + c] = ((ar0[i] + ar1[i]) + bb) + As[(size_t)r * lda + c];      // leaves after the next barrier
            }
        } else {
            store_out(SGs, SG, n0, true);
            store_out(Gs, TH, n0, true);                          // the gate product (p.TH: train_common.h)
        }
        // the next tile's rows have had a whole tile of matrix work to arrive: into the OTHER buffer (its last readers were
        // the previous trip's contractions, two barriers ago)
        store_rows(t + 1, sm + ((ti + 1) & 1) * 16 * lda);
    }
    if (!LAST) {
        TR_LDS_BARRIER();
        store_out(Xs, Xout, ly.s_out + t_last * 16, true);
    }
}

// dynamic LDS: St[TM][lda(S)] | Yt[TM][max(lda(S), 2 lda(C))]  (the two G_l staging buffers alias Yt)
template <int MT>
__global__ __launch_bounds__(512) void k_post_fwd(TrainParams p) {
    constexpr int TM = 16 * MT;
    extern __shared__ float sm[];
    const int C = p.C, S = p.S, Q = p.Q, L = p.L;
    const int lds = tr_lda(S), ldg = tr_lda(C);
    float* St = sm; float* Yt = sm + TM * lds;
    float* Gb[2] = {Yt, Yt + TM * ldg};
    const int b = blockIdx.y, t0 = blockIdx.x * TM;      // rows of the last-BL window
    const int nbase = p.N1 - p.BL + t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NTS = S / 16, NTQ = Q / 16, NCG = C / 16;
    // the biases of the three epilogues, requested before anything else and pinned (empty asm) in front of the first contraction:
    // loaded in the epilogues they sat behind an s_waitcnt each (an exposed L2 round trip after every MFMA loop)
    float bpre[3][2];
    {
        const int npq = wave < (NTS + 1) / 2 ? wave : 0, nqq = wave < (NTQ + 1) / 2 ? wave : 0;
        const int cs0 = 16 * (2 * npq) + (lane & 15), cs1 = 16 * ((2 * npq + 1 < NTS) ? 2 * npq + 1 : 2 * npq) + (lane & 15);
        const int cq0 = 16 * (2 * nqq) + (lane & 15), cq1 = 16 * ((2 * nqq + 1 < NTQ) ? 2 * nqq + 1 : 2 * nqq) + (lane & 15);
        bpre[0][0] = p.bp[p.bias_s + cs0]; bpre[0][1] = p.bp[p.bias_s + cs1];
        bpre[1][0] = p.bp[p.bias_p1 + cs0]; bpre[1][1] = p.bp[p.bias_p1 + cs1];
        bpre[2][0] = p.bp[p.bias_p2 + cq0]; bpre[2][1] = p.bp[p.bias_p2 + cq1];
    }
#define POST_PIN_BIAS asm volatile("" :: "v"(bpre[0][0]), "v"(bpre[0][1]), "v"(bpre[1][0]), "v"(bpre[1][1]), "v"(bpre[2][0]), "v"(bpre[2][1]))
    // ---------- skip sum: K = L*C over the saved gates
    const int npairs = (NTS + 1) / 2;
    const int LC = L * C, ldall = tr_lda(LC);
    if (npairs <= 8 && (size_t)TM * ldall <= (size_t)TM * (lds + (lds > 2 * ldg ? lds : 2 * ldg))) {
        // All L gate tiles fit the workgroup's LDS at once (they alias St / Yt, which are written only afterwards): every load of
        // the tile is in flight together and the sum over layers is ONE K = L*C contraction -- one exposed memory round trip and
        // two barriers instead of L of each.
        float* Gall = sm;
        const int per_row = LC / 2;                              // float2 pairs per row
        for (int idx = tid; idx < TM * per_row; idx += 512) {
            const int r = idx / per_row, kk = (idx - r * per_row) * 2;
            const int l = kk / C, k = kk - l * C;
            float2 g = make_float2(0.f, 0.f);
            if (t0 + r < p.BL) {
                const size_t o = ((size_t)(l * p.B + b) * p.N1 + nbase + r) * C + k;
                g = *(const float2*)(p.TH + o);                     // (p.TH holds the gate product sigma * tanh: train_common.h)
            }
            *(float2*)(Gall + (size_t)r * ldall + kk) = g;
        }
        __syncthreads();
        const int np = wave;
        const bool active = np < npairs;
        const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
        f32x4 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
        POST_PIN_BIAS;
        if (active) { const int nts[2] = {nt0, nt1}; wave_gemm<MT, 2>(acc, Gall, ldall, p.wp + p.ws_f4, NTS, nts, LC, lane); }
        __syncthreads();                                         // every wave is done reading Gall before St overwrites it
        if (active) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = j ? nt1 : nt0;
                if (j && nt1 == nt0) break;
                const int c = 16 * nt + (lane & 15);
                const float bs = wave < npairs ? bpre[0][j] : p.bp[p.bias_s + c];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        const float v = acc[mt][j][i] + bs;
                        if (t0 + r < p.BL) p.S0[((size_t)b * p.BL + t0 + r) * S + c] = v;
                        St[(size_t)r * lds + c] = v > 0.f ? v : 0.f;
                    }
            }
        }
        __syncthreads();
    } else
    for (int pb = 0; pb < npairs; pb += 8) {                // pair batch (one batch when S <= 256)
        const int np = pb + wave;
        const bool active = np < npairs;
        const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
        f32x4 acc[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
        // the gate rows of layer l+1 are requested before the matrix cores start on layer l and land in LDS after it:
        // one exposed memory round trip per tile instead of one per layer
        constexpr int NG = (TM * 32 + 511) / 512;           // float2 pairs per thread for C <= 64 (more: second pass below)
        const int gper = TM * (C / 2);
        float2 gt[NG];
        auto gfetch = [&](int l) {
            const float* TH = p.TH + ((size_t)(l * p.B + b) * p.N1) * C;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const int idx = tid + k * 512;
                const int r = idx / (C / 2), kk = (idx - r * (C / 2)) * 2;
                const bool ok = idx < gper && t0 + r < p.BL;
                const size_t o = ok ? (size_t)(nbase + r) * C + kk : 0;
                const float2 t = *(const float2*)(TH + o);
                gt[k] = ok ? t : make_float2(0.f, 0.f);
            }
        };
        const bool fits = gper <= NG * 512;
        if (fits) gfetch(0);
        for (int l = 0; l < L; ++l) {
            float* G = Gb[l & 1];
            if (fits) {
#pragma unroll
                for (int k = 0; k < NG; ++k) {
                    const int idx = tid + k * 512;
                    if (idx < gper) { const int r = idx / (C / 2), kk = (idx - r * (C / 2)) * 2; *(float2*)(G + (size_t)r * ldg + kk) = gt[k]; }
                }
            } else {
                const float* TH = p.TH + ((size_t)(l * p.B + b) * p.N1) * C;
                for (int idx = tid; idx < gper; idx += 512) {
                    const int r = idx / (C / 2), k = (idx - r * (C / 2)) * 2;
                    float2 g = make_float2(0.f, 0.f);
                    if (t0 + r < p.BL) g = *(const float2*)(TH + (size_t)(nbase + r) * C + k);
                    *(float2*)(G + (size_t)r * ldg + k) = g;
                }
            }
            __syncthreads();
            if (fits && l + 1 < L) gfetch(l + 1);
            if (active) {
                const int nts[2] = {nt0, nt1};
                wave_gemm<MT, 2>(acc, G, ldg, p.wp + p.ws_f4 + (size_t)l * NCG * NTS * 64, NTS, nts, C, lane);
            }
        }
        if (active) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = j ? nt1 : nt0;
                if (j && nt1 == nt0) break;
                const int c = 16 * nt + (lane & 15);
                const float bs = pb == 0 ? bpre[0][j] : p.bp[p.bias_s + c];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        const float v = acc[mt][j][i] + bs;
                        if (t0 + r < p.BL) p.S0[((size_t)b * p.BL + t0 + r) * S + c] = v;
                        St[(size_t)r * lds + c] = v > 0.f ? v : 0.f;
                    }
            }
        }
        __syncthreads();   // also protects the G buffers (alias of Yt) before the next batch / phase
    }
    // ---------- post 1x1 #1: y0 = W1 relu(s0) + b1
    for (int pb = 0; pb < npairs; pb += 8) {
        const int np = pb + wave;
        if (np < npairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTS) ? 2 * np + 1 : 2 * np;
            f32x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<MT, 2>(acc, St, lds, p.wp + p.p1_f4, NTS, nts, S, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = j ? nt1 : nt0;
                if (j && nt1 == nt0) break;
                const int c = 16 * nt + (lane & 15);
                const float bb = pb == 0 ? bpre[1][j] : p.bp[p.bias_p1 + c];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        const float v = acc[mt][j][i] + bb;
                        if (t0 + r < p.BL) p.Y0[((size_t)b * p.BL + t0 + r) * S + c] = v;
                        Yt[(size_t)r * lds + c] = v > 0.f ? v : 0.f;
                    }
            }
        }
    }
    __syncthreads();
    // ---------- post 1x1 #2: logits = W2 relu(y0) + b2
    const int qpairs = (NTQ + 1) / 2;
    for (int pb = 0; pb < qpairs; pb += 8) {
        const int np = pb + wave;
        if (np < qpairs) {
            const int nt0 = 2 * np, nt1 = (2 * np + 1 < NTQ) ? 2 * np + 1 : 2 * np;
            f32x4 acc[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
            const int nts[2] = {nt0, nt1};
            wave_gemm<MT, 2>(acc, Yt, lds, p.wp + p.p2_f4, NTQ, nts, S, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nt = j ? nt1 : nt0;
                if (j && nt1 == nt0) break;
                const int c = 16 * nt + (lane & 15);
                const float bb = pb == 0 ? bpre[2][j] : p.bp[p.bias_p2 + c];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 16 * mt + 4 * (lane >> 4) + i;
                        const float v = acc[mt][j][i] + bb;
                        if (p.logits && t0 + r < p.BL) p.logits[((size_t)b * p.BL + t0 + r) * Q + c] = v;
                        if (p.ce_tgt) St[(size_t)r * lds + c] = v;          // St (relu(s0)) is dead since the first 1x1; Q <= S checked by the host
                    }
            }
        }
    }
    if (!p.ce_tgt) return;
    // ---------- fused torch.nn.CrossEntropyLoss() (mean) and its gradient on the tile's rows while the logits are in LDS
    // (same arithmetic as k_ce: one wave per row, four classes per lane; reference qpnet_train.py:430,526-528)
    __syncthreads();
    {
        const int64_t rows = (int64_t)p.B * p.BL;
        const float inv = 1.0f / (float)rows;
        double lsum = 0.0;
        for (int r = wave; r < TM; r += 8) {
            if (t0 + r >= p.BL) break;
            const float* lg = St + (size_t)r * lds;
            const int64_t row = (int64_t)b * p.BL + t0 + r;
            int64_t tg = p.ce_tgt[(size_t)b * p.ce_stride + (p.ce_stride - p.BL) + t0 + r];
            if (tg < 0 || tg >= Q) { if (lane == 0) atomicOr(p.status, 2); tg = tg < 0 ? 0 : Q - 1; }
            float m = -INFINITY;
            for (int q = lane * 4; q < Q; q += 256) { const float2 v01 = *(const float2*)(lg + q), v23 = *(const float2*)(lg + q + 2); const float4 v = make_float4(v01.x, v01.y, v23.x, v23.y);   /* rows are 8-byte aligned in LDS (ld = 2 mod 32) */ m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w)); }
            m = tr_wave_max(m);
            float se = 0.f;
            for (int q = lane * 4; q < Q; q += 256) { const float2 v01 = *(const float2*)(lg + q), v23 = *(const float2*)(lg + q + 2); const float4 v = make_float4(v01.x, v01.y, v23.x, v23.y);   /* rows are 8-byte aligned in LDS (ld = 2 mod 32) */ se += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m)); }
            se = tr_wave_sum(se);
            const float lse = logf(se) + m;
            if (p.ce_dlogits) for (int q = lane * 4; q < Q; q += 256) {
                const float2 v01 = *(const float2*)(lg + q), v23 = *(const float2*)(lg + q + 2); const float4 v = make_float4(v01.x, v01.y, v23.x, v23.y);   /* rows are 8-byte aligned in LDS (ld = 2 mod 32) */
                float4 gq = make_float4(__expf(v.x - lse), __expf(v.y - lse), __expf(v.z - lse), __expf(v.w - lse));
                const int dq = (int)tg - q;
                if (dq == 0) gq.x -= 1.0f; else if (dq == 1) gq.y -= 1.0f; else if (dq == 2) gq.z -= 1.0f; else if (dq == 3) gq.w -= 1.0f;
                *(float4*)(p.ce_dlogits + (size_t)row * Q + q) = make_float4(gq.x * inv, gq.y * inv, gq.z * inv, gq.w * inv);
            }
            lsum += (double)(lse - lg[tg]);
        }
        double* part = (double*)Yt;                 // relu(y0) is dead: every wave passed the barrier above after its last read
        if (lane == 0) part[wave] = lsum;
        __syncthreads();
        if (tid == 0) {
            double sacc = 0.0;
            for (int w8 = 0; w8 < 8; ++w8) sacc += part[w8];
            atomicAdd(p.ce_loss + (blockIdx.x & 63), sacc / (double)rows);
        }
    }
}

// ------------------------------------------------------------------------------------------------ post-net, wide tiles (S = Q = 256): train_post.h
template <int MT>
__global__ __launch_bounds__(512) void k_post_fwd_w(TrainParams p) {
    extern __shared__ float sm[];
    unsigned long long mS, mY; float4 bq[2];
    post_fwd_w_tile<MT, false>(p, sm, mS, mY, bq);
}

// mean cross entropy + its gradient, one wave per row, rpw rows per wave (reference qpnet_train.py:430,526-528)
__global__ __launch_bounds__(256) void k_ce(const float* __restrict__ logits, const int64_t* __restrict__ tgt, int64_t tgt_stride,
                                            int BL, int Q, int64_t rows, float* __restrict__ dlogits, double* __restrict__ loss, int rpw, int* __restrict__ status) {
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / (float)rows;
    double lsum = 0.0;
    for (int it = 0; it < rpw; ++it) {
        const int64_t row = ((int64_t)blockIdx.x * rpw + it) * 4 + wave;
        if (row >= rows) break;
        const float* lg = logits + (size_t)row * Q;
        const int64_t b = row / BL, t = row - b * BL;
        int64_t tg = tgt[(size_t)b * tgt_stride + (tgt_stride - BL) + t];
        if (tg < 0 || tg >= Q) { if (lane == 0) atomicOr(status, 2); tg = tg < 0 ? 0 : Q - 1; }     // reference: assert max(batch_t) < n_quantize
        float m = -INFINITY;
        const bool vec = (Q & 255) == 0;             // 4 contiguous logits per lane per pass (16-byte accesses)
        if (vec) for (int q = lane * 4; q < Q; q += 256) { const float4 v = *(const float4*)(lg + q); m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w)); }
        else for (int q = lane; q < Q; q += 64) m = fmaxf(m, lg[q]);
        m = tr_wave_max(m);
        float se = 0.f;
        if (vec) for (int q = lane * 4; q < Q; q += 256) { const float4 v = *(const float4*)(lg + q); se += (__expf(v.x - m) + __expf(v.y - m)) + (__expf(v.z - m) + __expf(v.w - m)); }
        else for (int q = lane; q < Q; q += 64) se += expf(lg[q] - m);
        se = tr_wave_sum(se);
        const float lse = logf(se) + m;
        if (dlogits) {
            if (vec) for (int q = lane * 4; q < Q; q += 256) {
                const float4 v = *(const float4*)(lg + q);
                float4 g = make_float4(__expf(v.x - lse), __expf(v.y - lse), __expf(v.z - lse), __expf(v.w - lse));
                const int dq = (int)tg - q;
                if (dq == 0) g.x -= 1.0f; else if (dq == 1) g.y -= 1.0f; else if (dq == 2) g.z -= 1.0f; else if (dq == 3) g.w -= 1.0f;
                *(float4*)(dlogits + (size_t)row * Q + q) = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
            } else for (int q = lane; q < Q; q += 64) {
                const float pr = expf(lg[q] - lse);
                dlogits[(size_t)row * Q + q] = (pr - (q == tg ? 1.0f : 0.0f)) * inv;
            }
        }
        lsum += (double)(lse - lg[tg]);
    }
    if (lane == 0) part[wave] = lsum;
    __syncthreads();
    // 64 accumulator slots (summed by the reader): one double atomic per workgroup on ONE address serialises at the memory side
    if (threadIdx.x == 0) atomicAdd(loss + (blockIdx.x & 63), (part[0] + part[1] + part[2] + part[3]) / (double)rows);
}

// ------------------------------------------------------------------ host launchers (called from train_host.hip)
void qpn_launch_prep(const TrainParams& p, const AuxGeom& ag, hipStream_t stream) {
    const long items = (long)p.N1 * (p.C / 4 + (p.hoist ? 0 : p.Ap / 4) + p.L);
    const int blocks = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
    // aux hoist: the frame-rate projections PA are extra workgroups of this launch (they need k_refresh's packed biases: the launch before)
    const int naux = p.hoist ? (p.nfr + 1 + AUXP_FPB - 1) / AUXP_FPB * p.L : 0;
    const size_t lds = p.hoist ? (size_t)(2 * p.C * p.A + p.A * AUXP_FPB + 8) * sizeof(float) : 0;
    hipLaunchKernelGGL(k_train_prep, dim3(blocks + naux, p.B), dim3(256), lds, stream, p, ag, blocks);
}

bool qpn_stack_fwd_fits(const TrainParams& p);
int qpn_launch_stack_fwd(const TrainParams& p, const StackQ& q, const TrainKnobs& k, hipStream_t stream);
bool qpn_stack_fwd_t_fits(const TrainParams& p);
int qpn_launch_stack_fwd_t(const TrainParams& p, const StackQ& q, const TrainKnobs& k, hipStream_t stream);

void qpn_launch_post_fb(const TrainParams& p, const TrainBwd& bw, hipStream_t stream);      // train_bwd.hip: k_post_fb_w

// fuse_post_bwd: the post-net's backward runs inside this launch sequence too (the caller has checked the geometry: the wide tiles, cross entropy fused)
int qpn_launch_fwd(const TrainParams& p, const TrainKnobs& k, const AuxGeom& ag, const StackQ* sq, hipStream_t stream, const TrainBwd* fuse_post_bwd) {
    const int C = p.C, S = p.S;
    qpn_launch_prep(p, ag, stream);
    qpn_prof_mark(PG_PREP, stream);
    // 16-row tiles everywhere (twice the workgroups of 32-row tiles, all co-resident: measured 8-15 % faster for the layer and post-net kernels)
    const size_t lds_layer = (size_t)16 * (tr_lda(p.Ktp) + tr_lda(C)) * sizeof(float);
    const size_t lds_post = (size_t)16 * (tr_lda(S) + (tr_lda(S) > 2 * tr_lda(C) ? tr_lda(S) : 2 * tr_lda(C))) * sizeof(float);
    if (lds_layer > 160 * 1024 || lds_post > 160 * 1024) {
        qpn_set_error("training kernels: tiles do not fit the 160 KiB LDS for n_resch=%d n_skipch=%d (n_resch <= 128 supported)", C, S);
        return QPN_EINVAL;
    }
    {
        const int flags0 = k.xcd_swizzle ? 2 : 0;
        // persistent register-resident form (n_resch 64, K = 176): 2 workgroups per CU; QPN_LAYER_PERSIST=0 keeps the tile-per-workgroup launches
        const bool persist = C == 64 && (p.Ktp == 176 || p.hoist) && p.N1 < (1 << 24) && k.persist_fwd;      // (N1 < 2^24: 32-bit element offsets of one batch item)
        if (p.hoist && !persist) { qpn_set_error("internal: the frame-rate aux term needs the register-resident layer kernels"); return QPN_EINVAL; }
        // the whole stack as ONE persistent launch over a (layer, tile) work queue (train_stack.hip); QPN_STACK_QUEUE=0 keeps a launch per layer
        const bool stack_q = persist && k.stack_q_fwd && sq && sq->flags && p.qctl && qpn_stack_fwd_fits(p);
        if (stack_q) {      // the transposed-product tile body where that form exists (train_stackw.hip), else the LDS-staged one
            const int rcq = (k.stack_wave_fwd && qpn_stack_fwd_t_fits(p)) ? qpn_launch_stack_fwd_t(p, *sq, k, stream) : qpn_launch_stack_fwd(p, *sq, k, stream);
            if (rcq) return rcq;
        }
        for (int l = 0; l < p.L && !stack_q; ++l) {
            const int rows = p.N1 - p.layers[l].s_out;
            const int tiles = (rows + 15) / 16;
            if (persist) {
                int G = qpn_num_cus() * 2; if (G > tiles) G = tiles;
                if (G > 1024) G = 1024;                            // (scratch_rows holds a pair of rows for 1024 workgroups per batch item)
                const size_t ldsp = (size_t)(32 * tr_lda(p.Ktp) + 4 * 16 * tr_lda(64)) * sizeof(float);
                const bool last = l == p.L - 1;
                if (p.hoist) {
                    if (last) hipLaunchKernelGGL((k_layer_fwd_p<8, true>), dim3(G, p.B), dim3(256), ldsp, stream, p, l, flags0, tiles, p.scratch_rows);
                    else hipLaunchKernelGGL((k_layer_fwd_p<8, false>), dim3(G, p.B), dim3(256), ldsp, stream, p, l, flags0, tiles, p.scratch_rows);
                } else {
                    if (last) hipLaunchKernelGGL((k_layer_fwd_p<11, true>), dim3(G, p.B), dim3(256), ldsp, stream, p, l, flags0, tiles, p.scratch_rows);
                    else hipLaunchKernelGGL((k_layer_fwd_p<11, false>), dim3(G, p.B), dim3(256), ldsp, stream, p, l, flags0, tiles, p.scratch_rows);
                }
            } else {
                if (lds_layer > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_layer_fwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_layer);
                hipLaunchKernelGGL((k_layer_fwd<1>), dim3(tiles, p.B), dim3(256), lds_layer, stream, p, l, (l == p.L - 1 ? 1 : 0) | flags0);
            }
        }
    }
    qpn_prof_mark(PG_LAYER_FWD, stream);
    // wide post-net tiles (S = Q = 256, n_resch 64): 16 * MT rows per workgroup, MT chosen so that the chunk is one round of workgroups
    const bool post_wide = S == 256 && p.Q == 256 && C == 64 && k.post_wide;
    if (post_wide && fuse_post_bwd) qpn_launch_post_fb(p, *fuse_post_bwd, stream);
    else if (post_wide) {
        constexpr int MTW = 5;
        const size_t ldsw = (size_t)16 * MTW * (tr_lda(256) + 2 * tr_lda(64)) * sizeof(float);
        QPN_HIP(hipFuncSetAttribute((const void*)k_post_fwd_w<MTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
        hipLaunchKernelGGL((k_post_fwd_w<MTW>), dim3((p.BL + 16 * MTW - 1) / (16 * MTW), p.B), dim3(512), ldsw, stream, p);
    } else {
        if (lds_post > 48 * 1024) (void)hipFuncSetAttribute((const void*)k_post_fwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_post);
        hipLaunchKernelGGL((k_post_fwd<1>), dim3((p.BL + 15) / 16, p.B), dim3(512), lds_post, stream, p);
    }
    qpn_prof_mark(PG_POST_FWD, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}

int qpn_launch_ce(const float* logits, const int64_t* tgt, int64_t tgt_stride, int B, int BL, int Q, float* dlogits, double* loss, int* status, bool loss_cleared, hipStream_t stream) {
    const int64_t rows = (int64_t)B * BL;
    // (the loss accumulator was cleared by this step's k_refresh; a CE call without a forward in front clears it itself)
    if (!loss_cleared) QPN_HIP(hipMemsetAsync(loss, 0, 64 * sizeof(double), stream));
    const int rpw = 4;                // 16 rows per workgroup: ~1250 workgroups for a 20 k-row chunk (64 rows per workgroup left most CUs with one)
    hipLaunchKernelGGL(k_ce, dim3((unsigned)((rows + 4 * rpw - 1) / (4 * rpw))), dim3(256), 0, stream, logits, tgt, tgt_stride, BL, Q, rows, dlogits, loss, rpw, status);
    qpn_prof_mark(PG_CE, stream);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
