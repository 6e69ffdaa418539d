// train_post.h -- the wide post-net tiles (S = Q = 256, n_resch 64): forward and backward of 16 * MT rows as device functions, so that the fused training step can run both
// in ONE kernel (k_post_fb_w, train_bwd.hip) next to the separate k_post_fwd_w (train_fwd.hip) / k_post_bwd_w (train_bwd.hip) the autograd path uses.
#pragma once
#include "train_common.h"

// dev aid (-DQPN_POST_STAMPS): s_memtime of thread 0 of workgroup `PS_WG` at the stage boundaries of the post-net forward kernels, into the stack queues' control
// words [600 + 16 * slot + i] (read back with qpn_train_stack_stats; tools/post_stamps.py)
#ifdef QPN_POST_STAMPS
#define POST_STAMP(slot, i) do { if (p.qctl && ps_on && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); p.qctl[600 + 16 * (slot) + (i)] = (unsigned)t_; } } while (0)
#else
#define POST_STAMP(slot, i) do { } while (0)
#endif
// ------------------------------------------------------------------------------------------------ post-net, wide tiles (S = Q = 256)
// k_post_fwd streams the post-net's 1 MB of weight fragments from L2 once per 16-row tile: 1250 tiles = 1.25 GB per launch, and a
// 16-deep step's two fragment loads feed only 8 MFMAs (256 cycles) -- the launch runs at the L2's pace, 0.5 of the matrix-core rate.
// Here a workgroup takes 16 * MT rows (MT = 5: 80 rows, so a 20 000-row chunk is 250 workgroups = ONE round on 256 CUs): the same
// two loads feed 8 * MT MFMAs, weight traffic drops MT-fold, and the K = L*C skip sum runs as one continuous fragment stream with
// the gate rows of layer l+1 landing in the other LDS buffer under layer l's contraction.  One [16 MT][S] LDS tile is reused IN
// PLACE by the three stages (every wave holds its outputs in registers across the barrier that ends the reads), and relu(s0) /
// relu(y0) leave from it as whole rows -- the backward only needs their sign (the ReLU mask) and the rectified values (weight
// gradients), so the rectified values are what is stored.   (reference: _postprocess, src/nets/qpnet.py:566-571, 283-309)
// FUSED: the tile's backward follows in the same kernel (k_post_fb_w): the ReLU signs of S0 / Y0 stay in registers as bits (mS, mY: bit (4 mt + i) * 2 + j of
// the lane's (row tile mt, row i, column tile j) element), dL/dlogits is ALSO left in the LDS tile (rows past the chunk end as zeros), and the weight-fragment
// registers bq hold the first step of the backward's first contraction on return.
template <int MT, bool FUSED>
__device__ __forceinline__ void post_fwd_w_tile(const TrainParams& p, float* sm, unsigned long long& mS, unsigned long long& mY, float4 (&bq)[2]) {
    constexpr int TM = 16 * MT, S = 256, Q = 256, C = 64, NTS = S / 16;
    constexpr int lds = ((S + 29) / 32) * 32 + 2, ldg = ((C + 29) / 32) * 32 + 2;
    constexpr int NG = (TM * (C / 2) + 511) / 512;               // float2 pairs of a [TM][C] gate tile per thread
    float* T = sm;                                                // [TM][lds]; gate staging: T + TM*lds + {0, TM*ldg}
    float* Gb = sm + TM * lds;
    const int L = p.L, b = blockIdx.y, t0 = blockIdx.x * TM;
    const int nbase = p.N1 - p.BL + t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt0 = 2 * wave;                                     // this wave's two column tiles (of 16) in every stage
    const int c0 = 16 * nt0 + (lane & 15), c1 = c0 + 16;
    const float bs0 = p.bp[p.bias_s + c0], bs1 = p.bp[p.bias_s + c1], bp10 = p.bp[p.bias_p1 + c0], bp11 = p.bp[p.bias_p1 + c1],
                bp20 = p.bp[p.bias_p2 + c0], bp21 = p.bp[p.bias_p2 + c1];
    const float4* Ws = p.wp + p.ws_f4; const float4* P1 = p.wp + p.p1_f4; const float4* P2 = p.wp + p.p2_f4;
    bq[0] = Ws[(size_t)nt0 * 64 + lane]; bq[1] = Ws[(size_t)(nt0 + 1) * 64 + lane];
    // ---- gate rows of one layer: thread -> NG (row, column pair) items of the layer's gate product (p.TH holds sigma * tanh: train_common.h)
    float2 gt[NG];
    auto gfetch = [&](int l) {
        const float* TH = p.TH + ((size_t)(l * p.B + b) * p.N1) * C;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int idx = tid + k * 512, r = idx / (C / 2), kk = (idx - r * (C / 2)) * 2;
            const bool ok = idx < TM * (C / 2) && t0 + r < p.BL;
            const size_t o = ok ? (size_t)(nbase + r) * C + kk : 0;
            gt[k] = *(const float2*)(TH + o);
        }
    };
    auto gstore = [&](float* G) {
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int idx = tid + k * 512, r = idx / (C / 2), kk = (idx - r * (C / 2)) * 2;
            if (idx < TM * (C / 2)) *(float2*)(G + (size_t)r * ldg + kk) = t0 + r < p.BL ? gt[k] : make_float2(0.f, 0.f);
        }
    };
    // a finished stage: rectified outputs into T (in place: every wave has passed the barrier that ends the stage's reads) ...
    auto put = [&](const f32x4 (&acc)[MT][2], float bias0, float bias1, bool relu, unsigned long long* signs) {
        unsigned long long m = 0ull;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i;
                float v0 = acc[mt][0][i] + bias0, v1 = acc[mt][1][i] + bias1;
                if (relu) {
                    if (FUSED) m |= (unsigned long long)(v0 > 0.f ? 1u : 0u) << ((4 * mt + i) * 2) | (unsigned long long)(v1 > 0.f ? 1u : 0u) << ((4 * mt + i) * 2 + 1);
                    v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f;
                }
                T[(size_t)r * lds + c0] = v0; T[(size_t)r * lds + c1] = v1;
            }
        if (FUSED && signs) *signs = m;
    };
    // ... and from there to a [BL][256] array as whole 1 KB rows
    auto rows_out = [&](float* dst) {
        for (int idx = tid; idx < TM * (S / 2); idx += 512) {
            const int r = idx / (S / 2), kk = (idx - r * (S / 2)) * 2;
            if (t0 + r < p.BL) *(float2*)(dst + ((size_t)b * p.BL + t0 + r) * S + kk) = *(const float2*)(T + (size_t)r * lds + kk);
        }
    };
    const bool ps_on = blockIdx.x == 5 && blockIdx.y == 0; (void)ps_on;
    POST_STAMP(0, 0);
    f32x4 acc[MT][2];
#define POSTW_ZERO() _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4){0, 0, 0, 0}; acc[mt][1] = (f32x4){0, 0, 0, 0}; }
    // ---------- skip sum: one K = L*C contraction, layer l's gates in Gb[l & 1]
    POSTW_ZERO();
    gfetch(0);
    for (int l = 0; l < L; ++l) {
        float* G = Gb + (l & 1) * TM * ldg;
        gstore(G);
        TR_LDS_BARRIER();                                          // G complete; the other buffer's readers (layer l-1) are done
        if (l + 1 < L) gfetch(l + 1);
        post_gemm<MT>(acc, G, ldg, Ws + (size_t)l * (C / 16) * NTS * 64, NTS, nt0, C / 16, lane, bq,
                      l + 1 < L ? Ws + ((size_t)(l + 1) * (C / 16) * NTS + nt0) * 64 + lane : P1 + (size_t)nt0 * 64 + lane);
    }
    POST_STAMP(0, 1);
    put(acc, bs0, bs1, true, &mS);
    TR_LDS_BARRIER();
    POST_STAMP(0, 2);
    rows_out(p.S0);
    POST_STAMP(0, 3);                                               // relu(s0): sign = the backward's mask, value = the weight gradient's operand
    // ---------- post 1x1 #1
    POSTW_ZERO();
    post_gemm<MT>(acc, T, lds, P1, NTS, nt0, S / 16, lane, bq, P2 + (size_t)nt0 * 64 + lane);
    POST_STAMP(0, 4);
    TR_LDS_BARRIER();                                              // every wave (and rows_out) is done reading T
    put(acc, bp10, bp11, true, &mY);
    TR_LDS_BARRIER();
    POST_STAMP(0, 5);
    rows_out(p.Y0);
    POST_STAMP(0, 6);
    // the targets of this wave's cross-entropy rows, asked for a whole contraction ahead (one dependent global load per row inside the row loop was most of that phase)
    constexpr int NCE = TM / 8;
    int64_t tgv[NCE];
#pragma unroll
    for (int k = 0; k < NCE; ++k) {
        const int r = wave + 8 * k;
        tgv[k] = (p.ce_tgt && t0 + r < p.BL) ? p.ce_tgt[(size_t)b * p.ce_stride + (p.ce_stride - p.BL) + t0 + r] : 0;
    }
    // ---------- post 1x1 #2
    POSTW_ZERO();
    post_gemm<MT>(acc, T, lds, P2, Q / 16, nt0, S / 16, lane, bq, (FUSED ? p.wp + p.p2t_f4 : P2) + (size_t)nt0 * 64 + lane);
    POST_STAMP(0, 7);
    TR_LDS_BARRIER();
    put(acc, bp20, bp21, false, nullptr);
    TR_LDS_BARRIER();
    POST_STAMP(0, 8);
#undef POSTW_ZERO
    if (p.logits)
        for (int idx = tid; idx < TM * (Q / 2); idx += 512) {
            const int r = idx / (Q / 2), kk = (idx - r * (Q / 2)) * 2;
            if (t0 + r < p.BL) *(float2*)(p.logits + ((size_t)b * p.BL + t0 + r) * Q + kk) = *(const float2*)(T + (size_t)r * lds + kk);
        }
    if (!p.ce_tgt) return;
    // ---------- fused torch.nn.CrossEntropyLoss() (mean) and its gradient while the logits are in LDS (same arithmetic as k_ce)
    {
        const int64_t rows = (int64_t)p.B * p.BL;
        const float inv = 1.0f / (float)rows;
        double lsum = 0.0;
        // every row's logits (and the target's logit) are read first, THEN the ten rows' reductions run as independent chains (one row at a time, each chain -- two wave
        // reductions, a logarithm, eight exponentials -- waited for the row before it: the LDS stores of one row and the loads of the next could not be reordered)
        float lv[NCE][4], ltg[NCE];
        int tgi[NCE];
#pragma unroll
        for (int k = 0; k < NCE; ++k) {
            const int r = wave + 8 * k;
            const float* lg = T + (size_t)r * lds;
            int64_t tg = tgv[k];
            if (t0 + r < p.BL && (tg < 0 || tg >= Q)) { if (lane == 0) atomicOr(p.status, 2); tg = tg < 0 ? 0 : Q - 1; }
            tgi[k] = (int)tg;
            const float2 v01 = *(const float2*)(lg + lane * 4), v23 = *(const float2*)(lg + lane * 4 + 2);
            lv[k][0] = v01.x; lv[k][1] = v01.y; lv[k][2] = v23.x; lv[k][3] = v23.y;
            ltg[k] = lg[tgi[k]];
        }
#pragma unroll
        for (int k = 0; k < NCE; ++k) {
            const int r = wave + 8 * k;
            const bool in = t0 + r < p.BL;
            const int64_t row = (int64_t)b * p.BL + t0 + r;
            const int q = lane * 4;
            // (the wave reductions are k_ce's: DPP inside the 16-lane rows + v_readlane across them, same order, bit-identical dL/dlogits; as six
            //  dependent ds_bpermute each they made the cross entropy 25 k of the tile's 232 k cycles: profiles/r05_post_fwd_stamps.txt)
            const float m = tr_wave_max(fmaxf(fmaxf(lv[k][0], lv[k][1]), fmaxf(lv[k][2], lv[k][3])));
            const float se = tr_wave_sum((__expf(lv[k][0] - m) + __expf(lv[k][1] - m)) + (__expf(lv[k][2] - m) + __expf(lv[k][3] - m)));
            const float lse = logf(se) + m;
            if (p.ce_dlogits && in) {
                float4 gq = make_float4(__expf(lv[k][0] - lse), __expf(lv[k][1] - lse), __expf(lv[k][2] - lse), __expf(lv[k][3] - lse));
                const int dq = tgi[k] - q;
                if (dq == 0) gq.x -= 1.0f; else if (dq == 1) gq.y -= 1.0f; else if (dq == 2) gq.z -= 1.0f; else if (dq == 3) gq.w -= 1.0f;
                *(float4*)(p.ce_dlogits + (size_t)row * Q + q) = make_float4(gq.x * inv, gq.y * inv, gq.z * inv, gq.w * inv);
                if (FUSED) {
                    float* dl = T + (size_t)r * lds + q;
                    *(float2*)dl = make_float2(gq.x * inv, gq.y * inv); *(float2*)(dl + 2) = make_float2(gq.z * inv, gq.w * inv);
                }
            }
            if (in) lsum += (double)(lse - ltg[k]);
        }
        if (FUSED) for (int r = wave; r < TM; r += 8) if (t0 + r >= p.BL) { float* dl = T + (size_t)r * lds + lane * 4; *(float2*)dl = make_float2(0.f, 0.f); *(float2*)(dl + 2) = make_float2(0.f, 0.f); }
        double* part = (double*)Gb;                               // the gate staging is dead
        if (lane == 0) part[wave] = lsum;
        __syncthreads();
        if (tid == 0) {
            double sacc = 0.0;
            for (int w8 = 0; w8 < 8; ++w8) sacc += part[w8];
            atomicAdd(p.ce_loss + (blockIdx.x & 63), sacc / (double)rows);
        }
    }
    POST_STAMP(0, 9);
}

// ------------------------------------------------------------------------------------------------ backward of the same tile
// Counterpart of post_fwd_w_tile: the transposed post-net weights stream from L2 once per 16 * MT rows, one [16 MT][256] LDS tile reused in place by the stages, every output
// array written as whole rows.  Separate kernel: the ReLU masks are the signs of the rectified activations the forward stored (p.Y0, p.S0), requested ahead of the contraction they follow.
// FUSED: behind post_fwd_w_tile<MT, true> in the same kernel: dL/dlogits is in the LDS tile, the ReLU signs are the bits mS / mY, bq holds the first fragments.
template <int MT, bool FUSED>
__device__ __forceinline__ void post_bwd_w_tile(const TrainParams& p, const TrainBwd& bw, float* sm, unsigned long long mS, unsigned long long mY, float4 (&bq)[2]) {
    constexpr int TM = 16 * MT, S = 256, Q = 256;
    constexpr int lds = ((S + 29) / 32) * 32 + 2;
    float* T = sm;
    const int LC = p.LC, b = blockIdx.y, t0 = blockIdx.x * TM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt0 = 2 * wave, NTS = S / 16;
    const int c0 = 16 * nt0 + (lane & 15), c1 = c0 + 16;
    const float4* P2t = p.wp + p.p2t_f4; const float4* P1t = p.wp + p.p1t_f4; const float4* Wst = p.wp + p.wst_f4;
    const int NTL = LC / 16;
    if (!FUSED) { bq[0] = P2t[(size_t)nt0 * 64 + lane]; bq[1] = P2t[(size_t)(nt0 + 1) * 64 + lane]; }
    const bool ps_on = blockIdx.x == 5 && blockIdx.y == 0; (void)ps_on;
    POST_STAMP(1, 0);
    float mk[MT][4][2];
    auto mask_load = [&](const float* src) {                      // rows past the chunk end read the arena's padding rows (masked at the store)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const size_t o = ((size_t)b * p.BL + t0 + 16 * mt + 4 * (lane >> 4) + i) * S;
                mk[mt][i][0] = src[o + c0]; mk[mt][i][1] = src[o + c1];
            }
    };
    unsigned long long mbits = mY;
    auto put = [&](const f32x4 (&acc)[MT][2], bool masked) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 16 * mt + 4 * (lane >> 4) + i;
                const bool in = t0 + r < p.BL;
                const bool k0 = FUSED ? ((mbits >> ((4 * mt + i) * 2)) & 1ull) != 0ull : mk[mt][i][0] > 0.f;
                const bool k1 = FUSED ? ((mbits >> ((4 * mt + i) * 2 + 1)) & 1ull) != 0ull : mk[mt][i][1] > 0.f;
                T[(size_t)r * lds + c0] = (in && (!masked || k0)) ? acc[mt][0][i] : 0.f;
                T[(size_t)r * lds + c1] = (in && (!masked || k1)) ? acc[mt][1][i] : 0.f;
            }
    };
    auto rows_out = [&](float* dst, size_t ld, int col0) {       // T -> rows t0.. of dst[B][BL][ld], columns col0 .. col0 + 255
        for (int idx = tid; idx < TM * (S / 2); idx += 512) {
            const int r = idx / (S / 2), kk = (idx - r * (S / 2)) * 2;
            if (t0 + r < p.BL) *(float2*)(dst + ((size_t)b * p.BL + t0 + r) * ld + col0 + kk) = *(const float2*)(T + (size_t)r * lds + kk);
        }
    };
    if (!FUSED) {
        mask_load(p.Y0);
        for (int idx = tid; idx < TM * (Q / 2); idx += 512) {      // stage the dlogits rows
            const int r = idx / (Q / 2), kk = (idx - r * (Q / 2)) * 2;
            float2 v = make_float2(0.f, 0.f);
            if (t0 + r < p.BL) v = *(const float2*)(bw.dlogits + ((size_t)b * p.BL + t0 + r) * Q + kk);
            *(float2*)(T + (size_t)r * lds + kk) = v;
        }
        TR_LDS_BARRIER();
    }
    f32x4 acc[MT][2], acc2[MT][2];
#define POSTW_ZERO(a) _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { a[mt][0] = (f32x4){0, 0, 0, 0}; a[mt][1] = (f32x4){0, 0, 0, 0}; }
    // ---------- dY0 = (dlogits . W2) * (Y0 > 0)
    POSTW_ZERO(acc);
    POST_STAMP(1, 1);
    post_gemm<MT>(acc, T, lds, P2t, NTS, nt0, Q / 16, lane, bq, P1t + (size_t)nt0 * 64 + lane);
    POST_STAMP(1, 2);
    TR_LDS_BARRIER();
    put(acc, true);
    if (FUSED) mbits = mS; else mask_load(p.S0);                  // (the Y0 signs are consumed: same registers)
    TR_LDS_BARRIER();
    POST_STAMP(1, 3);
    rows_out(bw.DY0, S, 0);
    POST_STAMP(1, 4);
    // ---------- dS0 = (dY0 . W1post) * (S0 > 0)
    POSTW_ZERO(acc);
    post_gemm<MT>(acc, T, lds, P1t, NTS, nt0, S / 16, lane, bq, Wst + (size_t)nt0 * 64 + lane);
    POST_STAMP(1, 5);
    TR_LDS_BARRIER();
    put(acc, true);
    TR_LDS_BARRIER();
    POST_STAMP(1, 6);
    rows_out(bw.DS0, S, 0);
    POST_STAMP(1, 7);
    // ---------- DGS[t][l*C + c] = sum_s dS0[t][s] Ws_l[s][c]: L*C columns in passes of 2 x 256, both held in registers and stored from there (through the LDS tile,
    //            one half after the other, the kernel's last 27 k cycles were two barrier-separated round trips with nothing to overlap them: 1460 -> 1465 steps/s)
    for (int cb = 0; cb < LC; cb += 512) {
        const int nta = cb / 16 + nt0, ntb = nta + NTS;
        const bool two = cb + 256 < LC;
        POSTW_ZERO(acc); POSTW_ZERO(acc2);
        post_gemm<MT>(acc, T, lds, Wst, NTL, nta, S / 16, lane, bq, Wst + (size_t)(two ? ntb : nta) * 64 + lane);
        if (two) post_gemm<MT>(acc2, T, lds, Wst, NTL, ntb, S / 16, lane, bq, Wst + (size_t)(cb + 512 < LC ? nta + 2 * NTS : nta) * 64 + lane);
        POST_STAMP(1, 8);
        // the kernel's last outputs leave straight from the accumulators (nothing is left to overlap an LDS round trip with)
        auto direct = [&](const f32x4 (&a)[MT][2], int col0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * mt + 4 * (lane >> 4) + i;
                    if (t0 + r < p.BL) { float* d = bw.DGS + ((size_t)b * p.BL + t0 + r) * LC + col0; d[c0] = a[mt][0][i]; d[c1] = a[mt][1][i]; }
                }
        };
        direct(acc, cb);
        if (two) direct(acc2, cb + 256);
        if (cb + 512 < LC) TR_LDS_BARRIER();
    }
#undef POSTW_ZERO
    POST_STAMP(1, 9);
}
