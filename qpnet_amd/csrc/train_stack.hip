// train_stack.hip -- the residual stack of a training step (n_resch = 64) as ONE persistent launch per direction.
//
// k_layer_fwd_p / k_layer_bwd_p (train_fwd.hip, train_bwd.hip) run a layer per launch: 16 launches of ~11 us of matrix work each,
// every one paying a kernel boundary, a fill and a drain.  Here the (layer, batch item, 16-row tile) triples of the
// whole stack are the POSITIONS of one work queue, in layer-major order; 2 workgroups per CU pull positions with a returning
// atomic add and run the same tile body as the per-layer kernels.  What lets the layers overlap is that the stack is causal and
// its dependencies are local in time: tile (l, t) needs the rows of layer l-1 that its own rows and their pitch-dependent taps
// touch (reference src/nets/qpnet.py:271-306: xC / xP of layer l are slices / gathers of layer l-1's output) -- at most
// reach_l / 16 + 2 producer tiles, all of them ~T - reach_l / 16 positions back in the queue, i.e. ~2.4 rounds of the 512
// resident workgroups.  So:
//   * one flag word per position: the producer stores the tile's rows WRITE-THROUGH (sc1), every storing wave drains them
//     (in-order vmcnt: the wave's later flag loads return behind them), the workgroup's barrier, then ONE lane stores
//     flag[pos] = epoch (sc1).  Consumers read the flags of their producer range with ONE sc1 load per wave and the rows with
//     sc1 buffer loads (CDNA4 guide, Guideline 16 R1: no release / acquire fence, correct for any placement of the workgroups).
//   * the rows of a workgroup's NEXT position are requested half a tile ahead, behind a NON-blocking look at its flags.  When a
//     flag is missing the workgroup finishes and PUBLISHES everything it holds and only then blocks on the flags: every blocking
//     wait is for positions lower than any unfinished position the waiter holds, so the lowest unfinished position of the queue
//     can always run -- no deadlock whatever the dispatch order or the number of resident workgroups (positions are handed out
//     by the atomic, never assumed).
//   * epoch = the forward's generation number: nothing is zeroed per launch except the queue head (by k_train_prep).
//   * every wait is bounded: a timeout raises the abort word, all workgroups drain, the status word reports it (bit 4).
// Arithmetic and its order are those of k_layer_fwd_p / k_layer_bwd_p: the results are bit-identical to the per-layer launches.
#include "train_stackq.h"

// ------------------------------------------------------------------------------------------------ forward
// dynamic LDS: As[2][16][lda] | Gs | SGs | THs | Xs ([16][ldg] each) | control words
template <int KS>
__global__ __launch_bounds__(256, 2) void k_stack_fwd(TrainParams p, StackQ q) {
    constexpr int C = 64, Ktp = 16 * KS;
    constexpr bool HOIST = KS == 8;                               // K = 2C: the auxiliary 1x1 at frame rate (TrainParams::hoist), as in k_layer_fwd_p
    constexpr int lda = ((Ktp + 29) / 32) * 32 + 2, ldg = ((C + 29) / 32) * 32 + 2;
    extern __shared__ float sm[];
    float* Gs = sm + 32 * lda;
    float* SGs = Gs + 16 * ldg; float* THs = SGs + 16 * ldg; float* Xs = THs + 16 * ldg;
    int* ctl = (int*)(Xs + 16 * ldg);          // [0..3] position ring, [4..7] per wave: the next tile's rows were NOT requested, [8] arrivals at the publish point
    const int Ap = p.Ap, N1 = p.N1;
    const int tid = threadIdx.x, lane = tid & 63, wave = sq_rfl(tid >> 6);
    const int srow = tid >> 4, sc4 = tid & 15;
    const bool aux_thread = sc4 < (Ktp - 2 * C) / 4, aux_real = 4 * sc4 < Ap;
    const int orow = tid >> 5, oc2 = (tid & 31) * 2;
    const int c = 16 * wave + (lane & 15);
    const int arow = lane & 15, ak = lane >> 4;
    float* const dmy = p.scratch_rows + (size_t)blockIdx.x * 2 * C + oc2;
    const unsigned xbytes = (unsigned)N1 * C * 4u;                // one batch item of one layer's activations
    const size_t xlayer = (size_t)p.B * N1 * C;                   // floats between consecutive layers' activations

    // ---- resident weight fragments of the layer in hand
    float4 w1[KS][2], wr[4];
    float bs = 0.f, bt = 0.f, bb = 0.f;
    auto load_weights = [&](int l) {
        const TrLayer ly = p.layers[l];
        const float4* W1 = p.wp + ly.w1_f4; const float4* Wr = p.wp + ly.wr_f4;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { w1[ks][0] = W1[((size_t)ks * 8 + wave) * 64 + lane]; w1[ks][1] = W1[((size_t)ks * 8 + 4 + wave) * 64 + lane]; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wr[ks] = Wr[((size_t)ks * 4 + wave) * 64 + lane];
        bs = p.bp[ly.bias1 + c]; bt = p.bp[ly.bias1 + C + c]; bb = p.bp[ly.biasr + c];
        // waited for HERE, on the layer change's own path: left to the first MFMA that reads them, hipcc's wait for the two paths' merged state lands
        // in front of EVERY tile's first MFMA (`s_waitcnt vmcnt(1)`), i.e. behind the previous tile's write-through block output.  (The builtin, not
        // `asm`: SIInsertWaitcnts reads an S_WAITCNT that is an instruction and takes everything older as complete; it cannot see into `asm`.)
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0), nothing else
    };
    auto xrsrc = [&](const float* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, xbytes, 0x00020000); };
    int tp = 0; float4 rc, rp, rx = make_float4(0.f, 0.f, 0.f, 0.f);
    float2 rwj = make_float2(0.f, 0.f); float rpb0 = 0.f, rpb1 = 0.f, xaux_a = 0.f, xaux_b0 = 0.f, xaux_b1 = 0.f;      // aux hoist: operands of the extra MFMA step (in flight / in hand)
    auto load_tap = [&](const SqTile& d, int& out) {
        const int n = d.n0 + srow;
        out = (p.TAP + d.tapb)[n < N1 ? n : N1 - 1];
    };
    auto load_rows = [&](const SqTile& d, bool go) {           // go (wave-uniform) false: the two activation loads are dropped by the range check
        const int n = d.n0 + srow, nn = n < N1 ? n : N1 - 1;
        const auto rs = xrsrc(p.X + (size_t)d.xrow * C);
        const unsigned oc = go ? (__umul24((unsigned)nn, (unsigned)C) + 4u * sc4) * 4u : SQ_OOB;
        const unsigned op = go ? (__umul24((unsigned)tp, (unsigned)C) + 4u * sc4) * 4u : SQ_OOB;
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oc, 0, SQ_SC1);
        const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)op, 0, SQ_SC1);
        rc = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        rp = make_float4(__uint_as_float(b4.x), __uint_as_float(b4.y), __uint_as_float(b4.z), __uint_as_float(b4.w));
        if constexpr (HOIST) {       // (d.hrow: float offset of the tile's first frame in PA; WJ has 16 rows of padding behind row N1 - 1)
            const float* pa = p.PA + d.hrow + ((lane >> 4) & 1) * 2 * C + 16 * wave + (lane & 15);
            rwj = p.WJ[d.n0 + (lane & 15)];
            rpb0 = pa[0]; rpb1 = pa[C];
        } else rx = *(const float4*)(p.HUP + (size_t)d.hrow * Ap + (__umul24((unsigned)nn, (unsigned)Ap) + (aux_real ? 4u * sc4 : 0u)));
        __builtin_amdgcn_sched_barrier(0);                       // (all requests before anything waits for one of them)
    };
    auto store_rows = [&](const SqTile& d, float* As) {
        // (all three requests are taken up here by EVERY lane: the lanes without an auxiliary column never read theirs, and a register hipcc believes
        //  pending is waited for where it is next written -- the next tile's first MFMAs, behind the write-through block output)
        if constexpr (HOIST) asm volatile("" : "+v"(rpb1)); else asm volatile("" : "+v"(rx.w));
        const bool in = d.n0 + srow < N1;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 vc = in ? rc : z, vp = in ? rp : z, vx = (in && aux_real) ? rx : z;
        float* dd = As + (size_t)srow * lda + 4 * sc4;
        *(float2*)dd = make_float2(vc.x, vc.y); *(float2*)(dd + 2) = make_float2(vc.z, vc.w);
        *(float2*)(dd + C) = make_float2(vp.x, vp.y); *(float2*)(dd + C + 2) = make_float2(vp.z, vp.w);
        if constexpr (HOIST) { xaux_a = tr_aux_a(rwj, lane); xaux_b0 = lane < 32 ? rpb0 : 0.f; xaux_b1 = lane < 32 ? rpb1 : 0.f; }
        else if (aux_thread) { *(float2*)(dd + 2 * C) = make_float2(vx.x, vx.y); *(float2*)(dd + 2 * C + 2) = make_float2(vx.z, vx.w); }
    };
    auto store_out = [&](const float* T, float* dst, int n0) {      // a [16][64] LDS tile -> rows n0.. of a [N1][64] array (plain stores: read by later kernels only)
        const float2 v0 = *(const float2*)(T + (size_t)orow * ldg + oc2), v1 = *(const float2*)(T + (size_t)(orow + 8) * ldg + oc2);
        float* d0 = dst + (__umul24((unsigned)(n0 + orow), (unsigned)C) + oc2);
        float* d1 = d0 + 8 * C;
        d0 = n0 + orow < N1 ? d0 : dmy;
        d1 = n0 + orow + 8 < N1 ? d1 : dmy + C;
        *(float2*)d0 = v0; *(float2*)d1 = v1;
    };
    // (the registers a store read its data from are not written again before the store has completed -- hipcc waits for that --, and they were the
    //  first accumulators of the next tile: `s_waitcnt vmcnt(1)` behind its first MFMA.  The value stays live until the next publish point instead.)
    u32x4 xkeep = {0u, 0u, 0u, 0u};
    auto store_x = [&](const SqTile& d, bool go) {                // the block output of tile d (in Xs) -> X[l + 1], write-through, 16 bytes per thread
        const float2 v0 = *(const float2*)(Xs + (size_t)srow * ldg + 4 * sc4), v1 = *(const float2*)(Xs + (size_t)srow * ldg + 4 * sc4 + 2);
        const int n = d.n0 + srow;
        const unsigned o = (go && n < N1) ? (__umul24((unsigned)n, (unsigned)C) + 4u * sc4) * 4u : SQ_OOB;
        const u32x4 v = {__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v1.x), __float_as_uint(v1.y)};
        __builtin_amdgcn_raw_buffer_store_b128(v, xrsrc(p.X + (size_t)d.xrow * C + xlayer), (int)o, 0, SQ_SC1);
        xkeep = v;
    };
    auto publishes = [&](const SqTile& d) { return sq_valid(d) && !sq_last(d); };     // (nothing reads the last block's residual output: no rows, no flag)

    int zero_v; asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    // ---- prologue: three positions, the first tile staged
    // positions come from SQ_NQ sub-queues (position = ticket * SQ_NQ + sub-queue, each handed out in increasing order; one head word
    // saturates at ~88 returning atomics per microsecond chip-wide: CDNA4 guide, price list 'dequeue'); a workgroup's home is blockIdx % SQ_NQ
    const int NQ = q.nq, sub = (blockIdx.x / 8) % NQ;      // (blockIdx % 8 tells the XCD under round-robin dispatch: a sub-queue served by ONE XCD drifts away from the others)
    unsigned* const head = q.head + sub * TR_QHEAD_STRIDE + zero_v;
    // (three separate tickets, each requested when the previous one has returned: every workgroup does the same, so the first tickets of
    //  all of them come before the second ones -- ONE add of 3 gave a workgroup three neighbouring tiles, and the workgroups that started
    //  with tiles of layer 1 waited for layer 0 tiles their neighbours had third in line)
    if (tid == 0) {
        ctl[8] = 0;
        const unsigned k0 = atomicAdd(head, 1u); ctl[0] = (int)(k0 * NQ + sub);
        asm volatile("" ::: "memory");
        const unsigned k1 = atomicAdd(head, 1u); ctl[1] = (int)(k1 * NQ + sub);
        asm volatile("" ::: "memory");
        const unsigned k2 = atomicAdd(head, 1u); ctl[2] = (int)(k2 * NQ + sub);
    }
    __syncthreads();
    SqTile cur = sq_take(sq_fetch(q, sq_rfl(ctl[0]))), next = sq_take(sq_fetch(q, sq_rfl(ctl[1]))), nn = sq_take(sq_fetch(q, sq_rfl(ctl[2])));
    if (!sq_valid(cur)) return;
    SqTile prev = cur; prev.meta = 0;
    int lw = sq_layer(cur);
    load_weights(lw);
    load_tap(cur, tp);
    sq_wait(q, cur.dfirst, cur.dn, lane, p.status, 0u);
    load_rows(cur, true);
    store_rows(cur, sm);
    load_tap(next, tp);
    // Per tile, three barriers:
    //   B1  the tile's staged rows are complete
    //       gate contraction; then (the previous tile's write-through rows left a whole contraction ago) this wave's counted wait, an
    //       arrival on the LDS counter, and the wave that arrives LAST publishes the previous tile -- no barrier of its own
    //       gate epilogue
    //   B2  the gate tile is complete
    //       the flags of the NEXT tile's producers are requested here -- as late as the tile allows: a producer is ~2.45 rounds of the
    //       resident workgroups ahead of its consumer, of which one round is its own tile and ~0.5 its rows' way to memory -- and looked
    //       at ONCE behind the residual contraction; a wave that finds them all requests the next tile's rows
    //   B3  the block output is complete in LDS and leaves (write-through); every wave knows whether all four found their flags.
    //       If not: the stores drain, a barrier, the tile is PUBLISHED, and only then do the waves without rows wait (so every wait is for
    //       positions lower than any unpublished position the workgroup holds).  Then the next tile's rows are requested
    bool cur_published = false;         // (the tile just finished was published by the slow path: this trip's publish point skips it)
    for (int it = 0;; ++it) {
        float* As = sm + (it & 1) * 16 * lda;
        const bool last = sq_last(cur);
        SQ_STAMP(0);
        // the position three tiles ahead.  (The address goes through an opaque zero: with a provably uniform address LLVM's atomic optimizer
        // turns the add into a wave reduction whose v_readfirstlane needs the returned value AT ONCE -- s_waitcnt vmcnt(0) at the top of every tile.)
        unsigned rtk = 0;                                         // (the raw ticket: any arithmetic on it here would wait for the atomic at once)
        if (tid == 0) rtk = atomicAdd(head, 1u);
        if (sq_layer(cur) != lw) { lw = sq_layer(cur); load_weights(lw); }
        int tpn; load_tap(nn, tpn);
        TR_LDS_BARRIER();                                          // B1
        const bool pub_prev = publishes(prev) && !cur_published;
        SQ_STAMP(1);
        float xa[KS][4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float* ap = As + (size_t)arow * lda + 16 * ks + ak;
            xa[ks][0] = ap[0]; xa[ks][1] = ap[4]; xa[ks][2] = ap[8]; xa[ks][3] = ap[12];
        }
        __builtin_amdgcn_sched_barrier(0);
        SQ_PRIO(0);
        f32x4 a0 = (f32x4){0, 0, 0, 0}, a1 = (f32x4){0, 0, 0, 0};
        if constexpr (HOIST) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xaux_a, xaux_b0, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xaux_a, xaux_b1, a1, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][0], w1[ks][0].x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][0], w1[ks][1].x, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][1], w1[ks][0].y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][1], w1[ks][1].y, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][2], w1[ks][0].z, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][2], w1[ks][1].z, a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][3], w1[ks][0].w, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][3], w1[ks][1].w, a1, 0, 0, 0);
        }
        SQ_PRIO(2);
        __builtin_amdgcn_sched_barrier(0);
        SQ_STAMP(2);
        // publish point: younger than the previous tile's row stores are only this trip's tap load and, in wave 0, the ticket
        __builtin_amdgcn_s_waitcnt(0x0F71);                        // vmcnt(1) (the builtin: hipcc's own bookkeeping sees it)
        asm volatile("" :: "v"(xkeep));
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(ctl + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & 3) == 3 && pub_prev) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
        }
        __builtin_amdgcn_sched_barrier(0);
        SQ_STAMP(3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = (4 * (lane >> 4) + i) * ldg + c;
            const float sg = sq_sigmoid(a0[i] + bs), th = sq_tanh(a1[i] + bt);
            Gs[o] = sg * th; SGs[o] = sg;
        }
        SQ_STAMP(4);
        if (tid == 0) ctl[(it + 3) & 3] = (int)(rtk * NQ + sub);
        TR_LDS_BARRIER();                                          // B2
        SQ_STAMP(5);
        const SqRaw raw3 = sq_fetch(q, sq_rfl(ctl[(it + 3) & 3]));       // the table entry of the tile three ahead: taken at the end of the trip
        const int fn = next.dn, fnm1 = fn > 0 ? fn - 1 : 0, fbase = fn > 0 ? next.dfirst : cur.pos;      // (no producers: a word of its own)
        unsigned fv0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), fv1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
        asm volatile("" ::: "memory");
        float* SG = p.SG + (size_t)cur.xrow * C;
        float* TH = p.TH + (size_t)cur.xrow * C;
        if (!last) {                                               // the last block's residual output is never used (qpnet.py:306-309)
            float ga[4][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float* gp = Gs + (size_t)arow * ldg + 16 * ks + ak;
                ga[ks][0] = gp[0]; ga[ks][1] = gp[4]; ga[ks][2] = gp[8]; ga[ks][3] = gp[12];
            }
            __builtin_amdgcn_sched_barrier(0);
            SQ_PRIO(0);
            f32x4 ar0 = (f32x4){0, 0, 0, 0}, ar1 = (f32x4){0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][0], wr[ks].x, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][0], wr[ks + 1].x, ar1, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][1], wr[ks].y, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][1], wr[ks + 1].y, ar1, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][2], wr[ks].z, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][2], wr[ks + 1].z, ar1, 0, 0, 0);
                ar0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks][3], wr[ks].w, ar0, 0, 0, 0); ar1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[ks + 1][3], wr[ks + 1].w, ar1, 0, 0, 0);
            }
            SQ_PRIO(2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (lane >> 4) + i;
                Xs[r * ldg + c] = ((ar0[i] + ar1[i]) + bb) + As[(size_t)r * lda + c];
            }
        }
        SQ_STAMP(6);
        asm volatile("" : "+v"(fv0), "+v"(fv1));                   // (the flag words are looked at HERE: left alone hipcc compares them, i.e. waits for them, right behind the loads)
        bool ready = fn == 0 || (fn <= 128 && __all(fv0 == q.epoch && fv1 == q.epoch));
#if SQ_EXP & 512
        ready = true;
#endif
        // sigma / tanh leave BEHIND the look at the flags: vmcnt counts in order, so stores issued in front of it are waited for with it
        asm volatile("" ::: "memory");
#if !(SQ_EXP & 64)
        store_out(SGs, SG, cur.n0);
        store_out(Gs, TH, cur.n0);                                 // the gate product (p.TH: train_common.h)
#endif
        if (lane == 0) ctl[4 + wave] = ready ? 0 : 1;
        TR_LDS_BARRIER();                                          // B3
        int any_slow = sq_rfl(ctl[4] | ctl[5] | ctl[6] | ctl[7]);
        cur_published = false;
        if (any_slow) {
            // Some wave did not find every flag.  Most such misses are near misses (the producer publishes within a microsecond or two): the
            // waves that missed look ONCE more before the workgroup pays for the hand-over below (a drain, two barriers, a wait).
            if (tid == 0) atomicAdd(q.stats + 2, 1u);
            if (!ready) {
                const unsigned g0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), g1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
                ready = fn <= 128 && __all(g0 == q.epoch && g1 == q.epoch);
            }
            if (lane == 0) ctl[12 + wave] = ready ? 0 : 1;        // (words of their own: a wave may still be reading the first look's)
            TR_LDS_BARRIER();
            any_slow = sq_rfl(ctl[12] | ctl[13] | ctl[14] | ctl[15]);
        }
        if (any_slow) {
            // a producer of the next tile has not published yet: hand over everything this workgroup holds, THEN wait (a workgroup that
            // waits while it holds finished, unpublished tiles makes its own consumers wait: measured, a convoy that tripled the launch)
            if (tid == 0) atomicAdd(q.stats, 1u);
            store_x(cur, publishes(cur));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TR_LDS_BARRIER();
            if (tid == 0 && publishes(cur)) sq_st(q.flags + sq_fidx((unsigned)cur.pos), q.epoch_pub);
            cur_published = true;
            if (!ready) sq_wait(q, next.dfirst, next.dn, lane, p.status, 0u);
        }
        // the next tile's rows: ONE request site (a second one inside the branch above makes the row registers phi nodes, and hipcc
        // resolves them with copies -- i.e. waits for the rows, and for the write-through stores in front of them, right here)
#if SQ_EXP & 256
        rc = rp = rx = make_float4(0.25f, 0.5f, 0.125f, 0.75f); rwj = make_float2(0.5f, 0.f); rpb0 = rpb1 = 0.25f;
#else
        load_rows(next, true);
#endif
        // ... and the block output leaves BEHIND them (vmcnt counts in order: in front of them, the staging below would wait for the
        // write-through stores' way to memory as well)
        // (ONE unconditional instruction -- its lanes aim beyond the range when the slow path has stored the tile already --: behind a conditional one
        //  hipcc's wait for the rows becomes vmcnt(0), i.e. a wait for this store's way to memory)
#if !(SQ_EXP & 128)
        store_x(cur, publishes(cur) && !any_slow);
#endif
        store_rows(next, sm + ((it + 1) & 1) * 16 * lda);
        SQ_STAMP(7);
        tp = tpn;
        prev = cur; cur = next; next = nn; nn = sq_take(raw3);
        if (!sq_valid(cur)) break;
    }
    // the last tile's output has left; publish it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_LDS_BARRIER();
    if (tid == 0 && publishes(prev) && !cur_published) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
}

// ------------------------------------------------------------------------------------------------ backward
// The layer backward of k_layer_bwd_p (train_bwd.hip: dg = dXout . Wr^T + skip-path grads, dz = dg * gate', d[x_cur | x_past | aux] = dZ . W1^T,
// the pitch-tap part scattered to its tap rows) over the same kind of queue, layers from the last to the first.  What is handed between
// workgroups here: the own-row part of the input gradient (write-through rows, a unique writer) and the scattered part (float atomics --
// they execute at the memory side -- for the adaptive blocks, write-through rows for the fixed ones); a tile is published when both have
// completed (vmcnt counts the atomics too), and its consumers -- the tiles of the layer below whose rows it touched -- read both parts with
// sc1 loads.  dZ, sigma / tanh, the skip-path grads and the aux-feature gradient are not handed over inside the launch (plain accesses / atomics).
// dynamic LDS: staging {Dx | Sg | Th | Dg}[2][16][ldx] | Dz [16][ldz] | Os [16][ldo] | control words
template <int NTK>
__global__ __launch_bounds__(256, 2) void k_stack_bwd(TrainParams p, TrainBwd bw, StackQ q) {
    constexpr int C = 64;
    constexpr int ldx = ((C + 29) / 32) * 32 + 2, ldz = ((2 * C + 29) / 32) * 32 + 2, ldo = ((16 * NTK + 29) / 32) * 32 + 2;
    constexpr int NJ = (NTK + 3) / 4;
    constexpr bool HOIST = NTK == 8;                              // the auxiliary 1x1 at frame rate (TrainParams::hoist): no aux columns in the input gradient; D / G instead (train_common.h, tr_aux_bwd)
    extern __shared__ float sm[];
    float* Dz = sm + 8 * 16 * ldx;
    float* Os = Dz + 16 * ldz;
    int* ctl = (int*)(Os + 16 * ldo);
    float* Gp = (float*)(ctl + 32);                               // hoist: [4 waves][16 rows] partial sums of dZ[row][.] * PA[frame(row)][.]
    const int Ap = p.Ap, N1 = p.N1;
    const int tid = threadIdx.x, lane = tid & 63, wave = sq_rfl(tid >> 6);
    const int srow = tid >> 4, sc4 = tid & 15;
    const int zrow = tid >> 6, zc2 = (tid & 63) * 2;
    const int arow = lane & 15, ak = lane >> 4;
    const int c = 16 * wave + (lane & 15);
    const int win0 = N1 - p.BL;
    const unsigned xbytes = (unsigned)N1 * C * 4u;
    const size_t nDX = (size_t)p.B * N1 * C;
    float* const dmy = p.scratch_rows + (size_t)blockIdx.x * 2 * 128;       // two 512-byte scratch rows per workgroup

    float4 wr[4], w1[NJ][8];
    auto load_weights = [&](int l, bool last) {
        const TrLayer ly = p.layers[l];
        const float4* Wrt = p.wp + ly.wrt_f4; const float4* W1t = p.wp + ly.w1t_f4;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const float4 v = Wrt[((size_t)ks * 4 + wave) * 64 + lane]; wr[ks] = last ? make_float4(0.f, 0.f, 0.f, 0.f) : v; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nt = wave + 4 * j < NTK ? wave + 4 * j : NTK - 1;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) w1[j][ks] = W1t[((size_t)ks * NTK + nt) * 64 + lane];
        }
        // waited for HERE, on the layer change's own path: left to the first MFMA that reads them, the wait lands on every tile's path (hipcc merges
        // the two paths' pending counts), where it waits for the previous tile's outputs instead
        static_assert(NJ == 3 || NJ == 2, "the operand lists below name every float4 of wr / w1");
        asm volatile("" :: "v"(wr[0].w), "v"(wr[1].w), "v"(wr[2].w), "v"(wr[3].w),
                     "v"(w1[0][0].w), "v"(w1[0][1].w), "v"(w1[0][2].w), "v"(w1[0][3].w), "v"(w1[0][4].w), "v"(w1[0][5].w), "v"(w1[0][6].w), "v"(w1[0][7].w),
                     "v"(w1[1][0].w), "v"(w1[1][1].w), "v"(w1[1][2].w), "v"(w1[1][3].w), "v"(w1[1][4].w), "v"(w1[1][5].w), "v"(w1[1][6].w), "v"(w1[1][7].w));
        if constexpr (NJ == 3)
            asm volatile("" :: "v"(w1[NJ - 1][0].w), "v"(w1[NJ - 1][1].w), "v"(w1[NJ - 1][2].w), "v"(w1[NJ - 1][3].w), "v"(w1[NJ - 1][4].w), "v"(w1[NJ - 1][5].w), "v"(w1[NJ - 1][6].w), "v"(w1[NJ - 1][7].w));
    };
    auto rsrc = [&](const float* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, xbytes, 0x00020000); };
    // the rows of a tile come from two places: what this layer's forward and the post-net's backward left (sigma, tanh, the skip-path gradient:
    // nobody inside the launch writes them) and the gradient the layer above hands over (write-through rows and atomics of other workgroups)
    float4 ra, rb2, rsg, rth, rdg;
    auto row_off = [&](const SqTile& d) { const int n = d.n0 + srow, nn_ = n < N1 ? n : N1 - 1; return __umul24((unsigned)nn_, (unsigned)C) + 4u * sc4; };
    auto load_own = [&](const SqTile& d) {
        const unsigned o = row_off(d);
        const int n = d.n0 + srow, nn_ = n < N1 ? n : N1 - 1;
        rsg = *(const float4*)(p.SG + (size_t)d.xrow * C + o); rth = *(const float4*)(p.TH + (size_t)d.xrow * C + o);
        const int nw = nn_ >= win0 ? nn_ - win0 : 0;
        rdg = *(const float4*)(bw.DGS + (size_t)d.dgs + (__umul24((unsigned)nw, (unsigned)p.LC) + 4u * sc4));
    };
    auto load_ab = [&](const SqTile& d, float4& a4, float4& b4_) {      // (nothing for the last layer: dropped by the range check)
        const unsigned ob = sq_last(d) ? SQ_OOB : row_off(d) * 4u;
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc(bw.DXA[0] + (size_t)d.xrow * C + nDX), (int)ob, 0, SQ_SC1);
        const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(rsrc(bw.DXB[0] + (size_t)d.xrow * C + nDX), (int)ob, 0, SQ_SC1);
        a4 = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        b4_ = make_float4(__uint_as_float(b4.x), __uint_as_float(b4.y), __uint_as_float(b4.z), __uint_as_float(b4.w));
    };
    auto store_own = [&](const SqTile& d, float* B) {
        const int n = d.n0 + srow;
        const bool in = n < N1;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 sg = in ? rsg : z, th = in ? rth : z, dg = (in && n >= win0) ? rdg : z;
        float* d1 = B + (size_t)(16 + srow) * ldx + 4 * sc4; *(float2*)d1 = make_float2(sg.x, sg.y); *(float2*)(d1 + 2) = make_float2(sg.z, sg.w);
        float* d2 = d1 + 16 * ldx; *(float2*)d2 = make_float2(th.x, th.y); *(float2*)(d2 + 2) = make_float2(th.z, th.w);
        float* d3 = d2 + 16 * ldx; *(float2*)d3 = make_float2(dg.x, dg.y); *(float2*)(d3 + 2) = make_float2(dg.z, dg.w);
    };
    auto store_dx = [&](const SqTile& d, float* B, const float4& a4, const float4& b4_) {
        const int n = d.n0 + srow;
        const bool in = n < N1 && !sq_last(d);
        const float4 dx = in ? make_float4(a4.x + b4_.x, a4.y + b4_.y, a4.z + b4_.z, a4.w + b4_.w) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* d0 = B + (size_t)srow * ldx + 4 * sc4;
        *(float2*)d0 = make_float2(dx.x, dx.y); *(float2*)(d0 + 2) = make_float2(dx.z, dx.w);
        // the sum replaces the own-row part IN PLACE: this tile is those rows' only reader inside the launch, and the residual 1x1's weight gradient behind
        // the launch then reads ONE array instead of two (41 MB less on the step's memory-bound tail; qpn_launch_bwd: build_wr)
        if (in) *(float4*)(bw.DXA[0] + (size_t)d.xrow * C + nDX + row_off(d)) = dx;
    };
    // aux hoist: the WJ entries of this lane's four dZ rows and its four PA values (tr_aux_bwd) -- c*: the tile in hand, n*: the next tile's, in flight
    float2 cwj[4], nwj[4]; float cpa[4], npa[4]; float dacc[4] = {0.f, 0.f, 0.f, 0.f};
    float eacc[2] = {0.f, 0.f};
    auto load_aux = [&](const SqTile& d, float2 (&wj)[4], float (&pa)[4]) {
        const float2* w = p.WJ + d.n0 + 4 * (lane >> 4);          // (16 rows of padding behind row N1 - 1)
        const float* q4 = p.PA + d.hrow + 16 * wave + (lane & 15);      // d.hrow: float offset of the tile's first frame in PA / DPA
#pragma unroll
        for (int i = 0; i < 4; ++i) wj[i] = w[i];
        pa[0] = q4[0]; pa[1] = q4[C]; pa[2] = q4[2 * C]; pa[3] = q4[3 * C];
    };
    int tprow[4];
    auto load_taps = [&](const SqTile& d, int (&tp)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int n = d.n0 + 4 * wave + i; tp[i] = (p.TAP + d.tapb)[n < N1 ? n : N1 - 1]; }
    };
    // the tile's outputs (behind B3: Os and the tile's Dx are complete in LDS)
    auto outputs = [&](const SqTile& d, const float* Dx) {
        const bool adaptive = (d.meta >> 26) & 1;
        const int dil = p.layers[sq_layer(d)].dilation;
#if !(SQ_EXP & 16)
        {   // own-row part (+ residual path): 16 bytes per thread, write-through
            const int n = d.n0 + srow;
            const float2 o0 = *(const float2*)(Os + (size_t)srow * ldo + 4 * sc4), o1 = *(const float2*)(Os + (size_t)srow * ldo + 4 * sc4 + 2);
            const float2 x0 = *(const float2*)(Dx + (size_t)srow * ldx + 4 * sc4), x1 = *(const float2*)(Dx + (size_t)srow * ldx + 4 * sc4 + 2);
            const u32x4 v = {__float_as_uint(o0.x + x0.x), __float_as_uint(o0.y + x0.y), __float_as_uint(o1.x + x1.x), __float_as_uint(o1.y + x1.y)};
            const unsigned off = n < N1 ? (__umul24((unsigned)n, (unsigned)C) + 4u * sc4) * 4u : SQ_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(bw.DXA[0] + (size_t)d.xrow * C), (int)off, 0, SQ_SC1);
            if (!adaptive) {   // fixed block: the tap row n - dilation has this one writer
                const float2 p0 = *(const float2*)(Os + (size_t)srow * ldo + C + 4 * sc4), p1 = *(const float2*)(Os + (size_t)srow * ldo + C + 4 * sc4 + 2);
                const u32x4 w = {__float_as_uint(p0.x), __float_as_uint(p0.y), __float_as_uint(p1.x), __float_as_uint(p1.y)};
                const int tr = n - dil;
                const unsigned offb = (n < N1 && tr >= 0) ? (__umul24((unsigned)tr, (unsigned)C) + 4u * sc4) * 4u : SQ_OOB;
                __builtin_amdgcn_raw_buffer_store_b128(w, rsrc(bw.DXB[0] + (size_t)d.xrow * C), (int)offb, 0, SQ_SC1);
            }
        }
#endif
        float* DBout = bw.DXB[0] + (size_t)d.xrow * C;
        float* DH = HOIST ? nullptr : bw.DHUP + (size_t)d.hrow * Ap;
#pragma unroll
        for (int i = 0; i < 4; ++i) {      // wave w owns rows 4w .. 4w+3, lane = channel: one 256-byte row per instruction
            const int r = 4 * wave + i, n = d.n0 + r;
            const bool in = n < N1;
#if !(SQ_EXP & 1)
            if (adaptive) {                // gather backward (collisions): ONE full-row float-atomic instruction per tap row
                float* db = DBout + (__umul24((unsigned)tprow[i], (unsigned)C) + lane);
                atomicAdd(in ? db : dmy + 128 + lane, Os[(size_t)r * ldo + C + lane]);
            }
            if constexpr (!HOIST) if (lane < Ap) {
                float* dh = DH + (__umul24((unsigned)n, (unsigned)Ap) + lane);
                atomicAdd(in ? dh : dmy + 192 + lane, Os[(size_t)r * ldo + 2 * C + lane]);
            }
#endif
        }
        if constexpr (HOIST) {
#if !(SQ_EXP & 1)
            // D_l[frame][gate row] += this tile's share (two frames x sigma / tanh: 64-byte strips of lanes 0..15), and the tile's 16 G values
            if (lane < 16) {
                float* dp = bw.DPA + d.hrow + 16 * wave + lane;
                atomicAdd(dp, dacc[0]); atomicAdd(dp + C, dacc[1]); atomicAdd(dp + 2 * C, dacc[2]); atomicAdd(dp + 3 * C, dacc[3]);
                float* ep = bw.EB + (size_t)(sq_layer(d) * TR_EB_SLOTS + (blockIdx.x & (TR_EB_SLOTS - 1))) * 2 * C + 16 * wave + lane;       // the gate-bias accumulators: a slab per workgroup index mod 32
                atomicAdd(ep, eacc[0]); atomicAdd(ep + C, eacc[1]);
            }
#endif
            if (tid < 16 && d.n0 + tid < N1) bw.GW[(size_t)d.xrow + d.n0 + tid] = (Gp[tid] + Gp[16 + tid]) + (Gp[32 + tid] + Gp[48 + tid]);
        }
    };
    auto publishes = [&](const SqTile& d) { return sq_valid(d) && sq_layer(d) > 0; };     // (layer 0's input gradient feeds later kernels only)

    int zero_v; asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const int NQ = q.nq, sub = (blockIdx.x / 8) % NQ;
    unsigned* const head = q.head + sub * TR_QHEAD_STRIDE + zero_v;
    if (tid == 0) {
        ctl[8] = 0;
        const unsigned k0 = atomicAdd(head, 1u); ctl[0] = (int)(k0 * NQ + sub);
        asm volatile("" ::: "memory");
        const unsigned k1 = atomicAdd(head, 1u); ctl[1] = (int)(k1 * NQ + sub);
        asm volatile("" ::: "memory");
        const unsigned k2 = atomicAdd(head, 1u); ctl[16] = (int)(k2 * NQ + sub);
    }
    __syncthreads();
    SqTile cur = sq_take(sq_fetch(q, sq_rfl(ctl[0]))), next = sq_take(sq_fetch(q, sq_rfl(ctl[1])));
    if (!sq_valid(cur)) return;
    SqTile prev = cur; prev.meta = 0;
    int lw = sq_layer(cur);
    load_weights(lw, sq_last(cur));
    sq_wait(q, cur.dfirst, cur.dn, lane, p.status, 0u);
    load_own(cur); load_ab(cur, ra, rb2);
    if constexpr (HOIST) load_aux(cur, cwj, cpa);
    load_taps(cur, tprow);
    store_own(cur, sm); store_dx(cur, sm, ra, rb2);
    bool cur_published = false;
    // the flag words live in a buffer descriptor too: the publishing store is ONE unconditional instruction (every lane but one aims beyond the range)
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void*)q.flags, 0, (int)((sq_fidx((unsigned)q.total) + 64u) * 4u), 0x00020000);
    // Per tile: three barriers, ONE wait for everything that can be asked early, one counted wait for the two rows that cannot.
    //   B1  the tile's staged rows are complete;  dg = dXout . Wr + skip-path gradient, dz = dg * gate'  -> Dz
    //   B2  the request group: the NEXT tile's own-layer rows and taps, its producers' flags, the table entry of the position two tiles ahead
    //       (its ticket was taken up at the previous tile's wait), the ticket of the one three ahead;  then d[x_cur | x_past | aux] = dZ . W1
    //       -- 96 MFMAs, ~3.7k cycles, which is what the group needs to come back --
    //       the wait (vmcnt 0): rows / taps / table entry / ticket are taken up; the next tile's handed-over rows are requested if its flags
    //       are all there; the previous tile is published (its outputs left in front of the group, a whole tile ago)
    //       own-layer rows and this tile's products -> LDS, dZ rows -> memory
    //   B3  every wave knows whether all four found the flags.  Yes: the handed-over rows (counted wait: only the publishing store and the dZ
    //       stores are younger) -> LDS, then the tile's outputs leave -- nobody looks at them before the next tile's wait.  No: outputs, drain,
    //       the tile is PUBLISHED, and only then do the waves wait and fetch those rows.
    // [vmcnt counts in order: a request consumed anywhere else costs a memory latency whatever its own age -- with the rows requested behind B3
    //  and the ticket at the top of the tile a tile took 15.7k cycles (3.7k of them MFMA issue), 10k with every global access removed.]
    for (int it = 0;; ++it) {
        float* Dx = sm + (it & 1) * 4 * 16 * ldx; float* Sg = Dx + 16 * ldx; float* Th = Sg + 16 * ldx; float* Dg = Th + 16 * ldx;
        float* Bn = sm + ((it + 1) & 1) * 4 * 16 * ldx;           // the next tile's staging buffer (last read by the previous tile's outputs, in front of B1)
        const bool last = sq_last(cur);
        if (sq_layer(cur) != lw) { lw = sq_layer(cur); load_weights(lw, last); }
        const bool pub_prev = publishes(prev) && !cur_published;
        SQ_STAMPB(0);
        TR_LDS_BARRIER();                                          // B1
        float xa[4][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const float* ap = Dx + (size_t)arow * ldx + 16 * ks + ak; xa[ks][0] = ap[0]; xa[ks][1] = ap[4]; xa[ks][2] = ap[8]; xa[ks][3] = ap[12]; }
        __builtin_amdgcn_sched_barrier(0);
        SQ_PRIO(0);
        f32x4 a0 = (f32x4){0, 0, 0, 0}, a1 = (f32x4){0, 0, 0, 0};
        if (!last) {
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][0], wr[ks].x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][0], wr[ks + 1].x, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][1], wr[ks].y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][1], wr[ks + 1].y, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][2], wr[ks].z, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][2], wr[ks + 1].z, a1, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks][3], wr[ks].w, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[ks + 1][3], wr[ks + 1].w, a1, 0, 0, 0);
            }
        }
        SQ_PRIO(2);
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
        if constexpr (HOIST) {      // the frame-rate aux term's backward (what the 48 aux columns of dZ . W1 and the dH atomics were)
            const TrAuxBwd ab = tr_aux_bwd(cwj, cpa, dzs, dzt, lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) dacc[k] = ab.d[k];
            eacc[0] = ab.e[0]; eacc[1] = ab.e[1];
            if ((lane & 15) == 0) { float* gq = Gp + 16 * wave + 4 * (lane >> 4); gq[0] = ab.gp[0]; gq[1] = ab.gp[1]; gq[2] = ab.gp[2]; gq[3] = ab.gp[3]; }
        }
        SQ_STAMPB(1);
        TR_LDS_BARRIER();                                          // B2
        // ---- the request group
        const SqRaw raw2 = sq_fetch(q, sq_rfl(ctl[16 + (it & 1)]));
        const int fn = next.dn, fnm1 = fn > 0 ? fn - 1 : 0, fbase = fn > 0 ? next.dfirst : cur.pos;
        unsigned fv0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), fv1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
        load_own(next);
        if constexpr (HOIST) load_aux(next, nwj, npa);
        int tpn[4]; load_taps(next, tpn);
        unsigned rtk = 0;
        if (tid == 0) rtk = atomicAdd(head, 1u);
        asm volatile("" ::: "memory");
        float za[8][4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { const float* zp = Dz + (size_t)arow * ldz + 16 * ks + ak; za[ks][0] = zp[0]; za[ks][1] = zp[4]; za[ks][2] = zp[8]; za[ks][3] = zp[12]; }
        __builtin_amdgcn_sched_barrier(0);
        SQ_PRIO(0);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0, 0, 0, 0};
#define SQB_MFMA(ks) do { \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][0], w1[j][ks].x, acc[j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][1], w1[j][ks].y, acc[j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][2], w1[j][ks].z, acc[j], 0, 0, 0); \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(za[ks][3], w1[j][ks].w, acc[j], 0, 0, 0); } while (0)
        SQB_MFMA(0); SQB_MFMA(1); SQB_MFMA(2); SQB_MFMA(3);
        __builtin_amdgcn_sched_barrier(0);
        // ---- the previous tile is published HALFWAY through the contraction: its outputs left ~3.7k cycles ago (the end of the previous tile, the gate
        // phase, half of this one), and every quarter of a tile by which a flag is early is one by which its consumer's look is less likely to miss.
        // Counted wait: the request group is younger than those outputs -- eleven requests (twelve in wave 0: the ticket); the count waited down to is
        // two BELOW that, so a group hipcc manages to issue with fewer instructions still cannot let an output through
        // (hoist: six more requests in the group -- the four WJ entries as two 16-byte loads, four PA values: seventeen, eighteen in wave 0)
        if constexpr (HOIST) { if (wave == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); }
        else { if (wave == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
        {
            int old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(ctl + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool pub = pub_prev && (sq_rfl(old) & 3) == 3;      // the wave that gets here LAST
            const unsigned foff = (pub && lane == 0) ? sq_fidx((unsigned)prev.pos) * 4u : SQ_OOB;
            __builtin_amdgcn_raw_buffer_store_b32(q.epoch_pub, frs, (int)foff, 0, SQ_SC1);
        }
        __builtin_amdgcn_sched_barrier(0);
        SQB_MFMA(4); SQB_MFMA(5); SQB_MFMA(6); SQB_MFMA(7);
#undef SQB_MFMA
        SQ_PRIO(2);
        __builtin_amdgcn_sched_barrier(0);
        SQ_STAMPB(2);
        // ---- the wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (hipcc does not see that wait: every register the group loaded is named here, or its own wait for the youngest of them -- the taps, the
        //  ticket -- lands behind the publishing store and the outputs, in the middle of their way to memory)
        asm volatile("" : "+v"(fv0), "+v"(fv1), "+v"(rtk), "+v"(tpn[0]), "+v"(tpn[1]), "+v"(tpn[2]), "+v"(tpn[3]));
        if constexpr (HOIST) asm volatile("" : "+v"(nwj[0].x), "+v"(nwj[1].x), "+v"(nwj[2].x), "+v"(nwj[3].x), "+v"(npa[0]), "+v"(npa[1]), "+v"(npa[2]), "+v"(npa[3]));
        SQ_STAMPB(3);
        bool ready = fn == 0 || (fn <= 128 && __all(fv0 == q.epoch && fv1 == q.epoch));
#if SQ_EXP & 32
        ready = true;
#endif
        load_ab(next, ra, rb2);                                    // (not final if the flags are not all there: fetched again on that path)
        store_own(next, Bn);
        const SqTile n2 = sq_take(raw2);
        if (tid == 0) ctl[16 + ((it + 1) & 1)] = (int)(rtk * NQ + sub);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nt = wave + 4 * j;
            if (nt >= NTK) continue;                               // wave-uniform
#pragma unroll
            for (int i = 0; i < 4; ++i) Os[(size_t)(4 * (lane >> 4) + i) * ldo + 16 * nt + (lane & 15)] = acc[j][i];
        }
        SQ_STAMPB(4);
#if !(SQ_EXP & 4)
        {   // dZ rows to global (512 B each; read by the weight-gradient kernels behind this launch)
            float* DZg = bw.DZ + (size_t)cur.xrow * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = zrow + 4 * k;
                float* d = DZg + (__umul24((unsigned)(cur.n0 + r), 2u * C) + zc2);
                d = cur.n0 + r < N1 ? d : dmy + zc2;
                *(float2*)d = *(const float2*)(Dz + (size_t)r * ldz + zc2);
            }
        }
#endif
        if (lane == 0) ctl[4 + wave] = ready ? 0 : 1;
        SQ_STAMPB(5);
        TR_LDS_BARRIER();                                          // B3
        SQ_STAMPB(6);
        int any_slow = sq_rfl(ctl[4] | ctl[5] | ctl[6] | ctl[7]);
        const int any_late = any_slow;                             // some wave's early request for the handed-over rows went out in front of a missing flag
        cur_published = false;
        if (any_slow) {                                            // a second look (one memory latency) before the tile gives its outputs up early
            if (tid == 0) atomicAdd(q.stats + 2, 1u);
            if (!ready) {
                const unsigned g0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), g1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
                ready = fn <= 128 && __all(g0 == q.epoch && g1 == q.epoch);
            }
            if (lane == 0) ctl[12 + wave] = ready ? 0 : 1;
            TR_LDS_BARRIER();
            any_slow = sq_rfl(ctl[12] | ctl[13] | ctl[14] | ctl[15]);
        }
        // the handed-over rows are waited for HERE, in front of the branch (counted: only the dZ stores are younger): hipcc's
        // control flow has a path around both arms, and a row register it believes pending is waited for at the top of the next tile -- behind the outputs
        asm volatile("" : "+v"(ra.w), "+v"(rb2.w));
        if (any_slow) {      // a producer of the next tile is still at work: this tile is published BEFORE anybody waits (a wait is only ever for positions below everything the workgroup holds back)
            if (tid == 0) atomicAdd(q.stats, 1u);
            outputs(cur, Dx);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TR_LDS_BARRIER();
            if (tid == 0 && publishes(cur)) sq_st(q.flags + sq_fidx((unsigned)cur.pos), q.epoch_pub);
            cur_published = true;
            if (!ready) sq_wait(q, next.dfirst, next.dn, lane, p.status, 0u);
            float4 la, lb; load_ab(next, la, lb);
            store_dx(next, Bn, la, lb);
        } else if (any_late) {                                     // the second look found them: the rows once more (a memory latency), nothing else changes
            float4 la, lb; load_ab(next, la, lb);
            store_dx(next, Bn, la, lb);
            outputs(cur, Dx);
        } else {
            store_dx(next, Bn, ra, rb2);
            outputs(cur, Dx);
        }
        SQ_STAMPB(7);
#pragma unroll
        for (int i = 0; i < 4; ++i) tprow[i] = tpn[i];
        if constexpr (HOIST) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { cwj[i] = nwj[i]; cpa[i] = npa[i]; }
        }
        prev = cur; cur = next; next = n2;
        if (!sq_valid(cur)) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_LDS_BARRIER();
    if (tid == 0 && publishes(prev) && !cur_published) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
}

// ------------------------------------------------------------------------------------------------ host side
// tests/test_train_gpu.py::test_stack_queue_that_gives_up_... (a -DQPN_TESTING build, QPN_TEST_STACK_GIVES_UP=1): the published flags carry another
// value than the consumers expect and the waits are short, so the first dependent tile runs out, raises the abort word and the launch drains
static void sq_arm(StackQ& q, const TrainKnobs& k) {
    q.epoch_pub = q.epoch; q.spin_limit = SQ_SPIN_LIMIT;
#ifdef QPN_TESTING
    if (k.test_stack_gives_up) { q.epoch_pub = q.epoch ^ 0x55555555u; q.spin_limit = 2000u; }
#else
    (void)k;
#endif
}
bool qpn_stack_fwd_fits(const TrainParams& p) {
    return p.C == 64 && (p.hoist ? p.Ktp == 128 : p.Ktp == 176) && p.L >= 1 && p.L <= TR_MAXL && p.B < 65536 && (int64_t)p.N1 * p.C * 4 <= (1ll << 30) &&
           (int64_t)(p.L + 1) * p.B * p.N1 < (1ll << 31);
}

// positions of the queue: layer-major, batch item, tile (rows ascending); the table itself is written on the device by k_train_prep
void qpn_stack_fill(TrainParams& p) {
    int pos = 0;
    for (int l = 0; l < p.L; ++l) {
        p.qT[l] = (p.N1 - p.layers[l].s_out + 15) / 16;
        p.qP[l] = pos; pos += p.B * p.qT[l];
    }
    for (int l = p.L; l <= TR_MAXL; ++l) p.qP[l] = pos;
    p.qtotal = pos;
}

template <int KS>
static int launch_stack_fwd_k(const TrainParams& p, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    constexpr int lda = ((16 * KS + 29) / 32) * 32 + 2, ldg = ((64 + 29) / 32) * 32 + 2;
    const size_t lds = (size_t)(32 * lda + 4 * 16 * ldg) * sizeof(float) + 64;
    int G = k.stack_wgs > 0 ? k.stack_wgs : qpn_num_cus() * 2;
    if (G > q.total) G = q.total;
    if (G > 1024) G = 1024;                                      // (scratch rows: one pair per workgroup)
    QPN_HIP(hipFuncSetAttribute((const void*)k_stack_fwd<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    StackQ qq = q; qq.nq = G / 8 < 1 ? 1 : (G / 8 < SQ_NQ ? G / 8 : SQ_NQ);      // every sub-queue needs a puller (a workgroup's home: (blockIdx / 8) % nq)
    sq_arm(qq, k);
    hipLaunchKernelGGL((k_stack_fwd<KS>), dim3(G), dim3(256), lds, stream, p, qq);
    return QPN_OK;
}
int qpn_launch_stack_fwd(const TrainParams& p, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    return p.hoist ? launch_stack_fwd_k<8>(p, q, k, stream) : launch_stack_fwd_k<11>(p, q, k, stream);
}

bool qpn_stack_bwd_fits(const TrainParams& p) {
    return qpn_stack_fwd_fits(p) && p.Ap <= 64 && (int64_t)p.B * p.BL * p.LC < (1ll << 31) && (int64_t)p.N1 * (p.LC > 2 * p.C ? p.LC : 2 * p.C) < (1ll << 32);
}

template <int NTK>
static int launch_stack_bwd_k(const TrainParams& p, const TrainBwd& bw, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    constexpr int ldx = ((64 + 29) / 32) * 32 + 2, ldz = ((128 + 29) / 32) * 32 + 2, ldo = ((16 * NTK + 29) / 32) * 32 + 2;
    const size_t lds = (size_t)(8 * 16 * ldx + 16 * ldz + 16 * ldo) * sizeof(float) + 128 + 256;      // + control words, + the hoist form's G partials
    // 1.5 workgroups per CU: the skip / post-net weight gradients run on the side stream while this launch is resident (qpn_launch_bwd), and a
    // launch that fills every CU twice over leaves them no room -- measured on the overlapped step: 0.846 ms with 2 per CU, 0.776 with 1.5,
    // 0.778 with 1, 0.790 for the eight per-layer launches
    // (round 5, the K = 128 form: 1.25 per CU -- 320 -- measured 1411-1415 steps/s against 1375 at 384 and 1365 at 256; counts whose groups of eight
    //  do not divide evenly over the eight sub-queues -- 304, 336, 352 -- are 3-4 % slower)
    int G = k.stack_wgs_bwd > 0 ? k.stack_wgs_bwd : (NTK == 8 ? qpn_num_cus() * 5 / 4 / 64 * 64 : qpn_num_cus() * 3 / 2);
    if (G < 8) G = qpn_num_cus();
    if (G > q.total) G = q.total;
    if (G > 1024) G = 1024;
    QPN_HIP(hipFuncSetAttribute((const void*)k_stack_bwd<NTK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    StackQ qq = q; qq.nq = G / 8 < 1 ? 1 : (G / 8 < SQ_NQ ? G / 8 : SQ_NQ);
    sq_arm(qq, k);
    hipLaunchKernelGGL((k_stack_bwd<NTK>), dim3(G), dim3(256), lds, stream, p, bw, qq);
    return QPN_OK;
}
int qpn_launch_stack_bwd(const TrainParams& p, const TrainBwd& bw, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    return p.hoist ? launch_stack_bwd_k<8>(p, bw, q, k, stream) : launch_stack_bwd_k<11>(p, bw, q, k, stream);
}
