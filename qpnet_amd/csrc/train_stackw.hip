// train_stackw.hip -- the residual stack's work queue (train_stackq.h, train_stack.hip) with ONE WAVE PER 16-ROW TILE (round 6).
//
// k_stack_fwd / k_stack_bwd split a tile over the four waves of a workgroup: every wave holds a quarter of the layer's weights in registers, the
// waves meet at three barriers per tile (the gate outputs of all four are every wave's next operand) and, because a batch-1 chunk has only
// ~1300 tiles per layer (the width of the dependency graph: a tile needs its producers' rows), a SIMD never holds more than ~1.25
// of those waves -- each runs one serial chain of LDS staging, 88 MFMAs and epilogue per tile, and the matrix cores idle for 0.6-0.7 of it
// (profiles/r05_pmc_by_kernel.json: 0.28 busy backward, 0.41 forward).  Here a wave owns a WHOLE tile:
//   * the layer's big weight block (backward: W1, 128 x 128 = 64 KB) lives in LDS, two layers' worth (the frontier of the queue spans two layers),
//     and is read as the MFMA's A operand, one ds_read_b128 per four MFMAs; the small one (Wr, 16 KB) comes straight from L1 / L2;
//   * both products are computed TRANSPOSED (out^T = W^T . in^T): the tile's data is the B operand, whose lane layout -- lane = row (lane & 15),
//     lane >> 4 = which four consecutive channels of every 16 -- is also the layout the result comes back in.  With the weights packed in a
//     matching k order (tr: fragA_pack) the gate outputs feed the second product from the registers they were computed in: no LDS transpose,
//     no barrier, no other wave.  Rows are loaded and stored as 16-byte pieces (16 rows x 64 B per instruction);
//   * 320 MFMAs per tile and wave instead of 88: measured 0.56 (one wave per SIMD) to 0.75 (two) of the fp32 matrix rate with everything else
//     of the tile in place (tools/wave_tile_bench.hip, profiles/r06_wave_tile_bench.txt), against 0.28.
// The queue protocol is unchanged (positions, tickets, flags, write-through hand-off, bounded waits): a wave publishes its tile before it looks
// at the next position, so every wait is for positions below everything the waiter holds back.
#include "train_stackq.h"

#define SW_W1F4 4096                                // float4 words of one layer's W1 image: 8 k-blocks x 8 m-tiles x 64 lanes
#define SW_WRF4 1024                                // ... of its Wr image: 4 x 4 x 64
#define SW_READY 0x8000u
// dev aid, TIMING ONLY (results are invalid): -DSW_EXP=<bits> removes parts of the tile, to see what each costs --
// 1 the LDS image protocol (no acquire / load / release: whatever the slot holds), 2 the aux reductions and their atomics, 4 the flag waits, 8 the tap scatter atomics,
// 16 the second product's MFMAs, 32 the dZ row stores
#ifndef SW_EXP
#define SW_EXP 0
#endif

// ---- the two LDS weight slots.  state = (layer + 1) << 16 | ready << 15 | users.  A wave that needs layer l looks at slot l & 1: the layer is there ->
// users + 1 (and wait for `ready`); another layer with no user left -> claim the slot (users = 1, not ready), load the image, set ready; another
// layer still in use -> wait (its users hold no lock while they wait for anything else: they are inside a tile's arithmetic).
__device__ __forceinline__ int sw_acquire(unsigned* state, int layer, int lane) {
    int res = 0;
    if (lane == 0) {
        const unsigned want = (unsigned)(layer + 1) << 16;
        for (;;) {
            const unsigned old = __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old >> 16) == (unsigned)(layer + 1)) {
                if (atomicCAS(state, old, old + 1u) == old) { res = 0; break; }
            } else if ((old & 0x7fffu) == 0u && (old == 0u || (old & SW_READY))) {      // (never a slot whose image is still on its way)
                if (atomicCAS(state, old, want | 1u) == old) { res = 1; break; }
            } else __builtin_amdgcn_s_sleep(4);
        }
    }
    return sq_rfl(res);
}
__device__ __forceinline__ void sw_wait_ready(unsigned* state, int lane) {
    int ok = 0;
    do {
        if (lane == 0) ok = (__hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & SW_READY) ? 1 : 0;
        ok = sq_rfl(ok);
        if (!ok) __builtin_amdgcn_s_sleep(2);
    } while (!ok);
}
// one wave starts the copy of a 64 KB image (global, fragment order) into its slot: 64 LDS-DMA pieces of 1 KB (64 lanes x 16 bytes, lane-linear -- the
// image's own order), no register in between; they complete behind the wave's next s_waitcnt vmcnt(0), after which it sets the slot's ready bit
__device__ __forceinline__ void sw_start_image(float4* slot, const float4* __restrict__ src, int lane) {
#pragma unroll 2
    for (int k = 0; k < SW_W1F4 / 64; ++k)
        __builtin_amdgcn_global_load_lds((const void*)(src + (size_t)k * 64 + lane), (__attribute__((address_space(3))) void*)(slot + k * 64), 16, 0, 0);
}
__device__ __forceinline__ float sw_xor_dpp8(float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x128, 0xf, 0xf, true)); }     // row_ror:8 = lane ^ 8
__device__ __forceinline__ float sw_xor_dpp2(float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xf, 0xf, true)); }      // quad_perm [2,3,0,1] = lane ^ 2
__device__ __forceinline__ float sw_xor_dpp1(float a) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, true)); }      // quad_perm [1,0,3,2] = lane ^ 1
// Column sums over the tile's 16 rows (= over lane & 15) of 32 per-lane values, by a halving exchange: at the step of row bit b a lane keeps the half of
// its registers that its own bit selects and adds the partner's copy of that half, so 32 registers become 16, 8, 4, 2 (~3 instructions per pair, 92 pairs'
// worth instead of 32 x 4 full reductions).  Result: o[q], q = 0 / 1, = the sum for register r = 16 b3 + 8 b2 + 4 b1 + 2 b0 + q of the lane's row bits b3..b0.
__device__ __forceinline__ void sw_colsum32(const float (&v)[32], int row, float (&o)[2]) {
    const bool b3 = row & 8, b2 = row & 4, b1 = row & 2, b0 = row & 1;
    float w[16], x[8], y[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = (b3 ? v[r + 16] : v[r]) + sw_xor_dpp8(b3 ? v[r] : v[r + 16]);
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = (b2 ? w[r + 8] : w[r]) + __shfl_xor(b2 ? w[r] : w[r + 8], 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) y[r] = (b1 ? x[r + 4] : x[r]) + sw_xor_dpp2(b1 ? x[r] : x[r + 4]);
#pragma unroll
    for (int r = 0; r < 2; ++r) o[r] = (b0 ? y[r + 2] : y[r]) + sw_xor_dpp1(b0 ? y[r] : y[r + 2]);
}

// ------------------------------------------------------------------------------------------------ backward (aux hoist form: K = 128), one wave per tile
// dynamic LDS: W1 images of two layers (2 x 64 KB) | slot states
// [Opt-in (QPN_STACK_WAVE_BWD=1): correct -- every training test passes with it -- and slower than k_stack_bwd, see the head of this file.  This is the plain
//  form, one tile after the other; a form that requested the next tile's rows a tile ahead was built as well and was slower still (its early look at the
//  producers' flags misses at a window of 1.25 rounds, and a lone wave pays ~180 register moves a tile for the rotation): MEASUREMENTS R6.]
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_stack_bwd_w(TrainParams p, TrainBwd bw, StackQ q) {
    constexpr int C = 64;
    extern __shared__ float4 swm[];
    unsigned* const state = (unsigned*)(swm + 2 * SW_W1F4);
    const int lane = threadIdx.x & 63, wave = sq_rfl(threadIdx.x >> 6);
    const int row = lane & 15, g = lane >> 4;
    const int N1 = p.N1, win0 = N1 - p.BL, U = p.U;
    const unsigned xbytes = (unsigned)N1 * C * 4u;
    const size_t nDX = (size_t)p.B * N1 * C;
    if (threadIdx.x < 2) state[threadIdx.x] = 0u;
    __syncthreads();
    auto rsrc = [&](const float* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, xbytes, 0x00020000); };
    int zero_v; asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const int NQ = q.nq, sub = (blockIdx.x / 8) % NQ;
    unsigned* const head = q.head + sub * TR_QHEAD_STRIDE + zero_v;
    const int eslot = (blockIdx.x * WAVES + wave) & (TR_EB_SLOTS - 1);

    unsigned tk = 0;
    if (lane == 0) tk = atomicAdd(head, 1u);
    SqTile d = sq_take(sq_fetch(q, sq_rfl((int)tk) * NQ + sub));
    while (sq_valid(d)) {
        // the ticket of the position after this one: asked for now, looked at when the tile is done
        unsigned tkn = 0;
        if (lane == 0) tkn = atomicAdd(head, 1u);
        const int layer = sq_layer(d);
        const bool last = sq_last(d), adaptive = (d.meta >> 26) & 1;
        const int n = d.n0 + row;
        const bool in = n < N1;
        const int nn = in ? n : N1 - 1;
        const size_t xb = (size_t)d.xrow * C;
        const unsigned ro = (unsigned)nn * C + 4u * g;                // float offset of this lane's first 16-byte piece in its row
        // ---- rows nobody inside the launch writes: sigma, the gate product, the skip-path gradient, taps, the aux tables, the residual 1x1's weights
        float4 sg4[4], th4[4], dq4[4], pas[4], pat[4], wr[4][4];
        const bool has_dg = in && n >= win0;
        const int nw = nn >= win0 ? nn - win0 : 0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            sg4[mt] = *(const float4*)(p.SG + xb + ro + 16 * mt);
            th4[mt] = *(const float4*)(p.TH + xb + ro + 16 * mt);
            dq4[mt] = *(const float4*)(bw.DGS + (size_t)d.dgs + ((size_t)nw * p.LC + 16 * mt + 4 * g));
        }
        // the frame slot and the upsampling weight of this lane's row: j(n) = (F U - N1 + n) mod U, consecutive over the tile's rows (U >= 16: at most one wrap)
        const int j0 = (int)((unsigned)(p.F * U - N1 + d.n0) % (unsigned)U), jr = j0 + row;
        const int fslot = jr >= U ? 1 : 0;
        const float wjx = p.flat[p.up_w + (jr >= U ? jr - U : jr)];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float* pq = p.PA + d.hrow + fslot * 2 * C + 16 * mt + 4 * g;
            pas[mt] = *(const float4*)pq; pat[mt] = *(const float4*)(pq + C);
        }
        const int tap = adaptive ? (p.TAP + d.tapb)[nn] : n - p.layers[layer].dilation;
        if (!last) {
            const float4* wq = p.wp + p.wrq_f4 + (size_t)layer * SW_WRF4 + lane;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) wr[s4][mt] = wq[(s4 * 4 + mt) * 64];
        }
        // ---- the producers' flags, then the rows they handed over: the two parts of the gradient w.r.t. this layer's output
#if !(SW_EXP & 4)
        if (!sq_wait(q, d.dfirst, d.dn, lane, p.status, 0u)) break;
#endif
        float4 dx[4];
        if (!last) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc(bw.DXA[0] + xb + nDX), (int)((ro + 16 * mt) * 4u), 0, SQ_SC1);
                const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc(bw.DXB[0] + xb + nDX), (int)((ro + 16 * mt) * 4u), 0, SQ_SC1);
                dx[mt] = in ? make_float4(__uint_as_float(a.x) + __uint_as_float(b.x), __uint_as_float(a.y) + __uint_as_float(b.y),
                                          __uint_as_float(a.z) + __uint_as_float(b.z), __uint_as_float(a.w) + __uint_as_float(b.w)) : make_float4(0.f, 0.f, 0.f, 0.f);
                // the sum replaces the own-row part IN PLACE (this tile is those rows' only reader inside the launch; dWr behind the launch reads one array)
                if (in) *(float4*)(bw.DXA[0] + xb + nDX + ro + 16 * mt) = dx[mt];
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dx[mt] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- dg^T = Wr^T . dXout^T  (A: the weights, B: the tile), + the skip-path gradient;  dz = dg * gate'
        f32x4 a1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) a1[mt] = (f32x4){0, 0, 0, 0};
        if (!last) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s4][mt].x, dx[s4].x, a1[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s4][mt].y, dx[s4].y, a1[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s4][mt].z, dx[s4].z, a1[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a1[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[s4][mt].w, dx[s4].w, a1[mt], 0, 0, 0);
            }
        }
        float dz[32];                                               // register r = 16 half + 4 mt + i: gate column 64 half + 16 mt + 4 g + i of row `row`
        float gsum = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float sgv[4] = {sg4[mt].x, sg4[mt].y, sg4[mt].z, sg4[mt].w}, thv[4] = {th4[mt].x, th4[mt].y, th4[mt].z, th4[mt].w};
            const float dqv[4] = {dq4[mt].x, dq4[mt].y, dq4[mt].z, dq4[mt].w};
            const float pasv[4] = {pas[mt].x, pas[mt].y, pas[mt].z, pas[mt].w}, patv[4] = {pat[mt].x, pat[mt].y, pat[mt].z, pat[mt].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float dgv = a1[mt][i] + (has_dg ? dqv[i] : 0.f);
                float zs, zt;
                tr_gate_bwd(dgv, in ? sgv[i] : 0.f, in ? thv[i] : 0.f, zs, zt);
                dz[4 * mt + i] = zs; dz[16 + 4 * mt + i] = zt;
                gsum += zs * pasv[i] + zt * patv[i];
            }
        }
        // ---- the frame-rate aux term's backward (tr_aux_bwd's three outputs in this layout): G per row, D per frame and gate column, E per gate column
#if !(SW_EXP & 2)
        {
            gsum += __shfl_xor(gsum, 16); gsum += __shfl_xor(gsum, 32);
            if (g == 0 && in) bw.GW[(size_t)d.xrow + n] = gsum;
            float e2[2], d0[2], d1[2], y[32];
            sw_colsum32(dz, row, e2);
            const bool two = __any(fslot != 0);                       // the tile reaches into a second frame (U >= 16: never a third)
#pragma unroll
            for (int r = 0; r < 32; ++r) y[r] = fslot ? 0.f : wjx * dz[r];
            sw_colsum32(y, row, d0);
            const int col = 64 * ((row >> 3) & 1) + 16 * ((row >> 1) & 3) + 4 * g + 2 * (row & 1);      // (+ q): the column the lane's two sums belong to
            float* dp = bw.DPA + d.hrow + col;
            float* ep = bw.EB + (size_t)(layer * TR_EB_SLOTS + eslot) * 2 * C + col;
            atomicAdd(dp, d0[0]); atomicAdd(dp + 1, d0[1]);
            atomicAdd(ep, e2[0]); atomicAdd(ep + 1, e2[1]);
            if (two) {
#pragma unroll
                for (int r = 0; r < 32; ++r) y[r] = fslot ? wjx * dz[r] : 0.f;
                sw_colsum32(y, row, d1);
                atomicAdd(dp + 2 * C, d1[0]); atomicAdd(dp + 2 * C + 1, d1[1]);
            }
        }
#endif
        // ---- dZ rows to memory (the gate contraction's weight gradient behind this launch reads them)
#if !(SW_EXP & 32)
        if (in) {
            float* zr = bw.DZ + (size_t)d.xrow * 2 * C + (size_t)n * 2 * C + 4 * g;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                *(float4*)(zr + 16 * mt) = make_float4(dz[4 * mt], dz[4 * mt + 1], dz[4 * mt + 2], dz[4 * mt + 3]);
                *(float4*)(zr + C + 16 * mt) = make_float4(dz[16 + 4 * mt], dz[16 + 4 * mt + 1], dz[16 + 4 * mt + 2], dz[16 + 4 * mt + 3]);
            }
        }
#endif
        // ---- d[x_cur | x_past]^T = W1^T . dZ^T: the weights from the layer's LDS image
        unsigned* const st = state + (layer & 1);
        float4* const slot = swm + (layer & 1) * SW_W1F4;
#if !(SW_EXP & 1)
        if (sw_acquire(st, layer, lane)) {
            sw_start_image(slot, p.wp + p.w1q_f4 + (size_t)layer * SW_W1F4, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) atomicOr(st, SW_READY);
        } else sw_wait_ready(st, lane);
#endif
        f32x4 a2[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) a2[m] = (f32x4){0, 0, 0, 0};
        {
            const float4* W = slot + lane;
            float4 wb[2][8];
#pragma unroll
            for (int m = 0; m < 8; ++m) wb[0][m] = W[m * 64];
#pragma unroll
            for (int s4 = 0; s4 < ((SW_EXP & 16) ? 1 : 8); ++s4) {
                if (s4 + 1 < 8) {
#pragma unroll
                    for (int m = 0; m < 8; ++m) wb[(s4 + 1) & 1][m] = W[((s4 + 1) * 8 + m) * 64];
                }
#pragma unroll
                for (int m = 0; m < 8; ++m) a2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s4 & 1][m].x, dz[4 * s4 + 0], a2[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) a2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s4 & 1][m].y, dz[4 * s4 + 1], a2[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) a2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s4 & 1][m].z, dz[4 * s4 + 2], a2[m], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < 8; ++m) a2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s4 & 1][m].w, dz[4 * s4 + 3], a2[m], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if !(SW_EXP & 1)
        if (lane == 0) atomicSub(st, 1u);
#endif
        // ---- outputs: own-row part (+ the residual path) and the tap part of the gradient w.r.t. this layer's INPUT, handed to the layer below
        {
            const unsigned off = in ? ((unsigned)n * C + 4u * g) * 4u : SQ_OOB;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const u32x4 v = {__float_as_uint(a2[mt][0] + dx[mt].x), __float_as_uint(a2[mt][1] + dx[mt].y), __float_as_uint(a2[mt][2] + dx[mt].z), __float_as_uint(a2[mt][3] + dx[mt].w)};
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(bw.DXA[0] + xb), (int)(in ? off + 64u * mt : SQ_OOB), 0, SQ_SC1);
            }
            if (!adaptive) {                                        // fixed block: the tap row n - dilation has this one writer
                const unsigned offb = (in && tap >= 0) ? ((unsigned)tap * C + 4u * g) * 4u : SQ_OOB;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const u32x4 v = {__float_as_uint(a2[4 + mt][0]), __float_as_uint(a2[4 + mt][1]), __float_as_uint(a2[4 + mt][2]), __float_as_uint(a2[4 + mt][3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(bw.DXB[0] + xb), (int)((in && tap >= 0) ? offb + 64u * mt : SQ_OOB), 0, SQ_SC1);
                }
            } else if (in && !(SW_EXP & 8)) {                       // gather backward (collisions): float atomics at the memory side
                float* db = bw.DXB[0] + xb + (size_t)tap * C + 4 * g;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int i = 0; i < 4; ++i) atomicAdd(db + 16 * mt + i, a2[4 + mt][i]);
            }
        }
        // ---- published once everything above has completed (vmcnt counts the atomics too); layer 0's input gradient feeds later kernels only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (layer > 0 && lane == 0) sq_st(q.flags + sq_fidx((unsigned)d.pos), q.epoch_pub);
        d = sq_take(sq_fetch(q, sq_rfl((int)tkn) * NQ + sub));
    }
}

// ------------------------------------------------------------------------------------------------ forward: four waves per tile, transposed products
// The workgroup-per-tile forward (k_stack_fwd) with the layout above instead of its LDS staging: every wave requests the tile's rows itself, as the B operand
// (16-byte pieces: lane = row, lane >> 4 = which four channels of every sixteen; the pitch-tap rows are a gather at row granularity, which this layout IS),
// computes the gate pre-activations of ITS sixteen channels for all sixteen rows (64 MFMAs, weights = A operand, resident in registers), adds bias and the
// frame-rate aux term elementwise, and stores sigma and the gate product straight from its registers.  The one exchange a tile needs -- every wave's residual
// 1x1 contracts over all 64 gate channels -- is a 4 KB LDS tile and ONE barrier (k_stack_fwd: the A tile staged through LDS, three barriers, ~90 LDS
// instructions a wave).  Queue protocol, ticket ring and hand-over as there.
// dynamic LDS: Gs[2][16][72] | control words
__global__ __launch_bounds__(256, 2) void k_stack_fwd_t(TrainParams p, StackQ q) {
    constexpr int C = 64, GLD = 72;
    extern __shared__ float smt[];
    float* const Gs = smt;
    int* const ctl = (int*)(smt + 2 * 16 * GLD);      // [0..3] position ring, [4..7] / [12..15] per wave: the next tile's producers were not all published, [8] arrivals at the publish point
    const int N1 = p.N1, U = p.U;
    const int tid = threadIdx.x, lane = tid & 63, wave = sq_rfl(tid >> 6);
    const int row = lane & 15, g = lane >> 4, cw = 16 * wave + 4 * g;      // this lane: row `row` of the tile, channels cw .. cw + 3 of its wave's sixteen
    const unsigned xbytes = (unsigned)N1 * C * 4u;
    const size_t xlayer = (size_t)p.B * N1 * C;
    auto xrsrc = [&](const float* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, xbytes, 0x00020000); };

    float4 w1s[8], w1t[8], wra[4], bs4, bt4, bb4;
    auto load_weights = [&](int l) {
        const TrLayer ly = p.layers[l];
        const float4* W1 = p.wp + p.w1p_f4 + (size_t)l * SW_W1F4 + lane; const float4* Wr = p.wp + p.wrp_f4 + (size_t)l * SW_WRF4 + lane;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) { w1s[s4] = W1[(s4 * 8 + wave) * 64]; w1t[s4] = W1[(s4 * 8 + 4 + wave) * 64]; }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) wra[s4] = Wr[(s4 * 4 + wave) * 64];
        bs4 = *(const float4*)(p.bp + ly.bias1 + cw); bt4 = *(const float4*)(p.bp + ly.bias1 + C + cw); bb4 = *(const float4*)(p.bp + ly.biasr + cw);
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): waited for on the layer change's own path (see k_stack_fwd)
    };
    u32x4 xc[4], xp[4]; float4 pas, pat; float wjx = 0.f;
    int tp = 0;
    auto load_tap = [&](const SqTile& d, int& out) { const int n = d.n0 + row; out = (p.TAP + d.tapb)[n < N1 ? n : N1 - 1]; };
    auto load_rows = [&](const SqTile& d) {
        const int n = d.n0 + row, nn = n < N1 ? n : N1 - 1;
        const auto rs = xrsrc(p.X + (size_t)d.xrow * C);
        const unsigned oc = ((unsigned)nn * C + 4u * g) * 4u, op = ((unsigned)tp * C + 4u * g) * 4u;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            xc[s4] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(oc + 64u * s4), 0, SQ_SC1);
            xp[s4] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(op + 64u * s4), 0, SQ_SC1);
        }
        // the frame slot and the upsampling weight of this lane's row: j(n) = (F U - N1 + n) mod U, consecutive over the tile's rows (U >= 16: at most one wrap)
        const int j0 = (int)((unsigned)(p.F * U - N1 + d.n0) % (unsigned)U), jr = j0 + row;
        const float* pq = p.PA + d.hrow + (jr >= U ? 2 * C : 0) + cw;
        pas = *(const float4*)pq; pat = *(const float4*)(pq + C);
        wjx = p.flat[p.up_w + (jr >= U ? jr - U : jr)];
        __builtin_amdgcn_sched_barrier(0);
    };
    auto publishes = [&](const SqTile& d) { return sq_valid(d) && !sq_last(d); };     // (nothing reads the last block's residual output: no rows, no flag)

    int zero_v; asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const int NQ = q.nq, sub = (blockIdx.x / 8) % NQ;
    unsigned* const head = q.head + sub * TR_QHEAD_STRIDE + zero_v;
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
    load_rows(cur);
    load_tap(next, tp);
    bool cur_published = false, rows_youngest = true;      // rows_youngest: nothing was issued behind the rows of `cur` (first tile, or the slow path requested them last)
    for (int it = 0;; ++it) {
        float* const G = Gs + (it & 1) * 16 * GLD;
        const bool last = sq_last(cur);
        const int n = cur.n0 + row;
        const bool in = n < N1;
        if (sq_layer(cur) != lw) { lw = sq_layer(cur); load_weights(lw); rows_youngest = true; }
        // ---- T1: the rows of cur are here (the three row stores of the previous tile, issued behind their request, stay in flight)
        if (rows_youngest) __builtin_amdgcn_s_waitcnt(0x0F70); else __builtin_amdgcn_s_waitcnt(0x0F73);      // vmcnt(0) / vmcnt(3)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(xc[s4]), "+v"(xp[s4]));
        asm volatile("" : "+v"(pas.w), "+v"(pat.w), "+v"(wjx));
        // the request group of this trip: the NEXT tile's producers' flags, the tap rows of the tile after it, the ticket of the one three ahead
        const int fn = next.dn, fnm1 = fn > 0 ? fn - 1 : 0, fbase = fn > 0 ? next.dfirst : cur.pos;      // (no producers: a word of its own)
        unsigned fv0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), fv1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
        int tpn; load_tap(nn, tpn);
        unsigned rtk = 0;
        if (tid == 0) rtk = atomicAdd(head, 1u);
        asm volatile("" ::: "memory");
        // ---- T2: z^T = W1^T . [x_cur | x_past]^T for this wave's sixteen channels (sigma and tanh rows), + bias + aux term; gate
        f32x4 zs = (f32x4){0, 0, 0, 0}, zt = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const u32x4 v = s4 < 4 ? xc[s4] : xp[s4 - 4];
            const float b0 = in ? __uint_as_float(v.x) : 0.f, b1 = in ? __uint_as_float(v.y) : 0.f, b2 = in ? __uint_as_float(v.z) : 0.f, b3 = in ? __uint_as_float(v.w) : 0.f;
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].x, b0, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].x, b0, zt, 0, 0, 0);
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].y, b1, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].y, b1, zt, 0, 0, 0);
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].z, b2, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].z, b2, zt, 0, 0, 0);
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].w, b3, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].w, b3, zt, 0, 0, 0);
        }
        const u32x4 xres = wave == 0 ? xc[0] : wave == 1 ? xc[1] : wave == 2 ? xc[2] : xc[3];      // x_cur of this lane's own channels: the residual path
        const float bsv[4] = {bs4.x, bs4.y, bs4.z, bs4.w}, btv[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
        const float psv[4] = {pas.x, pas.y, pas.z, pas.w}, ptv[4] = {pat.x, pat.y, pat.z, pat.w};
        float sgv[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sg = sq_sigmoid((zs[i] + bsv[i]) + wjx * psv[i]), th = sq_tanh((zt[i] + btv[i]) + wjx * ptv[i]);
            sgv[i] = sg; gv[i] = sg * th;
        }
        // ---- T3: everything older has completed: the previous tile's rows (-> the wave that gets here last publishes it), this trip's request group
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" : "+v"(fv0), "+v"(fv1), "+v"(tpn), "+v"(rtk));
        const bool pub_prev = publishes(prev) && !cur_published;
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(ctl + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & 3) == 3 && pub_prev) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
        }
        bool ready = fn == 0 || (fn <= 128 && __all(fv0 == q.epoch && fv1 == q.epoch));
        if (lane == 0) ctl[4 + wave] = ready ? 0 : 1;
        if (tid == 0) ctl[(it + 3) & 3] = (int)(rtk * NQ + sub);
        *(float4*)(G + row * GLD + cw) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        TR_LDS_BARRIER();
        // ---- T4: every wave knows whether all four found the flags; yes -> the next tile's rows are requested now
        const SqRaw raw3 = sq_fetch(q, sq_rfl(ctl[(it + 3) & 3]));
        int any_slow = sq_rfl(ctl[4] | ctl[5] | ctl[6] | ctl[7]);
        if (!any_slow) load_rows(next);
        // ---- T5: out^T = Wr^T . g^T for this wave's sixteen output channels, + bias + x_cur
        f32x4 ar = (f32x4){0, 0, 0, 0};
        if (!last) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 b4 = *(const float4*)(G + row * GLD + 16 * s4 + 4 * g);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].x, b4.x, ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].y, b4.y, ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].z, b4.z, ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].w, b4.w, ar, 0, 0, 0);
            }
        }
        // ---- T6: this wave's 64 bytes of every row leave: sigma and the gate product (plain: later kernels read them), the block output (write-through)
        auto store_rows = [&](bool with_x) {
            float* sgp = in ? p.SG + (size_t)cur.xrow * C + (size_t)n * C + cw : p.scratch_rows + (size_t)blockIdx.x * 256 + 4 * lane;
            float* thp = in ? p.TH + (size_t)cur.xrow * C + (size_t)n * C + cw : p.scratch_rows + (size_t)blockIdx.x * 256 + 4 * lane;
            *(float4*)sgp = make_float4(sgv[0], sgv[1], sgv[2], sgv[3]);
            *(float4*)thp = make_float4(gv[0], gv[1], gv[2], gv[3]);
            const u32x4 v = {__float_as_uint((ar[0] + bb4.x) + __uint_as_float(xres.x)), __float_as_uint((ar[1] + bb4.y) + __uint_as_float(xres.y)),
                             __float_as_uint((ar[2] + bb4.z) + __uint_as_float(xres.z)), __float_as_uint((ar[3] + bb4.w) + __uint_as_float(xres.w))};
            const unsigned o = (with_x && in && !last) ? ((unsigned)n * C + (unsigned)cw) * 4u : SQ_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(v, xrsrc(p.X + (size_t)cur.xrow * C + xlayer), (int)o, 0, SQ_SC1);
        };
        cur_published = false; rows_youngest = false;
        if (any_slow) {
            // Some wave did not find every flag.  Most such misses are near misses: the waves that missed look ONCE more before the workgroup pays for the hand-over
            if (tid == 0) atomicAdd(q.stats + 2, 1u);
            if (!ready) {
                const unsigned g0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), g1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
                ready = fn <= 128 && __all(g0 == q.epoch && g1 == q.epoch);
            }
            if (lane == 0) ctl[12 + wave] = ready ? 0 : 1;
            store_rows(true);
            TR_LDS_BARRIER();
            any_slow = sq_rfl(ctl[12] | ctl[13] | ctl[14] | ctl[15]);
            if (any_slow) {
                // a producer of the next tile has not published yet: hand over everything this workgroup holds, THEN wait
                if (tid == 0) atomicAdd(q.stats, 1u);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TR_LDS_BARRIER();
                if (tid == 0 && publishes(cur)) sq_st(q.flags + sq_fidx((unsigned)cur.pos), q.epoch_pub);
                cur_published = true;
                if (!ready) sq_wait(q, next.dfirst, next.dn, lane, p.status, 0u);
            }
            load_rows(next);
            rows_youngest = true;
        } else store_rows(true);
        tp = tpn;
        prev = cur; cur = next; next = nn; nn = sq_take(raw3);
        if (!sq_valid(cur)) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_LDS_BARRIER();
    if (tid == 0 && publishes(prev) && !cur_published) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
}

// ------------------------------------------------------------------------------------------------ forward: k_stack_fwd's skeleton, transposed products
// k_stack_fwd<8> with the tile body of k_stack_fwd_t: the tile's rows are still requested ONCE per workgroup as whole 256-byte rows and staged in LDS
// (k_stack_fwd_t's per-wave 64-byte pieces quadruple the row requests and cost it 60 % -- profiles/r06_stackw_probe.txt), but they are read back as the B
// operand (eight ds_read_b128 a wave instead of thirty-two ds_read_b32), the gate epilogue works on registers in the row layout, sigma / gate product / block
// output leave from registers, and the only other LDS traffic is the 4 KB gate tile every wave's residual 1x1 contracts over: two barriers and ~15 LDS
// instructions a wave and tile instead of three and ~90.  Ticket ring, flag look, publish point and hand-over exactly as in k_stack_fwd.
// dynamic LDS: As[2][16][136] | Gs[2][16][72] | control words
__global__ __launch_bounds__(256, 2) void k_stack_fwd_h(TrainParams p, StackQ q) {
    constexpr int C = 64, LDA = 136, GLD = 72;
    extern __shared__ float smh[];
    float* const Gs = smh + 2 * 16 * LDA;
    int* const ctl = (int*)(Gs + 2 * 16 * GLD);       // [0..3] position ring, [4..7] / [12..15] per wave: flags missing, [8] arrivals at the publish point
    const int N1 = p.N1, U = p.U;
    const int tid = threadIdx.x, lane = tid & 63, wave = sq_rfl(tid >> 6);
    const int srow = tid >> 4, sc4 = tid & 15;                    // staging: thread -> (row, 16-byte piece) of the tile
    const int row = lane & 15, g = lane >> 4, cw = 16 * wave + 4 * g;      // arithmetic: lane -> row `row`, channels cw .. cw + 3 of its wave's sixteen
    const unsigned xbytes = (unsigned)N1 * C * 4u;
    const size_t xlayer = (size_t)p.B * N1 * C;
    auto xrsrc = [&](const float* base) { return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, xbytes, 0x00020000); };

    float4 w1s[8], w1t[8], wra[4], bs4, bt4, bb4;
    auto load_weights = [&](int l) {
        const TrLayer ly = p.layers[l];
        const float4* W1 = p.wp + p.w1p_f4 + (size_t)l * SW_W1F4 + lane; const float4* Wr = p.wp + p.wrp_f4 + (size_t)l * SW_WRF4 + lane;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) { w1s[s4] = W1[(s4 * 8 + wave) * 64]; w1t[s4] = W1[(s4 * 8 + 4 + wave) * 64]; }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) wra[s4] = Wr[(s4 * 4 + wave) * 64];
        bs4 = *(const float4*)(p.bp + ly.bias1 + cw); bt4 = *(const float4*)(p.bp + ly.bias1 + C + cw); bb4 = *(const float4*)(p.bp + ly.biasr + cw);
        __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): waited for on the layer change's own path (see k_stack_fwd)
    };
    int tp = 0; float4 rc, rp, rpas, rpat; float rwjx = 0.f;                 // the next tile's rows (this thread's pieces) and aux operands (this lane's): requested at the end of a trip, used in the next
    auto load_tap = [&](const SqTile& d, int& out) { const int n = d.n0 + srow; out = (p.TAP + d.tapb)[n < N1 ? n : N1 - 1]; };
    auto load_rows = [&](const SqTile& d) {
        const int n = d.n0 + srow, nn = n < N1 ? n : N1 - 1;
        const auto rs = xrsrc(p.X + (size_t)d.xrow * C);
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((unsigned)nn * C + 4u * sc4) * 4u), 0, SQ_SC1);
        const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((unsigned)tp * C + 4u * sc4) * 4u), 0, SQ_SC1);
        rc = make_float4(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
        rp = make_float4(__uint_as_float(b4.x), __uint_as_float(b4.y), __uint_as_float(b4.z), __uint_as_float(b4.w));
        // the frame slot and the upsampling weight of this LANE's row: j(n) = (F U - N1 + n) mod U, consecutive over the tile's rows (U >= 16: at most one wrap)
        const int j0 = (int)((unsigned)(p.F * U - N1 + d.n0) % (unsigned)U), jr = j0 + row;
        const float* pq = p.PA + d.hrow + (jr >= U ? 2 * C : 0) + cw;
        rpas = *(const float4*)pq; rpat = *(const float4*)(pq + C);
        rwjx = p.flat[p.up_w + (jr >= U ? jr - U : jr)];
        __builtin_amdgcn_sched_barrier(0);                       // (all requests before anything waits for one of them)
    };
    auto store_rows = [&](const SqTile& d, float* As) {
        asm volatile("" : "+v"(rpas.w), "+v"(rpat.w), "+v"(rwjx));      // (every request of the group is taken up here: see k_stack_fwd)
        const bool in = d.n0 + srow < N1;
        float* dd = As + (size_t)srow * LDA + 4 * sc4;
        // (component by component: a select between two float4 OBJECTS becomes a load through a selected address, i.e. both live in scratch memory)
        *(float4*)dd = make_float4(in ? rc.x : 0.f, in ? rc.y : 0.f, in ? rc.z : 0.f, in ? rc.w : 0.f);
        *(float4*)(dd + C) = make_float4(in ? rp.x : 0.f, in ? rp.y : 0.f, in ? rp.z : 0.f, in ? rp.w : 0.f);
    };
    u32x4 xkeep = {0u, 0u, 0u, 0u};
    auto publishes = [&](const SqTile& d) { return sq_valid(d) && !sq_last(d); };

    int zero_v; asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const int NQ = q.nq, sub = (blockIdx.x / 8) % NQ;
    unsigned* const head = q.head + sub * TR_QHEAD_STRIDE + zero_v;
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
    load_rows(cur);
    store_rows(cur, smh);
    load_tap(next, tp);
    bool cur_published = false;
    for (int it = 0;; ++it) {
        const float* As = smh + (it & 1) * 16 * LDA;
        float* const G = Gs + (it & 1) * 16 * GLD;
        const bool last = sq_last(cur);
        const int n = cur.n0 + row;
        const bool in = n < N1;
        unsigned rtk = 0;                                         // the position three tiles ahead (the raw ticket: see k_stack_fwd)
        if (tid == 0) rtk = atomicAdd(head, 1u);
        if (sq_layer(cur) != lw) { lw = sq_layer(cur); load_weights(lw); }
        int tpn; load_tap(nn, tpn);
        TR_LDS_BARRIER();                                          // B1: the tile's staged rows are complete
        const bool pub_prev = publishes(prev) && !cur_published;
        float4 xb[8];
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) xb[s4] = *(const float4*)(As + (size_t)row * LDA + 16 * s4 + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
        SQ_PRIO(0);
        f32x4 zs = (f32x4){0, 0, 0, 0}, zt = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].x, xb[s4].x, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].x, xb[s4].x, zt, 0, 0, 0);
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].y, xb[s4].y, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].y, xb[s4].y, zt, 0, 0, 0);
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].z, xb[s4].z, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].z, xb[s4].z, zt, 0, 0, 0);
            zs = __builtin_amdgcn_mfma_f32_16x16x4f32(w1s[s4].w, xb[s4].w, zs, 0, 0, 0); zt = __builtin_amdgcn_mfma_f32_16x16x4f32(w1t[s4].w, xb[s4].w, zt, 0, 0, 0);
        }
        SQ_PRIO(2);
        __builtin_amdgcn_sched_barrier(0);
        // publish point: younger than the previous tile's row stores are only this trip's tap load and, in wave 0, the ticket
        __builtin_amdgcn_s_waitcnt(0x0F71);                        // vmcnt(1)
        asm volatile("" :: "v"(xkeep));
        if (lane == 0) {
            const int old = __hip_atomic_fetch_add(ctl + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((old & 3) == 3 && pub_prev) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
        }
        __builtin_amdgcn_sched_barrier(0);
        // x_cur of this lane's own channels (the residual path): read again from the staged tile -- selecting xb[wave] makes the array a memory object
        // (hipcc folds the select chain into ONE load through a selected address: the whole array went to scratch, every MFMA group behind a scratch load)
        const float4 xres = *(const float4*)(As + (size_t)row * LDA + cw);
        const float bsv[4] = {bs4.x, bs4.y, bs4.z, bs4.w}, btv[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
        const float psv[4] = {rpas.x, rpas.y, rpas.z, rpas.w}, ptv[4] = {rpat.x, rpat.y, rpat.z, rpat.w};
        const float wjx = rwjx;
        float sgv[4], gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sg = sq_sigmoid((zs[i] + bsv[i]) + wjx * psv[i]), th = sq_tanh((zt[i] + btv[i]) + wjx * ptv[i]);
            sgv[i] = sg; gv[i] = sg * th;
        }
        *(float4*)(G + row * GLD + cw) = make_float4(gv[0], gv[1], gv[2], gv[3]);
        if (tid == 0) ctl[(it + 3) & 3] = (int)(rtk * NQ + sub);
        TR_LDS_BARRIER();                                          // B2: the gate tile is complete
        const SqRaw raw3 = sq_fetch(q, sq_rfl(ctl[(it + 3) & 3]));
        const int fn = next.dn, fnm1 = fn > 0 ? fn - 1 : 0, fbase = fn > 0 ? next.dfirst : cur.pos;
        unsigned fv0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), fv1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
        asm volatile("" ::: "memory");
        f32x4 ar = (f32x4){0, 0, 0, 0};
        if (!last) {
            float4 gb[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) gb[s4] = *(const float4*)(G + row * GLD + 16 * s4 + 4 * g);
            __builtin_amdgcn_sched_barrier(0);
            SQ_PRIO(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].x, gb[s4].x, ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].y, gb[s4].y, ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].z, gb[s4].z, ar, 0, 0, 0);
                ar = __builtin_amdgcn_mfma_f32_16x16x4f32(wra[s4].w, gb[s4].w, ar, 0, 0, 0);
            }
            SQ_PRIO(2);
        }
        asm volatile("" : "+v"(fv0), "+v"(fv1));                   // (the flag words are looked at HERE)
        bool ready = fn == 0 || (fn <= 128 && __all(fv0 == q.epoch && fv1 == q.epoch));
        // sigma / the gate product leave BEHIND the look at the flags (vmcnt counts in order); plain stores: later kernels read them
        asm volatile("" ::: "memory");
        {
            float* const scr = p.scratch_rows + (size_t)blockIdx.x * 256 + 4 * lane;
            float* sgp = in ? p.SG + (size_t)cur.xrow * C + (size_t)n * C + cw : scr;
            float* thp = in ? p.TH + (size_t)cur.xrow * C + (size_t)n * C + cw : scr;
#if !(SQ_EXP & 64)
            *(float4*)sgp = make_float4(sgv[0], sgv[1], sgv[2], sgv[3]);
            *(float4*)thp = make_float4(gv[0], gv[1], gv[2], gv[3]);
#endif
        }
        if (lane == 0) ctl[4 + wave] = ready ? 0 : 1;
        TR_LDS_BARRIER();                                          // B3: every wave knows whether all four found their flags
        int any_slow = sq_rfl(ctl[4] | ctl[5] | ctl[6] | ctl[7]);
        cur_published = false;
        const u32x4 xo = {__float_as_uint((ar[0] + bb4.x) + xres.x), __float_as_uint((ar[1] + bb4.y) + xres.y), __float_as_uint((ar[2] + bb4.z) + xres.z), __float_as_uint((ar[3] + bb4.w) + xres.w)};
        auto store_x = [&](bool go) {                             // the block output of cur -> X[l + 1], write-through, from the registers it was computed in
            const unsigned o = (go && in && !(SQ_EXP & 128)) ? ((unsigned)n * C + (unsigned)cw) * 4u : SQ_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(xo, xrsrc(p.X + (size_t)cur.xrow * C + xlayer), (int)o, 0, SQ_SC1);
            xkeep = xo;
        };
        if (any_slow) {
            if (tid == 0) atomicAdd(q.stats + 2, 1u);
            if (!ready) {
                const unsigned g0 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane < fnm1 ? lane : fnm1)))), g1 = sq_ld(q.flags + sq_fidx((unsigned)(fbase + (lane + 64 < fnm1 ? lane + 64 : fnm1))));
                ready = fn <= 128 && __all(g0 == q.epoch && g1 == q.epoch);
            }
            if (lane == 0) ctl[12 + wave] = ready ? 0 : 1;
            TR_LDS_BARRIER();
            any_slow = sq_rfl(ctl[12] | ctl[13] | ctl[14] | ctl[15]);
        }
        if (any_slow) {
            // a producer of the next tile has not published yet: hand over everything this workgroup holds, THEN wait
            if (tid == 0) atomicAdd(q.stats, 1u);
            store_x(publishes(cur));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TR_LDS_BARRIER();
            if (tid == 0 && publishes(cur)) sq_st(q.flags + sq_fidx((unsigned)cur.pos), q.epoch_pub);
            cur_published = true;
            if (!ready) sq_wait(q, next.dfirst, next.dn, lane, p.status, 0u);
        }
        load_rows(next);                                           // ONE request site (see k_stack_fwd)
        store_x(publishes(cur) && !any_slow);                     // ... and the block output leaves BEHIND them (vmcnt counts in order)
        store_rows(next, smh + ((it + 1) & 1) * 16 * LDA);
        tp = tpn;
        prev = cur; cur = next; next = nn; nn = sq_take(raw3);
        if (!sq_valid(cur)) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TR_LDS_BARRIER();
    if (tid == 0 && publishes(prev) && !cur_published) sq_st(q.flags + sq_fidx((unsigned)prev.pos), q.epoch_pub);
}

// ------------------------------------------------------------------------------------------------ host side
bool qpn_stack_bwd_w_fits(const TrainParams& p) {
    return p.hoist && p.C == 64 && p.Ktp == 128 && p.w1q_f4 >= 0 && p.wrq_f4 >= 0 && p.U >= 16;
}

template <int WAVES>
static int launch_bwd_w(const TrainParams& p, const TrainBwd& bw, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    const size_t lds = (size_t)2 * SW_W1F4 * sizeof(float4) + 64;
    int G = k.stack_wgs_bwd > 0 ? k.stack_wgs_bwd : qpn_num_cus();
    if (G < 8) G = 8;
    if (G > 1024) G = 1024;
    QPN_HIP(hipFuncSetAttribute((const void*)k_stack_bwd_w<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    StackQ qq = q; qq.nq = G / 8 < 1 ? 1 : (G / 8 < SQ_NQ ? G / 8 : SQ_NQ);
    qq.epoch_pub = q.epoch; qq.spin_limit = SQ_SPIN_LIMIT;
#ifdef QPN_TESTING
    if (k.test_stack_gives_up) { qq.epoch_pub = q.epoch ^ 0x55555555u; qq.spin_limit = 2000u; }
#endif
    hipLaunchKernelGGL((k_stack_bwd_w<WAVES>), dim3(G), dim3(64 * WAVES), lds, stream, p, bw, qq);
    return QPN_OK;
}
bool qpn_stack_fwd_t_fits(const TrainParams& p) {
    return p.hoist && p.C == 64 && p.Ktp == 128 && p.w1p_f4 >= 0 && p.wrp_f4 >= 0 && p.U >= 16;
}
int qpn_launch_stack_fwd_t(const TrainParams& p, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    const size_t lds = (size_t)2 * 16 * 72 * sizeof(float) + 64;
    int G = k.stack_wgs > 0 ? k.stack_wgs : qpn_num_cus() * 2;
    if (G > q.total) G = q.total;
    if (G > 1024) G = 1024;
    StackQ qq = q; qq.nq = G / 8 < 1 ? 1 : (G / 8 < SQ_NQ ? G / 8 : SQ_NQ);
    qq.epoch_pub = q.epoch; qq.spin_limit = SQ_SPIN_LIMIT;
#ifdef QPN_TESTING
    if (k.test_stack_gives_up) { qq.epoch_pub = q.epoch ^ 0x55555555u; qq.spin_limit = 2000u; }
#endif
    if (k.stack_wave_fwd == 2) {
        const size_t ldsh = (size_t)(2 * 16 * 136 + 2 * 16 * 72) * sizeof(float) + 64;
        hipLaunchKernelGGL(k_stack_fwd_h, dim3(G), dim3(256), ldsh, stream, p, qq);
    } else hipLaunchKernelGGL(k_stack_fwd_t, dim3(G), dim3(256), lds, stream, p, qq);
    return QPN_OK;
}

int qpn_launch_stack_bwd_w(const TrainParams& p, const TrainBwd& bw, const StackQ& q, const TrainKnobs& k, hipStream_t stream) {
    // one wave per SIMD: the tile loop keeps the next tile's rows in registers while the current one computes (~470 of the 512 a lone wave may use)
    return launch_bwd_w<4>(p, bw, q, k, stream);
}
