// decode.hip -- persistent autoregressive decode of QPNet on gfx950 (MI355X).
//
// Replaces QPNet.batch_fast_generate (reference src/nets/qpnet.py:314-559) and the per-sample
// helpers it calls (_generate_fixed_residual_forward :672-685, _generate_adaptive_residual_forward
// :642-655, FDilatedConv1d :81-87, _generate_dilated_index :613-624, _preprocess :561-564,
// _postprocess :566-571).  The reference dispatches ~160 tiny torch ops per generated sample;
// here the whole utterance is ONE kernel launch: one 1024-thread workgroup per utterance walks a
// host-built micro-program ("tasks") once per sample.  A task is a 64-lane x 16-deep weight tile
// (4 KiB, streamed from L2 with coalesced 16-byte loads, prefetched one task ahead across the
// workgroup barrier) plus an epilogue (gate / residual / skip accumulate / post-net / argmax).
// All recurrent state of the current step lives in LDS; the per-layer input history ("ring
// buffers", up to maxd*2^k samples deep) lives in a small L2-resident global block because
// 15*maxd*C floats exceed one CU's LDS for low pitch (SURVEY.md §7 "Ring-buffer footprint").
//
// Arithmetic is the fixed-order fp32 "QPNet-f32" spec of DESIGN.md §3, so results are
// bit-identical to the CPU oracle; compile with -ffp-contract=off.
#include "qpn_common.h"
#include "qpn_handle.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>

#include "decode_dev.h"


// ================================================================== small setup kernels
__global__ void k_pack_gather(const float* __restrict__ flat, const int* __restrict__ map, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int m = map[i]; out[i] = m >= 0 ? flat[m] : 0.0f; }
}


// Qb[l][row] = ((b_up * dot(Va[row,:], 1) + ba) + b_conv) [+ b_convP]   
// one workgroup of 64*k lanes per layer; aux tiles packed natural-row, R = Ap/16.
__global__ void k_fold_bias(const float4* __restrict__ wpk, const float* __restrict__ flat, const BiasDesc* __restrict__ bd,
                            int aux_woff4_layer0, int aux_tiles_per_layer, int logRa, int A, int Ap, int C,
                            const float* __restrict__ up_b_ptr, float* __restrict__ qb) {
    const float up_b = up_b_ptr ? *up_b_ptr : 0.0f;      // read on the device: qpn_set_weights needs no read-back
    extern __shared__ float4 smem4[];
    float* sm = (float*)smem4;
    const int l = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < Ap; i += blockDim.x) sm[i] = i < A ? 1.0f : 0.0f;
    __syncthreads();
    const int R = 1 << logRa, rpt = 64 / R, q = lane & (R - 1);
    const BiasDesc b = bd[l];
    for (int t = wave; t < aux_tiles_per_layer; t += nw) {
        float4 w[4], x[4];
        load_tile(w, wpk, aux_woff4_layer0 + (l * aux_tiles_per_layer + t) * 256, lane);
        const float4* xv = (const float4*)(sm + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = xv[j];
        float rs = tree_reduce(chunk16(w, x), logRa);
        int row = t * rpt + (lane >> logRa);
        if (q == 0) {
            int half = row / C, r = row - half * C;
            float v = up_b * rs;
            v = v + flat[b.auxb[half] + r];
            v = v + flat[b.convb[half] + r];
            if (b.adaptive) v = v + flat[b.convPb[half] + r];
            qb[(size_t)l * 2 * C + row] = v;
        }
    }
}

// P[b][f][l][row] = dot(Va_l[row,:], h[b,:,f])  -- aux 1x1 convs hoisted to FRAME rate.
// grid (F, B); h is (B, A, F).
__global__ void k_aux_project(const float4* __restrict__ wpk, const float* __restrict__ h, int64_t F,
                              int aux_woff4_layer0, int aux_tiles_per_layer, int logRa, int A, int Ap, int C, int L,
                              float* __restrict__ pproj) {
    extern __shared__ float4 smem4[];
    float* sm = (float*)smem4;
    const int64_t f = blockIdx.x; const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < Ap; i += blockDim.x) sm[i] = i < A ? h[((size_t)b * A + i) * F + f] : 0.0f;
    __syncthreads();
    const int R = 1 << logRa, rpt = 64 / R, q = lane & (R - 1);
    float4 x[4];
    const float4* xv = (const float4*)(sm + 16 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = xv[j];
    float* out = pproj + ((size_t)b * F + f) * L * 2 * C;
    const int ntiles = L * aux_tiles_per_layer;
    for (int t = wave; t < ntiles; t += nw) {
        float4 w[4];
        load_tile(w, wpk, aux_woff4_layer0 + t * 256, lane);
        float v = tree_reduce(chunk16(w, x), logRa);
        int l = t / aux_tiles_per_layer, tt = t - l * aux_tiles_per_layer;
        int row = tt * rpt + (lane >> logRa);
        if (q == 0) out[(size_t)l * 2 * C + row] = v;
    }
}

// known[b][i]: n_pad copies of Q/2 (qpnet.py:358) then x % Q (OneHot, qpnet.py:76)
__global__ void k_known(const int64_t* __restrict__ x, int n_x, int n_pad, int Q, int* __restrict__ known) {
    const int b = blockIdx.y, n0 = n_pad + n_x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n0) {
        int v;
        if (i < n_pad) v = Q / 2;
        else { int64_t s = x[(size_t)b * n_x + (i - n_pad)] % Q; v = (int)(s < 0 ? s + Q : s); }
        known[(size_t)b * n0 + i] = v;
    }
}



// ================================================================== the persistent decode kernel

// aux terms a[t1] of every layer -> LDS (needs only the frame-rate projections)
__device__ __forceinline__ void stage_aux(const DecodeParams& p, const UttView& u, int64_t t1, int tid, int nthreads) {
    float* sm = SM;
    const int n = p.L * 2 * p.C;
    const int ut = aux_time(u, (int)t1);             // 32-bit: n0 + n_samples < 2^31 is checked on the host
    int f, j;
    if (ut < 0) { f = 0; j = 0; }                    // replicate pad of the upsampled h (qpnet.py:359)
    else if (p.U > 0) { f = (int)((unsigned)ut / (unsigned)p.U); j = ut - f * p.U; }
    else { f = ut; j = 0; }
    const float wj = p.U > 0 ? p.flat[p.up_w + j] : 1.0f;
    const float* pf = u.pproj + (size_t)f * n;
    for (int i = tid; i < n; i += nthreads) sm[p.o_auxv + i] = __builtin_fmaf(wj, pf[i], p.qb[i]);
}
// gathered past rows x_l[t2 - off_l(t2)] of every layer -> LDS xp; sel[l] = 1 when off == 1 (the row is
// produced during step t2-1 itself and is read from xbuf instead)
__device__ __forceinline__ void stage_taps(const DecodeParams& p, const UttView& u, int64_t t2, int tid, int nthreads, int* status) {
    const int L = p.L, C = p.C;
    float* sm = SM; int* smi = SMI;
    const int ut = aux_time(u, (int)t2);
    const int widx = (int)t2 < u.n0 - 1 ? (int)t2 - (u.n0 - 1) : 0;
    for (int i = tid; i < L * C; i += nthreads) {
        const int l = i / C, c = i - l * C;
        const RingDesc r = p.rings[l];
        int off = tap_offset(r, u, ut, widx);
        if (off < 1 || off >= r.len) { if (c == 0) atomicOr(status, 1); off = off < 1 ? 1 : r.len - 1; }
        if (c == 0) smi[p.o_sel + l] = off == 1;
        if (off > 1) {
            const int tp = (int)t2 - off;             // time of the past tap; < 0 -> a never-written (zero) slot
            const int slot = tp >= 0 ? (int)((unsigned)tp % (unsigned)r.len) : tp + r.len;
            sm[p.o_xp + l * p.Cp + c] = ld_agent(u.ring + r.base + (size_t)slot * C + c);
        }
    }
}
// causal conv on the one-hot input = two table rows (qpnet.py:110-132): input of layer 0 at time t1
__device__ __forceinline__ void causal_rows(const DecodeParams& p, const UttView& u, int s_prev, int s_cur, int64_t t1, int lane) {
    float* sm = SM;
    const RingDesc r0 = p.rings[0];
    for (int ch = lane; ch < p.C; ch += 64) {
        float v = p.flat[p.causal_w + ((size_t)ch * p.Q + s_prev) * 2] + p.flat[p.causal_w + ((size_t)ch * p.Q + s_cur) * 2 + 1];
        v = v + p.flat[p.causal_b + ch];
        sm[p.o_xbuf + ch] = v;
        st_agent(u.ring + r0.base + (size_t)((unsigned)t1 % (unsigned)r0.len) * p.C + ch, v);
    }
}

struct Ctx {
    const DecodeParams* p; const UttView* u;
    int lane, wave; int64_t Ttot;
};

// one slot of the per-step program for this wave; `w` holds the slot's weight tile and is refilled with
// the tile of slot+3 as soon as it has been consumed (three tiles in flight per wave).
__device__ __forceinline__ void run_slot(const Ctx& c, int slot, int64_t t, float4 (&w)[4]) {
    const DecodeParams& p = *c.p; const UttView& u = *c.u;
    float* sm = SM; int* smi = SMI;
    const int lane = c.lane;
    const int4* tl = (const int4*)(smi + p.o_tasks + (slot * QPN_NW + c.wave) * 8);
    const int4 q0 = tl[0], q1 = tl[1];
    const int opf = __builtin_amdgcn_readfirstlane(q0.x);
    int xoff = __builtin_amdgcn_readfirstlane(q0.z);
    const int row0 = __builtin_amdgcn_readfirstlane(q0.w);
    const int ta = __builtin_amdgcn_readfirstlane(q1.x), tb = __builtin_amdgcn_readfirstlane(q1.y);
    const int tc = __builtin_amdgcn_readfirstlane(q1.z), td = __builtin_amdgcn_readfirstlane(q1.w);
    // descriptor of the task three slots ahead (same register set)
    int ns = slot + 3; if (ns >= p.n_slots) ns -= p.n_slots;
    const int2 nq = *(const int2*)(smi + p.o_tasks + (ns * QPN_NW + c.wave) * 8);
    const int nopf = __builtin_amdgcn_readfirstlane(nq.x), nwoff = __builtin_amdgcn_readfirstlane(nq.y);
    if (opf & TF_BARRIER) {
        if (opf & TF_DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
    }
    const int op = opf & 0xff;
    const int C = p.C, Q = p.Q;
    if (opf & TF_HASW) {
        const int logR = (opf >> 16) & 0xf;
        const int R = 1 << logR, q = lane & (R - 1);
        if (op == OP_PAST && smi[p.o_sel + tc]) xoff = tb;        // tap distance 1: the row is this step's layer input
        float4 x[4];
        const float4* xv = (const float4*)(sm + xoff + 16 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = xv[j];
        float acc = chunk16(w, x);
        if (nopf & TF_HASW) load_tile(w, p.wpk, nwoff, lane);
        acc = tree_reduce(acc, logR);
        const int row = row0 + (lane >> logR);
        const bool lead = q == 0;
        const int par = (int)(t & 1) * p.L * 2 * C;
        switch (op) {
        case OP_PAST:      // a = layer*2C: past-tap dot of the NEXT step -> pd[(t+1)&1]
            if (lead) sm[p.o_pd + (p.L * 2 * C - par) + ta + row] = acc;
            break;
        case OP_Z: {       // rows interleaved (sigmoid_c, tanh_c); a = layer*2C, c = g
            const int ch = row >> 1, half = row & 1, nat = half * C + ch;
            const float z = (acc + sm[p.o_pd + par + ta + nat]) + sm[p.o_auxv + ta + nat];
            float zo;
            if (logR == 2) zo = dpp_f<0x104>(z);            // row_shl:4 -> lane i reads lane i+4 (the tanh group)
            else if (logR == 1) zo = dpp_f<0x102>(z);
            else if (logR == 3) zo = dpp_f<0x108>(z);
            else if (logR == 0) zo = dpp_f<0x101>(z);
            else zo = __shfl_xor(z, R);
            if (lead && !half) sm[tc + ch] = qgate(z, zo);
            break;
        }
        case OP_RES:       // a = x in (LDS), b = bias (LDS), c = x out (LDS), d = ring index of the consumer or -1
            if (lead) {
                const float v = (acc + sm[tb + row]) + sm[ta + row];
                sm[tc + row] = v;
                if (td >= 0) {
                    const RingDesc r = p.rings[td];
                    st_agent(u.ring + r.base + (size_t)(t % r.len) * C + row, v);
                }
            }
            break;
        case OP_SKIP:      // a = accumulator (LDS), b = bias (LDS), c = other accumulator, d = y1
            if (lead) {
                const float v = sm[ta + row] + (acc + sm[tb + row]);
                sm[ta + row] = v;
                if (opf & TF_LAST) {
                    const float tot = tc >= 0 ? sm[tc + row] + v : v;     // sumF + sumA (qpnet.py:505)
                    sm[td + row] = tot > 0.0f ? tot : 0.0f;
                }
            }
            break;
        case OP_POST1:     // b = bias, c = y2
            if (lead) { const float v = acc + sm[tb + row]; sm[tc + row] = v > 0.0f ? v : 0.0f; }
            break;
        case OP_POST2:     // b = bias, c = logits
            if (lead) sm[tc + row] = acc + sm[tb + row];
            break;
        default: break;
        }
        return;
    }
    if (op == OP_ARGMAX) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = lane; i < Q; i += 64) { const float v = sm[p.o_lg + i]; if (v > bv) { bv = v; bi = i; } }
        bi = wave_argmax(bv, bi);
        const int64_t i = t - (u.n0 - 1);
        if (i >= 0 && u.logits) for (int k = lane; k < Q; k += 64) u.logits[(size_t)i * Q + k] = sm[p.o_lg + k];
        int next;
        if (i >= 0) {
            if (p.mode == QPN_MODE_SAMPLING) bi = sample_wave(p.o_lg, Q, p.seed, (unsigned)u.row, (unsigned)i, lane);
            next = bi;
            if (u.teacher) { const int64_t sv = u.teacher[i] % Q; next = (int)(sv < 0 ? sv + Q : sv); }
            if (lane == 0) u.out[i] = bi;
        } else next = u.known[t + 1];
        const int cur = smi[p.o_samp + 1];
        if (t + 2 < c.Ttot) causal_rows(p, u, cur, next, t + 1, lane);     // layer-0 input of the next step
        if (lane == 0) { smi[p.o_samp] = cur; smi[p.o_samp + 1] = next; }
    } else if (op == OP_STAGE) {    // a = first wave of the staging group, b = waves in it
        const int stid = (c.wave - ta) * 64 + lane, nst = tb * 64;
        for (int i = stid; i < p.S; i += nst) { sm[p.o_skf + i] = 0.0f; sm[p.o_ska + i] = 0.0f; }
        if (t + 2 < c.Ttot) stage_aux(p, u, t + 1, stid, nst);
        if (t + 3 < c.Ttot) stage_taps(p, u, t + 2, stid, nst, p.status);
    }
    if (nopf & TF_HASW) load_tile(w, p.wpk, nwoff, lane);
}

__global__ __launch_bounds__(QPN_NT) void k_decode(DecodeParams p) {
    float* sm = SM;
    int* smi = SMI;
    const UttView u = make_view(p, p.utts[blockIdx.x]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < p.state_floats; i += QPN_NT) sm[i] = 0.0f;
    for (int i = tid; i < p.n_bias; i += QPN_NT) sm[p.o_bias + i] = p.flat[p.bias_src[i]];
    {
        const int* src = (const int*)p.tasks;
        for (int i = tid; i < p.n_slots * QPN_NW * 8; i += QPN_NT) smi[p.o_tasks + i] = src[i];
    }
    __syncthreads();
    const int64_t Ttot = (int64_t)u.n0 + u.n_samples;
    if (Ttot < 3) return;                                   // nothing to predict
    // state for the first step (t = 1): layer-0 input, aux terms; taps of step 2; pd of step 1 is all-zero
    if (wave == 0) {
        causal_rows(p, u, u.known[0], u.known[1], 1, lane);
        if (lane == 0) { smi[p.o_samp] = u.known[0]; smi[p.o_samp + 1] = u.known[1]; }
    }
    stage_aux(p, u, 1, tid, QPN_NT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (Ttot > 3) stage_taps(p, u, 2, tid, QPN_NT, p.status);
    __syncthreads();

    Ctx c; c.p = &p; c.u = &u; c.lane = lane; c.wave = wave; c.Ttot = Ttot;
    float4 w0[4], w1[4], w2[4];
    {
        const int* tk = smi + p.o_tasks + wave * 8;
        if (tk[0] & TF_HASW) load_tile(w0, p.wpk, tk[1], lane);
        if (tk[QPN_NW * 8] & TF_HASW) load_tile(w1, p.wpk, tk[QPN_NW * 8 + 1], lane);
        if (tk[2 * QPN_NW * 8] & TF_HASW) load_tile(w2, p.wpk, tk[2 * QPN_NW * 8 + 1], lane);
    }
    for (int64_t t = 1; t + 1 < Ttot; ++t) {
        for (int slot = 0; slot < p.n_slots; slot += 3) {
            run_slot(c, slot, t, w0);
            run_slot(c, slot + 1, t, w1);
            run_slot(c, slot + 2, t, w2);
        }
    }
}


// ================================================================== specialised straight-line decode kernel
// Same algorithm, state and bit-exact arithmetic as k_decode, but for geometries whose per-layer tiles fit one
// round of 16 waves (2*NZ <= 16: n_resch <= 64) the per-step program is compiled as straight-line code with a
// STATIC tile assignment instead of being interpreted from the task table: ~45 instructions per tile instead of
// ~170, addresses and LDS offsets folded, reductions chosen at compile time.  (In-kernel stamps showed the
// interpreter spends ~1500 cycles of pure instruction issue per slot; the memory system was idle half the time.)
//   Z phase (per layer)  waves [0,NZ): this step's z tiles + gate      waves [NZ,2NZ): next step's past-tap dots
//   R phase              residual 1x1 tiles first (-> next layer input), then the skip 1x1 tiles
//   tail                 skip total/relu -> post 1x1 #1 -> post 1x1 #2 -> argmax/causal/staging
// What bounds a step and why the kernel looks the way it does: DESIGN.md §4, profiles/r01_decode_fast_phase_timeline.txt.
template <int V> struct ILog2 { static constexpr int v = 1 + ILog2<V / 2>::v; };
template <> struct ILog2<1> { static constexpr int v = 0; };

template <int LOGR>
__device__ __forceinline__ float tree_reduce_c(float acc) {
    if constexpr (LOGR == 5) { acc = acc + __shfl_xor(acc, 16); }
    if constexpr (LOGR >= 4) { acc = acc + dpp_f<0x128>(acc); }
    if constexpr (LOGR >= 3) { if constexpr (LOGR == 3) acc = acc + __shfl_xor(acc, 4); else acc = acc + dpp_f<0x124>(acc); }
    if constexpr (LOGR >= 2) { acc = acc + dpp_f<0x4E>(acc); }
    if constexpr (LOGR >= 1) { acc = acc + dpp_f<0xB1>(acc); }
    return acc;
}
__device__ __forceinline__ void read_x(float4 (&x)[4], int xoff_plus_16q) {
    const float4* xv = (const float4*)(SM + xoff_plus_16q);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = xv[j];
}

// One tile's worth of work for the straight-line kernel: 4 coalesced 1 KiB loads / the spec dot product of a tile.
__device__ __forceinline__ void tile_load(float4 (&w)[4], const float4* tp) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = tp[j * 64];
}

// NWV waves per workgroup.  Per layer a wave owns ZT tiles of the Z phase (ids w, w+NWV, ..: z tiles first, then the
// next step's past-tap tiles) and RT tiles of the R phase (residual tiles first, then skip tiles); every tile is
// re-requested in place for the next layer right after it has been consumed (a full layer ahead of its use).
// What a wave does with its slots depends only on which interval of wave ids it falls in, so the step loop is
// instantiated once per interval ("role", W0 = its first wave): inside a role the code is straight-line with no
// wave-dependent branches around the tile requests, and hipcc's waitcnt pass can then count the outstanding loads
// (vmcnt(N)) instead of draining the queue (vmcnt(0)) in front of every tile.
template <int C, int S, int Q, int NWV>
struct FastGeo {
    static constexpr int R = C / 16, LOGR = ILog2<R>::v, RPT = 64 / R;
    static constexpr int NZ = 2 * C / RPT, NRES = C / RPT, NSK = S / RPT;
    static constexpr int ZT = (2 * NZ + NWV - 1) / NWV, RT = (NRES + NSK + NWV - 1) / NWV;
    static constexpr int RS = S / 16, LOGRS = ILog2<RS>::v, RPTS = 64 / RS;
    static constexpr int NP1 = S / RPTS, NP2 = Q / RPTS;
    static constexpr int T1 = (NP1 + NWV - 1) / NWV, T2 = (NP2 + NWV - 1) / NWV;
    static constexpr int zkind(int id) { return id < NZ ? 1 : id < 2 * NZ ? 2 : 0; }            // 1 z tile, 2 past-tap tile
    static constexpr int rkind(int id) { return id < NRES ? 1 : id < NRES + NSK ? 2 : 0; }      // 1 residual, 2 skip
    static constexpr int next_boundary(int w0) {
        int nb = NWV;
        for (int i = 0; i < 16; ++i) {
            const int th[6] = {NZ - i * NWV, 2 * NZ - i * NWV, NRES - i * NWV, NRES + NSK - i * NWV, NP1 - i * NWV, NP2 - i * NWV};
            for (int k = 0; k < 6; ++k) if (th[k] > w0 && th[k] < nb) nb = th[k];
        }
        return nb;
    }
};

template <int C, int S, int Q, int NWV, bool RESL, int W0>
__device__ __forceinline__ void fast_steps(const DecodeParams& p, const FastParams& f, const UttView& u,
                                           const int wave, const int lane0, const int tid0, const int Ttot) {
    using G = FastGeo<C, S, Q, NWV>;
    constexpr int NTH = NWV * 64;
    constexpr int R = G::R, LOGR = G::LOGR, RPT = G::RPT, NZ = G::NZ, NRES = G::NRES, ZT = G::ZT, RT = G::RT;
    constexpr int RS = G::RS, LOGRS = G::LOGRS, RPTS = G::RPTS, T1 = G::T1, T2 = G::T2;
    static_assert(ZT <= 4 && RT <= 4 && T1 <= 4 && T2 <= 4, "slot macros are expanded four times");
    float* sm = SM; int* smi = SMI;
    const int L = p.L;
    int stamp_i = 0;
#ifdef QPN_ENABLE_STAMPS   // dev aid (-DQPN_ENABLE_STAMPS + QPN_STAMPS=1): s_memtime at every phase boundary of step 3000,
                           // parked in LDS behind the kernel's own state (no global stores in the timed code), dumped after the step
#define QPN_STAMP() do { if (p.stamps && t == 3000 && lane == 0 && stamp_i < 120) smi[p.o_stamp + stamp_i * QPN_NW + (wave < QPN_NW ? wave : 0)] = (int)__builtin_amdgcn_s_memtime(); ++stamp_i; } while (0)
#define QPN_STAMP_DUMP() do { if (p.stamps && t == 3000 && lane == 0) for (int k = 0; k < stamp_i && k < 120; ++k) p.stamps[(size_t)k * QPN_NW + (wave < QPN_NW ? wave : 0)] = (long long)(unsigned)smi[p.o_stamp + k * QPN_NW + (wave < QPN_NW ? wave : 0)]; } while (0)
#else
#define QPN_STAMP() do { (void)stamp_i; } while (0)
#define QPN_STAMP_DUMP() do { } while (0)
#endif
    const float4* wl0 = p.wpk + lane0;
    const int C2 = 2 * C, LC2 = L * C2;
    float4 wz[ZT][4], wr[RT][4], wp[4];
#define QPN_ZPTR(wl, l, id) ((wl) + ((id) < NZ ? f.w_cur[l] + (id) * 256 : f.w_past[l] + ((id) - NZ) * 256))
#define QPN_RPTR(wl, l, id) ((wl) + ((id) < NRES ? f.w_res[l] + (id) * 256 : f.w_skip[l] + ((id) - NRES) * 256))
#define QPN_ZINIT(wl, i) if constexpr ((i) < ZT && G::zkind(W0 + (i) * NWV) != 0) tile_load(wz[(i) < ZT ? (i) : 0], QPN_ZPTR(wl, 0, wave + (i) * NWV));
#define QPN_RINIT(wl, i) if constexpr ((i) < RT && G::rkind(W0 + (i) * NWV) != 0 && !(RESL && G::rkind(W0 + (i) * NWV) == 1)) tile_load(wr[(i) < RT ? (i) : 0], QPN_RPTR(wl, 0, wave + (i) * NWV));
    QPN_ZINIT(wl0, 0) QPN_ZINIT(wl0, 1) QPN_ZINIT(wl0, 2) QPN_ZINIT(wl0, 3)
    QPN_RINIT(wl0, 0) QPN_RINIT(wl0, 1) QPN_RINIT(wl0, 2) QPN_RINIT(wl0, 3)

    // The post-net tiles of a wave form one sequence s = 0..T1+T2-1 (post 1x1 #1 tiles, then #2) that cycles through three
    // register sets: wp, and the layer sets wz[0] / wr[0] once the last layer has consumed them.  Tile s is requested as
    // soon as tile s-3 has been consumed, the first three during the last layer.
#define QPN_TB(s) ((s) % 3 == 0 ? wp : (s) % 3 == 1 ? wz[0] : wr[0])
#define QPN_TVALID(s) ((s) < T1 ? (W0 + (s) * NWV < G::NP1) : ((s) < T1 + T2 && W0 + ((s) - T1) * NWV < G::NP2))
#define QPN_TLOAD(s) if constexpr (QPN_TVALID(s)) tile_load(QPN_TB(s), wl + ((s) < T1 ? f.w_p1 + (wave + (s) * NWV) * 256 : f.w_p2 + (wave + ((s) - T1) * NWV) * 256));

    // ---- one layer; LAST = the final layer, which hands its tile registers to the post-net instead of re-requesting
#define QPN_ZSLOT(i, LAST)                                                                                        \
            if constexpr ((i) < ZT) {                                                                             \
                constexpr int zk = G::zkind(W0 + (i) * NWV);                                                      \
                const int id = wave + (i) * NWV;                                                                  \
                if constexpr (zk == 1) {                                                                          \
                    float4 x[4]; read_x(x, p.o_xbuf + l * p.Cp + 16 * q);                                         \
                    float acc;                                                                                    \
                    if constexpr (LAST) acc = chunk16(wz[(i) < ZT ? (i) : 0], x);                                 \
                    else acc = chunk16_reload(wz[(i) < ZT ? (i) : 0], x, QPN_ZPTR(wl, l + 1, id));                \
                    acc = tree_reduce_c<LOGR>(acc);                                                               \
                    const int row = id * RPT + grp, ch = row >> 1, half = row & 1, nat = half * C + ch;           \
                    const float z = (acc + sm[p.o_pd + par + l * C2 + nat]) + sm[p.o_auxv + l * C2 + nat];        \
                    const float zo = dpp_f<0x100 + R>(z);   /* row_shl:R -> the tanh group's pre-activation */    \
                    if (q == 0 && !half) sm[p.o_gl + l * p.Cp + ch] = qgate(z, zo);                               \
                } else if constexpr (zk == 2) {                                                                   \
                    const int xo = smi[p.o_sel + l] ? p.o_xbuf + l * p.Cp : p.o_xp + l * p.Cp;                    \
                    float4 x[4]; read_x(x, xo + 16 * q);                                                          \
                    float acc;                                                                                    \
                    if constexpr (LAST) acc = chunk16(wz[(i) < ZT ? (i) : 0], x);                                 \
                    else acc = chunk16_reload(wz[(i) < ZT ? (i) : 0], x, QPN_ZPTR(wl, l + 1, id));                \
                    acc = tree_reduce_c<LOGR>(acc);                                                               \
                    if (q == 0) sm[p.o_pd + (LC2 - par) + l * C2 + (id - NZ) * RPT + grp] = acc;  /* next parity */ \
                }                                                                                                 \
                if constexpr (LAST && (i) == 0) { QPN_TLOAD(1) }                                                     \
            }
#define QPN_RDOT(i, LAST)                                                                                         \
            if constexpr ((i) < RT) {                                                                             \
                constexpr int rk = G::rkind(W0 + (i) * NWV);                                                      \
                if constexpr (RESL && rk == 1) {                                                                  \
                    if constexpr (!(LAST)) {              /* resident tile: four 1 KiB LDS reads, no global request */ \
                        const float4* tp = (const float4*)(sm + p.o_wres) + (l * NRES + wave + (i) * NWV) * 256 + lane; \
                        /* the slot's register set is idle in this mode (the post-net borrows it later): use it as the landing zone */ \
                        _Pragma("unroll") for (int j = 0; j < 4; ++j) wr[(i) < RT ? (i) : 0][j] = tp[j * 64];       \
                        racc[(i) < RT ? (i) : 0] = tree_reduce_c<LOGR>(chunk16(wr[(i) < RT ? (i) : 0], xg));      \
                    }                                                                                             \
                } else if constexpr (rk != 0) {                                                                   \
                    if constexpr (LAST) racc[(i) < RT ? (i) : 0] = tree_reduce_c<LOGR>(chunk16(wr[(i) < RT ? (i) : 0], xg)); \
                    else racc[(i) < RT ? (i) : 0] = tree_reduce_c<LOGR>(chunk16_reload(wr[(i) < RT ? (i) : 0], xg, QPN_RPTR(wl, l + 1, wave + (i) * NWV))); \
                }                                                                                                 \
                if constexpr (LAST && (i) == 0) { QPN_TLOAD(2) }                                                  \
            }
#define QPN_RPUT(i, LASTL)                                                                                            \
            if constexpr ((i) < RT) {                                                                             \
                constexpr int rk = G::rkind(W0 + (i) * NWV);                                                      \
                const int id = wave + (i) * NWV;                                                                  \
                if constexpr (rk == 1 && !(RESL && LASTL)) {                                                      \
                    const int row = id * RPT + grp;                                                               \
                    /* x_{l+1}; its ring row is written once per step at the end (no stores in the load queue here) */ \
                    sm[p.o_xbuf + (l + 1) * p.Cp + row] = (racc[(i) < RT ? (i) : 0] + sm[f.b_res[l] + row]) + sm[p.o_xbuf + l * p.Cp + row]; \
                } else if constexpr (rk == 2) {                                                                   \
                    const int row = (id - NRES) * RPT + grp;                                                      \
                    const int a = (f.adaptive[l] ? p.o_ska : p.o_skf) + row;                                      \
                    sm[a] = sm[a] + (racc[(i) < RT ? (i) : 0] + sm[f.b_skip[l] + row]);                           \
                }                                                                                                 \
            }
#define QPN_LAYER(LAST)                                                                                           \
            if constexpr (LAST) { QPN_TLOAD(0) }                                                                  \
            wg_barrier();                                   /* layer input x_l, pd[par], aux terms are in LDS */  \
            QPN_STAMP();                                                                                          \
            QPN_ZSLOT(0, LAST) QPN_ZSLOT(1, LAST) QPN_ZSLOT(2, LAST) QPN_ZSLOT(3, LAST)                           \
            QPN_STAMP();                                                                                          \
            wg_barrier();                                   /* gate vector g_l is in LDS */                       \
            QPN_STAMP();                                                                                          \
            if constexpr (G::rkind(W0) != 0) {              /* every R tile of the wave reads the same g_l */     \
                float4 xg[4]; read_x(xg, p.o_gl + l * p.Cp + 16 * q);                                             \
                float racc[RT];                                                                                   \
                QPN_RDOT(0, LAST) QPN_RDOT(1, LAST) QPN_RDOT(2, LAST) QPN_RDOT(3, LAST)                           \
                if (q == 0) { QPN_RPUT(0, LAST) QPN_RPUT(1, LAST) QPN_RPUT(2, LAST) QPN_RPUT(3, LAST) }                                   \
            } else if constexpr (LAST) { QPN_TLOAD(2) }

    for (int t = 1; t + 1 < Ttot; ++t) {
        // hipcc hoists every lane-constant address out of the step loop and then spills them: an opaque zero
        // re-derives the lane ids per step instead, so nothing derived from them can be hoisted
        int zero = 0;
        asm volatile("" : "+s"(zero));
        const int lane = lane0 + zero, tid = tid0 + zero;
        const float4* wl = wl0 + zero;
        const int q = lane & (R - 1), grp = lane >> LOGR, qs = lane & (RS - 1), grps = lane >> LOGRS;
        const int par = (t & 1) * LC2;
        stamp_i = 0;
        for (int l = 0; l + 1 < L; ++l) { QPN_LAYER(false) }
        { const int l = L - 1; QPN_LAYER(true) }
        QPN_STAMP();
        // ---- tail: skip total / relu -> post 1x1 #1 -> post 1x1 #2
        wg_barrier();
        for (int row = tid; row < S; row += NTH) {
            const float tot = sm[p.o_skf + row] + sm[p.o_ska + row];      // sum(skip_F) + sum(skip_A)  (qpnet.py:505)
            sm[p.o_y1 + row] = tot > 0.0f ? tot : 0.0f;
        }
        wg_barrier();
#define QPN_PDOT(s, acc) if constexpr (QPN_TVALID(s)) { acc = tree_reduce_c<LOGRS>(chunk16(QPN_TB(s), xq)); } QPN_TLOAD((s) + 3)
        if constexpr (W0 < G::NP1) {
            float4 xq[4]; read_x(xq, p.o_y1 + 16 * qs);
            float pa[T1];
            if constexpr (0 < T1) { QPN_PDOT(0, pa[0]) } if constexpr (1 < T1) { QPN_PDOT(1, pa[1 < T1 ? 1 : 0]) }
            if constexpr (2 < T1) { QPN_PDOT(2, pa[2 < T1 ? 2 : 0]) } if constexpr (3 < T1) { QPN_PDOT(3, pa[3 < T1 ? 3 : 0]) }
            if (qs == 0) {
#pragma unroll
                for (int i = 0; i < T1; ++i) if (W0 + i * NWV < G::NP1) {
                    const int row = (wave + i * NWV) * RPTS + grps;
                    const float v = pa[i] + sm[f.b_p1 + row];
                    sm[p.o_y2 + row] = v > 0.0f ? v : 0.0f;
                }
            }
        } else {                                            // no post #1 tile here: its slots of the sequence are free at once
            if constexpr (0 < T1) { QPN_TLOAD(3) } if constexpr (1 < T1) { QPN_TLOAD(4) } if constexpr (2 < T1) { QPN_TLOAD(5) }
        }
        wg_barrier();
        if constexpr (W0 < G::NP2) {
            float4 xq[4]; read_x(xq, p.o_y2 + 16 * qs);
            float pa[T2];
            if constexpr (0 < T2) { QPN_PDOT(T1 + 0, pa[0]) } if constexpr (1 < T2) { QPN_PDOT(T1 + 1, pa[1 < T2 ? 1 : 0]) }
            if constexpr (2 < T2) { QPN_PDOT(T1 + 2, pa[2 < T2 ? 2 : 0]) } if constexpr (3 < T2) { QPN_PDOT(T1 + 3, pa[3 < T2 ? 3 : 0]) }
            if (qs == 0) {
#pragma unroll
                for (int i = 0; i < T2; ++i) if (W0 + i * NWV < G::NP2) sm[p.o_lg + (wave + i * NWV) * RPTS + grps] = pa[i] + sm[f.b_p2 + (wave + i * NWV) * RPTS + grps];
            }
        }
#undef QPN_PDOT
        QPN_STAMP();
        // ---- end of step: this step's layer inputs x_1..x_{L-1} go to their rings (one row each) ...
        for (int i = tid; i < (L - 1) * C; i += NTH) {
            const int l1 = 1 + i / C, row = i % C;
            const RingDesc r = p.rings[l1];
            st_agent(u.ring + r.base + (size_t)((unsigned)t % (unsigned)r.len) * C + row, sm[p.o_xbuf + l1 * p.Cp + row]);
        }
        QPN_STAMP();
        QPN_STAMP_DUMP();
        __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): the ring rows have left the wave
        wg_barrier();
        // ... layer-0 tiles of the next step fly while the sample is picked, the next layer-0 input looked up and the
        // aux terms / past rows staged
        QPN_ZINIT(wl, 0) QPN_ZINIT(wl, 1) QPN_ZINIT(wl, 2) QPN_ZINIT(wl, 3)
        QPN_RINIT(wl, 0) QPN_RINIT(wl, 1) QPN_RINIT(wl, 2) QPN_RINIT(wl, 3)
        if (wave == 0) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int i = lane; i < Q; i += 64) { const float v = sm[p.o_lg + i]; if (v > bv) { bv = v; bi = i; } }
            bi = wave_argmax(bv, bi);
            const int i = t - (u.n0 - 1);
            if (i >= 0 && u.logits) for (int k = lane; k < Q; k += 64) u.logits[(size_t)i * Q + k] = sm[p.o_lg + k];
            int next;
            if (i >= 0) {
                if (p.mode == QPN_MODE_SAMPLING) bi = sample_wave(p.o_lg, Q, p.seed, (unsigned)u.row, (unsigned)i, lane);
                next = bi;
                if (u.teacher) { const int64_t sv = u.teacher[i] % Q; next = (int)(sv < 0 ? sv + Q : sv); }
                if (lane == 0) u.out[i] = bi;
            } else next = u.known[t + 1];
            const int cur = smi[p.o_samp + 1];
            if (t + 2 < Ttot) causal_rows(p, u, cur, next, t + 1, lane);
            if (lane == 0) { smi[p.o_samp] = cur; smi[p.o_samp + 1] = next; }
        } else {
            const int stid = (wave - 1) * 64 + lane, nst = (NWV - 1) * 64;
            for (int i = stid; i < S; i += nst) { sm[p.o_skf + i] = 0.0f; sm[p.o_ska + i] = 0.0f; }
            if (t + 2 < Ttot) stage_aux(p, u, t + 1, stid, nst);
            if (t + 3 < Ttot) stage_taps(p, u, t + 2, stid, nst, p.status);
        }
    }
#undef QPN_STAMP
#undef QPN_STAMP_DUMP
#undef QPN_ZPTR
#undef QPN_RPTR
#undef QPN_ZINIT
#undef QPN_RINIT
#undef QPN_TB
#undef QPN_TVALID
#undef QPN_TLOAD
#undef QPN_ZSLOT
#undef QPN_RDOT
#undef QPN_RPUT
#undef QPN_LAYER
}

template <int C, int S, int Q, int NWV, bool RESL, int W0>
__device__ __forceinline__ void fast_dispatch(const DecodeParams& p, const FastParams& f, const UttView& u,
                                              const int wave, const int lane, const int tid, const int Ttot) {
    constexpr int NB = FastGeo<C, S, Q, NWV>::next_boundary(W0);
    if constexpr (NB >= NWV) fast_steps<C, S, Q, NWV, RESL, W0>(p, f, u, wave, lane, tid, Ttot);
    else {
        if (wave < NB) fast_steps<C, S, Q, NWV, RESL, W0>(p, f, u, wave, lane, tid, Ttot);
        else fast_dispatch<C, S, Q, NWV, RESL, NB>(p, f, u, wave, lane, tid, Ttot);
    }
}

// RESL: the residual-1x1 tiles of layers 0..L-2 (the ones on the critical path of every R phase) live in LDS for the whole
// launch: 16 fewer 1 KiB requests per layer through the serial vector-memory front end, and the waves that own them never
// block on it.  The host picks RESL when the tiles fit beside the step state (paper-size: 112 KiB + 38 KiB).
template <int C, int S, int Q, int NWV, bool RESL>
__global__ __launch_bounds__(NWV * 64) void k_decode_fast(DecodeParams p, FastParams f) {
    using G = FastGeo<C, S, Q, NWV>;
    static_assert(G::R <= 8 && G::RS <= 32 && G::ZT >= 1 && G::RT >= 1, "geometry not covered by the fast kernel");
    constexpr int NTH = NWV * 64;
    float* sm = SM; int* smi = SMI;
    const UttView u = make_view(p, p.utts[blockIdx.x]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < p.state_floats; i += NTH) sm[i] = 0.0f;
    for (int i = tid; i < p.n_bias; i += NTH) sm[p.o_bias + i] = p.flat[p.bias_src[i]];
    __syncthreads();
    const int Ttot = u.n0 + u.n_samples;
    if (Ttot < 3) return;
    if (wave == 0) {
        causal_rows(p, u, u.known[0], u.known[1], 1, lane);
        if (lane == 0) { smi[p.o_samp] = u.known[0]; smi[p.o_samp + 1] = u.known[1]; }
    }
    stage_aux(p, u, 1, tid, NTH);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (Ttot > 3) stage_taps(p, u, 2, tid, NTH, p.status);
    __syncthreads();
    if constexpr (RESL) {
        const int per = G::NRES * 256;                                   // float4s per layer
        float4* dst = (float4*)(sm + p.o_wres);
        for (int i = tid; i < (p.L - 1) * per; i += NTH) { const int l = i / per; dst[i] = p.wpk[f.w_res[l] + (i - l * per)]; }
        __syncthreads();
    }
    fast_dispatch<C, S, Q, NWV, RESL, 0>(p, f, u, wave, lane, tid, Ttot);
}

// ================================================================== host side
static thread_local char g_err[512] = "";
void qpn_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* qpn_last_error(void) { return g_err; }
extern "C" int qpn_version(void) { return 1000; }

int qpn_build_geom(const qpn_config* c, Geom* g) {
    memset(g, 0, sizeof(*g));
    if (!c) { qpn_set_error("null config"); return QPN_EINVAL; }
    g->cfg = *c;
    if (c->kernel_size != 2) { qpn_set_error("kernel_size must be 2 (reference runQP.py always uses 2)"); return QPN_EINVAL; }
    if (c->n_quantize < 2 || c->n_aux < 1 || c->n_resch < 1 || c->n_skipch < 1 || c->upsampling_factor < 0 ||
        c->dilationF_depth < 0 || c->dilationA_depth < 0 || c->dilationF_repeat < 0 || c->dilationA_repeat < 0) {
        qpn_set_error("bad geometry"); return QPN_EINVAL;
    }
    int C = g->C = c->n_resch, S = g->S = c->n_skipch, Q = g->Q = c->n_quantize, A = g->A = c->n_aux;
    g->U = c->upsampling_factor;
    g->LF = c->dilationF_depth * c->dilationF_repeat; g->LA = c->dilationA_depth * c->dilationA_repeat; g->L = g->LF + g->LA;
    if (g->L < 1 || g->L > QPN_MAX_LAYERS) { qpn_set_error("1..%d residual layers supported", QPN_MAX_LAYERS); return QPN_EINVAL; }
    g->Cp = qpn_pad_k(C); g->Sp = qpn_pad_k(S); g->Ap = qpn_pad_k(A);
    int64_t o = 0;
    g->causal_w = o; o += (int64_t)C * Q * 2; g->causal_b = o; o += C;
    if (g->U > 0) { g->up_w = o; o += g->U; g->up_b = o; o += 1; } else { g->up_w = g->up_b = -1; }
    const int64_t convsz = (int64_t)C * C * 2 + C, auxsz = (int64_t)C * A + C, sksz = (int64_t)S * C + S, rssz = (int64_t)C * C + C;
    const int64_t dconv = 2 * ((int64_t)C * C + C);
    int64_t Fs = o; o += g->LF * convsz; int64_t Ft = o; o += g->LF * convsz;
    int64_t Fas = o; o += g->LF * auxsz; int64_t Fat = o; o += g->LF * auxsz;
    int64_t Fsk = o; o += g->LF * sksz; int64_t Frs = o; o += g->LF * rssz;
    int64_t As = o; o += g->LA * dconv; int64_t At = o; o += g->LA * dconv;
    int64_t Aas = o; o += g->LA * auxsz; int64_t Aat = o; o += g->LA * auxsz;
    int64_t Ask = o; o += g->LA * sksz; int64_t Ars = o; o += g->LA * rssz;
    g->post1_w = o; o += (int64_t)S * S; g->post1_b = o; o += S;
    g->post2_w = o; o += (int64_t)Q * S; g->post2_b = o; o += Q;
    g->n_params = o;
    for (int l = 0; l < g->L; ++l) {
        LayerGeom& y = g->layers[l];
        y.adaptive = l >= g->LF;
        int i = y.adaptive ? l - g->LF : l;
        int depth = y.adaptive ? c->dilationA_depth : c->dilationF_depth;
        y.dilation = 1 << (i % depth);
        if (!y.adaptive) {
            y.wS = Fs + i * convsz; y.bS = y.wS + (int64_t)C * C * 2;
            y.wT = Ft + i * convsz; y.bT = y.wT + (int64_t)C * C * 2;
            y.auxS = Fas + i * auxsz; y.auxSb = y.auxS + (int64_t)C * A;
            y.auxT = Fat + i * auxsz; y.auxTb = y.auxT + (int64_t)C * A;
            y.skip = Fsk + i * sksz; y.skipb = y.skip + (int64_t)S * C;
            y.res = Frs + i * rssz; y.resb = y.res + (int64_t)C * C;
        } else {
            y.wS = As + i * dconv; y.bS = y.wS + (int64_t)C * C; y.wSP = y.bS + C; y.bSP = y.wSP + (int64_t)C * C;
            y.wT = At + i * dconv; y.bT = y.wT + (int64_t)C * C; y.wTP = y.bT + C; y.bTP = y.wTP + (int64_t)C * C;
            y.auxS = Aas + i * auxsz; y.auxSb = y.auxS + (int64_t)C * A;
            y.auxT = Aat + i * auxsz; y.auxTb = y.auxT + (int64_t)C * A;
            y.skip = Ask + i * sksz; y.skipb = y.skip + (int64_t)S * C;
            y.res = Ars + i * rssz; y.resb = y.res + (int64_t)C * C;
        }
    }
    g->recF = 0; g->recA = 0;
    for (int l = 0; l < g->L; ++l) (g->layers[l].adaptive ? g->recA : g->recF) += g->layers[l].dilation;
    return QPN_OK;
}

extern "C" int64_t qpn_param_count(const qpn_config* cfg) {
    Geom g; if (qpn_build_geom(cfg, &g) != QPN_OK) return -1; return g.n_params;
}



static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// append the tiles of one matrix to the gather map; returns the float4 offset of its first tile.
// src(row, k) -> flat index (k < K) ; rows in PACKED order.
template <class F>
static int pack_matrix(std::vector<int>& map, int rows, int K, int Kp, F src) {
    const int R = Kp / 16, rpt = 64 / R, tiles = rows / rpt;
    const int off4 = (int)(map.size() / 4);
    map.resize(map.size() + (size_t)tiles * 1024);
    int* m = map.data() + (size_t)off4 * 4;
    for (int t = 0; t < tiles; ++t)
        for (int j = 0; j < 4; ++j)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    int row = t * rpt + lane / R, q = lane % R, k = 16 * q + 4 * j + e;
                    int64_t s = k < K ? src(row, k) : -1;
                    m[((size_t)(t * 4 + j) * 64 + lane) * 4 + e] = (int)s;
                }
    return off4;
}

// 16-row tiles as the MFMA's A operand (decode_coopb.hip).  ALL the fragments of one workgroup are contiguous (blocks of `per_w` float4 from `base4`: the
// workgroup streams 1.5 MB per generated sample and should not walk a new page for every matrix); inside a block, the tile at `rel4` holds, per 16-deep
// chunk c, word c * 64 + lane: element e = M[row m = lane & 15][16 c + 4 e + (lane >> 4)].  src(w, m, k) -> flat index, or -1 (a zero: padding rows)
template <class F>
static void fill_mtiles(std::vector<int>& map, size_t base4, size_t per_w, size_t rel4, int G, int R, F src) {
    for (int w = 0; w < G; ++w) {
        int* mm = map.data() + (base4 + (size_t)w * per_w + rel4) * 4;
        for (int c = 0; c < R; ++c)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e)
                    mm[((size_t)c * 64 + lane) * 4 + e] = (int)src(w, lane & 15, 16 * c + 4 * e + (lane >> 4));
    }
}
bool qpn_coopb_supported(const Geom& g);

static int build_program(qpn_handle* h) {
    Geom& g = h->g;
    const int C = g.C, S = g.S, Q = g.Q, A = g.A, L = g.L;
    const int Rc = g.Cp / 16, Rs = g.Sp / 16, Ra = g.Ap / 16;
    if (Rc > 32 || Rs > 64 || Ra > 64) { qpn_set_error("n_resch <= 512, n_skipch/n_aux <= 1024 supported by the decode kernel"); return QPN_EINVAL; }
    if ((2 * C) % (64 / Rc) || C % (64 / Rc) || S % (64 / Rc) || S % (64 / Rs) || Q % (64 / Rs) || (2 * C) % (64 / Ra)) {
        qpn_set_error("channel counts must be multiples of the tile height (n_resch %% %d, n_skipch %% %d, n_quantize %% %d)",
                      64 / Rc, std::max(64 / Rc, 64 / Rs), 64 / Rs);
        return QPN_EINVAL;
    }
    if (g.n_params >= (int64_t)1 << 31) { qpn_set_error("model too large for 32-bit gather map"); return QPN_EINVAL; }
    // ---- LDS layout (float offsets, all multiples of 4): step state first (zeroed at start) ...
    DecodeParams& p = h->dp;
    memset(&p, 0, sizeof(p));
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    p.o_xbuf = take((L + 1) * g.Cp); p.o_xp = take(L * g.Cp); p.o_pd = take(2 * L * 2 * C); p.o_auxv = take(L * 2 * C);
    p.o_g = take(g.Cp); p.o_skf = take(S); p.o_ska = take(S); p.o_y1 = take(g.Sp); p.o_y2 = take(g.Sp); p.o_lg = take(Q);
    p.o_samp = take(4); p.o_sel = take(L); p.o_gl = take(L * g.Cp); p.state_floats = o;
    // ... then the biases the epilogues add (so no global load sits behind the weight prefetch queue)
    std::vector<int>& bsrc = h->h_bias_src; bsrc.clear();
    auto bias_block = [&](int64_t flat_off, int n) { int r = p.o_bias + (int)bsrc.size(); for (int i = 0; i < n; ++i) bsrc.push_back((int)(flat_off + i)); return r; };
    p.o_bias = o;
    std::vector<int> lds_br(L), lds_bs(L);
    for (int l = 0; l < L; ++l) { lds_br[l] = bias_block(g.layers[l].resb, C); lds_bs[l] = bias_block(g.layers[l].skipb, S); }
    const int lds_b1 = bias_block(g.post1_b, S), lds_b2 = bias_block(g.post2_b, Q);
    p.n_bias = (int)bsrc.size();
    o += (p.n_bias + 3) & ~3;
    p.C = C; p.Cp = g.Cp; p.S = S; p.Q = Q; p.L = L; p.U = g.U;
    p.causal_w = g.causal_w; p.causal_b = g.causal_b; p.up_w = g.up_w;

    // ---- packed weight tiles + task list
    std::vector<int>& map = h->h_map; map.clear();
    const bool cb_ok = qpn_coopb_supported(g);
    h->cb_ok = cb_ok; h->cb_groups = h->cb_per = 0;
    const int cbG = C / 8, cbRC = g.Cp / 16, cbRS = g.Sp / 16;
    const size_t cb_per_w = cb_ok ? ((size_t)3 * L * cbRC + 2 * cbRS) * 64 : 0;      // float4 per workgroup: three tiles per layer + the two post-net tiles
    const size_t cb_base4 = map.size() / 4;
    if (cb_ok) {
        map.resize(map.size() + cb_per_w * cbG * 4, -1);
        h->cb_base4 = (long long)cb_base4; h->cb_per_w = (int)cb_per_w;
    }
    std::vector<std::vector<Task>> phases;
    auto tile_tasks = [&](std::vector<Task>& ph, int op, int off4, int rows, int Kp, int xoff, int a, int b, int c, int d, int flags) {
        const int R = Kp / 16, rpt = 64 / R, tiles = rows / rpt;
        for (int t = 0; t < tiles; ++t) {
            Task k; k.op = op | TF_HASW | flags | (ilog2(R) << 16); k.woff4 = off4 + t * 256; k.xoff = xoff; k.row0 = t * rpt;
            k.a = a; k.b = b; k.c = c; k.d = d; ph.push_back(k);
        }
    };
    for (int l = 0; l < L; ++l) {
        const LayerGeom y = g.layers[l];
        auto srcw = [&](int half, int r, int k, int past) -> int64_t {
            if (!y.adaptive) return (half ? y.wT : y.wS) + ((int64_t)r * C + k) * 2 + (past ? 0 : 1);
            if (past) return (half ? y.wTP : y.wSP) + (int64_t)r * C + k;
            return (half ? y.wT : y.wS) + (int64_t)r * C + k;
        };
        int off_past = pack_matrix(map, 2 * C, C, g.Cp, [&](int row, int k) { return srcw(row / C, row % C, k, 1); });
        int off_cur = pack_matrix(map, 2 * C, C, g.Cp, [&](int row, int k) { return srcw(row & 1, row >> 1, k, 0); });
        h->w_past_il[l] = pack_matrix(map, 2 * C, C, g.Cp, [&](int row, int k) { return srcw(row & 1, row >> 1, k, 1); });   // cooperative kernel: (sigma_c, tanh_c) rows together
        int off_res = pack_matrix(map, C, C, g.Cp, [&](int row, int k) { return y.res + (int64_t)row * C + k; });
        int off_skip = pack_matrix(map, S, C, g.Cp, [&](int row, int k) { return y.skip + (int64_t)row * C + k; });
        if (cb_ok) {      // the batched cooperative kernel's A-operand fragments (decode_coopb.hip): workgroup w = channels 8 w .. 8 w + 7
            const int SBc = S / cbG;
            h->cb_zc[l] = (3 * l + 0) * cbRC * 64; h->cb_zp[l] = (3 * l + 1) * cbRC * 64; h->cb_rs[l] = (3 * l + 2) * cbRC * 64;      // (relative to the workgroup's block)
            fill_mtiles(map, cb_base4, cb_per_w, h->cb_zc[l], cbG, cbRC, [&](int w, int m, int k) -> int64_t { return k < C ? srcw(m >> 3, 8 * w + (m & 7), k, 0) : -1; });
            fill_mtiles(map, cb_base4, cb_per_w, h->cb_zp[l], cbG, cbRC, [&](int w, int m, int k) -> int64_t { return k < C ? srcw(m >> 3, 8 * w + (m & 7), k, 1) : -1; });
            fill_mtiles(map, cb_base4, cb_per_w, h->cb_rs[l], cbG, cbRC, [&](int w, int m, int k) -> int64_t {
                if (k >= C) return -1;
                if (m < 8) return y.res + (int64_t)(8 * w + m) * C + k;
                return m - 8 < SBc ? y.skip + (int64_t)(SBc * w + m - 8) * C + k : -1; });
        }
        // Z phase: this step's pre-activations + the NEXT step's past-tap dots of the same layer (the
        // past rows are known one step early, so these tiles fill the waves the z tiles leave idle)
        std::vector<Task> zp, rp;
        h->fp.w_cur[l] = off_cur; h->fp.w_past[l] = off_past; h->fp.w_res[l] = off_res; h->fp.w_skip[l] = off_skip;
        h->fp.b_res[l] = lds_br[l]; h->fp.b_skip[l] = lds_bs[l]; h->fp.adaptive[l] = y.adaptive;
        tile_tasks(zp, OP_Z, off_cur, 2 * C, g.Cp, p.o_xbuf + l * g.Cp, l * 2 * C, 0, p.o_g, 0, 0);
        tile_tasks(zp, OP_PAST, off_past, 2 * C, g.Cp, p.o_xp + l * g.Cp, l * 2 * C, p.o_xbuf + l * g.Cp, l, 0, 0);
        tile_tasks(rp, OP_RES, off_res, C, g.Cp, p.o_g, p.o_xbuf + l * g.Cp, lds_br[l], p.o_xbuf + (l + 1) * g.Cp, l + 1 < L ? l + 1 : -1, 0);
        const bool last = l == L - 1;
        int acc = y.adaptive ? p.o_ska : p.o_skf;
        int other = last ? (y.adaptive ? (g.LF > 0 ? p.o_skf : -1) : -1) : 0;
        tile_tasks(rp, OP_SKIP, off_skip, S, g.Cp, p.o_g, acc, lds_bs[l], other, p.o_y1, last ? TF_LAST : 0);
        phases.push_back(zp); phases.push_back(rp);
    }
    if (cb_ok) {
        const int SBc = S / cbG, QBc = Q / cbG;
        h->cb_p1 = 3 * L * cbRC * 64; h->cb_p2 = h->cb_p1 + cbRS * 64;
        fill_mtiles(map, cb_base4, cb_per_w, h->cb_p1, cbG, cbRS, [&](int w, int m, int k) -> int64_t { return (m < SBc && k < S) ? g.post1_w + (int64_t)(SBc * w + m) * S + k : -1; });
        fill_mtiles(map, cb_base4, cb_per_w, h->cb_p2, cbG, cbRS, [&](int w, int m, int k) -> int64_t { return (m < QBc && k < S) ? g.post2_w + (int64_t)(QBc * w + m) * S + k : -1; });
    }
    int off_p1 = pack_matrix(map, S, S, g.Sp, [&](int row, int k) { return g.post1_w + (int64_t)row * S + k; });
    int off_p2 = pack_matrix(map, Q, S, g.Sp, [&](int row, int k) { return g.post2_w + (int64_t)row * S + k; });
    h->fp.w_p1 = off_p1; h->fp.w_p2 = off_p2; h->fp.b_p1 = lds_b1; h->fp.b_p2 = lds_b2;
    std::vector<Task> p1, p2, pa;
    tile_tasks(p1, OP_POST1, off_p1, S, g.Sp, p.o_y1, 0, lds_b1, p.o_y2, 0, 0);
    tile_tasks(p2, OP_POST2, off_p2, Q, g.Sp, p.o_y2, 0, lds_b2, p.o_lg, 0, 0);
    { Task k; memset(&k, 0, sizeof(k)); k.op = OP_ARGMAX; pa.push_back(k);
      for (int w = 1; w < QPN_NW; ++w) { Task s; memset(&s, 0, sizeof(s)); s.op = OP_STAGE; s.a = 1; s.b = QPN_NW - 1; pa.push_back(s); } }
    // aux matrices (natural rows) for the frame-rate projection kernels
    h->logRa = ilog2(Ra); h->aux_tiles = 2 * C / (64 / Ra);
    h->aux_woff4 = -1;
    for (int l = 0; l < L; ++l) {
        const LayerGeom y = g.layers[l];
        int off = pack_matrix(map, 2 * C, A, g.Ap, [&](int row, int k) { return (row / C ? y.auxT : y.auxS) + (int64_t)(row % C) * A + k; });
        if (l == 0) h->aux_woff4 = off;
    }
    phases.push_back(p1); phases.push_back(p2); phases.push_back(pa);
    // lay the phases out as slots x waves; the ARGMAX/STAGE phase keeps its fixed wave assignment
    h->h_tasks.clear();
    int slot = 0;
    for (size_t pi = 0; pi < phases.size(); ++pi) {
        const std::vector<Task>& ph = phases[pi];
        const bool is_last = pi + 1 == phases.size();
        int ns = (int)((ph.size() + QPN_NW - 1) / QPN_NW);
        h->h_tasks.resize((size_t)(slot + ns) * QPN_NW);
        for (int s = 0; s < ns; ++s)
            for (int w = 0; w < QPN_NW; ++w) {
                size_t idx = (size_t)s * QPN_NW + w;
                Task k; memset(&k, 0, sizeof(k)); k.op = OP_NOP;
                if (idx < ph.size()) k = ph[idx];
                if (s == 0) k.op |= TF_BARRIER | (is_last ? TF_DRAIN : 0);
                h->h_tasks[(size_t)(slot + s) * QPN_NW + w] = k;
            }
        slot += ns;
    }
    while (slot % 3) {      // the kernel unrolls the slot loop by 3 (three weight tiles in flight per wave)
        h->h_tasks.resize((size_t)(slot + 1) * QPN_NW);
        for (int w = 0; w < QPN_NW; ++w) { Task k; memset(&k, 0, sizeof(k)); h->h_tasks[(size_t)slot * QPN_NW + w] = k; }
        ++slot;
    }
    h->n_slots = slot;
    p.n_slots = slot;
    p.o_tasks = o; o += slot * QPN_NW * 8;
    p.lds_floats = o;
    h->single_cu_ok = (size_t)o * 4 <= 160 * 1024;        // otherwise: several workgroups per utterance (decode_coop.hip)
    return QPN_OK;
}

bool qpn_pipe_supported(const Geom& g);
int qpn_pipe_rows_resident(int n_cus);
int qpn_launch_decode_pipe(qpn_handle* h, DecodeParams& p, int B, int groups, hipStream_t stream);
int qpn_coop_group_size(const Geom& g, int limit);
int qpn_launch_decode_coop(qpn_handle* h, DecodeParams& p, int B, int G, hipStream_t stream);
bool qpn_coopb_supported(const Geom& g);
int qpn_launch_decode_coopb(qpn_handle* h, DecodeParams& p, int B, hipStream_t stream);

extern "C" int qpn_create(const qpn_config* cfg, qpn_handle** out) {
    if (!out) { qpn_set_error("null out"); return QPN_EINVAL; }
    *out = nullptr;
    qpn_handle* h = new qpn_handle();
    int rc = qpn_build_geom(cfg, &h->g);
    if (rc != QPN_OK) { delete h; return rc; }
    h->d_map = nullptr; h->d_wpk = nullptr; h->d_tasks = nullptr; h->d_qb = nullptr; h->d_bd = nullptr; h->d_status = nullptr; h->d_bias_src = nullptr;
    h->d_flat = nullptr; h->have_weights = false;
    h->d_pproj = nullptr; h->pproj_cap = 0; h->d_ring = nullptr; h->ring_cap = 0; h->d_known = nullptr; h->known_cap = 0;
    h->d_xch = nullptr; h->xch_cap = 0; h->single_cu_ok = true;
    h->d_utts = nullptr; h->utts_cap = 0; h->ev0 = h->ev1 = nullptr; h->last_ms = 0; h->pending = false; h->device = -1; h->train = nullptr;
    h->n_cus = 0; h->pipe_rows = 0; h->h_utts_pinned = nullptr; h->h_utts_cap = 0; h->dec_side = nullptr; h->dec_fork = h->dec_join = nullptr;
    {   // the environment is read HERE, once per handle: no decode or training call looks at it again
        DecodeKnobs& k = h->dk;
        k.generic = getenv("QPN_DECODE_GENERIC") != nullptr;
        k.no_resl = getenv("QPN_DECODE_NO_RESL") != nullptr;
        k.coop = 0; if (const char* e = getenv("QPN_DECODE_COOP")) k.coop = atoi(e) > 0 ? atoi(e) : 0;
        // (measured on one MI355X, repo-default geometry, us per sample step of the whole batch, batched / per-utterance kernel: B = 1: 89 / 92, 2: 90 / 97, 4: 98 / 110,
        //  8: 115 / 123, 16: 137 / 143, 20: 147 / 200, 32: 165 / 204, 64: 216 / 357 -- profiles/r06_coopb_batches.txt)
        k.coopb = 1; if (const char* e = getenv("QPN_DECODE_COOPB")) k.coopb = atoi(e) > 0 ? atoi(e) : 0;
        k.coopb_per = 0; if (const char* e = getenv("QPN_DECODE_COOPB_PER")) k.coopb_per = atoi(e);
        k.coopb_delay[0] = 8; k.coopb_delay[1] = 4; k.coopb_delay[2] = 0;
        if (const char* e = getenv("QPN_COOPB_DELAY_G")) k.coopb_delay[0] = atoi(e);
        if (const char* e = getenv("QPN_COOPB_DELAY_X")) k.coopb_delay[1] = atoi(e);
        if (const char* e = getenv("QPN_COOPB_DELAY_T")) k.coopb_delay[2] = atoi(e);
        k.pipe = 1; if (const char* e = getenv("QPN_DECODE_PIPE")) k.pipe = atoi(e) != 0 ? 1 : 0;
        k.hybrid = getenv("QPN_DECODE_HYBRID") != nullptr;
        k.stamps = getenv("QPN_STAMPS") != nullptr;
        k.test_pipe_gives_up = false;
#ifdef QPN_TESTING
        k.test_pipe_gives_up = getenv("QPN_TEST_PIPE_GIVES_UP") != nullptr;
#endif
    }
    rc = build_program(h);
    h->decode_ok = rc == QPN_OK;
    if (rc != QPN_OK) h->decode_err = g_err;            // reported by the decode entry points; the training path has its own limits
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        // geometry-only handle: usable for qpn_param_count-style queries, every compute call fails loudly
        h->device = -1; *out = h; return QPN_OK;
    }
    QPN_HIP(hipGetDevice(&h->device));
    QPN_HIP(hipDeviceGetAttribute(&h->n_cus, hipDeviceAttributeMultiprocessorCount, h->device));
    if (h->n_cus < 1) h->n_cus = 1;
    h->pipe_rows = qpn_pipe_rows_resident(h->n_cus);
    h->pipe_nu = 3;                                      // most utterances a five-role group steps alternately when the batch exceeds the groups
    if (const char* e = getenv("QPN_PIPE_NU")) { const int v = atoi(e); if (v >= 2 && v <= 3) h->pipe_nu = v; }
    *out = h;
    return QPN_OK;
}

extern "C" void qpn_destroy(qpn_handle* h) {
    if (!h) return;
    if (h->device >= 0) {
        void* bufs[] = {h->d_map, h->d_wpk, h->d_tasks, h->d_qb, h->d_bd, h->d_status, h->d_bias_src, h->d_pproj, h->d_ring, h->d_known, h->d_utts, h->d_xch};
        for (void* b : bufs) if (b) (void)hipFree(b);
        qpn_train_destroy(h->train);
        if (h->ev0) (void)hipEventDestroy(h->ev0);
        if (h->ev1) (void)hipEventDestroy(h->ev1);
        if (h->h_utts_pinned) (void)hipHostFree(h->h_utts_pinned);
        if (h->dec_side) (void)hipStreamDestroy(h->dec_side);
        if (h->dec_fork) (void)hipEventDestroy(h->dec_fork);
        if (h->dec_join) (void)hipEventDestroy(h->dec_join);
    }
    delete h;
}

static int need_device(qpn_handle* h) {
    if (!h) { qpn_set_error("null handle"); return QPN_EINVAL; }
    if (h->device < 0) { qpn_set_error("no HIP device: libqpnet_hip has no CPU fallback"); return QPN_ENODEV; }
    return QPN_OK;
}

extern "C" int qpn_set_weights(qpn_handle* h, const float* d_flat, size_t n, void* stream_) {
    int rc = need_device(h); if (rc) return rc;
    if (!h->decode_ok) { qpn_set_error("%s", h->decode_err.c_str()); return QPN_EINVAL; }
    if (!d_flat || (int64_t)n != h->g.n_params) { qpn_set_error("flat parameter vector must have %lld floats, got %zu", (long long)h->g.n_params, n); return QPN_EINVAL; }
    hipStream_t stream = (hipStream_t)stream_;
    const Geom& g = h->g;
    const size_t nmap = h->h_map.size();
    if (!h->d_map) {
        QPN_HIP(hipMalloc(&h->d_map, nmap * sizeof(int)));
        QPN_HIP(hipMalloc(&h->d_wpk, nmap * sizeof(float)));
        QPN_HIP(hipMalloc(&h->d_tasks, h->h_tasks.size() * sizeof(Task)));
        QPN_HIP(hipMalloc(&h->d_qb, (size_t)g.L * 2 * g.C * sizeof(float)));
        QPN_HIP(hipMalloc(&h->d_bd, (size_t)g.L * sizeof(BiasDesc)));
        QPN_HIP(hipMalloc(&h->d_status, 64 + 128 * QPN_NW * sizeof(long long)));
        QPN_HIP(hipMalloc(&h->d_bias_src, h->h_bias_src.size() * sizeof(int)));
        QPN_HIP(hipMemcpy(h->d_bias_src, h->h_bias_src.data(), h->h_bias_src.size() * sizeof(int), hipMemcpyHostToDevice));
        QPN_HIP(hipMemcpy(h->d_map, h->h_map.data(), nmap * sizeof(int), hipMemcpyHostToDevice));
        QPN_HIP(hipMemcpy(h->d_tasks, h->h_tasks.data(), h->h_tasks.size() * sizeof(Task), hipMemcpyHostToDevice));
        std::vector<BiasDesc> bd(g.L);
        for (int l = 0; l < g.L; ++l) {
            const LayerGeom& y = g.layers[l];
            bd[l].auxb[0] = y.auxSb; bd[l].auxb[1] = y.auxTb; bd[l].convb[0] = y.bS; bd[l].convb[1] = y.bT;
            bd[l].convPb[0] = y.bSP; bd[l].convPb[1] = y.bTP; bd[l].adaptive = y.adaptive; bd[l].pad = 0;
        }
        QPN_HIP(hipMemcpy(h->d_bd, bd.data(), bd.size() * sizeof(BiasDesc), hipMemcpyHostToDevice));
        QPN_HIP(hipEventCreate(&h->ev0)); QPN_HIP(hipEventCreate(&h->ev1));
    }
    h->d_flat = d_flat;
    hipLaunchKernelGGL(k_pack_gather, dim3((unsigned)((nmap + 255) / 256)), dim3(256), 0, stream, d_flat, h->d_map, h->d_wpk, (int64_t)nmap);
    hipLaunchKernelGGL(k_fold_bias, dim3(g.L), dim3(256), g.Ap * sizeof(float), stream, (const float4*)h->d_wpk, d_flat, h->d_bd,
                       h->aux_woff4, h->aux_tiles, h->logRa, g.A, g.Ap, g.C, g.U > 0 ? d_flat + g.up_b : (const float*)nullptr, h->d_qb);
    QPN_HIP(hipGetLastError());
    h->have_weights = true;                              // (stream-ordered: later decode calls on this stream see the packed tiles)
    return QPN_OK;
}

template <class T>
static int grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap) return QPN_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    hipError_t e = hipMalloc(p, need * sizeof(T));
    if (e != hipSuccess) { qpn_set_error("hipMalloc(%zu bytes) failed: %s", need * sizeof(T), hipGetErrorString(e)); return QPN_ENOMEM; }
    *cap = need;
    return QPN_OK;
}

// One launch of the one-CU kernels (one 1024-thread workgroup per utterance) over the descriptors p.utts[0 .. n)
static int launch_one_cu(qpn_handle* h, DecodeParams& p, int n, hipStream_t stream) {
    const Geom& g = h->g;
    const bool generic = h->dk.generic;
    const bool fast64 = !generic && g.C == 64 && g.S == 256 && g.Q == 256, fast32 = !generic && g.C == 32 && g.S == 32 && g.Q == 256;
    // the specialised kernel does not use the task table: the residual-1x1 tiles of layers 0..L-2 take its place (and more) when they fit
    const int stamp_floats = p.stamps ? 120 * QPN_NW : 0;
    const int nres_tiles = (g.C * g.C / 1024) * (g.L - 1);
    p.o_wres = (p.o_tasks + 3) & ~3;
    const bool resl = (fast64 || fast32) && g.L >= 2 && !h->dk.no_resl &&
                      ((size_t)p.o_wres + (size_t)nres_tiles * 1024 + stamp_floats) * sizeof(float) <= 160 * 1024;
    const int use_floats = resl ? p.o_wres + nres_tiles * 1024 : p.lds_floats;
    p.o_stamp = use_floats;
    const size_t lds_bytes = ((size_t)use_floats + stamp_floats) * sizeof(float);
    const void* kfn = fast64 ? (resl ? (const void*)k_decode_fast<64, 256, 256, 16, true> : (const void*)k_decode_fast<64, 256, 256, 16, false>)
                    : fast32 ? (resl ? (const void*)k_decode_fast<32, 32, 256, 16, true> : (const void*)k_decode_fast<32, 32, 256, 16, false>)
                    : (const void*)k_decode;
    if (lds_bytes > 48 * 1024) QPN_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (fast64) {
        if (resl) hipLaunchKernelGGL((k_decode_fast<64, 256, 256, 16, true>), dim3(n), dim3(1024), lds_bytes, stream, p, h->fp);
        else hipLaunchKernelGGL((k_decode_fast<64, 256, 256, 16, false>), dim3(n), dim3(1024), lds_bytes, stream, p, h->fp);
    } else if (fast32) {
        if (resl) hipLaunchKernelGGL((k_decode_fast<32, 32, 256, 16, true>), dim3(n), dim3(1024), lds_bytes, stream, p, h->fp);
        else hipLaunchKernelGGL((k_decode_fast<32, 32, 256, 16, false>), dim3(n), dim3(1024), lds_bytes, stream, p, h->fp);
    } else
        hipLaunchKernelGGL(k_decode, dim3(n), dim3(QPN_NT), lds_bytes, stream, p);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}

// force_one_cu: the retry of qpn_decode_finish (a multi-workgroup launch gave up) -- one-CU kernels only
static int decode_enqueue_impl(qpn_handle* h, int B, int n_x, int64_t F, int64_t Td,
                               const int64_t* d_x, const float* d_h, const void* d_dfac, int d_is_f32,
                               const int64_t* h_n_samples, int maxd, int mode, uint64_t seed,
                               const int64_t* d_teacher, int64_t* d_out, float* d_logits, hipStream_t stream, bool force_one_cu, int coop_limit) {
    const Geom& g = h->g;
    int rc;
    int64_t max_n = 0;
    for (int b = 0; b < B; ++b) {
        int64_t n = h_n_samples[b];
        if (n < 0) { qpn_set_error("negative n_samples"); return QPN_EINVAL; }
        if (n_x - 1 + n > Td || (g.U > 0 ? (n_x - 1 + n > F * g.U) : (n_x - 1 + n > F))) {
            qpn_set_error("row %d: n_samples=%lld exceeds the features/dilated factors provided", b, (long long)n); return QPN_EINVAL;
        }
        max_n = std::max(max_n, n);
    }
    // receptive field and left padding (qpnet.py:351-357)
    const int64_t RF = (int64_t)g.recA * maxd + g.recF + 1;
    int64_t n_pad = RF - n_x + 1; if (n_pad < 0) n_pad = 0;
    const int64_t n0 = n_pad + n_x;
    if (n0 + max_n >= ((int64_t)1 << 31)) { qpn_set_error("sequence too long"); return QPN_EINVAL; }
    // ring geometry
    DecodeParams p = h->dp;
    size_t ring_floats = 0;
    for (int l = 0; l < g.L; ++l) {
        const LayerGeom& y = g.layers[l];
        int len = (y.adaptive ? y.dilation * maxd : y.dilation) + 1;
        p.rings[l].base = (int)ring_floats; p.rings[l].len = len; p.rings[l].mult = y.dilation; p.rings[l].adaptive = y.adaptive;
        ring_floats += (size_t)len * g.C;
    }
    ring_floats = (ring_floats + 63) & ~(size_t)63;
    // several workgroups per utterance when one CU cannot hold the step state (or QPN_DECODE_COOP=<G> asks for it)
    int coopG = h->dk.coop;
    // (default: up to half the CUs for one utterance -- measured at C = 512, B = 1: 96.7 / 91.6 / 93.1 us per sample with 64 / 128 / 256 workgroups; QPN_DECODE_COOP=<G> asks for more)
    if (!h->single_cu_ok && coopG == 0) coopG = h->n_cus >= 128 ? h->n_cus / 2 : h->n_cus;
    if (force_one_cu && h->single_cu_ok) coopG = 0;
    // wide geometries, batches: the utterances batched into the contractions (decode_coopb.hip) -- up to 16 per group of n_resch / 8 workgroups
    // (a launch holds 16 utterances per group and as many groups as fit the chip: larger batches take several launches of equal size, longest rows first)
    const bool coopb = coopG > 0 && !force_one_cu && coop_limit == 0 && h->cb_ok && h->dk.coopb > 0 && B >= h->dk.coopb && h->n_cus >= g.C / 8;
    const int cb_cap = 16 * (h->n_cus / (g.C / 8) > 0 ? h->n_cus / (g.C / 8) : 1), cb_launches = (B + cb_cap - 1) / cb_cap, cb_rows = (B + cb_launches - 1) / cb_launches;
    if (coopG > 0) {
        int cap = coopG; if (B < h->n_cus && h->n_cus / B < cap) cap = h->n_cus / B;       // the whole batch in one launch when it fits the chip
        if (coop_limit > 0 && cap > coop_limit) cap = coop_limit;                          // retry with fewer workgroups per utterance
        coopG = qpn_coop_group_size(g, cap < 1 ? 1 : cap);
    }
    // ---- launch plan.  Rows are handed to the kernels longest first (descriptors are a permutation of the batch):
    //   PIPE  rows [0, n_pipe) in waves of <= pipe_rows (five resident workgroups per utterance, decode_pipe.hip)
    //   ONE   rows [n_pipe, B) on one-CU kernels, on a side stream BESIDE the first pipelined wave when there is one
    std::vector<int> order(B);
    for (int b = 0; b < B; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b2) { return h_n_samples[a] > h_n_samples[b2]; });
    const bool pipe_ok = !coopG && !force_one_cu && h->dk.pipe != 0 && qpn_pipe_supported(g) && !h->dk.generic && h->pipe_rows >= 1;
    int n_pipe = 0, n_waves = 0, wave_rows = 0;
    if (pipe_ok) {
        const int cap = h->pipe_rows;                    // five-role groups one launch holds resident
        const bool hybrid_knob = h->dk.hybrid;      // dev aid: rows beyond `cap` on one-CU kernels beside the launch (round-3 first form)
        if (B <= cap) { n_pipe = B; n_waves = 1; wave_rows = B; }
        else if (hybrid_knob && B - cap <= h->n_cus - 5 * cap) { n_pipe = cap; n_waves = 1; wave_rows = cap; }
        else {
            // a group takes a second (third) utterance, stepped alternately with the first (a role is busy ~2 us of an utterance's ~8 us
            // step): up to pipe_nu * cap rows per launch at 10 us (12 us) per step instead of 8.4; beyond that, equal-sized launches
            // measured step times with 1 / 2 / 3 utterances per group: 8.4 / 10.0 / 15.3 us (profiles/r03_decode_batches.txt); the plan with
            // the smallest (launches x step time) wins -- three per group only pays where it saves a launch (97..144 rows on 48 groups)
            static const double step_us[4] = {0.0, 8.4, 10.0, 15.3};
            double best = 1e300;
            for (int nu_max = 2; nu_max <= h->pipe_nu; ++nu_max) {
                const int nw = (B + nu_max * cap - 1) / (nu_max * cap), per = (B + nw - 1) / nw, nu = (per + cap - 1) / cap;
                const double tt = nw * step_us[nu];
                if (tt < best) { best = tt; n_waves = nw; wave_rows = per; }
            }
            n_pipe = B;
        }
    }
    const int n_one = coopG ? 0 : B - n_pipe;
    if (!coopG) { rc = grow(&h->d_ring, &h->ring_cap, ring_floats * (size_t)B); if (rc) return rc; }      // pitch-tap histories (one-CU kernels; role S1 of the pipelined one)
    rc = grow(&h->d_pproj, &h->pproj_cap, (size_t)B * F * g.L * 2 * g.C); if (rc) return rc;
    rc = grow(&h->d_known, &h->known_cap, (size_t)B * n0); if (rc) return rc;
    rc = grow(&h->d_utts, &h->utts_cap, (size_t)B); if (rc) return rc;
    if ((size_t)B > h->h_utts_cap) {
        if (h->h_utts_pinned) (void)hipHostFree(h->h_utts_pinned);
        h->h_utts_pinned = nullptr; h->h_utts_cap = 0;
        QPN_HIP(hipHostMalloc((void**)&h->h_utts_pinned, (size_t)B * sizeof(UttDesc), hipHostMallocDefault));
        h->h_utts_cap = (size_t)B;
    }
    UttDesc* utts = h->h_utts_pinned;      // pinned and owned by the handle (one decode in flight per handle): no host synchronisation here
    for (int k = 0; k < B; ++k) {
        const int b = order[k];
        UttDesc& u = utts[k];
        u.pproj = (int64_t)b * F * g.L * 2 * g.C;
        u.dfac = (int64_t)b * Td;
        u.known = (int64_t)b * n0;
        u.teacher = d_teacher ? (int64_t)b * max_n : -1;
        u.out = (int64_t)b * max_n;
        u.logits = d_logits ? (int64_t)b * max_n * g.Q : -1;
        u.ring = (int64_t)k * ring_floats;
        u.n_pad = (int)n_pad; u.n0 = (int)n0; u.n_samples = (int)h_n_samples[b]; u.d_is_f32 = d_is_f32; u.F = F;
        u.row = b; u.pad_ = 0;
    }
    QPN_HIP(hipMemcpyAsync(h->d_utts, utts, (size_t)B * sizeof(UttDesc), hipMemcpyHostToDevice, stream));
    if (!coopG) QPN_HIP(hipMemsetAsync(h->d_ring, 0, ring_floats * (size_t)B * sizeof(float), stream));
    QPN_HIP(hipMemsetAsync(h->d_status, 0, 64, stream));
    hipLaunchKernelGGL(k_known, dim3((unsigned)((n0 + 255) / 256), B), dim3(256), 0, stream, d_x, n_x, (int)n_pad, g.Q, h->d_known);
    hipLaunchKernelGGL(k_aux_project, dim3((unsigned)F, B), dim3(256), g.Ap * sizeof(float), stream, (const float4*)h->d_wpk, d_h, F,
                       h->aux_woff4, h->aux_tiles, h->logRa, g.A, g.Ap, g.C, g.L, h->d_pproj);
    p.wpk = (const float4*)h->d_wpk; p.flat = h->d_flat; p.qb = h->d_qb; p.tasks = h->d_tasks; p.utts = h->d_utts;
    p.status = h->d_status; p.mode = mode; p.seed = seed; p.bias_src = h->d_bias_src;
    p.stamps = h->dk.stamps ? (long long*)(h->d_status + 16) : nullptr;
    p.pproj = h->d_pproj; p.dfac = d_dfac; p.known = h->d_known; p.teacher = d_teacher; p.out = d_out; p.logits = d_logits; p.ring = h->d_ring;
    QPN_HIP(hipEventRecord(h->ev0, stream));
    char plan[160];
    rc = 1;                                                           // (1: the batched kernel does not apply to this call)
    if (coopb)
        for (int first = 0; first < B; first += cb_rows) {
            DecodeParams pw = p; pw.utts = h->d_utts + first;
            rc = qpn_launch_decode_coopb(h, pw, std::min(cb_rows, B - first), stream);
            if (rc) break;                                            // (what makes it not apply does not depend on the rows: the first launch decides)
        }
    if (rc < 0) return rc;
    if (rc == 0) {
        if (cb_launches > 1) snprintf(plan, sizeof(plan), "coopb G=%d launches=%d x %d rows=%d", g.C / 8, cb_launches, cb_rows, B);
        else snprintf(plan, sizeof(plan), "coopb G=%d groups=%d x %d rows=%d", g.C / 8, h->cb_groups, h->cb_per, B);
        h->call.multi_wg = 2; h->call.coopG = coopG > 2 ? coopG : 2;      // (a launch that gives up is re-run per utterance: decode_coop.hip)
    } else if (coopG) {
        rc = qpn_launch_decode_coop(h, p, B, coopG, stream); if (rc) return rc;
        snprintf(plan, sizeof(plan), "coop G=%d rows=%d", coopG, B);
        h->call.multi_wg = coopG > 1 ? 2 : 0; h->call.coopG = coopG;
    } else {
        const bool beside = n_pipe > 0 && n_one > 0;       // hybrid: the one-CU rows run on the side stream while the pipelined wave runs
        if (beside) {
            if (!h->dec_side) {
                QPN_HIP(hipStreamCreateWithFlags(&h->dec_side, hipStreamNonBlocking));
                QPN_HIP(hipEventCreateWithFlags(&h->dec_fork, hipEventDisableTiming));
                QPN_HIP(hipEventCreateWithFlags(&h->dec_join, hipEventDisableTiming));
            }
            QPN_HIP(hipEventRecord(h->dec_fork, stream));
            QPN_HIP(hipStreamWaitEvent(h->dec_side, h->dec_fork, 0));
        }
        for (int w = 0; w < n_waves; ++w) {
            const int first = w * wave_rows, n = std::min(wave_rows, n_pipe - first);
            if (n <= 0) break;
            DecodeParams pw = p; pw.utts = h->d_utts + first;
            rc = qpn_launch_decode_pipe(h, pw, n, std::min(n, h->pipe_rows), stream); if (rc) return rc;
        }
        if (n_one > 0) {
            DecodeParams po = p; po.utts = h->d_utts + n_pipe;
            rc = launch_one_cu(h, po, n_one, beside ? h->dec_side : stream); if (rc) return rc;
        }
        if (beside) {
            QPN_HIP(hipEventRecord(h->dec_join, h->dec_side));
            QPN_HIP(hipStreamWaitEvent(stream, h->dec_join, 0));
        }
        snprintf(plan, sizeof(plan), "pipe rows=%d waves=%d x %d (%d per group); one-cu rows=%d%s", n_pipe, n_waves, wave_rows,
                 n_pipe > 0 ? (wave_rows + h->pipe_rows - 1) / h->pipe_rows : 1, n_one, beside ? " (beside)" : "");
        h->call.multi_wg = n_pipe > 0 ? 1 : 0; h->call.coopG = 0;
    }
    h->plan = plan;
    QPN_HIP(hipEventRecord(h->ev1, stream));
    h->pending = true;
    return QPN_OK;
}

extern "C" int qpn_decode_enqueue(qpn_handle* h, int B, int n_x, int64_t F, int64_t Td,
                                  const int64_t* d_x, const float* d_h, const void* d_dfac, int d_is_f32,
                                  const int64_t* h_n_samples, int maxd, int mode, uint64_t seed,
                                  const int64_t* d_teacher, int64_t* d_out, float* d_logits, void* stream_) {
    int rc = need_device(h); if (rc) return rc;
    if (!h->have_weights) { qpn_set_error("qpn_set_weights must be called before qpn_decode"); return QPN_ESTATE; }
    if (h->pending) { qpn_set_error("one decode in flight per handle: call qpn_decode_finish first"); return QPN_ESTATE; }
    if (B < 1 || n_x < 1 || F < 1 || !d_x || !d_h || !d_dfac || !h_n_samples || !d_out || maxd < 1) { qpn_set_error("bad decode arguments"); return QPN_EINVAL; }
    if (mode != QPN_MODE_ARGMAX && mode != QPN_MODE_SAMPLING) { qpn_set_error("mode must be QPN_MODE_ARGMAX or QPN_MODE_SAMPLING"); return QPN_EINVAL; }
    if (mode == QPN_MODE_SAMPLING && (h->g.Q % 64 || h->g.Q > 256)) { qpn_set_error("sampling needs n_quantize in {64,128,192,256}"); return QPN_EINVAL; }
    qpn_handle::DecodeCall& c = h->call;
    c.B = B; c.n_x = n_x; c.F = F; c.Td = Td; c.d_x = d_x; c.d_h = d_h; c.d_dfac = d_dfac; c.d_is_f32 = d_is_f32;
    c.n_samples.assign(h_n_samples, h_n_samples + B); c.maxd = maxd; c.mode = mode; c.seed = seed;
    c.d_teacher = d_teacher; c.d_out = d_out; c.d_logits = d_logits; c.multi_wg = 0; c.coopG = 0;
    return decode_enqueue_impl(h, B, n_x, F, Td, d_x, d_h, d_dfac, d_is_f32, c.n_samples.data(), maxd, mode, seed, d_teacher, d_out, d_logits,
                               (hipStream_t)stream_, false, 0);
}

extern "C" const char* qpn_last_decode_plan(qpn_handle* h) { return h ? h->plan.c_str() : ""; }

extern "C" int qpn_decode_finish(qpn_handle* h, void* stream_) {
    int rc = need_device(h); if (rc) return rc;
    if (!h->pending) { qpn_set_error("no decode in flight"); return QPN_ESTATE; }
    hipStream_t stream = (hipStream_t)stream_;
    QPN_HIP(hipStreamSynchronize(stream));
    h->pending = false;
    int status = 0;
    QPN_HIP(hipMemcpy(&status, h->d_status, sizeof(int), hipMemcpyDeviceToHost));
    QPN_HIP(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    if ((status & 4) && h->call.multi_wg) {
        // A multi-workgroup launch gave up: its workgroups were not all resident together (CU-masked or shared GPU, another
        // kernel holding CUs).  Every wait in those kernels is bounded, the grid has drained; run the batch again with a
        // smaller footprint -- the one-CU kernels for the paper-size geometry, half the workgroups per utterance otherwise.
        const qpn_handle::DecodeCall c = h->call;
        const bool to_one_cu = h->single_cu_ok;
        if (to_one_cu || c.coopG >= 2) {
            const std::string first_plan = h->plan;
            h->call.multi_wg = 0;
            rc = decode_enqueue_impl(h, c.B, c.n_x, c.F, c.Td, c.d_x, c.d_h, c.d_dfac, c.d_is_f32, c.n_samples.data(), c.maxd, c.mode, c.seed,
                                     c.d_teacher, c.d_out, c.d_logits, stream, to_one_cu, to_one_cu ? 0 : c.coopG / 2);
            if (rc) return rc;
            QPN_HIP(hipStreamSynchronize(stream));
            h->pending = false;
            h->plan = first_plan + " -> timed out, retried: " + h->plan;
            QPN_HIP(hipMemcpy(&status, h->d_status, sizeof(int), hipMemcpyDeviceToHost));
            float ms2 = 0.f; QPN_HIP(hipEventElapsedTime(&ms2, h->ev0, h->ev1)); h->last_ms += ms2;
        }
    }
    if (h->dk.stamps) {      // dev aid: stamp times (cycles, relative to the first stamp of wave 0) of step 3000
        std::vector<long long> st((size_t)120 * QPN_NW);
        QPN_HIP(hipMemcpy(st.data(), h->d_status + 16, st.size() * sizeof(long long), hipMemcpyDeviceToHost));
        for (int s = 0; s < 40; ++s) {
            fprintf(stderr, "stamp %2d:", s);
            for (int w = 0; w < QPN_NW; w += 5) fprintf(stderr, " w%-2d %7lld", w, st[(size_t)s * QPN_NW + w] - st[0]);
            fprintf(stderr, "\n");
        }
    }
    if (status & 4) { qpn_set_error("decode: a workgroup timed out waiting for its peers (is another job holding CUs of this GPU?)"); return QPN_ENODEV; }
    if (status & 1) { qpn_set_error("pitch-dependent tap left its ring buffer (dilated factor <= 0.5 or > maxd)"); return QPN_ERANGE; }
    return QPN_OK;
}

extern "C" int qpn_decode(qpn_handle* h, int B, int n_x, int64_t F, int64_t Td,
                          const int64_t* d_x, const float* d_h, const void* d_dfac, int d_is_f32,
                          const int64_t* h_n_samples, int maxd, int mode, uint64_t seed,
                          const int64_t* d_teacher, int64_t* d_out, float* d_logits, void* stream) {
    int rc = qpn_decode_enqueue(h, B, n_x, F, Td, d_x, d_h, d_dfac, d_is_f32, h_n_samples, maxd, mode, seed, d_teacher, d_out, d_logits, stream);
    if (rc) return rc;
    return qpn_decode_finish(h, stream);
}

extern "C" float qpn_last_decode_kernel_ms(qpn_handle* h) { return h ? h->last_ms : 0.0f; }

// ---------------------------------------------------------------- dilated index builders
__global__ void k_didx_train(const float* __restrict__ d, int64_t L, int dilation, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; const int b = blockIdx.y;
    if (i < L) {
        float dil = -d[(size_t)b * L + i] * (float)dilation;
        float s = __fadd_rn(dil, (float)(i - L));
        out[(size_t)b * L + i] = (int64_t)rintf(s);
    }
}
__global__ void k_didx_gen32(const float* __restrict__ d, int64_t n, int dilation, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)rintf(-d[i] * (float)dilation);
}
__global__ void k_didx_gen64(const double* __restrict__ d, int64_t n, int dilation, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)rint(-d[i] * (double)dilation);
}
static int dev_ok() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { qpn_set_error("no HIP device: libqpnet_hip has no CPU fallback"); return QPN_ENODEV; }
    return QPN_OK;
}
extern "C" int qpn_dilated_index_train(const float* d_d, int B, int64_t L, int dilation, int64_t* d_out, void* stream) {
    int rc = dev_ok(); if (rc) return rc;
    if (!d_d || !d_out || B < 1 || L < 1) { qpn_set_error("bad arguments"); return QPN_EINVAL; }
    hipLaunchKernelGGL(k_didx_train, dim3((unsigned)((L + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, d_d, L, dilation, d_out);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
extern "C" int qpn_dilated_index_gen_f32(const float* d_d, int64_t n, int dilation, int64_t* d_out, void* stream) {
    int rc = dev_ok(); if (rc) return rc;
    if (!d_d || !d_out || n < 1) { qpn_set_error("bad arguments"); return QPN_EINVAL; }
    hipLaunchKernelGGL(k_didx_gen32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_d, n, dilation, d_out);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
extern "C" int qpn_dilated_index_gen_f64(const double* d_d, int64_t n, int dilation, int32_t* d_out, void* stream) {
    int rc = dev_ok(); if (rc) return rc;
    if (!d_d || !d_out || n < 1) { qpn_set_error("bad arguments"); return QPN_EINVAL; }
    hipLaunchKernelGGL(k_didx_gen64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_d, n, dilation, d_out);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
