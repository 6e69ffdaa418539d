// decode_pipe.hip -- layer-PIPELINED autoregressive decode for the paper-size QPNet (C=64, S=256, Q=256, 4 fixed + 4
// pitch-adaptive layers): five workgroups (five CUs) per utterance, every weight tile that sits on the critical path RESIDENT
// in registers / LDS.  Replaces QPNet.batch_fast_generate (reference src/nets/qpnet.py:314-559), same arithmetic spec as
// decode.hip (bit-identical streams).
//
// Why: one CU per utterance (decode.hip) re-streams the 1.7 MB of weight tiles from L2 for every generated sample and is
// bounded by its 64 B/clk L1 port at 11.65 us per sample.  Four CUs hold ALL tiles (512 KB of VGPRs + 160 KB of LDS each):
//   S0  fixed layers 0-3: current-tap tiles in VGPRs, residual tiles in LDS, past-tap tiles streamed while it waits for its input
//   K0  skip rows of layers 0-3 (all tiles in VGPRs), fed with S0's gate vectors: the fixed-stack skip sum accF, off the critical path
//   S1  adaptive layers 4-7: same residency; the pitch-dependent history rings stay in a private, L2-resident global block
//   K   skip rows of layers 4-7 (tiles of layers 4, 5 in LDS, of 6, 7 in VGPRs) + post 1x1 #1 (VGPRs)
//   P   post 1x1 #2 (VGPRs), argmax / sampling, the two causal-conv table rows of the picked sample (tables in LDS)
// A generated sample travels S0 -> S1 -> K -> P -> S0 (S0 -> K0 -> K beside it): four hand-offs on the critical path (0.44 us each inside an XCD, 0.55 across:
// profiles/r02_hop_microbench.txt) instead of 1.7 MB through one L1 port.  Hand-offs are data-tagged 8-byte granules
// {tag = step + 1, value} written with agent-scope (sc1) stores and polled with agent-scope loads (CDNA4 guide, Guideline 16
// R2): correct for any placement; blocks g, g+8, g+16, g+24 serve one utterance so that round-robin dispatch puts them on
// one XCD (speed only).  Every wait is bounded; a timeout raises the abort flag and everyone drains.
#include "decode_dev.h"
#include "qpn_handle.h"
#include <string.h>

typedef unsigned long long u64;
#define PIPE_NT 512
#define PIPE_NW 8
#define PIPE_SPIN (1u << 22)

// exchange block of one utterance (granule offsets)
#define PX_X4 0          // [64]   S0 -> S1   layer-4 input
#define PX_ACK 64        // [1]    S1 -> S0   x4 of step t consumed (flow control while S0 is not throttled by P: warm-up)
#define PX_G 72          // [4][64] S1 -> K   gate vectors of layers 4..7
#define PX_ACCF 328      // [256]  K0 -> K    skip sum of the fixed stack
#define PX_Y2 584        // [256]  K  -> P    post 1x1 #1 output
#define PX_NX 840        // [1 + 64 + 64] P -> S0  picked sample, its tap-1 table row, its tap-0 table row
#define PX_G0 976        // [4][64] S0 -> K0  gate vectors of layers 0..3
#define PX_STRIDE 1232

struct PipeParams {
    u64* xch; int* abort;
    int w_past_il[8];            // float4 offsets: past-tap tiles with (sigma_c, tanh_c) rows interleaved
    int f_resb[8], f_skipb[8], f_p1b, f_p2b;
    int nutt;                    // utterances of this launch (descriptors p.utts[0 .. nutt))
    int groups, base, rem;       // five-role groups; the LAST rem groups serve base + 1 utterances, the others base (contiguous rows, longest first)
};
// per-utterance context of a role that serves NU utterances
struct PipeUtt { UttView u; u64* X; int Ttot; };

__device__ __forceinline__ void pst(u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 pld(const u64* g) { return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#ifndef PIPE_POLL_SLEEP
#define PIPE_POLL_SLEEP 0             // s_sleep between two polls of a wait (x 64 clocks; dev: -DPIPE_POLL_SLEEP=<n> build variants, tools/pipe_poll_sleep.sh)
#endif
__device__ __forceinline__ void ppause() { if (PIPE_POLL_SLEEP) __builtin_amdgcn_s_sleep(PIPE_POLL_SLEEP); }
// wait for ONE granule of step `tag`; returns its value (on timeout / abort: raises the flag and returns what is there)
__device__ __forceinline__ float pwait(const u64* g, unsigned tag, int* abort, int* status) {
    u64 v = pld(g);
    unsigned spins = 0;
    while ((unsigned)(v >> 32) != tag) {
        if (++spins > PIPE_SPIN || ((spins & 255u) == 0 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(status, 4);
            break;
        }
        ppause();
        v = pld(g);
    }
    return __uint_as_float((unsigned)v);
}
// ... for TWO granules of the same step at once: both loads are in flight together (one round trip, not two)
__device__ __forceinline__ float2 pwait2(const u64* g0, const u64* g1, unsigned tag, int* abort, int* status) {
    u64 v0 = pld(g0), v1 = pld(g1);
    unsigned spins = 0;
    while ((unsigned)(v0 >> 32) != tag || (unsigned)(v1 >> 32) != tag) {
        if (++spins > PIPE_SPIN || ((spins & 255u) == 0 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(status, 4);
            break;
        }
        ppause();
        v0 = pld(g0); v1 = pld(g1);
    }
    return make_float2(__uint_as_float((unsigned)v0), __uint_as_float((unsigned)v1));
}
__device__ __forceinline__ void rd4(float4 (&x)[4], const float* v) {
    const float4* p = (const float4*)v;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = p[j];
}
__device__ __forceinline__ float red4(float a) {          // R = 4 lanes per row (K = 64)
    a = a + dpp_f<0x4E>(a); a = a + dpp_f<0xB1>(a); return a;
}
__device__ __forceinline__ float red16(float a) {         // R = 16 lanes per row (K = 256)
    a = a + dpp_f<0x128>(a); a = a + dpp_f<0x124>(a); a = a + dpp_f<0x4E>(a); a = a + dpp_f<0xB1>(a); return a;
}

// ------------------------------------------------------------------------------------------------ S0 / S1: a four-layer stack
// ADAPT = false: layers 0..3 (fixed taps, history in LDS); true: layers 4..7 (pitch-dependent taps, history in the global ring block)
// NU utterances share the role's resident tiles and are stepped alternately (A(t), B(t), A(t+1), ...): a role is busy ~2 us of an
// utterance's ~8 us step, so the second utterance's work runs where the first one's sample is on its way round the other roles
template <bool ADAPT, int NU>
__device__ __forceinline__ void stack_role(const DecodeParams& p, const FastParams& f, const PipeParams& pp, const PipeUtt (&cx)[NU]) {
    constexpr int C = 64, S = 256, L0 = ADAPT ? 4 : 0;
    float* sm = SM; int* smi = SMI;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 3, grp = lane >> 2;
    const int Q = p.Q;
    // LDS (floats), per utterance: xbuf[5][64] | g[4][64] | xp[4][64] | hist[4][16][64] (fixed stack) | t0row[64] | misc[8]; then wres[4 layers][4 tiles][1024]
    constexpr int o_x = 0, o_g = 320, o_xp = 576, o_hist = 832, o_t0 = o_hist + 4 * 16 * 64, o_misc = o_t0 + 64, o_pd = (o_misc + 8 + 3) & ~3,
                  o_ublk = o_pd + (NU > 1 ? 4 * 4 * 64 : 0), o_br = NU * o_ublk, o_wres = o_br + 4 * 64;
    // (o_pd, two utterances only: [layer][pd.sigma | pd.tanh | aux.sigma | aux.tanh][64] -- the coming step's past-tap dots and aux terms
    //  live in LDS there; in registers next to the second utterance's they push the resident current-tap tiles into scratch memory)
    for (int i = tid; i < o_br; i += PIPE_NT) sm[i] = 0.0f;
    for (int i = tid; i < 4 * 64; i += PIPE_NT) sm[o_br + i] = p.flat[pp.f_resb[L0 + (i >> 6)] + (i & 63)];      // residual biases (registers are tight: 2-3 spilled otherwise)
    {   // residual 1x1 tiles of my layers -> LDS (layer 7's residual output is never used)
        float4* dst = (float4*)(sm + o_wres);
        for (int i = tid; i < 4 * 4 * 256; i += PIPE_NT) { const int l = i >> 10; dst[i] = p.wpk[f.w_res[L0 + l] + (i & 1023)]; }
    }
    // FOUR waves run the stack (waves 4..7 leave after the set-up): lane (wave, grp, q) owns channel ch = 16*wave + grp, i.e. BOTH its
    // sigma row and its tanh row (rows 2ch, 2ch+1 of the interleaved matrices: tile ch/8, rows 2(ch%8), +1).  The two 16-deep chains
    // issue as v_pk_fma_f32, the gate needs no cross-lane exchange, and with one wave per SIMD a dependent VALU chain issues every
    // 4 cycles instead of every 8 (two waves per SIMD take turns): the gate stage was 700 of a layer's 1100 cycles.
    typedef float f2v __attribute__((ext_vector_type(2)));
    const int zch = (wave & 3) * 16 + grp;
    const int ztile = zch >> 3, zls = (2 * (zch & 7)) * 4 + q;          // tile, and the lane of load_tile's layout that holds my sigma row (+4: tanh row)
    f2v wcp[4][16];
#pragma unroll
    for (int l = 0; l < 4; ++l) {
        const float4* tb = p.wpk + f.w_cur[L0 + l] + ztile * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 a = tb[j * 64 + zls], b = tb[j * 64 + zls + 4];
            wcp[l][4 * j + 0] = (f2v){a.x, b.x}; wcp[l][4 * j + 1] = (f2v){a.y, b.y}; wcp[l][4 * j + 2] = (f2v){a.z, b.z}; wcp[l][4 * j + 3] = (f2v){a.w, b.w};
        }
    }
    const int rrow = zch;
    const float cbias = (!ADAPT && tid < C) ? p.flat[p.causal_b + tid] : 0.0f;
    int Tmax = 0;
#pragma unroll
    for (int v = 0; v < NU; ++v) Tmax = cx[v].Ttot > Tmax ? cx[v].Ttot : Tmax;
    __syncthreads();
    if (Tmax < 3 || wave >= 4) return;                          // (a finished wave no longer counts at the workgroup's barriers)
    f2v pdv[NU > 1 ? 1 : NU][4], auxv[NU > 1 ? 1 : NU][4];       // past-tap dots / aux terms of the coming step, per layer: {sigma row, tanh row}
    auto prepare = [&](const PipeUtt& c, int vb, int t, f2v (&pd)[4], f2v (&ax)[4]) {      // everything step t needs that does not depend on step t's own input
        const UttView& u = c.u;
        const int ut = aux_time(u, t);
        int fr, j;
        if (ut < 0) { fr = 0; j = 0; }
        else if (p.U > 0) { fr = (int)((unsigned)ut / (unsigned)p.U); j = ut - fr * p.U; }
        else { fr = ut; j = 0; }
        const float wj = p.U > 0 ? p.flat[p.up_w + j] : 1.0f;
        const float* pf = u.pproj + (size_t)fr * p.L * 2 * C;
#pragma unroll
        for (int l = 0; l < 4; ++l) ax[l] = (f2v){__builtin_fmaf(wj, pf[(L0 + l) * 2 * C + zch], p.qb[(L0 + l) * 2 * C + zch]),
                                                  __builtin_fmaf(wj, pf[(L0 + l) * 2 * C + C + zch], p.qb[(L0 + l) * 2 * C + C + zch])};
        // past rows x_l[t - off] -> LDS xp
        if (ADAPT) {
            const int widx = t < u.n0 - 1 ? t - (u.n0 - 1) : 0;
            if (tid < 4 * C) {
                const int l = tid >> 6, c2 = tid & 63;
                const RingDesc r = p.rings[L0 + l];
                int off = tap_offset(r, u, ut, widx);
                if (off < 1 || off >= r.len) { if (c2 == 0) atomicOr(p.status, 1); off = off < 1 ? 1 : r.len - 1; }
                const int tp = t - off;
                const int slot = tp >= 0 ? (int)((unsigned)tp % (unsigned)r.len) : tp + r.len;      // < 0: a never-written (zero) slot
                sm[vb + o_xp + l * C + c2] = ld_agent(u.ring + r.base + (size_t)slot * C + c2);
            }
        } else {
            if (tid < 4 * C) {
                const int l = tid >> 6, c2 = tid & 63;
                const int tp = t - (1 << l);
                sm[vb + o_xp + l * C + c2] = tp >= 1 ? sm[vb + o_hist + (l * 16 + (tp & 15)) * C + c2] : 0.0f;     // time 0 and before: zeros
            }
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < 4; ++l) {
            float4 x[4], wqs[4], wqt[4];
            const float4* tb = p.wpk + pp.w_past_il[L0 + l] + ztile * 256;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) { wqs[j2] = tb[j2 * 64 + zls]; wqt[j2] = tb[j2 * 64 + zls + 4]; }
            rd4(x, sm + vb + o_xp + l * C + 16 * q);
            pd[l] = (f2v){red4(chunk16(wqs, x)), red4(chunk16(wqt, x))};
        }
        if (NU > 1 && q == 0) {
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                float* d = sm + vb + o_pd + l * 256 + zch;
                d[0] = pd[l].x; d[64] = pd[l].y; d[128] = ax[l].x; d[192] = ax[l].y;
            }
        }
    };
#pragma unroll
    for (int v = 0; v < NU; ++v) {
        if (cx[v].Ttot >= 3) prepare(cx[v], v * o_ublk, 1, pdv[NU > 1 ? 0 : v], auxv[NU > 1 ? 0 : v]);
        if (!ADAPT && tid == 0 && cx[v].Ttot >= 3) { smi[v * o_ublk + o_misc] = cx[v].u.known[0]; }        // s[t-1] of step 1
    }
    __syncthreads();
    for (int t = 1; t + 1 < Tmax; ++t) {
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll
        for (int v = 0; v < NU; ++v) {
            if (NU > 1 && t + 1 >= cx[v].Ttot) continue;              // this utterance is done (wave-uniform; with one utterance the loop bound says so)
            const UttView& u = cx[v].u; u64* X = cx[v].X;
            const int vb = v * o_ublk;
            const bool gen = t >= u.n0 - 1;                           // this step's output is a generated sample
            // ---------------- A. this step's input
            if (!ADAPT) {
                if (wave == 0) {
                    float vv;
                    if (t <= u.n0 - 1) {                              // known sample: two rows of the causal table from memory (qpnet.py:110-132)
                        const int sp = smi[vb + o_misc], sc = u.known[t];
                        vv = p.flat[p.causal_w + ((size_t)lane * Q + sp) * 2] + p.flat[p.causal_w + ((size_t)lane * Q + sc) * 2 + 1];
                        if (lane == 0) smi[vb + o_misc] = sc;
                        if (t == u.n0 - 1) sm[vb + o_t0 + lane] = p.flat[p.causal_w + ((size_t)lane * Q + sc) * 2];     // tap-0 row of the last known sample
                    } else {                                          // picked by P at step t-1: id + its tap-1 row + its tap-0 row (for the next step)
                        const u64* nx = X + PX_NX;
                        const float2 rows = pwait2(nx + 1 + lane, nx + 65 + lane, (unsigned)t, pp.abort, p.status);     // tap-1 row, tap-0 row (for the next step)
                        vv = sm[vb + o_t0 + lane] + rows.x;
                        sm[vb + o_t0 + lane] = rows.y;
                    }
                    vv = vv + cbias;
                    sm[vb + o_x + lane] = vv;
                    sm[vb + o_hist + (0 * 16 + (t & 15)) * C + lane] = vv;
                }
                if (wave == 1 && lane == 0 && t > 1) pwait(X + PX_ACK, (unsigned)t, pp.abort, p.status);     // S1 took x4 of step t-1: the slot is free
            } else {
                if (wave == 0) {
                    const float vv = pwait(X + PX_X4 + lane, tag, pp.abort, p.status);
                    sm[vb + o_x + lane] = vv;
                    const RingDesc r = p.rings[L0];
                    st_agent(u.ring + r.base + (size_t)((unsigned)t % (unsigned)r.len) * C + lane, vv);
                    if (lane == 0) pst(X + PX_ACK, tag, 0.0f);
                }
            }
            wg_barrier();
            // ---------------- B. four gated residual blocks, tiles resident
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                {
                    float4 x[4]; rd4(x, sm + vb + o_x + l * C + 16 * q);
                    const float xe[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w, x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
                    f2v acc = wcp[l][0] * (f2v){xe[0], xe[0]};                    // the spec's chunk per row: one product, then 15 fma in k order
#pragma unroll
                    for (int e = 1; e < 16; ++e) acc = __builtin_elementwise_fma(wcp[l][e], (f2v){xe[e], xe[e]}, acc);
                    float pds, pdt, axs, axt;
                    if (NU > 1) { const float* d = sm + vb + o_pd + l * 256 + zch; pds = d[0]; pdt = d[64]; axs = d[128]; axt = d[192]; }
                    else { pds = pdv[0][l].x; pdt = pdv[0][l].y; axs = auxv[0][l].x; axt = auxv[0][l].y; }
                    const float z = (red4(acc.x) + pds) + axs;         // sigma row
                    const float zo = (red4(acc.y) + pdt) + axt;        // tanh row of the same channel
                    if (q == 0) {
                        const float g = qgate(z, zo);
                        sm[vb + o_g + l * C + zch] = g;
                        if (gen) pst(X + (ADAPT ? PX_G : PX_G0) + l * C + zch, tag, g);
                    }
                }
                // operands of the residual phase that do not depend on the gate: requested before the barrier
                float4 wr4[4]; float xres = 0.0f, brl = 0.0f;
                const bool do_res = !(ADAPT && l == 3);
                if (do_res) {
                    const float4* tp = (const float4*)(sm + o_wres) + (l * 4 + wave) * 256 + lane;
#pragma unroll
                    for (int j = 0; j < 4; ++j) wr4[j] = tp[j * 64];
                    xres = sm[vb + o_x + l * C + rrow];
                    brl = sm[o_br + l * 64 + rrow];
                }
                wg_barrier();                                         // LDS only: the hand-off stores stay in flight
                if (do_res) {                                         // residual 1x1 (+ residual add): next layer's input
                    float4 x[4];
                    rd4(x, sm + vb + o_g + l * C + 16 * q);
                    const float acc = red4(chunk16(wr4, x));
                    if (q == 0) {
                        const float vv = (acc + brl) + xres;
                        if (!ADAPT && l == 3) pst(X + PX_X4 + rrow, tag, vv);            // hand x_4 to the adaptive stack
                        else {
                            sm[vb + o_x + (l + 1) * C + rrow] = vv;
                            if (ADAPT) { const RingDesc r = p.rings[L0 + l + 1]; st_agent(u.ring + r.base + (size_t)((unsigned)t % (unsigned)r.len) * C + rrow, vv); }
                            else sm[vb + o_hist + ((l + 1) * 16 + (t & 15)) * C + rrow] = vv;
                        }
                    }
                }
                if (l < 3) wg_barrier();
            }
        }
        // ---------------- C. off the critical path: the coming step's past-tap dots / aux terms
        if (ADAPT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // ring rows have left the wave (workgroup-scope visibility across the barrier)
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NU; ++v) if (t + 2 < cx[v].Ttot) prepare(cx[v], v * o_ublk, t + 1, pdv[NU > 1 ? 0 : v], auxv[NU > 1 ? 0 : v]);
        if (tid == 0) smi[o_misc + 1] = __hip_atomic_load(pp.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (smi[o_misc + 1]) break;
    }
    (void)S;
}

// ------------------------------------------------------------------------------------------------ K: adaptive skip rows + post 1x1 #1
template <int NU>
__device__ __forceinline__ void skip_post1_role(const DecodeParams& p, const FastParams& f, const PipeParams& pp, const PipeUtt (&cx)[NU]) {
    constexpr int C = 64, S = 256;
    float* sm = SM; int* smi = SMI;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 3, grp = lane >> 2, qs = lane & 15, grps = lane >> 4;
    // per utterance: g[4][64] | y1[256] | misc[8]; then the skip tiles of layers 4, 5 (128 KB) and the biases
    constexpr int o_g = 0, o_y1 = 256, o_misc = 512, o_ublk = 520, o_sk = NU * o_ublk;
    for (int i = tid; i < o_sk; i += PIPE_NT) sm[i] = 0.0f;
    {
        float4* dst = (float4*)(sm + o_sk);
        for (int i = tid; i < 2 * 16 * 256; i += PIPE_NT) { const int l = i >> 12; dst[i] = p.wpk[f.w_skip[4 + l] + (i & 4095)]; }
    }
    float4 w6[2][4], w7[2][4], wp1[8][4];                                 // layers 6, 7 and post 1x1 #1 resident in registers
#pragma unroll
    for (int j = 0; j < 2; ++j) { load_tile(w6[j], p.wpk, f.w_skip[6] + (wave + 8 * j) * 256, lane); load_tile(w7[j], p.wpk, f.w_skip[7] + (wave + 8 * j) * 256, lane); }
#pragma unroll
    for (int j = 0; j < 8; ++j) load_tile(wp1[j], p.wpk, f.w_p1 + (wave * 8 + j) * 256, lane);
    // the biases live in LDS (5 KB): 16 more registers per lane on top of the 192 of resident tiles spilled to scratch memory, and the
    // reloads sat in the per-sample loop (scratch_load + s_waitcnt in front of the skip sums and the post-1 epilogue)
    constexpr int o_bs = o_sk + 2 * 16 * 1024, o_b1 = o_bs + 4 * 256;
    // layouts: skip biases [layer][wave][grp][j] (one 8-byte read per layer), post-1 biases [grps][wave][j] (two 16-byte reads)
    for (int i = tid; i < 4 * 256; i += PIPE_NT) {
        const int l = i >> 8, r = i & 255, j = r & 1, g16 = (r >> 1) & 15, w8 = r >> 5;
        sm[o_bs + i] = p.flat[pp.f_skipb[4 + l] + (w8 + 8 * j) * 16 + g16];
    }
    for (int i = tid; i < 256; i += PIPE_NT) {
        const int j = i & 7, w8 = (i >> 3) & 7, g4 = i >> 6;
        sm[o_b1 + i] = p.flat[pp.f_p1b + (w8 * 8 + j) * 4 + g4];
    }
    int Tmax = 0, t_begin = 0x7fffffff;
#pragma unroll
    for (int v = 0; v < NU; ++v) { Tmax = cx[v].Ttot > Tmax ? cx[v].Ttot : Tmax; const int tb = cx[v].u.n0 - 1 > 1 ? cx[v].u.n0 - 1 : 1; t_begin = tb < t_begin ? tb : t_begin; }
    __syncthreads();
    if (Tmax < 3) return;
    for (int t = t_begin; t + 1 < Tmax; ++t) {
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll
        for (int v = 0; v < NU; ++v) {
            if (NU > 1 && (t + 1 >= cx[v].Ttot || t < cx[v].u.n0 - 1)) continue;      // done / not generating yet (wave-uniform)
            u64* X = cx[v].X;
            const int vb = v * o_ublk;
            float acc2[2] = {0.0f, 0.0f};
            u64 af[2] = {0, 0};                                   // the fixed stack's skip sums (from K0, long before S1 is done): requested early
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (tid < C) sm[vb + o_g + l * C + tid] = pwait(X + PX_G + l * C + tid, tag, pp.abort, p.status);
                if (l == 2 && q == 0) {
                    int oz; asm volatile("v_mov_b32 %0, 0" : "=v"(oz));       // opaque zero: the lane's granule address is re-derived here (hoisted out of the
                                                                              // sample loop it is spilled to scratch memory and reloaded behind an s_waitcnt)
                    const int lo = (grp | oz) + wave * 16;                    // (a 32-bit lane offset, added to the uniform base in here)
                    af[0] = pld(X + (PX_ACCF + lo)); af[1] = pld(X + (PX_ACCF + 128 + lo));
                }
                wg_barrier();                                         // LDS only (the polls' / hand-offs' global traffic is not waited for)
                float4 x[4]; rd4(x, sm + vb + o_g + l * C + 16 * q);
                const float2 bsl = *(const float2*)(sm + o_bs + ((l * 8 + wave) * 16 + grp) * 2);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float d;
                    if (l < 2) {
                        float4 w[4];
                        const float4* tp = (const float4*)(sm + o_sk) + (l * 16 + wave + 8 * j) * 256 + lane;
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) w[jj] = tp[jj * 64];
                        d = red4(chunk16(w, x));
                    } else if (l == 2) d = red4(chunk16(w6[j], x));
                    else d = red4(chunk16(w7[j], x));
                    acc2[j] = acc2[j] + (d + (j ? bsl.y : bsl.x));
                }
            }
            if (q == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = (wave + 8 * j) * 16 + grp;
                    const float accf = (unsigned)(af[j] >> 32) == tag ? __uint_as_float((unsigned)af[j]) : pwait(X + PX_ACCF + row, tag, pp.abort, p.status);
                    const float tot = accf + acc2[j];                                               // sum(skip_F) + sum(skip_A)  (qpnet.py:505)
                    sm[vb + o_y1 + row] = tot > 0.0f ? tot : 0.0f;
                }
            }
            wg_barrier();                                         // LDS only (the polls' / hand-offs' global traffic is not waited for)
            {
                float4 x[4]; rd4(x, sm + vb + o_y1 + 16 * qs);
                float pa[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) pa[j] = red16(chunk16(wp1[j], x));
                if (qs == 0) {
                    const float4 bA = *(const float4*)(sm + o_b1 + (grps * 8 + wave) * 8), bB = *(const float4*)(sm + o_b1 + (grps * 8 + wave) * 8 + 4);
                    const float bj[8] = {bA.x, bA.y, bA.z, bA.w, bB.x, bB.y, bB.z, bB.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float vv = pa[j] + bj[j]; pst(X + PX_Y2 + (wave * 8 + j) * 4 + grps, tag, vv > 0.0f ? vv : 0.0f); }
                }
            }
        }
        if (tid == 0) smi[o_misc] = __hip_atomic_load(pp.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (smi[o_misc]) break;
    }
    (void)S;
}

// ------------------------------------------------------------------------------------------------ K0: skip rows of the fixed stack
// (a CU of its own: the fixed stack's skip sum is needed only when the adaptive stack has finished, but computing it inside S0
// either delays x_4 or arrives late at K -- measured 0.9 us of stall per sample)
template <int NU>
__device__ __forceinline__ void skip_fixed_role(const DecodeParams& p, const FastParams& f, const PipeParams& pp, const PipeUtt (&cx)[NU]) {
    constexpr int C = 64;
    float* sm = SM; int* smi = SMI;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 3, grp = lane >> 2;
    constexpr int o_g = 0, o_misc = 256, o_ublk = 264;
    for (int i = tid; i < NU * o_ublk; i += PIPE_NT) sm[i] = 0.0f;
    float4 wsk[4][2][4];                                          // two skip tiles per wave per layer, resident (128 VGPRs)
    float bsk[4][2];
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            load_tile(wsk[l][j], p.wpk, f.w_skip[l] + (wave + 8 * j) * 256, lane);
            bsk[l][j] = p.flat[pp.f_skipb[l] + (wave + 8 * j) * 16 + grp];
        }
    int Tmax = 0, t_begin = 0x7fffffff;
#pragma unroll
    for (int v = 0; v < NU; ++v) { Tmax = cx[v].Ttot > Tmax ? cx[v].Ttot : Tmax; const int tb = cx[v].u.n0 - 1 > 1 ? cx[v].u.n0 - 1 : 1; t_begin = tb < t_begin ? tb : t_begin; }
    __syncthreads();
    if (Tmax < 3) return;
    for (int t = t_begin; t + 1 < Tmax; ++t) {
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll
        for (int v = 0; v < NU; ++v) {
            if (NU > 1 && (t + 1 >= cx[v].Ttot || t < cx[v].u.n0 - 1)) continue;
            u64* X = cx[v].X;
            const int vb = v * o_ublk;
            float acc2[2] = {0.0f, 0.0f};
#pragma unroll
            for (int l = 0; l < 4; ++l) {
                if (tid < C) sm[vb + o_g + l * C + tid] = pwait(X + PX_G0 + l * C + tid, tag, pp.abort, p.status);
                wg_barrier();
                float4 x[4]; rd4(x, sm + vb + o_g + l * C + 16 * q);
#pragma unroll
                for (int j = 0; j < 2; ++j) acc2[j] = acc2[j] + (red4(chunk16(wsk[l][j], x)) + bsk[l][j]);
            }
            if (q == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) pst(X + PX_ACCF + (wave + 8 * j) * 16 + grp, tag, acc2[j]);
            }
        }
        if (tid == 0) smi[o_misc] = __hip_atomic_load(pp.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (smi[o_misc]) break;
    }
}

// ------------------------------------------------------------------------------------------------ P: post 1x1 #2, pick, causal rows
template <int NU>
__device__ __forceinline__ void post2_pick_role(const DecodeParams& p, const FastParams& f, const PipeParams& pp, const PipeUtt (&cx)[NU]) {
    constexpr int C = 64;
    float* sm = SM; int* smi = SMI;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qs = lane & 15, grps = lane >> 4;
    const int Q = p.Q;
    // per utterance: y2[256] | lg[256] | misc[8]; then tab[2][Q][64]: tap-0 rows, then tap-1 rows, one row per class
    constexpr int o_y2 = 0, o_lg = 256, o_misc = 512, o_ublk = 520, o_b2 = NU * o_ublk, o_tab = o_b2 + 256;
    for (int i = tid; i < o_b2; i += PIPE_NT) sm[i] = 0.0f;
    // post-2 biases in LDS, in the order the lanes read them ([grps][wave][j]: two 16-byte reads): eight registers the resident tiles need
    for (int i = tid; i < 256; i += PIPE_NT) { const int j = i & 7, w8 = (i >> 3) & 7, g4 = i >> 6; sm[o_b2 + i] = p.flat[pp.f_p2b + (w8 * 8 + j) * 4 + g4]; }
    for (int i = tid; i < 2 * Q * C; i += PIPE_NT) {
        const int tp = i / (Q * C), r = i - tp * Q * C, s = r / C, c = r - s * C;
        sm[o_tab + i] = p.flat[p.causal_w + ((size_t)c * Q + s) * 2 + tp];
    }
    // post 1x1 #2: this wave's eight rows resident, kept as four PAIRS of rows ({row 2i, row 2i+1} per element) so that the two
    // independent 16-deep FMA chains of a pair issue as v_pk_fma_f32 -- with two waves per SIMD the 128 scalar FMAs were the phase
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v wpp[4][16];
    {
        float4 wa[4], wb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            load_tile(wa, p.wpk, f.w_p2 + (wave * 8 + 2 * i) * 256, lane);
            load_tile(wb, p.wpk, f.w_p2 + (wave * 8 + 2 * i + 1) * 256, lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wpp[i][4 * k + 0] = (f2v){wa[k].x, wb[k].x}; wpp[i][4 * k + 1] = (f2v){wa[k].y, wb[k].y};
                wpp[i][4 * k + 2] = (f2v){wa[k].z, wb[k].z}; wpp[i][4 * k + 3] = (f2v){wa[k].w, wb[k].w};
            }
        }
    }
    int Tmax = 0, t_begin = 0x7fffffff;
#pragma unroll
    for (int v = 0; v < NU; ++v) { Tmax = cx[v].Ttot > Tmax ? cx[v].Ttot : Tmax; const int tb = cx[v].u.n0 - 1 > 1 ? cx[v].u.n0 - 1 : 1; t_begin = tb < t_begin ? tb : t_begin; }
    const bool sampling = p.mode == QPN_MODE_SAMPLING;
    __syncthreads();
    if (Tmax < 3) return;
    for (int t = t_begin; t + 1 < Tmax; ++t) {
        const unsigned tag = (unsigned)t + 1u;
#pragma unroll
        for (int v = 0; v < NU; ++v) {
            if (NU > 1 && (t + 1 >= cx[v].Ttot || t < cx[v].u.n0 - 1)) continue;
            const UttView& u = cx[v].u; u64* X = cx[v].X;
            const int vb = v * o_ublk;
            const int i = t - (u.n0 - 1);
            float uni = 0.0f;
            if (sampling && wave == 0) uni = sample_uniform(p.seed, (unsigned)u.row, (unsigned)i);   // ahead of the wait: independent of the logits
            if (wave == 0) {      // ONE wave gathers all 256 granules, four per lane in one poll loop: four waves polling on their own each
                                  // sample their granules once per L2 round trip, and the barrier waited for the unluckiest phase
                const u64* g = X + PX_Y2 + lane;
                u64 v0 = pld(g), v1 = pld(g + 64), v2 = pld(g + 128), v3 = pld(g + 192);
                unsigned spins = 0;
                while ((unsigned)(v0 >> 32) != tag || (unsigned)(v1 >> 32) != tag || (unsigned)(v2 >> 32) != tag || (unsigned)(v3 >> 32) != tag) {
                    if (++spins > PIPE_SPIN || ((spins & 255u) == 0 && __hip_atomic_load(pp.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(pp.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicOr(p.status, 4);
                        break;
                    }
                    ppause();
                    v0 = pld(g); v1 = pld(g + 64); v2 = pld(g + 128); v3 = pld(g + 192);
                }
                sm[vb + o_y2 + lane] = __uint_as_float((unsigned)v0); sm[vb + o_y2 + 64 + lane] = __uint_as_float((unsigned)v1);
                sm[vb + o_y2 + 128 + lane] = __uint_as_float((unsigned)v2); sm[vb + o_y2 + 192 + lane] = __uint_as_float((unsigned)v3);
            }
            wg_barrier();                                         // LDS only (the polls' / hand-offs' global traffic is not waited for)
            {
                float4 x[4]; rd4(x, sm + vb + o_y2 + 16 * qs);
                float pa[8];
#pragma unroll
                for (int i2 = 0; i2 < 4; ++i2) {
                    const float xe[16] = {x[0].x, x[0].y, x[0].z, x[0].w, x[1].x, x[1].y, x[1].z, x[1].w, x[2].x, x[2].y, x[2].z, x[2].w, x[3].x, x[3].y, x[3].z, x[3].w};
                    f2v acc = wpp[i2][0] * (f2v){xe[0], xe[0]};                    // the spec's chunk: one product, then 15 fma in k order -- per row, as before
#pragma unroll
                    for (int e = 1; e < 16; ++e) acc = __builtin_elementwise_fma(wpp[i2][e], (f2v){xe[e], xe[e]}, acc);
                    pa[2 * i2] = red16(acc.x); pa[2 * i2 + 1] = red16(acc.y);
                }
                if (qs == 0) {
                    const float4 bA = *(const float4*)(sm + o_b2 + (grps * 8 + wave) * 8), bB = *(const float4*)(sm + o_b2 + (grps * 8 + wave) * 8 + 4);
                    const float b2[8] = {bA.x, bA.y, bA.z, bA.w, bB.x, bB.y, bB.z, bB.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) sm[vb + o_lg + (wave * 8 + j) * 4 + grps] = pa[j] + b2[j];
                }
            }
            wg_barrier();                                         // LDS only (the polls' / hand-offs' global traffic is not waited for)
            if (wave == 0) {
                int bi;
                if (sampling) bi = sample_wave_u(vb + o_lg, Q, uni, lane);
                else {
                    float bv;
                    {   // lane owns four consecutive classes (one ds_read_b128); lowest index among maxima
                        const float4 v4 = *(const float4*)(sm + vb + o_lg + 4 * lane);
                        bv = v4.x; bi = 4 * lane;
                        if (v4.y > bv) { bv = v4.y; bi = 4 * lane + 1; }
                        if (v4.z > bv) { bv = v4.z; bi = 4 * lane + 2; }
                        if (v4.w > bv) { bv = v4.w; bi = 4 * lane + 3; }
                    }
#define PIPE_AMAX(CTRL) { const float ov = dpp_f<CTRL>(bv); const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, true); \
                          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
                    PIPE_AMAX(0xB1) PIPE_AMAX(0x4E) PIPE_AMAX(0x124) PIPE_AMAX(0x128)   // within rows of 16 lanes: quad perms, row rotations
#undef PIPE_AMAX
                    {   // across the four rows: every lane of a row holds its row's best, so four v_readlane pairs and uniform compares
                        // replace two ds_bpermute round trips (rows hold ascending class ranges: the lowest index wins ties by order)
                        float rv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), 0)); int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
                        for (int r = 1; r < 4; ++r) {
                            const float ov = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bv), 16 * r)); const int oi = __builtin_amdgcn_readlane(bi, 16 * r);
                            if (ov > rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
                        }
                        bv = rv; bi = ri;
                    }
                }
                int next = bi;
                if (u.teacher) { const int64_t sv = u.teacher[i] % Q; next = (int)(sv < 0 ? sv + Q : sv); }
                // the next step's layer-0 input needs the tap-1 row of `next`; the step after that its tap-0 row
                u64* nx = X + PX_NX;
                pst(nx + 65 + lane, tag, sm[o_tab + next * C + lane]);
                pst(nx + 1 + lane, tag, sm[o_tab + (Q + next) * C + lane]);
                if (lane == 0) { pst(nx, tag, __int_as_float(next)); u.out[i] = bi; }
                if (u.logits) for (int k = lane; k < Q; k += 64) u.logits[(size_t)i * Q + k] = sm[vb + o_lg + k];
            }
        }
        if (tid == 64) smi[o_misc] = __hip_atomic_load(pp.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (smi[o_misc]) break;
    }
}

template <int NU>
__device__ __forceinline__ void pipe_group(const DecodeParams& p, const FastParams& f, const PipeParams& pp, int role, int row0, int n_active) {
    PipeUtt cx[NU];
#pragma unroll
    for (int v = 0; v < NU; ++v) {
        const bool on = v < n_active && row0 + v < pp.nutt;
        const int row = on ? row0 + v : row0;                          // (a switched-off slot repeats the first utterance's descriptors and never runs)
        cx[v].u = make_view(p, p.utts[row]);
        cx[v].X = pp.xch + (size_t)row * PX_STRIDE;
        cx[v].Ttot = on ? cx[v].u.n0 + cx[v].u.n_samples : 0;
    }
    if (role == 0) stack_role<false, NU>(p, f, pp, cx);
    else if (role == 1) stack_role<true, NU>(p, f, pp, cx);
    else if (role == 2) skip_post1_role<NU>(p, f, pp, cx);
    else if (role == 3) post2_pick_role<NU>(p, f, pp, cx);
    else skip_fixed_role<NU>(p, f, pp, cx);
}

// 40 consecutive blocks serve 8 five-role groups; the five roles of a group are 8 blocks apart (one XCD under round-robin dispatch)
__global__ __launch_bounds__(PIPE_NT) void k_decode_pipe(DecodeParams p, FastParams f, PipeParams pp) {      // every group serves one utterance
    const int chunk = blockIdx.x / 40, within = blockIdx.x - chunk * 40, role = within >> 3, gi = chunk * 8 + (within & 7);
    if (gi >= pp.nutt) return;
    pipe_group<1>(p, f, pp, role, gi, 1);
}
// ... and the launches for more utterances than groups: a group serves up to NU of them, stepped alternately.  Kernels of their own so
// that the one-utterance roles keep their register allocation (one instantiation per role here: a group with fewer utterances runs the
// same code with slots switched off)
template <int NU>
__global__ __launch_bounds__(PIPE_NT) void k_decode_pipe_n(DecodeParams p, FastParams f, PipeParams pp) {
    const int chunk = blockIdx.x / 40, within = blockIdx.x - chunk * 40, role = within >> 3, gi = chunk * 8 + (within & 7);
    if (gi >= pp.groups) return;
    const int first_big = pp.groups - pp.rem;
    const int row0 = gi * pp.base + (gi > first_big ? gi - first_big : 0), cnt = pp.base + (gi >= first_big ? 1 : 0);
    pipe_group<NU>(p, f, pp, role, row0, cnt);
}

// ------------------------------------------------------------------------------------------------ host side
bool qpn_pipe_supported(const Geom& g) {
    if (g.C != 64 || g.S != 256 || g.Q != 256 || g.LF != 4 || g.LA != 4 || g.L != 8) return false;
    for (int l = 0; l < 8; ++l) if (g.layers[l].dilation != (1 << (l & 3)) || g.layers[l].adaptive != (l >= 4)) return false;
    return true;
}

#define PIPE_MAX_NU 3
static size_t pipe_lds_bytes(int nu) {      // the largest role: P (the causal tables) / K (two layers' skip tiles + biases) / the stacks (residual tiles + per-utterance state)
    const size_t pk = (size_t)nu * 520 + 2 * 256 * 64 + 6 * 256, st = (size_t)nu * (5000 + (nu > 1 ? 1024 : 0)) + 256 + 4 * 4 * 1024 + 64;
    return (pk > st ? pk : st) * sizeof(float);
}

// utterances one launch can serve with all 5 * rows workgroups resident TOGETHER (they spin on each other): the device's CU
// count times the blocks of k_decode_pipe a CU admits (1: 137 KB of LDS), in whole 8-utterance chunks of the block mapping
int qpn_pipe_rows_resident(int n_cus) {
    int per_cu = 0;
    (void)hipFuncSetAttribute((const void*)k_decode_pipe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe_lds_bytes(1));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_decode_pipe, PIPE_NT, pipe_lds_bytes(1)) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        return 0;
    }
    if (per_cu > 1) per_cu = 1;       // one role per CU: the roles were tuned as the only tenant of their CU
    return (n_cus * per_cu / 40) * 8;
}

// B utterances on `groups` five-role groups (B <= PIPE_MAX_NU * groups): as even as possible, the shortest rows share
int qpn_launch_decode_pipe(qpn_handle* h, DecodeParams& p, int B, int groups, hipStream_t stream) {
    const Geom& g = h->g;
    PipeParams pp; memset(&pp, 0, sizeof(pp));
    for (int l = 0; l < 8; ++l) { pp.w_past_il[l] = h->w_past_il[l]; pp.f_resb[l] = (int)g.layers[l].resb; pp.f_skipb[l] = (int)g.layers[l].skipb; }
    pp.f_p1b = (int)g.post1_b; pp.f_p2b = (int)g.post2_b; pp.nutt = B;
    if (groups > B) groups = B;
    if (groups < 1 || PIPE_MAX_NU * groups < B) { qpn_set_error("internal: %d utterances do not fit %d pipelined groups", B, groups); return QPN_EINVAL; }
    pp.groups = groups; pp.base = B / groups; pp.rem = B % groups;
    const int nu = pp.base + (pp.rem ? 1 : 0);
    const size_t xwords = (size_t)PX_STRIDE * B + 16;
    if (xwords > h->xch_cap) {
        if (h->d_xch) (void)hipFree(h->d_xch);
        h->d_xch = nullptr; h->xch_cap = 0;
        if (hipMalloc(&h->d_xch, xwords * sizeof(unsigned long long)) != hipSuccess) { qpn_set_error("hipMalloc for the decode exchange buffers failed"); return QPN_ENOMEM; }
        h->xch_cap = xwords;
    }
    pp.xch = h->d_xch + 16; pp.abort = (int*)h->d_xch;
    QPN_HIP(hipMemsetAsync(h->d_xch, 0, xwords * sizeof(unsigned long long), stream));
#ifdef QPN_TESTING
    if (h->dk.test_pipe_gives_up) {
        // test hook (a -DQPN_TESTING build only): the launch behaves as if a wait had timed out at once (abort flag raised, status bit 4): exercises the
        // re-run on the one-CU kernels without needing a CU-masked device (tests/test_decode_gpu.py)
        static const int one = 1, four = 4;
        QPN_HIP(hipMemcpyAsync(h->d_xch, &one, sizeof(int), hipMemcpyHostToDevice, stream));
        QPN_HIP(hipMemcpyAsync(h->d_status, &four, sizeof(int), hipMemcpyHostToDevice, stream));
    }
#endif
    const size_t lds = pipe_lds_bytes(nu);
    const void* kfn = nu == 1 ? (const void*)k_decode_pipe : nu == 2 ? (const void*)k_decode_pipe_n<2> : (const void*)k_decode_pipe_n<3>;
    QPN_HIP(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));    // per device: set at every launch
    const int nchunks = (groups + 7) / 8;
    if (nu == 1) hipLaunchKernelGGL(k_decode_pipe, dim3(40 * nchunks), dim3(PIPE_NT), lds, stream, p, h->fp, pp);
    else if (nu == 2) hipLaunchKernelGGL(k_decode_pipe_n<2>, dim3(40 * nchunks), dim3(PIPE_NT), lds, stream, p, h->fp, pp);
    else hipLaunchKernelGGL(k_decode_pipe_n<3>, dim3(40 * nchunks), dim3(PIPE_NT), lds, stream, p, h->fp, pp);
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
