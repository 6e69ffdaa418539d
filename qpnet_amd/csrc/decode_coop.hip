// decode_coop.hip -- autoregressive decode with SEVERAL cooperating workgroups per utterance (gfx950 / MI355X).
//
// Replaces QPNet.batch_fast_generate (reference src/nets/qpnet.py:314-559) for geometries whose per-sample state and
// weight stream are too large for the one-CU-per-utterance kernels of decode.hip: the repo-default QPNet (n_resch 512,
// 16 layers: src/utils/param_model.py:58-64) touches 96 MB of weights and 1 MB of state per generated sample.
//
// Partition: workgroup g of the G that serve an utterance owns rows [g/G, (g+1)/G) of EVERY matrix (1/G of the weight
// stream, so the G CUs stream their slices from L2 / Infinity Cache in parallel).  Each dependent stage therefore ends
// with an all-gather of a short vector (C or S floats) among the G workgroups: 2 per layer + 3 for the post-net.
// The exchange is the data-tagged granule of the CDNA4 guide (Guideline 16, R2): every float travels as ONE aligned
// 8-byte {tag = step + 1, value} agent-scope store; consumers re-read their granules until the tag matches -- no flag,
// no fence, correct for any placement of the workgroups on CUs / XCDs.  The layer-input history ("ring buffers") IS the
// exchange buffer of x_l: slot t % len of ring l receives step t's x_l and is re-read later as the pitch-dependent tap.
// Buffers are reused once per generated sample; the sample-to-sample dependency (every workgroup needs ALL logits of
// step t before it can produce anything of step t+1) is the barrier that makes the reuse safe.
// Every wait is bounded: a timeout raises the abort flag, all workgroups drain and the host reports an error.
//
// Arithmetic: the fixed-order "QPNet-f32" spec -- bit-identical to the CPU oracle and to decode.hip.
#include "decode_dev.h"
#include "qpn_handle.h"
#include <string.h>

typedef unsigned long long u64;
#ifndef COOP_NT
#define COOP_NT 768      // 12 waves: 1024 threads cap the kernel at 128 registers (29 spilled, 112 bytes of scratch per lane re-read inside the
#endif                   // per-sample loop); measured at C = 512: 8.98 / 52.9 / 81.5 k samples/s for 1 / 8 / 20 utterances vs 8.84 / 45.4 / 75.2 k (512 threads: 8.69 / 49.9 / 66.7 k)
#define COOP_NW (COOP_NT / 64)
#define COOP_SPIN_LIMIT (1u << 22)

__device__ __forceinline__ void gr_store(u64* g, unsigned tag, float v) {
    __hip_atomic_store(g, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 gr_load(const u64* g) {
    return __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The first ceil(n / 256) waves fetch n granules of step `tag` into LDS: four granules per lane per sweep, re-read until every tag
// matches (a few polling waves instead of one polling lane per granule: pollers next to a weight stream cost bandwidth)
__device__ __forceinline__ void gather_vec(const u64* src, int n, unsigned tag, float* dst, int tid, int* abort, int* status) {
    const int base = (tid >> 6) * 256 + (tid & 63);
    if ((tid >> 6) * 256 >= n) return;
    u64 v[4];
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + 64 * k;
            v[k] = i < n ? gr_load(src + i) : ((u64)tag << 32);
            ok &= (unsigned)(v[k] >> 32) == tag;
        }
        if (__all(ok)) break;
        if (++spins > COOP_SPIN_LIMIT || ((spins & 255u) == 0 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(status, 4);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = base + 64 * k; if (i < n) dst[i] = __uint_as_float((unsigned)v[k]); }
}

__global__ __launch_bounds__(COOP_NT) void k_decode_coop(DecodeParams p, FastParams f, CoopParams c, int b0) {
    float* sm = SM; int* smi = SMI;
    const int gidx = blockIdx.x, b = b0 + blockIdx.y;
    const UttView u = make_view(p, p.utts[b]);
    u64* X = c.xch + (size_t)blockIdx.y * c.utt_stride;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, Cp = p.Cp, S = p.S, Q = p.Q, L = p.L;
    const int CB = c.CB, SB = c.SB, QB = c.QB;
    const int c0 = gidx * CB, s0 = gidx * SB, q0 = gidx * QB;
    const int logR = c.logR, R = 1 << logR, rpt = c.rpt, logRs = c.logRs, Rs = 1 << logRs, rpts = c.rpts;
    // LDS layout (floats)
    int o = 0;
    const int o_xa = o; o += Cp; const int o_xb = o; o += Cp; const int o_xp = o; o += L * Cp; const int o_g = o; o += Cp;
    const int o_y1 = o; o += c.Sp; const int o_y2 = o; o += c.Sp; const int o_lg = o; o += (Q + 3) & ~3;
    const int o_acc = o; o += (2 * SB + 3) & ~3; const int o_aux = o; o += L * 2 * CB; const int o_samp = o; o += 4;
    // the biases my rows add, in LDS: read from global memory where they are used, each sat behind its own wait right after a row's
    // reduction (an exposed L2 round trip per tile pass)
    const int o_bres = o; o += L * CB; const int o_bsk = o; o += L * SB; const int o_bp1 = o; o += SB; const int o_bp2 = o; o += QB;
    for (int i = tid; i < o_bres; i += COOP_NT) sm[i] = 0.0f;
    for (int i = tid; i < L * CB; i += COOP_NT) sm[o_bres + i] = p.flat[c.f_resb[i / CB] + c0 + i % CB];
    for (int i = tid; i < L * SB; i += COOP_NT) sm[o_bsk + i] = p.flat[c.f_skipb[i / SB] + s0 + i % SB];
    for (int i = tid; i < SB; i += COOP_NT) sm[o_bp1 + i] = p.flat[c.f_p1b + s0 + i];
    for (int i = tid; i < QB; i += COOP_NT) sm[o_bp2 + i] = p.flat[c.f_p2b + q0 + i];
    __syncthreads();
    const int Ttot = u.n0 + u.n_samples;
    if (Ttot < 3) return;
    if (tid == 0) { smi[o_samp] = u.known[0]; smi[o_samp + 1] = u.known[1]; }
    __syncthreads();
    const int q = lane & (R - 1), grp = lane >> logR, qs = lane & (Rs - 1), grps = lane >> logRs;
    const int n_aux = L * 2 * CB;
    for (int t = 1; t + 1 < Ttot; ++t) {
        const unsigned tag = (unsigned)t + 1u;
        // ---------------- A. this step's layer-0 input (two rows of the causal table, qpnet.py:110-132), aux terms, past rows
        {
            const int s_prev = smi[o_samp], s_cur = smi[o_samp + 1];
            const RingDesc r0 = p.rings[0];
            for (int ch = tid; ch < C; ch += COOP_NT) {
                float v = p.flat[p.causal_w + ((size_t)ch * Q + s_prev) * 2] + p.flat[p.causal_w + ((size_t)ch * Q + s_cur) * 2 + 1];
                v = v + p.flat[p.causal_b + ch];
                sm[o_xa + ch] = v;
                if (ch >= c0 && ch < c0 + CB) gr_store(X + c.o_ring[0] + (size_t)((unsigned)t % (unsigned)r0.len) * C + ch, tag, v);
            }
            const int ut = aux_time(u, t);
            int fr, j;
            if (ut < 0) { fr = 0; j = 0; }
            else if (p.U > 0) { fr = (int)((unsigned)ut / (unsigned)p.U); j = ut - fr * p.U; }
            else { fr = ut; j = 0; }
            const float wj = p.U > 0 ? p.flat[p.up_w + j] : 1.0f;
            const float* pf = u.pproj + (size_t)fr * L * 2 * C;
            for (int i = tid; i < n_aux; i += COOP_NT) {
                const int l = i / (2 * CB), r = i - l * 2 * CB, half = r / CB, nat = half * C + c0 + (r - half * CB);
                sm[o_aux + i] = __builtin_fmaf(wj, pf[l * 2 * C + nat], p.qb[l * 2 * C + nat]);
            }
            const int widx = t < u.n0 - 1 ? t - (u.n0 - 1) : 0;
            for (int i = tid; i < L * C; i += COOP_NT) {
                const int l = i / C, ch = i - l * C;
                const RingDesc r = p.rings[l];
                int off = tap_offset(r, u, ut, widx);
                if (off < 1 || off >= r.len) { if (ch == 0) atomicOr(p.status, 1); off = off < 1 ? 1 : r.len - 1; }
                const int tp = t - off;                   // a row of an earlier step (slot tp carries tag tp + 1)
                float v = 0.0f;                           // time 0 and before: never written, zeros
                if (tp >= 1) {
                    const u64* src = X + c.o_ring[l] + (size_t)((unsigned)tp % (unsigned)r.len) * C + ch;
                    u64 gv = gr_load(src);
                    unsigned spins = 0;
                    while ((unsigned)(gv >> 32) != (unsigned)tp + 1u && ++spins < 4096u) gv = gr_load(src);     // published a whole step ago: normally no spin
                    if (spins >= 4096u) { __hip_atomic_store(c.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); atomicOr(p.status, 4); }
                    v = __uint_as_float((unsigned)gv);
                }
                sm[o_xp + l * Cp + ch] = v;
            }
            for (int i = tid; i < 2 * SB; i += COOP_NT) sm[o_acc + i] = 0.0f;
        }
        __syncthreads();
        int xin = o_xa, xnext = o_xb;
        const int npair = (2 * CB) / rpt, zt0 = (2 * c0) / rpt;
        // the tiles of a phase's FIRST pass are requested before the wait that precedes the phase (they do not depend on it)
        float4 wcN[4], wqN[4], wrN[4];
        if (wave < npair) { load_tile(wcN, p.wpk, f.w_cur[0] + (zt0 + wave) * 256, lane); load_tile(wqN, p.wpk, c.w_past_il[0] + (zt0 + wave) * 256, lane); }
        for (int l = 0; l < L; ++l) {
            // ---------------- Z: pre-activations of my channels (current + past tap + aux), gate
            for (int i = wave; i < npair; i += COOP_NW) {
                const int ti = zt0 + i;
                float4 x[4], xp[4];
                if (i != wave) {                              // (the first pass's tiles were requested before the wait)
                    load_tile(wcN, p.wpk, f.w_cur[l] + ti * 256, lane);
                    load_tile(wqN, p.wpk, c.w_past_il[l] + ti * 256, lane);
                }
                const float4* xv = (const float4*)(sm + xin + 16 * q);
                const float4* pv = (const float4*)(sm + o_xp + l * Cp + 16 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) { x[k] = xv[k]; xp[k] = pv[k]; }
                const float ac = tree_reduce(chunk16(wcN, x), logR);
                const float ap = tree_reduce(chunk16(wqN, xp), logR);
                const int row = ti * rpt + grp, ch = row >> 1, half = row & 1;
                const float z = (ac + ap) + sm[o_aux + l * 2 * CB + half * CB + (ch - c0)];
                float zo;                                     // the tanh row of the same channel (R lanes up)
                if (R == 32) { const int zi = __float_as_int(z); const auto sw = __builtin_amdgcn_permlane32_swap(zi, zi, false, false); zo = __int_as_float((int)sw[1]); }
                else zo = __shfl_down(z, R);
                if (q == 0 && !half) gr_store(X + c.o_g + (size_t)l * C + ch, tag, qgate(z, zo));
            }
            const bool has_res = l + 1 < L;               // the last block's residual output is unused (qpnet.py:505)
            // (a slice smaller than a 4 KiB tile -- G = 128 / 256 at C = 512: one or two skip / post-net rows per workgroup -- takes the tile that holds its rows
            //  and keeps the lane groups of those rows)
            const int nres = has_res ? CB / rpt : 0, nsk = (SB + rpt - 1) / rpt;
            if (wave < nres + nsk)
                load_tile(wrN, p.wpk, wave < nres ? f.w_res[l] + (c0 / rpt + wave) * 256 : f.w_skip[l] + (s0 / rpt + (wave - nres)) * 256, lane);
            gather_vec(X + c.o_g + (size_t)l * C, C, tag, sm + o_g, tid, c.abort, p.status);
            __syncthreads();
            // ---------------- R: residual 1x1 rows of my channels (-> next layer's input), skip 1x1 rows of my slice
            {
                float4 xg[4];
                const float4* gv = (const float4*)(sm + o_g + 16 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) xg[k] = gv[k];
                for (int i = wave; i < nres + nsk; i += COOP_NW) {
                    if (i != wave) load_tile(wrN, p.wpk, i < nres ? f.w_res[l] + (c0 / rpt + i) * 256 : f.w_skip[l] + (s0 / rpt + (i - nres)) * 256, lane);
                    const float acc = tree_reduce(chunk16(wrN, xg), logR);
                    if (i < nres) {
                        const int row = (c0 / rpt + i) * rpt + grp;
                        if (q == 0) {
                            const float v = (acc + sm[o_bres + l * CB + (row - c0)]) + sm[xin + row];
                            const RingDesc rn = p.rings[l + 1];
                            gr_store(X + c.o_ring[l + 1] + (size_t)((unsigned)t % (unsigned)rn.len) * C + row, tag, v);
                        }
                    } else {
                        const int row = (s0 / rpt + (i - nres)) * rpt + grp;
                        if (q == 0 && row >= s0 && row < s0 + SB) {
                            const int a = o_acc + (f.adaptive[l] ? SB : 0) + (row - s0);
                            sm[a] = sm[a] + (acc + sm[o_bsk + l * SB + (row - s0)]);
                        }
                    }
                }
                if (has_res) {
                    if (wave < npair) { load_tile(wcN, p.wpk, f.w_cur[l + 1] + (zt0 + wave) * 256, lane); load_tile(wqN, p.wpk, c.w_past_il[l + 1] + (zt0 + wave) * 256, lane); }
                    const RingDesc rn = p.rings[l + 1];
                    gather_vec(X + c.o_ring[l + 1] + (size_t)((unsigned)t % (unsigned)rn.len) * C, C, tag, sm + xnext, tid, c.abort, p.status);
                }
            }
            __syncthreads();
            const int tmp = xin; xin = xnext; xnext = tmp;
        }
        // ---------------- tail: relu(skip total) -> post 1x1 #1 -> relu -> post 1x1 #2 (qpnet.py:566-571)
        for (int r = tid; r < SB; r += COOP_NT) {
            const float tot = sm[o_acc + r] + sm[o_acc + SB + r];     // sum(skip_F) + sum(skip_A)  (qpnet.py:505)
            gr_store(X + c.o_y1 + s0 + r, tag, tot > 0.0f ? tot : 0.0f);
        }
        const int np1 = (SB + rpts - 1) / rpts, np2 = (QB + rpts - 1) / rpts;
        if (wave < np1) load_tile(wrN, p.wpk, f.w_p1 + (s0 / rpts + wave) * 256, lane);
        gather_vec(X + c.o_y1, S, tag, sm + o_y1, tid, c.abort, p.status);
        __syncthreads();
        {
            float4 xq[4];
            const float4* yv = (const float4*)(sm + o_y1 + 16 * qs);
#pragma unroll
            for (int k = 0; k < 4; ++k) xq[k] = yv[k];
            for (int i = wave; i < np1; i += COOP_NW) {
                const int ti = s0 / rpts + i;
                if (i != wave) load_tile(wrN, p.wpk, f.w_p1 + ti * 256, lane);
                const float acc = tree_reduce(chunk16(wrN, xq), logRs);
                const int row = ti * rpts + grps;
                if (qs == 0 && row >= s0 && row < s0 + SB) { const float v = acc + sm[o_bp1 + (row - s0)]; gr_store(X + c.o_y2 + row, tag, v > 0.0f ? v : 0.0f); }
            }
        }
        if (wave < np2) load_tile(wrN, p.wpk, f.w_p2 + (q0 / rpts + wave) * 256, lane);
        gather_vec(X + c.o_y2, S, tag, sm + o_y2, tid, c.abort, p.status);
        __syncthreads();
        {
            float4 xq[4];
            const float4* yv = (const float4*)(sm + o_y2 + 16 * qs);
#pragma unroll
            for (int k = 0; k < 4; ++k) xq[k] = yv[k];
            for (int i = wave; i < np2; i += COOP_NW) {
                const int ti = q0 / rpts + i;
                if (i != wave) load_tile(wrN, p.wpk, f.w_p2 + ti * 256, lane);
                const float acc = tree_reduce(chunk16(wrN, xq), logRs);
                const int row = ti * rpts + grps;
                if (qs == 0 && row >= q0 && row < q0 + QB) gr_store(X + c.o_lg + row, tag, acc + sm[o_bp2 + (row - q0)]);
            }
        }
        gather_vec(X + c.o_lg, Q, tag, sm + o_lg, tid, c.abort, p.status);
        __syncthreads();
        // ---------------- pick: every workgroup holds all logits and derives the same next sample
        if (wave == 0) {
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int i = lane; i < Q; i += 64) { const float v = sm[o_lg + i]; if (v > bv) { bv = v; bi = i; } }
            bi = wave_argmax(bv, bi);
            const int i = t - (u.n0 - 1);
            if (i >= 0 && u.logits && gidx == 0) for (int k = lane; k < Q; k += 64) u.logits[(size_t)i * Q + k] = sm[o_lg + k];
            int next;
            if (i >= 0) {
                if (p.mode == QPN_MODE_SAMPLING) bi = sample_wave(o_lg, Q, p.seed, (unsigned)u.row, (unsigned)i, lane);
                next = bi;
                if (u.teacher) { const int64_t sv = u.teacher[i] % Q; next = (int)(sv < 0 ? sv + Q : sv); }
                if (lane == 0 && gidx == 0) u.out[i] = bi;
            } else next = u.known[t + 1];
            if (lane == 0) {
                smi[o_samp] = smi[o_samp + 1]; smi[o_samp + 1] = next;
                smi[o_samp + 2] = __hip_atomic_load(c.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        if (smi[o_samp + 2]) break;            // a peer gave up: leave together (the flag was read once, by one lane)
    }
}

// ------------------------------------------------------------------------------------------ host side
static int ilog2c(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// largest power-of-two group size G <= limit: the gate and residual row slices are whole 4 KiB tiles; a skip / post-net slice may be a fraction of ONE tile
// (round 5: G = 128 / 256 for the C = 512 geometry -- the workgroup takes the tile that holds its one or two rows and keeps their lane groups)
int qpn_coop_group_size(const Geom& g, int limit) {
    const int rpt = 64 / (g.Cp / 16), rpts = 64 / (g.Sp / 16);
    auto slice_ok = [](int rows, int per_tile) { return rows % per_tile == 0 || (rows < per_tile && per_tile % rows == 0); };
    int best = 1;
    for (int G = 1; G <= limit; G *= 2) {
        if (g.C % G || g.S % G || g.Q % G) break;
        const int CB = g.C / G, SB = g.S / G, QB = g.Q / G;
        if ((2 * CB) % rpt || CB % rpt || !slice_ok(SB, rpt) || !slice_ok(SB, rpts) || !slice_ok(QB, rpts)) break;
        best = G;
    }
    return best;
}

int qpn_launch_decode_coop(qpn_handle* h, DecodeParams& p, int B, int G, hipStream_t stream) {
    const Geom& g = h->g;
    const int L = g.L, C = g.C, S = g.S, Q = g.Q;
    CoopParams c; memset(&c, 0, sizeof(c));
    c.G = G; c.CB = C / G; c.SB = S / G; c.QB = Q / G; c.Sp = g.Sp;
    c.logR = ilog2c(g.Cp / 16); c.rpt = 64 / (g.Cp / 16); c.logRs = ilog2c(g.Sp / 16); c.rpts = 64 / (g.Sp / 16);
    long o = 0;
    for (int l = 0; l < L; ++l) { c.o_ring[l] = (int)o; o += (long)p.rings[l].len * C; c.w_past_il[l] = h->w_past_il[l];
                                  c.f_resb[l] = (int)g.layers[l].resb; c.f_skipb[l] = (int)g.layers[l].skipb; }
    c.o_g = (int)o; o += (long)L * C; c.o_y1 = (int)o; o += S; c.o_y2 = (int)o; o += S; c.o_lg = (int)o; o += Q;
    o = (o + 15) & ~15L;
    if (o >= (1L << 31)) { qpn_set_error("cooperative decode: exchange block too large"); return QPN_EINVAL; }
    c.utt_stride = o; c.f_p1b = (int)g.post1_b; c.f_p2b = (int)g.post2_b;
    const int per_launch = h->n_cus / G > 0 ? h->n_cus / G : 1;        // all workgroups of a launch must be resident together (one per CU)
    const int nb = B < per_launch ? B : per_launch;
    const size_t xwords = (size_t)o * nb + 16;
    if (xwords > h->xch_cap) {
        if (h->d_xch) (void)hipFree(h->d_xch);
        h->d_xch = nullptr; h->xch_cap = 0;
        if (hipMalloc(&h->d_xch, xwords * sizeof(unsigned long long)) != hipSuccess) { qpn_set_error("hipMalloc(%zu MiB) for the decode exchange buffers failed", xwords * 8 >> 20); return QPN_ENOMEM; }
        h->xch_cap = xwords;
    }
    c.xch = h->d_xch + 16; c.abort = (int*)h->d_xch;          // first 128 bytes: the abort flag
    // LDS of the kernel
    int lds = 2 * g.Cp + L * g.Cp + g.Cp + 2 * g.Sp + ((Q + 3) & ~3) + ((2 * c.SB + 3) & ~3) + L * 2 * c.CB + 4 + L * (c.CB + c.SB) + c.SB + c.QB;
    const size_t lds_bytes = (size_t)lds * sizeof(float);
    if (lds_bytes > 160 * 1024) { qpn_set_error("cooperative decode: %zu KiB of step state per workgroup exceed LDS (use more workgroups per utterance)", lds_bytes >> 10); return QPN_EINVAL; }
    QPN_HIP(hipFuncSetAttribute((const void*)k_decode_coop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    for (int b0 = 0; b0 < B; b0 += per_launch) {
        const int n = B - b0 < per_launch ? B - b0 : per_launch;
        QPN_HIP(hipMemsetAsync(h->d_xch, 0, xwords * sizeof(unsigned long long), stream));      // tags, rings, abort flag
        hipLaunchKernelGGL(k_decode_coop, dim3(G, n), dim3(COOP_NT), lds_bytes, stream, p, h->fp, c, b0);
    }
    QPN_HIP(hipGetLastError());
    return QPN_OK;
}
