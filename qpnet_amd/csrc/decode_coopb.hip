// decode_coopb.hip -- cooperative autoregressive decode of the wide geometries with the UTTERANCES BATCHED INTO THE CONTRACTIONS (round 6; gfx950 / MI355X).
//
// decode_coop.hip gives every utterance its own group of workgroups: at the reference's decode batch (20 utterances, src/runQP.py:66) twenty groups each
// stream the repo-default model's 96.6 MB of weights per generated sample (9.8 TB/s aggregate -- the Infinity Cache's rate), and a row's 16-term FMA chain +
// cross-lane tree is a ~350-cycle dependent chain per (row tile, utterance).  Here ONE group of G = n_resch / 8 workgroups serves up to 16 utterances: the
// utterances are the N dimension of v_mfma_f32_16x16x4_f32, a workgroup's 16 rows of a matrix the M dimension, and the weights are read ONCE per sample step
// for all of them.  Reference: QPNet.batch_fast_generate, src/nets/qpnet.py:314-559 (the per-sample loop :446-557).
//
// Arithmetic: the fixed-order "QPNet-f32" spec, bit for bit.  Four chained MFMAs (accumulator 0, k ascending) ARE the spec's 16-term
// chain p = w0 x0, fma, fma, ... (tools/mfma_chain_test.hip: 5.1 M outputs, no difference), so a chunk's accumulator tile holds chunk16 of 16 rows x 16
// utterances; the spec's stride-halving tree over the K / 16 chunks is elementwise adds of such tiles in the same association: the four waves of a dot
// product take the chunks c = j (mod 4) in the order j, j+16, j+8, j+24, j+4, j+20, j+12, j+28 (three live tiles), and (W0 + W2) + (W1 + W3) closes it.
//
// Partition and exchange as in decode_coop.hip: workgroup w owns 8 channels (gate rows sigma / tanh of them: one 16-row tile; their residual rows + its
// S / G skip rows: one tile), every dependent stage ends in an all-gather of {tag, value} granules (2 per layer + 3), the layer-input history IS the
// exchange buffer.  Vectors live in LDS as B-operand images: float4 ((chunk * 4 + k-slot) * 16 + utterance) = x[utterance][16 chunk + 4 e + k-slot], e = 0..3.
#include "decode_dev.h"
#include "qpn_handle.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned long long u64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CBB_NT 512
#define CBB_NW (CBB_NT / 64)
#define CBB_NU 16                     // utterance columns of a tile
#define CBB_SPIN_LIMIT (1u << 20)

struct CoopbParams {
    u64* xch; long utt_stride;
    int o_ring[QPN_MAX_LAYERS]; int o_g, o_y1, o_y2, o_lg;
    int* abort; unsigned wpk_bytes; long long base4; int per_w, dev_nostream, dev_nocheck, poll_delay_g, poll_delay_x, poll_delay_t;      // workgroup w's fragments: per_w float4 from base4 + w * per_w
    int G, SB, QB, NBper, B;          // workgroups of a group (8 channels each); skip / logit rows per workgroup; utterances per group; utterances of the call
    int RC, RS;                       // 16-deep chunks of K = n_resch / K = n_skipch
    int zc[QPN_MAX_LAYERS], zp[QPN_MAX_LAYERS], rs[QPN_MAX_LAYERS], p1, p2;      // float4 offsets of the A-operand tiles inside the workgroup's fragment block: word chunk * 64 + lane
    int f_resb[QPN_MAX_LAYERS], f_skipb[QPN_MAX_LAYERS], f_p1b, f_p2b, adaptive[QPN_MAX_LAYERS];
};


typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CBB_SC1 16                    // cache-policy bit of the buffer accesses: sc1 (agent scope -- what the atomic accesses of the granules compile to)
#define CBB_XT 256                    // threads of the exchange waves

#define CBB_LT 10                     // ints per layer of the LDS layer table: granule offset of the ring, its length, adaptive?, fragment blocks zc / zp / rs, the step's slot (either parity), dilation
// (CBB_FRESH: a thread index made opaque, so that the compiler recomputes what derives from it at every use instead of keeping dozens of
//  loop-invariant offsets alive across the whole step -- they spilled)
#define CBB_FRESH(x) asm volatile("" : "+v"(x))

__host__ __device__ static inline int cbb_lds_floats(int C, int L) {
    return 4 * C * CBB_NU + 4 * 256 + 2 * 4 * 256 + 2 * 256 + 2 * 8 * CBB_NU + 2 * L * 8 + 8 + 16 + 2 * L * CBB_NU + L * CBB_LT + 2 * CBB_NU + 4 * CBB_NU + 2 * CBB_NU + 2 * CBB_NU + 4 + 32 + 1 + CBB_NU * 22;
}
// float offset, inside a B-operand image, of element (channel ch, utterance n)
__device__ __forceinline__ int img_idx(int ch, int n) { return (((((ch >> 4) << 2) + (ch & 3)) * CBB_NU + n) << 2) + ((ch >> 2) & 3); }
// two granules (adjacent channels, one tag) with one 16-byte store; each half validates itself, so the halves may become visible apart
__device__ __forceinline__ void gb_store2(__amdgpu_buffer_rsrc_t rs, int goff, unsigned tag, float v0, float v1) {
    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(v0), tag, __float_as_uint(v1), tag}, rs, goff * 8, 0, CBB_SC1);
}
__device__ __forceinline__ void gb_store1(__amdgpu_buffer_rsrc_t rs, int goff, unsigned tag, float v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(v), tag}, rs, goff * 8, 0, CBB_SC1);
}

// ---- all-gather of one vector of every utterance of the group, by the 256 threads of the exchange waves.  An item is a PAIR of granules (channels 2 pr,
// 2 pr + 1 of utterance n: one 16-byte load; each half carries its own tag, so a torn pair is simply read again); a thread owns the items (n0 + k * per, pr),
// k = 0 .. NK - 1 (NK compiled for the group's size), and has them all in flight.  tags[n] == 0: the utterance has nothing here (a tap before time 1, or finished) -- zeros, no load.
// IMG: into a B-operand image, else plain [n][count].
struct GSrc { int src0, stride; const int* soff; const unsigned* tags; int hshift; int nocheck; int nlo; };      // granule offset of utterance 0's vector, utterance stride, per-utterance slots, tags, log2(pairs), (dev) no tag check, first utterance of this call's share
template <int NK> struct GBatch { u32x4 v[NK]; unsigned want[NK]; int voff[NK]; };
template <int NK>
__device__ __forceinline__ void g_issue(GBatch<NK>& b, __amdgpu_buffer_rsrc_t rs, const GSrc& s, int nb, int xt, int kb) {
    CBB_FRESH(xt);
    const int per = CBB_XT >> s.hshift, n0 = xt >> s.hshift, pr = xt & ((1 << s.hshift) - 1);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int n = s.nlo + n0 + (kb + k) * per;
        const bool in = n < nb;
        b.want[k] = in ? s.tags[n] : 0u;
        b.voff[k] = (s.src0 + (in ? n * s.stride + (s.soff ? s.soff[n] : 0) : 0) + 2 * pr) * 8;
        b.v[k] = (u32x4){0u, 0u, 0u, 0u};
        if (b.want[k]) b.v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, b.voff[k], 0, CBB_SC1);
    }
}
template <int NK, bool IMG>
__device__ __forceinline__ void g_finish(GBatch<NK>& b, __amdgpu_buffer_rsrc_t rs, const GSrc& s, float* dstbase, int count, int nb, int xt, int kb, int* abort, int* status) {
    unsigned spins = 0;
    if (!s.nocheck)
    for (;;) {
        unsigned bad = 0u;      // (one test for the round that succeeds: no branch per item)
#pragma unroll
        for (int k = 0; k < NK; ++k) bad |= b.want[k] ? ((b.v[k].y ^ b.want[k]) | (b.v[k].w ^ b.want[k])) : 0u;
        if (!bad) break;
#pragma unroll
        for (int k = 0; k < NK; ++k)
            if (b.want[k] && (b.v[k].y != b.want[k] || b.v[k].w != b.want[k])) b.v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, b.voff[k], 0, CBB_SC1);
        if (++spins > CBB_SPIN_LIMIT || ((spins & 255u) == 0 && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(status, 4);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    CBB_FRESH(xt);
    const int per = CBB_XT >> s.hshift, n0 = xt >> s.hshift, pr = xt & ((1 << s.hshift) - 1);
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int n = s.nlo + n0 + (kb + k) * per;
        if (n < nb) {      // (an utterance without a tag loaded nothing: its registers are the zeros of g_issue)
            const float a = __uint_as_float(b.v[k].x), c = __uint_as_float(b.v[k].z);
            if (IMG) { const int d = img_idx(2 * pr, n); dstbase[d] = a; dstbase[d + 4 * CBB_NU] = c; }      // channel 2 pr + 1: the next k-slot
            else *(float2*)(dstbase + n * count + 2 * pr) = make_float2(a, c);
        }
    }
}
template <int NK, bool IMG>
__device__ __forceinline__ void gather1_n(__amdgpu_buffer_rsrc_t rs, const GSrc& sa, float* da, int count, int nb, int xt, int kb, int* abort, int* status) {
    GBatch<NK> a;
    g_issue(a, rs, sa, nb, xt, kb);
    g_finish<NK, IMG>(a, rs, sa, da, count, nb, xt, kb, abort, status);
}
template <int NK>
__device__ __forceinline__ void gather2_n(__amdgpu_buffer_rsrc_t rs, const GSrc& sa, float* da, const GSrc& sb, float* db, bool two, int count, int nb, int xt, int kb, int* abort, int* status) {
    GBatch<NK> a, b;
    g_issue(a, rs, sa, nb, xt, kb);
    if (two) g_issue(b, rs, sb, nb, xt, kb);
    g_finish<NK, true>(a, rs, sa, da, count, nb, xt, kb, abort, status);
    if (two) g_finish<NK, true>(b, rs, sb, db, count, nb, xt, kb, abort, status);
}
// (the poll loop's instruction count is on the critical path of every exchange: it is compiled for exactly 1 .. 8 items per thread, and for 10 and 12; 13 .. 16 take two rounds)
#define CBB_BY_ITEMS(n_, kb_, CALL) do { switch (n_) { case 1: CALL(1, kb_); break; case 2: CALL(2, kb_); break; case 3: CALL(3, kb_); break; case 4: CALL(4, kb_); break; \
    case 5: CALL(5, kb_); break; case 6: CALL(6, kb_); break; case 7: CALL(7, kb_); break; default: CALL(8, kb_); break; } } while (0)
template <bool IMG>
__device__ __forceinline__ void gather1(__amdgpu_buffer_rsrc_t rs, const GSrc& sa, float* da, int count, int nb, int xt, int* abort, int* status) {
    const int per = CBB_XT >> sa.hshift, iters = (nb - sa.nlo + per - 1) / per;
#define CBB_G1(NK, KB) gather1_n<NK, IMG>(rs, sa, da, count, nb, xt, KB, abort, status)
    if (iters <= 8) { CBB_BY_ITEMS(iters, 0, CBB_G1); }
    else if (iters <= 10) CBB_G1(10, 0);      // (groups of 9..16 utterances: still one round -- a second one costs a round trip per edge)
    else if (iters <= 12) CBB_G1(12, 0);
    else { CBB_G1(8, 0); const int m = iters - 8; CBB_BY_ITEMS(m, 8, CBB_G1); }      // (16 in flight spill)
#undef CBB_G1
}
// two vectors of the same length at once (the second only if `two`)
__device__ __forceinline__ void gather2(__amdgpu_buffer_rsrc_t rs, const GSrc& sa, float* da, const GSrc& sb, float* db, bool two, int count, int nb, int xt, int* abort, int* status) {
    const int per = CBB_XT >> sa.hshift, iters = (nb - sa.nlo + per - 1) / per;
#define CBB_G2(NK, KB) gather2_n<NK>(rs, sa, da, sb, db, two, count, nb, xt, KB, abort, status)
    for (int kb = 0; kb < iters; kb += 8) { const int m = iters - kb; CBB_BY_ITEMS(m, kb, CBB_G2); }
#undef CBB_G2
}

// ---- a compute wave's share of a dot product over R chunks: the chunks c = j + 4 i (i = 0 .. R / 4 - 1), their A fragments in registers, each a chain of
// four MFMAs from accumulator 0, combined in the spec's tree order
// (buffer loads: the block's offset is a scalar, the lane's a 32-bit register -- no 64-bit per-lane addresses to keep)
__device__ __forceinline__ void load_frags(float4 (&an)[8], __amdgpu_buffer_rsrc_t rw, unsigned blk4, int R, int j, int lane) {
    CBB_FRESH(lane);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (4 * i < R) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, (int)((blk4 + (unsigned)(j + 4 * i) * 64u) * 16u), 0);
            an[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        }
}
// (written step-major: the k-th MFMA of all the chunks, then the next -- eight independent accumulator chains in flight instead of one chain after another)
template <int NCH>
__device__ __forceinline__ void chunk_tiles(const float4 (&an)[8], const float* img, int j, int lane, f32x4 (&acc)[8]) {
    float4 b[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) b[i] = *(const float4*)(img + ((((j + 4 * i) << 2) + (lane >> 4)) * CBB_NU + (lane & 15)) * 4);
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[i].x, b[i].x, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[i].y, b[i].y, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[i].z, b[i].z, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NCH; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(an[i].w, b[i].w, acc[i], 0, 0, 0);
}
__device__ __forceinline__ f32x4 wave_dot(const float4 (&an)[8], const float* img, int R, int j, int lane) {
    CBB_FRESH(lane);
    f32x4 t[8];
    if (R == 32) {      // fragment i = chunk j + 4 i; the tree pairs chunks 16 apart, then 8, then 4: (t0 + t4) + (t2 + t6), (t1 + t5) + (t3 + t7)
        chunk_tiles<8>(an, img, j, lane, t);
        return ((t[0] + t[4]) + (t[2] + t[6])) + ((t[1] + t[5]) + (t[3] + t[7]));
    }
    // R == 16: chunks j, j+4, j+8, j+12: (t0 + t2) + (t1 + t3)
    chunk_tiles<4>(an, img, j, lane, t);
    return (t[0] + t[2]) + (t[1] + t[3]);
}
// element (row r, utterance n) of a dot product from the four waves' partial tiles (the tree's last two levels)
__device__ __forceinline__ float close_elem(const float* P, int r, int n) {
    const int o = (((r >> 2) * 16 + n) << 2) + (r & 3);
    return (P[o] + P[512 + o]) + (P[256 + o] + P[768 + o]);
}
// Waves 0..3 COMPUTE: every dot product, A fragments prefetched a layer ahead into three register sets (current tap / past tap / residual + skip) -- these
// waves issue no other global access, so the in-order return of their loads never holds up anything but the weight stream itself.
// Waves 4..7 EXCHANGE: tags and tap distances, layer 0's input, the aux terms, every publish and every gather.
__global__ __launch_bounds__(CBB_NT) void k_decode_coopb(const DecodeParams p, const CoopbParams c) {
    float* sm = SM; int* smi = SMI;
    const int w = blockIdx.x, grp = blockIdx.y;
    const int wblk = (int)(c.base4 + (long long)w * c.per_w);    // float4 offset of my fragments (relative to the buffer descriptor's base)
    const int b0 = grp * c.NBper;
    const int nb = c.B - b0 < c.NBper ? c.B - b0 : c.NBper;
    const int ustride = (int)c.utt_stride;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(c.xch + (size_t)b0 * c.utt_stride), 0, (int)((size_t)nb * c.utt_stride * 8), 0x00020000);
#ifdef QPN_ENABLE_STAMPS
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)c.wpk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, 0, 0x00020000);      // (zero records: every load is out of range -- returns 0 without touching memory)
    const __amdgpu_buffer_rsrc_t rw = rw_;
#else
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, 0, (int)c.wpk_bytes, 0x00020000);
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = p.C, S = p.S, Q = p.Q, L = p.L;
    const int SB = c.SB, QB = c.QB, RC = c.RC, RS = c.RS;
    const int hC = 30 - __clz(C), hS = 30 - __clz(S), hQ = 30 - __clz(Q);      // log2 of the PAIRS of a vector (C, S, Q: powers of two)
    const int c0 = w * 8, s0 = w * SB, q0 = w * QB;
    // LDS layout (floats)
    const int IMG = C * CBB_NU;
    const int o_x = 0, o_gv = IMG, o_xp = 2 * IMG;            // layer input; gate vector; past rows of the layers of either parity
    int o = 4 * IMG;
    const int o_part = o; o += 4 * 256;                       // partial tiles of the four waves of a dot product
    const int o_pp = o; o += 2 * 4 * 256;                     // ... of the past-tap dot products, layers of either parity
    const int o_auxz = o; o += 2 * 256;                       // aux terms of my gate rows, layers of either parity: [16 rows][16 utterances]
    const int o_acc = o; o += 2 * 8 * CBB_NU;                 // skip totals of the fixed / adaptive stacks: [2][8 rows][16 utterances]
    const int o_bres = o; o += L * 8; const int o_bsk = o; o += L * 8; const int o_bp1 = o; o += 8; const int o_bp2 = o; o += 16;
    const int o_off = o; o += 2 * L * CBB_NU;                 // (int) tap distances of the step (steps of either parity: the next step's are computed during this step's post-net)
    const int o_lt = o; o += L * CBB_LT;                      // (int) layer table
    const int o_samp = o; o += 2 * CBB_NU;                    // (int) the two newest samples of every utterance
    const int o_tag = o; o += 4 * CBB_NU;                     // (unsigned) tags: [0], [1] the step's (either parity; 0: inactive), [2], [3] of the past rows being gathered
    const int o_so = o; o += 2 * CBB_NU;                      // (int) ring slots (granule offsets) of the past rows being gathered
    const int o_fr = o; o += 2 * CBB_NU;                      // (int, either parity) the step's aux frame of every utterance: its layer-0 row of pproj (rows of 2 C floats)
    const int o_ctl = o; o += 4;
    const int o_prof = o; o += 32;                            // (dev: phase times, -DQPN_ENABLE_STAMPS builds)
    const int o_ud = (o + 1) & ~1; o = o_ud + CBB_NU * 22;                     // the group's utterance descriptors (UttDesc, 88 bytes each): read every step, a dependent global load otherwise
    const int o_lg = o_x;                                     // logits [nb][Q] (the images are free by then)
    for (int i = tid; i < o; i += CBB_NT) sm[i] = 0.0f;
    __syncthreads();
    for (int i = tid; i < L * 8; i += CBB_NT) sm[o_bres + i] = p.flat[c.f_resb[i >> 3] + c0 + (i & 7)];
    for (int i = tid; i < L * 8; i += CBB_NT) if ((i & 7) < SB) sm[o_bsk + i] = p.flat[c.f_skipb[i >> 3] + s0 + (i & 7)];
    if (tid < SB) sm[o_bp1 + tid] = p.flat[c.f_p1b + s0 + tid];
    if (tid < QB) sm[o_bp2 + tid] = p.flat[c.f_p2b + q0 + tid];
    if (tid < L) {
        int* e = smi + o_lt + tid * CBB_LT;
        e[0] = c.o_ring[tid]; e[1] = p.rings[tid].len; e[2] = c.adaptive[tid]; e[3] = wblk + c.zc[tid]; e[4] = wblk + c.zp[tid]; e[5] = wblk + c.rs[tid]; e[8] = p.rings[tid].mult;
    }
    static_assert(sizeof(UttDesc) == 88, "the LDS copy of the descriptors assumes 22 ints");
    for (int i = tid; i < nb * 22; i += CBB_NT) smi[o_ud + i] = ((const int*)(p.utts + b0))[i];
    const UttDesc* const uds = (const UttDesc*)(smi + o_ud);
    int Tmax = 0;
    for (int k = 0; k < nb; ++k) { const UttDesc ud = p.utts[b0 + k]; const int tt = ud.n0 + ud.n_samples; Tmax = tt > Tmax ? tt : Tmax; }
    if (tid < nb) { const UttView u = make_view(p, p.utts[b0 + tid]); smi[o_samp + 2 * tid] = u.known[0]; smi[o_samp + 2 * tid + 1] = u.n0 + u.n_samples >= 3 ? u.known[1] : 0; }
    __syncthreads();
    if (Tmax < 3) return;
    unsigned* const tagbuf = (unsigned*)(smi + o_tag);
    unsigned* const ptags = tagbuf + 2 * CBB_NU;             // the two rows of past-row tags
    const int j4 = wave & 3;
    const bool cw = wave < 4;
    const int xt = tid - CBB_XT;                              // index among the exchange threads (negative: a compute thread)
    const int am = (tid & 255) >> 4, ak = tid & 15;           // an exchange thread's aux term: gate row am, utterance ak
    const int anat = (am >> 3) * C + c0 + (am & 7);
    const unsigned p1blk = (unsigned)(wblk + c.p1), p2blk = (unsigned)(wblk + c.p2);
    auto LT = [&](int l, int f) { return __builtin_amdgcn_readfirstlane(smi[o_lt + l * CBB_LT + f]); };
#ifdef QPN_ENABLE_STAMPS      // dev (QPN_COOPB_NOCHECK=1): gathers take whatever they read -- wrong samples, the time of an exchange without the wait for its producers
    const int NOCHK = c.dev_nocheck;
#else
    const int NOCHK = 0;
#endif
#ifdef QPN_ENABLE_STAMPS      // dev aid (QPN_STAMPS=1): time per phase as the first exchange wave of workgroup (0, 0) sees it, summed over the steps -> p.stamps[phase * QPN_NW]
    const bool prof = p.stamps && w == 0 && grp == 0 && xt == 0;
    long long prof_last = prof ? (long long)__builtin_readcyclecounter() : 0;
#define CB_T(k) do { if (prof) { const long long now_ = (long long)__builtin_readcyclecounter(); ((long long*)(sm + o_prof))[k] += now_ - prof_last; prof_last = now_; } } while (0)
    const bool profc = p.stamps && w == 0 && grp == 0 && tid == 0;      // ... and the first compute wave: slots 11..15
    long long profc_last = 0;
#define CB_C0() do { if (profc) profc_last = (long long)__builtin_readcyclecounter(); } while (0)
#define CB_C(k) do { if (profc) { const long long now_ = (long long)__builtin_readcyclecounter(); ((long long*)(sm + o_prof))[k] += now_ - profc_last; profc_last = now_; } } while (0)
#else
#define CB_T(k) do { } while (0)
#define CB_C0() do { } while (0)
#define CB_C(k) do { } while (0)
#endif
    // the state of step ts that does not depend on its samples -- per utterance: active?, aux frame, tap distances; per layer: the ring slot -- into the buffers of
    // parity ts & 1; computed by the compute waves while the exchange waves gather the last layer's gate vector (nothing else for them to do there) (qpnet.py:450-452, 592-624)
    auto step_state = [&](int ts) {
        const int pr_ = ts & 1;
        const int k = tid & 15;      // (the 256 threads of the compute waves: utterance k, layers tid >> 4, + 16, ...)
        {
            unsigned tg = 0u; int fr = 0;
            if (k < nb) {
                const UttDesc ud = uds[k];
                const UttView u = make_view(p, ud);
                if (ts + 1 < u.n0 + u.n_samples) {
                    tg = (unsigned)ts + 1u;
                    const int ut = aux_time(u, ts);
                    int f = 0;
                    if (ut >= 0) f = p.U > 0 ? (int)((unsigned)ut / (unsigned)p.U) : ut;
                    fr = (int)(ud.pproj / (2 * C)) + f * L;
                    const int widx = ts < u.n0 - 1 ? ts - (u.n0 - 1) : 0;
                    for (int l = tid >> 4; l < L; l += 16) {
                        const int* e = smi + o_lt + l * CBB_LT;
                        RingDesc r; r.base = 0; r.len = e[1]; r.mult = e[8]; r.adaptive = e[2];
                        int off = tap_offset(r, u, ut, widx);
                        if (off < 1 || off >= r.len) { atomicOr(p.status, 1); off = off < 1 ? 1 : r.len - 1; }
                        smi[o_off + pr_ * L * CBB_NU + l * CBB_NU + k] = off;
                    }
                }
            }
            if (tid < CBB_NU) { tagbuf[pr_ * CBB_NU + k] = tg; smi[o_fr + pr_ * CBB_NU + k] = fr; }
        }
        if (wave == 3 && lane < L) { int* e = smi + o_lt + lane * CBB_LT; e[6 + pr_] = e[0] + (int)((unsigned)ts % (unsigned)e[1]) * C; }
        if (tid == 255) {      // aux time's upsampling weight: the same for every utterance of the call (one t, one n_pad)
            float wv = 1.0f;
            if (p.U > 0) {
                const UttDesc& ud = uds[0];
                const int ut = ts - ud.n_pad - (ts < ud.n0 - 1 ? 1 : 0);
                wv = p.flat[p.up_w + (ut >= 0 ? ut - (int)((unsigned)ut / (unsigned)p.U) * p.U : 0)];
            }
            sm[o_ctl + 2 + pr_] = wv;
        }
    };
    // ---------------- pick (qpnet.py:505-516) from the logits in LDS: a wave per utterance, utterances kfirst, kfirst + 8, ...
    auto pick_utts = [&](int t, int kfirst) {
        const unsigned* tg = tagbuf + (t & 1) * CBB_NU;
        for (int k = kfirst; k < nb; k += 8) {
            if (!tg[k]) continue;
            const UttView u = make_view(p, uds[k]);
            float bv = -INFINITY; int bi = 0x7fffffff;
            for (int i = lane; i < Q; i += 64) { const float v = sm[o_lg + k * Q + i]; if (v > bv) { bv = v; bi = i; } }
            bi = wave_argmax(bv, bi);
            const int i = t - (u.n0 - 1);
            if (i >= 0 && u.logits && w == 0) for (int q = lane; q < Q; q += 64) u.logits[(size_t)i * Q + q] = sm[o_lg + k * Q + q];
            int next;
            if (i >= 0) {
                if (p.mode == QPN_MODE_SAMPLING) bi = sample_wave(o_lg + k * Q, Q, p.seed, (unsigned)u.row, (unsigned)i, lane);
                next = bi;
                if (u.teacher) { const int64_t sv = u.teacher[i] % Q; next = (int)(sv < 0 ? sv + Q : sv); }
                if (lane == 0 && w == 0) u.out[i] = bi;
            } else next = u.known[t + 1];
            if (lane == 0) { smi[o_samp + 2 * k] = smi[o_samp + 2 * k + 1]; smi[o_samp + 2 * k + 1] = next; }
        }
    };
    // The two roles run their OWN loop nests (the register sets of one never meet the other's in a merge); both execute the same sequence of barriers:
    //   per step: P1 | P2 | P3 | per layer: A | B | C | D | tail: T1 .. T6
    if (cw) {
        // =========================================================================== compute waves
        // fragments of the next current-tap / past-tap / residual + skip dot products.  Every set has ONE load site per phase (block and chunk count selected
        // as scalars): two sites merging into one register set made the compiler wait for the loads on the spot and copy them.
        float4 an_c[8], an_p[8], an_r[8];
        load_frags(an_c, rw, (unsigned)LT(0, 3), RC, j4, lane);
        load_frags(an_p, rw, (unsigned)LT(0, 4), RC, j4, lane);
        load_frags(an_r, rw, (unsigned)LT(0, 5), RC, j4, lane);
        step_state(1);
        __syncthreads();          // (the first step's state)
        for (int t = 1; t + 1 < Tmax; ++t) {
#ifdef QPN_ENABLE_STAMPS      // dev (QPN_COOPB_NOSTREAM=1): the fragments are loaded in the first steps and never again -- wrong samples, the time without the weight stream
            const __amdgpu_buffer_rsrc_t rw = (c.dev_nostream && t > 3) ? rw0 : rw_;
#endif
            __syncthreads();      // P2
            {      // (idle until P3: the past rows of layers 0 and 1, beside the exchange waves' gather of layer 0's input)
                const GSrc sa = {LT(0, 0), ustride, smi + o_so, ptags, hC, NOCHK, 0}, sb = {LT(1, 0), ustride, smi + o_so + CBB_NU, ptags + CBB_NU, hC, NOCHK, 0};
                gather2(rs, sa, sm + o_xp, sb, sm + o_xp + IMG, true, C, nb, tid, c.abort, p.status);
            }
            __syncthreads();      // P3: layer 0's input and the past rows of layers 0, 1 are in LDS
            {      // layer 0's past-row dot (the other layers': a layer ahead, between A and B)
                const f32x4 part = wave_dot(an_p, sm + o_xp, RC, j4, lane);
                load_frags(an_p, rw, (unsigned)LT(1, 4), RC, j4, lane);
                *(f32x4*)(sm + o_pp + j4 * 256 + lane * 4) = part;
            }
            for (int l = 0; l < L; ++l) {
                const bool has_res = l + 1 < L;
                {      // this layer's current-row dot
                    const f32x4 part = wave_dot(an_c, sm + o_x, RC, j4, lane);
                    *(f32x4*)(sm + o_part + j4 * 256 + lane * 4) = part;
                }
                __syncthreads();      // A
                // (the next fragments are requested AFTER the barrier the dot product's consumers wait at: with the exchange waves' gathers in the CU's memory
                //  pipeline, issuing eight loads took 0.4 us -- measured -- and the compute waves have slack in the phases that follow)
                load_frags(an_c, rw, has_res ? (unsigned)LT(l + 1, 3) : p1blk, has_res ? RC : RS, j4, lane);
                if (has_res) {        // the next layer's past-row dot (its rows arrived during the previous layer)
                    const f32x4 part = wave_dot(an_p, sm + o_xp + ((l + 1) & 1) * IMG, RC, j4, lane);
                    load_frags(an_p, rw, l + 2 < L ? (unsigned)LT(l + 2, 4) : p2blk, l + 2 < L ? RC : RS, j4, lane);
                    *(f32x4*)(sm + o_pp + ((l + 1) & 1) * 1024 + j4 * 256 + lane * 4) = part;
                } else step_state(t + 1);      // (the last layer has no successor whose past-tap dot would run here)
                __syncthreads();      // B: the gate vector is in LDS
                {      // residual 1x1 rows of my channels and my skip 1x1 rows, one tile
                    CB_C0();
                    const f32x4 part = wave_dot(an_r, sm + o_gv, RC, j4, lane);
                    CB_C(11);
                    *(f32x4*)(sm + o_part + j4 * 256 + lane * 4) = part;
                    CB_C(13);
                }
                __syncthreads();      // C
                CB_C(14);
                load_frags(an_r, rw, (unsigned)LT(has_res ? l + 1 : 0, 5), RC, j4, lane);
                CB_C(12);
                if (l + 2 < L) {      // (idle until D: the past rows of layer l + 2 -- rows of earlier steps, nothing to wait for -- so that the exchange waves gather one vector per edge)
                    const GSrc sb = {LT(l + 2, 0), ustride, smi + o_so, ptags, hC, NOCHK, 0};
                    gather1<true>(rs, sb, sm + o_xp + (l & 1) * IMG, C, nb, tid, c.abort, p.status);
                }
                __syncthreads();      // D: the next layer's input is in LDS
            }
            __syncthreads();          // T1: relu(skip total) of every utterance is in LDS
            {
                const f32x4 part = wave_dot(an_c, sm + o_xp, RS, j4, lane);
                *(f32x4*)(sm + o_part + j4 * 256 + lane * 4) = part;
            }
            __syncthreads();          // T2
            load_frags(an_c, rw, (unsigned)LT(0, 3), RC, j4, lane);
            __syncthreads();          // T3
            {
                const f32x4 part = wave_dot(an_p, sm + o_gv, RS, j4, lane);
                *(f32x4*)(sm + o_part + j4 * 256 + lane * 4) = part;
            }
            __syncthreads();          // T4
            load_frags(an_p, rw, (unsigned)LT(0, 4), RC, j4, lane);
            __syncthreads();          // T5: the logits of every utterance are in LDS
            pick_utts(t, wave + 4);
            __syncthreads();          // T6
            if (smi[o_ctl + 1]) break;            // a peer gave up: leave together
        }
    } else {
        // =========================================================================== exchange waves
        __syncthreads();          // (the first step's state: the compute waves)
        for (int t = 1; t + 1 < Tmax; ++t) {
            const unsigned tag = (unsigned)t + 1u;
            // (P1 -- per utterance: active?, aux frame, tap distances; per layer: the ring slot -- was computed during the previous step's post-net)
            const int par = t & 1;
            unsigned* const tags = tagbuf + par * CBB_NU;
            const int ofr = o_fr + par * CBB_NU, ooff = o_off + par * L * CBB_NU, sl = 6 + par;
            const float wj = sm[o_ctl + 2 + par];
            CB_T(0);
            // tags and ring slots of the past rows of layer l (rows of EARLIER steps: published long ago; time 0 and before: never written, zeros) -> set `which`
            auto past_rows = [&](int l, int which) {
                const int k = xt - 128;
                if (k >= 0 && k < CBB_NU) {
                    unsigned tg = 0u; int so = 0;
                    if (tags[k]) {
                        const int tp = t - smi[ooff + l * CBB_NU + k];
                        if (tp >= 1) { tg = (unsigned)tp + 1u; so = (int)((unsigned)tp % (unsigned)smi[o_lt + l * CBB_LT + 1]) * C; }
                    }
                    ptags[which * CBB_NU + k] = tg; smi[o_so + which * CBB_NU + k] = so;
                }
            };
            // ---------------- P2. my channels of layer 0's input (two rows of the causal table, qpnet.py:110-132) -> ring 0; layer 0's aux terms; the past rows of layers 0 and 1
            past_rows(0, 0);
            past_rows(1, 1);
            if (xt < 4 * CBB_NU) {            // (channel pair, utterance)
                const int m2 = xt >> 4, k = xt & 15;
                if (k < nb && tags[k]) {
                    const int s_prev = smi[o_samp + 2 * k], s_cur = smi[o_samp + 2 * k + 1];
                    float v[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int ch = c0 + 2 * m2 + i;
                        v[i] = p.flat[p.causal_w + ((size_t)ch * Q + s_prev) * 2] + p.flat[p.causal_w + ((size_t)ch * Q + s_cur) * 2 + 1];
                        v[i] = v[i] + p.flat[p.causal_b + ch];
                    }
                    gb_store2(rs, k * ustride + smi[o_lt + sl] + c0 + 2 * m2, tag, v[0], v[1]);
                }
            }
            {
                float v = 0.0f;
                if (ak < nb && tags[ak]) v = __builtin_fmaf(wj, p.pproj[(size_t)smi[ofr + ak] * (2 * C) + anat], p.qb[anat]);
                sm[o_auxz + xt] = v;
                sm[o_acc + xt] = 0.0f;
            }
            __syncthreads();      // P2
            CB_T(1);
            {
                const GSrc sx = {LT(0, sl), ustride, nullptr, tags, hC, NOCHK, 0};
                gather1<true>(rs, sx, sm + o_x, C, nb, xt, c.abort, p.status);
            }
            __syncthreads();      // P3
            CB_T(2);
            for (int l = 0; l < L; ++l) {
                const bool has_res = l + 1 < L;               // the last block's residual output is unused (qpnet.py:505)
                if (l + 2 < L) past_rows(l + 2, 0);
                // one (channel, utterance) per thread of the first two exchange waves.  What does not depend on this layer's current-row dot is read while it runs:
                // the past-tap dot product (closed from its partial tiles), the aux terms, the layer input of my channel (the gather after C overwrites it)
                int xv = xt; CBB_FRESH(xv);
                const int gm = xv >> 4, gn = xv & 15;
                const bool gate_thread = gm < 8 && gn < nb && tags[gn];
                const float* Pp = sm + o_pp + (l & 1) * 1024; const float* Ax = sm + o_auxz + (l & 1) * 256;
                float aps = 0.0f, apt = 0.0f, axs = 0.0f, axt = 0.0f, xres = 0.0f;
                if (gate_thread) {
                    axs = Ax[gm * 16 + gn]; axt = Ax[(gm + 8) * 16 + gn]; xres = sm[o_x + img_idx(c0 + gm, gn)];
                    if (l > 0) { aps = close_elem(Pp, gm, gn); apt = close_elem(Pp, gm + 8, gn); }      // (layer 0's: computed in this very phase)
                }
                __syncthreads();      // A: the partial tiles of this layer's current-row dot are in LDS
                CB_T(3);
                // ---------------- gate of my channels -> all-gather; beside it the aux terms of layer l + 1 (the past rows of layer l + 2: the compute waves, between C and D)
                if (gate_thread) {
                    const float* Pc = sm + o_part;
                    if (l == 0) { aps = close_elem(Pp, gm, gn); apt = close_elem(Pp, gm + 8, gn); }
                    const float zs = (close_elem(Pc, gm, gn) + aps) + axs;
                    const float zt = (close_elem(Pc, gm + 8, gn) + apt) + axt;
                    gb_store1(rs, gn * ustride + c.o_g + l * C + c0 + gm, tag, qgate(zs, zt));
                }
                CB_T(4);
                float aux_p = 0.0f, aux_q = 0.0f;      // the next layer's aux term: requested AFTER the gather (loads return in order: in front of it they would hold it up -- the
                {                                      // weight stream evicts those rows from L2 every step), consumed after the next gather
                    const GSrc sa = {c.o_g + l * C, ustride, nullptr, tags, hC, NOCHK, 0};
                    for (int i = 0; i < c.poll_delay_g; ++i) __builtin_amdgcn_s_sleep(2);
                    gather1<true>(rs, sa, sm + o_gv, C, nb, xt, c.abort, p.status);
                    if (has_res && ak < nb && tags[ak]) { aux_p = p.pproj[(size_t)(smi[ofr + ak] + l + 1) * (2 * C) + anat]; aux_q = p.qb[(l + 1) * 2 * C + anat]; }
                }
                __syncthreads();      // B
                CB_T(5);
                __syncthreads();      // C: the partial tiles of the residual + skip tile are in LDS
                CB_T(6);
                // ---------------- block output of my channels -> all-gather (the next layer's input); my skip rows accumulate here
                {      // one (row, utterance) per exchange thread: rows 0..7 the residual rows of my channels, 8.. my skip rows
                    int xv = xt; CBB_FRESH(xv);
                    const int m = xv >> 4, n = xv & 15;
                    if (n < nb && tags[n]) {
                        const float acc = close_elem(sm + o_part, m, n);
                        if (m < 8) {
                            if (has_res) gb_store1(rs, n * ustride + smi[o_lt + (l + 1) * CBB_LT + sl] + c0 + m, tag, (acc + sm[o_bres + l * 8 + m]) + xres);
                        } else if (m - 8 < SB) {
                            const int a = o_acc + (smi[o_lt + l * CBB_LT + 2] ? 8 * CBB_NU : 0) + (m - 8) * CBB_NU + n;
                            sm[a] = sm[a] + (acc + sm[o_bsk + l * 8 + (m - 8)]);
                        }
                    }
                }
                CB_T(7);
                if (has_res) {
                    const GSrc sa = {LT(l + 1, sl), ustride, nullptr, tags, hC, NOCHK, 0};
                    for (int i = 0; i < c.poll_delay_x; ++i) __builtin_amdgcn_s_sleep(2);
                    gather1<true>(rs, sa, sm + o_x, C, nb, xt, c.abort, p.status);
                    sm[o_auxz + ((l + 1) & 1) * 256 + xt] = (ak < nb && tags[ak]) ? __builtin_fmaf(wj, aux_p, aux_q) : 0.0f;
                }
                __syncthreads();      // D
                CB_T(8);
            }
            // ---------------- tail: relu(skip total) -> post 1x1 #1 -> relu -> post 1x1 #2 (qpnet.py:566-571)
            if (xt < SB * CBB_NU) {
                const int r = xt >> 4, k = xt & 15;
                if (k < nb && tags[k]) {
                    const float tot = sm[o_acc + r * CBB_NU + k] + sm[o_acc + 8 * CBB_NU + r * CBB_NU + k];      // sum(skip_F) + sum(skip_A)  (qpnet.py:505)
                    gb_store1(rs, k * ustride + c.o_y1 + s0 + r, tag, tot > 0.0f ? tot : 0.0f);
                }
            }
            {
                const GSrc sa = {c.o_y1, ustride, nullptr, tags, hS, NOCHK, 0};
                for (int i = 0; i < c.poll_delay_t; ++i) __builtin_amdgcn_s_sleep(2);
                gather1<true>(rs, sa, sm + o_xp, S, nb, xt, c.abort, p.status);
            }
            __syncthreads();      // T1
            __syncthreads();      // T2: the partial tiles of post 1x1 #1
            {
                int xv = xt; CBB_FRESH(xv);
                const int m = xv >> 4, n = xv & 15;
                if (m < SB && n < nb && tags[n]) { const float v = close_elem(sm + o_part, m, n) + sm[o_bp1 + m]; gb_store1(rs, n * ustride + c.o_y2 + s0 + m, tag, v > 0.0f ? v : 0.0f); }
                const GSrc sa = {c.o_y2, ustride, nullptr, tags, hS, NOCHK, 0};
                for (int i = 0; i < c.poll_delay_t; ++i) __builtin_amdgcn_s_sleep(2);
                gather1<true>(rs, sa, sm + o_gv, S, nb, xt, c.abort, p.status);
            }
            __syncthreads();      // T3
            __syncthreads();      // T4: the partial tiles of post 1x1 #2
            {
                int xv = xt; CBB_FRESH(xv);
                const int m = xv >> 4, n = xv & 15;
                if (m < QB && n < nb && tags[n]) gb_store1(rs, n * ustride + c.o_lg + q0 + m, tag, close_elem(sm + o_part, m, n) + sm[o_bp2 + m]);
                // all logits of all utterances -> LDS [nb][Q] (plain): every workgroup derives the same next samples
                const GSrc sa = {c.o_lg, ustride, nullptr, tags, hQ, NOCHK, 0};
                for (int i = 0; i < c.poll_delay_t; ++i) __builtin_amdgcn_s_sleep(2);
                gather1<false>(rs, sa, sm + o_lg, Q, nb, xt, c.abort, p.status);
            }
            __syncthreads();      // T5
            CB_T(9);
            pick_utts(t, wave - 4);      // (utterances 0..3 mod 8; the compute waves: 4..7 mod 8)
            if (xt == 0) smi[o_ctl + 1] = __hip_atomic_load(c.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();      // T6
            CB_T(10);
            if (smi[o_ctl + 1]) break;            // a peer gave up: leave together
        }
    }
#ifdef QPN_ENABLE_STAMPS
    __syncthreads();
    if (prof) for (int k = 0; k < 17; ++k) p.stamps[(size_t)k * QPN_NW] = k == 0 ? 0 : ((long long*)(sm + o_prof))[k - 1];
#endif
}

// ------------------------------------------------------------------------------------------ host side
// geometries this kernel takes: 8 channels per workgroup make the gate tile; S / G and Q / G rows fit their tiles; K = 256 or 512 chunks trees
bool qpn_coopb_supported(const Geom& g) {
    if (g.L < 2 || g.C % 8 || g.C < 128 || g.S > g.C || g.Q > g.C) return false;      // (the skip / logit vectors reuse the channel images)
    if ((g.C & (g.C - 1)) || (g.S & (g.S - 1)) || (g.Q & (g.Q - 1)) || g.Q < 64) return false;      // (the gathers split a vector's granule pairs over the threads by shifts)
    const int G = g.C / 8;
    if (g.S % G || g.Q % G || g.S / G < 1 || g.S / G > 8 || g.Q / G < 1 || g.Q / G > 16) return false;
    const int RC = g.Cp / 16, RS = g.Sp / 16;
    if ((RC != 16 && RC != 32) || (RS != 16 && RS != 32) || g.Cp != g.C || g.Sp != g.S) return false;
    return (size_t)cbb_lds_floats(g.C, g.L) * sizeof(float) <= 160 * 1024;
}

int qpn_launch_decode_coopb(qpn_handle* h, DecodeParams& p, int B, hipStream_t stream) {
    const Geom& g = h->g;
    const int L = g.L, C = g.C, S = g.S, Q = g.Q;
    CoopbParams c; memset(&c, 0, sizeof(c));
    c.G = C / 8; c.SB = S / c.G; c.QB = Q / c.G; c.RC = g.Cp / 16; c.RS = g.Sp / 16; c.B = B;
    long o = 0;
    for (int l = 0; l < L; ++l) {
        c.o_ring[l] = (int)o; o += (long)p.rings[l].len * C;
        c.zc[l] = h->cb_zc[l]; c.zp[l] = h->cb_zp[l]; c.rs[l] = h->cb_rs[l];
        c.f_resb[l] = (int)g.layers[l].resb; c.f_skipb[l] = (int)g.layers[l].skipb; c.adaptive[l] = g.layers[l].adaptive;
    }
    c.p1 = h->cb_p1; c.p2 = h->cb_p2; c.base4 = h->cb_base4; c.per_w = h->cb_per_w;
    if (h->h_map.size() * sizeof(float) >= (1ull << 32)) return 1;      // (the fragment blocks are addressed with 32-bit byte offsets)
    c.wpk_bytes = (unsigned)(h->h_map.size() * sizeof(float));
    c.o_g = (int)o; o += (long)L * C; c.o_y1 = (int)o; o += S; c.o_y2 = (int)o; o += S; c.o_lg = (int)o; o += Q;
    o = (o + 15) & ~15L;
    if (o * CBB_NU * 8 >= (1L << 31)) return 1;             // (a group's exchange block is addressed with 32-bit byte offsets: the caller takes decode_coop.hip)
    c.utt_stride = o; c.f_p1b = (int)g.post1_b; c.f_p2b = (int)g.post2_b;
    // groups: as many as fit the chip (all workgroups of a launch must be resident together, one per CU), the utterances spread evenly over them
    const int max_groups = h->n_cus / c.G > 0 ? h->n_cus / c.G : 1;
    int ngroups = (B + CBB_NU - 1) / CBB_NU;
    if (ngroups < max_groups) ngroups = max_groups < B ? max_groups : B;
    if (ngroups > max_groups) ngroups = max_groups;           // (more than 16 per group would be needed: the caller splits the batch)
    c.NBper = (B + ngroups - 1) / ngroups;
    if (h->dk.coopb_per > 0 && h->dk.coopb_per <= CBB_NU && (B + h->dk.coopb_per - 1) / h->dk.coopb_per <= max_groups) c.NBper = h->dk.coopb_per;      // (dev knob)
    if (c.NBper > CBB_NU) { qpn_set_error("batched cooperative decode: %d utterances exceed one launch (%d groups of 16)", B, max_groups); return QPN_EINVAL; }
    ngroups = (B + c.NBper - 1) / c.NBper;
    const size_t xwords = (size_t)o * B + 16;
    if (xwords > h->xch_cap) {
        if (h->d_xch) (void)hipFree(h->d_xch);
        h->d_xch = nullptr; h->xch_cap = 0;
        if (hipMalloc(&h->d_xch, xwords * sizeof(unsigned long long)) != hipSuccess) { qpn_set_error("hipMalloc(%zu MiB) for the decode exchange buffers failed", xwords * 8 >> 20); return QPN_ENOMEM; }
        h->xch_cap = xwords;
    }
    c.xch = h->d_xch + 16; c.abort = (int*)h->d_xch;
    const size_t lds_bytes = (size_t)cbb_lds_floats(C, L) * sizeof(float);
    // first poll of an edge's gather: 8 / 4 x 128 clocks after the exchange waves reach it (QPN_COOPB_DELAY_G / _X / _T, dev).  Polls sent before anything can have been
    // published return stale a round trip later and load the memory side the publishes go through: B = 20: 151 -> 113 us per step, B = 4: 98.6 -> 91.3, B = 64: 220 -> 189
    // (tools/coopb_delay_sweep.py, profiles/r06_coopb_poll_delay.txt)
    c.poll_delay_g = h->dk.coopb_delay[0]; c.poll_delay_x = h->dk.coopb_delay[1]; c.poll_delay_t = h->dk.coopb_delay[2];
#ifdef QPN_ENABLE_STAMPS
    c.dev_nostream = getenv("QPN_COOPB_NOSTREAM") ? 1 : 0; c.dev_nocheck = getenv("QPN_COOPB_NOCHECK") ? 1 : 0;
#endif
    QPN_HIP(hipFuncSetAttribute((const void*)k_decode_coopb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    QPN_HIP(hipMemsetAsync(h->d_xch, 0, xwords * sizeof(unsigned long long), stream));      // tags, rings, abort flag
#ifdef QPN_TESTING
    if (h->dk.test_pipe_gives_up) {
        // test hook (a -DQPN_TESTING build only): the launch behaves as if a wait had timed out at once (abort flag raised, status bit 4): exercises the
        // re-run on the per-utterance cooperative kernel without needing a CU-masked device (tests/test_decode_gpu.py)
        static const int one = 1, four = 4;
        QPN_HIP(hipMemcpyAsync(h->d_xch, &one, sizeof(int), hipMemcpyHostToDevice, stream));
        QPN_HIP(hipMemcpyAsync(h->d_status, &four, sizeof(int), hipMemcpyHostToDevice, stream));
    }
#endif
    hipLaunchKernelGGL(k_decode_coopb, dim3(c.G, ngroups), dim3(CBB_NT), lds_bytes, stream, p, c);
    QPN_HIP(hipGetLastError());
    h->cb_groups = ngroups; h->cb_per = c.NBper;
    return QPN_OK;
}
