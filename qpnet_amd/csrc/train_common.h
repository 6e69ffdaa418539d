// train_common.h -- shared pieces of the MFMA training kernels (forward / backward).
//
// All activations are TIME-MAJOR [n][channels] fp32 (one 256-byte row per time step for C=64):
// the pitch-dependent gather then moves whole contiguous rows, and every contraction is a
// GEMM with M = time.  Contractions run on the fp32 matrix cores: v_mfma_f32_16x16x4_f32
// (exact fp32, 256 FLOP/clk/CU), A operand from an LDS tile, B operand (weights) from a
// fragment-ordered copy that a gather kernel refreshes from the flat parameter vector
// every step (weights change every step; 0.5 M floats).
#pragma once
#include "qpn_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TR_MAXL QPN_MAX_LAYERS   // layers supported by the training kernels (the reference's deepest shipped stack, Rd10Rr3Ed4Er1, has 34: src/utils/param_model.py:66-72)

// leading dimension for an LDS A-tile read "row = lane&15, k = lane>>4" with ds_read_b32:
// ld == 2 (mod 32) makes the 32-lane groups conflict free (MI355X_MICROARCH.md §LDS).
__host__ __device__ static inline int tr_lda(int K) { return ((K + 29) / 32) * 32 + 2; }
// leading dimension for transposed reads "k-row = lane>>4, col = lane&15": ld == 16 (mod 32)
__host__ __device__ static inline int tr_ldt(int N) { return ((N + 15) / 32) * 32 + 16; }

struct TrLayer {
    int adaptive, dilation;
    int s_in, s_out;          // first valid local row of the layer input / output
    int tap_off;              // offset (ints) of this layer's tap table in the TAP buffer, or -1 (fixed)
    // fragment-ordered weight blocks (float4 offsets into the packed training weights)
    int w1_f4, wr_f4;         // fwd: [Ktp x 2C], [C x C]
    int w1t_f4, wrt_f4;       // bwd: [2C x Ktp] (dA = dZ.W1), [C x C] (dg = dXout.Wr)
    int bias1, biasr;         // float offsets into the packed bias block
};

struct TrainParams {
    int C, S, Q, A, Ap, L, U;
    int B, T, F, Td, BL, N1, N0, maxd;
    int Kt, Ktp;              // 2C + Ap, padded to 16
    int LC;                   // L * C (skip contraction depth)
    // global buffers
    const float* flat;        // parameters (state_dict order)
    const float4* wp;         // fragment-ordered weights
    const float* ct;          // causal conv table transposed to [tap][class][C] (a row per looked-up class: coalesced)
    float* bp;                // packed biases (hoist: k_aux_proj adds b_up * sum_a Va[n][a] to the gate biases)
    const int64_t* x; const float* h; const float* d;
    float* X;                 // [L+1][B][N1][C]
    float* SG; float* TH;     // [L][B][N1][C]: sigma, and -- on the tile path -- the gate PRODUCT sigma * tanh (tr_gate_bwd below; the GEMM path of the wide stacks keeps tanh here and the product in TrainGemm::G)
    float* HUP;               // [B][N1][Ap]
    // Auxiliary 1x1 at FRAME rate (hoist != 0; n_resch 64, upsampling_factor >= 16).  The upsampling deconvolution is rank 1
    // (h_up[a][U f + j] = h[a][f] w_up[j] + b_up, reference src/nets/qpnet.py:134-158), so the aux 1x1 of a gated block (qpnet.py:215-216,
    // 663-664) commutes with it:  Va . h_up[:, t] + ba = w_up[j(t)] * (Va . h[:, f(t)]) + (b_up * Va . 1 + ba).  The second term rides in
    // the packed bias; the first is a rank-<=2 update of a 16-row tile (U >= 16: at most two frames per tile) -- ONE extra 16x16x4 MFMA per
    // accumulator instead of the 48 padded aux columns of the K = 176 contraction (the decode kernels have always done this).
    int hoist;                // Ktp = 2C (no aux columns in the A tile), PA / WJ valid, HUP not written
    int nfr, ffirst;          // frames the N1 rows touch, first of them: row n -> q = F U - N1 + n, f = q / U, j = q - f U
    float* PA;                // [L][B][nfr + 1][2C]: PA_l[b][f - ffirst][n] = sum_a Va_l[n][a] h[b][a][f]; row nfr: padding (finite, never used by a real row)
    float2* WJ;               // [N1 + 16]: {w_up[j(n)], j(n) as int bits} (same for every batch item); tail = row N1 - 1
    int* TAP;                 // [LA][B][N1]
    int* XC;                  // [B][N1+1] sample classes of the rows (x % Q), for the causal conv's weight gradient
    float* S0; float* Y0;     // [B][BL][S] pre-relu skip sum / post1
    float* logits;            // [B][BL][Q] (nullptr: not stored -- only with the fused cross entropy below)
    // fused cross entropy in the post-net kernel (qpn_train_forward_loss): targets = last BL columns of ce_tgt rows; nullptr = off
    const int64_t* ce_tgt; int64_t ce_stride; float* ce_dlogits; double* ce_loss;
    int* status;
    float* scratch_rows;      // [B * 1024 workgroups][2][128] floats nobody reads: target of the persistent kernels' out-of-range row stores
    unsigned* qctl;           // control words of the stack work queues (train_stack.hip), control words and sub-queue heads zeroed by k_train_prep; or nullptr
    int4* qtab;               // [qtotal + 1][2] tile table of the forward queue, written by k_train_prep (tr_queue_entry_fwd); or nullptr
    int4* qtab_b;             // ... and of the backward queue (tr_queue_entry_bwd)
    int qtotal;               // positions of a queue: sum over layers of B * qT[l]
    int qP[TR_MAXL + 1];      // first position of layer l (forward order)
    int qT[TR_MAXL];          // 16-row tiles per batch item in layer l
    // post-net packed blocks
    int ws_f4, p1_f4, p2_f4;          // fwd: [LC x S], [S x S], [S x Q]
    int wst_f4, p1t_f4, p2t_f4;       // bwd: [S x LC], [S x S], [Q x S]
    int bias_s, bias_p1, bias_p2;     // packed bias offsets (bias_s = sum over layers of skip biases)
    int64_t causal_w, causal_b, up_w, up_b;
    // wave-per-tile stack kernels (train_stackw.hip; hoist form): per-layer weight images in A-operand order with the permuted k of fragA_pack, layer l at
    // offset l * 4096 (w1q: the backward's 128 x 128 block d[x_cur | x_past] = dZ . W1^T; w1p: the forward's gate block) / l * 1024 (wrq / wrp: the 64 x 64 residual 1x1)
    // float4 words behind these; -1: not packed
    int w1q_f4, wrq_f4, w1p_f4, wrp_f4;
    TrLayer layers[TR_MAXL];
};

// slab offsets (floats) of every weight-grad block; host side only
struct TrainSlabs {
    int g_w1[TR_MAXL], g_b1[TR_MAXL], g_wr[TR_MAXL], g_br[TR_MAXL], g_ws[TR_MAXL], g_bs;
    int g_p1, g_bp1, g_p2, g_bp2;
    int g_early0, g_early1;           // slab range [g_early0, g_early1): skip 1x1 / skip bias / post-net blocks, complete before the layer backward ends
    int g_cw, g_cb;                   // causal conv table [tap][C][Q] and bias, or -1: histogram kernel (k_causal_bwd)
};

struct TrainBwd {            // backward-only buffers / maps (see train_bwd.hip)
    const float* dlogits; float* gflat;
    float* DXA[2]; float* DXB[2];     // [B][N1][C] grads w.r.t. a layer input: own-position part / scattered pitch-tap part
    float* DZ;                        // [B][N1][2C] gate pre-activation grads of the current layer
    float* DS0; float* DY0;           // [B][BL][S]
    float* DGS;                       // [B][BL][L*C]
    float* DHUP;                      // [B][N1][Ap]  (hoist == 0)
    float* DPA;                       // hoist: [L][B][nfr + 1][2C], D_l[b][fx][n] = sum_{t in frame} w_up[j(t)] dZ_l[t][n] (float atomics; zeroed per backward)
    float* GW;                        // hoist: [L][B][N1], G_l[b][t] = sum_n dZ_l[t][n] PA_l[b][fx(t)][n]  (the upsampling kernel's gradient, per row)
    float* EB;                        // hoist: [L][TR_EB_SLOTS][2C] directly behind DPA, zeroed with it; sum over the slots = E_l[n] = sum_{b, t} dZ_l[t][n] (the gate-bias gradient, accumulated by the layer backward itself)
    float* slab;                      // [NCH][gstage] split-time partial weight grads
    const int* gdst; const int* gdst_list;   // inverse map in CSR form: slab element s feeds flat-grad entries gdst_list[gdst[s] .. gdst[s+1])
    const int* gzero; int n_gzero;    // flat-grad entries no slab element feeds (written by their own kernels afterwards): zeroed by the reduction
    int nch, gstage;
    int64_t n_params;
    const struct TrainSlabs* sl;      // HOST pointer (launchers only): slab offsets of every weight-grad block -- kept out of the kernel arguments (4 KB budget at 48 layers)
    float gscale;                     // the flat gradient is multiplied by this (data-parallel: the rank's row count)
    int append_scale;                 // ... and gflat[n_params .. n_params+3] = {gscale, flagged ? 1 : 0, 0, 0} (rides in the all-reduce); 2: the trailer leaves with an early exchange bucket (never rewritten by the last launch)
    const int* status;                // the handle's sticky status word (read for the trailer's flag)
    hipStream_t side; hipEvent_t ev_fork, ev_join, ev_mid;   // side stream of the weight gradients that run under the layer backward (owned by TrainState)
    hipEvent_t ev_early; int* early_recorded;                // recorded on the side stream behind the early reduction when it also wrote the trailer (qpn_train_early_bucket)
};


// aux hoist: what k_aux_proj / k_aux_tail need of every layer (flat offsets of the aux 1x1 weights, offset of the packed gate bias, first valid row of the call) ...
struct AuxGeom { int auxS[TR_MAXL], auxT[TR_MAXL], bias1[TR_MAXL], s_out[TR_MAXL]; };
// ... and their (compact) kernel arguments
struct AuxArgs {
    const float* flat; const float* h; float* PA; float* bp; const float* DPA; const float* EB; const float* GW; float* gflat;
    float gscale; int C, A, L, B, F, U, N1, nfr, ffirst; int64_t up_w, up_b;
    AuxGeom g;
};
static inline AuxArgs tr_aux_args(const TrainParams& p, const TrainBwd* bw, const AuxGeom& ag) {
    AuxArgs a; a.flat = p.flat; a.h = p.h; a.PA = p.PA; a.bp = p.bp; a.DPA = bw ? bw->DPA : nullptr; a.EB = bw ? bw->EB : nullptr; a.GW = bw ? bw->GW : nullptr;
    a.gflat = bw ? bw->gflat : nullptr; a.gscale = bw ? bw->gscale : 1.f;
    a.C = p.C; a.A = p.A; a.L = p.L; a.B = p.B; a.F = p.F; a.U = p.U; a.N1 = p.N1; a.nfr = p.nfr; a.ffirst = p.ffirst; a.up_w = p.up_w; a.up_b = p.up_b; a.g = ag;
    return a;
}
// Launch-plan knobs, parsed ONCE per handle from the environment (train_init): nothing in the per-step launch path calls getenv().
// All optional; the defaults are the measured best.  tests/test_train_gpu.py builds a fresh model (= a fresh handle) per arrangement.
struct TrainKnobs {
    bool serial;                      // QPN_TRAIN_SERIAL=1: every launch of a step on the caller's stream (also forced while a profile is taken)
    bool stack_q_fwd, stack_q_bwd;    // QPN_STACK_QUEUE=0 / QPN_STACK_QUEUE_BWD=0: a launch per layer instead of the work-queue launches (train_stack.hip)
    int stack_wgs, stack_wgs_bwd;     // QPN_STACK_WGS / QPN_STACK_WGS_BWD: grid sizes of the queue launches (0: 2 / 1.5 workgroups per CU)
    bool persist_fwd, persist_bwd;    // QPN_LAYER_PERSIST=0 / QPN_LAYER_BWD_PERSIST=0: the tile-per-workgroup layer kernels at n_resch 64
    bool wgrad_generic;               // QPN_WGRAD_GENERIC=1: the run-time-tiled weight-gradient kernel (k_wgrad2)
    int wgrad_chunks, wgrad_chunks_side;   // QPN_WGRAD_CHUNKS / QPN_WGRAD_CHUNKS_SIDE: time chunks (= partial slabs) of the weight gradients
    bool post_fuse;                        // QPN_POST_FUSE=0: qpn_train_step runs the post-net's forward and backward as two kernels (k_post_fwd_w, k_post_bwd_w) instead of k_post_fb_w
    bool up_side, reduce_early, wr_side;   // QPN_UP_SIDE=0 / QPN_REDUCE_EARLY=0 / QPN_WR_SIDE=0: where the backward's small launches run
    bool post_pair, zero_in_post, post_wide;   // QPN_POST_WGRAD_PAIR=0 / QPN_ZERO_IN_POST=0 / QPN_POST_WIDE=0
    bool xcd_swizzle;                 // QPN_NO_XCD_SWIZZLE=1 clears it
    bool ce_separate;                 // QPN_CE_SEPARATE=1: cross entropy as its own kernel behind the forward
    bool event_fence;                 // QPN_EVENT_FENCE=1: system-scope fence at the fork / join events
    bool aux_hoist;                   // QPN_AUX_HOIST=0: the auxiliary 1x1 contracted at sample rate (K = 176) even where the frame-rate form applies
    int stack_wave_fwd; bool stack_wave_bwd;   // QPN_STACK_WAVE_FWD=1 / QPN_STACK_WAVE_BWD=1: the transposed-product queue kernels of train_stackw.hip (opt-in experiments) instead of k_stack_fwd / k_stack_bwd
    int stack_waves;                  // waves per workgroup (= per CU) of the wave-per-tile kernels: 4, one per SIMD
    bool test_stack_gives_up;         // -DQPN_TESTING builds only (QPN_TEST_STACK_GIVES_UP=1): the queue launches' published flags are made unrecognisable
};
void qpn_train_knobs_parse(TrainKnobs& k);

#define TR_QHEAD_STRIDE 1056   // words between sub-queue heads (4224 bytes: different memory channels)
#define TR_QHDR_WORDS 17920    // control block: [1] abort, counters [4..6] forward, [8..10] backward, [1024 + (8 dir + q) * TR_QHEAD_STRIDE]: sub-queue heads
// Work queue of the one-launch residual stack (train_stack.hip): positions = (layer, batch item, 16-row tile) in layer-major order
struct StackQ {
    unsigned* flags;          // [sq_fidx(position)] == epoch once the position's output rows are visible device-wide
    unsigned* head;           // 8 sub-queue heads, TR_QHEAD_STRIDE words apart (zeroed before the launch): next ticket of each
    unsigned* stats;          // dev counters: [0] escalations (a bounded wait ran out), [1] polls of waits, [2] waves that did not find the flags at first look
    unsigned* abort;          // raised when a wait timed out: every workgroup stops waiting and drains
    const int4* tab;          // tile table [total + 1][2] (entry total: the invalid tile)
    unsigned epoch;           // never 0; a new one per forward
    unsigned epoch_pub;       // what a published flag carries: the epoch (a test hook makes it differ, so that every wait runs out)
    unsigned spin_limit;      // polls before a wait gives up
    int total, nq;            // positions; sub-queues in use
};

// float offset in PA / DPA of the frame that holds row n0 (layer l, batch item b)
__host__ __device__ __forceinline__ int tr_pa_off(const TrainParams& p, int l, int b, int n0) {
    const int q = p.F * p.U - p.N1 + n0;                       // (F U < 2^31: checked by the host)
    return ((l * p.B + b) * (p.nfr + 1) + (q / p.U - p.ffirst)) * 2 * p.C;
}

// One entry of the forward tile table (device side of k_train_prep; read by k_stack_fwd):
//   a = {first row n0, layer | batch item << 8 | last layer << 24 | valid << 25, first producer position, producer positions}
//   b = {row offset of the layer's input in X (= of its sigma / tanh rows), tap table offset, row offset of the aux features (hoist: float offset of the tile's first frame in PA), 0}
// producers of tile (l, t): the tiles of layer l - 1 that hold the rows n0 - reach .. n0 + 15 (own rows and every row a tap can touch;
// reach = dilation (fixed) or dilation * maxd (adaptive), reference src/nets/qpnet.py:271-306)
__device__ __forceinline__ void tr_queue_entry_fwd(const TrainParams& p, int pos, int4& a, int4& b) {
    const bool valid = pos < p.qtotal;
    const int ps = valid ? pos : 0;
    int l = 0;
    for (int k = 1; k < p.L; ++k) l += ps >= p.qP[k] ? 1 : 0;
    const int r = ps - p.qP[l], T = p.qT[l];
    const int bi = r / T, t = r - bi * T;
    const TrLayer ly = p.layers[l];
    const int n0 = ly.s_out + 16 * t;
    int first = 0, n = 0;
    if (valid && l > 0) {
        const int reach = ly.s_out - ly.s_in, s_prev = p.layers[l - 1].s_out;
        int lo = n0 - reach; if (lo < s_prev) lo = s_prev;
        int hi = n0 + 15; if (hi > p.N1 - 1) hi = p.N1 - 1;
        const int t_lo = (lo - s_prev) >> 4, t_hi = (hi - s_prev) >> 4;
        first = p.qP[l - 1] + bi * p.qT[l - 1] + t_lo; n = t_hi - t_lo + 1;
    }
    a = make_int4(n0, l | (bi << 8) | (l == p.L - 1 ? 1 << 24 : 0) | (valid ? 1 << 25 : 0), first, n);
    b = make_int4((l * p.B + bi) * p.N1, ly.tap_off + bi * p.N1, p.hoist ? tr_pa_off(p, l, bi, n0) : bi * p.N1, 0);
}

// The backward queue runs the layers from the last to the first (position blocks in that order, tiles by ascending rows).  Entry of tile
// (l, t), rows n0 .. n0 + 15 of layer l's OUTPUT gradient (= grads w.r.t. X[l + 1]):
//   a = {n0, layer | batch item << 8 | last layer << 24 | valid << 25 | adaptive << 26, first producer position, producer positions}
//   b = {row offset (l * B + b) * N1 of the layer's arrays, tap table offset, b * N1, float offset of the skip-path gate grads b * BL * LC + l * C}
// producers: the tiles of layer l + 1 whose own rows or pitch-tap scatter touch rows n0 .. n0 + 15 of its INPUT gradient, i.e. its rows
// n0 .. n0 + 15 + reach(l + 1) (autograd of the gather, reference src/nets/qpnet.py:295-298, 626-640)
__device__ __forceinline__ void tr_queue_entry_bwd(const TrainParams& p, int pos, int4& a, int4& b) {
    const bool valid = pos < p.qtotal;
    const int ps = valid ? pos : 0;
    int k = 0, base = 0;                              // k-th processed layer = layer L - 1 - k
    for (int j = 0; j + 1 < p.L; ++j) { const int n = p.B * p.qT[p.L - 1 - j]; if (ps >= base + n) { base += n; k = j + 1; } else break; }
    const int l = p.L - 1 - k;
    const int r = ps - base, T = p.qT[l];
    const int bi = r / T, t = r - bi * T;
    const TrLayer ly = p.layers[l];
    const int n0 = ly.s_out + 16 * t;
    int first = 0, n = 0;
    if (valid && l < p.L - 1) {
        const TrLayer up = p.layers[l + 1];
        const int reach = up.s_out - up.s_in;
        int lo = n0 - up.s_out; if (lo < 0) lo = 0;
        int hi = n0 + 15 + reach; if (hi > p.N1 - 1) hi = p.N1 - 1;
        hi -= up.s_out;
        const int t_lo = lo >> 4, t_hi = hi >> 4;
        const int pbase = base - p.B * p.qT[l + 1];   // layer l + 1 is the block in front of this one
        first = pbase + bi * p.qT[l + 1] + t_lo; n = t_hi - t_lo + 1;
    }
    a = make_int4(n0, l | (bi << 8) | (l == p.L - 1 ? 1 << 24 : 0) | (valid ? 1 << 25 : 0) | (ly.adaptive ? 1 << 26 : 0), first, n);
    b = make_int4((l * p.B + bi) * p.N1, ly.tap_off + bi * p.N1, p.hoist ? tr_pa_off(p, l, bi, n0) : bi * p.N1, bi * p.BL * p.LC + l * p.C);
}

// K-major, zero-padded weight blocks of the GEMM path (train_gemm.hip; n_resch > 128): float offsets into `wp`
struct TrainGemm {
    const float* wp;                  // packed weights (refreshed from the flat parameters every step)
    float* G;                         // [L][B][N1][C] gate outputs sigma*tanh (A operand of the residual / skip contractions)
    int K1;                           // 2C + n_aux padded to 32: K of the gate GEMM
    int N1g, Cg, Sg, Qg, LCg, Ktg;    // column counts padded to 128: gate tiles, C, S, Q, L*C, 2C + n_aux
    long w1[TR_MAXL], w1t[TR_MAXL], wr[TR_MAXL], wrt[TR_MAXL];
    long ws, wst, p1, p1t, p2, p2t;
};

// Workgroup barrier that orders LDS traffic only.  hipcc's __syncthreads() also drains vmcnt: every global load / store the wave
// has in flight -- in the persistent kernels that is the NEXT tile's prefetch, requested a moment earlier precisely so that it
// flies under this tile's contractions -- would be waited for at every barrier.  The LDS writes of this wave are complete
// (lgkmcnt(0)) before it arrives; registers still being loaded are tracked by the compiler's own s_waitcnt at their first use.
#define TR_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

int qpn_num_cus();                                   // compute units of the current device (cached per device)

// optional per-kernel-group timing (HIP events on the launch stream; bench.py roofline)
// (one group per LAUNCH of the heavy kernels, so that bench.py can name the single longest kernel: PG_WGRAD = the gate contraction's weight
//  gradient dW1, then the residual 1x1's, the skip 1x1's, the post-net pair's and the causal table's)
enum { PG_PREP = 0, PG_LAYER_FWD, PG_POST_FWD, PG_CE, PG_POST_BWD, PG_WGRAD, PG_LAYER_BWD, PG_GRAD_TAIL, PG_ADAM, PG_ALLREDUCE,
       PG_WGRAD_WR, PG_WGRAD_SKIP, PG_WGRAD_POST, PG_WGRAD_CAUSAL, PG_COUNT };
void qpn_prof_mark(int group, hipStream_t stream);
bool qpn_prof_serial();                              // a SERIAL per-group timing is in progress (keeps the step on one stream; the overlapped mode does not)
bool qpn_prof_active();                              // per-group timing in progress (keeps a step on one stream)   // attributes the work enqueued since the previous mark to `group`

// XCD-aware tile index: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2); remapped so that the
// workgroups sharing an XCD own a CONTIGUOUS range of time tiles -- a tile's pitch-tap rows are then the rows a neighbour on the
// same L2 fetched a moment ago (bijective form for any grid size; QPN_NO_XCD_SWIZZLE=1 at launch passes swz = 0).
__device__ __forceinline__ int tr_xcd_tile(int orig, int nwg, int swz) {
    if (!swz || nwg <= 8) return orig;
    const int q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + orig / 8;
}

// tile range of workgroup `orig` (hardware id) of a grid of G over `tiles` tiles: ranges are contiguous in the XCD-aware order
// g = tr_xcd_tile(orig) and as even as possible; the `rem` workgroups that take one tile more are the ones with the LOWEST
// hardware ids -- with two workgroups per CU dealt in id order, a CU then gets (base + 1) + base tiles rather than 2 (base + 1)
__device__ __forceinline__ void tr_tile_range(int orig, int G, int tiles, int swz, int& t_first, int& t_count) {
    const int base = tiles / G, rem = tiles - base * G;
    if (!swz || G <= 8) { t_first = orig * base + (orig < rem ? orig : rem); t_count = base + (orig < rem ? 1 : 0); return; }
    const int q = G / 8, r = G % 8, xcd = orig % 8, j = orig / 8;
    const int g = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;          // = tr_xcd_tile(orig, G, 1)
    int before = 0;                                                                       // workgroups with an extra tile among g' < g
    for (int x = 0; x < xcd; ++x) before += rem > x ? (rem - x + 7) / 8 : 0;
    const int mine = rem > xcd ? (rem - xcd + 7) / 8 : 0;
    before += j < mine ? j : mine;
    t_first = g * base + before; t_count = base + (orig < rem ? 1 : 0);
}

// ---- auxiliary 1x1 at frame rate (TrainParams::hoist): pieces shared by the per-layer and the work-queue kernels
// frame slot (0: the tile's first frame, 1: the next one) of tile row `row` whose within-frame offset is j: the rows of a tile are consecutive
// samples and U >= 16, so the offset has wrapped exactly when it is smaller than the row index
__device__ __forceinline__ int tr_aux_slot(float2 wj, int row) { return __float_as_int(wj.y) < row ? 1 : 0; }
// forward, A operand of the extra MFMA step: lane (row = lane & 15, k = lane >> 4) holds w_up[j(row)] in the k-slot of its frame, 0 elsewhere
__device__ __forceinline__ float tr_aux_a(float2 wj, int lane) { return tr_aux_slot(wj, lane & 15) == (lane >> 4) ? wj.x : 0.f; }
__device__ __forceinline__ float tr_dpp_row_sum(float a) {   // sum over the 16 lanes of a row, in every lane of it
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x128, 0xf, 0xf, true));      // row_ror:8
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x124, 0xf, 0xf, true));      // row_ror:4
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
    a += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
    return a;
}
// max / sum over the 64 lanes of a wave, the result in every lane: four DPP steps inside the 16-lane rows, then the four row results through
// v_readlane (scalar operands) -- ~12 VALU instructions and no trip through the LDS crossbar, against six dependent ds_bpermute of __shfl_xor
__device__ __forceinline__ float tr_wave_max(float a) {
    a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x128, 0xf, 0xf, true)));
    a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x124, 0xf, 0xf, true)));
    a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xf, 0xf, true)));
    a = fmaxf(a, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0xB1, 0xf, 0xf, true)));
    const int i = __float_as_int(a);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}
__device__ __forceinline__ float tr_wave_sum(float a) {
    const int i = __float_as_int(tr_dpp_row_sum(a));
    return (__int_as_float(__builtin_amdgcn_readlane(i, 0)) + __int_as_float(__builtin_amdgcn_readlane(i, 16))) +
           (__int_as_float(__builtin_amdgcn_readlane(i, 32)) + __int_as_float(__builtin_amdgcn_readlane(i, 48)));
}
// backward of the frame-rate aux term for one wave's share of a 16-row tile.  The lane holds dZ of rows 4 (lane >> 4) + i, i = 0..3, gate
// columns c (sigma) and C + c (tanh), c = 16 wave + (lane & 15); wj[i] = the WJ entries of those rows, pa = {PA[f0][c], PA[f0][C + c],
// PA[f0 + 1][c], PA[f0 + 1][C + c]}.
//   d[0..3]  D contributions {frame 0 sigma, frame 0 tanh, frame 1 sigma, frame 1 tanh} for column lane & 15 -- valid in lanes 0..15 --:
//            sum over the tile's rows of w_up[j(row)] dZ[row][col] per frame, as 8 MFMAs (A' = the [frame][row] weights, B' = the lane's dZ)
//   gp[0..3] partial row sums of dZ[row][.] * PA[frame(row)][.] over this wave's 32 columns (the same value in all 16 lanes of a row group)
//   e[0..1]  the tile's column sums of dZ (sigma, tanh) for column lane & 15 -- valid in lanes 0..15 --: a third row of A' holds ones
struct TrAuxBwd { float d[4]; float gp[4]; float e[2]; };
#define TR_EB_SLOTS 32       // the gate-bias accumulators E are spread over this many slabs per layer (workgroup index mod): ~40 float atomics per address and layer
__device__ __forceinline__ TrAuxBwd tr_aux_bwd(const float2 (&wj)[4], const float (&pa)[4], const float (&dzs)[4], const float (&dzt)[4], int lane) {
    TrAuxBwd o;
    f32x4 as = (f32x4){0, 0, 0, 0}, at = (f32x4){0, 0, 0, 0};
    const int m = lane & 15, g4 = 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tr_aux_slot(wj[i], g4 + i);
        const float a = m == 2 ? 1.f : (k == m ? wj[i].x : 0.f);       // A'[m][kk = lane >> 4] of step i (row 4 kk + i): m = 0 / 1 the frame slots, m = 2 ones
        as = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dzs[i], as, 0, 0, 0);
        at = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dzt[i], at, 0, 0, 0);
        o.gp[i] = tr_dpp_row_sum(dzs[i] * (k ? pa[2] : pa[0]) + dzt[i] * (k ? pa[3] : pa[1]));
    }
    o.d[0] = as[0]; o.d[1] = at[0]; o.d[2] = as[1]; o.d[3] = at[1];        // rows 0 / 1 of the result = frame slots 0 / 1 (lanes 0..15)
    o.e[0] = as[2]; o.e[1] = at[2];
    return o;
}

// ---- one wave: acc[mt][j] += A_lds[16*mt.., :K] * Bfrag[:, nt_j]   (K multiple of 16)
// A_lds row-major with leading dim lda (floats); Bp fragment order: [(ks4*NT + nt)*64 + lane] float4,
// element e of the float4 = B[4*(4*ks4+e) + (lane>>4)][16*nt + (lane&15)].
// one 16-deep step: 4 x MT x NJ MFMAs on fragments b[j] (A rows of the step read from LDS here)
template <int MT, int NJ>
__device__ __forceinline__ void wave_gemm_step(f32x4 (&acc)[MT][NJ], const float* __restrict__ A_lds, int lda, int ks4, int arow, int ak, const float4 (&b)[NJ]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float* ap = A_lds + (size_t)(16 * mt + arow) * lda + 16 * ks4 + ak;
        const float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[j].x, acc[mt][j], 0, 0, 0);
            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[j].y, acc[mt][j], 0, 0, 0);
            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b[j].z, acc[mt][j], 0, 0, 0);
            acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b[j].w, acc[mt][j], 0, 0, 0);
        }
    }
}
// The weight fragments of step k+1 are requested BEFORE step k's MFMAs and waited for after them.  Two register sets used
// alternately and scheduling barriers around each group: written as "load next; copy; compute" hipcc rotated the loop so that
// every step loaded its OWN fragments and waited for them at once (s_waitcnt right behind the global_load: the whole L2 round
// trip exposed 11 times per K = 176 contraction -- the reason these contractions ran at 0.6 of the matrix-core rate).
template <int MT, int NJ>
__device__ __forceinline__ void wave_gemm(f32x4 (&acc)[MT][NJ], const float* __restrict__ A_lds, int lda,
                                          const float4* __restrict__ Bp, int NT, const int (&nts)[NJ], int K, int lane) {
    const int arow = lane & 15, ak = lane >> 4;
    const int nk = K / 16;
    float4 b0[NJ], b1[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) b0[j] = Bp[(size_t)nts[j] * 64 + lane];
    int ks4 = 0;
    for (; ks4 + 1 < nk; ks4 += 2) {          // two steps per trip, ONE basic block (a conditional second half lets LLVM sink the loads to their use)
#pragma unroll
        for (int j = 0; j < NJ; ++j) b1[j] = Bp[((size_t)(ks4 + 1) * NT + nts[j]) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        wave_gemm_step<MT, NJ>(acc, A_lds, lda, ks4, arow, ak, b0);
        __builtin_amdgcn_sched_barrier(0);
        const int kn = ks4 + 2 < nk ? ks4 + 2 : nk - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) b0[j] = Bp[((size_t)kn * NT + nts[j]) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
        wave_gemm_step<MT, NJ>(acc, A_lds, lda, ks4 + 1, arow, ak, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (ks4 < nk) wave_gemm_step<MT, NJ>(acc, A_lds, lda, ks4, arow, ak, b0);      // odd step count: the last fragments are already here
}

// Variant for the narrow GEMMs (one n-tile per wave): the weight fragments come from L2 (~1 k cycles away) and are requested PD
// 16-deep steps ahead through a ring of PD register sets.  [With one step of lookahead a 16-row tile's step has only 4-8 MFMAs (128-256 cycles) to hide that latency
// behind: in-kernel stamps showed ~1.2 k cycles per step, 10 k cycles for a 32-MFMA GEMM.  For the two-n-tile GEMMs the
// extra registers cost more than the latency they hide (measured), so those keep wave_gemm.]
// split form: wave_b_preload requests the first PD steps' fragments (callable long before the A tile is ready, e.g. under an
// epilogue or a barrier), wave_gemm_run consumes them and refills the ring only when K needs more than PD steps
template <int NJ, int PD>
__device__ __forceinline__ void wave_b_preload(float4 (&bq)[PD][NJ], const float4* __restrict__ Bp, int NT, const int (&nts)[NJ], int K, int lane) {
    const int nk = K / 16;
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        const int kd = d < nk ? d : nk - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) bq[d][j] = Bp[((size_t)kd * NT + nts[j]) * 64 + lane];
    }
}
template <int MT, int NJ, int PD>
__device__ __forceinline__ void wave_gemm_run(f32x4 (&acc)[MT][NJ], const float* __restrict__ A_lds, int lda, float4 (&bq)[PD][NJ],
                                              const float4* __restrict__ Bp, int NT, const int (&nts)[NJ], int K, int lane) {
    const int arow = lane & 15, ak = lane >> 4;
    const int nk = K / 16;
    for (int ks0 = 0; ks0 < nk; ks0 += PD) {
        const bool more = ks0 + PD < nk;               // a later pass of the ring needs refills (wave-uniform)
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int ks4 = ks0 + d;
            if (ks4 < nk) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float* ap = A_lds + (size_t)(16 * mt + arow) * lda + 16 * ks4 + ak;
                    const float a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[d][j].x, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bq[d][j].y, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bq[d][j].z, acc[mt][j], 0, 0, 0);
                        acc[mt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, bq[d][j].w, acc[mt][j], 0, 0, 0);
                    }
                }
            }
            if (more) {
                const int kn = ks4 + PD < nk ? ks4 + PD : nk - 1;
#pragma unroll
                for (int j = 0; j < NJ; ++j) bq[d][j] = Bp[((size_t)kn * NT + nts[j]) * 64 + lane];
            }
        }
    }
}
template <int MT, int NJ, int PD>
__device__ __forceinline__ void wave_gemm_deep(f32x4 (&acc)[MT][NJ], const float* __restrict__ A_lds, int lda,
                                               const float4* __restrict__ Bp, int NT, const int (&nts)[NJ], int K, int lane) {
    float4 bq[PD][NJ];
    wave_b_preload<NJ, PD>(bq, Bp, NT, nts, K, lane);
    wave_gemm_run<MT, NJ, PD>(acc, A_lds, lda, bq, Bp, NT, nts, K, lane);
}

// Gate backward from what the tile path's forward keeps: sigma (p.SG) and the gate PRODUCT g = sigma * tanh (p.TH -- the product is what the skip sum, the skip / residual
// weight gradients and the residual 1x1 read, so they read ONE array; only this derivative wants tanh itself).  tanh = g / sigma; where sigma has underflowed to 0 the
// product is 0 as well and both derivatives vanish.  dz_sigma = dg tanh sigma (1 - sigma) = dg g (1 - sigma);  dz_tanh = dg sigma (1 - tanh^2).
// (the GEMM path of the wide stacks keeps sigma, tanh and the product: train_gemm.hip)
__device__ __forceinline__ void tr_gate_bwd(float dg, float sg, float g, float& dzs, float& dzt) {
    const float th = sg > 0.f ? g * __builtin_amdgcn_rcpf(sg) : 0.f;
    dzs = dg * g * (1.0f - sg);
    dzt = dg * sg * (1.0f - th * th);
}

// ---- wide post-net tiles (k_post_fwd_w / k_post_bwd_w): a wave owns two column tiles and 16 MT rows
template <int MT>
__device__ __forceinline__ void post_load_a(float (&x)[MT][4], const float* __restrict__ A, int lda, int ks, int arow, int ak) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const float* ap = A + (size_t)(16 * mt + arow) * lda + 16 * ks + ak;
        x[mt][0] = ap[0]; x[mt][1] = ap[4]; x[mt][2] = ap[8]; x[mt][3] = ap[12];
    }
}
template <int MT>
__device__ __forceinline__ void post_mfma(f32x4 (&acc)[MT][2], const float (&x)[MT][4], const float4& b0, const float4& b1) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][0], b0.x, acc[mt][0], 0, 0, 0); acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][0], b1.x, acc[mt][1], 0, 0, 0);
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][1], b0.y, acc[mt][0], 0, 0, 0); acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][1], b1.y, acc[mt][1], 0, 0, 0);
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][2], b0.z, acc[mt][0], 0, 0, 0); acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][2], b1.z, acc[mt][1], 0, 0, 0);
        acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][3], b0.w, acc[mt][0], 0, 0, 0); acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[mt][3], b1.w, acc[mt][1], 0, 0, 0);
    }
}
// acc += A[16 MT][K] . B[:, two column tiles].  Software pipeline, one 16-deep step ahead on BOTH operands: the weight fragments
// (global, b) and the A fragments (LDS, x) of step ks+1 are requested before step ks's 8 MT MFMAs, every group pinned in place with
// scheduling barriers -- left alone hipcc sinks each ds_read2_b32 to its use and reuses ONE register pair for all of them: an LDS
// round trip in front of every fourth MFMA (the matrix cores 0.6 busy).  b holds step 0 on entry and, on return, the first step
// of the FOLLOWING contraction (address `next`), so no contraction starts with an exposed L2 round trip.
template <int MT>
__device__ __forceinline__ void post_gemm(f32x4 (&acc)[MT][2], const float* __restrict__ A, int lda, const float4* __restrict__ Bp, int NT, int nt0, int nk,
                                          int lane, float4 (&b)[2], const float4* __restrict__ next) {
    const int arow = lane & 15, ak = lane >> 4;
    float x0[MT][4], x1[MT][4];
    post_load_a<MT>(x0, A, lda, 0, arow, ak);
    for (int ks = 0; ks < nk; ks += 2) {                          // nk is even (K a multiple of 32)
        float4 c0 = Bp[((size_t)(ks + 1) * NT + nt0) * 64 + lane], c1 = Bp[((size_t)(ks + 1) * NT + nt0 + 1) * 64 + lane];
        post_load_a<MT>(x1, A, lda, ks + 1, arow, ak);
        __builtin_amdgcn_sched_barrier(0);
        post_mfma<MT>(acc, x0, b[0], b[1]);
        __builtin_amdgcn_sched_barrier(0);
        const float4* nx = ks + 2 < nk ? Bp + ((size_t)(ks + 2) * NT + nt0) * 64 + lane : next;
        b[0] = nx[0]; b[1] = nx[64];
        post_load_a<MT>(x0, A, lda, ks + 2 < nk ? ks + 2 : ks, arow, ak);      // (past the end: a harmless re-read)
        __builtin_amdgcn_sched_barrier(0);
        post_mfma<MT>(acc, x1, c0, c1);
        __builtin_amdgcn_sched_barrier(0);
    }
}
