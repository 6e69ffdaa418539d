// train_host.hip -- host orchestration of the training step behind the C ABI:
//   qpn_train_forward / qpn_train_backward  (QPNet.forward + autograd, reference qpnet.py:239-312)
//   qpn_ce_loss                              (CrossEntropyLoss mean, reference qpnet_train.py:430,526-528)
//   qpn_adam_step                            (torch.optim.Adam, reference qpnet_train.py:426-429,531)
#include "train_common.h"
#include "qpn_handle.h"
#include <algorithm>
#include <string.h>

int qpn_launch_fwd(const TrainParams& p, const TrainKnobs& k, const AuxGeom& ag, const StackQ* sq, hipStream_t stream, const TrainBwd* fuse_post_bwd);
void qpn_stack_fill(TrainParams& p);
int qpn_launch_ce(const float* logits, const int64_t* tgt, int64_t tgt_stride, int B, int BL, int Q, float* dlogits, double* loss, int* status, bool loss_cleared, hipStream_t stream);
int qpn_launch_bwd(const TrainParams& p, const TrainBwd& bw, const TrainKnobs& k, const AuxGeom& ag, const StackQ* sq, hipStream_t stream, bool post_done);
int qpn_launch_fwd_gemm(const TrainParams& p, const TrainGemm& w, hipStream_t stream);
int qpn_launch_bwd_gemm(const TrainParams& p, const TrainBwd& bw, const TrainGemm& w, hipStream_t stream);
int qpn_launch_adam(float* w, const float* g, float* m, float* v, int64_t n, int step, float lr, float b1, float b2, float eps, float wd, const float* den, int* status,
                    int* h_status, double* h_loss, const double* d_loss, hipStream_t stream);


struct TrainState {
    std::vector<int> h_wmap;                  // gather map of the fragment-ordered weights
    std::vector<int> h_bstart, h_blist;       // CSR of the packed biases
    std::vector<int> h_gsrc;
    int* d_wmap; float* d_wp; int* d_bstart; int* d_blist; float* d_bp; int n_bias;
    int* d_gdst; int* d_gdst_list; int* d_gzero; int n_gzero;
    bool hoist;                               // the auxiliary 1x1 runs at frame rate (TrainParams::hoist)
    TrainParams tp;                           // template with block offsets filled in
    TrainBwd bw;
    TrainSlabs slabs;                         // slab offsets of the weight-gradient blocks (bw.sl points here)
    TrainKnobs knobs;                         // launch-plan knobs, parsed once (qpn_train_knobs_parse)
    AuxGeom ag;                               // aux hoist: per-layer offsets for k_aux_proj / k_aux_tail (s_out refreshed per forward)
    // workspaces (grow only)
    float* d_ws; size_t ws_cap;               // one arena, carved per call
    int* d_tap; size_t tap_cap;
    int* d_status; double* d_loss;            // d_status: 16 words -- [0] the sticky status word, [2..3] the count of Adam updates applied (k_adam)
    hipStream_t last_stream;                  // the stream of the latest training call (where a collected status word is cleared)
    int* h_status_pinned; hipEvent_t ev_status[2]; bool status_pending[2]; int status_newest;      // qpn_train_status_enqueue / _collect: the deferred check, two slots
    double* h_loss_pinned; hipEvent_t ev_loss[2]; bool loss_pending[2]; int loss_newest;            // qpn_train_loss_enqueue / _collect: the loss read one step late, two slots of 64 partial sums
    bool fwd_valid;
    bool loss_clear;                          // the loss accumulator is zero (cleared by the forward's refresh kernel, consumed by one CE call)
    bool use_gemm;                            // wide stacks (n_resch > 128, or QPN_TRAIN_GEMM=1): the LDS-tiled GEMM path of train_gemm.hip
    int* d_ctmap; float* d_ct;                // causal conv table [tap][class][C] and its gather map
    std::vector<int> h_gmap; int* d_gmap; float* d_gwp;   // its K-major weight blocks
    TrainGemm gm;
    int64_t generation;                       // bumped by every qpn_train_forward: identifies whose activations the workspace holds
    hipStream_t side; hipEvent_t ev_fork, ev_join, ev_mid;   // side stream for the weight gradients that overlap the layer backward
    hipEvent_t ev_early; int early_recorded; int64_t early_first;   // qpn_train_early_bucket: the flat-gradient tail that is final before the layer backward ends
    int64_t bwd_generation;                                  // generation of the forward the last backward belonged to
    const float* post_bwd_dlogits;                           // ... on this dL/dlogits buffer
    bool post_bwd_done;                                      // the last forward ran the post-net's backward too (k_post_fb_w: qpn_train_step)
    bool stack_disabled;                                     // a stack-queue launch gave up once (status bit 4): this handle keeps to a launch per layer
    unsigned* d_sq; size_t sq_pos_cap, sq_per_dir;                           // stack work queues (train_stack.hip): [16 control words | forward flags | backward flags]
    StackQ sqf, sqb;
};

// The environment is read HERE, once per handle: the launch path of a step never calls getenv().
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
void qpn_train_knobs_parse(TrainKnobs& k) {
    k.serial = env_int("QPN_TRAIN_SERIAL", 0) != 0;
    k.stack_q_fwd = env_int("QPN_STACK_QUEUE", 1) != 0;
    k.stack_q_bwd = k.stack_q_fwd && env_int("QPN_STACK_QUEUE_BWD", 1) != 0;
    k.stack_wgs = env_int("QPN_STACK_WGS", 0); if (k.stack_wgs < 0 || k.stack_wgs > 4096) k.stack_wgs = 0;
    k.stack_wgs_bwd = env_int("QPN_STACK_WGS_BWD", 0); if (k.stack_wgs_bwd < 0 || k.stack_wgs_bwd > 4096) k.stack_wgs_bwd = 0;
    k.persist_fwd = env_int("QPN_LAYER_PERSIST", 1) != 0;
    k.persist_bwd = env_int("QPN_LAYER_BWD_PERSIST", 1) != 0;
    k.wgrad_generic = getenv("QPN_WGRAD_GENERIC") != nullptr;
    k.wgrad_chunks = env_int("QPN_WGRAD_CHUNKS", 0); if (k.wgrad_chunks < 1 || k.wgrad_chunks > 512) k.wgrad_chunks = 0;
    k.wgrad_chunks_side = env_int("QPN_WGRAD_CHUNKS_SIDE", 0); if (k.wgrad_chunks_side < 1 || k.wgrad_chunks_side > 512) k.wgrad_chunks_side = 0;
    k.up_side = env_int("QPN_UP_SIDE", 1) != 0;
    k.reduce_early = env_int("QPN_REDUCE_EARLY", 1) != 0;
    k.wr_side = env_int("QPN_WR_SIDE", 1) != 0;
    k.post_fuse = env_int("QPN_POST_FUSE", 1) != 0;
    k.post_pair = env_int("QPN_POST_WGRAD_PAIR", 1) != 0;
    k.zero_in_post = env_int("QPN_ZERO_IN_POST", 1) != 0;
    k.post_wide = env_int("QPN_POST_WIDE", 1) != 0;
    k.xcd_swizzle = getenv("QPN_NO_XCD_SWIZZLE") == nullptr;
    k.ce_separate = getenv("QPN_CE_SEPARATE") != nullptr;
    k.event_fence = env_int("QPN_EVENT_FENCE", 0) == 1;
    k.aux_hoist = env_int("QPN_AUX_HOIST", 1) != 0;
    // (both opt-in: correct, measured slower than k_stack_fwd / k_stack_bwd on the batch-1 chunk -- MEASUREMENTS R6.2)
    k.stack_wave_fwd = env_int("QPN_STACK_WAVE_FWD", 0);      // 1: k_stack_fwd_t, 2: k_stack_fwd_h
    k.stack_wave_bwd = env_int("QPN_STACK_WAVE_BWD", 0) != 0;
    k.stack_waves = 4;
    k.test_stack_gives_up = false;
#ifdef QPN_TESTING
    k.test_stack_gives_up = env_int("QPN_TEST_STACK_GIVES_UP", 0) == 1;
#endif
}

int qpn_num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) { int n = 0; if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256; cached[dev] = n; }
    return cached[dev];
}

// ---- per-group timing
// Marks are HIP events on the stream a launch went to; a group's time is the sum, over its marks, of the time since the PREVIOUS mark on the same stream
// (group -1: a baseline only -- behind a cross-stream wait, so that the wait is nobody's time).  Serial mode (qpn_train_profile_begin) puts the whole
// step on one stream: a kernel's time alone.  Overlapped mode (qpn_train_profile_begin_overlapped) leaves the step as the timed loop runs it -- the skip /
// post-net weight gradients, the early reduction, the aux tail and dWr on the side stream next to the stack backward and dW1 -- and times every launch
// where it runs: what a kernel costs in the step, which is what the bench line's roofline.kernel is chosen by.
struct Prof { bool on = false; bool overlapped = false; std::vector<hipEvent_t> ev; std::vector<int> grp; std::vector<hipStream_t> st; size_t used = 0; };
static thread_local Prof g_prof;
bool qpn_prof_active() { return g_prof.on; }
bool qpn_prof_serial() { return g_prof.on && !g_prof.overlapped; }
void qpn_prof_mark(int group, hipStream_t stream) {
    Prof& P = g_prof;
    if (!P.on) return;
    if (P.used == P.ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; P.ev.push_back(e); P.grp.push_back(0); P.st.push_back(nullptr); }
    P.grp[P.used] = group; P.st[P.used] = stream;
    (void)hipEventRecord(P.ev[P.used++], stream);
}
extern "C" int qpn_train_profile_begin(qpn_handle* h, void* stream) {
    (void)h; g_prof.on = true; g_prof.overlapped = false; g_prof.used = 0;
    qpn_prof_mark(-1, (hipStream_t)stream);
    return QPN_OK;
}
extern "C" int qpn_train_profile_begin_overlapped(qpn_handle* h, void* stream) {
    (void)h; g_prof.on = true; g_prof.overlapped = true; g_prof.used = 0;
    qpn_prof_mark(-1, (hipStream_t)stream);
    return QPN_OK;
}
extern "C" int qpn_train_profile_mark(qpn_handle* h, int group, void* stream) {
    (void)h; qpn_prof_mark(group, (hipStream_t)stream); return QPN_OK;
}
extern "C" int qpn_train_profile_end(qpn_handle* h, float* h_ms, int n, void* stream) {
    Prof& P = g_prof;
    QPN_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (h && h->train && h->train->side) QPN_HIP(hipStreamSynchronize(h->train->side));
    for (int i = 0; i < n; ++i) h_ms[i] = 0.f;
    for (size_t i = 1; i < P.used; ++i) {
        if (P.grp[i] < 0 || P.grp[i] >= n) continue;
        size_t j = i;
        while (j > 0 && P.st[j - 1] != P.st[i]) --j;              // the previous mark on the same stream
        if (j == 0) continue;
        float ms = 0.f;
        QPN_HIP(hipEventElapsedTime(&ms, P.ev[j - 1], P.ev[i]));
        h_ms[P.grp[i]] += ms;
    }
    P.on = false;
    return QPN_OK;
}

__global__ void k_gather_f(const float* __restrict__ flat, const int* __restrict__ map, float* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { int m = map[i]; out[i] = m >= 0 ? flat[m] : 0.0f; }
}
// the per-step refresh of everything derived from the parameters, in ONE launch: weight blocks (fragment order or K-major), the
// transposed causal table, the packed bias sums; also clears the loss accumulator of the coming step
__global__ void k_refresh(const float* __restrict__ flat, const int* __restrict__ wmap, float* __restrict__ wout, int64_t nw,
                          const int* __restrict__ ctmap, float* __restrict__ ct, int64_t nct,
                          const int* __restrict__ bstart, const int* __restrict__ blist, float* __restrict__ bp, int nb,
                          int* __restrict__ status, double* __restrict__ loss) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nw) { const int m = wmap[i]; wout[i] = m >= 0 ? flat[m] : 0.0f; return; }
    i -= nw;
    if (i < nct) { ct[i] = flat[ctmap[i]]; return; }
    i -= nct;
    if (i < nb) { float a = 0.f; for (int j = bstart[i]; j < bstart[i + 1]; ++j) a += flat[blist[j]]; bp[i] = a; return; }
    i -= nb;
    // (the status word is STICKY: kernels OR into it, only qpn_train_status clears it when it reads it -- a check every N steps
    //  then still sees an out-of-range tap / target of any step in between; the reference asserts on every step)
    (void)status;
    if (i >= 16 && i < 16 + 64) loss[i - 16] = 0.0;
}
// B[k][n] (K x N valid) -> K-major block [Kp][Np], zero padded; returns the float offset
template <class F>
static long kmajor_pack(std::vector<int>& map, int K, int Kp, int N, int Np, F src) {
    const long off = (long)map.size();
    map.resize(map.size() + (size_t)Kp * Np, -1);
    int* m = map.data() + off;
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) m[(size_t)k * Np + n] = (int)src(k, n);
    return off;
}
static inline int pad_to(int v, int q) { return (v + q - 1) / q * q; }

// B[k][n] (K x N, both multiples of 16) -> fragment order; returns float4 offset
template <class F>
static int frag_pack(std::vector<int>& map, int K, int N, F src) {
    const int NT = N / 16, off4 = (int)(map.size() / 4);
    map.resize(map.size() + (size_t)K * N);
    int* m = map.data() + (size_t)off4 * 4;
    for (int ks4 = 0; ks4 < K / 16; ++ks4)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    int k = 16 * ks4 + 4 * e + (lane >> 4), n = 16 * nt + (lane & 15);
                    m[(((size_t)ks4 * NT + nt) * 64 + lane) * 4 + e] = (int)src(k, n);
                }
    return off4;
}

// W[m][k] (M x K, both multiples of 16) as the A operand of the TRANSPOSED products of train_stackw.hip: word ((s4 * M/16 + mt) * 64 + lane), element e
// = W[16 mt + (lane & 15)][16 s4 + 4 (lane >> 4) + e] -- the k order in which a lane of the tile's B operand holds its row (four consecutive
// channels of every sixteen per lane group).  Returns the float4 offset.
template <class F>
static int fragA_pack(std::vector<int>& map, int M, int K, F src) {
    const int MT = M / 16, off4 = (int)(map.size() / 4);
    map.resize(map.size() + (size_t)M * K);
    int* mp = map.data() + (size_t)off4 * 4;
    for (int s4 = 0; s4 < K / 16; ++s4)
        for (int mt = 0; mt < MT; ++mt)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int m = 16 * mt + (lane & 15), k = 16 * s4 + 4 * (lane >> 4) + e;
                    mp[(((size_t)s4 * MT + mt) * 64 + lane) * 4 + e] = (int)src(m, k);
                }
    return off4;
}

static int train_init(qpn_handle* h) {
    if (h->train) return QPN_OK;
    const Geom& g = h->g;
    const int C = g.C, S = g.S, Q = g.Q, A = g.A, L = g.L;
    if (L > TR_MAXL) { qpn_set_error("training kernels support up to %d residual layers", TR_MAXL); return QPN_EINVAL; }
    if (C % 16 || S % 16 || Q % 16) { qpn_set_error("training kernels need n_resch, n_skipch, n_quantize multiples of 16"); return QPN_EINVAL; }
    TrainState* t = new TrainState();
    memset(&t->tp, 0, sizeof(t->tp)); memset(&t->bw, 0, sizeof(t->bw));
    t->d_wmap = nullptr; t->d_wp = nullptr; t->d_bstart = t->d_blist = nullptr; t->d_bp = nullptr;
    t->d_gdst = t->d_gdst_list = t->d_gzero = nullptr; t->n_gzero = 0;
    t->d_ws = nullptr; t->ws_cap = 0; t->d_tap = nullptr; t->tap_cap = 0; t->d_status = nullptr; t->d_loss = nullptr; t->fwd_valid = false; t->loss_clear = false;
    t->generation = 0; t->side = nullptr; t->ev_fork = t->ev_join = t->ev_mid = nullptr; t->ev_early = nullptr; t->early_recorded = 0; t->early_first = -1;
    t->d_sq = nullptr; t->sq_pos_cap = 0; t->sq_per_dir = 0; t->bwd_generation = -1; t->stack_disabled = false; memset(&t->sqf, 0, sizeof(t->sqf)); memset(&t->sqb, 0, sizeof(t->sqb));
    qpn_train_knobs_parse(t->knobs);
    memset(&t->slabs, 0, sizeof(t->slabs));
    t->use_gemm = (C > 128 || getenv("QPN_TRAIN_GEMM")) && !getenv("QPN_TRAIN_TILES");
    t->d_gmap = nullptr; t->d_gwp = nullptr; t->d_ctmap = nullptr; t->d_ct = nullptr; memset(&t->gm, 0, sizeof(t->gm));
    if (t->use_gemm && (C % 32 || S % 32 || Q % 32)) { qpn_set_error("the GEMM training path needs n_resch, n_skipch, n_quantize multiples of 32"); delete t; return QPN_EINVAL; }
    TrainParams& p = t->tp;
    const int Ap = (A + 3) / 4 * 4;               // aux columns padded to the MFMA k-step
    // the auxiliary 1x1 at frame rate (TrainParams::hoist): the n_resch-64 register-resident kernels, at most two frames per 16-row tile
    t->hoist = !t->use_gemm && C == 64 && g.U >= 16 && A <= 64 && t->knobs.aux_hoist && t->knobs.persist_fwd && t->knobs.persist_bwd;
    p.hoist = t->hoist ? 1 : 0;
    p.C = C; p.S = S; p.Q = Q; p.A = A; p.Ap = Ap; p.L = L; p.U = g.U;
    p.Kt = t->hoist ? 2 * C : 2 * C + Ap; p.Ktp = (p.Kt + 15) / 16 * 16; p.LC = L * C;
    p.causal_w = g.causal_w; p.causal_b = g.causal_b; p.up_w = g.up_w; p.up_b = g.up_b;
    std::vector<int>& map = t->h_wmap;
    std::vector<std::vector<int>> biases;      // packed bias i <- list of flat indices
    auto add_bias = [&](int n, auto f) { int off = (int)biases.size(); for (int i = 0; i < n; ++i) biases.push_back(f(i)); return off; };
    const int Ktp = p.Ktp;
    int n_adapt = 0;
    for (int l = 0; l < L; ++l) {
        const LayerGeom y = g.layers[l];
        TrLayer& ly = p.layers[l];
        ly.adaptive = y.adaptive; ly.dilation = y.dilation;
        ly.tap_off = y.adaptive ? n_adapt++ : -1;     // scaled by B*N1 per call
        auto w1src = [&](int k, int n) -> int64_t {   // B[k][n] of z = [x_cur | x_past | aux] . W1
            const int half = n / C, r = n % C;
            if (k < C) return y.adaptive ? (half ? y.wT : y.wS) + (int64_t)r * C + k : (half ? y.wT : y.wS) + ((int64_t)r * C + k) * 2 + 1;
            if (k < 2 * C) { const int kk = k - C; return y.adaptive ? (half ? y.wTP : y.wSP) + (int64_t)r * C + kk : (half ? y.wT : y.wS) + ((int64_t)r * C + kk) * 2; }
            if (k < 2 * C + A) return (half ? y.auxT : y.auxS) + (int64_t)r * A + (k - 2 * C);
            return -1;
        };
        ly.w1_f4 = frag_pack(map, Ktp, 2 * C, w1src);
        ly.w1t_f4 = frag_pack(map, 2 * C, Ktp, [&](int k, int n) { return w1src(n, k); });
        ly.wr_f4 = frag_pack(map, C, C, [&](int k, int n) { return y.res + (int64_t)n * C + k; });
        ly.wrt_f4 = frag_pack(map, C, C, [&](int k, int n) { return y.res + (int64_t)k * C + n; });
        ly.bias1 = add_bias(2 * C, [&](int n) {
            const int half = n / C, r = n % C;
            std::vector<int> v{(int)((half ? y.bT : y.bS) + r), (int)((half ? y.auxTb : y.auxSb) + r)};
            if (y.adaptive) v.push_back((int)((half ? y.bTP : y.bSP) + r));
            return v; });
        ly.biasr = add_bias(C, [&](int n) { return std::vector<int>{(int)(y.resb + n)}; });
    }
    p.w1q_f4 = p.wrq_f4 = p.w1p_f4 = p.wrp_f4 = -1;
    if (t->hoist) {      // images of the wave-per-tile stack kernels (train_stackw.hip): all layers of a kind contiguous, 4096 / 1024 float4 words each
        auto w1of = [&](const LayerGeom& y, int k, int n) -> int64_t {      // the same B[k][n] of z = [x_cur | x_past] . W1 as w1src above (k < 2C)
            const int half = n / C, r = n % C;
            if (k < C) return y.adaptive ? (half ? y.wT : y.wS) + (int64_t)r * C + k : (half ? y.wT : y.wS) + ((int64_t)r * C + k) * 2 + 1;
            const int kk = k - C; return y.adaptive ? (half ? y.wTP : y.wSP) + (int64_t)r * C + kk : (half ? y.wT : y.wS) + ((int64_t)r * C + kk) * 2;
        };
        for (int l = 0; l < L; ++l) {      // backward: out^T[n_in][row] = sum_kz W1[n_in][kz] dZ[row][kz]  -> A'[m = n_in][k = kz]
            const LayerGeom y = g.layers[l];
            const int o = fragA_pack(map, 2 * C, 2 * C, [&](int m, int k) { return w1of(y, m, k); });
            if (l == 0) p.w1q_f4 = o;
        }
        for (int l = 0; l < L; ++l) {      // backward: dg^T[c][row] = sum_j res[j][c] dXout[row][j]  -> A'[m = c][k = j]
            const LayerGeom y = g.layers[l];
            const int o = fragA_pack(map, C, C, [&](int m, int k) { return y.res + (int64_t)k * C + m; });
            if (l == 0) p.wrq_f4 = o;
        }
        for (int l = 0; l < L; ++l) {      // forward: z^T[nz][row] = sum_k W1[k][nz] [x_cur | x_past][row][k]  -> A'[m = nz][k = k_in]
            const LayerGeom y = g.layers[l];
            const int o = fragA_pack(map, 2 * C, 2 * C, [&](int m, int k) { return w1of(y, k, m); });
            if (l == 0) p.w1p_f4 = o;
        }
        for (int l = 0; l < L; ++l) {      // forward: out^T[o][row] = sum_c res[o][c] g[row][c]  -> A'[m = o][k = c]
            const LayerGeom y = g.layers[l];
            const int o = fragA_pack(map, C, C, [&](int m, int k) { return y.res + (int64_t)m * C + k; });
            if (l == 0) p.wrp_f4 = o;
        }
    }
    p.ws_f4 = frag_pack(map, L * C, S, [&](int k, int n) { return g.layers[k / C].skip + (int64_t)n * C + (k % C); });
    p.wst_f4 = frag_pack(map, S, L * C, [&](int k, int n) { return g.layers[n / C].skip + (int64_t)k * C + (n % C); });
    p.p1_f4 = frag_pack(map, S, S, [&](int k, int n) { return g.post1_w + (int64_t)n * S + k; });
    p.p1t_f4 = frag_pack(map, S, S, [&](int k, int n) { return g.post1_w + (int64_t)k * S + n; });
    p.p2_f4 = frag_pack(map, S, Q, [&](int k, int n) { return g.post2_w + (int64_t)n * S + k; });
    p.p2t_f4 = frag_pack(map, Q, S, [&](int k, int n) { return g.post2_w + (int64_t)k * S + n; });
    p.bias_s = add_bias(S, [&](int n) { std::vector<int> v; for (int l = 0; l < L; ++l) v.push_back((int)(g.layers[l].skipb + n)); return v; });
    p.bias_p1 = add_bias(S, [&](int n) { return std::vector<int>{(int)(g.post1_b + n)}; });
    p.bias_p2 = add_bias(Q, [&](int n) { return std::vector<int>{(int)(g.post2_b + n)}; });
    if (t->use_gemm) {
        TrainGemm& gm = t->gm;
        std::vector<int>& gmap = t->h_gmap;
        const int Apad = pad_to(Ap, 32);
        gm.K1 = 2 * C + Apad; gm.N1g = pad_to(C, 64) * 2; gm.Cg = pad_to(C, 128); gm.Sg = pad_to(S, 128); gm.Qg = pad_to(Q, 128);
        gm.LCg = pad_to(L * C, 128); gm.Ktg = pad_to(2 * C + Ap, 128);
        for (int l = 0; l < L; ++l) {
            const LayerGeom y = g.layers[l];
            auto w1src = [&](int k, int n) -> int64_t {        // z column n (natural: half*C + r), input column k of [x_cur | x_past | aux]
                const int half = n / C, r = n % C;
                if (k < C) return y.adaptive ? (half ? y.wT : y.wS) + (int64_t)r * C + k : (half ? y.wT : y.wS) + ((int64_t)r * C + k) * 2 + 1;
                if (k < 2 * C) { const int kk = k - C; return y.adaptive ? (half ? y.wTP : y.wSP) + (int64_t)r * C + kk : (half ? y.wT : y.wS) + ((int64_t)r * C + kk) * 2; }
                if (k < 2 * C + A) return (half ? y.auxT : y.auxS) + (int64_t)r * A + (k - 2 * C);
                return -1;
            };
            // gate GEMM: rows k = [x_cur C | x_past C | aux (n_aux, padded to 32)], columns in 128-wide tiles [sigma c0..c0+63 | tanh c0..c0+63]
            gm.w1[l] = kmajor_pack(gmap, 2 * C + A, gm.K1, gm.N1g, gm.N1g, [&](int k, int q) -> int64_t {
                const int tile = q / 128, w = q % 128, half = w / 64, c = 64 * tile + w % 64;
                return c < C ? w1src(k, half * C + c) : -1; });
            gm.w1t[l] = kmajor_pack(gmap, 2 * C, 2 * C, 2 * C + A, gm.Ktg, [&](int k, int n) { return w1src(n, k); });
            gm.wr[l] = kmajor_pack(gmap, C, C, C, gm.Cg, [&](int k, int n) { return y.res + (int64_t)n * C + k; });
            gm.wrt[l] = kmajor_pack(gmap, C, C, C, gm.Cg, [&](int k, int n) { return y.res + (int64_t)k * C + n; });
        }
        gm.ws = kmajor_pack(gmap, L * C, L * C, S, gm.Sg, [&](int k, int n) { return g.layers[k / C].skip + (int64_t)n * C + (k % C); });
        gm.wst = kmajor_pack(gmap, S, S, L * C, gm.LCg, [&](int k, int n) { return g.layers[n / C].skip + (int64_t)k * C + (n % C); });
        gm.p1 = kmajor_pack(gmap, S, S, S, gm.Sg, [&](int k, int n) { return g.post1_w + (int64_t)n * S + k; });
        gm.p1t = kmajor_pack(gmap, S, S, S, gm.Sg, [&](int k, int n) { return g.post1_w + (int64_t)k * S + n; });
        gm.p2 = kmajor_pack(gmap, S, S, Q, gm.Qg, [&](int k, int n) { return g.post2_w + (int64_t)n * S + k; });
        gm.p2t = kmajor_pack(gmap, Q, Q, S, gm.Sg, [&](int k, int n) { return g.post2_w + (int64_t)k * S + n; });
    }
    memset(&t->ag, 0, sizeof(t->ag));
    for (int l = 0; l < L; ++l) { t->ag.auxS[l] = (int)g.layers[l].auxS; t->ag.auxT[l] = (int)g.layers[l].auxT; t->ag.bias1[l] = p.layers[l].bias1; }
    t->n_bias = (int)biases.size();
    t->h_bstart.assign(1, 0);
    for (auto& v : biases) { for (int x : v) t->h_blist.push_back(x); t->h_bstart.push_back((int)t->h_blist.size()); }

    // ---- weight-grad staging space ("slab") and the flat-grad gather map
    TrainBwd& bw = t->bw;
    TrainSlabs& sl = t->slabs;
    bw.sl = &t->slabs;
    int go = 0;
    auto gtake = [&](int n) { int r = go; go += (n + 63) & ~63; return r; };
    std::vector<int>& gs = t->h_gsrc;          // flat-grad entry <- slab element; -1: no slab feeds it (zeroed, then written by a kernel of its own); -2: k_aux_tail stores it
    gs.assign(g.n_params, -1);
    for (int l = 0; l < L; ++l) {
        sl.g_w1[l] = gtake(2 * C * Ktp);   // dW1[n][k] (n = z row, k = A-tile column), row-major [2C][Ktp]
        sl.g_b1[l] = gtake(2 * C);
        sl.g_wr[l] = gtake(C * C);         // dWr[o][c]
        sl.g_br[l] = gtake(C);
    }
    sl.g_early0 = go;
    for (int l = 0; l < L; ++l) sl.g_ws[l] = gtake(S * C);         // dWs_l[s][c]
    for (int l = 0; l < L; ++l) {
        const LayerGeom y = g.layers[l];
        // (slab order: first the blocks that need the layer backward -- dW1, dWr of every layer -- then, contiguous, the ones the side
        //  stream finishes early: skip 1x1 of every layer, skip bias, post-net; that range is reduced early as well, under the layer backward)
        for (int n = 0; n < 2 * C; ++n) {
            const int half = n / C, r = n % C;
            if (t->hoist) for (int a = 0; a < A; ++a) gs[(half ? y.auxT : y.auxS) + (int64_t)r * A + a] = -2;
            for (int k = 0; k < (t->hoist ? 2 * C : 2 * C + A); ++k) {
                int64_t dst;
                if (k < C) dst = y.adaptive ? (half ? y.wT : y.wS) + (int64_t)r * C + k : (half ? y.wT : y.wS) + ((int64_t)r * C + k) * 2 + 1;
                else if (k < 2 * C) { const int kk = k - C; dst = y.adaptive ? (half ? y.wTP : y.wSP) + (int64_t)r * C + kk : (half ? y.wT : y.wS) + ((int64_t)r * C + kk) * 2; }
                else dst = (half ? y.auxT : y.auxS) + (int64_t)r * A + (k - 2 * C);
                gs[dst] = sl.g_w1[l] + n * Ktp + k;
            }
            gs[(half ? y.bT : y.bS) + r] = sl.g_b1[l] + n;
            gs[(half ? y.auxTb : y.auxSb) + r] = sl.g_b1[l] + n;
            if (y.adaptive) gs[(half ? y.bTP : y.bSP) + r] = sl.g_b1[l] + n;
        }
        for (int o = 0; o < C; ++o) { for (int c = 0; c < C; ++c) gs[y.res + (int64_t)o * C + c] = sl.g_wr[l] + o * C + c; gs[y.resb + o] = sl.g_br[l] + o; }
        for (int s = 0; s < S; ++s) for (int c = 0; c < C; ++c) gs[y.skip + (int64_t)s * C + c] = sl.g_ws[l] + s * C + c;
    }
    sl.g_bs = gtake(S);
    for (int l = 0; l < L; ++l) for (int s = 0; s < S; ++s) gs[g.layers[l].skipb + s] = sl.g_bs + s;
    sl.g_p1 = gtake(S * S); sl.g_bp1 = gtake(S); sl.g_p2 = gtake(Q * S); sl.g_bp2 = gtake(Q);
    sl.g_early1 = go;
    // the post-net blocks close the flat parameter order (state_dict order, qpnet.py): with the trailer behind them one contiguous early bucket
    t->early_first = (g.post1_b == g.post1_w + (int64_t)S * S && g.post2_w == g.post1_b + S && g.post2_b == g.post2_w + (int64_t)Q * S && g.post2_b + Q == g.n_params) ? g.post1_w : -1;
    for (int o = 0; o < S; ++o) { for (int s = 0; s < S; ++s) gs[g.post1_w + (int64_t)o * S + s] = sl.g_p1 + o * S + s; gs[g.post1_b + o] = sl.g_bp1 + o; }
    for (int q = 0; q < Q; ++q) { for (int s = 0; s < S; ++s) gs[g.post2_w + (int64_t)q * S + s] = sl.g_p2 + q * S + s; gs[g.post2_b + q] = sl.g_bp2 + q; }
    // causal conv table: dW[c][q][tap] = sum_t dX0[t][c] * onehot(x[t-1+tap])[q] is one more time contraction (k_wgrad3 mode 4)
    // when its tiles fit (C = 64, Q a multiple of 128); otherwise the LDS-histogram kernel owns it (gs stays -1)
    sl.g_cw = sl.g_cb = -1;
    if (C == 64 && Q % 128 == 0 && !t->use_gemm) {
        sl.g_cw = gtake(2 * C * Q); sl.g_cb = gtake(C);
        for (int c = 0; c < C; ++c) {
            for (int q = 0; q < Q; ++q) for (int tp = 0; tp < 2; ++tp) gs[g.causal_w + ((int64_t)c * Q + q) * 2 + tp] = sl.g_cw + tp * C * Q + c * Q + q;
            gs[g.causal_b + c] = sl.g_cb + c;
        }
    }
    bw.gstage = go; bw.nch = t->use_gemm ? 4 : 64; bw.n_params = g.n_params;
    if (t->knobs.wgrad_chunks > 0) bw.nch = t->knobs.wgrad_chunks;      // tuning knob: time chunks (= partial slabs)
    // the upsampling kernel's gradient is written by a dedicated kernel (gs stays -1: k_up_bwd adds onto the zeroed entries; hoist: k_aux_tail stores them)
    if (t->hoist) for (int j = 0; j < g.U; ++j) gs[g.up_w + j] = -2;      // (the upsampling bias stays -1: zeroed by the reduction, k_aux_tail adds onto it)

    const size_t nmap = t->use_gemm ? 4 : map.size();      // (the GEMM path keeps its own K-major blocks)
    QPN_HIP(hipMalloc(&t->d_wmap, nmap * sizeof(int)));
    QPN_HIP(hipMalloc(&t->d_wp, nmap * sizeof(float)));
    QPN_HIP(hipMalloc(&t->d_bstart, t->h_bstart.size() * sizeof(int)));
    QPN_HIP(hipMalloc(&t->d_blist, t->h_blist.size() * sizeof(int)));
    QPN_HIP(hipMalloc(&t->d_bp, (size_t)t->n_bias * sizeof(float)));
    QPN_HIP(hipMalloc(&t->d_status, 64));
    QPN_HIP(hipMemset(t->d_status, 0, 64));
    t->last_stream = nullptr;
    t->h_status_pinned = nullptr; t->ev_status[0] = t->ev_status[1] = nullptr; t->status_pending[0] = t->status_pending[1] = false; t->status_newest = 0;
    QPN_HIP(hipHostMalloc((void**)&t->h_status_pinned, 64, hipHostMallocDefault));
    t->h_loss_pinned = nullptr; t->ev_loss[0] = t->ev_loss[1] = nullptr; t->loss_pending[0] = t->loss_pending[1] = false; t->loss_newest = 0;
    QPN_HIP(hipHostMalloc((void**)&t->h_loss_pinned, 2 * 64 * sizeof(double), hipHostMallocDefault));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_loss[0], hipEventDisableTiming));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_loss[1], hipEventDisableTiming));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_status[0], hipEventDisableTiming));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_status[1], hipEventDisableTiming));
    QPN_HIP(hipMalloc(&t->d_loss, 64 * sizeof(double)));
    if (!t->use_gemm) QPN_HIP(hipMemcpy(t->d_wmap, map.data(), nmap * sizeof(int), hipMemcpyHostToDevice));
    QPN_HIP(hipMemcpy(t->d_bstart, t->h_bstart.data(), t->h_bstart.size() * sizeof(int), hipMemcpyHostToDevice));
    QPN_HIP(hipMemcpy(t->d_blist, t->h_blist.data(), t->h_blist.size() * sizeof(int), hipMemcpyHostToDevice));
    {   // inverse of gsrc for the slab-order reduction, CSR: a slab element feeds one parameter, or several (a gate's conv / aux /
        // past-tap conv biases share one gradient, every layer's skip bias shares one); entries nothing feeds are listed for zeroing
        std::vector<int> start((size_t)go + 1, 0), list, gz;
        for (int64_t i = 0; i < g.n_params; ++i) { if (gs[i] >= 0) ++start[(size_t)gs[i] + 1]; else if (gs[i] == -1) gz.push_back((int)i); }
        for (int k = 0; k < go; ++k) start[(size_t)k + 1] += start[k];
        list.resize((size_t)start[go] + 1);
        std::vector<int> fill(start.begin(), start.end() - 1);
        for (int64_t i = 0; i < g.n_params; ++i) if (gs[i] >= 0) list[(size_t)fill[gs[i]]++] = (int)i;
        QPN_HIP(hipMalloc(&t->d_gdst, start.size() * sizeof(int)));
        QPN_HIP(hipMemcpy(t->d_gdst, start.data(), start.size() * sizeof(int), hipMemcpyHostToDevice));
        QPN_HIP(hipMalloc(&t->d_gdst_list, list.size() * sizeof(int)));
        QPN_HIP(hipMemcpy(t->d_gdst_list, list.data(), list.size() * sizeof(int), hipMemcpyHostToDevice));
        t->n_gzero = (int)gz.size();
        QPN_HIP(hipMalloc(&t->d_gzero, (gz.size() + 1) * sizeof(int)));
        if (!gz.empty()) QPN_HIP(hipMemcpy(t->d_gzero, gz.data(), gz.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    if (t->use_gemm) {
        QPN_HIP(hipMalloc(&t->d_gmap, t->h_gmap.size() * sizeof(int)));
        QPN_HIP(hipMalloc(&t->d_gwp, t->h_gmap.size() * sizeof(float)));
        QPN_HIP(hipMemcpy(t->d_gmap, t->h_gmap.data(), t->h_gmap.size() * sizeof(int), hipMemcpyHostToDevice));
        t->gm.wp = t->d_gwp;
        std::vector<int>().swap(t->h_wmap);                 // the fragment-ordered blocks of the tile kernels are not used on this path
    }
    {
        std::vector<int> ctm((size_t)2 * Q * C);
        for (int tp = 0; tp < 2; ++tp) for (int q = 0; q < Q; ++q) for (int c = 0; c < C; ++c)
            ctm[((size_t)tp * Q + q) * C + c] = (int)(g.causal_w + ((int64_t)c * Q + q) * 2 + tp);
        QPN_HIP(hipMalloc(&t->d_ctmap, ctm.size() * sizeof(int)));
        QPN_HIP(hipMalloc(&t->d_ct, ctm.size() * sizeof(float)));
        QPN_HIP(hipMemcpy(t->d_ctmap, ctm.data(), ctm.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    QPN_HIP(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));          // on the handle's device (current at this call)
    // the fork / join events only order kernels of THIS device's streams: no system-scope fence (cache write-back for the host) at each record
    // (QPN_EVENT_FENCE=1 restores it)
    const unsigned evf = hipEventDisableTiming | (t->knobs.event_fence ? 0u : (unsigned)hipEventDisableSystemFence);
    QPN_HIP(hipEventCreateWithFlags(&t->ev_fork, evf));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_join, evf));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_mid, evf));
    QPN_HIP(hipEventCreateWithFlags(&t->ev_early, evf));
    h->train = t;
    return QPN_OK;
}

void qpn_train_destroy(TrainState* t) {
    if (!t) return;
    void* bufs[] = {t->d_wmap, t->d_wp, t->d_bstart, t->d_blist, t->d_bp, t->d_gdst, t->d_gdst_list, t->d_gzero, t->d_ws, t->d_tap, t->d_status, t->d_loss, t->d_gmap, t->d_gwp, t->d_ctmap, t->d_ct, t->d_sq};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (t->h_status_pinned) (void)hipHostFree(t->h_status_pinned);
    if (t->h_loss_pinned) (void)hipHostFree(t->h_loss_pinned);
    for (int i = 0; i < 2; ++i) if (t->ev_loss[i]) (void)hipEventDestroy(t->ev_loss[i]);
    for (int i = 0; i < 2; ++i) if (t->ev_status[i]) (void)hipEventDestroy(t->ev_status[i]);
    if (t->side) (void)hipStreamDestroy(t->side);
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    if (t->ev_mid) (void)hipEventDestroy(t->ev_mid);
    if (t->ev_early) (void)hipEventDestroy(t->ev_early);
    delete t;
}

static int need_dev(qpn_handle* h) {
    if (!h) { qpn_set_error("null handle"); return QPN_EINVAL; }
    if (h->device < 0) { qpn_set_error("no HIP device: libqpnet_hip has no CPU fallback"); return QPN_ENODEV; }
    return QPN_OK;
}

// targets == nullptr: plain forward.  Otherwise the mean cross entropy (and d_dlogits) is computed too: inside the post-net kernel
// when the tile path can hold a row of logits in LDS, by a k_ce launch behind the forward otherwise.
// fuse_bwd: the caller runs this forward's backward next, with the same d_dlogits, whatever the loss is (qpn_train_step): the post-net's backward may run inside the forward's launch
static int train_forward_impl(qpn_handle* h, const float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                              const int64_t* d_x, const float* d_h, const float* d_dfac, float* d_logits,
                              const int64_t* d_targets, int64_t tgt_stride, float* d_dlogits, int want_logits, void* stream_, bool fuse_bwd = false) {
    int rc = need_dev(h); if (rc) return rc;
    rc = train_init(h); if (rc) return rc;
    TrainState* t = h->train;
    hipStream_t stream = (hipStream_t)stream_;
    const Geom& g = h->g;
    if (!d_flat || !d_x || !d_h || !d_dfac || !d_logits || B < 1 || BL < 1 || maxd < 1) { qpn_set_error("bad train_forward arguments"); return QPN_EINVAL; }
    if (d_targets && tgt_stride < BL) { qpn_set_error("bad train_forward_loss arguments: target rows shorter than batch_length"); return QPN_EINVAL; }
    const int64_t N0 = (int64_t)g.recA * maxd + g.recF + 1 + BL;           // qpnet.py:254-262
    if (N0 > T || N0 - 1 > Td || N0 - 1 > (g.U > 0 ? F * g.U : F)) {
        qpn_set_error("chunk too short: need receptive field (%lld) + batch_length (%d) = %lld samples, have x:%lld d:%lld h_up:%lld",
                      (long long)(N0 - BL), BL, (long long)N0, (long long)T, (long long)Td, (long long)(g.U > 0 ? F * g.U : F));
        return QPN_EINVAL;
    }
    if (T >= ((int64_t)1 << 31) || Td >= ((int64_t)1 << 31) || F * (g.U > 0 ? g.U : 1) >= ((int64_t)1 << 31) || (int64_t)B * N0 >= ((int64_t)1 << 31)) {
        qpn_set_error("chunk too long: the training kernels index rows with 32 bits"); return QPN_EINVAL;
    }
    TrainParams& p = t->tp;
    p.B = B; p.T = (int)T; p.F = (int)F; p.Td = (int)Td; p.BL = BL; p.N0 = (int)N0; p.N1 = (int)N0 - 1; p.maxd = maxd;
    const int C = g.C, S = g.S, L = g.L, N1 = p.N1;
    int s = 0, nA = 0;
    for (int l = 0; l < L; ++l) {
        TrLayer& ly = p.layers[l];
        ly.s_in = s; s += ly.adaptive ? ly.dilation * maxd : ly.dilation; ly.s_out = s;
        ly.tap_off = (nA++) * B * N1;          // every layer gets a tap table (fixed layers: n - dilation), so the kernels load taps unconditionally
        t->ag.s_out[l] = ly.s_out;
    }
    if (t->hoist) {
        if (N1 >= (1 << 24)) { qpn_set_error("chunk too long for the register-resident layer kernels (%d rows; QPN_AUX_HOIST=0 QPN_LAYER_PERSIST=0 QPN_LAYER_BWD_PERSIST=0 selects the tile-per-workgroup kernels)", N1); return QPN_EINVAL; }
        p.ffirst = (int)(((int64_t)F * g.U - N1) / g.U); p.nfr = (int)F - p.ffirst;
    } else { p.ffirst = 0; p.nfr = 0; }
    // ---- carve the arena
    const size_t nPA = t->hoist ? (size_t)L * B * (p.nfr + 1) * 2 * C : 0, nWJ = t->hoist ? (size_t)2 * (N1 + 16) : 0, nGW = t->hoist ? (size_t)L * B * N1 : 0;
    const size_t nX = (size_t)(L + 1) * B * N1 * C, nG = (size_t)L * B * N1 * C, nH = t->hoist ? 64 : (size_t)B * N1 * p.Ap, nS = (size_t)B * BL * S;
    TrainBwd& bw = t->bw;
    const size_t nDX = (size_t)B * N1 * C, nDZ = (size_t)B * N1 * 2 * C, nDGS = (size_t)B * BL * L * C, nSlab = (size_t)bw.nch * bw.gstage;
    const size_t nXC = (size_t)B * (N1 + 1);
    const size_t nScr = (size_t)B * 1024 * 2 * 128;
    size_t need = nScr + nX + 2 * nG + nH + 2 * (nS + 96 * (size_t)S + 64) + 2 * (size_t)(L + 1) * nDX + (size_t)L * nDZ + 2 * nS + nDGS + nH + nSlab + nXC + (t->use_gemm ? nG : 0) + 2 * nPA + nWJ + nGW + (size_t)L * TR_EB_SLOTS * 2 * C + 8192;
    if (need > t->ws_cap) {
        if (t->d_ws) (void)hipFree(t->d_ws);
        t->d_ws = nullptr; t->ws_cap = 0;
        hipError_t e = hipMalloc(&t->d_ws, need * sizeof(float));
        if (e != hipSuccess) { qpn_set_error("hipMalloc(%zu MiB) for the training workspace failed", need * 4 >> 20); return QPN_ENOMEM; }
        t->ws_cap = need;
    }
    const size_t ntap = (size_t)std::max(nA, 1) * B * N1;
    if (ntap > t->tap_cap) {
        if (t->d_tap) (void)hipFree(t->d_tap);
        t->d_tap = nullptr; t->tap_cap = 0;
        QPN_HIP(hipMalloc(&t->d_tap, ntap * sizeof(int)));
        t->tap_cap = ntap;
    }
    float* w = t->d_ws;
    auto carve = [&](size_t n) { float* r = w; w += (n + 63) & ~(size_t)63; return r; };
    p.X = carve(nX); p.SG = carve(nG); p.TH = carve(nG); p.HUP = carve(nH); p.S0 = carve(nS + 96 * (size_t)S); p.Y0 = carve(nS + 96 * (size_t)S);       // + one (80-row) tile of rows: the post-net backward prefetches its masks unclamped
    bw.DXA[0] = carve((size_t)(L + 1) * nDX); bw.DXB[0] = carve((size_t)(L + 1) * nDX); bw.DXA[1] = bw.DXB[1] = nullptr;   // DXB directly follows DXA (one memset)
    bw.DZ = carve((size_t)L * nDZ); bw.DS0 = carve(nS); bw.DY0 = carve(nS); bw.DGS = carve(nDGS); bw.DHUP = carve(nH); bw.slab = carve(nSlab);
    p.XC = (int*)carve(nXC);
    p.scratch_rows = carve(nScr);
    if (t->use_gemm) t->gm.G = carve(nG);
    p.PA = nullptr; p.WJ = nullptr; bw.DPA = nullptr; bw.GW = nullptr;
    bw.EB = nullptr;
    if (t->hoist) { p.PA = carve(nPA); bw.DPA = carve(nPA + (size_t)L * TR_EB_SLOTS * 2 * C); bw.EB = bw.DPA + nPA; p.WJ = (float2*)carve(nWJ); bw.GW = carve(nGW); }
    p.TAP = t->d_tap; p.status = t->d_status;
    t->last_stream = stream;
    {   // stack work queues (train_stack.hip): a flag word per (layer, batch item, 16-row tile) and direction, compared with a per-forward epoch
        // (zeroed only when (re)allocated), and the tile tables k_train_prep writes.  The regions keep their places for the life of an
        // allocation (sized for 1.5x the positions that forced it): a table word of an earlier step must never be read as a flag
        qpn_stack_fill(p);
        if ((size_t)p.qtotal > t->sq_pos_cap) {
            if (t->d_sq) (void)hipFree(t->d_sq);
            t->d_sq = nullptr; t->sq_pos_cap = 0;
            const size_t cap = (size_t)p.qtotal + (size_t)p.qtotal / 2 + 64;
            const size_t per_dir = (cap / 32 + 2) * 1056;              // (sq_fidx: 32 flags per 128-byte line, lines 4224 bytes apart)
            const size_t nsq = TR_QHDR_WORDS + 2 * per_dir + 2 * (cap + 1) * 8;      // + two int4 per position and direction
            QPN_HIP(hipMalloc(&t->d_sq, nsq * sizeof(unsigned)));
            QPN_HIP(hipMemsetAsync(t->d_sq, 0, nsq * sizeof(unsigned), stream));
            t->sq_pos_cap = cap; t->sq_per_dir = per_dir;
        }
        // A step captured into a hipGraph replays its launches with the arguments of the capture: the queues' epoch -- a launch argument, a new one
        // per forward, which is what lets the flags go unzeroed -- would repeat, and every flag of the previous replay would read as published.
        // While the stream is capturing the stack therefore runs as a launch per layer.
        hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(stream, &cap_st);
        const bool no_queue = t->stack_disabled || cap_st != hipStreamCaptureStatusNone;
        p.qctl = no_queue ? nullptr : t->d_sq;                   // (nullptr: no queue launches, no tables)
        p.qtab = no_queue ? nullptr : (int4*)(t->d_sq + TR_QHDR_WORDS + 2 * t->sq_per_dir);
        p.qtab_b = no_queue ? nullptr : p.qtab + 2 * (t->sq_pos_cap + 1);
        const unsigned epoch = (unsigned)((t->generation + 1) % 0xFFFFFFFFll) + 1u;      // (generation is bumped below; never 0)
        StackQ& f = t->sqf; StackQ& bq = t->sqb;
        f.flags = t->d_sq + TR_QHDR_WORDS; f.head = t->d_sq + 1024; f.abort = t->d_sq + 1; f.stats = t->d_sq + 4; f.epoch = epoch; f.total = p.qtotal; f.tab = p.qtab;
        bq.flags = f.flags + t->sq_per_dir; bq.head = f.head + 8 * TR_QHEAD_STRIDE; bq.abort = f.abort; bq.stats = t->d_sq + 8; bq.epoch = epoch; bq.total = p.qtotal; bq.tab = p.qtab_b;
    }
    p.flat = d_flat; p.wp = (const float4*)t->d_wp; p.bp = t->d_bp; p.x = d_x; p.h = d_h; p.d = d_dfac; p.logits = d_logits;
    // ---- refresh the fragment-ordered weights / packed biases from the current parameters
    {
        const int* wmap = t->use_gemm ? t->d_gmap : t->d_wmap;
        float* wout = t->use_gemm ? t->d_gwp : t->d_wp;
        const int64_t nw = (int64_t)(t->use_gemm ? t->h_gmap.size() : t->h_wmap.size()), nct = (int64_t)2 * g.Q * g.C;
        const int64_t tot = nw + nct + t->n_bias + 16 + 64;
        hipLaunchKernelGGL(k_refresh, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream, d_flat, wmap, wout, nw, t->d_ctmap, t->d_ct, nct,
                           t->d_bstart, t->d_blist, t->d_bp, t->n_bias, t->d_status, t->d_loss);
        p.ct = t->d_ct;
    }
    qpn_prof_mark(PG_PREP, stream);
    t->fwd_valid = false; t->loss_clear = true;
    ++t->generation;
    const bool fuse_ce = d_targets && !t->use_gemm && g.Q <= g.S && g.Q % 256 == 0 && !t->knobs.ce_separate;
    p.ce_tgt = fuse_ce ? d_targets : nullptr; p.ce_stride = tgt_stride; p.ce_dlogits = d_dlogits; p.ce_loss = t->d_loss;
    if (fuse_ce && !want_logits) p.logits = nullptr;
    const bool fuse_post = fuse_bwd && fuse_ce && d_dlogits && t->knobs.post_fuse && t->knobs.post_wide && t->knobs.zero_in_post && g.S == 256 && g.Q == 256 && g.C == 64 && (p.LC == 256 || p.LC == 512);
    t->post_bwd_done = false;
    rc = t->use_gemm ? qpn_launch_fwd_gemm(p, t->gm, stream) : qpn_launch_fwd(p, t->knobs, t->ag, &t->sqf, stream, fuse_post ? &t->bw : nullptr);
    if (!rc) { t->post_bwd_done = fuse_post; t->post_bwd_dlogits = d_dlogits; }
    p.ce_tgt = nullptr; p.logits = d_logits;
    if (rc) return rc;
    t->fwd_valid = true;
    if (fuse_ce) { t->loss_clear = false; qpn_prof_mark(PG_CE, stream); }
    else if (d_targets) {
        rc = qpn_launch_ce(d_logits, d_targets, tgt_stride, B, BL, g.Q, d_dlogits, t->d_loss, t->d_status, t->loss_clear, stream); if (rc) return rc;
        t->loss_clear = false;
    }
    return QPN_OK;
}

extern "C" int qpn_train_forward(qpn_handle* h, const float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                                 const int64_t* d_x, const float* d_h, const float* d_dfac, float* d_logits, void* stream_) {
    return train_forward_impl(h, d_flat, B, T, F, Td, BL, maxd, d_x, d_h, d_dfac, d_logits, nullptr, 0, nullptr, 1, stream_);
}

extern "C" int qpn_train_forward_loss(qpn_handle* h, const float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                                      const int64_t* d_x, const float* d_h, const float* d_dfac, const int64_t* d_targets, int64_t tgt_stride,
                                      float* d_logits, int want_logits, float* d_dlogits, void* stream_) {
    if (!d_targets) { qpn_set_error("bad train_forward_loss arguments: no targets"); return QPN_EINVAL; }
    return train_forward_impl(h, d_flat, B, T, F, Td, BL, maxd, d_x, d_h, d_dfac, d_logits, d_targets, tgt_stride, d_dlogits, want_logits & 1, stream_, (want_logits & QPN_FWD_BACKWARD_FOLLOWS) != 0);
}

extern "C" int qpn_train_loss(qpn_handle* h, double* h_loss, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train || !h_loss) { qpn_set_error("qpn_train_loss needs a preceding qpn_train_forward_loss / qpn_ce_loss"); return QPN_ESTATE; }
    hipStream_t stream = (hipStream_t)stream_;
    double parts[64];
    QPN_HIP(hipMemcpyAsync(parts, h->train->d_loss, sizeof(parts), hipMemcpyDeviceToHost, stream));
    QPN_HIP(hipStreamSynchronize(stream));
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) sum += parts[i];
    *h_loss = sum;
    return QPN_OK;
}

// The loss without draining the stream every step: _enqueue copies this step's 64 partial sums to a pinned slot behind the step's kernels (two
// slots: the copy of two steps ago is long done when its slot is reused); _collect(newest = 0) returns the sum enqueued one call EARLIER -- complete
// by the time the host has enqueued another step, so it does not wait in practice --, newest = 1 the last one (waits for it).  *h_valid = 0: no such copy.
extern "C" int qpn_train_loss_enqueue(qpn_handle* h, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train) { qpn_set_error("qpn_train_loss_enqueue needs a preceding qpn_train_forward_loss / qpn_ce_loss"); return QPN_ESTATE; }
    TrainState* t = h->train;
    const int slot = t->loss_newest ^ 1;
    if (t->loss_pending[slot]) QPN_HIP(hipEventSynchronize(t->ev_loss[slot]));      // (an uncollected copy of two calls ago: dropped)
    QPN_HIP(hipMemcpyAsync(t->h_loss_pinned + 64 * slot, t->d_loss, 64 * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    QPN_HIP(hipEventRecord(t->ev_loss[slot], (hipStream_t)stream_));
    t->loss_pending[slot] = true; t->loss_newest = slot;
    return QPN_OK;
}
extern "C" int qpn_train_loss_collect(qpn_handle* h, int newest, double* h_loss, int* h_valid) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h_loss || !h_valid) { qpn_set_error("bad loss_collect arguments"); return QPN_EINVAL; }
    *h_valid = 0; *h_loss = 0.0;
    if (!h->train) return QPN_OK;
    TrainState* t = h->train;
    const int slot = newest ? t->loss_newest : t->loss_newest ^ 1;
    if (!t->loss_pending[slot]) return QPN_OK;
    QPN_HIP(hipEventSynchronize(t->ev_loss[slot]));
    t->loss_pending[slot] = false;
    double sum = 0.0;
    for (int i = 0; i < 64; ++i) sum += t->h_loss_pinned[64 * slot + i];
    *h_loss = sum; *h_valid = 1;
    return QPN_OK;
}

static int status_to_rc(int st);
extern "C" int qpn_train_status(qpn_handle* h, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train) { qpn_set_error("no training call yet"); return QPN_ESTATE; }
    QPN_HIP(hipStreamSynchronize((hipStream_t)stream_));
    int st = 0;
    QPN_HIP(hipMemcpy(&st, h->train->d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) QPN_HIP(hipMemsetAsync(h->train->d_status, 0, sizeof(int), (hipStream_t)stream_));          // sticky until read: reported once
    h->train->last_stream = (hipStream_t)stream_;
    if (st & 4) h->train->stack_disabled = true;
    h->train->status_pending[0] = h->train->status_pending[1] = false;
    return status_to_rc(st);
}

static int status_to_rc(int st) {
    if (st & 1) { qpn_set_error("pitch-dependent tap outside the layer input (dilated factor > maxd or < 0; reference assert qpnet.py:294)"); return QPN_ERANGE; }
    if (st & 4) { qpn_set_error("the one-launch residual stack gave up waiting for a peer workgroup: the flagged step's results are invalid (its Adam update, and that of the steps enqueued behind it until this report, were skipped on the device: parameters and moments are those of the last clean step); the handle runs a launch per layer from here on (QPN_STACK_QUEUE=0 selects that from the start)"); return QPN_ENODEV; }
    if (st & 2) { qpn_set_error("target class outside [0, n_quantize) (reference assert qpnet_train.py:525)"); return QPN_ERANGE; }
    if (st & 8) { qpn_set_error("a peer rank flagged its chunk of this data-parallel step (a tap or target out of range, or an abandoned stack launch there): every rank skipped the update, the replicas are unchanged"); return QPN_ERANGE; }
    return QPN_OK;
}

// The same check without draining the stream: _enqueue copies the (sticky) status word to pinned memory behind the work enqueued so
// far and returns; _collect waits for the copies only and reports them.  Two slots: _collect_lagged looks at every enqueued check
// EXCEPT the newest -- a training loop that calls it at the start of step i + 1 never waits for step i (whose kernels the device still has
// queued while the host enqueues the next step), and reports a bad chunk two steps late at most.
static int status_collect_slot(TrainState* t, int slot) {
    if (!t->status_pending[slot]) return QPN_OK;
    QPN_HIP(hipEventSynchronize(t->ev_status[slot]));
    t->status_pending[slot] = false;
    const int st = t->h_status_pinned[slot];
    if (st & 4) t->stack_disabled = true;
    if (st) {
        // sticky until read: reported once.  Cleared ON the stream the training calls run on, i.e. behind every step enqueued so far (their Adam kernels all see
        // the word set and skip) and in front of the next one -- a null-stream memset is not ordered against a non-blocking stream and could land in the middle
        // of a k_adam launch, whose blocks each read the word
        QPN_HIP(hipMemsetAsync(t->d_status, 0, sizeof(int), t->last_stream));
        const int other = slot ^ 1;                                   // ... also where the other slot copied the same sticky bits before this clear
        if (t->status_pending[other]) {
            QPN_HIP(hipEventSynchronize(t->ev_status[other]));
            t->h_status_pinned[other] &= ~st;
            if (!t->h_status_pinned[other]) t->status_pending[other] = false;
        }
    }
    return status_to_rc(st);
}
extern "C" int qpn_train_status_enqueue(qpn_handle* h, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train) { qpn_set_error("no training call yet"); return QPN_ESTATE; }
    TrainState* t = h->train;
    const int slot = t->status_newest ^ 1;
    t->last_stream = (hipStream_t)stream_;
    rc = status_collect_slot(t, slot); if (rc) return rc;            // (two enqueues old: long done)
    QPN_HIP(hipMemcpyAsync(t->h_status_pinned + slot, t->d_status, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    QPN_HIP(hipEventRecord(t->ev_status[slot], (hipStream_t)stream_));
    t->status_pending[slot] = true; t->status_newest = slot;
    return QPN_OK;
}
extern "C" int qpn_train_status_collect(qpn_handle* h) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train) return QPN_OK;
    TrainState* t = h->train;
    rc = status_collect_slot(t, t->status_newest ^ 1); if (rc) return rc;
    return status_collect_slot(t, t->status_newest);
}
// ... and without waiting at all: only the checks whose copies have already landed are looked at (hipEventQuery); *pending = checks still in flight
extern "C" int qpn_train_status_poll(qpn_handle* h, int* pending) {
    int rc = need_dev(h); if (rc) return rc;
    if (pending) *pending = 0;
    if (!h->train) return QPN_OK;
    TrainState* t = h->train;
    for (int k = 0; k < 2; ++k) {
        const int slot = k == 0 ? t->status_newest ^ 1 : t->status_newest;      // older first
        if (!t->status_pending[slot]) continue;
        if (hipEventQuery(t->ev_status[slot]) != hipSuccess) { (void)hipGetLastError(); if (pending) ++*pending; continue; }
        rc = status_collect_slot(t, slot); if (rc) return rc;
    }
    return QPN_OK;
}
extern "C" int qpn_train_status_collect_lagged(qpn_handle* h) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train) return QPN_OK;
    return status_collect_slot(h->train, h->train->status_newest ^ 1);
}

extern "C" int qpn_ce_loss(qpn_handle* h, const float* d_logits, const int64_t* d_targets, int64_t tgt_stride, int B, int BL,
                           float* d_dlogits, double* h_loss, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    rc = train_init(h); if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (!d_logits || !d_targets || B < 1 || BL < 1 || tgt_stride < BL) { qpn_set_error("bad ce_loss arguments"); return QPN_EINVAL; }
    rc = qpn_launch_ce(d_logits, d_targets, tgt_stride, B, BL, h->g.Q, d_dlogits, h->train->d_loss, h->train->d_status, h->train->loss_clear, stream); if (rc) return rc;
    h->train->loss_clear = false;
    if (h_loss) {
        double parts[64];
        QPN_HIP(hipMemcpyAsync(parts, h->train->d_loss, sizeof(parts), hipMemcpyDeviceToHost, stream));
        QPN_HIP(hipStreamSynchronize(stream));
        double sum = 0.0;
        for (int i = 0; i < 64; ++i) sum += parts[i];
        *h_loss = sum;
    }
    return QPN_OK;
}

// dev / diagnostics: the control words of the stack work queues as the last step left them (synchronises the stream)
extern "C" int qpn_train_stack_stats(qpn_handle* h, unsigned* h_out, int n, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train || !h->train->d_sq || !h_out || n < 1) { qpn_set_error("no stack-queue launch yet"); return QPN_ESTATE; }
    QPN_HIP(hipStreamSynchronize((hipStream_t)stream_));
    QPN_HIP(hipMemcpy(h_out, h->train->d_sq, sizeof(unsigned) * (size_t)(n < 1024 ? n : 1024), hipMemcpyDeviceToHost));      // (words 600.. : stamps of a -DQPN_STACK_STAMPS build)
    return QPN_OK;
}

// Adam updates APPLIED on this handle so far (k_adam counts on the device: a launch that found the status word set -- its own rank's or, data-parallel, a peer's --
// applies nothing).  Drains `stream`.  A caller that counts steps on the host (the bias correction's step number) re-bases its count on this after a status error.
extern "C" int qpn_train_applied_updates(qpn_handle* h, int64_t* applied, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!applied) { qpn_set_error("bad applied_updates arguments"); return QPN_EINVAL; }
    *applied = 0;
    if (!h->train) return QPN_OK;
    QPN_HIP(hipStreamSynchronize((hipStream_t)stream_));
    unsigned long long v = 0;
    QPN_HIP(hipMemcpy(&v, h->train->d_status + 2, sizeof(v), hipMemcpyDeviceToHost));
    *applied = (int64_t)v;
    return QPN_OK;
}

#ifdef QPN_TESTING
// test hook (a -DQPN_TESTING build only; tests/f64_child.py): the rectified post-net activations relu(s0), relu(y0) of the last forward ([B][BL][S] floats each) --
// their signs are the ReLU sides this forward took, which a float64 yardstick of the gradient must be given (a unit within rounding of zero falls on either side)
extern "C" int qpn_test_postnet_activations(qpn_handle* h, float* d_s0, float* d_y0, int64_t n, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train || !h->train->fwd_valid || !d_s0 || !d_y0) { qpn_set_error("no forward to read"); return QPN_ESTATE; }
    const TrainParams& p = h->train->tp;
    if (n != (int64_t)p.B * p.BL * p.S) { qpn_set_error("postnet_activations: n != B * BL * S"); return QPN_EINVAL; }
    QPN_HIP(hipMemcpyAsync(d_s0, p.S0, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream_));
    QPN_HIP(hipMemcpyAsync(d_y0, p.Y0, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream_));
    return QPN_OK;
}
#endif

extern "C" int64_t qpn_train_generation(qpn_handle* h) { return (h && h->train) ? h->train->generation : 0; }

extern "C" int qpn_train_backward(qpn_handle* h, const float* d_dlogits, float* d_flatgrad, void* stream_) {
    return qpn_train_backward_ex(h, d_dlogits, d_flatgrad, 1.0f, 0, stream_);
}

extern "C" int qpn_train_backward_ex(qpn_handle* h, const float* d_dlogits, float* d_flatgrad, float grad_scale, int append_scale, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!h->train || !h->train->fwd_valid) { qpn_set_error("qpn_train_backward needs a preceding qpn_train_forward"); return QPN_ESTATE; }
    if (!d_dlogits || !d_flatgrad) { qpn_set_error("bad train_backward arguments"); return QPN_EINVAL; }
    TrainState* t = h->train;
    TrainBwd& bw = t->bw;
    bw.dlogits = d_dlogits; bw.gflat = d_flatgrad; bw.gdst = t->d_gdst; bw.gdst_list = t->d_gdst_list; bw.gzero = t->d_gzero; bw.n_gzero = t->n_gzero;
    bw.gscale = grad_scale; bw.append_scale = append_scale; bw.status = h->train->d_status;
    bw.side = t->side; bw.ev_fork = t->ev_fork; bw.ev_join = t->ev_join; bw.ev_mid = t->ev_mid;
    t->early_recorded = 0; bw.ev_early = t->ev_early; bw.early_recorded = (append_scale && t->early_first >= 0) ? &t->early_recorded : nullptr;
    // (the backward queue's heads and flags belong to ONE backward per forward: a repeated backward of the same forward runs a launch per layer)
    const bool first_bwd = t->bwd_generation != t->generation;
    t->bwd_generation = t->generation;
    const bool post_done = t->post_bwd_done && first_bwd && d_dlogits == t->post_bwd_dlogits;      // (a repeated backward, or another dL/dlogits: the separate kernel)
    t->post_bwd_done = false;
    return t->use_gemm ? qpn_launch_bwd_gemm(t->tp, bw, t->gm, (hipStream_t)stream_)
                       : qpn_launch_bwd(t->tp, bw, t->knobs, t->ag, first_bwd ? &t->sqb : nullptr, (hipStream_t)stream_, post_done);
}

extern "C" int qpn_train_early_bucket(qpn_handle* h, int64_t* first, int64_t* count, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!first || !count) { qpn_set_error("bad early_bucket arguments"); return QPN_EINVAL; }
    *first = 0; *count = 0;
    TrainState* t = h->train;
    if (!t || !t->early_recorded) return QPN_OK;            // this backward finished nothing early (one stream, no upsampling kernel, a plain backward): one exchange
    *first = t->early_first; *count = t->bw.n_params - t->early_first + 4;
    QPN_HIP(hipStreamWaitEvent((hipStream_t)stream_, t->ev_early, 0));
    return QPN_OK;
}

// first element of the flat-gradient tail that a backward finishes early (the post-net blocks; the 4-float trailer follows them), or -1: a
// property of the model's parameter layout, known once the handle has trained a step -- what data-parallel ranks agree on before they split the exchange
extern "C" int64_t qpn_train_early_first(qpn_handle* h) { return (h && h->train && h->device >= 0) ? h->train->early_first : -1; }

extern "C" int qpn_adam_step(qpn_handle* h, float* d_flat, const float* d_grad, float* d_m, float* d_v, int64_t n,
                             int step, float lr, float beta1, float beta2, float eps, float weight_decay, void* stream_) {
    return qpn_adam_step_ex(h, d_flat, d_grad, d_m, d_v, n, step, lr, beta1, beta2, eps, weight_decay, nullptr, stream_);
}

extern "C" int qpn_adam_step_ex(qpn_handle* h, float* d_flat, const float* d_grad, float* d_m, float* d_v, int64_t n,
                                int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                                const float* d_grad_denominator, void* stream_) {
    int rc = need_dev(h); if (rc) return rc;
    if (!d_flat || !d_grad || !d_m || !d_v || n < 1 || step < 1) { qpn_set_error("bad adam_step arguments"); return QPN_EINVAL; }
    if (h->train) h->train->last_stream = (hipStream_t)stream_;
    return qpn_launch_adam(d_flat, d_grad, d_m, d_v, n, step, lr, beta1, beta2, eps, weight_decay, d_grad_denominator, h->train ? h->train->d_status : nullptr, nullptr, nullptr, nullptr, (hipStream_t)stream_);
}

// One optimisation step behind ONE call -- what a training loop's body is (reference src/bin/qpnet_train.py:517-531: forward, CrossEntropyLoss, backward,
// Adam.step, loss.item()): the calls FusedTrainer.step used to make one by one, in their order.  The host side of a step is then a single foreign call
// (a Python caller's other threads -- the loader -- run while it is in progress) instead of seven with interpreter work between them.
//   loss_mode 0: no loss;  1 "lagged": this step's loss is copied out behind its kernels and *h_loss receives the PREVIOUS step's (*h_valid = 0 at the first
//   step or right behind a flush: qpn_train_loss_collect(newest = 1) fetches the last one), the stream is never drained;  2: this step's loss, read in the call
//   (drains the stream, as the reference's loss.item() does).
// The device-side status word is handled as FusedTrainer.step documents: collected two steps late at most (mode 0 / 1), in the call (mode 2).
extern "C" int qpn_train_step(qpn_handle* h, float* d_flat, int B, int64_t T, int64_t F, int64_t Td, int BL, int maxd,
                              const int64_t* d_x, const float* d_h, const float* d_dfac, const int64_t* d_targets, int64_t tgt_stride,
                              float* d_logits, float* d_dlogits, float* d_grad, float* d_m, float* d_v, int64_t n,
                              int step, float lr, float beta1, float beta2, float eps, float weight_decay,
                              int loss_mode, double* h_loss, int* h_valid, void* stream) {
    if (h_valid) *h_valid = 0;
    if (h_loss) *h_loss = 0.0;
    if (loss_mode < 0 || loss_mode > 2 || (loss_mode && (!h_loss || !h_valid))) { qpn_set_error("bad train_step arguments"); return QPN_EINVAL; }
    int rc = need_dev(h); if (rc) return rc;
    rc = qpn_train_status_collect_lagged(h); if (rc) return rc;          // the check of the step before the previous one (never waits for queued work)
    if (!d_targets || !d_dlogits) { qpn_set_error("qpn_train_forward_loss needs targets and a dlogits buffer"); return QPN_EINVAL; }
    rc = train_forward_impl(h, d_flat, B, T, F, Td, BL, maxd, d_x, d_h, d_dfac, d_logits, d_targets, tgt_stride, d_dlogits, 0, stream, true); if (rc) return rc;
    rc = qpn_train_backward(h, d_dlogits, d_grad, stream); if (rc) return rc;
    if (loss_mode == 2) {
        rc = qpn_adam_step_ex(h, d_flat, d_grad, d_m, d_v, n, step, lr, beta1, beta2, eps, weight_decay, nullptr, stream); if (rc) return rc;
        rc = qpn_train_loss(h, h_loss, stream); if (rc) return rc;
        *h_valid = 1;
        return qpn_train_status(h, stream);
    }
    // modes 0 / 1: the Adam kernel -- the step's last -- writes the status word (and the loss partials) into the pinned slots itself: what
    // qpn_train_status_enqueue / qpn_train_loss_enqueue do with a copy each, without the two copy-engine launches behind the step
    if (!d_flat || !d_grad || !d_m || !d_v || n < 1 || step < 1) { qpn_set_error("bad adam_step arguments"); return QPN_EINVAL; }
    TrainState* t = h->train;
    const int sslot = t->status_newest ^ 1;
    rc = status_collect_slot(t, sslot); if (rc) return rc;               // (two enqueues old: long done)
    const int lslot = t->loss_newest ^ 1;
    if (loss_mode == 1 && t->loss_pending[lslot]) QPN_HIP(hipEventSynchronize(t->ev_loss[lslot]));      // (an uncollected copy of two calls ago: dropped)
    rc = qpn_launch_adam(d_flat, d_grad, d_m, d_v, n, step, lr, beta1, beta2, eps, weight_decay, nullptr, t->d_status, t->h_status_pinned + sslot,
                         loss_mode == 1 ? t->h_loss_pinned + 64 * lslot : nullptr, loss_mode == 1 ? t->d_loss : nullptr, (hipStream_t)stream);
    if (rc) return rc;
    QPN_HIP(hipEventRecord(t->ev_status[sslot], (hipStream_t)stream));
    t->status_pending[sslot] = true; t->status_newest = sslot;
    if (loss_mode == 1) {
        QPN_HIP(hipEventRecord(t->ev_loss[lslot], (hipStream_t)stream));
        t->loss_pending[lslot] = true; t->loss_newest = lslot;
        return qpn_train_loss_collect(h, 0, h_loss, h_valid);
    }
    return QPN_OK;
}
