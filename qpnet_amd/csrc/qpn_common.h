// qpn_common.h -- internal definitions shared by the HIP translation units of libqpnet_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/qpnet_hip.h"

#define QPN_MAX_LAYERS 48
#define QPN_NW 16                 // waves per decode workgroup
#define QPN_NT (QPN_NW * 64)

// ---------------------------------------------------------------- error plumbing
void qpn_set_error(const char* fmt, ...);
#define QPN_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t _e = (call);                                                            \
        if (_e != hipSuccess) {                                                            \
            qpn_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? QPN_ENOMEM : QPN_ENODEV;                    \
        }                                                                                  \
    } while (0)

// ---------------------------------------------------------------- geometry
struct LayerGeom {
    int adaptive;      // 0 fixed, 1 pitch-adaptive
    int dilation;      // 2^k
    // flat offsets (floats) into the state_dict-ordered parameter vector
    // fixed:    convS/convT = (C,C,2) weights (tap0 older), biasS/biasT
    // adaptive: convC/convP per half
    int64_t wS, bS, wT, bT;            // fixed: dil*_sigmoid/tanh .conv.weight/.bias ; adaptive: convC
    int64_t wSP, bSP, wTP, bTP;        // adaptive only: convP
    int64_t auxS, auxSb, auxT, auxTb;  // (C,A,1), (C)
    int64_t skip, skipb, res, resb;    // (S,C,1),(S),(C,C,1),(C)
};

struct Geom {
    qpn_config cfg;
    int C, S, Q, A, U, LF, LA, L;
    int Cp, Sp, Ap;            // K paddings (16 * power of two)
    int recF, recA;
    int64_t causal_w, causal_b, up_w, up_b, post1_w, post1_b, post2_w, post2_b;
    int64_t n_params;
    LayerGeom layers[QPN_MAX_LAYERS];
};

int qpn_build_geom(const qpn_config* cfg, Geom* g);
__host__ __device__ static inline int qpn_pad_k(int K) { int R = 1; while (16 * R < K) R *= 2; return 16 * R; }

// ---------------------------------------------------------------- decode program
enum {
    OP_NOP = 0, OP_PAST, OP_Z, OP_RES, OP_SKIP, OP_POST1, OP_POST2, OP_CAUSAL, OP_ARGMAX, OP_STAGE
};
#define TF_BARRIER 0x100      // workgroup barrier before this task
#define TF_DRAIN   0x200      // ... and drain this wave's global stores first
#define TF_LAST    0x400      // OP_SKIP of the last layer: also emit relu(skip total)
#define TF_HASW    0x800      // task consumes a weight tile (prefetchable)

struct Task {          // 32 bytes, wave-uniform; the table is copied to LDS at kernel start
    int op;            // opcode | flags | (log2 R << 16)
    int woff4;         // float4 offset of the weight tile in the packed buffer
    int xoff;          // LDS float offset of the input vector
    int row0;          // first row of the tile (in the matrix' packed row order)
    int a, b, c, d;    // op specific
};

struct RingDesc { int base; int len; int mult; int adaptive; };   // base: float offset in the utterance's ring block

struct UttDesc {            // offsets into the base pointers of DecodeParams (kernel-argument pointers keep
                            // the GLOBAL address space; pointers loaded from memory would become FLAT accesses)
    int64_t pproj;          // floats: [F][L][2C] per-frame aux projections
    int64_t dfac;           // elements: dilated factors row (double or float)
    int64_t known;          // ints: [n0] padded known prefix (sample ids)
    int64_t teacher;        // int64s: optional [n_samples], -1 = none
    int64_t out;            // int64s: [n_samples]
    int64_t logits;         // floats: optional [n_samples][Q], -1 = none
    int64_t ring;           // floats: ring block of this utterance (zeroed before launch)
    int n_pad, n0, n_samples, d_is_f32;
    int64_t F;
    int row, pad_;          // row of the caller's batch (the descriptors of a launch are a permuted slice of the batch): Philox key
};

// per-layer tile / bias locations for the specialised straight-line decode kernel (k_decode_fast)
struct FastParams {
    int w_cur[QPN_MAX_LAYERS], w_past[QPN_MAX_LAYERS], w_res[QPN_MAX_LAYERS], w_skip[QPN_MAX_LAYERS];   // float4 offsets
    int b_res[QPN_MAX_LAYERS], b_skip[QPN_MAX_LAYERS];                                                   // LDS float offsets
    int adaptive[QPN_MAX_LAYERS];
    int w_p1, w_p2, b_p1, b_p2;
};

// cooperative decode (decode_coop.hip): G workgroups per utterance, each owning 1/G of every layer's output rows; full
// vectors are exchanged through 8-byte {tag, value} granules in global memory (tag = step + 1, zeroed before every launch)
struct CoopParams {
    unsigned long long* xch;          // exchange memory
    long utt_stride;                  // granules per utterance
    int o_ring[QPN_MAX_LAYERS];       // granule offsets inside an utterance block: layer-input history [len_l][C]
    int o_g, o_y1, o_y2, o_lg;        // gate vectors [L][C], relu(skip total) [S], post-1 output [S], logits [Q]
    int* abort;                       // set when a wait timed out: every workgroup drains instead of spinning on
    int G, CB, SB, QB, Sp;            // workgroups per utterance; channels / skip rows / logit rows per workgroup
    int logR, rpt, logRs, rpts;       // lanes per row (log2) and rows per 4 KiB tile for K = C and K = S
    int w_past_il[QPN_MAX_LAYERS];    // float4 offsets of the past-tap tiles with (sigma_c, tanh_c) rows interleaved
    int f_resb[QPN_MAX_LAYERS], f_skipb[QPN_MAX_LAYERS], f_p1b, f_p2b;   // flat offsets of the biases the epilogues add
};

struct DecodeParams {
    const float4* wpk;
    const float* flat;
    const float* qb;        // [L][2C]
    const Task* tasks;
    const UttDesc* utts;
    const float* pproj; const void* dfac; const int* known; const int64_t* teacher; int64_t* out; float* logits; float* ring;
    int* status;
    long long* stamps;      // dev aid (QPN_STAMPS=1): s_memtime stamps of one step, [stamp][wave]
    int n_slots;
    int C, Cp, S, Q, L, U, mode;
    int64_t causal_w, causal_b, up_w;
    int o_xbuf, o_xp, o_pd, o_auxv, o_g, o_skf, o_ska, o_y1, o_y2, o_lg, o_samp, o_sel, state_floats;
    int o_bias, n_bias, o_tasks, lds_floats;
    int o_gl;               // specialised kernel: per-layer gate vectors [L][Cp]
    int o_stamp;            // dev stamps: first LDS float behind everything the launched kernel uses
    int o_wres;             // specialised kernel: residual-1x1 tiles of layers 0..L-2 resident in LDS (float offset, 16-byte aligned)
    const int* bias_src;    // [n_bias] flat indices of the biases mirrored in LDS
    unsigned long long seed;
    RingDesc rings[QPN_MAX_LAYERS];
};
