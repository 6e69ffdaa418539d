// qpn_handle.h -- the library handle (shared by decode.hip and train_host.hip)
#pragma once
#include "qpn_common.h"

struct BiasDesc { int64_t auxb[2], convb[2], convPb[2]; int adaptive; int pad; };
struct TrainState;
void qpn_train_destroy(TrainState* t);

// decode launch-plan knobs, parsed once in qpn_create (all optional; tests / tools build a fresh handle per setting)
struct DecodeKnobs {
    bool generic;          // QPN_DECODE_GENERIC: the interpreter kernel k_decode instead of the straight-line k_decode_fast
    bool no_resl;          // QPN_DECODE_NO_RESL: no LDS-resident residual tiles in the one-CU kernel
    int coop;              // QPN_DECODE_COOP=<G>: cooperative decode with up to G workgroups per utterance (0: only where one CU cannot hold the state)
    int coopb;             // QPN_DECODE_COOPB: smallest batch that takes the utterance-batched cooperative kernel (decode_coopb.hip) where it applies (default 1: always; 0 = never -- the per-utterance kernel decode_coop.hip)
    int coopb_delay[3];    // QPN_COOPB_DELAY_G / _X / _T (dev): first poll of the gate / block-output / post-net gathers of that kernel, x 128 clocks after the exchange waves reach it (8 / 4 / 0)
    int coopb_per;         // QPN_DECODE_COOPB_PER (dev): utterances per group of that kernel (default: the batch spread over as many groups as fit the chip)
    int pipe;              // QPN_DECODE_PIPE: 0 = one-CU kernels, 1 / unset = the five-role pipelined kernel where it applies
    bool hybrid;           // QPN_DECODE_HYBRID (dev): rows beyond the pipelined capacity on one-CU kernels beside the launch
    bool stamps;           // QPN_STAMPS (dev, -DQPN_ENABLE_STAMPS builds)
    bool test_pipe_gives_up;   // -DQPN_TESTING builds only (QPN_TEST_PIPE_GIVES_UP=1)
};

struct qpn_handle {
    Geom g;
    int device;
    DecodeKnobs dk;
    // decode program
    std::vector<int> h_map;          // gather map of the packed tile buffer
    std::vector<Task> h_tasks;
    int n_slots;
    int aux_woff4, aux_tiles, logRa;
    DecodeParams dp;                 // template (pointers filled per call)
    FastParams fp;                   // tile map of the specialised kernel
    int* d_map; float* d_wpk; Task* d_tasks; float* d_qb; BiasDesc* d_bd; int* d_status; int* d_bias_src;
    std::vector<int> h_bias_src;
    const float* d_flat; bool have_weights;
    // per-call workspaces (grow only)
    float* d_pproj; size_t pproj_cap;
    float* d_ring; size_t ring_cap;
    int* d_known; size_t known_cap;
    UttDesc* d_utts; size_t utts_cap;
    hipEvent_t ev0, ev1; float last_ms;
    bool pending;
    // device facts queried at qpn_create (nothing assumes a 256-CU chip that is all ours)
    int n_cus;                       // hipDeviceAttributeMultiprocessorCount
    int pipe_rows;                   // five-role groups one pipelined launch can hold resident (5 workgroups each), a multiple of 8
    int pipe_nu;                     // utterances per group when the batch exceeds them (1..3)
    // pinned staging of the utterance descriptors (enqueue does not synchronise) + the side stream of a hybrid launch
    UttDesc* h_utts_pinned; size_t h_utts_cap;
    hipStream_t dec_side; hipEvent_t dec_fork, dec_join;
    struct DecodeCall {              // arguments of the decode in flight: qpn_decode_finish re-runs it on the one-CU kernel when a
        int B, n_x; int64_t F, Td;   // multi-workgroup launch gave up (peers not co-resident: masked / shared GPU)
        const int64_t* d_x; const float* d_h; const void* d_dfac; int d_is_f32;
        std::vector<int64_t> n_samples; int maxd, mode; uint64_t seed;
        const int64_t* d_teacher; int64_t* d_out; float* d_logits;
        int multi_wg;                // 0: one-CU kernels only, 1: pipelined launch(es) involved, 2: cooperative (G > 1)
        int coopG;
    } call;
    std::string plan;                // human-readable launch plan of the last decode (qpn_last_decode_plan)
    bool single_cu_ok;               // the step state fits one CU's LDS (decode.hip kernels); otherwise decode_coop.hip only
    int w_past_il[QPN_MAX_LAYERS];   // channel-interleaved past-tap tiles (cooperative kernel)
    unsigned long long* d_xch; size_t xch_cap;   // exchange granules of the cooperative kernels
    // batched cooperative kernel (decode_coopb.hip): workgroup w's A-operand fragments are the cb_per_w float4 from cb_base4 + w * cb_per_w of the packed weights;
    // cb_zc.. = float4 offsets of the tiles inside that block; or cb_ok = false
    bool cb_ok; int cb_zc[QPN_MAX_LAYERS], cb_zp[QPN_MAX_LAYERS], cb_rs[QPN_MAX_LAYERS], cb_p1, cb_p2; long long cb_base4; int cb_per_w;
    int cb_groups, cb_per;                       // plan of the last batched launch
    bool decode_ok; std::string decode_err;   // geometries the decode kernels do not cover still train (and report why on decode calls)
    struct TrainState* train;        // lazily created by the training entry points (train_host.hip)
};
