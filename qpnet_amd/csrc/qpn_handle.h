// qpn_handle.h -- the library handle (shared by decode.hip and train_host.hip)
#pragma once
#include "qpn_common.h"

struct BiasDesc { int64_t auxb[2], convb[2], convPb[2]; int adaptive; int pad; };
struct TrainState;
void qpn_train_destroy(TrainState* t);

struct qpn_handle {
    Geom g;
    int device;
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
    bool single_cu_ok;               // the step state fits one CU's LDS (decode.hip kernels); otherwise decode_coop.hip only
    int w_past_il[QPN_MAX_LAYERS];   // channel-interleaved past-tap tiles (cooperative kernel)
    unsigned long long* d_xch; size_t xch_cap;   // exchange granules of the cooperative kernel
    bool decode_ok; std::string decode_err;   // geometries the decode kernels do not cover still train (and report why on decode calls)
    struct TrainState* train;        // lazily created by the training entry points (train_host.hip)
};
