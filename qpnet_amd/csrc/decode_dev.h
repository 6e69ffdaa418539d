// decode_dev.h -- device-side pieces shared by the decode kernels (decode.hip: one workgroup per utterance;
// decode_coop.hip: several cooperating workgroups per utterance).  Arithmetic = the fixed-order "QPNet-f32" spec of
// DESIGN.md section 3 (bit-identical to oracle/qpnet_oracle.c); compile with -ffp-contract=off.
#pragma once
#include "qpn_common.h"

// ================================================================== device helpers
// File-scope LDS symbol: device functions index it directly, so the compiler keeps address space 3
// (a generic float* would turn every LDS access into a FLAT op that also waits on the global-load queue).
extern __shared__ float4 qpn_lds[];
#define SM ((float*)qpn_lds)
#define SMI ((int*)qpn_lds)
__device__ __forceinline__ float qexp(float x) {
    x = fminf(fmaxf(x, -87.0f), 88.0f);
    float n = rintf(x * 0x1.715476p+0f);
    float r = __builtin_fmaf(n, -0x1.63p-1f, x);
    r = __builtin_fmaf(n, 0x1.bd0106p-13f, r);
    float p = 0x1.a01a02p-13f;
    p = __builtin_fmaf(p, r, 0x1.6c16c2p-10f);
    p = __builtin_fmaf(p, r, 0x1.111112p-7f);
    p = __builtin_fmaf(p, r, 0x1.555556p-5f);
    p = __builtin_fmaf(p, r, 0x1.555556p-3f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return p * __int_as_float(((int)n + 127) << 23);
}
__device__ __forceinline__ float qgate(float zs, float zt) {
    float ea = qexp(-zs);
    float eb = qexp(-2.0f * fabsf(zt));
    float num = 1.0f - eb;
    float den = (1.0f + ea) * (1.0f + eb);
    return copysignf(num / den, zt);
}

// 16-deep chunk of the spec dot product: p = w0*x0, then 15 fma in k order.
__device__ __forceinline__ float chunk16(const float4 (&w)[4], const float4 (&x)[4]) {
    float acc = w[0].x * x[0].x;
    acc = __builtin_fmaf(w[0].y, x[0].y, acc);
    acc = __builtin_fmaf(w[0].z, x[0].z, acc);
    acc = __builtin_fmaf(w[0].w, x[0].w, acc);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        acc = __builtin_fmaf(w[j].x, x[j].x, acc);
        acc = __builtin_fmaf(w[j].y, x[j].y, acc);
        acc = __builtin_fmaf(w[j].z, x[j].z, acc);
        acc = __builtin_fmaf(w[j].w, x[j].w, acc);
    }
    return acc;
}
// Same dot product, and each 1 KiB quarter of the tile is re-requested (from tp) as soon as its four FMAs have
// been issued: the tile's registers free up a quarter at a time, so the next request reaches the (serial, 16
// cycles per load) vector-memory front end ~150 cycles earlier than after the whole chain.  The scheduling
// barriers pin the loads where they are written; hipcc otherwise sinks them below the epilogue.
__device__ __forceinline__ float chunk16_reload(float4 (&w)[4], const float4 (&x)[4], const float4* __restrict__ tp) {
    float acc = w[0].x * x[0].x;
    acc = __builtin_fmaf(w[0].y, x[0].y, acc);
    acc = __builtin_fmaf(w[0].z, x[0].z, acc);
    acc = __builtin_fmaf(w[0].w, x[0].w, acc);
    __builtin_amdgcn_sched_barrier(0);
    w[0] = tp[0];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        acc = __builtin_fmaf(w[j].x, x[j].x, acc);
        acc = __builtin_fmaf(w[j].y, x[j].y, acc);
        acc = __builtin_fmaf(w[j].z, x[j].z, acc);
        acc = __builtin_fmaf(w[j].w, x[j].w, acc);
        __builtin_amdgcn_sched_barrier(0);
        w[j] = tp[j * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
// stride-halving tree over the R lanes of a row group (R = 1 << logR, lanes contiguous).
// The spec order is "p_i += p_{i+s} for s = R/2 .. 1"; fp add is commutative, so every lane of the
// group ends with the same bits whether the partner is reached by xor, rotation or quad permute.
__device__ __forceinline__ float rl_f(float v, int lane_) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_)); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// (value, index) maximum over the 64 lanes, lowest index among equal values; every lane returns the winner's index.
// Four DPP steps inside the 16-lane rows, then four v_readlane pairs and uniform compares across the rows (a butterfly of
// six ds_bpermute pairs is a ~900-cycle dependent chain on the pick's critical path).
__device__ __forceinline__ int wave_argmax(float bv, int bi) {
#define QPN_AMAX(CTRL) { const float ov_ = dpp_f<CTRL>(bv); const int oi_ = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, true); \
                         if (ov_ > bv || (ov_ == bv && oi_ < bi)) { bv = ov_; bi = oi_; } }
    QPN_AMAX(0xB1) QPN_AMAX(0x4E) QPN_AMAX(0x124) QPN_AMAX(0x128)
#undef QPN_AMAX
    float rv = rl_f(bv, 0); int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int r = 1; r < 4; ++r) {
        const float ov = rl_f(bv, 16 * r); const int oi = __builtin_amdgcn_readlane(bi, 16 * r);
        if (ov > rv || (ov == rv && oi < ri)) { rv = ov; ri = oi; }
    }
    return ri;
}
__device__ __forceinline__ float tree_reduce(float acc, int logR) {
    if (logR == 2) {                       // K = 64: two quad permutes, no LDS crossbar
        acc = acc + dpp_f<0x4E>(acc);      // quad_perm [2,3,0,1]  (stride 2)
        acc = acc + dpp_f<0xB1>(acc);      // quad_perm [1,0,3,2]  (stride 1)
        return acc;
    }
    if (logR == 4) {                       // K = 256: row rotations by 8 and 4, then the quad permutes
        acc = acc + dpp_f<0x128>(acc);     // row_ror:8
        acc = acc + dpp_f<0x124>(acc);     // row_ror:4
        acc = acc + dpp_f<0x4E>(acc);
        acc = acc + dpp_f<0xB1>(acc);
        return acc;
    }
    if (logR == 1) return acc + dpp_f<0xB1>(acc);
    if (logR == 5) {                       // K = 512: the stride-16 partner through v_permlane16_swap_b32 (a VALU op of gfx950), then as K = 256.
        // swap(a, a) leaves {row0, row0, row2, row2} in the first result and {row1, row1, row3, row3} in the second: the lane's
        // partner (lane ^ 16) is the second on even rows, the first on odd rows -- no trip through the LDS crossbar (a ds_bpermute
        // per step was a 5-deep ~120-cycle chain in front of every row of the wide models' matvecs)
        const int a = __float_as_int(acc);
        const auto sw = __builtin_amdgcn_permlane16_swap(a, a, false, false);
        const float partner = __int_as_float((threadIdx.x & 16) ? (int)sw[0] : (int)sw[1]);
        acc = acc + partner;
        acc = acc + dpp_f<0x128>(acc);
        acc = acc + dpp_f<0x124>(acc);
        acc = acc + dpp_f<0x4E>(acc);
        acc = acc + dpp_f<0xB1>(acc);
        return acc;
    }
    for (int s = (1 << logR) >> 1; s >= 1; s >>= 1) acc = acc + __shfl_xor(acc, s);
    return acc;
}
__device__ __forceinline__ void load_tile(float4 (&w)[4], const float4* __restrict__ wpk, int woff4, int lane) {
    const float4* p = wpk + (size_t)woff4 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = p[j * 64];
}
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// Ring-buffer traffic stays inside ONE workgroup (= one CU): producer waves store, drain vmcnt and pass a workgroup
// barrier before any wave loads the row -- the visibility HIP guarantees for global memory across __syncthreads().
// Workgroup scope keeps the rows in the CU's L1 / the XCD's L2 (write-back).  Agent scope (sc1, write-through) was
// measured to push every 4-byte store to the fabric: 2.0 KB written + ~1.8 KB fetched per generated sample
// (profiles/r01_decode_traffic_pmc.txt) against 172 B of algorithmic HBM bytes.
__device__ __forceinline__ float ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---------------------------------------------------------------- sampling mode (reference qpnet.py:507-510)
// softmax + categorical draw by inverse CDF with a counter-based generator (Philox4x32-10, counter = (step, row),
// key = seed), in the fixed order of DESIGN.md §3 so that the CPU oracle reproduces every draw bit for bit.
// (Parity with the reference's torch.Generator stream is statistical only.)  One wave, Q = 64 * per, per <= 4.
__device__ __forceinline__ unsigned philox_first(unsigned c0, unsigned c1, unsigned k0, unsigned k1) {
    unsigned c2 = 0, c3 = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c0;
}
__device__ __forceinline__ float sample_uniform(unsigned long long seed, unsigned row, unsigned step) {
    return (float)(philox_first(step, row, (unsigned)seed, (unsigned)(seed >> 32)) >> 8) * 0x1p-24f;
}

// inverse-CDF draw over the Q logits at SM[o_lg..]; `u` is the step's uniform (sample_uniform), which does not depend on the
// logits and can be computed while the producer is still working
__device__ __forceinline__ int sample_wave_u(int o_lg, int Q, float u, int lane) {
    const float* lg = SM + o_lg;
    const int per = Q >> 6;
    float l[4], e[4];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) { l[j] = j < per ? lg[lane * per + j] : -INFINITY; m = fmaxf(m, l[j]); }
    m = fmaxf(m, dpp_f<0xB1>(m)); m = fmaxf(m, dpp_f<0x4E>(m)); m = fmaxf(m, dpp_f<0x124>(m)); m = fmaxf(m, dpp_f<0x128>(m));
    // (every lane of a 16-lane row now holds the row's maximum: four v_readlane instead of two ds_bpermute round trips; max and
    //  min are order-independent, so the bits are the spec's)
    m = fmaxf(fmaxf(rl_f(m, 0), rl_f(m, 16)), fmaxf(rl_f(m, 32), rl_f(m, 48)));
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = j < per ? qexp(l[j] - m) : 0.0f;
    float a = e[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) if (j < per) a = a + e[j];
    float v = a;
    {   // d = 1 through the wave-wide DPP shift (wave_shr:1, lane 0 reads 0), the wider strides through the LDS crossbar
        const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); if (lane >= 1) v = v + up;
    }
    for (int d = 2; d < 64; d <<= 1) { const float up = __shfl_up(v, d); if (lane >= d) v = v + up; }
    const float total = rl_f(v, 63);
    float c = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
    if (lane == 0) c = 0.0f;
    const float th = u * total;
    int idx = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j) if (j < per) { c = c + e[j]; if (idx == 0x7fffffff && c > th) idx = lane * per + j; }
#define QPN_IMIN(CTRL) { const int o_ = __builtin_amdgcn_update_dpp(0x7fffffff, idx, CTRL, 0xf, 0xf, false); idx = o_ < idx ? o_ : idx; }
    QPN_IMIN(0xB1) QPN_IMIN(0x4E) QPN_IMIN(0x124) QPN_IMIN(0x128)            // within rows of 16 lanes
#undef QPN_IMIN
    {
        const int i0 = __builtin_amdgcn_readlane(idx, 0), i1 = __builtin_amdgcn_readlane(idx, 16), i2 = __builtin_amdgcn_readlane(idx, 32), i3 = __builtin_amdgcn_readlane(idx, 48);
        const int a01 = i0 < i1 ? i0 : i1, a23 = i2 < i3 ? i2 : i3;
        idx = a01 < a23 ? a01 : a23;
    }
    return idx == 0x7fffffff ? Q - 1 : idx;
}

__device__ __forceinline__ int sample_wave(int o_lg, int Q, unsigned long long seed, unsigned row, unsigned step, int lane) {
    return sample_wave_u(o_lg, Q, sample_uniform(seed, row, step), lane);
}

struct UttView {            // per-utterance pointers derived from kernel-argument bases (global address space)
    const float* pproj; const void* dfac; const int* known; const int64_t* teacher; int64_t* out; float* logits; float* ring;
    int n_pad, n0, n_samples, d_is_f32, row;
};
__device__ __forceinline__ UttView make_view(const DecodeParams& p, const UttDesc& d) {
    UttView u;
    u.pproj = p.pproj + d.pproj;
    u.dfac = d.d_is_f32 ? (const void*)((const float*)p.dfac + d.dfac) : (const void*)((const double*)p.dfac + d.dfac);
    u.known = p.known + d.known;
    u.teacher = d.teacher >= 0 ? p.teacher + d.teacher : nullptr;
    u.out = p.out + d.out;
    u.logits = d.logits >= 0 ? p.logits + d.logits : nullptr;
    u.ring = p.ring + d.ring;
    u.n_pad = d.n_pad; u.n0 = d.n0; u.n_samples = d.n_samples; u.d_is_f32 = d.d_is_f32; u.row = d.row;
    return u;
}

// pitch-dependent tap distance of ring `r` at (padded) time t  (qpnet.py:613-624)
// `widx` != 0: warm-up step over the known prefix -- the reference takes those taps from _dilated_index (qpnet.py:416,
// 592-611: rint(-d*dil + idx), idx = position from the end of the prefix), not from _generate_dilated_index
__device__ __forceinline__ int tap_offset(const RingDesc& r, const UttView& u, int ut, int widx) {
    if (!r.adaptive) return r.mult;
    if (ut < 0) return r.mult;                       // d := 1.0 in the left padding (qpnet.py:361-364)
    if (u.d_is_f32) {
        float d = ((const float*)u.dfac)[ut];
        if (widx) return widx - (int)rintf(__fadd_rn(-d * (float)r.mult, (float)widx));
        return -(int)rintf(-d * (float)r.mult);
    }
    double d = ((const double*)u.dfac)[ut];
    if (widx) return widx - (int)rint(__dadd_rn(-d * (double)r.mult, (double)widx));
    return -(int)rint(-d * (double)r.mult);
}
// un-padded time whose aux features / dilated factor step t uses: the newest sample's own in the generation loop
// (qpnet.py:450-452); one EARLIER during the warm-up over the known prefix, where the reference pairs layer output p with
// h[p-1], d[p-1] (h_ = h[:, :, :causal_output.size(-1)], qpnet.py:366-368; visible only with seeds of >= 3 samples)
__device__ __forceinline__ int aux_time(const UttView& u, int t) { return t - u.n_pad - (t < u.n0 - 1 ? 1 : 0); }
