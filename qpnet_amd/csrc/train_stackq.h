// train_stackq.h -- the work queue of the one-launch residual stack, shared by its two kernel families: the workgroup-per-tile kernels of train_stack.hip and the
// wave-per-tile kernels of train_stackw.hip.  Positions, flags, tickets and the bounded waits (see the head of train_stack.hip for the protocol).
#pragma once
#include "train_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// dev aid, TIMING ONLY (results are invalid): -DSQ_EXP=<bits> removes parts of the queues' memory traffic, to see what each costs --
// backward: 1 float atomics, 2 plain instead of write-through hand-off (both directions), 4 dZ rows, 16 own-row / fixed-tap stores, 32 every look "ready";
// forward: 64 sigma / tanh rows, 128 block output, 256 row requests, 512 every look "ready"
#ifndef SQ_EXP
#define SQ_EXP 0
#endif
#if SQ_EXP & 2
#define SQ_SC1 0
#else
#define SQ_SC1 16
#endif                                   // aux bits of the raw buffer builtins: sc1 (agent-scope / write-through)
#define SQ_OOB 0x80000000u                          // a buffer offset beyond every descriptor's range: the access is dropped
#define SQ_SPIN_LIMIT (1u << 22)                    // polls of ~1-2 us: several seconds
#define SQ_FLINE 1056u                              // words between the 128-byte lines of the flag array
#define SQ_NQ 8                                     // sub-queues (head words 128 bytes apart)

__device__ __forceinline__ unsigned sq_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sq_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int sq_rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float sq_sigmoid(float z) { return __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float sq_tanh(float z) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * z)) - 1.0f; }

// dev aid (build with -DQPN_STACK_STAMPS): s_memtime (low word) of wave 0 of workgroup 5 at the phase boundaries of its first 25 tiles,
// into control words [600 + 8 * tile + phase]
#ifdef QPN_STACK_STAMPS
#define SQ_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 5 && tid == 0 && it < 25) q.stats[596 + 8 * it + (i)] = (unsigned)t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SQ_STAMP(i) do { } while (0)
#endif
#ifdef QPN_STACK_STAMPS_BWD                         // the same for the backward queue (its counters start 4 words later: same control words)
#define SQ_STAMPB(i) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); if (blockIdx.x == 5 && tid == 0 && it < 25) q.stats[592 + 8 * it + (i)] = (unsigned)t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SQ_STAMPB(i) do { } while (0)
#endif

// Flag words: 32 consecutive positions share a 128-byte line (a consumer's producer range is 2-3 lines), consecutive LINES lie 4224 bytes
// apart: the ~1000 flags the frontier of the queue polls and publishes at any moment would otherwise sit in ONE 4 KB stretch of memory,
// i.e. behind one memory channel (measured: waits of 20-180 us on flags published long before).
__host__ __device__ __forceinline__ static unsigned sq_fidx(unsigned pos) { return (pos >> 5) * SQ_FLINE + (pos & 31u); }

// Wave priority: two workgroups share a CU, so every SIMD holds one wave of each; while one of them issues its MFMA block back to back the other
// one's address arithmetic, LDS and memory instructions have to get between them -- the wave in an MFMA block runs at the lowest priority, everything
// else above it
#ifdef SQ_NOPRIO
#define SQ_PRIO(x) do { } while (0)
#else
#define SQ_PRIO(x) __builtin_amdgcn_s_setprio(x)
#endif

struct SqTile { int n0, meta, dfirst, dn, xrow, tapb, hrow, dgs, pos; };       // wave-uniform (SGPRs); meta: layer | batch item << 8 | last << 24 | valid << 25
__device__ __forceinline__ bool sq_valid(const SqTile& d) { return (d.meta >> 25) & 1; }
__device__ __forceinline__ bool sq_last(const SqTile& d) { return (d.meta >> 24) & 1; }
__device__ __forceinline__ int sq_layer(const SqTile& d) { return d.meta & 255; }

// the table entry of a position (positions past the end: the invalid tile, which addresses like position 0 and stores nothing), in two steps so that
// the load is in flight for most of a tile: sq_fetch issues it (every lane the same address), sq_take makes the words wave-uniform
struct SqRaw { int4 a, b; int pos; };
__device__ __forceinline__ SqRaw sq_fetch(const StackQ& q, int pos) {
    const int i = (pos >= 0 && pos < q.total) ? pos : q.total;
    SqRaw r; r.a = q.tab[2 * i]; r.b = q.tab[2 * i + 1]; r.pos = i;
    return r;
}
__device__ __forceinline__ SqTile sq_take(const SqRaw& r) {
    SqTile d;
    d.n0 = sq_rfl(r.a.x); d.meta = sq_rfl(r.a.y); d.dfirst = sq_rfl(r.a.z); d.dn = sq_rfl(r.a.w);
    d.xrow = sq_rfl(r.b.x); d.tapb = sq_rfl(r.b.y); d.hrow = sq_rfl(r.b.z); d.dgs = sq_rfl(r.b.w); d.pos = r.pos;
    return d;
}

// wait for flags [first, first + n): every lane polls one flag.  `bound` > 0: give up after that many polls per 64-flag group (returns 2:
// the caller escalates -- publishes what it holds and comes back with bound = 0); bound = 0: until SQ_SPIN_LIMIT, then the abort word is
// raised (returns 0, as it does when another workgroup raised it).  1 = every flag carries the epoch.
__device__ __forceinline__ int sq_wait(const StackQ& q, int first, int n, int lane, int* status, unsigned bound) {
    unsigned total_spins = 0;
    int rc = 1;
    for (int base = 0; base < n && rc == 1; base += 64) {
        const int i = base + lane < n ? base + lane : n - 1;
        const unsigned* fp = q.flags + sq_fidx((unsigned)(first + i));
        unsigned spins = 0;
        for (;;) {
            const unsigned v = sq_ld(fp);
            if (__all(v == q.epoch)) break;
            ++total_spins;
            if ((spins & 63u) == 63u && sq_ld(q.abort)) { rc = 0; break; }      // (one word for the whole chip: looked at rarely)
            ++spins;
            if (bound && spins >= bound) { rc = 2; break; }
            if (spins > q.spin_limit) {
#ifdef QPN_STACK_DEBUG
                if (sq_ld(q.abort) == 0u) { const unsigned long long miss = __ballot(v != q.epoch); if (lane == 0) { q.stats[24] = (unsigned)first; q.stats[25] = (unsigned)n; q.stats[26] = (unsigned)base; q.stats[27] = (unsigned)miss; q.stats[28] = (unsigned)(miss >> 32); q.stats[29] = blockIdx.x; q.stats[30] = q.epoch; } if (lane == (int)__ffsll(miss) - 1) q.stats[31] = v; }
#endif
                if (lane == 0) { sq_st(q.abort, 1u); atomicOr(status, 4); } rc = 0; break; }
            if (spins > 2) __builtin_amdgcn_s_sleep(8);
        }
    }
    if (total_spins && lane == 0) atomicAdd(q.stats + 1, total_spins);
    return rc;
}

