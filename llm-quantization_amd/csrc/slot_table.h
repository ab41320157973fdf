// The slot table of the fresh-statistics activation quantizer, shared by its producers (the statistics kernel and the
// norm kernels that publish the minima / maxima of their own output) and its consumer (the quantize kernel).
//
// A "slot" is one statistics unit: cluster k for the channels of cluster k, n_clusters + (c - covered) for a channel c
// outside every cluster.  Producers combine per-slot minima / maxima with 64-bit atomic maxima on order-preserving
// keys; every word carries, above the 32-bit key, a call number that grows with every call, so that what an earlier
// call left always loses and the table never has to be cleared (it only has to be zero when it is first used).
//   ws[0]  call number of the last completed call  (read by the producer, written by the consumer)
//   ws[1]  call number of the call in flight       (written by the producer, read by the consumer)
//   ws[16 + slab * table_words + 16 * (slot * nrep + rep) + {0, 1}]   {min, max} of replica rep of a slot of a slab
//                                                                      (one 128-byte line each)
// Replicas spread the atomics of a slot that many blocks feed.
#pragma once
#include "miq_common.h"

namespace miq {

typedef unsigned long long u64;

// order-preserving key of a float for an unsigned maximum; NaN maps to the largest key (it wins, like torch.amax).
// A zero enters as +0 (the keys would order -0 below +0, where the float comparisons of the partial-plane kernels
// keep whichever zero came first: k_cluster_scale canonicalises the same way, so the two paths agree in every bit).
__device__ __forceinline__ unsigned max_key(float f) {
    if (f != f) return 0xFFFFFFFFu;
    const unsigned u = __float_as_uint(f + 0.f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float max_key_value(unsigned k) {
    if (k == 0xFFFFFFFFu) return __builtin_nanf("");
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// the same for a minimum: the complemented key, NaN again the largest
__device__ __forceinline__ unsigned min_key(float f) { return (f != f) ? 0xFFFFFFFFu : ~max_key(f); }
__device__ __forceinline__ float min_key_value(unsigned k) {
    if (k == 0xFFFFFFFFu) return __builtin_nanf("");
    return max_key_value(~k);
}

// slot of channel c: by arithmetic for equal-sized clusters (csize > 0), through the map otherwise
struct SlotMap {
    const int* slot_of;  // [C] or nullptr when csize > 0
    int csize;           // size of every cluster, or 0
    int covered;         // channels that belong to a cluster
    int ncl;
    __device__ __forceinline__ int operator()(long long c) const {
        if (csize > 0) return c < covered ? (int)(c / csize) : ncl + (int)(c - covered);
        return slot_of[c];
    }
};
constexpr int kSlotStride = 16;   // 64-bit words per slot replica: one 128-byte line each, so that the atomics of
                                  // different slots (and replicas) are served by different cache lines
constexpr int kSlotReplicas = 8;  // replicas the workspace is sized for
constexpr int kMaxSlabs = 8;      // slabs of one shape per launch (grid.z)
struct SlabPtrs { const void* p[kMaxSlabs]; };
struct SlabOuts { void* p[kMaxSlabs]; };

static inline int64_t fresh_slots(int64_t C, int n_clusters, int64_t covered) { return (int64_t)n_clusters + (C - covered); }
static inline int64_t fresh_table_words(int64_t nslot) { return (int64_t)kSlotStride * nslot * kSlotReplicas; }

// What a row kernel needs to publish the statistics of the rows it writes into slab 0 of `ws` (kSlotReplicas
// replicas): either equal clusters of `packs_per_slot` packs (of the kernel's pack width) that cover all channels, or
// the channel -> slot map of any other layout.
struct RowStatSink {
    u64* ws;             // nullptr: no statistics wanted
    int packs_per_slot;  // > 0: slot of pack j is j / packs_per_slot
    const int* slot_of;  // [C], used when packs_per_slot == 0
};

// rows a team of a statistics-publishing row kernel walks before it publishes.  Measured on [2048, 4096] slabs (norm +
// quantizer forward, against the plain norm followed by the two-kernel quantizer at 34 us fp32 / 30 us fp16): 1 row
// 37 / 34 us with 32 clusters and 33 / 30 us per tensor; 4 rows 43 / 29 us and 41 / 27 us (fewer exchanges and atomics,
// but two blocks per CU no longer cover the memory latency in fp32).
constexpr int kStatRows = 1;

// The running minima / maxima of a team's rows folded into the table.  Thread `lane` of the team holds in lo[k][e] /
// hi[k][e] the extremes of channel (k * TEAM + lane) * VEC + e over the rows it has seen (anything when the pack lies past
// the row's end: j >= npack).  A lane whose pack lies inside one slot first joins its neighbours of the same slot (slots
// are contiguous channel ranges, so equal slots sit in consecutive lanes of a wave: a segmented scan) and the last lane
// of every run issues the two atomics; a lane whose pack straddles slots publishes its channels one by one.
// lds: 32 floats, used (with block barriers: every thread of the block must be here) when the team spans the block and
// there is one slot only, so that one lane of the block publishes instead of one per wave.
template <int TEAM, int NV, int VEC>
__device__ __forceinline__ void publish_minmax(const RowStatSink& sink, const float (&lo)[NV][VEC], const float (&hi)[NV][VEC],
                                               int lane, int npack, int rep, unsigned tag, float* lds) {
    u64* const table = sink.ws + kSlotStride;
    const u64 tag_hi = (u64)tag << 32;
    const int wl = threadIdx.x & 63;
    auto publish = [&](int slot, float lo_v, float hi_v) {
        u64* q = table + ((long long)slot * kSlotReplicas + rep) * kSlotStride;
        atomicMax(q, tag_hi | min_key(lo_v));
        atomicMax(q + 1, tag_hi | max_key(hi_v));
    };
    const bool one_slot = !sink.slot_of && sink.packs_per_slot >= npack;
    if (one_slot) {  // one cluster over all channels: a plain reduction
        float v[2] = {__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k * TEAM + lane >= npack) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[0] = nmin(v[0], lo[k][e]); v[1] = nmax(v[1], hi[k][e]); }
        }
        team_reduce<TEAM, 2>(v, OpMinMax(), lds);
        if (lane == 0) publish(0, v[0], v[1]);
        return;
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * TEAM + lane;
        const bool valid = j < npack;
        float a = __builtin_inff(), b = -__builtin_inff();
        if (valid) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { a = nmin(a, lo[k][e]); b = nmax(b, hi[k][e]); }
        }
        int sl[VEC];
        bool uniform = true;
        if (sink.slot_of) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) sl[e] = valid ? sink.slot_of[j * VEC + e] : -1;
#pragma unroll
            for (int e = 1; e < VEC; ++e) uniform = uniform && sl[e] == sl[0];
        } else {
            sl[0] = j / sink.packs_per_slot;
        }
        const int key = valid ? (uniform ? sl[0] : -2 - wl) : -1;  // unique keys below -1 join nobody
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int k2 = __shfl_up(key, o);
            const float a2 = __shfl_up(a, o), b2 = __shfl_up(b, o);
            if (wl >= o && k2 == key) { a = nmin(a, a2); b = nmax(b, b2); }
        }
        const int knext = __shfl_down(key, 1);
        if (key >= 0) {
            if (wl == 63 || knext != key) publish(key, a, b);
        } else if (valid && !uniform) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) publish(sl[e], lo[k][e], hi[k][e]);
        }
    }
}

}  // namespace miq
