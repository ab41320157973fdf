// Static activation quantisation over the channel (last) axis of a [R, C] activation slab:
//   k_colstat        per-channel min/max over the R rows            (reads x once)
//   k_cluster_scale  combine with the cached statistics (running min/max or EMA), merge channels
//                    into contiguous clusters, derive scale / zero point per channel
//   k_colquant_fwd   quantize-dequantize with the per-channel parameters (reads x, writes out)
//   k_colquant_bwd   straight-through backward incl. the min/max routing
//   k_gather_*       RPTQ channel reorder: out[r, c] = x[r, idx[c]] staged through LDS
//
// Semantics: UniformAffineQuantizer.per_cluster_minmax_calibration / per_tensor_minmax_calibration /
// calibration / fake_quant (reference quantize/quantizer.py:157-238,99-121) and the index_select in
// OmniLayerNorm.forward (quantize/omni_norm.py:36-40).
#include <mutex>
#include <type_traits>
#include "miq_common.h"
#include "slot_table.h"

namespace miq {

// ---------------------------------------------------------------------------------------------
// per-channel min/max.  Block = 4 waves; lane l owns one 16-byte pack of columns, wave w owns every
// 4th row of the block's row chunk, so each wave-load is one contiguous 1 KiB line segment.
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_colstat(const T* __restrict__ x, long long R, int C, int rows_per_chunk, float* __restrict__ pmin,
          float* __restrict__ pmax) {
    __shared__ float smin[4][64 * VEC];
    __shared__ float smax[4][64 * VEC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    const bool valid = col0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float mn[VEC], mx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mn[e] = __builtin_inff(); mx[e] = -__builtin_inff(); }
    if (valid) {
        const T* p = x + col0;
        long long r = r0 + w;
        for (; r + 12 < r1; r += 16) {
            Pack<T, VEC> a = ld_pack<T, VEC>(p + r * C);
            Pack<T, VEC> b = ld_pack<T, VEC>(p + (r + 4) * C);
            Pack<T, VEC> c = ld_pack<T, VEC>(p + (r + 8) * C);
            Pack<T, VEC> d = ld_pack<T, VEC>(p + (r + 12) * C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float fa = to_f<T>(a.e[e]), fb = to_f<T>(b.e[e]), fc = to_f<T>(c.e[e]), fd = to_f<T>(d.e[e]);
                mn[e] = nmin(nmin(mn[e], fa), nmin(fb, nmin(fc, fd)));
                mx[e] = nmax(nmax(mx[e], fa), nmax(fb, nmax(fc, fd)));
            }
        }
        for (; r < r1; r += 4) {
            Pack<T, VEC> a = ld_pack<T, VEC>(p + r * C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float fa = to_f<T>(a.e[e]);
                mn[e] = nmin(mn[e], fa);
                mx[e] = nmax(mx[e], fa);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        smin[w][lane * VEC + e] = mn[e];
        smax[w][lane * VEC + e] = mx[e];
    }
    __syncthreads();
    if (w == 0 && valid) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int i = lane * VEC + e;
            const float a = nmin(nmin(smin[0][i], smin[1][i]), nmin(smin[2][i], smin[3][i]));
            const float b = nmax(nmax(smax[0][i], smax[1][i]), nmax(smax[2][i], smax[3][i]));
            pmin[(long long)blockIdx.y * C + col0 + e] = a;
            pmax[(long long)blockIdx.y * C + col0 + e] = b;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cluster merge + scale.  One block per cluster (plus one block for channels that belong to no
// cluster, which keep their own statistics).  `idx` (optional) maps an output channel to the source
// column whose statistics it takes (reorder fused into the statistics pass).
// ---------------------------------------------------------------------------------------------
template <typename T>
struct ScaleArgs {
    float* pmin;         // [nchunk + 1, C] partial minima; row nchunk receives the per-channel result
    float* pmax;
    int nchunk;
    int C;
    const int* idx;      // [C] or nullptr
    const T* cached_min; // [C] previous statistics or nullptr
    const T* cached_max;
    int metric;          // 0 running min/max, 1 EMA 0.99/0.01
    const int* starts;   // [ncl + 1] prefix sums of the cluster sizes
    int ncl;
    int n_bits;
    T* cmin_out;         // [C] statistics after combine + merge (what the caller caches)
    T* cmax_out;
    T* scale;            // [C]
    T* zp;               // [C] clamped, un-rounded zero point
    T* rzp;              // [C]
};

// step 1: fold the row-chunk partials of every output channel's source column and apply the
// running-min/max or EMA update against the cached statistics.  Block = 64 channels x 16 waves; wave w
// folds chunks w, w+16, ... (coalesced 256-byte reads, eight in flight): with the ~128 chunks of a
// [2048, 4096] slab that is one round of loads per wave instead of eight dependent ones.  LDS joins the waves.
constexpr int kCombineWaves = 16;
template <typename T>
__global__ void __launch_bounds__(64 * kCombineWaves) k_colstat_combine(ScaleArgs<T> a) {
    __shared__ float smn[kCombineWaves][64], smx[kCombineWaves][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const bool valid = c < a.C;
    float mn = __builtin_inff(), mx = -__builtin_inff();
    if (valid) {
        const int src = a.idx ? a.idx[c] : c;
        int j = w;
        for (; j + 3 * kCombineWaves < a.nchunk; j += 4 * kCombineWaves) {
            const float a0 = a.pmin[(long long)j * a.C + src], b0 = a.pmax[(long long)j * a.C + src];
            const float a1 = a.pmin[(long long)(j + kCombineWaves) * a.C + src], b1 = a.pmax[(long long)(j + kCombineWaves) * a.C + src];
            const float a2 = a.pmin[(long long)(j + 2 * kCombineWaves) * a.C + src], b2 = a.pmax[(long long)(j + 2 * kCombineWaves) * a.C + src];
            const float a3 = a.pmin[(long long)(j + 3 * kCombineWaves) * a.C + src], b3 = a.pmax[(long long)(j + 3 * kCombineWaves) * a.C + src];
            mn = nmin(nmin(mn, a0), nmin(a1, nmin(a2, a3)));
            mx = nmax(nmax(mx, b0), nmax(b1, nmax(b2, b3)));
        }
        for (; j < a.nchunk; j += kCombineWaves) {
            mn = nmin(mn, a.pmin[(long long)j * a.C + src]);
            mx = nmax(mx, a.pmax[(long long)j * a.C + src]);
        }
    }
    smn[w][lane] = mn;
    smx[w][lane] = mx;
    __syncthreads();
    if (w != 0 || !valid) return;
#pragma unroll
    for (int u = 1; u < kCombineWaves; ++u) {
        mn = nmin(mn, smn[u][lane]);
        mx = nmax(mx, smx[u][lane]);
    }
    if (a.cached_min) {
        const float pm = to_f<T>(a.cached_min[c]), px = to_f<T>(a.cached_max[c]);
        if (a.metric == 0) {
            mx = nmax(px, mx);
            mn = nmin(pm, mn);
        } else {
            mx = rnd<T>(rnd<T>(px * 0.99f) + rnd<T>(mx * 0.01f));
            mn = rnd<T>(rnd<T>(pm * 0.99f) + rnd<T>(mn * 0.01f));
        }
    }
    a.pmin[(long long)a.nchunk * a.C + c] = mn;
    a.pmax[(long long)a.nchunk * a.C + c] = mx;
}

// step 2: gridDim.y blocks per cluster (plus as many for the channels that belong to no cluster).  Every block
// of a cluster reduces the whole cluster (its per-channel statistics sit in L2; four loads in flight) and writes
// the parameters of its own slice of channels, so that a single large cluster (per-tensor statistics: all C
// channels) is not one block walking the channels 256 at a time.
template <typename T>
__global__ void __launch_bounds__(256) k_cluster_scale(ScaleArgs<T> a) {
    __shared__ float lds[32];
    const int k = blockIdx.x;
    const bool own = k >= a.ncl;  // leftover channels: every channel is its own cluster
    const int st = own ? a.starts[a.ncl] : a.starts[k];
    const int en = own ? a.C : a.starts[k + 1];
    const float* cmn = a.pmin + (long long)a.nchunk * a.C;
    const float* cmx = a.pmax + (long long)a.nchunk * a.C;
    float mm[2] = {__builtin_inff(), -__builtin_inff()};
    if (!own) {
        for (int c = st + threadIdx.x; c < en; c += 1024) {
            float vn[4], vx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = c + 256 * u;
                vn[u] = cc < en ? cmn[cc] : __builtin_inff();
                vx[u] = cc < en ? cmx[cc] : -__builtin_inff();
            }
            mm[0] = nmin(nmin(mm[0], vn[0]), nmin(vn[1], nmin(vn[2], vn[3])));
            mm[1] = nmax(nmax(mm[1], vx[0]), nmax(vx[1], nmax(vx[2], vx[3])));
        }
        team_reduce<256, 2>(mm, OpMinMax(), lds);
    }
    const float qmax = exp2f((float)a.n_bits) - 1.0f;
    const float lo = rnd<T>(kClipLo);
    for (int c = st + blockIdx.y * 256 + threadIdx.x; c < en; c += 256 * gridDim.y) {
        const float mn = (own ? cmn[c] : mm[0]) + 0.f;  // a zero extreme is +0 (see max_key)
        const float mx = (own ? cmx[c] : mm[1]) + 0.f;
        a.cmin_out[c] = from_f<T>(mn);
        a.cmax_out[c] = from_f<T>(mx);
        const float s = clampf(rnd<T>(rnd<T>(mx - mn) / qmax), lo, kClipHi);
        const float z = clampf(rnd<T>(-mn / s), -kClipHi, kClipHi);
        a.scale[c] = from_f<T>(s);
        a.zp[c] = from_f<T>(z);
        a.rzp[c] = from_f<T>(rintf(z));
    }
}

// ---------------------------------------------------------------------------------------------
// quantize-dequantize with per-channel parameters
// ---------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_colquant_fwd(const T* __restrict__ x, const T* __restrict__ scale, const T* __restrict__ rzp,
               T* __restrict__ out, uint8_t* __restrict__ codes, long long R, int C, int rows_per_chunk, float qmax,
               int stream_out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    if (col0 >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    // x is read for the last time here (its statistics, if any were taken, came from an earlier kernel).  fp32: the first
    // batch of rows is requested before the parameters, so that the two round trips overlap, and a wave of a 32-row
    // chunk has all of its rows in flight at once ([2048, 4096]: 17.5 -> 14.7 us).  fp16 packs (twice the registers
    // per row) keep four rows per batch behind the parameters: the early batch costs a wave per SIMD (10.7 -> 12.7 us).
    constexpr bool EARLY = sizeof(T) == 4;
    constexpr int DEPTH = EARLY ? 8 : 4;
    long long r = r0 + w;
    Pack<T, VEC> first[DEPTH];
    if constexpr (EARLY) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if (r + 4 * u < r1) first[u] = ld_pack_nt<T, VEC>(x + (r + 4 * u) * C + col0);
    }
    float s[VEC], z[VEC], rs[VEC];
    bool ok = true;
    {
        Pack<T, VEC> ps = ld_pack<T, VEC>(scale + col0), pz = ld_pack<T, VEC>(rzp + col0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            s[e] = to_f<T>(ps.e[e]);
            z[e] = to_f<T>(pz.e[e]);
            rs[e] = 1.0f / s[e];
            ok = ok && ints_exact<T>(z[e], qmax);
        }
    }
    // wave-uniform: every integer intermediate of this wave's channels is exact in T
    const bool exact_int = __all(ok);
    // the streaming loop is instantiated for both values of exact_int: a per-element test of it would
    // become a scalar branch around every fp16 re-rounding
    auto stream_rows = [&](auto exact_tag) {
        constexpr bool EX = decltype(exact_tag)::value;
        auto body = [&](const Pack<T, VEC>& a, long long rr) {
            Pack<T, VEC> o;
            Pack<uint8_t, VEC> cd;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xi = col_level<T>(to_f<T>(a.e[e]), s[e], rs[e], z[e], EX);
                const float xc = clampf(xi, 0.f, qmax);
                o.e[e] = from_f<T>(affine_dequant<T>(xc, s[e], z[e], EX));
                if (codes) cd.e[e] = (xc == xc) ? (uint8_t)xc : (uint8_t)0;
            }
            st_out<T, VEC>(out + rr * C + col0, o, stream_out);
            if (codes) st_out<uint8_t, VEC>(codes + rr * C + col0, cd, stream_out);
        };
        if constexpr (EARLY) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
                if (r + 4 * u < r1) body(first[u], r + 4 * u);
            r += 4 * DEPTH;
        }
        // whole and partial batches alike (the bounds are wave-uniform): a tail row at a time would be a round trip each
        for (; r < r1; r += 4 * DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
                if (r + 4 * u < r1) first[u] = ld_pack_nt<T, VEC>(x + (r + 4 * u) * C + col0);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
                if (r + 4 * u < r1) body(first[u], r + 4 * u);
        }
    };
    if (sizeof(T) == 4 || exact_int) stream_rows(std::true_type{});
    else stream_rows(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
// Fresh statistics (no cached values: every call of a training step): two kernels instead of four.
//
// The statistics of a cluster are a minimum / maximum over all of its elements, which is associative and
// commutative, so the row-chunk blocks combine them with integer atomic maxima on order-preserving keys: the result
// does not depend on the arrival order and equals the fold of per-channel partials bit for bit.  A "slot" is one
// statistics unit: cluster k for the channels of cluster k, n_clusters + (c - covered) for a channel c outside every
// cluster; slot_of[c] maps channels to slots.  The slot table in the workspace is never cleared: every 64-bit word
// carries, above the 32-bit key, a call number that grows with every call, so a word written by an earlier call
// always loses against the current one (the table only has to be zero when it is first used).
//   ws[0]  call number of the last completed call  (read by the statistics kernel, written by the quantize kernel)
//   ws[1]  call number of the call in flight       (written by the statistics kernel, read by the quantize kernel)
//   ws[16 + slab * table_words + 16 * (slot * nrep + rep) + {0, 1}]   {min, max} of replica rep of a slot of a slab
//                                                                      (one 128-byte line each)
// One pair of counters serves every slab and every layout the workspace is used with: whatever an earlier call left
// anywhere in the table carries a smaller call number.
// Replicas spread the atomics of a slot that many blocks feed (one cluster spanning all channels: every block).
// ---------------------------------------------------------------------------------------------
// Statistics pass.  Block = 4 waves; lane l owns one 16-byte pack of channels, wave w the rows r0 + w, r0 + w + 4, ...
// of the block's row chunk with DEPTH rows in flight (every wave-load is one contiguous 1 KiB segment; rows past the
// chunk's end are replaced by its last row, which changes no minimum or maximum).  blockIdx.z selects the slab; slab i
// uses the table at ws + 16 + i * ws_stride.  UNIFORM: every lane's pack lies inside one slot (equal cluster sizes that
// are multiples of VEC, no channel outside a cluster), so the pack is folded to one pair before the waves meet.
template <typename T, int VEC, int DEPTH, bool UNIFORM>
__global__ void __launch_bounds__(256)
k_colstat_slots(SlabPtrs xs, long long R, int C, int rows_per_chunk, SlotMap sm, u64* ws_base, long long ws_stride, int nrep) {
    constexpr int NP = UNIFORM ? 1 : VEC;  // pairs a lane hands to wave 0
    __shared__ float smin[3][64 * NP];
    __shared__ float smax[3][64 * NP];
    const T* __restrict__ x = static_cast<const T*>(xs.p[blockIdx.z]);
    u64* ws = ws_base;  // the two call counters, shared by all slabs
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    const bool valid = col0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    unsigned tag = 0;
    if (w == 0) tag = (unsigned)ws[0] + 1u;  // requested ahead of the rows: it has arrived when they have
    float mn[VEC], mx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mn[e] = __builtin_inff(); mx[e] = -__builtin_inff(); }
    if (valid) {
        const T* p = x + col0;
        for (long long r = r0 + w; r < r1; r += DEPTH * 4) {
            Pack<T, VEC> a[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const long long ru = r + u * 4;
                a[u] = ld_pack<T, VEC>(p + (ru < r1 ? ru : r1 - 1) * C);
            }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float f = to_f<T>(a[u].e[e]);
                    mn[e] = nmin(mn[e], f);
                    mx[e] = nmax(mx[e], f);
                }
        }
    }
    if (UNIFORM) {
#pragma unroll
        for (int e = 1; e < VEC; ++e) { mn[0] = nmin(mn[0], mn[e]); mx[0] = nmax(mx[0], mx[e]); }
    }
    if (w > 0) {
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            smin[w - 1][lane * NP + e] = mn[e];
            smax[w - 1][lane * NP + e] = mx[e];
        }
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int e = 0; e < NP; ++e)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            mn[e] = nmin(mn[e], smin[u][lane * NP + e]);
            mx[e] = nmax(mx[e], smax[u][lane * NP + e]);
        }
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) ws[1] = tag;
    u64* const table = ws + kSlotStride + blockIdx.z * ws_stride;
    const int rep = blockIdx.y % nrep;
    const u64 hi = (u64)tag << 32;
    auto publish = [&](int slot, float lo_v, float hi_v) {
        u64* q = table + ((long long)slot * nrep + rep) * kSlotStride;
        atomicMax(q, hi | min_key(lo_v));
        atomicMax(q + 1, hi | max_key(hi_v));
    };
    // A lane whose channels all belong to one slot first joins its neighbours of the same slot (slots are contiguous
    // channel ranges, so equal slots sit in consecutive lanes: a segmented scan); the last lane of every run issues the
    // two atomics.  A lane that straddles slots publishes its channels one by one.
    int sl[NP];
    bool uniform = valid;
    if (UNIFORM) {
        sl[0] = valid ? (int)(col0 / sm.csize) : -1;
    } else {
#pragma unroll
        for (int e = 0; e < NP; ++e) sl[e] = valid ? sm(col0 + e) : -1;
#pragma unroll
        for (int e = 1; e < NP; ++e) uniform = uniform && (sl[e] == sl[0]);
    }
    float lo_v = mn[0], hi_v = mx[0];
#pragma unroll
    for (int e = 1; e < NP; ++e) { lo_v = nmin(lo_v, mn[e]); hi_v = nmax(hi_v, mx[e]); }
    const int key = uniform ? sl[0] : -2 - lane;  // unique negative key: joins nobody
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int k2 = __shfl_up(key, o);
        const float a = __shfl_up(lo_v, o), b = __shfl_up(hi_v, o);
        if (lane >= o && k2 == key) { lo_v = nmin(lo_v, a); hi_v = nmax(hi_v, b); }
    }
    const int knext = __shfl_down(key, 1);
    if (uniform) {
        if (lane == 63 || knext != key) publish(key, lo_v, hi_v);
    } else if (valid) {
#pragma unroll
        for (int e = 0; e < NP; ++e) publish(sl[e], mn[e], mx[e]);
    }
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// the {min, max} words of the NREP replicas of a slot, requested together; finish() folds the ones of the current call
// and derives the reference's scale / zero point
template <typename T, int NREP>
struct SlotParams {
    u64x2 raw[NREP];
    float mn, mx, s, zp, z, rs;
    __device__ __forceinline__ void request(const u64* table, int slot) {
#pragma unroll
        for (int rep = 0; rep < NREP; ++rep)
            raw[rep] = *reinterpret_cast<const u64x2*>(table + ((long long)slot * NREP + rep) * kSlotStride);
    }
    __device__ __forceinline__ void finish(unsigned tag, float qmax) {
        unsigned klo = 0, khi = 0;  // keys of real values are never 0
#pragma unroll
        for (int rep = 0; rep < NREP; ++rep) {
            if ((unsigned)(raw[rep].x >> 32) == tag && (unsigned)raw[rep].x > klo) klo = (unsigned)raw[rep].x;
            if ((unsigned)(raw[rep].y >> 32) == tag && (unsigned)raw[rep].y > khi) khi = (unsigned)raw[rep].y;
        }
        mn = min_key_value(klo);
        mx = max_key_value(khi);
        s = clampf(rnd<T>(rnd<T>(mx - mn) / qmax), rnd<T>(kClipLo), kClipHi);
        zp = clampf(rnd<T>(-mn / s), -kClipHi, kClipHi);
        z = rintf(zp);
        rs = 1.0f / s;
    }
};

// quantize-dequantize with the parameters derived from the slot table; the blocks of row chunk 0 also write the
// [5, C] vectors the caller keeps (merged min, merged max, scale, zero point, rounded zero point).  Lane / wave / row
// mapping of k_colstat_slots.  UNIFORM: every lane's pack lies inside one slot (equal cluster sizes that are multiples
// of VEC, no channel outside a cluster): one parameter set per lane.  The first rows are requested between the table
// loads and their use, so the parameter chain costs no extra memory round trip.
template <typename T, int VEC, int DEPTH, int NREP, bool UNIFORM>
__global__ void __launch_bounds__(256)
k_colquant_slots(SlabPtrs xs, SlotMap sm, u64* ws_base, long long ws_stride, SlabOuts outs, SlabOuts codes_tab, SlabOuts vecs,
                 long long R, int C, int rows_per_chunk, float qmax, int stream_out) {
    constexpr int NP = UNIFORM ? 1 : VEC;  // parameter sets per lane
    const T* __restrict__ x = static_cast<const T*>(xs.p[blockIdx.z]);
    T* __restrict__ out = static_cast<T*>(outs.p[blockIdx.z]);
    uint8_t* __restrict__ codes = static_cast<uint8_t*>(codes_tab.p[blockIdx.z]);
    T* __restrict__ vec = static_cast<T*>(vecs.p[blockIdx.z]);
    u64* ws = ws_base;
    const u64* table = ws + kSlotStride + blockIdx.z * ws_stride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    if (col0 >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    const unsigned tag = (unsigned)ws[1];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ws[0] = tag;
    SlotParams<T, NREP> sp[NP];
    // (general form: a lane whose channels share one slot, i.e. nearly every lane, still asks for one parameter set)
    int sl[NP];
    bool lane_uniform = true;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        sl[k] = sm(col0 + k);
        lane_uniform = lane_uniform && sl[k] == sl[0];
    }
    if (lane_uniform) {
        sp[0].request(table, sl[0]);
    } else {
#pragma unroll
        for (int k = 0; k < NP; ++k) sp[k].request(table, sl[k]);
    }
    long long r = r0 + w;
    Pack<T, VEC> first[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u)
        if (r + 4 * u < r1) first[u] = ld_pack_nt<T, VEC>(x + (r + 4 * u) * C + col0);
    float s[NP], z[NP], rs[NP];
    bool ok = true;
    {
        Pack<T, VEC> v0, v1, v2, v3, v4;
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            if (k == 0 || !lane_uniform) sp[k].finish(tag, qmax);
            else sp[k] = sp[0];
            s[k] = sp[k].s; z[k] = sp[k].z; rs[k] = sp[k].rs;
            ok = ok && ints_exact<T>(sp[k].z, qmax);
#pragma unroll
            for (int e = (UNIFORM ? 0 : k); e < (UNIFORM ? VEC : k + 1); ++e) {
                v0.e[e] = from_f<T>(sp[k].mn); v1.e[e] = from_f<T>(sp[k].mx); v2.e[e] = from_f<T>(sp[k].s);
                v3.e[e] = from_f<T>(sp[k].zp); v4.e[e] = from_f<T>(sp[k].z);
            }
        }
        if (blockIdx.y == 0 && w == 0) {
            st_pack<T, VEC>(vec + col0, v0); st_pack<T, VEC>(vec + C + col0, v1); st_pack<T, VEC>(vec + 2ll * C + col0, v2);
            st_pack<T, VEC>(vec + 3ll * C + col0, v3); st_pack<T, VEC>(vec + 4ll * C + col0, v4);
        }
    }
    const bool exact_int = __all(ok);
    auto stream_rows = [&](auto exact_tag) {
        constexpr bool EX = decltype(exact_tag)::value;
        auto body = [&](const Pack<T, VEC>& v, long long rr) {
            Pack<T, VEC> o;
            Pack<uint8_t, VEC> cd;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int k = UNIFORM ? 0 : e;
                const float xi = col_level<T>(to_f<T>(v.e[e]), s[k], rs[k], z[k], EX);
                const float xc = clampf(xi, 0.f, qmax);
                o.e[e] = from_f<T>(affine_dequant<T>(xc, s[k], z[k], EX));
                if (codes) cd.e[e] = (xc == xc) ? (uint8_t)xc : (uint8_t)0;
            }
            st_out<T, VEC>(out + rr * C + col0, o, stream_out);
            if (codes) st_out<uint8_t, VEC>(codes + rr * C + col0, cd, stream_out);
        };
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
            if (r + 4 * u < r1) body(first[u], r + 4 * u);
        r += 4 * DEPTH;
        for (; r + (DEPTH - 1) * 4 < r1; r += DEPTH * 4) {
            Pack<T, VEC> a[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) a[u] = ld_pack_nt<T, VEC>(x + (r + 4 * u) * C + col0);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) body(a[u], r + 4 * u);
        }
        for (; r < r1; r += 4) body(ld_pack_nt<T, VEC>(x + r * C + col0), r);
    };
    if (sizeof(T) == 4 || exact_int) stream_rows(std::true_type{});
    else stream_rows(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
// channel gather out[r, c] = x[r, idx[c]]: rows are staged in LDS with coalesced 16-byte loads and
// read back through the index, stores are coalesced again.
// ---------------------------------------------------------------------------------------------
// up to kMaxSlabs tensors of one shape per launch (blockIdx.y), each with its own index
struct GatherSlabs {
    const void* x[kMaxSlabs];
    const int* idx[kMaxSlabs];
    void* out[kMaxSlabs];
};

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_gather(GatherSlabs slabs, long long R, int C, int rows_per_block, int stream_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);  // [rows_per_block][C]
    const T* __restrict__ x = static_cast<const T*>(slabs.x[blockIdx.y]);
    const int* __restrict__ idx = slabs.idx[blockIdx.y];
    T* __restrict__ out = static_cast<T*>(slabs.out[blockIdx.y]);
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    int nr = rows_per_block;
    if (r0 + nr > R) nr = (int)(R - r0);
    const int npack = C / VEC;
    const long long total = (long long)nr * npack;
    const T* src = x + r0 * C;  // the nr rows are contiguous in memory
    for (long long i = threadIdx.x; i < total; i += 256)
        st_pack<T, VEC>(tile + i * VEC, ld_pack_nt<T, VEC>(src + i * VEC));
    __syncthreads();
    T* dst = out + r0 * C;
    for (int p = threadIdx.x; p < npack; p += 256) {
        int id[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) id[e] = idx[p * VEC + e];
        for (int rr = 0; rr < nr; ++rr) {
            Pack<T, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.e[e] = tile[(long long)rr * C + id[e]];
            st_out<T, VEC>(dst + (long long)rr * C + (long long)p * VEC, o, stream_out);
        }
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr long long kFreshBlocks = 2048;  // block target per slab of the fresh-statistics forward when a launch holds several slabs

static inline int pick_chunks(long long R, int col_tiles, long long target = 1024) {
    // aim at ~target blocks (1024: 4 per CU), at least 16 rows per chunk
    long long want = target / (col_tiles > 0 ? col_tiles : 1);
    if (want < 1) want = 1;
    long long maxc = (R + 15) / 16;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename T>
static int launch_colstat(const void* x, long long R, long long C, float* pmin, float* pmax, int nchunk,
                          hipStream_t stream) {
    constexpr int N = VecOf<T>::N;
    const bool vec_ok = (C % N == 0) && aligned16(x);
    const int rows_per_chunk = (int)((R + nchunk - 1) / nchunk);
    if (vec_ok) {
        dim3 grid((unsigned)((C / N + 63) / 64), (unsigned)nchunk);
        hipLaunchKernelGGL((k_colstat<T, N>), grid, dim3(256), 0, stream, (const T*)x, R, (int)C, rows_per_chunk, pmin, pmax);
    } else {
        dim3 grid((unsigned)((C + 63) / 64), (unsigned)nchunk);
        hipLaunchKernelGGL((k_colstat<T, 1>), grid, dim3(256), 0, stream, (const T*)x, R, (int)C, rows_per_chunk, pmin, pmax);
    }
    MIQ_CHECK_LAUNCH("miq_colstat_minmax");
    return 0;
}

template <typename T>
static int launch_colquant_fwd(const void* x, const void* scale, const void* rzp, void* out, void* codes,
                               long long R, long long C, int n_bits, hipStream_t stream) {
    constexpr int N = VecOf<T>::N;
    const bool vec_ok = (C % N == 0) && aligned16(x) && aligned16(out) && aligned16(scale) && aligned16(rzp) &&
                        (!codes || (reinterpret_cast<uintptr_t>(codes) % N) == 0);
    const float qmax = (float)((1 << n_bits) - 1);
    const int so = stream_out_of(R * C * (long long)sizeof(T));
    if (vec_ok) {
        const int tiles = (int)((C / N + 63) / 64);
        const int nchunk = pick_chunks(R, tiles);
        const int rpc = (int)((R + nchunk - 1) / nchunk);
        hipLaunchKernelGGL((k_colquant_fwd<T, N>), dim3(tiles, nchunk), dim3(256), 0, stream, (const T*)x,
                           (const T*)scale, (const T*)rzp, (T*)out, (uint8_t*)codes, R, (int)C, rpc, qmax, so);
    } else {
        const int tiles = (int)((C + 63) / 64);
        const int nchunk = pick_chunks(R, tiles);
        const int rpc = (int)((R + nchunk - 1) / nchunk);
        hipLaunchKernelGGL((k_colquant_fwd<T, 1>), dim3(tiles, nchunk), dim3(256), 0, stream, (const T*)x,
                           (const T*)scale, (const T*)rzp, (T*)out, (uint8_t*)codes, R, (int)C, rpc, qmax, so);
    }
    MIQ_CHECK_LAUNCH("miq_colquant_fwd");
    return 0;
}

template <typename T>
static int launch_gather(int n, const void* const* x, const int* const* idx, void* const* out, long long R, long long C,
                         hipStream_t stream) {
    constexpr int N = VecOf<T>::N;
    GatherSlabs slabs{};
    bool vec_ok = (C % N == 0);
    for (int i = 0; i < n; ++i) {
        MIQ_CHECK_ARG(x[i] && idx[i] && out[i], "miq_gather_lastdim: null pointer in tensor %d", i);
        slabs.x[i] = x[i];
        slabs.idx[i] = idx[i];
        slabs.out[i] = out[i];
        vec_ok = vec_ok && aligned16(x[i]) && aligned16(out[i]);
    }
    const long long row_bytes = C * (long long)sizeof(T);
    MIQ_CHECK_ARG(row_bytes <= 160 * 1024, "miq_gather_lastdim: a row of %lld bytes does not fit in LDS", row_bytes);
    int rows = (int)(65536 / row_bytes);
    if (rows < 1) rows = 1;
    if (rows > 16) rows = 16;
    // at least ~1024 blocks when the tensors have that many rows: several small tiles per CU stream better than one big
    // one; ~2048 for a batch (two [2048, 4096] fp16 tensors: 22.6 us with 2 rows per block against 23.5 with 4, 25.3 with 1)
    const long long want_rows = R * n / (n >= 2 ? 2048 : 1024);
    if (rows > want_rows) rows = (int)(want_rows < 1 ? 1 : want_rows);
    if (rows > R) rows = (int)R;
    const size_t lds = (size_t)rows * row_bytes;
    const dim3 grid((unsigned)((R + rows - 1) / rows), (unsigned)n);
    const int so = stream_out_of(R * row_bytes * n);
    if (vec_ok) {
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_gather<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_gather<T, N>), grid, dim3(256), lds, stream, slabs, R, (int)C, rows, so);
    } else {
        if (lds > 65536) (void)hipFuncSetAttribute((const void*)k_gather<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_gather<T, 1>), grid, dim3(256), lds, stream, slabs, R, (int)C, rows, so);
    }
    MIQ_CHECK_LAUNCH("miq_gather_lastdim");
    return 0;
}

}  // namespace miq

using namespace miq;

extern "C" int miq_colstat_chunks(int64_t R, int64_t C, int dtype) {
    if (R <= 0 || C <= 0) return 1;
    const int N = dtype == MIQ_F16 ? 8 : 4;
    const int tiles = (int)((C % N == 0) ? (C / N + 63) / 64 : (C + 63) / 64);
    return pick_chunks(R, tiles);
}

extern "C" int miq_colstat_minmax(const void* x, int64_t R, int64_t C, float* part_min, float* part_max,
                                  int nchunk, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(x && part_min && part_max && R > 0 && C > 0 && C < (1ll << 31) && nchunk >= 1,
                  "miq_colstat_minmax: null pointer or bad shape (R=%lld C=%lld nchunk=%d)", (long long)R, (long long)C, nchunk);
    if (dtype == MIQ_F32) return launch_colstat<float>(x, R, C, part_min, part_max, nchunk, (hipStream_t)stream);
    if (dtype == MIQ_F16) return launch_colstat<__half>(x, R, C, part_min, part_max, nchunk, (hipStream_t)stream);
    miq_set_error("miq_colstat_minmax: unsupported dtype %d", dtype);
    return -22;
}

template <typename T>
static int cluster_scale_t(float* pmin, float* pmax, int nchunk, int64_t C, const int* idx,
                           const void* cached_min, const void* cached_max, int metric, const int* starts, int ncl,
                           int covered, int n_bits, void* cmin_out, void* cmax_out, void* scale, void* zp, void* rzp,
                           hipStream_t stream) {
    ScaleArgs<T> a;
    a.pmin = pmin; a.pmax = pmax; a.nchunk = nchunk; a.C = (int)C; a.idx = idx;
    a.cached_min = (const T*)cached_min; a.cached_max = (const T*)cached_max; a.metric = metric;
    a.starts = starts; a.ncl = ncl; a.n_bits = n_bits;
    a.cmin_out = (T*)cmin_out; a.cmax_out = (T*)cmax_out; a.scale = (T*)scale; a.zp = (T*)zp; a.rzp = (T*)rzp;
    const unsigned grid = (unsigned)(ncl + (covered < C ? 1 : 0));
    if (grid == 0) return 0;
    hipLaunchKernelGGL((k_colstat_combine<T>), dim3((unsigned)((C + 63) / 64)), dim3(64 * kCombineWaves), 0, stream, a);
    hipLaunchKernelGGL((k_cluster_scale<T>), dim3(grid, cluster_slices(C, grid)), dim3(256), 0, stream, a);
    MIQ_CHECK_LAUNCH("miq_cluster_scale");
    return 0;
}

extern "C" int miq_cluster_scale(float* part_min, float* part_max, int nchunk, int64_t C, const int* idx,
                                 const void* cached_min, const void* cached_max, int metric, const int* starts,
                                 int n_clusters, int64_t covered, int n_bits, void* cmin_out, void* cmax_out,
                                 void* scale_out, void* zp_out, void* rzp_out, int dtype, void* stream) {
    if (C == 0) return 0;
    MIQ_CHECK_ARG(part_min && part_max && starts && cmin_out && cmax_out && scale_out && zp_out && rzp_out,
                  "miq_cluster_scale: null pointer");
    MIQ_CHECK_ARG((cached_min == nullptr) == (cached_max == nullptr), "miq_cluster_scale: cached min/max must come together");
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 15 && (metric == 0 || metric == 1) && n_clusters >= 0 && covered >= 0 && covered <= C,
                  "miq_cluster_scale: bad n_bits/metric/cluster layout");
    if (dtype == MIQ_F32)
        return cluster_scale_t<float>(part_min, part_max, nchunk, C, idx, cached_min, cached_max, metric, starts,
                                      n_clusters, (int)covered, n_bits, cmin_out, cmax_out, scale_out, zp_out, rzp_out, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return cluster_scale_t<__half>(part_min, part_max, nchunk, C, idx, cached_min, cached_max, metric, starts,
                                       n_clusters, (int)covered, n_bits, cmin_out, cmax_out, scale_out, zp_out, rzp_out, (hipStream_t)stream);
    miq_set_error("miq_cluster_scale: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_colquant_fwd(const void* x, const void* scale, const void* rzp, void* out, void* codes_out,
                                int64_t R, int64_t C, int n_bits, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(x && scale && rzp && out && R > 0 && C > 0 && C < (1ll << 31), "miq_colquant_fwd: null pointer or bad shape");
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 15 && (!codes_out || n_bits <= 8), "miq_colquant_fwd: n_bits=%d unsupported", n_bits);
    if (dtype == MIQ_F32) return launch_colquant_fwd<float>(x, scale, rzp, out, codes_out, R, C, n_bits, (hipStream_t)stream);
    if (dtype == MIQ_F16) return launch_colquant_fwd<__half>(x, scale, rzp, out, codes_out, R, C, n_bits, (hipStream_t)stream);
    miq_set_error("miq_colquant_fwd: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int64_t miq_act_workspace(int64_t R, int64_t C, int dtype) {
    if (R <= 0 || C <= 0) return 0;
    return 2ll * (miq_colstat_chunks(R, C, dtype) + 1) * C;
}

extern "C" int miq_act_calib_quant_fwd(const void* x, const void* cached_min, const void* cached_max, int metric,
                                       const int* starts, int n_clusters, int64_t covered, int n_bits, float* workspace,
                                       void* out, void* codes_out, void* cmin_out, void* cmax_out, void* scale_out,
                                       void* zp_out, void* rzp_out, int64_t R, int64_t C, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(workspace, "miq_act_calib_quant_fwd: null workspace");
    const int nchunk = miq_colstat_chunks(R, C, dtype);
    float* pmin = workspace;
    float* pmax = workspace + (int64_t)(nchunk + 1) * C;
    int rc = miq_colstat_minmax(x, R, C, pmin, pmax, nchunk, dtype, stream);
    if (rc) return rc;
    rc = miq_cluster_scale(pmin, pmax, nchunk, C, nullptr, cached_min, cached_max, metric, starts, n_clusters, covered, n_bits,
                           cmin_out, cmax_out, scale_out, zp_out, rzp_out, dtype, stream);
    if (rc) return rc;
    return miq_colquant_fwd(x, scale_out, rzp_out, out, codes_out, R, C, n_bits, dtype, stream);
}

// ---------------------------------------------------------------------------------------------
// fresh statistics: statistics kernel + quantize kernel over up to kMaxSlabs slabs of one shape
// ---------------------------------------------------------------------------------------------

extern "C" int64_t miq_act_fresh_workspace(int n, int64_t C, int n_clusters, int64_t covered) {
    if (n <= 0 || C <= 0 || n_clusters < 0 || covered < 0 || covered > C) return 0;
    return (kSlotStride + (int64_t)n * fresh_table_words(fresh_slots(C, n_clusters, covered))) * (int64_t)sizeof(u64);
}

template <typename T, int VEC, int NREP>
static void launch_fresh_quant(bool uniform, dim3 grid, hipStream_t stream, const SlabPtrs& xs, const SlotMap& sm, u64* ws,
                               long long ws_stride, const SlabOuts& outs, const SlabOuts& codes, const SlabOuts& vecs, long long R,
                               int C, int rpc, float qmax) {
    const int stream_out = stream_out_of(R * C * (long long)sizeof(T) * grid.z);  // all slabs of the launch
    // a single fp32 slab is short of loads in flight with four rows per wave (25 -> 21 us for [2048, 4096]); with two or
    // three slabs per launch the larger batch only costs occupancy (21 -> 22 us per slab)
    if (uniform && sizeof(T) == 4 && grid.z == 1)
        hipLaunchKernelGGL((k_colquant_slots<T, VEC, 8, NREP, true>), grid, dim3(256), 0, stream, xs, sm, ws, ws_stride, outs, codes,
                           vecs, R, C, rpc, qmax, stream_out);
    else if (uniform)
        hipLaunchKernelGGL((k_colquant_slots<T, VEC, 4, NREP, true>), grid, dim3(256), 0, stream, xs, sm, ws, ws_stride, outs, codes,
                           vecs, R, C, rpc, qmax, stream_out);
    else
        hipLaunchKernelGGL((k_colquant_slots<T, VEC, 4, NREP, false>), grid, dim3(256), 0, stream, xs, sm, ws, ws_stride, outs, codes,
                           vecs, R, C, rpc, qmax, stream_out);
}

template <typename T>
static int act_fresh_t(int n, const void* const* x, int64_t cluster_size, const int* slot_of, int n_clusters, int64_t covered,
                       int n_bits, void* workspace, void* const* out, void* const* codes_out, void* const* stats_out, long long R,
                       long long C, hipStream_t stream) {
    constexpr int N = VecOf<T>::N;
    SlabPtrs xs{};
    SlabOuts outs{}, codes{}, vecs{};
    bool vec_ok = (C % N == 0);
    for (int i = 0; i < n; ++i) {
        MIQ_CHECK_ARG(x[i] && out[i] && stats_out[i], "miq_act_fresh_quant_fwd: null pointer in slab %d", i);
        xs.p[i] = x[i];
        outs.p[i] = out[i];
        vecs.p[i] = stats_out[i];
        codes.p[i] = codes_out ? codes_out[i] : nullptr;
        vec_ok = vec_ok && aligned16(x[i]) && aligned16(out[i]) && aligned16(stats_out[i]) &&
                 (!codes.p[i] || (reinterpret_cast<uintptr_t>(codes.p[i]) % N) == 0);
    }
    const int vec = vec_ok ? N : 1;
    const int tiles = (int)((C / vec + 63) / 64);
    // The slot table decouples the two grids from every other kernel's partials, so a launch over several slabs takes
    // twice the blocks per slab, in row chunks that are whole 16-row batches (measured on the Llama-2-7B sites, policies
    // side by side on one GPU: three fp32 [2048, 4096] slabs +6 %, two +4 %, three fp16 [2048, 11008] slabs +2 %, fp16
    // [2048, 4096] unchanged; a single slab loses 2 % and keeps the 1024-block grid).
    int nchunk = pick_chunks(R, tiles, n >= 2 ? kFreshBlocks : 1024);
    int rpc = (int)((R + nchunk - 1) / nchunk);
    if (n >= 2) {
        rpc = (rpc + 15) / 16 * 16;
        nchunk = (int)((R + rpc - 1) / rpc);
    }
    const int64_t nslot = fresh_slots(C, n_clusters, covered);
    const long long ws_stride = fresh_table_words(nslot);
    // every row chunk feeds every slot its column tile touches: spread them over replicas when there are many
    const int nrep = nchunk > 16 ? kSlotReplicas : 1;
    const SlotMap sm{cluster_size > 0 ? nullptr : slot_of, (int)cluster_size, (int)covered, n_clusters};
    const bool uniform = vec == 1 || (cluster_size > 0 && cluster_size % vec == 0 && covered == C);
    const float qmax = (float)((1 << n_bits) - 1);
    const dim3 grid(tiles, nchunk, n);
    u64* ws = static_cast<u64*>(workspace);
    if (vec_ok) {
        // fp32 keeps 8 rows per lane in flight, fp16 (twice the channels per pack, twice the conversions) 4
        constexpr int D = sizeof(T) == 4 ? 8 : 4;
        if (uniform) hipLaunchKernelGGL((k_colstat_slots<T, N, D, true>), grid, dim3(256), 0, stream, xs, R, (int)C, rpc, sm, ws, ws_stride, nrep);
        else hipLaunchKernelGGL((k_colstat_slots<T, N, D, false>), grid, dim3(256), 0, stream, xs, R, (int)C, rpc, sm, ws, ws_stride, nrep);
        if (nrep == 1) launch_fresh_quant<T, N, 1>(uniform, grid, stream, xs, sm, ws, ws_stride, outs, codes, vecs, R, (int)C, rpc, qmax);
        else launch_fresh_quant<T, N, kSlotReplicas>(uniform, grid, stream, xs, sm, ws, ws_stride, outs, codes, vecs, R, (int)C, rpc, qmax);
    } else {
        hipLaunchKernelGGL((k_colstat_slots<T, 1, 4, false>), grid, dim3(256), 0, stream, xs, R, (int)C, rpc, sm, ws, ws_stride, nrep);
        if (nrep == 1) launch_fresh_quant<T, 1, 1>(true, grid, stream, xs, sm, ws, ws_stride, outs, codes, vecs, R, (int)C, rpc, qmax);
        else launch_fresh_quant<T, 1, kSlotReplicas>(true, grid, stream, xs, sm, ws, ws_stride, outs, codes, vecs, R, (int)C, rpc, qmax);
    }
    MIQ_CHECK_LAUNCH("miq_act_fresh_quant_fwd");
    return 0;
}

// The quantize half alone, for one slab whose statistics a producer other than k_colstat_slots has published in slab 0
// of `workspace` with kSlotReplicas replicas (the norm kernels do, for their own output).
template <typename T>
static int act_from_table_t(const void* x, int64_t cluster_size, const int* slot_of, int n_clusters, int64_t covered, int n_bits,
                            void* workspace, void* out, void* codes_out, void* stats_out, long long R, long long C,
                            hipStream_t stream) {
    constexpr int N = VecOf<T>::N;
    MIQ_CHECK_ARG(C % N == 0 && aligned16(x) && aligned16(out) && aligned16(stats_out) &&
                      (!codes_out || (reinterpret_cast<uintptr_t>(codes_out) % N) == 0),
                  "miq_act_quant_from_table: C must be a multiple of %d and the pointers 16-byte aligned", N);
    SlabPtrs xs{};
    SlabOuts outs{}, codes{}, vecs{};
    xs.p[0] = x; outs.p[0] = out; codes.p[0] = codes_out; vecs.p[0] = stats_out;
    const int tiles = (int)((C / N + 63) / 64);
    const int nchunk = pick_chunks(R, tiles);
    const int rpc = (int)((R + nchunk - 1) / nchunk);
    const SlotMap sm{cluster_size > 0 ? nullptr : slot_of, (int)cluster_size, (int)covered, n_clusters};
    const long long ws_stride = fresh_table_words(fresh_slots(C, n_clusters, covered));
    const bool uniform = cluster_size > 0 && cluster_size % N == 0 && covered == C;
    const float qmax = (float)((1 << n_bits) - 1);
    launch_fresh_quant<T, N, kSlotReplicas>(uniform, dim3(tiles, nchunk, 1), stream, xs, sm, static_cast<u64*>(workspace), ws_stride,
                                            outs, codes, vecs, R, (int)C, rpc, qmax);
    MIQ_CHECK_LAUNCH("miq_act_quant_from_table");
    return 0;
}

extern "C" int miq_act_quant_from_table(const void* x, int64_t cluster_size, const int* slot_of, int n_clusters, int64_t covered,
                                        int n_bits, void* workspace, void* out, void* codes_out, void* stats_out, int64_t R,
                                        int64_t C, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(x && out && stats_out && workspace && R > 0 && C > 0 && C < (1ll << 31),
                  "miq_act_quant_from_table: null pointer or bad shape (R=%lld C=%lld)", (long long)R, (long long)C);
    MIQ_CHECK_ARG(n_clusters >= 0 && covered >= 0 && covered <= C && (cluster_size > 0 || slot_of || n_clusters == 0),
                  "miq_act_quant_from_table: unequal clusters need the channel -> slot map");
    MIQ_CHECK_ARG(cluster_size <= 0 || cluster_size * n_clusters == covered, "miq_act_quant_from_table: cluster_size * n_clusters != covered");
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 15 && (!codes_out || n_bits <= 8), "miq_act_quant_from_table: n_bits=%d unsupported", n_bits);
    MIQ_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 127u) == 0, "miq_act_quant_from_table: workspace must be 128-byte aligned");
    if (dtype == MIQ_F32)
        return act_from_table_t<float>(x, cluster_size, slot_of, n_clusters, covered, n_bits, workspace, out, codes_out, stats_out, R, C,
                                       (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return act_from_table_t<__half>(x, cluster_size, slot_of, n_clusters, covered, n_bits, workspace, out, codes_out, stats_out, R, C,
                                        (hipStream_t)stream);
    miq_set_error("miq_act_quant_from_table: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_act_fresh_quant_fwd(int n, const void* const* x, int64_t cluster_size, const int* slot_of, int n_clusters,
                                       int64_t covered, int n_bits, void* workspace, void* const* out, void* const* codes_out,
                                       void* const* stats_out, int64_t R, int64_t C, int dtype, void* stream) {
    if (n == 0 || R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(n > 0 && n <= kMaxSlabs, "miq_act_fresh_quant_fwd: %d slabs (1..%d per call)", n, kMaxSlabs);
    MIQ_CHECK_ARG(x && out && stats_out && workspace && R > 0 && C > 0 && C < (1ll << 31),
                  "miq_act_fresh_quant_fwd: null pointer or bad shape (R=%lld C=%lld)", (long long)R, (long long)C);
    MIQ_CHECK_ARG(n_clusters >= 0 && covered >= 0 && covered <= C && (cluster_size > 0 || slot_of || n_clusters == 0),
                  "miq_act_fresh_quant_fwd: unequal clusters need the channel -> slot map");
    MIQ_CHECK_ARG(cluster_size <= 0 || cluster_size * n_clusters == covered, "miq_act_fresh_quant_fwd: cluster_size * n_clusters != covered");
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 15 && (!codes_out || n_bits <= 8), "miq_act_fresh_quant_fwd: n_bits=%d unsupported", n_bits);
    MIQ_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 127u) == 0, "miq_act_fresh_quant_fwd: workspace must be 128-byte aligned");
    if (dtype == MIQ_F32)
        return act_fresh_t<float>(n, x, cluster_size, slot_of, n_clusters, covered, n_bits, workspace, out, codes_out, stats_out, R, C,
                                  (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return act_fresh_t<__half>(n, x, cluster_size, slot_of, n_clusters, covered, n_bits, workspace, out, codes_out, stats_out, R, C,
                                   (hipStream_t)stream);
    miq_set_error("miq_act_fresh_quant_fwd: unsupported dtype %d", dtype);
    return -22;
}

// ---------------------------------------------------------------------------------------------
// The same call for callers that cannot allocate (HIP graph capture, serving loops): the partial
// statistics live in a file-scope device array instead of a caller-provided workspace.  Every GPU
// the library runs on holds its own copy of the array at its own address; host code only ever
// uses the per-device address looked up with hipGetSymbolAddress, never the array's name.
// ---------------------------------------------------------------------------------------------
constexpr int64_t kStaticScratchFloats = 2ll << 20;  // 8 MiB per GPU
static __device__ __align__(256) float scratch[kStaticScratchFloats];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static float* g_scratch_ptr[kMaxDevices];  // device address of `scratch` on each GPU, looked up once

// device address of this GPU's copy of `scratch` (the current device must be the tensors' device)
static int static_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    MIQ_CHECK_ARG(e == hipSuccess && dev >= 0 && dev < kMaxDevices, "miq_act_calib_quant_fwd_static: no current device (%s)",
                  hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            miq_set_error("miq_act_calib_quant_fwd_static: hipGetSymbolAddress failed on device %d: %s", dev, hipGetErrorString(e));
            return -5;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

extern "C" int64_t miq_act_static_capacity(void) { return kStaticScratchFloats; }

extern "C" int miq_act_static_init(void) {
    float* ws = nullptr;
    return static_scratch(&ws);
}

extern "C" int miq_act_calib_quant_fwd_static(const void* x, const void* cached_min, const void* cached_max, int metric,
                                              const int* starts, int n_clusters, int64_t covered, int n_bits,
                                              void* out, void* codes_out, void* cmin_out, void* cmax_out, void* scale_out,
                                              void* zp_out, void* rzp_out, float* chan_min_out, float* chan_max_out,
                                              int64_t R, int64_t C, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(R > 0 && C > 0 && C < (1ll << 31), "miq_act_calib_quant_fwd_static: bad shape (R=%lld C=%lld)", (long long)R, (long long)C);
    const int64_t need = miq_act_workspace(R, C, dtype);
    MIQ_CHECK_ARG(need <= kStaticScratchFloats,
                  "miq_act_calib_quant_fwd_static: a [%lld, %lld] slab needs %lld floats of scratch, the static buffer holds %lld; "
                  "use miq_act_calib_quant_fwd with a caller workspace", (long long)R, (long long)C, (long long)need,
                  (long long)kStaticScratchFloats);
    MIQ_CHECK_ARG((chan_min_out == nullptr) == (chan_max_out == nullptr), "miq_act_calib_quant_fwd_static: chan_min_out/chan_max_out must come together");
    float* ws = nullptr;
    int rc = static_scratch(&ws);
    if (rc) return rc;
    rc = miq_act_calib_quant_fwd(x, cached_min, cached_max, metric, starts, n_clusters, covered, n_bits, ws, out, codes_out,
                                 cmin_out, cmax_out, scale_out, zp_out, rzp_out, R, C, dtype, stream);
    if (rc) return rc;
    if (chan_min_out) {
        const int nchunk = miq_colstat_chunks(R, C, dtype);
        const size_t bytes = (size_t)C * sizeof(float);
        hipError_t e = hipMemcpyAsync(chan_min_out, ws + (int64_t)nchunk * C, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e == hipSuccess)
            e = hipMemcpyAsync(chan_max_out, ws + (int64_t)(2 * nchunk + 1) * C, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        if (e != hipSuccess) {
            miq_set_error("miq_act_calib_quant_fwd_static: copy of the per-channel statistics failed: %s", hipGetErrorString(e));
            return -5;
        }
    }
    return 0;
}

extern "C" int miq_gather_lastdim_batch(int n, const void* const* x, const int* const* idx, void* const* out, int64_t R, int64_t C,
                                        int dtype, void* stream) {
    if (n == 0 || R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(n > 0 && n <= kMaxSlabs, "miq_gather_lastdim_batch: %d tensors (1..%d per call)", n, kMaxSlabs);
    MIQ_CHECK_ARG(x && idx && out && R > 0 && C > 0 && C < (1ll << 31), "miq_gather_lastdim_batch: null pointer or bad shape");
    if (dtype == MIQ_F32) return launch_gather<float>(n, x, idx, out, R, C, (hipStream_t)stream);
    if (dtype == MIQ_F16) return launch_gather<__half>(n, x, idx, out, R, C, (hipStream_t)stream);
    miq_set_error("miq_gather_lastdim_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_gather_lastdim(const void* x, const int* idx, void* out, int64_t R, int64_t C, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(x && idx && out && R > 0 && C > 0 && C < (1ll << 31), "miq_gather_lastdim: null pointer or bad shape");
    return miq_gather_lastdim_batch(1, &x, &idx, &out, R, C, dtype, stream);
}
