// Backward of the static activation quantizer (what autograd derives for
// UniformAffineQuantizer.calibration + fake_quant, reference quantize/quantizer.py:157-238,99-121).
//
// With per-channel scale s[c], zero point z[c], m = [0 <= round(x/s)+z <= qmax], xd = clamp(..) - z:
//   direct path      dx  = (g*s)*m/s
//   scale path       dS_k = sum over the cluster slab of g*(xd - m*x/s)   (0 when the scale clamp is active)
//                    d_kmax = dS_k/qmax, d_kmin = -dS_k/qmax
//   routing          per cluster the maximum is an amax over channels of the per-channel amax over rows
//                    (quantizer.py:190-210): autograd splits evenly among tied channels, then evenly among tied
//                    rows, so every element equal to its channel max, in a channel whose max equals the
//                    cluster max, receives (d_kmax / #tied channels) / #tied rows (same for min).
//                    Per-tensor statistics are ONE reduction over all elements (torch.min(x) / torch.max(x),
//                    quantizer.py:222-223, and amin / amax over every axis when no cluster axis is set):
//                    autograd then splits evenly among all tied elements (`flat`: d_kmax / #tied elements).
// One streaming pass plus a sparse correction:
//   k_colbwd_stream   reads x and g once, writes the direct path into dx and leaves per (row chunk,
//                     channel) partial sums of dS and of the number of elements at the channel max / min
//   k_colbwd_fold     sums the partials over the row chunks (deterministic order)
//   k_colbwd_shares   per cluster: dS -> the share every routed element receives
//   k_colbwd_fixup    revisits only the (row chunk, column tile) pairs that hold a routed element and
//                     adds the share to dx there (typically a few tiles out of ~1000)
// so the activation and its gradient cross HBM once (3 passes of traffic in total: x, g, dx).
// In eval mode (frozen scales) a single kernel k_colbwd_apply writes dx = g * m.
#include <type_traits>
#include "miq_common.h"

namespace miq {

// Quotient x/s of the backward pass.  fp16: the correctly rounded fp16 quotient (col_div).  fp32: reciprocal
// multiply plus one FMA correction step, within one ulp of (and almost always equal to) the IEEE quotient at
// a third of its instructions; the backward only sums it and takes round(q), so the forward's bit-exact
// division is not needed here.
template <typename T> __device__ __forceinline__ float bwd_div(float x, float s, float rs);
template <> __device__ __forceinline__ float bwd_div<__half>(float x, float s, float rs) { return col_div<__half>(x, s, rs); }
template <> __device__ __forceinline__ float bwd_div<float>(float x, float s, float rs) {
    const float q0 = x * rs;
    return __builtin_fmaf(__builtin_fmaf(-q0, s, x), rs, q0);
}

// the direct path of one element: fp32 follows autograd's (g*s)/s; for fp16 gradients the plain g is within
// their rounding (and is passed through without conversion).  Outside the grid autograd computes 0/s, which
// is 0 for every valid scale and NaN for a NaN scale (statistics poisoned by a NaN activation): `off`.
template <typename T> __device__ __forceinline__ T direct_out(T graw, float gv, float s, float rs, bool m, T off);
template <> __device__ __forceinline__ float direct_out<float>(float, float gv, float s, float rs, bool m, float off) {
    return m ? bwd_div<float>(gv * s, s, rs) : off;
}
template <> __device__ __forceinline__ __half direct_out<__half>(__half graw, float, float, float, bool m, __half off) {
    return m ? graw : off;
}

// One activation slab of a backward launch (blockIdx.z selects it; the weight-side entry points pass one).
// The extremes an element is compared with are the per-channel statistics (chan_min / chan_max, fp32) when the forward
// kept them, else the merged statistics kmin / kmax themselves: an element equals its cluster's maximum exactly when it
// equals its channel's maximum and that channel attains the cluster maximum, so both forms select the same elements.
struct BwdSlab {
    const void* x;
    const void* g;
    const void* scale;
    const void* rzp;
    const void* kmin;        // [C] merged statistics in the tensor dtype
    const void* kmax;
    const float* chan_min;   // [C] or nullptr
    const float* chan_max;
    void* dx;
    float* ws;               // part [3, nchunk, C] | sums [3, C] | share_min [C] | share_max [C] | flag
};
constexpr int kMaxBwdSlabs = 8;  // per launch (blockIdx.z)
struct BwdSlabs { BwdSlab s[kMaxBwdSlabs]; };

template <typename T>
__device__ __forceinline__ float extreme_at(const float* chan, const void* merged, long long c) {
    return chan ? chan[c] : to_f<T>(static_cast<const T*>(merged)[c]);
}

// dx = direct path; partial sums per (row chunk, channel): [0] dS, [1] #x==chan_max, [2] #x==chan_min.
// The per-element arithmetic is what bounds this kernel for fp16 (6 bytes per element), so it is kept short:
// tie counts are integer adds of the compare result, the clamp is a median-of-three (a NaN level, which
// only +-inf inputs produce, is carried into the sum through the masked quotient instead), and the fp16
// re-roundings of integer intermediates are skipped when every level is exactly representable.
// UNIFORM: the channels of a lane's pack share one cluster (equal clusters whose size is a multiple of VEC, merged
// statistics): one scale / zero point / pair of extremes per lane instead of VEC of each, which is what lets the
// vectorised kernels stay inside 128 registers without spilling.
// LANEPART (with UNIFORM): the lane's pack lies inside one cluster, and a cluster only needs the SUM of dS over its
// channels and to know WHERE elements equal to its extremes sit.  The block then leaves one {dS, tie bits} pair per lane
// and row chunk (8 bytes instead of 3 * VEC floats: the three per-channel planes are 6 MB of a [2048, 4096] fp16 slab's
// 50 MB) and does not count ties at all: a lane whose batch reaches an extreme sets bit 0 (maximum) / bit 1 (minimum) and
// k_colbwd_pairs looks at the rows of exactly those (row chunk, pack) pairs.
template <typename T, int VEC, bool UNIFORM, bool LANEPART = false>
__global__ void __launch_bounds__(256, 4)  // four waves per SIMD: all four blocks of a CU resident
k_colbwd_stream(BwdSlabs slabs, long long R, int C, int rows_per_chunk, float qmax, int flagged) {
    static_assert(!LANEPART || UNIFORM, "one pair per lane needs the lane's pack inside one cluster");
    constexpr int NP = UNIFORM ? 1 : VEC;  // parameter sets per lane
    __shared__ float sm[LANEPART ? 2 : 3][4][LANEPART ? 64 : 64 * VEC];
    const BwdSlab& sb = slabs.s[blockIdx.z];
    const T* __restrict__ x = static_cast<const T*>(sb.x);
    const T* __restrict__ g = static_cast<const T*>(sb.g);
    const T* __restrict__ scale = static_cast<const T*>(sb.scale);
    const T* __restrict__ rzp = static_cast<const T*>(sb.rzp);
    float* __restrict__ part = sb.ws;
    T* __restrict__ dx = static_cast<T*>(sb.dx);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    const bool valid = col0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    if (flagged && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        sb.ws[(3ll * gridDim.y + 5) * C] = 0.f;  // "some cluster left its shares to the correction kernel": set by k_colbwd_cluster
    float acc[VEC];
    int nmx[LANEPART ? 1 : VEC], nmn[LANEPART ? 1 : VEC];
    int tbits = 0;  // LANEPART: bit 0 / 1: a batch of this lane reached the cluster maximum / minimum
    float s[NP], rs[NP], z[NP], cmn[NP], cmx[NP];
    bool ok = true;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int e = 0; e < (LANEPART ? 1 : VEC); ++e) nmx[e] = nmn[e] = 0;
    // A wave owns rows r0 + w, r0 + w + 4, ...; it handles them UNR at a time with all 2*UNR loads in
    // flight (a chunk is typically 32 rows: one or two batches per wave).  The first batch is issued before the per-channel parameters are fetched.
    // 32 elements per batch keep the kernel under 128 registers (four waves per SIMD) without spills; larger
    // batches were measured: no faster in fp32, slower in fp16
    // (fp16 with 4 rows: 64 bytes of spills and 3 % slower in the per-channel form; in the per-lane form, which holds no
    // tie counters, 128 registers without spills and 5-10 % slower)
    constexpr int UNR = VEC == 8 ? 2 : 4;
    // fp16 (two rows per batch: half the bytes in flight of the fp32 form) requests the next batch before it works on
    // the current one
    constexpr bool PREFETCH = VEC == 8;
    Pack<T, VEC> a[UNR], b[UNR], na[PREFETCH ? UNR : 1], nb[PREFETCH ? UNR : 1];
    auto load_into = [&](Pack<T, VEC>* pa, Pack<T, VEC>* pb, long long r) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const long long rr = r + 4 * u;
            if (rr < r1) {
                pa[u] = ld_pack_nt<T, VEC>(x + rr * C + col0);
                pb[u] = ld_pack_nt<T, VEC>(g + rr * C + col0);
            }
        }
    };
    auto load_batch = [&](long long r) { load_into(a, b, r); };
    if (valid) {
        load_batch(r0 + w);
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            s[e] = to_f<T>(scale[col0 + e]);
            rs[e] = 1.0f / s[e];
            z[e] = to_f<T>(rzp[col0 + e]);
            cmn[e] = extreme_at<T>(sb.chan_min, sb.kmin, col0 + e);
            cmx[e] = extreme_at<T>(sb.chan_max, sb.kmax, col0 + e);
            ok = ok && ints_exact<T>(z[e], qmax);
        }
    }
    const bool exact_int = __all(ok);  // wave-uniform
    // NaN scales are rare enough for a wave-uniform special case: only there the masked-out gradient is
    // not the constant 0, and keeping it per channel in the common case costs registers this kernel lacks
    bool finite_scales = true;
#pragma unroll
    for (int e = 0; e < NP; ++e) finite_scales = finite_scales && !(valid && s[e] != s[e]);
    const bool no_nan = __all(finite_scales);
    auto stream_rows = [&](auto exact_tag, auto nan_tag) {
        constexpr bool EX = decltype(exact_tag)::value;
        constexpr bool NANS = decltype(nan_tag)::value;
        // Ties with the extremes are rare: per element only a running maximum / minimum of the batch is kept (the
        // hardware's three-operand forms: one operation per element for both), and the elements are compared and
        // counted one by one only in a wave whose batch reaches an extreme.
        float bmx[NP], bmn[NP];
        auto body = [&](const Pack<T, VEC>& pa, const Pack<T, VEC>& pb, long long r) {
            Pack<T, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int k = UNIFORM ? 0 : e;
                const float xv = to_f<T>(pa.e[e]), gv = to_f<T>(pb.e[e]);
                const float q = bwd_div<T>(xv, s[k], rs[k]);
                const float rq = rint_ste(q);
                float xi = rq + z[k];
                if (!EX) xi = rnd<T>(xi);
                const bool m = (xi >= 0.f) && (xi <= qmax);
                float xd = __builtin_amdgcn_fmed3f(xi, 0.f, qmax) - z[k];
                if (!EX) xd = rnd<T>(xd);
                // outside the grid the quotient drops out; rq - rq is 0, or NaN when the level is NaN
                acc[e] += gv * (xd - (m ? q : rq - rq));
                bmx[k] = __builtin_fmaxf(bmx[k], xv);
                bmn[k] = __builtin_fminf(bmn[k], xv);
                o.e[e] = direct_out<T>(pb.e[e], gv, s[k], rs[k], m, NANS ? from_f<T>(0.f / s[k]) : from_f<T>(0.f));
            }
            st_pack<T, VEC>(dx + r * C + col0, o);
        };
        // (a separate unguarded copy of the batch for the interior of the chunk doubles the code and, with
        // it, the register pressure: measured 2x slower)
        for (long long r = r0 + w; r < r1;) {
            if constexpr (PREFETCH) {
                if (r + 4 * UNR < r1) load_into(na, nb, r + 4 * UNR);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) { bmx[k] = -__builtin_inff(); bmn[k] = __builtin_inff(); }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (r + 4 * u < r1) body(a[u], b[u], r + 4 * u);
            bool tie = false;
#pragma unroll
            for (int k = 0; k < NP; ++k) tie = tie || bmx[k] == cmx[k] || bmn[k] == cmn[k];
            if constexpr (LANEPART) {
                tbits |= (bmx[0] == cmx[0] ? 1 : 0) | (bmn[0] == cmn[0] ? 2 : 0);
            } else if (__any(tie)) {
                // (the rows are read again, from cache, rather than kept in registers for a case this rare)
#pragma unroll 1
                for (int u = 0; u < UNR; ++u) {
                    if (r + 4 * u >= r1) break;
                    const Pack<T, VEC> again = ld_pack<T, VEC>(x + (r + 4 * u) * C + col0);
#pragma unroll
                    for (int e = 0; e < (LANEPART ? 1 : VEC); ++e) {
                        const float xv = to_f<T>(again.e[e]);
                        nmx[e] += (xv == cmx[UNIFORM ? 0 : e]) ? 1 : 0;
                        nmn[e] += (xv == cmn[UNIFORM ? 0 : e]) ? 1 : 0;
                    }
                }
            }
            r += 4 * UNR;
            if constexpr (PREFETCH) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) { a[u] = na[u]; b[u] = nb[u]; }
            } else {
                if (r < r1) load_batch(r);
            }
        }
    };
    if (valid) {
        const bool ex = sizeof(T) == 4 || exact_int;
        if (ex && no_nan) stream_rows(std::true_type{}, std::false_type{});
        else if (ex) stream_rows(std::true_type{}, std::true_type{});
        else if (no_nan) stream_rows(std::false_type{}, std::false_type{});
        else stream_rows(std::false_type{}, std::true_type{});
    }
    if constexpr (LANEPART) {
        // the pack's channels in a fixed tree, then the four waves in a fixed order: one sum per lane and row chunk
        float a = 0.f;
        if constexpr (VEC == 8) a = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        else if constexpr (VEC == 4) a = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) a += acc[e];
        }
        sm[0][w][lane] = a;
        sm[1][w][lane] = __int_as_float(tbits);
        __syncthreads();
        if (w == 0 && valid) {
            const int bits = __float_as_int(sm[1][0][lane]) | __float_as_int(sm[1][1][lane]) | __float_as_int(sm[1][2][lane]) |
                             __float_as_int(sm[1][3][lane]);
            float2 o;
            o.x = (sm[0][0][lane] + sm[0][1][lane]) + (sm[0][2][lane] + sm[0][3][lane]);
            o.y = __int_as_float(bits);
            const long long P = C / VEC;  // packs per row
            reinterpret_cast<float2*>(part)[(long long)blockIdx.y * P + (long long)blockIdx.x * 64 + lane] = o;
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            sm[0][w][lane * VEC + e] = acc[e];
            sm[1][w][lane * VEC + e] = (float)nmx[e];
            sm[2][w][lane * VEC + e] = (float)nmn[e];
        }
        __syncthreads();
        if (w == 0 && valid) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int i = lane * VEC + e;
                    part[((long long)k * gridDim.y + blockIdx.y) * C + col0 + e] =
                        (sm[k][0][i] + sm[k][1][i]) + (sm[k][2][i] + sm[k][3][i]);
                }
        }
    }
}

// fold the row-chunk partials into sums[3, C] (the per-chunk counts stay: the correction kernel reads
// them).  Block = 64 channels x 4 waves, wave w sums chunks w, w+4, ...
__global__ void __launch_bounds__(256)
k_colbwd_fold(BwdSlabs slabs, int nchunk, int C) {
    __shared__ float sm[4][64];
    const float* __restrict__ part = slabs.s[blockIdx.z].ws;
    float* __restrict__ sums = slabs.s[blockIdx.z].ws + 3ll * nchunk * C;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int k = blockIdx.y;  // plane: dS, #max, #min
    const bool valid = c < C;
    float acc = 0.f;
    if (valid) {
        const float* p = part + ((long long)k * nchunk) * C + c;
        // wave w sums chunks w, w+4, ... in this fixed order; up to 16 loads in flight per thread
        for (int j0 = w; j0 < nchunk; j0 += 64) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int j = j0 + 4 * u;
                v[u] = p[(long long)(j < nchunk ? j : nchunk - 1) * C];  // unconditional load, masked below
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j0 + 4 * u >= nchunk) v[u] = 0.f;
            acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])) +
                   (((v[8] + v[9]) + (v[10] + v[11])) + ((v[12] + v[13]) + (v[14] + v[15])));
        }
    }
    sm[w][lane] = acc;
    __syncthreads();
    if (w == 0 && valid)
        sums[(long long)k * C + c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// gridDim.y blocks per cluster (and for the un-clustered channels): per-channel shares for the max / min elements.
// Every block of a cluster sums the whole cluster (L2-resident vectors, four loads in flight, the same fixed order
// in every block) and writes the shares of its own slice of channels: a single large cluster (per-tensor
// statistics) is spread over up to 16 blocks instead of one block walking all channels.
// CHAN: the forward kept per-channel statistics (a channel attains the cluster extreme when its own extreme equals the
// merged one); otherwise a channel attains it when it holds at least one element equal to the merged extreme.
template <typename T, bool CHAN>
__global__ void __launch_bounds__(256)
k_colbwd_shares(BwdSlabs slabs, int nchunk, int C, const int* __restrict__ starts, int ncl, float qmax, int flat) {
    __shared__ float lds[48];
    const BwdSlab& sb = slabs.s[blockIdx.z];
    const float* __restrict__ sums = sb.ws + 3ll * nchunk * C;
    float* __restrict__ share_min = sb.ws + (3ll * nchunk + 3) * C;
    float* __restrict__ share_max = share_min + C;
    const T* __restrict__ kmin = static_cast<const T*>(sb.kmin);
    const T* __restrict__ kmax = static_cast<const T*>(sb.kmax);
    const float* __restrict__ chan_min = sb.chan_min;
    const float* __restrict__ chan_max = sb.chan_max;
    const int k = blockIdx.x;
    const bool own = k >= ncl;
    const int st = own ? starts[ncl] : starts[k];
    const int en = own ? C : starts[k + 1];
    const float lo = rnd<T>(kClipLo);
    const int first = st + blockIdx.y * 256 + threadIdx.x, step = 256 * gridDim.y;
    auto colsum = [&](int which, int c) { return sums[(long long)which * C + c]; };
    if (own) {
        for (int c = first; c < en; c += step) {
            const float mn = to_f<T>(kmin[c]), mx = to_f<T>(kmax[c]);
            const float s_raw = rnd<T>(rnd<T>(mx - mn) / qmax);
            const bool live = s_raw >= lo && s_raw <= kClipHi;
            const float d = live ? colsum(0, c) / qmax : 0.f;
            share_max[c] = d / colsum(1, c);
            share_min[c] = -d / colsum(2, c);
        }
        return;
    }
    if (en <= st) return;
    const float mn = to_f<T>(kmin[st]), mx = to_f<T>(kmax[st]);  // constant over the cluster
    // cluster totals: dS, number of channels (flat: of elements) attaining the cluster max / min.  Every load is
    // unconditional (an out-of-range channel re-reads the cluster's first one and is masked afterwards): guarded loads
    // would be issued one by one behind a branch each.
    float acc[3] = {0.f, 0.f, 0.f};
    for (int c = st + threadIdx.x; c < en; c += 1024) {
        float ds[4], n1[4], n2[4], cx[4], cn[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cc = c + 256 * u;
            const int j = cc < en ? cc : st;
            ds[u] = colsum(0, j);
            n1[u] = colsum(1, j);
            n2[u] = colsum(2, j);
            if (CHAN) { cx[u] = chan_max[j]; cn[u] = chan_min[j]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = c + 256 * u < en;
            const bool ax = in && (CHAN ? cx[u] == mx : n1[u] > 0.f);
            const bool an = in && (CHAN ? cn[u] == mn : n2[u] > 0.f);
            acc[0] += in ? ds[u] : 0.f;
            acc[1] += ax ? (flat ? n1[u] : 1.f) : 0.f;
            acc[2] += an ? (flat ? n2[u] : 1.f) : 0.f;
        }
    }
    team_sum<256, 3>(acc, lds);
    const float s_raw = rnd<T>(rnd<T>(mx - mn) / qmax);
    const bool live = s_raw >= lo && s_raw <= kClipHi;
    const float d = live ? acc[0] / qmax : 0.f;
    for (int c = first; c < en; c += step) {
        const float n1 = colsum(1, c), n2 = colsum(2, c);
        const bool ax = CHAN ? chan_max[c] == mx : n1 > 0.f;
        const bool an = CHAN ? chan_min[c] == mn : n2 > 0.f;
        share_max[c] = ax ? (flat ? d / acc[1] : (d / acc[1]) / n1) : 0.f;
        share_min[c] = an ? (flat ? -d / acc[2] : (-d / acc[2]) / n2) : 0.f;
    }
}

// Fold + shares + sparse correction in one launch, for clusters of at most kClusterMax channels: one block per
// cluster.  The arithmetic is that of k_colbwd_fold followed by k_colbwd_shares, in the same order (per channel: four
// interleaved chunk sequences summed 16 at a time, then (s0 + s1) + (s2 + s3); per cluster: the strided sum and tree of
// k_colbwd_shares), so the shares carry the same bits.  While folding the tie counts the block notes which (row chunk,
// channel) pairs hold an element at a cluster extreme; up to kPairMax of them are corrected right here (one wave scans
// the pair's rows), and the shares it publishes for k_colbwd_fixup are then zero.  A cluster with more pairs (ReLU-style
// mass ties) either walks its whole slab right here (inline_mass: the host chose it because the slab is small; no
// correction launch follows) or publishes its shares and raises the slab's flag, and k_colbwd_fixup streams over it.
constexpr int kClusterMax = 1024;
constexpr int kPairMax = 64;
constexpr long long kInlineMass = 1ll << 20;  // elements of a cluster slab one block corrects itself under mass ties
template <typename T>
__global__ void __launch_bounds__(1024)
k_colbwd_cluster(BwdSlabs slabs, int nchunk, long long R, int C, int rows_per_chunk, const int* __restrict__ starts, int csize,
                 float qmax, int flat, int inline_mass) {
    __shared__ float part_l[3][4][256];
    __shared__ float tot[3][kClusterMax];
    __shared__ float shr[2][kClusterMax];
    __shared__ float lds[48];
    __shared__ int pairs[kPairMax];
    __shared__ int npairs;
    const BwdSlab& sb = slabs.s[blockIdx.z];
    const float* __restrict__ part = sb.ws;
    float* __restrict__ share_min = sb.ws + (3ll * nchunk + 3) * C;
    float* __restrict__ share_max = share_min + C;
    // equal clusters: the bounds by arithmetic (the loads below then do not wait for `starts`)
    const int st = csize > 0 ? (int)blockIdx.x * csize : starts[blockIdx.x];
    const int en = csize > 0 ? st + csize : starts[blockIdx.x + 1];
    const int W = en - st;
    if (W <= 0) return;
    const int tid = threadIdx.x;
    // requested now, used after the fold
    const T kmin_raw = static_cast<const T*>(sb.kmin)[st], kmax_raw = static_cast<const T*>(sb.kmax)[st];
    if (tid == 0) npairs = 0;
    __syncthreads();
    // per-channel fold, TW = blockDim.x / 4 channels at a time: thread (w, c) sums chunks w, w + 4, ... of channel c
    const int TW = blockDim.x >> 2, nwaves = blockDim.x >> 6;
    for (int c0 = 0; c0 < W; c0 += TW) {
        const int c = tid % TW, w = tid / TW;
        const bool valid = c0 + c < W;
        float acc[3] = {0.f, 0.f, 0.f};
        {
            // every load is unconditional (an out-of-range chunk or channel re-reads a valid element and is zeroed
            // afterwards): guarded loads would be issued one by one behind a branch each
            const float* p = part + st + (valid ? c0 + c : 0);
            for (int j0 = w; j0 < nchunk; j0 += 64) {
                float v[3][16];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int j = j0 + 4 * u;
                        v[k][u] = p[((long long)k * nchunk + (j < nchunk ? j : nchunk - 1)) * C];
                    }
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (!valid || j0 + 4 * u >= nchunk) v[k][u] = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    acc[k] += ((v[k][0] + v[k][1]) + (v[k][2] + v[k][3])) + ((v[k][4] + v[k][5]) + (v[k][6] + v[k][7])) +
                              (((v[k][8] + v[k][9]) + (v[k][10] + v[k][11])) + ((v[k][12] + v[k][13]) + (v[k][14] + v[k][15])));
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int bits = (v[1][u] > 0.f ? 1 : 0) | (v[2][u] > 0.f ? 2 : 0);
                    if (bits) {
                        const int i = atomicAdd(&npairs, 1);
                        if (i < kPairMax) pairs[i] = ((j0 + 4 * u) << 12) | ((c0 + c) << 2) | bits;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) part_l[k][w][c] = acc[k];
        __syncthreads();
        if (w == 0 && valid) {
#pragma unroll
            for (int k = 0; k < 3; ++k) tot[k][c0 + c] = (part_l[k][0][c] + part_l[k][1][c]) + (part_l[k][2][c] + part_l[k][3][c]);
        }
        __syncthreads();
    }
    // cluster totals in the order of k_colbwd_shares: 256 strided partial sums, then the tree of team_sum<256, 3>
    const float mn = to_f<T>(kmin_raw), mx = to_f<T>(kmax_raw);
    float acc[3] = {0.f, 0.f, 0.f};
    if (tid < 256) {
        for (int c = tid; c < W; c += 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = c + 256 * u;
                const bool in = cc < W;
                acc[0] += in ? tot[0][cc] : 0.f;
                acc[1] += (in && tot[1][cc] > 0.f) ? (flat ? tot[1][cc] : 1.f) : 0.f;
                acc[2] += (in && tot[2][cc] > 0.f) ? (flat ? tot[2][cc] : 1.f) : 0.f;
            }
        }
    }
    team_sum<256, 3>(acc, lds);  // every wave takes part in the barriers; the sums are those of waves 0..3
    const float lo = rnd<T>(kClipLo);
    const float s_raw = rnd<T>(rnd<T>(mx - mn) / qmax);
    const bool live = s_raw >= lo && s_raw <= kClipHi;
    const float d = live ? acc[0] / qmax : 0.f;
    const bool mass = npairs > kPairMax;
    const bool publish = mass && !inline_mass;
    for (int c = tid; c < W; c += blockDim.x) {
        const float hx = tot[1][c] > 0.f ? (flat ? d / acc[1] : (d / acc[1]) / tot[1][c]) : 0.f;
        const float hn = tot[2][c] > 0.f ? (flat ? -d / acc[2] : (-d / acc[2]) / tot[2][c]) : 0.f;
        shr[0][c] = hx;
        shr[1][c] = hn;
        if (!inline_mass) {
            share_max[st + c] = publish ? hx : 0.f;
            share_min[st + c] = publish ? hn : 0.f;
        }
    }
    if (publish) {
        if (tid == 0) sb.ws[(3ll * nchunk + 5) * C] = 1.f;
        return;
    }
    __syncthreads();
    const T* __restrict__ x = static_cast<const T*>(sb.x);
    T* __restrict__ dx = static_cast<T*>(sb.dx);
    if (mass) {
        // mass ties in a cluster slab small enough for one block: walk all of its elements, eight rows in flight per
        // thread, consecutive threads on consecutive channels.  (direct + share_max) + share_min as everywhere.
        const long long total = R * W;
        const long long step = blockDim.x;
        for (long long i0 = tid; i0 < total; i0 += 8 * step) {
            T xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long i = i0 + u * step < total ? i0 + u * step : i0;
                xv[u] = x[(i / W) * C + st + (int)(i % W)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long i = i0 + u * step;
                if (i >= total) break;
                const int c = (int)(i % W);
                const float xf = to_f<T>(xv[u]), hx = shr[0][c], hn = shr[1][c];
                const bool hit_x = hx != 0.f && xf == mx, hit_n = hn != 0.f && xf == mn;
                if (hit_x || hit_n) {
                    const long long at = (i / W) * C + st + c;
                    float v = to_f<T>(dx[at]);
                    if (hit_x) v += hx;
                    if (hit_n) v += hn;
                    dx[at] = from_f<T>(v);
                }
            }
        }
        return;
    }
    // sparse correction: wave v takes pairs v, v + nwaves, ...; its lanes scan the rows of the pair's chunk
    const int lane = tid & 63;
    for (int i = tid >> 6; i < npairs; i += nwaves) {
        const int e = pairs[i];
        const int j = e >> 12, c = (e >> 2) & 1023;
        const float hx = shr[0][c], hn = shr[1][c];
        const bool amx = (e & 1) && hx != 0.f, amn = (e & 2) && hn != 0.f;  // a NaN share is added like any other value
        if (!amx && !amn) continue;
        const long long r0 = (long long)j * rows_per_chunk;
        long long r1 = r0 + rows_per_chunk;
        if (r1 > R) r1 = R;
        for (long long r = r0 + lane; r < r1; r += 64) {
            const long long at = r * C + st + c;
            const T xr = x[at], dr = dx[at];  // together: one trip to memory instead of two
            const float xv = to_f<T>(xr);
            const bool hit_x = amx && xv == mx, hit_n = amn && xv == mn;
            if (hit_x || hit_n) {
                float v = to_f<T>(dr);
                if (hit_x) v += hx;
                if (hit_n) v += hn;
                dx[at] = from_f<T>(v);
            }
        }
    }
}

// Second (and last) launch behind k_colbwd_stream<.., LANEPART>: one block per cluster and slab.
//   fold     the cluster's {dS, tie bits} pairs over all row chunks and its packs, in a fixed order (every thread its
//            strided entries, then the block tree): nchunk * csize / VEC pairs, a few KB
//   pairs    (row chunk, pack) entries with a tie bit name the only places where an element can equal a cluster
//            extreme: their rows are read back and the tied elements counted per channel (integer counts in LDS: the
//            order of arrival does not matter)
//   shares   d_kmax = dS / qmax, split over the tied channels and then over the tied rows of a channel (flat: over all
//            tied elements), as autograd splits amax / amin (d_kmin likewise, negated); masked by the scale clamp
//   apply    dx += share at those elements: (direct + share_max) + share_min like every other form of this backward
// More than kPairMax entries (ReLU-style mass ties: every chunk holds a minimum) make the block walk its whole cluster
// slab instead, once to count and once to apply.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_colbwd_pairs(BwdSlabs slabs, int nchunk, long long R, int C, int rows_per_chunk, int csize, float qmax, int flat) {
    __shared__ int cnt[2][kClusterMax];
    __shared__ float lds[64];
    __shared__ int pairs[kPairMax];
    __shared__ int npairs;
    const BwdSlab& sb = slabs.s[blockIdx.z];
    const float2* __restrict__ plane = reinterpret_cast<const float2*>(sb.ws);
    const T* __restrict__ x = static_cast<const T*>(sb.x);
    T* __restrict__ dx = static_cast<T*>(sb.dx);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int st = (int)blockIdx.x * csize;
    const long long P = C / VEC;
    const int L = csize / VEC, lane0 = st / VEC;
    const T kmin_raw = static_cast<const T*>(sb.kmin)[st], kmax_raw = static_cast<const T*>(sb.kmax)[st];  // used after the fold
    if (tid == 0) npairs = 0;
    for (int c = tid; c < csize; c += 256) cnt[0][c] = cnt[1][c] = 0;
    __syncthreads();
    const int entries = nchunk * L;
    float acc = 0.f;
    for (int i0 = tid; i0 < entries; i0 += 256 * 8) {
        float2 v[8];
        int jj[8], ll[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // unconditional loads (an entry past the end re-reads the last one and is skipped below)
            const int i = i0 + 256 * u < entries ? i0 + 256 * u : entries - 1;
            jj[u] = i / L;
            ll[u] = i - jj[u] * L;
            v[u] = plane[(long long)jj[u] * P + lane0 + ll[u]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + 256 * u >= entries) break;
            acc += v[u].x;
            const int bits = __float_as_int(v[u].y);
            if (bits) {
                const int k = atomicAdd(&npairs, 1);
                if (k < kPairMax) pairs[k] = (jj[u] << 12) | (ll[u] << 2) | bits;
            }
        }
    }
    // block sum in a fixed tree (and a barrier behind the list)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) lds[wv] = acc;
    __syncthreads();
    const float dS = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    const int np = npairs;
    const bool mass = np > kPairMax;
    const float mn = to_f<T>(kmin_raw), mx = to_f<T>(kmax_raw);
    // the elements of (row chunk j, pack l) that equal a cluster extreme: f(row, channel inside the cluster, is_max)
    auto scan_pair = [&](int e, auto f) {
        const int j = e >> 12, l = (e >> 2) & 1023;
        const long long r0 = (long long)j * rows_per_chunk;
        long long r1 = r0 + rows_per_chunk;
        if (r1 > R) r1 = R;
        for (long long r = r0 + lane; r < r1; r += 64) {
            const Pack<T, VEC> a = ld_pack<T, VEC>(x + r * C + st + l * VEC);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float xv = to_f<T>(a.e[q]);
                if ((e & 1) && xv == mx) f(r, l * VEC + q, true);
                if ((e & 2) && xv == mn) f(r, l * VEC + q, false);
            }
        }
    };
    auto scan_slab = [&](auto f) {
        const long long total = R * L;
        for (long long i = tid; i < total; i += 256) {
            const long long r = i / L;
            const int l = (int)(i - r * L);
            const Pack<T, VEC> a = ld_pack<T, VEC>(x + r * C + st + l * VEC);
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                const float xv = to_f<T>(a.e[q]);
                if (xv == mx) f(r, l * VEC + q, true);
                if (xv == mn) f(r, l * VEC + q, false);
            }
        }
    };
    auto count = [&](long long, int c, bool is_max) { atomicAdd(&cnt[is_max ? 0 : 1][c], 1); };
    if (mass) scan_slab(count);
    else
        for (int i = wv; i < np; i += 4) scan_pair(pairs[i], count);
    __syncthreads();
    // number of channels that attain each extreme (flat: of elements)
    int n[2] = {0, 0};
    for (int c = tid; c < csize; c += 256) {
        n[0] += flat ? cnt[0][c] : (cnt[0][c] > 0 ? 1 : 0);
        n[1] += flat ? cnt[1][c] : (cnt[1][c] > 0 ? 1 : 0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { n[0] += __shfl_xor(n[0], o); n[1] += __shfl_xor(n[1], o); }
    if (lane == 0) { lds[8 + wv] = (float)n[0]; lds[16 + wv] = (float)n[1]; }
    __syncthreads();
    const float nmax = (lds[8] + lds[9]) + (lds[10] + lds[11]), nmin_ = (lds[16] + lds[17]) + (lds[18] + lds[19]);
    const float lo = rnd<T>(kClipLo);
    const float s_raw = rnd<T>(rnd<T>(mx - mn) / qmax);
    const bool live = s_raw >= lo && s_raw <= kClipHi;
    const float d = live ? dS / qmax : 0.f;
    auto apply = [&](long long r, int c, bool is_max) {
        const float k = (float)cnt[is_max ? 0 : 1][c];
        const float h = is_max ? (flat ? d / nmax : (d / nmax) / k) : (flat ? -d / nmin_ : (-d / nmin_) / k);
        // an element that is both (a constant cluster) is visited twice by one lane, maximum first
        const long long at = r * C + st + c;
        dx[at] = from_f<T>(to_f<T>(dx[at]) + h);
    };
    if (mass) scan_slab(apply);
    else
        for (int i = wv; i < np; i += 4) scan_pair(pairs[i], apply);
}

// eval mode (frozen scales): dx = g * [level inside the grid]
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_colbwd_apply(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ scale,
               const T* __restrict__ rzp, T* __restrict__ dx, long long R, int C, int rows_per_chunk, float qmax) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    if (col0 >= C) return;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float s[VEC], rs[VEC], z[VEC];
    T off[VEC];
    Pack<T, VEC> ps = ld_pack<T, VEC>(scale + col0), pz = ld_pack<T, VEC>(rzp + col0);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        s[e] = to_f<T>(ps.e[e]);
        rs[e] = 1.0f / s[e];
        off[e] = from_f<T>(0.f / s[e]);
        z[e] = to_f<T>(pz.e[e]);
    }
    auto body = [&](const Pack<T, VEC>& a, const Pack<T, VEC>& b, long long r) {
        Pack<T, VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xv = to_f<T>(a.e[e]), gv = to_f<T>(b.e[e]);
            const float xi = rnd<T>(rint_ste(bwd_div<T>(xv, s[e], rs[e])) + z[e]);
            const bool m = (xi >= 0.f) && (xi <= qmax);
            o.e[e] = direct_out<T>(b.e[e], gv, s[e], rs[e], m, off[e]);
        }
        st_pack<T, VEC>(dx + r * C + col0, o);
    };
    long long r = r0 + w;
    for (; r + 12 < r1; r += 16) {
        Pack<T, VEC> a0 = ld_pack<T, VEC>(x + r * C + col0), b0 = ld_pack<T, VEC>(g + r * C + col0);
        Pack<T, VEC> a1 = ld_pack<T, VEC>(x + (r + 4) * C + col0), b1 = ld_pack<T, VEC>(g + (r + 4) * C + col0);
        Pack<T, VEC> a2 = ld_pack<T, VEC>(x + (r + 8) * C + col0), b2 = ld_pack<T, VEC>(g + (r + 8) * C + col0);
        Pack<T, VEC> a3 = ld_pack<T, VEC>(x + (r + 12) * C + col0), b3 = ld_pack<T, VEC>(g + (r + 12) * C + col0);
        body(a0, b0, r);
        body(a1, b1, r + 4);
        body(a2, b2, r + 8);
        body(a3, b3, r + 12);
    }
    for (; r < r1; r += 4) {
        Pack<T, VEC> a0 = ld_pack<T, VEC>(x + r * C + col0), b0 = ld_pack<T, VEC>(g + r * C + col0);
        body(a0, b0, r);
    }
}

// VEC consecutive floats, as 16-byte loads when VEC is a multiple of 4 (p 16-byte aligned then)
template <int VEC>
__device__ __forceinline__ void ld_floats(const float* __restrict__ p, float* out) {
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const Pack<float, 4> v = ld_pack<float, 4>(p + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) out[4 * q + e] = v.e[e];
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = p[e];
    }
}

// sparse correction after k_colbwd_stream: same (column tile, row chunk) grid, a lane owns one pack of
// channels, wave w every 4th row of the chunk.  A lane goes back over its rows only if one of its channels
// holds, inside this chunk, an element equal to the channel max / min (the per-chunk counts of the
// streaming pass say so) whose share is not zero.  dx += share at those elements: the sum
// (direct + share_max) + share_min of a single-pass kernel, in the same order.  Usually a handful of lanes
// do any work (their loads are issued eight rows at a time: the kernel is latency-bound); with ReLU-style
// mass ties every block streams its tile with coalesced 16-byte accesses.
template <typename T, int VEC>
__device__ __forceinline__ void fixup_tile(const BwdSlab& sb, int nchunk, long long R, int C, int rows_per_chunk, int tile, int chunk) {
    const T* __restrict__ x = static_cast<const T*>(sb.x);
    const float* __restrict__ part = sb.ws;
    const float* __restrict__ share_min = sb.ws + (3ll * nchunk + 3) * C;
    const float* __restrict__ share_max = share_min + C;
    T* __restrict__ dx = static_cast<T*>(sb.dx);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)tile * 64 + lane) * VEC;
    if (col0 >= C) return;
    const long long r0 = (long long)chunk * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float hmn[VEC], hmx[VEC], nmx[VEC], nmn[VEC], cmn[VEC], cmx[VEC];
    ld_floats<VEC>(part + ((long long)1 * nchunk + chunk) * C + col0, nmx);
    ld_floats<VEC>(part + ((long long)2 * nchunk + chunk) * C + col0, nmn);
    ld_floats<VEC>(share_max + col0, hmx);
    ld_floats<VEC>(share_min + col0, hmn);
    if (sb.chan_max) {
        ld_floats<VEC>(sb.chan_max + col0, cmx);
        ld_floats<VEC>(sb.chan_min + col0, cmn);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            cmx[e] = to_f<T>(static_cast<const T*>(sb.kmax)[col0 + e]);
            cmn[e] = to_f<T>(static_cast<const T*>(sb.kmin)[col0 + e]);
        }
    }
    bool amx[VEC], amn[VEC], any = false;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        // a NaN share compares unequal to zero and is added, like every other value
        amx[e] = nmx[e] > 0.f && hmx[e] != 0.f;
        amn[e] = nmn[e] > 0.f && hmn[e] != 0.f;
        any = any || amx[e] || amn[e];
    }
    if (!any) return;
    auto fix = [&](const Pack<T, VEC>& a, long long r) {
        bool hit = false;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xv = to_f<T>(a.e[e]);
            hit = hit || (amx[e] && xv == cmx[e]) || (amn[e] && xv == cmn[e]);
        }
        if (!hit) return;
        Pack<T, VEC> o = ld_pack<T, VEC>(dx + r * C + col0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xv = to_f<T>(a.e[e]);
            float d = to_f<T>(o.e[e]);
            if (amx[e] && xv == cmx[e]) d += hmx[e];
            if (amn[e] && xv == cmn[e]) d += hmn[e];
            o.e[e] = from_f<T>(d);
        }
        st_pack<T, VEC>(dx + r * C + col0, o);
    };
    long long r = r0 + w;
    for (; r + 28 < r1; r += 32) {
        Pack<T, VEC> a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = ld_pack<T, VEC>(x + (r + 4 * u) * C + col0);
#pragma unroll
        for (int u = 0; u < 8; ++u) fix(a[u], r + 4 * u);
    }
    for (; r < r1; r += 4) fix(ld_pack<T, VEC>(x + r * C + col0), r);
}

// The correction as a kernel.  flagged == 0: one block per (column tile, row chunk, slab).  flagged != 0 (behind
// k_colbwd_cluster): a small one-dimensional grid whose blocks first read the flag words of all slabs and leave at
// once when no cluster with mass ties left its shares (the usual case: the launch costs as little as a launch can);
// otherwise they walk the (tile, chunk) pairs of the flagged slabs.
template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_colbwd_fixup(BwdSlabs slabs, int n, int tiles, int nchunk, long long R, int C, int rows_per_chunk, int flagged) {
    if (!flagged) {
        fixup_tile<T, VEC>(slabs.s[blockIdx.z], nchunk, R, C, rows_per_chunk, blockIdx.x, blockIdx.y);
        return;
    }
    float f[kMaxBwdSlabs];
#pragma unroll
    for (int i = 0; i < kMaxBwdSlabs; ++i) f[i] = slabs.s[i < n ? i : 0].ws[(3ll * nchunk + 5) * C];
    unsigned live = 0;
#pragma unroll
    for (int i = 0; i < kMaxBwdSlabs; ++i) live |= (i < n && f[i] != 0.f) ? (1u << i) : 0u;
    if (!live) return;
    const int per_slab = tiles * nchunk;
    for (int t = blockIdx.x; t < per_slab * n; t += gridDim.x) {
        const int i = t / per_slab, rem = t - i * per_slab;
        if (live & (1u << i)) fixup_tile<T, VEC>(slabs.s[i], nchunk, R, C, rows_per_chunk, rem % tiles, rem / tiles);
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Launches the backward over `n` slabs of one shape.  merged: fold + shares + sparse correction in k_colbwd_cluster
// (needs clusters of at most kClusterMax channels that cover every channel, and the flag word behind the workspace).
template <typename T>
static int colquant_bwd_launch(const BwdSlabs& slabs, int n, bool routed, const int* starts, int ncl, long long covered,
                               int nchunk, long long R, long long C, int n_bits, int phase, bool merged, bool uniform,
                               int flat, hipStream_t stream, const char* who) {
    constexpr int N = VecOf<T>::N;
    bool vec_ok = (C % N == 0);
    for (int i = 0; i < n; ++i) {
        const BwdSlab& b = slabs.s[i];
        vec_ok = vec_ok && aligned16(b.x) && aligned16(b.g) && aligned16(b.dx) && aligned16(b.scale) && aligned16(b.rzp) &&
                 aligned16(b.ws) && aligned16(b.chan_min) && aligned16(b.chan_max);
    }
    const float qmax = (float)((1 << n_bits) - 1);
    const int tiles = (int)(vec_ok ? (C / N + 63) / 64 : (C + 63) / 64);
    const int rpc = (int)((R + nchunk - 1) / nchunk);
    const dim3 grid(tiles, nchunk, n), block(256);
    if (!routed) {  // eval mode: one slab
        const BwdSlab& b = slabs.s[0];
        if (vec_ok)
            hipLaunchKernelGGL((k_colbwd_apply<T, N>), dim3(tiles, nchunk), block, 0, stream, (const T*)b.x, (const T*)b.g,
                               (const T*)b.scale, (const T*)b.rzp, (T*)b.dx, R, (int)C, rpc, qmax);
        else
            hipLaunchKernelGGL((k_colbwd_apply<T, 1>), dim3(tiles, nchunk), block, 0, stream, (const T*)b.x, (const T*)b.g,
                               (const T*)b.scale, (const T*)b.rzp, (T*)b.dx, R, (int)C, rpc, qmax);
        MIQ_CHECK_LAUNCH(who);
        return 0;
    }
    // a cluster slab of up to kInlineMass elements is corrected by its own block of k_colbwd_cluster even under mass ties:
    // no flag and no correction launch then
    const bool inline_mass = merged && R * (covered / (ncl > 0 ? ncl : 1)) <= kInlineMass;
    const int flagged = merged && !inline_mass ? 1 : 0;
    if (merged && uniform && vec_ok && phase == 0) {
        // two launches: per-lane partials, then one block per cluster folds them and corrects the tied elements
        const int csize = (int)(covered / (ncl > 0 ? ncl : 1));
        hipLaunchKernelGGL((k_colbwd_stream<T, N, true, true>), grid, block, 0, stream, slabs, R, (int)C, rpc, qmax, 0);
        hipLaunchKernelGGL((k_colbwd_pairs<T, N>), dim3(ncl, 1, n), block, 0, stream, slabs, nchunk, R, (int)C, rpc, csize, qmax, flat);
        MIQ_CHECK_LAUNCH(who);
        return 0;
    }
    if (phase != 2) {
        if (vec_ok && uniform) hipLaunchKernelGGL((k_colbwd_stream<T, N, true>), grid, block, 0, stream, slabs, R, (int)C, rpc, qmax, flagged);
        else if (vec_ok) hipLaunchKernelGGL((k_colbwd_stream<T, N, false>), grid, block, 0, stream, slabs, R, (int)C, rpc, qmax, flagged);
        else hipLaunchKernelGGL((k_colbwd_stream<T, 1, true>), grid, block, 0, stream, slabs, R, (int)C, rpc, qmax, flagged);
        if (!merged)
            hipLaunchKernelGGL(k_colbwd_fold, dim3((unsigned)((C + 63) / 64), 3, n), block, 0, stream, slabs, nchunk, (int)C);
    }
    if (phase != 1) {
        if (merged) {
            // four chunk sequences x a tile of the cluster's channels (64 .. 256, a multiple of 64) per block
            long long tw = (covered / (ncl > 0 ? ncl : 1) + 63) / 64 * 64;
            tw = tw < 64 ? 64 : (tw > 256 ? 256 : tw);
            const int csize = (int)(covered / (ncl > 0 ? ncl : 1));  // merged: equal clusters over all channels
            hipLaunchKernelGGL((k_colbwd_cluster<T>), dim3(ncl, 1, n), dim3((unsigned)(4 * tw)), 0, stream, slabs, nchunk, R, (int)C, rpc,
                               starts, csize, qmax, flat, inline_mass ? 1 : 0);
            if (inline_mass) {
                MIQ_CHECK_LAUNCH(who);
                return 0;
            }
        } else {
            const unsigned nblk = (unsigned)(ncl + (covered < C ? 1 : 0));
            if (slabs.s[0].chan_max)
                hipLaunchKernelGGL((k_colbwd_shares<T, true>), dim3(nblk, cluster_slices(C, nblk), n), block, 0, stream, slabs, nchunk,
                                   (int)C, starts, ncl, qmax, flat);
            else
                hipLaunchKernelGGL((k_colbwd_shares<T, false>), dim3(nblk, cluster_slices(C, nblk), n), block, 0, stream, slabs, nchunk,
                                   (int)C, starts, ncl, qmax, flat);
        }
        const long long pairs = (long long)tiles * nchunk * n;
        const dim3 fgrid = flagged ? dim3((unsigned)(pairs < 512 ? pairs : 512)) : grid;
        if (vec_ok) hipLaunchKernelGGL((k_colbwd_fixup<T, N>), fgrid, block, 0, stream, slabs, n, tiles, nchunk, R, (int)C, rpc, flagged);
        else hipLaunchKernelGGL((k_colbwd_fixup<T, 1>), fgrid, block, 0, stream, slabs, n, tiles, nchunk, R, (int)C, rpc, flagged);
    }
    MIQ_CHECK_LAUNCH(who);
    return 0;
}

}  // namespace miq

using namespace miq;

extern "C" int miq_colquant_bwd(const void* grad_out, const void* x, const void* scale, const void* rzp,
                                const float* chan_min, const float* chan_max, const void* kmin, const void* kmax,
                                const int* starts, int n_clusters, int64_t covered, float* workspace, int nchunk,
                                void* dx, int64_t R, int64_t C, int n_bits, int phase, int flat_ties, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(grad_out && x && scale && rzp && dx && R > 0 && C > 0 && C < (1ll << 31),
                  "miq_colquant_bwd: null pointer or bad shape");
    MIQ_CHECK_ARG(phase >= 0 && phase <= 2 && (phase == 0 || chan_min), "miq_colquant_bwd: phase %d needs the routing inputs", phase);
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 15 && nchunk >= 1, "miq_colquant_bwd: bad n_bits/nchunk");
    MIQ_CHECK_ARG((chan_min == nullptr) == (chan_max == nullptr), "miq_colquant_bwd: chan_min/chan_max must come together");
    MIQ_CHECK_ARG(!chan_min || (kmin && kmax && starts && workspace && covered >= 0 && covered <= C),
                  "miq_colquant_bwd: routing needs kmin, kmax, starts and a workspace");
    BwdSlabs slabs{};
    slabs.s[0] = BwdSlab{x, grad_out, scale, rzp, kmin, kmax, chan_min, chan_max, dx, workspace};
    if (dtype == MIQ_F32)
        return colquant_bwd_launch<float>(slabs, 1, chan_min != nullptr, starts, n_clusters, covered, nchunk, R, C, n_bits, phase,
                                          false, false, flat_ties ? 1 : 0, (hipStream_t)stream, "miq_colquant_bwd");
    if (dtype == MIQ_F16)
        return colquant_bwd_launch<__half>(slabs, 1, chan_min != nullptr, starts, n_clusters, covered, nchunk, R, C, n_bits, phase,
                                           false, false, flat_ties ? 1 : 0, (hipStream_t)stream, "miq_colquant_bwd");
    miq_set_error("miq_colquant_bwd: unsupported dtype %d", dtype);
    return -22;
}

// floats of workspace per slab of the fresh-statistics backward: the partial planes, the folded sums, the shares and
// the flag word, padded to a multiple of four floats
static inline int64_t fresh_bwd_ws_floats(int64_t R, int64_t C, int dtype) {
    const int64_t nchunk = miq_colstat_chunks(R, C, dtype);
    return ((3 * nchunk + 5) * C + 4 + 3) / 4 * 4;
}

extern "C" int64_t miq_act_fresh_bwd_workspace(int n, int64_t R, int64_t C, int dtype) {
    if (n <= 0 || R <= 0 || C <= 0) return 0;
    return (int64_t)n * fresh_bwd_ws_floats(R, C, dtype) * (int64_t)sizeof(float);
}

extern "C" int miq_act_fresh_quant_bwd(int n, const void* const* grad_out, const void* const* x, const void* const* stats,
                                       int64_t cluster_size, const int* starts, int n_clusters, int64_t covered, int n_bits,
                                       void* workspace, void* const* dx, int64_t R, int64_t C, int flat_ties, int dtype,
                                       void* stream) {
    if (n == 0 || R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(n > 0 && n <= kMaxBwdSlabs, "miq_act_fresh_quant_bwd: %d slabs (1..%d per call)", n, kMaxBwdSlabs);
    MIQ_CHECK_ARG(grad_out && x && stats && dx && starts && workspace && R > 0 && C > 0 && C < (1ll << 31),
                  "miq_act_fresh_quant_bwd: null pointer or bad shape");
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 15 && n_clusters >= 0 && covered >= 0 && covered <= C, "miq_act_fresh_quant_bwd: bad n_bits / cluster layout");
    MIQ_CHECK_ARG(cluster_size <= 0 || cluster_size * n_clusters == covered, "miq_act_fresh_quant_bwd: cluster_size * n_clusters != covered");
    MIQ_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15u) == 0, "miq_act_fresh_quant_bwd: workspace must be 16-byte aligned");
    if (dtype != MIQ_F32 && dtype != MIQ_F16) {
        miq_set_error("miq_act_fresh_quant_bwd: unsupported dtype %d", dtype);
        return -22;
    }
    const size_t es = dtype == MIQ_F16 ? 2 : 4;
    const int64_t stride = fresh_bwd_ws_floats(R, C, dtype);
    BwdSlabs slabs{};
    for (int i = 0; i < n; ++i) {
        MIQ_CHECK_ARG(grad_out[i] && x[i] && stats[i] && dx[i], "miq_act_fresh_quant_bwd: null pointer in slab %d", i);
        const char* v = static_cast<const char*>(stats[i]);  // [5, C]: merged min, merged max, scale, zero point, rounded zero point
        slabs.s[i] = BwdSlab{x[i], grad_out[i], v + 2 * C * es, v + 4 * C * es, v, v + C * es, nullptr, nullptr, dx[i],
                             static_cast<float*>(workspace) + i * stride};
    }
    int nchunk = miq_colstat_chunks(R, C, dtype);
    const bool merged = cluster_size > 0 && cluster_size <= kClusterMax && covered == C;
    // every 16-byte pack of channels inside one cluster: one parameter set per lane in the streaming kernel
    const bool uniform = cluster_size > 0 && covered == C && cluster_size % (dtype == MIQ_F16 ? 8 : 4) == 0;
    // The per-lane partials are so small that the row chunks can shrink: fp32 slabs stream 5 % faster in twice the
    // blocks (16-row chunks: one four-row batch per wave), fp16 slabs 8 % slower (measured on the Llama-2-7B sites)
    if (merged && uniform && dtype == MIQ_F32 && (R + 2 * nchunk - 1) / (2 * nchunk) >= 16) nchunk *= 2;
    if (dtype == MIQ_F32)
        return colquant_bwd_launch<float>(slabs, n, true, starts, n_clusters, covered, nchunk, R, C, n_bits, 0, merged, uniform,
                                          flat_ties ? 1 : 0, (hipStream_t)stream, "miq_act_fresh_quant_bwd");
    return colquant_bwd_launch<__half>(slabs, n, true, starts, n_clusters, covered, nchunk, R, C, n_bits, 0, merged, uniform,
                                       flat_ties ? 1 : 0, (hipStream_t)stream, "miq_act_fresh_quant_bwd");
}
