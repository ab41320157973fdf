// Row-wise uniform affine fake quantization with learnable weight clipping: forward and
// straight-through backward.  One team of threads owns one quantization row, keeps it in registers
// (16-byte global loads), reduces min/max with wave shuffles (+ one LDS exchange for rows wider than
// a wave), derives scale / zero point, then quantizes from registers and streams the result out.
// HBM traffic per call = read x once + write out once (forward); read g, x + write dx (backward).
//
// Semantics follow UniformAffineQuantizer.per_token_dynamic_calibration + fake_quant
// (reference quantize/quantizer.py:248-273,99-121) and Mix2AffineQuantizer.forward
// (quantize/quantizer_mix_v2.py:116-146) when per-row bit-widths are supplied.
#include <type_traits>
#include "miq_common.h"

namespace miq {

template <typename T>
struct UniArgs {
    const T* x;
    const T* up;            // [n_rows] activated LWC factors sigmoid(upbound_factor) / sigmoid(lowbound_factor), or nullptr
    const T* low;
    const float* row_bits;  // [n_rows] per-row integer bit-width (float) or nullptr
    int n_bits;
    int symmetric;
    int clamp_asym;         // clamp the asymmetric scale to [1e-5, 1e4] (fixed mixed-bit quantizer only)
};

struct UniCtx {
    float s;      // scale actually used
    float rzp;    // rounded zero point
    float qmax;   // 2^n - 1
    // kept for the backward pass
    float su, sl;      // sigmoid(up), sigmoid(low) (1 when LWC is off)
    float xmax, xmin;  // clipped bounds
    float s_raw;       // scale before clamping
    float qdiv;        // divisor that produced s_raw: qmax (asym) or 2^(n-1)-1 (sym)
};

template <typename T>
__device__ __forceinline__ UniCtx uniform_ctx(const UniArgs<T>& a, long long row, float rmin, float rmax) {
    UniCtx c;
    float n = a.row_bits ? a.row_bits[row] : (float)a.n_bits;
    c.qmax = exp2f(n) - 1.0f;
    c.su = 1.f;
    c.sl = 1.f;
    c.xmax = rmax;
    c.xmin = rmin;
    if (a.up) {
        c.su = to_f<T>(a.up[row]);
        c.sl = to_f<T>(a.low[row]);
        c.xmax = rnd<T>(c.su * rmax);
        c.xmin = rnd<T>(c.sl * rmin);
    }
    const float lo = rnd<T>(kClipLo), hi = kClipHi;
    if (a.symmetric) {
        const float amax = nmax(fabsf(c.xmax), fabsf(c.xmin));
        c.qdiv = exp2f(n - 1.0f) - 1.0f;
        c.s_raw = rnd<T>(amax / c.qdiv);
        // per-row bits (fixed mixed-bit quantizer) leave the symmetric scale unclamped
        c.s = a.row_bits ? c.s_raw : clampf(c.s_raw, lo, hi);
        c.rzp = c.qdiv;
    } else {
        c.qdiv = c.qmax;
        c.s_raw = rnd<T>(rnd<T>(c.xmax - c.xmin) / c.qmax);
        c.s = a.clamp_asym ? clampf(c.s_raw, lo, hi) : c.s_raw;
        const float zp = rnd<T>(-c.xmin / c.s);
        c.rzp = rintf(clampf(zp, -kClipHi, kClipHi));
    }
    return c;
}

// level before clamping: round_ste(x/s) + zp
template <typename T, class D>
__device__ __forceinline__ float quant_level(float x, const D& dv, const UniCtx& c, bool exact_int) {
    return affine_level<T>(x, dv, c.rzp, exact_int);
}

// One row per team, one launch wave of blocks after the other: a persistent variant that prefetches
// the team's next row while computing the current one was measured slower (the second register tile
// costs more occupancy than the overlap returns), and so was a one-wave-per-row tiling without the
// block barrier.
// the work of one team on one quantization row
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void uniform_fwd_row(const UniArgs<T>& a, const RowGeom& geo, long long row, int lane, float* lds,
                                                T* __restrict__ out, T* __restrict__ scale_out, T* __restrict__ rzp_out,
                                                uint8_t* __restrict__ codes) {
    long long base;
    int len;
    row_span(geo, row, base, len);
    RowTile<T, VEC, TEAM, NV> x;
    x.load_nt(a.x + base, len, lane);
    float rmin, rmax;
    row_minmax(x, len < geo.g, lds, rmin, rmax);
    const UniCtx c = uniform_ctx<T>(a, row, rmin, rmax);
    if (lane == 0) {
        scale_out[row] = from_f<T>(c.s);
        rzp_out[row] = from_f<T>(c.rzp);
    }
    const bool exact_int = ints_exact<T>(c.rzp, c.qmax);
    // instantiated for both values of exact_int (a per-element test would be a branch around every fp16 re-rounding)
    // and, in fp32, for the two forms of the quotient (a scale outside level_div's range is a row-uniform rarity)
    auto quantize_row = [&](auto exact_tag, const auto& dv) {
        constexpr bool EX = decltype(exact_tag)::value;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<T, VEC> o;
            Pack<uint8_t, VEC> cd;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xi = quant_level<T>(to_f<T>(x.v[k].e[e]), dv, c, EX);
                const float xc = clampf(xi, 0.f, c.qmax);
                o.e[e] = from_f<T>(affine_dequant<T>(xc, c.s, c.rzp, EX));
                if (codes) cd.e[e] = (xc == xc) ? (uint8_t)xc : (uint8_t)0;
            }
            const long long off = base + x.offset(k);
            st_out<T, VEC>(out + off, o, geo.stream_out);
            if (codes) st_pack<uint8_t, VEC>(codes + off, cd);
        }
    };
    if constexpr (sizeof(T) == 4) {
        if (level_div_ok(c.s)) quantize_row(std::true_type{}, LevelDivisor(c.s));
        else quantize_row(std::true_type{}, Divisor<T>(c.s));
    } else {
        // the usual row: integer levels exact in fp16 and a finite, non-zero scale; anything else takes the general form
        // (re-rounding every integer step is correct for every row, only slower)
        if (exact_int && half_div_ok(c.s)) quantize_row(std::true_type{}, HalfDivisor(c.s));
        else quantize_row(std::false_type{}, Divisor<T>(c.s));
    }
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_uniform_fwd(UniArgs<T> a, RowGeom geo, T* __restrict__ out, T* __restrict__ scale_out,
              T* __restrict__ rzp_out, uint8_t* __restrict__ codes) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;  // whole teams leave together; TEAM > 64 implies RPB == 1
    uniform_fwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, out, scale_out, rzp_out, codes);
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_uniform_fwd_batch(UniBatch<T> b, UniArgs<T> a, RowGeom geo) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= b.row_start[b.n]) return;
    long long local;
    const int m = batch_locate(b, row, local);
    a.x = b.x[m];
    a.up = b.up[m];
    a.low = b.low[m];
    uniform_fwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.out[m], b.a1[m], b.a2[m], nullptr);
}

// per-matrix vectors of per-row bit-widths for the batched launches of the fixed mixed-bit quantizer
struct BatchBits {
    const float* p[kMaxBatch];
};

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_uniform_fwd_bits_batch(UniBatch<T> b, BatchBits bits, UniArgs<T> a, RowGeom geo) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= b.row_start[b.n]) return;
    long long local;
    const int m = batch_locate(b, row, local);
    a.x = b.x[m];
    a.up = b.up[m];
    a.low = b.low[m];
    a.row_bits = bits.p[m];
    uniform_fwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.out[m], b.a1[m], b.a2[m], nullptr);
}

// Backward.  With m = [0 <= round(x/s)+zp <= qmax] and xd = clamp(..) - zp:
//   dx      = g*m                       (direct path)
//   dS      = sum_row g*(xd - m*x/s)    (gradient reaching the scale)
//   asym:   d_range = dS/qmax (masked by the scale clamp when it is active)
//           d_xmax = d_range, d_xmin = -d_range
//   sym:    d_amax = mask*dS/(2^(n-1)-1), routed to |xmax| or |xmin| (half each on a tie), times sign
//   LWC:    d_su = d_xmax*rowmax, d_sl = d_xmin*rowmin (gradients of the activated factors su, sl)
//   and the row max/min elements of x receive d_xmax*su / d_xmin*sl, split evenly among ties
//   (zero padding counts as a tie candidate and swallows its share).
template <typename T, int VEC, int TEAM, int NV, bool WANT_DX>
__device__ __forceinline__ void uniform_bwd_row(const UniArgs<T>& a, const RowGeom& geo, long long row, int lane, float* lds,
                                                const T* __restrict__ gout, T* __restrict__ dx, T* __restrict__ d_up,
                                                T* __restrict__ d_low) {
    long long base;
    int len;
    row_span(geo, row, base, len);
    RowTile<T, VEC, TEAM, NV> x, g;
    x.load_nt(a.x + base, len, lane);
    g.load_nt(gout + base, len, lane);
    const bool padded = len < geo.g;
    float rmin, rmax;
    row_minmax(x, padded, lds, rmin, rmax);
    const UniCtx c = uniform_ctx<T>(a, row, rmin, rmax);

    const bool exact_int = ints_exact<T>(c.rzp, c.qmax);
    float acc[3] = {0.f, 0.f, 0.f};  // dS, #elements == rmax, #elements == rmin
    static_assert(NV * VEC <= 64, "mask of un-clamped elements is one 64-bit word");
    unsigned long long inside = 0ull;  // bit k*VEC+e: the level of element (k, e) lies inside the grid
    auto row_sums = [&](auto exact_tag, const auto& dv) {
        constexpr bool EX = decltype(exact_tag)::value;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f<T>(x.v[k].e[e]);
                const float gv = to_f<T>(g.v[k].e[e]);
                const float q = dv.div(xv);
                const float xi = quant_level<T>(xv, dv, c, EX);
                const bool m = (xi >= 0.f) && (xi <= c.qmax);
                const float xd = clampf(xi, 0.f, c.qmax) - c.rzp;
                acc[0] += gv * (xd - (m ? q : 0.f));
                inside |= (m ? 1ull : 0ull) << (k * VEC + e);
                if (WANT_DX) {
                    acc[1] += (xv == rmax) ? 1.f : 0.f;
                    acc[2] += (xv == rmin) ? 1.f : 0.f;
                }
            }
        }
    };
    if constexpr (sizeof(T) == 4) {
        // (the quotient also enters the scale gradient, a tolerance-checked sum: below level_div's exact range it is
        // ~1e-24 of a level)
        if (level_div_ok(c.s)) row_sums(std::true_type{}, LevelDivisor(c.s));
        else row_sums(std::true_type{}, Divisor<T>(c.s));
    } else {
        if (exact_int && half_div_ok(c.s)) row_sums(std::true_type{}, HalfDivisor(c.s));
        else row_sums(std::false_type{}, Divisor<T>(c.s));
    }
    team_sum<TEAM, 3>(acc, lds);
    if (WANT_DX && padded) {
        const float npad = (float)(geo.g - len);
        if (rmax == 0.f) acc[1] += npad;
        if (rmin == 0.f) acc[2] += npad;
    }
    const float dS = acc[0];
    float d_xmax, d_xmin;
    if (a.symmetric) {
        const float lo = rnd<T>(kClipLo);
        const bool live = a.row_bits ? true : (c.s_raw >= lo && c.s_raw <= kClipHi);
        const float d_amax = live ? dS / c.qdiv : 0.f;
        const float A = fabsf(c.xmax), B = fabsf(c.xmin);
        const float da = (A > B) ? d_amax : (A == B ? 0.5f * d_amax : 0.f);
        const float db = (B > A) ? d_amax : (A == B ? 0.5f * d_amax : 0.f);
        d_xmax = da * sgnf(c.xmax);
        d_xmin = db * sgnf(c.xmin);
    } else {
        const float lo = rnd<T>(kClipLo);
        const bool live = a.clamp_asym ? (c.s_raw >= lo && c.s_raw <= kClipHi) : true;
        const float d_range = live ? dS / c.qmax : 0.f;
        d_xmax = d_range;
        d_xmin = -d_range;
    }
    if (a.up && lane == 0) {
        d_up[row] = from_f<T>(d_xmax * rmax);   // w.r.t. the activated factors
        d_low[row] = from_f<T>(d_xmin * rmin);
    }
    if (WANT_DX) {
        const float share_max = (d_xmax * c.su) / acc[1];
        const float share_min = (d_xmin * c.sl) / acc[2];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<T, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f<T>(x.v[k].e[e]);
                const float gv = to_f<T>(g.v[k].e[e]);
                // straight-through: the gradient passes where the level was not clamped (autograd forms
                // (g*s)/s here, which equals g to within one rounding)
                float d = ((inside >> (k * VEC + e)) & 1ull) ? gv : 0.f;
                if (xv == rmax) d += share_max;
                if (xv == rmin) d += share_min;
                o.e[e] = from_f<T>(d);
            }
            st_out<T, VEC>(dx + base + x.offset(k), o, geo.stream_out);
        }
    }
}

template <typename T, int VEC, int TEAM, int NV, bool WANT_DX>
// register budget: >= 4 waves per SIMD (two 512-thread blocks per CU) except for the widest fp16 tiles,
// whose 64 elements per thread would spill under that cap
__global__ void __launch_bounds__(block_of<TEAM>(), (sizeof(T) == 2 && NV == 8 && TEAM < 1024) ? 2 : 4)
k_uniform_bwd(UniArgs<T> a, RowGeom geo, const T* __restrict__ gout, T* __restrict__ dx,
              T* __restrict__ d_up, T* __restrict__ d_low) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    uniform_bwd_row<T, VEC, TEAM, NV, WANT_DX>(a, geo, row, lane, lds, gout, dx, d_up, d_low);
}

template <typename T, int VEC, int TEAM, int NV, bool WANT_DX>
__global__ void __launch_bounds__(block_of<TEAM>(), (sizeof(T) == 2 && NV == 8 && TEAM < 1024) ? 2 : 4)
k_uniform_bwd_batch(UniBatch<T> b, UniArgs<T> a, RowGeom geo) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= b.row_start[b.n]) return;
    long long local;
    const int m = batch_locate(b, row, local);
    a.x = b.x[m];
    a.up = b.up[m];
    a.low = b.low[m];
    uniform_bwd_row<T, VEC, TEAM, NV, WANT_DX>(a, geo, local, lane, lds, b.gout[m], b.out[m], b.a1[m], b.a2[m]);
}

template <typename T, int VEC, int TEAM, int NV, bool WANT_DX>
__global__ void __launch_bounds__(block_of<TEAM>(), (sizeof(T) == 2 && NV == 8 && TEAM < 1024) ? 2 : 4)
k_uniform_bwd_bits_batch(UniBatch<T> b, BatchBits bits, UniArgs<T> a, RowGeom geo) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= b.row_start[b.n]) return;
    long long local;
    const int m = batch_locate(b, row, local);
    a.x = b.x[m];
    a.up = b.up[m];
    a.low = b.low[m];
    a.row_bits = bits.p[m];
    uniform_bwd_row<T, VEC, TEAM, NV, WANT_DX>(a, geo, local, lane, lds, b.gout[m], b.out[m], b.a1[m], b.a2[m]);
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <typename T>
static int launch_uniform(bool backward, const void* gout, const void* x, const void* up, const void* low,
                          const float* row_bits, void* out_or_dx, void* a1, void* a2, void* codes,
                          long long O, long long I, long long group, int n_bits, int symmetric, int flags,
                          hipStream_t stream) {
    const char* name = backward ? "miq_rowquant_uniform_bwd" : "miq_rowquant_uniform_fwd";
    if (O == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(x && O > 0 && I > 0, "%s: null input pointer or bad shape (O=%lld I=%lld)", name, O, I);
    MIQ_CHECK_ARG(out_or_dx || backward, "%s: null output pointer", name);
    MIQ_CHECK_ARG(backward || (a1 && a2), "%s: null scale/rzp output pointer", name);
    MIQ_CHECK_ARG((up == nullptr) == (low == nullptr), "%s: up/low must both be given or both be null", name);
    MIQ_CHECK_ARG(row_bits || (n_bits >= 2 && n_bits <= 15), "%s: n_bits=%d outside [2,15]", name, n_bits);
    MIQ_CHECK_ARG(!codes || row_bits || n_bits <= 8, "%s: integer codes are uint8, n_bits must be <= 8", name);
    RowGeom geo;
    geo.I = (int)I;
    geo.g = group > 0 ? (int)group : (int)I;
    geo.G = (int)((I + geo.g - 1) / geo.g);
    geo.n_rows = O * geo.G;
    geo.stream_out = stream_out_of(O * I * (long long)sizeof(T));
    MIQ_CHECK_ARG(I < (1ll << 31) && geo.n_rows < (1ll << 31) * 4, "%s: matrix too large", name);
    constexpr int N = VecOf<T>::N;
    bool vec_ok = (I % N == 0) && (geo.g % N == 0) && aligned16(x) && aligned16(out_or_dx) &&
                  (!backward || aligned16(gout)) && (!codes || (reinterpret_cast<uintptr_t>(codes) % N) == 0);
    UniArgs<T> a;
    a.x = (const T*)x;
    a.up = (const T*)up;
    a.low = (const T*)low;
    a.row_bits = row_bits;
    a.n_bits = n_bits;
    a.symmetric = symmetric;
    a.clamp_asym = flags & 1;
    const int npack = vec_ok ? geo.g / N : geo.g;
    MIQ_CHECK_ARG(npack <= kRowMaxPacks, "%s: row of %d elements exceeds the register-tile capacity (%d packs)",
                  name, geo.g, kRowMaxPacks);
#define LAUNCH_FWD(VECN, TEAM, NV)                                                                     \
    {                                                                                                  \
        constexpr int B = block_of<TEAM>();                                                            \
        const long long nb = (geo.n_rows + (B / TEAM) - 1) / (B / TEAM);                               \
        hipLaunchKernelGGL((k_uniform_fwd<T, VECN, TEAM, NV>), dim3((unsigned)nb), dim3(B), 0, stream, \
                           a, geo, (T*)out_or_dx, (T*)a1, (T*)a2, (uint8_t*)codes);                    \
    }
#define LAUNCH_BWD(VECN, TEAM, NV)                                                                     \
    {                                                                                                  \
        constexpr int B = block_of<TEAM>();                                                            \
        const long long nb = (geo.n_rows + (B / TEAM) - 1) / (B / TEAM);                               \
        if (out_or_dx)                                                                                 \
            hipLaunchKernelGGL((k_uniform_bwd<T, VECN, TEAM, NV, true>), dim3((unsigned)nb), dim3(B),  \
                               0, stream, a, geo, (const T*)gout, (T*)out_or_dx, (T*)a1, (T*)a2);      \
        else                                                                                           \
            hipLaunchKernelGGL((k_uniform_bwd<T, VECN, TEAM, NV, false>), dim3((unsigned)nb), dim3(B), \
                               0, stream, a, geo, (const T*)gout, (T*)nullptr, (T*)a1, (T*)a2);        \
    }
    if (!backward) {
        if (vec_ok) {
#define L(TEAM, NV) LAUNCH_FWD(N, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        } else {
#define L(TEAM, NV) LAUNCH_FWD(1, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        }
    } else {
        if (vec_ok) {
#define L(TEAM, NV) LAUNCH_BWD(N, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        } else {
#define L(TEAM, NV) LAUNCH_BWD(1, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        }
    }
#undef LAUNCH_FWD
#undef LAUNCH_BWD
    MIQ_CHECK_LAUNCH(name);
    return 0;
}

template <typename T>
static int launch_uniform_batch(bool backward, int n, const void* const* gout, const void* const* x, const void* const* up,
                                const void* const* low, void* const* out_or_dx, void* const* a1, void* const* a2,
                                const int64_t* O, long long I, int n_bits, int symmetric, int flags, hipStream_t stream,
                                const float* const* row_bits = nullptr) {
    const char* name = row_bits ? (backward ? "miq_rowquant_mix2_bwd_batch" : "miq_rowquant_mix2_fwd_batch")
                                : (backward ? "miq_rowquant_uniform_bwd_batch" : "miq_rowquant_uniform_fwd_batch");
    if (n == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(n > 0 && n <= kMaxBatch && x && O && I > 0 && I < (1ll << 31), "%s: 1..%d matrices of one width expected", name, kMaxBatch);
    MIQ_CHECK_ARG(row_bits || (n_bits >= 2 && n_bits <= 15), "%s: n_bits=%d outside [2,15]", name, n_bits);
    MIQ_CHECK_ARG(backward ? (gout && a1 && a2) : (out_or_dx && a1 && a2), "%s: null pointer table", name);
    constexpr int N = VecOf<T>::N;
    UniBatch<T> b{};
    BatchBits bits{};
    const bool lwc = up != nullptr && up[0] != nullptr;
    const bool want_dx = backward && out_or_dx != nullptr && out_or_dx[0] != nullptr;
    bool vec_ok = (I % N == 0);
    long long rows = 0;
    b.n = 0;
    for (int i = 0; i < n; ++i) {
        MIQ_CHECK_ARG(O[i] >= 0, "%s: negative row count", name);
        if (O[i] == 0) continue;
        const int m = b.n++;
        b.x[m] = (const T*)x[i];
        b.up[m] = lwc ? (const T*)up[i] : nullptr;
        b.low[m] = lwc ? (const T*)low[i] : nullptr;
        b.gout[m] = backward ? (const T*)gout[i] : nullptr;
        b.out[m] = (backward && !want_dx) ? nullptr : (T*)out_or_dx[i];
        b.a1[m] = (backward && !lwc) ? nullptr : (T*)a1[i];
        b.a2[m] = (backward && !lwc) ? nullptr : (T*)a2[i];
        MIQ_CHECK_ARG(b.x[m] && (!lwc || (b.up[m] && b.low[m])) && (!backward || b.gout[m]) && (backward || (b.out[m] && b.a1[m] && b.a2[m])) &&
                          (!want_dx || b.out[m]) && (!(backward && lwc) || (b.a1[m] && b.a2[m])),
                      "%s: null pointer in entry %d (LWC and dx must be given for all matrices or for none)", name, i);
        if (row_bits) {
            bits.p[m] = row_bits[i];
            MIQ_CHECK_ARG(bits.p[m], "%s: null bit-width vector in entry %d", name, i);
        }
        vec_ok = vec_ok && aligned16(b.x[m]) && aligned16(b.out[m]) && aligned16(b.gout[m]);
        b.row_start[m] = rows;
        rows += O[i];
    }
    if (b.n == 0) return 0;
    for (int m = b.n; m <= kMaxBatch; ++m) b.row_start[m] = rows;
    RowGeom geo;
    geo.I = (int)I;
    geo.g = (int)I;
    geo.G = 1;
    geo.n_rows = rows;
    geo.stream_out = stream_out_of(rows * I * (long long)sizeof(T));
    UniArgs<T> a{};
    a.row_bits = nullptr;  // the kernels with bit-width vectors set it per matrix
    a.n_bits = n_bits;
    a.symmetric = symmetric;
    a.clamp_asym = flags & 1;
    const int npack = vec_ok ? geo.g / N : geo.g;
    MIQ_CHECK_ARG(npack <= kRowMaxPacks, "%s: row of %d elements exceeds the register-tile capacity", name, geo.g);
#define LAUNCH_FWD(VECN, TEAM, NV)                                                                                  \
    {                                                                                                               \
        constexpr int B = block_of<TEAM>();                                                                         \
        const long long nb = (rows + (B / TEAM) - 1) / (B / TEAM);                                                  \
        if (row_bits)                                                                                               \
            hipLaunchKernelGGL((k_uniform_fwd_bits_batch<T, VECN, TEAM, NV>), dim3((unsigned)nb), dim3(B), 0, stream, b, bits, a, geo); \
        else                                                                                                        \
            hipLaunchKernelGGL((k_uniform_fwd_batch<T, VECN, TEAM, NV>), dim3((unsigned)nb), dim3(B), 0, stream, b, a, geo); \
    }
#define LAUNCH_BWD(VECN, TEAM, NV)                                                                                  \
    {                                                                                                               \
        constexpr int B = block_of<TEAM>();                                                                         \
        const long long nb = (rows + (B / TEAM) - 1) / (B / TEAM);                                                  \
        if (row_bits && want_dx)                                                                                    \
            hipLaunchKernelGGL((k_uniform_bwd_bits_batch<T, VECN, TEAM, NV, true>), dim3((unsigned)nb), dim3(B), 0, stream, b, bits, a, geo); \
        else if (row_bits)                                                                                          \
            hipLaunchKernelGGL((k_uniform_bwd_bits_batch<T, VECN, TEAM, NV, false>), dim3((unsigned)nb), dim3(B), 0, stream, b, bits, a, geo); \
        else if (want_dx)                                                                                           \
            hipLaunchKernelGGL((k_uniform_bwd_batch<T, VECN, TEAM, NV, true>), dim3((unsigned)nb), dim3(B), 0, stream, b, a, geo); \
        else                                                                                                        \
            hipLaunchKernelGGL((k_uniform_bwd_batch<T, VECN, TEAM, NV, false>), dim3((unsigned)nb), dim3(B), 0, stream, b, a, geo); \
    }
    if (!backward) {
        if (vec_ok) {
#define L(TEAM, NV) LAUNCH_FWD(N, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        } else {
#define L(TEAM, NV) LAUNCH_FWD(1, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        }
    } else {
        if (vec_ok) {
#define L(TEAM, NV) LAUNCH_BWD(N, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        } else {
#define L(TEAM, NV) LAUNCH_BWD(1, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        }
    }
#undef LAUNCH_FWD
#undef LAUNCH_BWD
    MIQ_CHECK_LAUNCH(name);
    return 0;
}

}  // namespace miq

extern "C" int miq_rowquant_uniform_fwd(const void* x, const void* up, const void* low, const float* row_bits,
                                        void* out, void* scale_out, void* rzp_out, void* codes_out,
                                        int64_t O, int64_t I, int64_t group, int n_bits, int symmetric,
                                        int flags, int dtype, void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_uniform<float>(false, nullptr, x, up, low, row_bits, out, scale_out, rzp_out, codes_out,
                                          O, I, group, n_bits, symmetric, flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_uniform<__half>(false, nullptr, x, up, low, row_bits, out, scale_out, rzp_out, codes_out,
                                           O, I, group, n_bits, symmetric, flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_uniform_fwd: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_uniform_bwd(const void* grad_out, const void* x, const void* up, const void* low,
                                        const float* row_bits, void* dx, void* d_up, void* d_low,
                                        int64_t O, int64_t I, int64_t group, int n_bits, int symmetric,
                                        int flags, int dtype, void* stream) {
    if (O == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(grad_out, "miq_rowquant_uniform_bwd: null grad_out");
    MIQ_CHECK_ARG(!up || (d_up && d_low), "miq_rowquant_uniform_bwd: d_up/d_low required when up/low are given");
    if (dtype == MIQ_F32)
        return miq::launch_uniform<float>(true, grad_out, x, up, low, row_bits, dx, d_up, d_low, nullptr,
                                          O, I, group, n_bits, symmetric, flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_uniform<__half>(true, grad_out, x, up, low, row_bits, dx, d_up, d_low, nullptr,
                                           O, I, group, n_bits, symmetric, flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_uniform_bwd: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_uniform_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                              void* const* out, void* const* scale_out, void* const* rzp_out,
                                              const int64_t* O, int64_t I, int n_bits, int symmetric, int flags, int dtype,
                                              void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_uniform_batch<float>(false, n, nullptr, x, up, low, out, scale_out, rzp_out, O, I, n_bits, symmetric,
                                                flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_uniform_batch<__half>(false, n, nullptr, x, up, low, out, scale_out, rzp_out, O, I, n_bits, symmetric,
                                                 flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_uniform_fwd_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_uniform_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                              const void* const* low, void* const* dx, void* const* d_up, void* const* d_low,
                                              const int64_t* O, int64_t I, int n_bits, int symmetric, int flags, int dtype,
                                              void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_uniform_batch<float>(true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, n_bits, symmetric, flags,
                                                (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_uniform_batch<__half>(true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, n_bits, symmetric, flags,
                                                 (hipStream_t)stream);
    miq_set_error("miq_rowquant_uniform_bwd_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_mix2_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                           const float* const* row_bits, void* const* out, void* const* scale_out,
                                           void* const* rzp_out, const int64_t* O, int64_t I, int symmetric, int flags,
                                           int dtype, void* stream) {
    MIQ_CHECK_ARG(row_bits, "miq_rowquant_mix2_fwd_batch: null bit-width table");
    if (dtype == MIQ_F32)
        return miq::launch_uniform_batch<float>(false, n, nullptr, x, up, low, out, scale_out, rzp_out, O, I, 0, symmetric, flags,
                                                (hipStream_t)stream, row_bits);
    if (dtype == MIQ_F16)
        return miq::launch_uniform_batch<__half>(false, n, nullptr, x, up, low, out, scale_out, rzp_out, O, I, 0, symmetric, flags,
                                                 (hipStream_t)stream, row_bits);
    miq_set_error("miq_rowquant_mix2_fwd_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_mix2_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                           const void* const* low, const float* const* row_bits, void* const* dx,
                                           void* const* d_up, void* const* d_low, const int64_t* O, int64_t I, int symmetric,
                                           int flags, int dtype, void* stream) {
    MIQ_CHECK_ARG(row_bits, "miq_rowquant_mix2_bwd_batch: null bit-width table");
    if (dtype == MIQ_F32)
        return miq::launch_uniform_batch<float>(true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, 0, symmetric, flags,
                                                (hipStream_t)stream, row_bits);
    if (dtype == MIQ_F16)
        return miq::launch_uniform_batch<__half>(true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, 0, symmetric, flags,
                                                 (hipStream_t)stream, row_bits);
    miq_set_error("miq_rowquant_mix2_bwd_batch: unsupported dtype %d", dtype);
    return -22;
}
