// Exponential / logarithmic row-wise weight quantizers (power-of-two levels), forward + backward.
//
//   v2 ("exp2"): levels {0, +-1, +-2, ..., +-top}, top = 2^(2^(n-1)-2); scale = clamp(absmax/(2*top));
//                forward and the hand-written "STE" backward follow FakeQuant.forward/backward
//                (reference quantize/quantizer_exp_v2.py:21-105) and per_channel_dynamic_calibration
//                (:186-204).  This is the class QuantLinear instantiates (quantize/int_linear.py:5,48).
//   v1 ("exp1"): levels +-2^e, e in [-(2^(n-1)-1), 0], no zero level; symmetric (scale = clamp(absmax))
//                or asymmetric around the row midpoint (reference quantize/quantizer_exp.py:92-166);
//                the backward is what autograd derives through log2 / pow / clamp / round_ste.
//
// The level decision uses log2f; a value within a few ulp of a level threshold (|q| = 1.5*2^k for v2,
// 2^(k+0.5) for v1) can land on the neighbouring level compared with another math library.
#include "miq_common.h"

namespace miq {

constexpr float kLn2 = 0.693147180559945309f;  // (float)math.log(2)
constexpr float kInvLn2 = 1.442695040888963407f;

template <typename T>
struct ExpArgs {
    const T* x;
    const T* up;
    const T* low;
    int n_bits;
    int symmetric;  // v1 only
};

// v / 1.5f as the IEEE quotient through the corrected reciprocal product (see CorrectedDivisor; v lies in
// [1e-4, 1e4] here, far inside the range where the correction is exact)
__device__ __forceinline__ float div_1p5(float v) {
    constexpr float kInv = 1.0f / 1.5f;  // correctly rounded at compile time
    const float q0 = v * kInv;
    return __builtin_fmaf(__builtin_fmaf(-q0, 1.5f, v), kInv, q0);
}

// ------------------------------------------------------------------------------------------ v2
struct Exp2Ctx {
    ClipCtx clip;
    float s, s_raw, top, qdiv;
};

template <typename T>
__device__ __forceinline__ Exp2Ctx exp2_ctx(const ExpArgs<T>& a, long long row, float rmin, float rmax) {
    Exp2Ctx c;
    c.clip = lwc_clip<T>(a.up, a.low, row, rmin, rmax);
    c.top = (float)(1 << ((1 << (a.n_bits - 1)) - 2));
    c.qdiv = 2.f * c.top;
    const float amax = nmax(fabsf(c.clip.xmax), fabsf(c.clip.xmin));
    c.s_raw = rnd<T>(amax / c.qdiv);
    c.s = clampf(c.s_raw, rnd<T>(kClipLo), kClipHi);
    return c;
}

struct Exp2Elem {
    float sign, mag, lvl, signed_lvl;  // lvl: level before clamping (0 for the zero bin)
    bool is_zero, below;
};

template <typename T, class D>
__device__ __forceinline__ Exp2Elem exp2_elem(float xv, const Exp2Ctx& c, const D& dv) {
    Exp2Elem r;
    const float q = dv.div(xv);
    r.sign = sgnf(q);
    const float a0 = fabsf(q);
    r.is_zero = a0 < 0.5f;
    r.mag = clampf(a0, rnd<T>(1e-4f), 1e4f);
    float t = rnd<T>(rnd<T>(__log2f(rnd<T>(div_1p5(r.mag)))) + 0.5f);
    r.below = t < -0.5f;
    if (r.below) t = 0.f;
    float e = (t == t) ? ldexpf(1.0f, (int)rintf(t)) : t;  // 2^integer: exact and representable in T (t <= 14); NaN stays NaN
    if (r.is_zero) e = 0.f;
    r.lvl = e;
    r.signed_lvl = rnd<T>(clampf(e, -c.top, c.top) * r.sign);
    return r;
}

// fp32 rows in which every quotient is finite (finite extremes, a scale that is not NaN, |x / s| far below the range
// where the residual of the corrected product could overflow) take element functions without the NaN-preserving forms
// and without the logarithm.
//
// The reference picks the level 2^round(log2(a / 1.5) + 0.5) of a magnitude a = |x / s| >= 0.5.  With y = RN(a / 1.5) that
// is the smallest power of two >= y, and for y an exact power of two 2^j the tie of round-half-to-even: 2^(j + (j & 1)).
// RN(a / 1.5) is monotone in a and crosses 2^j exactly at a = 1.5 * 2^j (the neighbouring floats of 1.5 * 2^j map below
// and above 2^j), so the level is a function of comparisons of a itself:
//   a < 0.5 -> 0 | a <= 1.5 -> 1 | a < 3 -> 2 | a <= 6 -> 4 | a < 12 -> 8 | a <= 24 -> 16 | ...
// In the reference's float arithmetic t = fl(fl(log2 y) + 0.5) the addition rounds too, which moves each threshold by a
// few ulp: with a correctly rounded log2 the float just below 3 already gives t = 1.5 and level 4, while 1.5 and 6 stay
// on the lower level.  Up to 3 bits (levels 0, 1, 2, 4, and "beyond the top") three or four compare-and-selects against
// those exact floats replace divide, log2, round, ldexp.  From 4 bits on the exponent field of y decides, with the
// same exactness (exp2_level_of).  Rows that are not plain, and fp16 rows, go through the logarithm instruction, whose
// last bit may put an element within a few ulp of a threshold on the neighbouring level (the stated tolerance of the
// exponent grids).
__device__ __forceinline__ bool exp2_plain_row(const Exp2Ctx& c, float rmin, float rmax) {
    const float bound = __builtin_fmaxf(fabsf(rmin), fabsf(rmax)) / c.s;
    return (rmin == rmin) && (rmax == rmax) && (bound <= 0x1p60f);  // a NaN scale gives a NaN bound
}
// unclamped level of a magnitude a0 >= 0.5 (anything for smaller ones: the caller zeroes them)
template <bool SMALL>
__device__ __forceinline__ float exp2_level_of(float a0) {
    if constexpr (SMALL) {  // n_bits <= 3: top <= 4; 8 stands for every level beyond it
        return a0 <= 1.5f ? 1.f : (a0 < 0x1.7ffffep+1f ? 2.f : (a0 <= 6.f ? 4.f : 8.f));  // 0x1.7ffffep+1: the float below 3
    } else {
        // y in [2^j, 2^(j+1)) has level 2^(j+1), except within an ulp or two of the ends of the interval, where the
        // rounding of fl(log2 y) + 0.5 decides (a correctly rounded log2; found by brute force over +-16 ulp of every
        // power of two): the lower end of an even j stays on level 2^j for the first d floats, its upper end is on
        // level 2^(j+2) for the last d floats, d = 1 for j < 4 and 2 for j = 4, 6; an odd j has no exceptions.  (From
        // j = 8 on the offsets grow to 3 and 5 ulp; those levels lie beyond the top of every supported bit-width.)
        const unsigned b = __float_as_uint(div_1p5(__builtin_amdgcn_fmed3f(a0, 1e-4f, 1e4f)));  // y > 0
        const int eb = (int)(b >> 23);  // biased exponent, j = eb - 127
        const unsigned m = b & 0x7FFFFFu;
        const unsigned d = eb < 127 + 4 ? 1u : 2u;
        const int even = (eb & 1);      // eb odd <=> j even
        int kb = eb + 1 + (even ? (int)(m >= 0x800000u - d) - (int)(m < d) : 0);
        kb = kb < 127 ? 127 : kb;       // y < 1: level 2^0 (incl. the reference's branch t < -0.5)
        return __uint_as_float((unsigned)kb << 23);
    }
}
template <bool SMALL>
__device__ __forceinline__ float exp2_plain_level(float xv, const Exp2Ctx& c, float rs) {
    const float q0 = xv * rs;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, c.s, xv), rs, q0);  // the IEEE quotient for 2^-80 <= |q0|; below it
    const float a0 = fabsf(q);                                             // the level is 0 whichever way it rounds
    float e = exp2_level_of<SMALL>(a0);
    if (a0 < 0.5f) e = 0.f;
    return __builtin_copysignf(__builtin_fminf(e, c.top), q + 0.f);
}

// The backward of such a row whose magnitudes also stay below the reference's clamp at 1e4.  An element is ALIVE when its
// level is neither the zero bin (a < 0.5), nor taken from the threshold branch (t < -0.5: a < 0.75), nor beyond the top.
// For an alive element the reference's chain gives grad_q = g * s * lvl / a and dx = grad_q / s = g * lvl / a, and its
// two contributions to the scale gradient, g * lvl * sign and grad_q * (-x / s^2) = -g * lvl * sign, cancel: only
// elements that are not alive contribute g * clamp(lvl) * sign (zero for the zero bin).  dx: the gradient w.r.t. x;
// ds: the element's term of the scale gradient.
template <bool SMALL>
__device__ __forceinline__ void exp2_plain_grad(float xv, float gv, const Exp2Ctx& c, float rs, float& dx, float& ds) {
    const float q0 = xv * rs;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, c.s, xv), rs, q0);
    const float a0 = fabsf(q);
    const float lvl = exp2_level_of<SMALL>(a0);
    const bool alive = (a0 >= 0.75f) & (lvl <= c.top);
    const float sat = a0 < 0.5f ? 0.f : __builtin_fminf(lvl, c.top);
    dx = alive ? (gv * lvl) * __builtin_amdgcn_rcpf(a0) : 0.f;
    ds = alive ? 0.f : gv * __builtin_copysignf(sat, q + 0.f);
}
__device__ __forceinline__ bool exp2_plain_grad_row(const Exp2Ctx& c, float rmin, float rmax) {
    const float bound = __builtin_fmaxf(fabsf(rmin), fabsf(rmax)) / c.s;
    return (rmin == rmin) && (rmax == rmax) && (bound <= 1e4f);
}

// the work of one team on one quantization row
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void exp2_fwd_row(const ExpArgs<T>& a, const RowGeom& geo, long long row, int lane, float* lds,
                                             T* __restrict__ out, T* __restrict__ scale_out) {
    long long base;
    int len;
    row_span(geo, row, base, len);
    RowTile<T, VEC, TEAM, NV> x;
    x.load_nt(a.x + base, len, lane);
    float rmin, rmax;
    row_minmax(x, len < geo.g, lds, rmin, rmax);
    const Exp2Ctx c = exp2_ctx<T>(a, row, rmin, rmax);
    const typename FastDivisorOf<T>::type dv(c.s);
    if (lane == 0) scale_out[row] = from_f<T>(c.s);
    if constexpr (sizeof(T) == 4) {
        if (exp2_plain_row(c, rmin, rmax)) {  // a property of the row: tested once, not per element
            const float rs = 1.0f / c.s;
            auto plain_row = [&](auto small_tag) {
                constexpr bool SMALL = decltype(small_tag)::value;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
                    Pack<T, VEC> o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o.e[e] = from_f<T>(exp2_plain_level<SMALL>(to_f<T>(x.v[k].e[e]), c, rs) * c.s);
                    st_out<T, VEC>(out + base + x.offset(k), o, geo.stream_out);
                }
            };
            if (a.n_bits <= 3) plain_row(std::true_type{});
            else plain_row(std::false_type{});
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!x.has(k)) continue;
        Pack<T, VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.e[e] = from_f<T>(exp2_elem<T>(to_f<T>(x.v[k].e[e]), c, dv).signed_lvl * c.s);
        st_out<T, VEC>(out + base + x.offset(k), o, geo.stream_out);
    }
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_exp2_fwd(ExpArgs<T> a, RowGeom geo, T* __restrict__ out, T* __restrict__ scale_out) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    exp2_fwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, out, scale_out);
}

// several matrices of one row width in one launch (the table of the uniform quantizer's batched kernels)
template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_exp2_fwd_batch(UniBatch<T> b, ExpArgs<T> a, RowGeom geo) {
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
    exp2_fwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.out[m], b.a1[m]);
}

// Minimum waves per SIMD asked of the backward kernels (fp32): the 512-lane tiles of 6 and 7 packs must stay inside 128
// registers (two blocks per CU; at 130 to 154 registers a CU holds ONE 512-lane block), the 256 x 4 tile of v2 inside 96
// (five waves per SIMD instead of four).
template <typename T, int TEAM, int NV, int WHICH>
constexpr int exp_bwd_waves() {
    if (sizeof(T) != 4) return 1;
    if (TEAM == 512 && (NV == 6 || NV == 7)) return 4;
    // (v2: 99 -> 96 registers with 20 bytes of spills, backward 0.69 -> 0.71 of peak on the Llama-2-7B layer; v1 would
    // spill 52 bytes and lose: 0.68 -> 0.61)
    if (WHICH == 2 && TEAM == 256 && NV == 4) return 5;
    return 1;
}

// Reference backward (quantizer_exp_v2.py:65-105), element by element:
//   gq   = g*s, negated where sign == -1; zeroed where the level saturates, is the zero bin, or
//          the threshold branch was taken;  grad_q = ((ln2*lvl)*gq / ln2 / mag) * sign
//   dx   = grad_q / s;   d_scale = g*signed_lvl + grad_q * (-x / s^2)   (summed over the row)
// followed by autograd through clamp(scale), max(|xmax|,|xmin|), the sigmoid clipping and amax/amin.
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void exp2_bwd_row(const ExpArgs<T>& a, const RowGeom& geo, long long row, int lane, float* lds,
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
    // fp32: the elements at the row extremes (rarely in more than two lanes of a row) are counted and corrected only by
    // the lanes that hold them
    constexpr bool kLaneTies = sizeof(T) == 4;
    bool at_extreme = true;
    if constexpr (kLaneTies) row_minmax(x, padded, lds, rmin, rmax, at_extreme);
    else row_minmax(x, padded, lds, rmin, rmax);
    const Exp2Ctx c = exp2_ctx<T>(a, row, rmin, rmax);
    const typename BwdDivisorOf<T>::type dv(c.s);
    const float hi_dead = rnd<T>(c.top + 0.0001f), lo_dead = rnd<T>(-c.top + 0.0001f);
    auto grad_q = [&](float gv, const Exp2Elem& el) {
        float gq = gv * c.s;
        if (el.sign == -1.f) gq = -gq;
        // the comparison constants are rounded to the tensor dtype, as the reference's `q_exp.ge(qmax+0.0001)` does
        // (bitwise | and a select: with || and an early return the compiler branches around the product for every element)
        const bool dead = (el.lvl >= hi_dead) | (el.lvl <= lo_dead) | el.is_zero | el.below;
        // gradients are tolerance-checked sums: the hardware reciprocal (1 ulp) instead of an IEEE division (__frcp_rn is one)
        const float v = ((kLn2 * el.lvl) * gq) * kInvLn2 * __builtin_amdgcn_rcpf(el.mag) * el.sign;
        return dead ? 0.f : v;
    };
    const float inv_s = 1.0f / c.s, neg_inv_s2 = -1.0f / (c.s * c.s);
    // The per-element gradient w.r.t. x/s is needed again for dx: keep it instead of redoing the log2 / level
    // search.  fp32 reuses the registers of g (not read again); fp16 uses a side array while it fits.
    constexpr bool kInPlace = sizeof(T) == 4;
    constexpr bool kSide = !kInPlace && NV * VEC <= 32;
    float side[kSide ? NV * VEC : 1];
    float acc[3] = {0.f, 0.f, 0.f};
    auto row_sums = [&]() {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f<T>(x.v[k].e[e]), gv = to_f<T>(g.v[k].e[e]);
                const Exp2Elem el = exp2_elem<T>(xv, c, dv);
                const float gqq = grad_q(gv, el);
                if constexpr (kInPlace) g.v[k].e[e] = from_f<T>(gqq);
                if constexpr (kSide) side[k * VEC + e] = gqq;
                acc[0] += gv * el.signed_lvl + gqq * (xv * neg_inv_s2);
                if constexpr (!kLaneTies) {
                    acc[1] += (xv == rmax) ? 1.f : 0.f;
                    acc[2] += (xv == rmin) ? 1.f : 0.f;
                }
            }
        }
    };
    // fp32 rows of ordinary values: the closed form per element (g's registers then hold dx itself, not grad_q)
    float post = inv_s;
    if constexpr (sizeof(T) == 4) {
        if (exp2_plain_grad_row(c, rmin, rmax)) {  // a property of the row, tested once
            auto plain_sums = [&](auto small_tag) {
                constexpr bool SMALL = decltype(small_tag)::value;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        float d, ds;
                        exp2_plain_grad<SMALL>(to_f<T>(x.v[k].e[e]), to_f<T>(g.v[k].e[e]), c, inv_s, d, ds);
                        g.v[k].e[e] = from_f<T>(d);
                        acc[0] += ds;
                    }
                }
            };
            if (a.n_bits <= 3) plain_sums(std::true_type{});
            else plain_sums(std::false_type{});
            post = 1.f;
        } else {
            row_sums();
        }
    } else {
        row_sums();
    }
    if constexpr (kLaneTies) {
        if (at_extreme) {
            x.for_each([&](int, int, float xv) {
                acc[1] += (xv == rmax) ? 1.f : 0.f;
                acc[2] += (xv == rmin) ? 1.f : 0.f;
            });
        }
    }
    team_sum<TEAM, 3>(acc, lds);
    if (padded) {
        const float npad = (float)(geo.g - len);
        if (rmax == 0.f) acc[1] += npad;
        if (rmin == 0.f) acc[2] += npad;
    }
    const bool live = c.s_raw >= rnd<T>(kClipLo) && c.s_raw <= kClipHi;
    float d_xmax, d_xmin;
    route_absmax(live ? acc[0] / c.qdiv : 0.f, c.clip, d_xmax, d_xmin);
    if (a.up && lane == 0) write_lwc_grads<T>(d_up, d_low, row, rmin, rmax, d_xmax, d_xmin);
    if constexpr (kLaneTies) {  // (fp32: g holds the per-element gradient w.r.t. x / s)
        if (dx) {
            if (at_extreme) {  // one lane-level branch around two compares per element
                const float share_max = (d_xmax * c.clip.su) / acc[1];
                const float share_min = (d_xmin * c.clip.sl) / acc[2];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float xv = to_f<T>(x.v[k].e[e]);
                        float d = to_f<T>(g.v[k].e[e]) * post;
                        if (xv == rmax) d += share_max;
                        if (xv == rmin) d += share_min;
                        g.v[k].e[e] = from_f<T>(d);
                    }
                }
            } else if (post != 1.f) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g.v[k].e[e] = from_f<T>(to_f<T>(g.v[k].e[e]) * post);
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!x.has(k)) continue;
                st_out<T, VEC>(dx + base + x.offset(k), g.v[k], geo.stream_out);
            }
        }
        return;
    }
    if (dx) {
        const float share_max = (d_xmax * c.clip.su) / acc[1];
        const float share_min = (d_xmin * c.clip.sl) / acc[2];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<T, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f<T>(x.v[k].e[e]), gv = to_f<T>(g.v[k].e[e]);
                float gqq;
                if constexpr (kInPlace) gqq = gv;  // overwritten above
                else if constexpr (kSide) gqq = side[k * VEC + e];
                else gqq = grad_q(gv, exp2_elem<T>(xv, c, dv));
                float d = gqq * inv_s;
                if (xv == rmax) d += share_max;
                if (xv == rmin) d += share_min;
                o.e[e] = from_f<T>(d);
            }
            st_out<T, VEC>(dx + base + x.offset(k), o, geo.stream_out);
        }
    }
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>(), (exp_bwd_waves<T, TEAM, NV, 2>()))
k_exp2_bwd(ExpArgs<T> a, RowGeom geo, const T* __restrict__ gout, T* __restrict__ dx, T* __restrict__ d_up,
           T* __restrict__ d_low) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    exp2_bwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, gout, dx, d_up, d_low);
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>(), (exp_bwd_waves<T, TEAM, NV, 2>()))
k_exp2_bwd_batch(UniBatch<T> b, ExpArgs<T> a, RowGeom geo) {
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
    exp2_bwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.gout[m], b.out[m], b.a1[m], b.a2[m]);
}

// ------------------------------------------------------------------------------------------ v1
struct Exp1Ctx {
    ClipCtx clip;
    float s, s_raw, mid, e_lo, p_lo;  // p_lo = 2^e_lo, the smallest level
    bool has_mid;
};

template <typename T>
__device__ __forceinline__ Exp1Ctx exp1_ctx(const ExpArgs<T>& a, long long row, float rmin, float rmax) {
    Exp1Ctx c;
    c.clip = lwc_clip<T>(a.up, a.low, row, rmin, rmax);
    c.e_lo = -(float)((1 << (a.n_bits - 1)) - 1);
    c.p_lo = ldexpf(1.0f, -((1 << (a.n_bits - 1)) - 1));
    const float lo = rnd<T>(kClipLo);
    if (a.symmetric) {
        c.s_raw = nmax(fabsf(c.clip.xmax), fabsf(c.clip.xmin));
        c.has_mid = false;
        c.mid = 0.f;
    } else {
        c.s_raw = rnd<T>(rnd<T>(c.clip.xmax - c.clip.xmin) / 2.f);
        c.mid = rnd<T>(rnd<T>(c.clip.xmax + c.clip.xmin) / 2.f);
        c.has_mid = true;
    }
    c.s = clampf(c.s_raw, lo, kClipHi);
    return c;
}

struct Exp1Elem {
    float v, u, sign, mag, lround, p;  // v = x - mid, u = v/s, mag = clamp|u|, lround = round(log2 mag), p = 2^e
};

template <typename T, class D>
__device__ __forceinline__ Exp1Elem exp1_elem(float xv, const Exp1Ctx& c, const D& dv) {
    Exp1Elem r;
    r.v = c.has_mid ? rnd<T>(xv - c.mid) : xv;
    r.sign = sgnf(r.v);
    r.u = dv.div(r.v);
    r.mag = clampf(fabsf(r.u), rnd<T>(kClipLo), kClipHi);
    const float l = rnd<T>(__log2f(r.mag));
    r.lround = rnd<T>(rnd<T>(rintf(l) - l) + l);
    const float ec = clampf(r.lround, c.e_lo, 0.f);
    r.p = (ec == ec) ? ldexpf(1.0f, (int)ec) : ec;  // 2^integer in [2^e_lo, 1]: exact in T
    return r;
}

template <typename T>
__device__ __forceinline__ float exp1_out(const Exp1Elem& el, const Exp1Ctx& c) {
    float o = rnd<T>(rnd<T>(el.p * el.sign) * c.s);
    if (c.has_mid) o = rnd<T>(o + c.mid);
    return o;
}

// fp32 rows in which every quotient is finite (see exp2_plain_row): the v1 element without the NaN-preserving forms.
// Same bits as exp1_elem + exp1_out there; the sign is copied from v, and a zero v gives the zero sgnf gives.
__device__ __forceinline__ bool exp1_plain_row(const Exp1Ctx& c, float rmin, float rmax) {
    const float bound = 2.f * __builtin_fmaxf(fabsf(rmin), fabsf(rmax)) / c.s;
    return (rmin == rmin) && (rmax == rmax) && (bound <= 0x1p60f) && (fabsf(c.mid) <= 0x1p100f);
}
// The level 2^clamp(round(log2 a), e_lo, 0) of a magnitude a without the logarithm: round(log2 a) = k exactly when
// 2^(k - 0.5) < a < 2^(k + 0.5).  The thresholds are irrational, so which of the two floats around one is the first of
// the upper level is decided by the last bit of fl(log2 a); kExp1From[j] is the first float that a correctly rounded
// log2 sends to level 2^-j.  Up to 3 bits (levels 1, 1/2, 1/4, 1/8) three compare-and-selects against those floats; from
// 4 bits on the exponent field and a comparison of the significand with the boundary's own first float.  Rows that are
// not plain, and fp16 rows, go through the logarithm instruction, whose last bit may put an element within a few ulp
// of a threshold on the neighbouring level (the stated tolerance of the exponent grids).
__device__ constexpr float kExp1From[4] = {0x1.6a09e6p-1f, 0x1.6a09e8p-2f, 0x1.6a09e6p-3f, 0x1.6a09eap-4f};
constexpr float kExp1Above = 0x1.6a09e8p+0f;  // the first float whose log2 rounds to 1
// the first float of level 2^k, k in [-16, 1] (see exp1_level_of)
__device__ __forceinline__ float exp1_first_of(int k) {
    const int idx = k + 16 < 0 ? 0 : (k + 16 > 17 ? 17 : k + 16);
    const unsigned off = (unsigned)((idx < 16 ? 0x8796a6a4c4c4c4c0ull >> (4 * idx) : 0x87ull >> (4 * (idx - 16))) & 15u);
    return __uint_as_float(((unsigned)(k - 1 + 127) << 23) | (0x3504ECu + off));
}
template <bool SMALL>
__device__ __forceinline__ float exp1_level_of(float a, const Exp1Ctx& c) {
    if constexpr (SMALL) {
        const float p = a >= 0x1.6a09e6p-1f ? 1.f : (a >= 0x1.6a09e8p-2f ? 0.5f : (a >= 0x1.6a09e6p-3f ? 0.25f : 0.125f));
        return __builtin_fmaxf(p, c.p_lo);
    } else {
        // a in [2^e, 2^(e + 1)) has level e + 1 from the first float that a correctly rounded log2 sends there: the
        // significand 0x3504EC + kExp1Sig(e + 1), a 4-bit entry per boundary -16 .. 1 (all boundaries of [1e-5, 1e4] that
        // the clamp to [e_lo, 0] leaves visible; brute force over +-24 ulp of every 2^(k - 0.5))
        const unsigned b = __float_as_uint(a);  // a in [1e-5, 1e4]
        const int e = (int)(b >> 23) - 127;
        const int idx = e + 17 < 0 ? 0 : (e + 17 > 17 ? 17 : e + 17);
        const unsigned sig = 0x3504ECu + (unsigned)(((idx < 16 ? 0x8796a6a4c4c4c4c0ull >> (4 * idx) : 0x87ull >> (4 * (idx - 16)))) & 15u);
        int k = e + ((b & 0x7FFFFFu) >= sig ? 1 : 0);
        const int lo = (int)c.e_lo;
        k = k < lo ? lo : (k > 0 ? 0 : k);
        return __uint_as_float((unsigned)(k + 127) << 23);
    }
}
template <bool SMALL>
__device__ __forceinline__ float exp1_plain_out(float xv, const Exp1Ctx& c, float rs) {
    const float v = c.has_mid ? xv - c.mid : xv;
    const float q0 = v * rs;
    const float u = __builtin_fmaf(__builtin_fmaf(-q0, c.s, v), rs, q0);  // below 2^-80 the magnitude clamps to 1e-5 either way
    const float p = exp1_level_of<SMALL>(__builtin_amdgcn_fmed3f(fabsf(u), kClipLo, kClipHi), c);
    const float ps = (v != 0.f) ? __builtin_copysignf(p, v) : 0.f;
    const float o = ps * c.s;
    return c.has_mid ? o + c.mid : o;
}
// The backward of such a row.  An element is LIVE when round(log2 |u|) lies in [e_lo, 0] and |u| in [1e-5, 1e4]:
// 2^(e_lo - 0.5) < |u| < 2^0.5 (the clamp bounds lie outside that interval).  For a live element autograd's chain
// through pow, round_ste, log2, clamp and abs gives du = g * s * p / |u|, so dv = du / s = g * p / |u|, and its two
// contributions to the scale gradient, g * p * sign and du * (-v / s^2) = -g * p * sign, cancel; an element that is not
// live contributes g * p * sign (p: the clamped level) and has dv = 0.  The midpoint gradient collects g - dv.
template <bool SMALL>
__device__ __forceinline__ void exp1_plain_grad(float xv, float gv, const Exp1Ctx& c, float rs, float t_lo, float& dv, float& ds) {
    const float v = c.has_mid ? xv - c.mid : xv;
    const float q0 = v * rs;
    const float u = __builtin_fmaf(__builtin_fmaf(-q0, c.s, v), rs, q0);
    const float a = fabsf(u);
    const float p = exp1_level_of<SMALL>(__builtin_amdgcn_fmed3f(a, kClipLo, kClipHi), c);
    const bool live = (a >= t_lo) & (a < kExp1Above);
    dv = live ? (gv * p) * __builtin_amdgcn_rcpf(a) : 0.f;
    ds = (live | (v == 0.f)) ? 0.f : gv * __builtin_copysignf(p, v);
}

// the work of one team on one quantization row
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void exp1_fwd_row(const ExpArgs<T>& a, const RowGeom& geo, long long row, int lane, float* lds,
                                             T* __restrict__ out, T* __restrict__ scale_out, T* __restrict__ mid_out) {
    long long base;
    int len;
    row_span(geo, row, base, len);
    RowTile<T, VEC, TEAM, NV> x;
    x.load_nt(a.x + base, len, lane);
    float rmin, rmax;
    row_minmax(x, len < geo.g, lds, rmin, rmax);
    const Exp1Ctx c = exp1_ctx<T>(a, row, rmin, rmax);
    const typename FastDivisorOf<T>::type dv(c.s);
    if (lane == 0) {
        scale_out[row] = from_f<T>(c.s);
        if (mid_out) mid_out[row] = from_f<T>(c.mid);
    }
    if constexpr (sizeof(T) == 4) {
        if (exp1_plain_row(c, rmin, rmax)) {  // a property of the row: tested once, not per element
            const float rs = 1.0f / c.s;
            auto plain_row = [&](auto small_tag) {
                constexpr bool SMALL = decltype(small_tag)::value;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
                    Pack<T, VEC> o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o.e[e] = from_f<T>(exp1_plain_out<SMALL>(to_f<T>(x.v[k].e[e]), c, rs));
                    st_out<T, VEC>(out + base + x.offset(k), o, geo.stream_out);
                }
            };
            if (a.n_bits <= 3) plain_row(std::true_type{});
            else plain_row(std::false_type{});
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!x.has(k)) continue;
        Pack<T, VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.e[e] = from_f<T>(exp1_out<T>(exp1_elem<T>(to_f<T>(x.v[k].e[e]), c, dv), c));
        st_out<T, VEC>(out + base + x.offset(k), o, geo.stream_out);
    }
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_exp1_fwd(ExpArgs<T> a, RowGeom geo, T* __restrict__ out, T* __restrict__ scale_out, T* __restrict__ mid_out) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    exp1_fwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, out, scale_out, mid_out);
}

// several matrices of one row width in one launch (a2 = the midpoints, or null)
template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_exp1_fwd_batch(UniBatch<T> b, ExpArgs<T> a, RowGeom geo) {
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
    exp1_fwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.out[m], b.a1[m], b.a2[m]);
}

// Autograd of v1.  With P = 2^e * sign, live = [e_lo <= round(log2 mag) <= 0] & [1e-5 <= |u| <= 1e4]:
//   du = live * (ln2 * 2^e * sign * s * g) / (mag * ln2) * sgn(u)          (through pow, round_ste, log2, clamp, abs)
//   dv = du / s;  d_s = sum g*P + sum du * (-v / s^2);  d_mid = sum g - sum dv
//   sym : d|max| = mask * d_s;      asym: d_range = mask * d_s / 2, d_xmax = d_range + d_mid/2, d_xmin = -d_range + d_mid/2
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void exp1_bwd_row(const ExpArgs<T>& a, const RowGeom& geo, long long row, int lane, float* lds,
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
    // fp32: the elements at the row extremes are counted and corrected only by the lanes that hold them (k_exp2_bwd)
    constexpr bool kLaneTies = sizeof(T) == 4;
    bool at_extreme = true;
    if constexpr (kLaneTies) row_minmax(x, padded, lds, rmin, rmax, at_extreme);
    else row_minmax(x, padded, lds, rmin, rmax);
    const Exp1Ctx c = exp1_ctx<T>(a, row, rmin, rmax);
    const typename BwdDivisorOf<T>::type dv(c.s);
    const float alo = rnd<T>(kClipLo);
    auto d_u = [&](float gv, const Exp1Elem& el) {
        // (bitwise & and a select: with && and an early return the compiler branches around the product for every element)
        const bool live = (el.lround >= c.e_lo) & (el.lround <= 0.f) & (fabsf(el.u) >= alo) & (fabsf(el.u) <= kClipHi);
        const float v = ((kLn2 * el.p) * ((gv * c.s) * el.sign)) * __builtin_amdgcn_rcpf(el.mag * kLn2) * sgnf(el.u);
        return live ? v : 0.f;
    };
    const float inv_s = 1.0f / c.s, neg_inv_s2 = -1.0f / (c.s * c.s);
    constexpr bool kInPlace = sizeof(T) == 4;  // see k_exp2_bwd
    constexpr bool kSide = !kInPlace && NV * VEC <= 32;
    float side[kSide ? NV * VEC : 1];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // d_s, d_mid, #max, #min
    auto row_sums = [&]() {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f<T>(x.v[k].e[e]), gv = to_f<T>(g.v[k].e[e]);
                const Exp1Elem el = exp1_elem<T>(xv, c, dv);
                const float du = d_u(gv, el);
                if constexpr (kInPlace) g.v[k].e[e] = from_f<T>(du);
                if constexpr (kSide) side[k * VEC + e] = du;
                acc[0] += gv * (el.p * el.sign) + du * (el.v * neg_inv_s2);
                acc[1] += gv - du * inv_s;
                if constexpr (!kLaneTies) {
                    acc[2] += (xv == rmax) ? 1.f : 0.f;
                    acc[3] += (xv == rmin) ? 1.f : 0.f;
                }
            }
        }
    };
    // fp32 rows of ordinary values: the closed form per element (g's registers then hold dx itself)
    float post = inv_s;
    if constexpr (sizeof(T) == 4) {
        if (exp1_plain_row(c, rmin, rmax)) {  // a property of the row, tested once
            // the first float of the lowest level e_lo = -(2^(n_bits - 1) - 1); from 6 bits on e_lo lies below the clamp of
            // the magnitude at 1e-5, which then ends the live interval
            const float t_lo = a.n_bits <= 3 ? kExp1From[(1 << (a.n_bits - 1)) - 1]
                                             : (c.e_lo >= -16.f ? exp1_first_of((int)c.e_lo) : kClipLo);
            auto plain_sums = [&](auto small_tag) {
                constexpr bool SMALL = decltype(small_tag)::value;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float gv = to_f<T>(g.v[k].e[e]);
                        float d, ds;
                        exp1_plain_grad<SMALL>(to_f<T>(x.v[k].e[e]), gv, c, inv_s, t_lo, d, ds);
                        g.v[k].e[e] = from_f<T>(d);
                        acc[0] += ds;
                        acc[1] += gv - d;
                    }
                }
            };
            if (a.n_bits <= 3) plain_sums(std::true_type{});
            else plain_sums(std::false_type{});
            post = 1.f;
        } else {
            row_sums();
        }
    } else {
        row_sums();
    }
    if constexpr (kLaneTies) {
        if (at_extreme) {
            x.for_each([&](int, int, float xv) {
                acc[2] += (xv == rmax) ? 1.f : 0.f;
                acc[3] += (xv == rmin) ? 1.f : 0.f;
            });
        }
    }
    team_sum<TEAM, 4>(acc, lds);
    if (padded) {
        const float npad = (float)(geo.g - len);
        if (rmax == 0.f) acc[2] += npad;
        if (rmin == 0.f) acc[3] += npad;
    }
    const bool live_s = c.s_raw >= alo && c.s_raw <= kClipHi;
    float d_xmax, d_xmin;
    if (a.symmetric) {
        route_absmax(live_s ? acc[0] : 0.f, c.clip, d_xmax, d_xmin);
    } else {
        const float d_range = live_s ? acc[0] / 2.f : 0.f;
        d_xmax = d_range + acc[1] / 2.f;
        d_xmin = -d_range + acc[1] / 2.f;
    }
    if (a.up && lane == 0) write_lwc_grads<T>(d_up, d_low, row, rmin, rmax, d_xmax, d_xmin);
    if constexpr (kLaneTies) {  // (fp32: g holds the per-element gradient w.r.t. the quotient)
        if (dx) {
            if (at_extreme) {
                const float share_max = (d_xmax * c.clip.su) / acc[2];
                const float share_min = (d_xmin * c.clip.sl) / acc[3];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float xv = to_f<T>(x.v[k].e[e]);
                        float d = to_f<T>(g.v[k].e[e]) * post;
                        if (xv == rmax) d += share_max;
                        if (xv == rmin) d += share_min;
                        g.v[k].e[e] = from_f<T>(d);
                    }
                }
            } else if (post != 1.f) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if (!x.has(k)) continue;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) g.v[k].e[e] = from_f<T>(to_f<T>(g.v[k].e[e]) * post);
                }
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!x.has(k)) continue;
                st_out<T, VEC>(dx + base + x.offset(k), g.v[k], geo.stream_out);
            }
        }
        return;
    }
    if (dx) {
        const float share_max = (d_xmax * c.clip.su) / acc[2];
        const float share_min = (d_xmin * c.clip.sl) / acc[3];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<T, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = to_f<T>(x.v[k].e[e]), gv = to_f<T>(g.v[k].e[e]);
                float du;
                if constexpr (kInPlace) du = gv;  // overwritten above
                else if constexpr (kSide) du = side[k * VEC + e];
                else du = d_u(gv, exp1_elem<T>(xv, c, dv));
                float d = du * inv_s;
                if (xv == rmax) d += share_max;
                if (xv == rmin) d += share_min;
                o.e[e] = from_f<T>(d);
            }
            st_out<T, VEC>(dx + base + x.offset(k), o, geo.stream_out);
        }
    }
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>(), (exp_bwd_waves<T, TEAM, NV, 1>()))
k_exp1_bwd(ExpArgs<T> a, RowGeom geo, const T* __restrict__ gout, T* __restrict__ dx, T* __restrict__ d_up,
           T* __restrict__ d_low) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    exp1_bwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, gout, dx, d_up, d_low);
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>(), (exp_bwd_waves<T, TEAM, NV, 1>()))
k_exp1_bwd_batch(UniBatch<T> b, ExpArgs<T> a, RowGeom geo) {
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
    exp1_bwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.gout[m], b.out[m], b.a1[m], b.a2[m]);
}

template <typename T>
static int launch_exp(int which, bool backward, const void* gout, const void* x, const void* up, const void* low,
                      void* out_or_dx, void* a1, void* a2, long long O, long long I, long long group, int n_bits,
                      int symmetric, int flags, hipStream_t stream) {
    const char* name = which == 2 ? (backward ? "miq_rowquant_exp2_bwd" : "miq_rowquant_exp2_fwd")
                                  : (backward ? "miq_rowquant_exp1_bwd" : "miq_rowquant_exp1_fwd");
    if (O == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(x && O > 0 && I > 0, "%s: null input pointer or bad shape", name);
    MIQ_CHECK_ARG((up == nullptr) == (low == nullptr), "%s: up/low must both be given or both be null", name);
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 4, "%s: n_bits=%d outside [2,4]", name, n_bits);
    MIQ_CHECK_ARG(backward ? (gout != nullptr) : (out_or_dx && a1), "%s: null pointer", name);
    MIQ_CHECK_ARG(!backward || !up || (a1 && a2), "%s: d_up/d_low required when up/low are given", name);
    RowGeom geo;
    MIQ_CHECK_ARG(make_geom(O, I, group, geo), "%s: matrix too large", name);
    geo.stream_out = stream_out_of(O * I * (long long)sizeof(T));
    constexpr int N = VecOf<T>::N;
    const bool vec_ok = (I % N == 0) && (geo.g % N == 0) && is_aligned16(x) && (!out_or_dx || is_aligned16(out_or_dx)) &&
                        (!backward || is_aligned16(gout));
    MIQ_CHECK_ARG((vec_ok ? geo.g / N : geo.g) <= kRowMaxPacks, "%s: row of %d elements exceeds the register tile", name, geo.g);
    ExpArgs<T> a;
    a.x = (const T*)x;
    a.up = (const T*)up;
    a.low = (const T*)low;
    a.n_bits = n_bits;
    a.symmetric = symmetric;
    if (which == 2) {
        if (!backward) MIQ_ROW_LAUNCH(k_exp2_fwd, T, vec_ok, geo, stream, a, geo, (T*)out_or_dx, (T*)a1);
        else MIQ_ROW_LAUNCH_SIZED(k_exp2_bwd, T, vec_ok, geo, stream, a, geo, (const T*)gout, (T*)out_or_dx, (T*)a1, (T*)a2);
    } else {
        if (!backward) MIQ_ROW_LAUNCH(k_exp1_fwd, T, vec_ok, geo, stream, a, geo, (T*)out_or_dx, (T*)a1, (T*)a2);
        else MIQ_ROW_LAUNCH_SIZED(k_exp1_bwd, T, vec_ok, geo, stream, a, geo, (const T*)gout, (T*)out_or_dx, (T*)a1, (T*)a2);
    }
    MIQ_CHECK_LAUNCH(name);
    return 0;
}

// several matrices of one row width (whole rows, no groups): one launch each way.  a2: backward d_low; forward of v1 the
// midpoints (may be null), unused by v2
template <typename T>
static int launch_exp_batch(int which, bool backward, int n, const void* const* gout, const void* const* x, const void* const* up,
                            const void* const* low, void* const* out_or_dx, void* const* a1, void* const* a2, const int64_t* O,
                            long long I, int n_bits, int symmetric, int flags, hipStream_t stream) {
    const char* name = which == 2 ? (backward ? "miq_rowquant_exp2_bwd_batch" : "miq_rowquant_exp2_fwd_batch")
                                  : (backward ? "miq_rowquant_exp1_bwd_batch" : "miq_rowquant_exp1_fwd_batch");
    if (n == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(n > 0 && n <= kMaxBatch && x && O && I > 0 && I < (1ll << 31), "%s: 1..%d matrices of one width expected", name, kMaxBatch);
    MIQ_CHECK_ARG(n_bits >= 2 && n_bits <= 4, "%s: n_bits=%d outside [2,4]", name, n_bits);
    MIQ_CHECK_ARG(backward ? (gout != nullptr) : (out_or_dx && a1), "%s: null pointer table", name);
    constexpr int N = VecOf<T>::N;
    UniBatch<T> b{};
    const bool lwc = up != nullptr && up[0] != nullptr;
    const bool want_dx = backward && out_or_dx != nullptr && out_or_dx[0] != nullptr;
    MIQ_CHECK_ARG(!backward || !lwc || (a1 && a2), "%s: d_up/d_low tables required when up/low are given", name);
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
        b.a2[m] = (backward && lwc) ? (T*)a2[i] : ((!backward && which == 1 && a2) ? (T*)a2[i] : nullptr);
        MIQ_CHECK_ARG(b.x[m] && (!lwc || (b.up[m] && b.low[m])) && (!backward || b.gout[m]) && (backward || (b.out[m] && b.a1[m])) &&
                          (!want_dx || b.out[m]) && (!(backward && lwc) || (b.a1[m] && b.a2[m])),
                      "%s: null pointer in entry %d (LWC and dx must be given for all matrices or for none)", name, i);
        vec_ok = vec_ok && is_aligned16(b.x[m]) && is_aligned16(b.out[m]) && is_aligned16(b.gout[m]);
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
    ExpArgs<T> a{};
    a.n_bits = n_bits;
    a.symmetric = which == 2 ? 1 : symmetric;
    const int npack = vec_ok ? geo.g / N : geo.g;
    MIQ_CHECK_ARG(npack <= kRowMaxPacks, "%s: row of %d elements exceeds the register tile", name, geo.g);
#define LAUNCH_ONE(KERN, VECN, TEAM, NV)                                                                       \
    {                                                                                                          \
        constexpr int B = block_of<TEAM>();                                                                    \
        const long long nb = (rows + (B / TEAM) - 1) / (B / TEAM);                                             \
        hipLaunchKernelGGL((KERN<T, VECN, TEAM, NV>), dim3((unsigned)nb), dim3(B), 0, stream, b, a, geo);      \
    }
#define DISPATCH(KERN, TABLE_ESZ)                       \
    if (vec_ok) {                                       \
        MIQ_ROW_DISPATCH_E(TABLE_ESZ, npack, KERN##_V); \
    } else {                                            \
        MIQ_ROW_DISPATCH_E(TABLE_ESZ, npack, KERN##_S); \
    }
#define F2_V(TEAM, NV) LAUNCH_ONE(k_exp2_fwd_batch, N, TEAM, NV)
#define F2_S(TEAM, NV) LAUNCH_ONE(k_exp2_fwd_batch, 1, TEAM, NV)
#define B2_V(TEAM, NV) LAUNCH_ONE(k_exp2_bwd_batch, N, TEAM, NV)
#define B2_S(TEAM, NV) LAUNCH_ONE(k_exp2_bwd_batch, 1, TEAM, NV)
#define F1_V(TEAM, NV) LAUNCH_ONE(k_exp1_fwd_batch, N, TEAM, NV)
#define F1_S(TEAM, NV) LAUNCH_ONE(k_exp1_fwd_batch, 1, TEAM, NV)
#define B1_V(TEAM, NV) LAUNCH_ONE(k_exp1_bwd_batch, N, TEAM, NV)
#define B1_S(TEAM, NV) LAUNCH_ONE(k_exp1_bwd_batch, 1, TEAM, NV)
    // (the forward kernels take the 4-byte tile table for every element size, the backward kernels the table of the
    // element's own size, as the single-matrix launches do)
    if (which == 2) {
        if (!backward) { DISPATCH(F2, 4) } else { DISPATCH(B2, sizeof(T)) }
    } else {
        if (!backward) { DISPATCH(F1, 4) } else { DISPATCH(B1, sizeof(T)) }
    }
#undef F2_V
#undef F2_S
#undef B2_V
#undef B2_S
#undef F1_V
#undef F1_S
#undef B1_V
#undef B1_S
#undef DISPATCH
#undef LAUNCH_ONE
    MIQ_CHECK_LAUNCH(name);
    return 0;
}

}  // namespace miq

#define MIQ_EXP_ENTRY(WHICH, BWD, ...)                                                                      \
    if (dtype == MIQ_F32) return miq::launch_exp<float>(WHICH, BWD, __VA_ARGS__, (hipStream_t)stream);      \
    if (dtype == MIQ_F16) return miq::launch_exp<__half>(WHICH, BWD, __VA_ARGS__, (hipStream_t)stream);     \
    miq_set_error("exp quantizer: unsupported dtype %d", dtype);                                            \
    return -22;

extern "C" int miq_rowquant_exp2_fwd(const void* x, const void* up, const void* low, void* out, void* scale_out,
                                     int64_t O, int64_t I, int64_t group, int n_bits, int flags, int dtype, void* stream) {
    MIQ_EXP_ENTRY(2, false, nullptr, x, up, low, out, scale_out, nullptr, O, I, group, n_bits, 1, flags)
}

extern "C" int miq_rowquant_exp2_bwd(const void* grad_out, const void* x, const void* up, const void* low, void* dx,
                                     void* d_up, void* d_low, int64_t O, int64_t I, int64_t group, int n_bits, int flags,
                                     int dtype, void* stream) {
    MIQ_EXP_ENTRY(2, true, grad_out, x, up, low, dx, d_up, d_low, O, I, group, n_bits, 1, flags)
}

extern "C" int miq_rowquant_exp1_fwd(const void* x, const void* up, const void* low, void* out, void* scale_out,
                                     void* mid_out, int64_t O, int64_t I, int64_t group, int n_bits, int symmetric,
                                     int flags, int dtype, void* stream) {
    MIQ_EXP_ENTRY(1, false, nullptr, x, up, low, out, scale_out, mid_out, O, I, group, n_bits, symmetric, flags)
}

extern "C" int miq_rowquant_exp1_bwd(const void* grad_out, const void* x, const void* up, const void* low, void* dx,
                                     void* d_up, void* d_low, int64_t O, int64_t I, int64_t group, int n_bits,
                                     int symmetric, int flags, int dtype, void* stream) {
    MIQ_EXP_ENTRY(1, true, grad_out, x, up, low, dx, d_up, d_low, O, I, group, n_bits, symmetric, flags)
}

extern "C" int miq_rowquant_exp2_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                           void* const* out, void* const* scale_out, const int64_t* O, int64_t I, int n_bits,
                                           int flags, int dtype, void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_exp_batch<float>(2, false, n, nullptr, x, up, low, out, scale_out, nullptr, O, I, n_bits, 1, flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_exp_batch<__half>(2, false, n, nullptr, x, up, low, out, scale_out, nullptr, O, I, n_bits, 1, flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_exp2_fwd_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_exp2_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                           const void* const* low, void* const* dx, void* const* d_up, void* const* d_low,
                                           const int64_t* O, int64_t I, int n_bits, int flags, int dtype, void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_exp_batch<float>(2, true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, n_bits, 1, flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_exp_batch<__half>(2, true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, n_bits, 1, flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_exp2_bwd_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_exp1_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                           void* const* out, void* const* scale_out, void* const* mid_out, const int64_t* O,
                                           int64_t I, int n_bits, int symmetric, int flags, int dtype, void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_exp_batch<float>(1, false, n, nullptr, x, up, low, out, scale_out, mid_out, O, I, n_bits, symmetric, flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_exp_batch<__half>(1, false, n, nullptr, x, up, low, out, scale_out, mid_out, O, I, n_bits, symmetric, flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_exp1_fwd_batch: unsupported dtype %d", dtype);
    return -22;
}

extern "C" int miq_rowquant_exp1_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                           const void* const* low, void* const* dx, void* const* d_up, void* const* d_low,
                                           const int64_t* O, int64_t I, int n_bits, int symmetric, int flags, int dtype,
                                           void* stream) {
    if (dtype == MIQ_F32)
        return miq::launch_exp_batch<float>(1, true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, n_bits, symmetric, flags, (hipStream_t)stream);
    if (dtype == MIQ_F16)
        return miq::launch_exp_batch<__half>(1, true, n, grad_out, x, up, low, dx, d_up, d_low, O, I, n_bits, symmetric, flags, (hipStream_t)stream);
    miq_set_error("miq_rowquant_exp1_bwd_batch: unsupported dtype %d", dtype);
    return -22;
}
