// Shared device-side building blocks for the fake-quantization kernels (gfx950 / CDNA4, wave64).
//
// Numerics contract (matches PyTorch's CPU kernels op by op):
//   * every elementwise op is computed in fp32 and, for fp16 tensors, rounded to fp16 right after
//     the op (rnd<T>()), because PyTorch materialises a half tensor after each op;
//   * division is IEEE-correct fp32 division (hipcc's default), rounding is round-half-to-even;
//   * min/max reductions and clamps propagate NaN the way torch.amin/amax/clamp do.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define MIQ_F32 0
#define MIQ_F16 1

namespace miq {

constexpr float kClipLo = 1e-5f;   // CLIPMIN of the reference (quantize/quantizer.py:10)
constexpr float kClipHi = 1e4f;

// ---------------------------------------------------------------------------------------------
// per-dtype rounding / conversion
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
// PyTorch rounds the fp32 result of an op to fp16 (two roundings).  The empty asm keeps the fp32 value
// in a register: otherwise the compiler folds "multiply/add, then convert" into v_fma_mixlo_f16, which
// rounds the exact result ONCE; that differs at fp16 midpoints whenever an operand is not itself an
// fp16 value (e.g. x * 0.99f gives 35.59375 instead of 35.5625 for x = 35.9375).
__device__ __forceinline__ __half f2h_twice_rounded(float v) {
    asm("" : "+v"(v));
    return __float2half(v);
}
template <> __device__ __forceinline__ float rnd<__half>(float v) { return __half2float(f2h_twice_rounded(v)); }

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }

template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return f2h_twice_rounded(v); }

// widest vector (16 bytes) per dtype
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<__half> { static constexpr int N = 8; };

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T e[VEC]; };

template <int BYTES> struct RawVec;
template <> struct RawVec<16> { typedef unsigned int type __attribute__((ext_vector_type(4))); };
template <> struct RawVec<8> { typedef unsigned int type __attribute__((ext_vector_type(2))); };
template <> struct RawVec<4> { typedef unsigned int type; };
template <> struct RawVec<2> { typedef unsigned short type; };
template <> struct RawVec<1> { typedef unsigned char type; };

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> ld_pack(const T* p) {
    return *reinterpret_cast<const Pack<T, VEC>*>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void st_pack(T* p, const Pack<T, VEC>& v) {
    *reinterpret_cast<Pack<T, VEC>*>(p) = v;
}
// Non-temporal store for outputs far larger than the caches (kStreamOutBytes and more per launch): the lines are not
// kept, dirty, in L2 and the memory-side cache, from where their write-back would fall on the kernels that follow.
// Measured on MI355X over the Llama-2-7B block's operator sequence (row quantizers forward and backward, activation
// quantizers forward and backward, gathers; 2.8 GB written per pass): 1.179 -> 1.144 ms with the row and activation
// quantizer outputs written this way; about 30 us of write-back leave whichever kernels came next.  Smaller outputs
// keep the default policy: their consumer finds them in cache.
// The 16-byte form is written as the instruction itself: next to a plain store of the same pack under a run-time
// policy flag (st_out), the compiler folds the two stores into one and drops the hint.  (A store the compiler does not
// count only makes its vmcnt waits for loads longer, never shorter: loads return in order.)
template <typename T, int VEC>
__device__ __forceinline__ void st_pack_nt(T* p, const Pack<T, VEC>& v) {
    using V = typename RawVec<sizeof(Pack<T, VEC>)>::type;
    V r;
    __builtin_memcpy(&r, &v, sizeof(r));
    if constexpr (sizeof(Pack<T, VEC>) == 16) {
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(r) : "memory");
    } else {
        __builtin_nontemporal_store(r, reinterpret_cast<V*>(p));
    }
}
// store of an output pack under the launch's policy (wave-uniform flag)
template <typename T, int VEC>
__device__ __forceinline__ void st_out(T* p, const Pack<T, VEC>& v, int stream_out) {
    if (stream_out) {
        st_pack_nt<T, VEC>(p, v);
    } else {
        st_pack<T, VEC>(p, v);
    }
}
constexpr long long kStreamOutBytes = 16ll << 20;
static inline int stream_out_of(long long bytes) { return bytes >= kStreamOutBytes ? 1 : 0; }
// Non-temporal load for the LAST use of a streamed operand (the weight and the incoming gradient of a
// weight backward, the source of a gather, the activation in the kernel that quantizes it, the activation and
// the gradient in the streaming activation backward, the operands of the exponent and mixed-bit row kernels).
// Measured on MI355X: +4 % on the row backward, +12 % on the gather, 181 -> 167 us on the activation quantize
// pair ([2048, 4096] x 9), 198 -> 193 us on its backward, +1 to +4 % on the exponent and mixed-bit kernels.
// Not for operands that a later kernel of the same pass reads again (the activation in its statistics kernel:
// -5 %), and not for the weight of the equivalent-transformation forward, whose small working set is read again
// from cache by the next launches (28 -> 31 us on [4096, 4096]).
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> ld_pack_nt(const T* p) {
    using V = typename RawVec<sizeof(Pack<T, VEC>)>::type;
    const V r = __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    Pack<T, VEC> o;
    __builtin_memcpy(&o, &r, sizeof(r));
    return o;
}

// ---------------------------------------------------------------------------------------------
// NaN-propagating min / max / clamp (torch semantics)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float nmin(float a, float b) { return (b < a || b != b) ? b : a; }
__device__ __forceinline__ float nmax(float a, float b) { return (b > a || b != b) ? b : a; }
// std::min(std::max(v, lo), hi): NaN stays NaN
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    v = (v < lo) ? lo : v;
    return (hi < v) ? hi : v;
}
__device__ __forceinline__ float sgnf(float v) { return (float)((0.f < v) - (v < 0.f)); }

// ---------------------------------------------------------------------------------------------
// x / s rounded to T, for a divisor that is reused many times.
//   float: plain IEEE division (the kernels that use it are bound by memory, the ~12 VALU ops are hidden).
//   half : x and s carry 11-bit significands, so one Newton correction on a product with the
//          correctly rounded reciprocal gives EXACTLY the IEEE fp32 quotient: the residual
//          e = x - q0*s is exact in an fma, and the true quotient of two 11-bit numbers can never sit
//          within 2^-36 of an fp32 rounding boundary.  3 VALU ops instead of the ~12 of a full
//          division; divisors that are 0, inf or NaN take the plain division.
// ---------------------------------------------------------------------------------------------
template <typename T> struct Divisor;
template <> struct Divisor<float> {
    float s;
    __device__ __forceinline__ explicit Divisor(float s_) : s(s_) {}
    __device__ __forceinline__ float div(float x) const { return x / s; }
};
// The fp32 quotient for kernels that are bound by arithmetic (exponent and mixed-bit grids): the correctly
// rounded reciprocal rs = RN(1/s) and one FMA correction (Markstein): q0 = x*rs, r = fma(-q0, s, x) (exact),
// q = fma(r, rs, q0); 5 VALU ops with the range test.  Bit-identical to x / s whenever 2^-40 <= |s| <= 2^40 and
// 2^-80 <= |q0| <= 2^100 (tools/ubench/divcheck.hip: 1.4e11 random operand pairs and all 2^23 significands of x
// against 32 special divisors, no mismatch); everything else (zeros, infinities, NaN, quotients near the ends of
// the exponent range) takes the plain division.  The branch costs a few percent where the division was hidden
// behind memory traffic, which is why the uniform kernels keep Divisor<float>.
struct CorrectedDivisor {
    float s, rs;
    bool fast;
    __device__ __forceinline__ explicit CorrectedDivisor(float s_) : CorrectedDivisor(s_, 1.0f / s_) {}
    // rs_ must be the IEEE quotient 1.0f / s_
    __device__ __forceinline__ CorrectedDivisor(float s_, float rs_) : s(s_), rs(rs_) {
        const float a = fabsf(s_);
        fast = (a >= 0x1p-40f) && (a <= 0x1p40f);
    }
    __device__ __forceinline__ float div(float x) const {
        const float q0 = x * rs;
        const float a = fabsf(q0);
        if (fast && a >= 0x1p-80f && a <= 0x1p100f) return __builtin_fmaf(__builtin_fmaf(-q0, s, x), rs, q0);
        return x / s;
    }
};
// The same product without the range test, for backward passes (gradients are tolerance-checked sums; equal to
// the IEEE quotient for every finite operand pair away from the ends of the exponent range)
struct UncheckedDivisor {
    float s, rs;
    __device__ __forceinline__ explicit UncheckedDivisor(float s_) : s(s_), rs(1.0f / s_) {}
    __device__ __forceinline__ float div(float x) const {
        const float q0 = x * rs;
        return __builtin_fmaf(__builtin_fmaf(-q0, s, x), rs, q0);
    }
};
// x / s for consumers that only round the quotient to an integer level and clamp the level (the uniform grids), given
// rs = RN(1 / s) and 2^-40 <= |s| <= 2^40 (or a NaN s): the corrected product above wherever it equals the IEEE quotient
// (2^-80 <= |x * rs| <= 2^100); for smaller products both are far below 0.5 in magnitude and round to the same level;
// for larger ones, infinite and NaN x the plain product is returned (beyond every level, infinite or NaN like the
// quotient).  5 VALU operations, no branch, no reciprocal per element.
__device__ __forceinline__ float level_div(float x, float s, float rs) {
    const float q0 = x * rs;
    const float q = __builtin_fmaf(__builtin_fmaf(-q0, s, x), rs, q0);
    return fabsf(q0) <= 0x1p100f ? q : q0;
}
__device__ __forceinline__ bool level_div_ok(float s) {
    const float a = fabsf(s);
    return a >= 0x1p-40f && a <= 0x1p40f;
}
struct LevelDivisor {
    float s, rs;
    __device__ __forceinline__ explicit LevelDivisor(float s_) : s(s_), rs(1.0f / s_) {}
    __device__ __forceinline__ float div(float x) const { return level_div(x, s, rs); }
};
// fp16 quotient for a divisor that is finite and not zero, or NaN (half_div_ok; a calibrated scale clamped to
// [1e-5, 1e4] always is): the corrected product of Divisor<__half> below without its per-element test.  The test is a
// property of the row, and left inside the element loop the compiler keeps both forms behind a branch per element.
struct HalfDivisor {
    float s, rs;
    __device__ __forceinline__ explicit HalfDivisor(float s_) : s(s_), rs(1.0f / s_) {}
    __device__ __forceinline__ float div(float x) const {
        const float q0 = x * rs;
        const float q = __builtin_fmaf(__builtin_fmaf(-q0, s, x), rs, q0);
        // an infinite x (or product) has no finite residual: the quotient is the product itself; NaN stays NaN
        return __half2float(__float2half(fabsf(q0) < __builtin_inff() ? q : q0));
    }
};
__device__ __forceinline__ bool half_div_ok(float s) {
    const float a = fabsf(s);
    return !((a == 0.f) || (a == __builtin_inff()));
}
// divisor types of the arithmetic-bound row kernels (their scales are clamped)
template <typename T> struct BwdDivisorOf;
template <> struct BwdDivisorOf<float> { using type = UncheckedDivisor; };
template <> struct BwdDivisorOf<__half> { using type = HalfDivisor; };
template <typename T> struct FastDivisorOf;
template <> struct FastDivisorOf<float> { using type = CorrectedDivisor; };
template <> struct FastDivisorOf<__half> { using type = HalfDivisor; };
template <> struct Divisor<__half> {
    float s, rs;
    bool fast;
    __device__ __forceinline__ explicit Divisor(float s_) : s(s_), rs(1.0f / s_) {
        const float a = fabsf(s_);
        fast = (a > 0.f) && (a < __builtin_inff());
    }
    __device__ __forceinline__ float div(float x) const {
        float q;
        if (fast) {
            const float q0 = x * rs;
            const float e = __builtin_fmaf(-q0, s, x);
            q = fabsf(q0) < __builtin_inff() ? __builtin_fmaf(e, rs, q0) : q0;
        } else {
            q = x / s;
        }
        return __half2float(__float2half(q));
    }
};

// round-to-nearest-even with the straight-through form (r - q) + q of the reference: equal to r for
// finite q, NaN for +-inf and NaN.  (r - q) and the sum are exact in every dtype, so no rounding step.
__device__ __forceinline__ float rint_ste(float q) {
    return (fabsf(q) == __builtin_inff()) ? __builtin_nanf("") : rintf(q);
}

// The affine quantization level of one element: clamp(round(x/s) + zp, 0, qmax) and the dequantized
// value (level - zp) * s, with the per-op rounding of T.  `exact_int` tells that |zp| + qmax <= 2048 and
// qmax < 2048, i.e. every integer intermediate is exactly representable in fp16 (always true for fp32).
template <typename T, class D>
__device__ __forceinline__ float affine_level(float x, const D& dv, float zp, bool exact_int) {
    const float xi = rint_ste(dv.div(x)) + zp;
    return exact_int ? xi : rnd<T>(xi);
}
template <typename T>
__device__ __forceinline__ float affine_dequant(float level_clamped, float s, float zp, bool exact_int) {
    const float d = level_clamped - zp;
    return (exact_int ? d : rnd<T>(d)) * s;  // the product is rounded by the store conversion
}
template <typename T> __device__ __forceinline__ bool ints_exact(float zp, float qmax);
template <> __device__ __forceinline__ bool ints_exact<float>(float, float) { return true; }
template <> __device__ __forceinline__ bool ints_exact<__half>(float zp, float qmax) {
    return (fabsf(zp) + qmax <= 2048.f) && (qmax < 2048.f);
}

// Level of one element for a per-channel scale that is known to be finite and positive (the
// calibrated scale is clamped to [1e-5, 1e4]; a NaN scale propagates through either path).
template <typename T>
__device__ __forceinline__ float col_div(float x, float s, float rs);
// (the calibrated scale lies in [1e-5, 1e4] or is NaN, and rs is the IEEE 1 / s: level_div applies)
template <> __device__ __forceinline__ float col_div<float>(float x, float s, float rs) { return level_div(x, s, rs); }
template <> __device__ __forceinline__ float col_div<__half>(float x, float s, float rs) {
    const float q0 = x * rs;
    const float e = __builtin_fmaf(-q0, s, x);   // exact residual: see Divisor<__half>
    return __half2float(__float2half(fabsf(q0) < __builtin_inff() ? __builtin_fmaf(e, rs, q0) : q0));
}
template <typename T>
__device__ __forceinline__ float col_level(float x, float s, float rs, float z, bool exact_int) {
    const float xi = rint_ste(col_div<T>(x, s, rs)) + z;
    return exact_int ? xi : rnd<T>(xi);
}

// Terms of the static quantizer's backward for one element: q = x/s, m = "level inside [0, qmax]",
// xd = clamp(level) - z.
template <typename T>
__device__ __forceinline__ void fq_terms(float xv, float s, float rs, float z, float qmax, float& q, bool& m, float& xd) {
    q = col_div<T>(xv, s, rs);
    const float xi = rnd<T>(rint_ste(q) + z);
    m = (xi >= 0.f) && (xi <= qmax);
    xd = rnd<T>(clampf(xi, 0.f, qmax) - z);
}

// ---------------------------------------------------------------------------------------------
// team reductions.  A "team" is TEAM consecutive threads that own one quantization row.
// TEAM <= 64: shuffle only (team lies inside one wave).  TEAM > 64: TEAM == blockDim.x, LDS exchange.
// ---------------------------------------------------------------------------------------------
// lds: >= 16*K floats of shared memory, only touched when TEAM > 64
template <int TEAM, int K, class F>
__device__ __forceinline__ void team_reduce(float (&v)[K], F f, float* lds) {
    constexpr int W = TEAM < 64 ? TEAM : 64;
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = f(k, v[k], __shfl_xor(v[k], o));
    }
    if constexpr (TEAM > 64) {
        constexpr int NW = TEAM / 64;
        const int wid = threadIdx.x >> 6;
        __syncthreads();  // protects lds against the previous reduction's readers
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) lds[k * 16 + wid] = v[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float r = lds[k * 16];
#pragma unroll
            for (int w = 1; w < NW; ++w) r = f(k, r, lds[k * 16 + w]);
            v[k] = r;
        }
    }
}

// element 0 is a running minimum, element 1 a running maximum (both NaN-propagating)
struct OpMinMax {
    __device__ float operator()(int k, float a, float b) const { return k == 0 ? nmin(a, b) : nmax(a, b); }
};
struct OpAdd {
    __device__ float operator()(int, float a, float b) const { return a + b; }
};

template <int TEAM, int K>
__device__ __forceinline__ void team_sum(float (&v)[K], float* lds) { team_reduce<TEAM, K>(v, OpAdd(), lds); }

// ---------------------------------------------------------------------------------------------
// Row geometry: how quantization rows map onto a row-major [O, I] matrix.
//   G == 1              : rows are matrix rows of length I (g == I)
//   G  > 1              : every matrix row is cut into G groups of g elements; the last group of a
//                         row may be short (I - (G-1)*g elements) and is zero-padded to g for the
//                         statistics only (padding never reaches the output).
// ---------------------------------------------------------------------------------------------
struct RowGeom {
    long long n_rows;  // O * G
    int I;             // matrix columns
    int g;             // nominal row (group) length
    int G;             // groups per matrix row
    int stream_out;    // the output (quantized matrix / input gradient) is written non-temporally: stream_out_of(its bytes)
};

__device__ __forceinline__ void row_span(const RowGeom& geo, long long row, long long& base, int& len) {
    if (geo.G == 1) {
        base = row * (long long)geo.I;
        len = geo.I;
    } else {
        const long long r = row / geo.G;
        const int j = (int)(row - r * geo.G);
        base = r * (long long)geo.I + (long long)j * geo.g;
        const int rest = geo.I - j * geo.g;
        len = rest < geo.g ? rest : geo.g;
    }
}

// A row held in registers by its team: thread t owns packs t, t+TEAM, t+2*TEAM, ...
template <typename T, int VEC, int TEAM, int NV>
struct RowTile {
    Pack<T, VEC> v[NV];
    int npack;  // packs in this row
    int t;      // lane inside the team

    __device__ __forceinline__ void load(const T* base, int len, int lane) {
        t = lane;
        npack = len / VEC;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + t;
            if (j < npack) v[k] = ld_pack<T, VEC>(base + (long long)j * VEC);
        }
    }
    __device__ __forceinline__ void load_nt(const T* base, int len, int lane) {
        t = lane;
        npack = len / VEC;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + t;
            if (j < npack) v[k] = ld_pack_nt<T, VEC>(base + (long long)j * VEC);
        }
    }
    template <class F> __device__ __forceinline__ void for_each(F f) const {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + t;
            if (j < npack) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) f(k, e, to_f<T>(v[k].e[e]));
            }
        }
    }
    __device__ __forceinline__ bool has(int k) const { return k * TEAM + t < npack; }
    __device__ __forceinline__ long long offset(int k) const { return (long long)(k * TEAM + t) * VEC; }
};

// Row min/max over the team (NaN-propagating), including the implicit zero padding.
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void row_minmax(const RowTile<T, VEC, TEAM, NV>& x, bool padded, float* lds,
                                           float& rmin, float& rmax) {
    // per element: one minimum, one maximum (the hardware's, which skip NaN; the compiler pairs them into three-operand
    // forms) and an unordered-compare that remembers a NaN, instead of two compare-and-select chains
    float mn = __builtin_inff(), mx = -__builtin_inff();
    bool any_nan = false;
    x.for_each([&](int, int, float a) {
        mn = __builtin_fminf(mn, a);
        mx = __builtin_fmaxf(mx, a);
        any_nan = any_nan || (a != a);
    });
    if (padded) { mn = __builtin_fminf(mn, 0.f); mx = __builtin_fmaxf(mx, 0.f); }
    if (any_nan) mn = mx = __builtin_nanf("");
    float mm[2] = {mn, mx};
    team_reduce<TEAM, 2>(mm, OpMinMax(), lds);
    rmin = mm[0];
    rmax = mm[1];
}

// The same, also telling the lane whether one of ITS elements can equal a row extreme (its own extremes reach the
// row's): kernels that count and correct the elements at the extremes can skip both for every other lane.
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void row_minmax(const RowTile<T, VEC, TEAM, NV>& x, bool padded, float* lds,
                                           float& rmin, float& rmax, bool& lane_at_extreme) {
    float mn = __builtin_inff(), mx = -__builtin_inff();
    bool any_nan = false;
    x.for_each([&](int, int, float a) {
        mn = __builtin_fminf(mn, a);
        mx = __builtin_fmaxf(mx, a);
        any_nan = any_nan || (a != a);
    });
    const float lmn = mn, lmx = mx;  // of the lane's own elements (padding is not an element)
    if (padded) { mn = __builtin_fminf(mn, 0.f); mx = __builtin_fmaxf(mx, 0.f); }
    if (any_nan) mn = mx = __builtin_nanf("");
    float mm[2] = {mn, mx};
    team_reduce<TEAM, 2>(mm, OpMinMax(), lds);
    rmin = mm[0];
    rmax = mm[1];
    lane_at_extreme = (lmn == rmin) || (lmx == rmax);  // never true in a NaN row, where no element equals the extremes
}

// ---------------------------------------------------------------------------------------------
// Learnable weight clipping shared by all row-wise quantizers:
//   xmax = sigmoid(up) * rowmax, xmin = sigmoid(low) * rowmin   (reference quantizer.py:259-261)
// The library receives the ACTIVATED factors su = sigmoid(up), sl = sigmoid(low): the caller evaluates the sigmoid
// with the same routine the reference uses (a device math library's exp differs from it in the last bit, and
// one bit of the clip factor moves the scale of the whole row), and chains sigmoid' onto the returned gradients.
// ---------------------------------------------------------------------------------------------
struct ClipCtx {
    float su, sl;      // activated clip factors; 1 when LWC is off
    float xmax, xmin;  // clipped bounds
};

template <typename T>
__device__ __forceinline__ ClipCtx lwc_clip(const T* up, const T* low, long long row, float rmin, float rmax) {
    ClipCtx c;
    c.su = 1.f;
    c.sl = 1.f;
    c.xmax = rmax;
    c.xmin = rmin;
    if (up) {
        c.su = to_f<T>(up[row]);
        c.sl = to_f<T>(low[row]);
        c.xmax = rnd<T>(c.su * rmax);
        c.xmin = rnd<T>(c.sl * rmin);
    }
    return c;
}

// gradient of max(|xmax|, |xmin|): all of it to the larger magnitude, half each on a tie, times sign
__device__ __forceinline__ void route_absmax(float d_amax, const ClipCtx& c, float& d_xmax, float& d_xmin) {
    const float A = fabsf(c.xmax), B = fabsf(c.xmin);
    const float da = (A > B) ? d_amax : (A == B ? 0.5f * d_amax : 0.f);
    const float db = (B > A) ? d_amax : (A == B ? 0.5f * d_amax : 0.f);
    d_xmax = da * sgnf(c.xmax);
    d_xmin = db * sgnf(c.xmin);
}

// gradients of the activated clip factors from the gradients of the clipped bounds (chain through the multiply)
template <typename T>
__device__ __forceinline__ void write_lwc_grads(T* d_up, T* d_low, long long row, float rmin, float rmax, float d_xmax,
                                                float d_xmin) {
    d_up[row] = from_f<T>(d_xmax * rmax);
    d_low[row] = from_f<T>(d_xmin * rmin);
}

// Several matrices of one row width in ONE launch (the weight matrices of a block are quantized back to
// back and are independent): a 67 MB matrix alone streams at ~5.0 TB/s because the launch's ramp-up and
// tail are a sizeable part of its 25 us; 38400 rows in one launch reach 5.7 TB/s.
constexpr int kMaxBatch = 16;
template <typename T>
struct UniBatch {
    const T* x[kMaxBatch];
    const T* up[kMaxBatch];
    const T* low[kMaxBatch];
    const T* gout[kMaxBatch];   // backward only
    T* out[kMaxBatch];          // forward: dequantized output; backward: dx (or nullptr)
    T* a1[kMaxBatch];           // forward: scale; backward: d_up
    T* a2[kMaxBatch];           // forward: rounded zero point; backward: d_low
    long long row_start[kMaxBatch + 1];
    int n;
};

template <typename T>
__device__ __forceinline__ int batch_locate(const UniBatch<T>& b, long long row, long long& local) {
    int m = 0;
    while (m + 1 < b.n && row >= b.row_start[m + 1]) ++m;
    local = row - b.row_start[m];
    return m;
}

// ---------------------------------------------------------------------------------------------
// Host-side launch helper: pick (TEAM, NV) from the number of 16-byte packs in the longest row.
// ---------------------------------------------------------------------------------------------
// 11008-wide fp32 rows (2752 packs) take a 6-pack tile on 512 lanes: the 8-pack tile's registers for packs that hold
// nothing cost the forward kernels 2-3 % (and the LET forward, which keeps three vectors per pack, a third of its time).
// Likewise 5120-wide rows (Llama-2-13B: 1280 packs = 256 x 5 exactly), 13824-wide (3456 packs: 512 x 7) and
// 28672-wide (Llama-2-70B: 7168 packs = 1024 x 7 exactly): a pack a lane never loads still costs the backward kernels
// two register quads, and with them a wave per SIMD.
// Teams that are not powers of two (384 x 8 for 11008-wide rows, 320 x 4, 448 x 8, 896 x 8) remove the idle
// lanes of the next larger tile but were measured slower in the backward kernels (6 or 7 waves per block
// spread unevenly over the 4 SIMDs), so the table stays with power-of-two teams.
// ESZ == 2 selects tiles with half the packs per lane on twice the lanes beyond 512 packs: a 2-byte pack carries 8
// elements, and 32 to 64 elements per lane cost the register-hungry fp16 kernels (the exponent grids' backward: 190 to
// 270 registers) their occupancy.  The light kernels (uniform grids) are faster on the ESZ == 4 table in fp16 as well
// (1024-thread blocks and their block-wide reductions cost them 5 to 25 %), so every caller names the table it wants.
#define MIQ_ROW_DISPATCH_E(ESZ, NPACK, LAUNCH)                           \
    do {                                                                 \
        if ((NPACK) <= 16) { LAUNCH(16, 1); }                            \
        else if ((NPACK) <= 64) { LAUNCH(16, 4); }                       \
        else if ((NPACK) <= 256) { LAUNCH(64, 4); }                      \
        else if ((NPACK) <= 512) { LAUNCH(256, 2); }                     \
        else if constexpr ((ESZ) == 2) {                                 \
            if ((NPACK) <= 1024) { LAUNCH(512, 2); }                     \
            else if ((NPACK) <= 2048) { LAUNCH(1024, 2); }               \
            else if ((NPACK) <= 4096) { LAUNCH(1024, 4); }               \
            else { LAUNCH(1024, 8); }                                    \
        } else {                                                         \
            if ((NPACK) <= 1024) { LAUNCH(256, 4); }                     \
            else if ((NPACK) <= 1280) { LAUNCH(256, 5); }                \
            else if ((NPACK) <= 2048) { LAUNCH(256, 8); }                \
            else if ((NPACK) <= 3072) { LAUNCH(512, 6); }                \
            else if ((NPACK) <= 3584) { LAUNCH(512, 7); }                \
            else if ((NPACK) <= 4096) { LAUNCH(512, 8); }                \
            else if ((NPACK) <= 7168) { LAUNCH(1024, 7); }               \
            else { LAUNCH(1024, 8); }                                    \
        }                                                                \
    } while (0)
#define MIQ_ROW_DISPATCH(NPACK, LAUNCH) MIQ_ROW_DISPATCH_E(4, NPACK, LAUNCH)

// Launch KERN<T, VEC, TEAM, NV> over all rows of GEO with the (TEAM, NV) that fits NPACK packs per row.
#define MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, TEAM, NV, GEO, STREAM, ...)                                        \
    {                                                                                                        \
        constexpr int B_ = miq::block_of<TEAM>();                                                            \
        const long long nb_ = ((GEO).n_rows + (B_ / TEAM) - 1) / (B_ / TEAM);                                \
        hipLaunchKernelGGL((KERN<T, VECN, TEAM, NV>), dim3((unsigned)nb_), dim3(B_), 0, STREAM, __VA_ARGS__); \
    }
#define MIQ_ROW_LAUNCH_VEC(ESZ, KERN, T, VECN, NPACK, GEO, STREAM, ...)                                         \
    do {                                                                                                   \
        if ((NPACK) <= 16) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 16, 1, GEO, STREAM, __VA_ARGS__)              \
        else if ((NPACK) <= 64) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 16, 4, GEO, STREAM, __VA_ARGS__)         \
        else if ((NPACK) <= 256) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 64, 4, GEO, STREAM, __VA_ARGS__)        \
        else if ((NPACK) <= 512) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 256, 2, GEO, STREAM, __VA_ARGS__)       \
        else if constexpr ((ESZ) == 2) {                                                                     \
            if ((NPACK) <= 1024) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 512, 2, GEO, STREAM, __VA_ARGS__)       \
            else if ((NPACK) <= 2048) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 1024, 2, GEO, STREAM, __VA_ARGS__) \
            else if ((NPACK) <= 4096) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 1024, 4, GEO, STREAM, __VA_ARGS__) \
            else MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 1024, 8, GEO, STREAM, __VA_ARGS__)                      \
        } else {                                                                                           \
            if ((NPACK) <= 1024) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 256, 4, GEO, STREAM, __VA_ARGS__)       \
            else if ((NPACK) <= 1280) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 256, 5, GEO, STREAM, __VA_ARGS__)  \
            else if ((NPACK) <= 2048) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 256, 8, GEO, STREAM, __VA_ARGS__)  \
            else if ((NPACK) <= 3072) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 512, 6, GEO, STREAM, __VA_ARGS__)  \
            else if ((NPACK) <= 3584) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 512, 7, GEO, STREAM, __VA_ARGS__)  \
            else if ((NPACK) <= 4096) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 512, 8, GEO, STREAM, __VA_ARGS__)  \
            else if ((NPACK) <= 7168) MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 1024, 7, GEO, STREAM, __VA_ARGS__) \
            else MIQ_ROW_LAUNCH_ONE(KERN, T, VECN, 1024, 8, GEO, STREAM, __VA_ARGS__)                      \
        }                                                                                                  \
    } while (0)
#define MIQ_ROW_LAUNCH_E(ESZ, KERN, T, VEC_OK, GEO, STREAM, ...)                                     \
    do {                                                                                             \
        if (VEC_OK) {                                                                                \
            MIQ_ROW_LAUNCH_VEC(ESZ, KERN, T, miq::VecOf<T>::N, (GEO).g / miq::VecOf<T>::N, GEO, STREAM, __VA_ARGS__); \
        } else {                                                                                     \
            MIQ_ROW_LAUNCH_VEC(ESZ, KERN, T, 1, (GEO).g, GEO, STREAM, __VA_ARGS__);                  \
        }                                                                                            \
    } while (0)
// the table of 4-byte elements for every element size (the light kernels are fastest on it in fp16 too)
#define MIQ_ROW_LAUNCH(KERN, T, VEC_OK, GEO, STREAM, ...) MIQ_ROW_LAUNCH_E(4, KERN, T, VEC_OK, GEO, STREAM, __VA_ARGS__)
// the table of the element's own size: for kernels that hold many registers per element (the exponent grids' backward)
#define MIQ_ROW_LAUNCH_SIZED(KERN, T, VEC_OK, GEO, STREAM, ...) MIQ_ROW_LAUNCH_E(sizeof(T), KERN, T, VEC_OK, GEO, STREAM, __VA_ARGS__)

constexpr int kRowMaxPacks = 8192;

template <int TEAM> constexpr int block_of() { return TEAM > 256 ? TEAM : 256; }

// blocks per cluster of the per-cluster parameter kernels: enough 256-channel slices for an average cluster, at most 16
static inline unsigned cluster_slices(long long C, unsigned n_blocks) {
    const long long per = (C / (n_blocks > 0 ? n_blocks : 1) + 255) / 256;
    return (unsigned)(per < 1 ? 1 : (per > 16 ? 16 : per));
}

static inline bool is_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// fills GEO for an [O, I] matrix with optional grouping; returns false when the matrix is too large
static inline bool make_geom(long long O, long long I, long long group, RowGeom& geo) {
    if (I >= (1ll << 31)) return false;
    geo.I = (int)I;
    geo.g = group > 0 ? (int)group : (int)I;
    geo.G = (int)((I + geo.g - 1) / geo.g);
    geo.n_rows = O * geo.G;
    geo.stream_out = 0;  // the caller knows the element size
    return geo.n_rows < (1ll << 33);
}

}  // namespace miq

// error reporting shared by all translation units (defined in capi.hip)
extern "C" void miq_set_error(const char* fmt, ...);
extern "C" int miq_colstat_chunks(int64_t R, int64_t C, int dtype);

#define MIQ_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            miq_set_error(__VA_ARGS__);          \
            return -22; /* -EINVAL */            \
        }                                        \
    } while (0)

#define MIQ_CHECK_LAUNCH(name)                                                          \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            miq_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));        \
            return -5; /* -EIO */                                                       \
        }                                                                               \
    } while (0)
