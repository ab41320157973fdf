// Learnable per-row bit-width ("FracBits" style) row-wise fake quantization: every row is quantized
// on the floor(n) and ceil(n) bit grids and the two results are blended with weights (1-frac, frac).
// Forward and backward of MixAffineQuantizer.forward / fake_quant
// (reference quantize/quantizer_mix.py:143-191,118-140); fp32 only, like the reference's use of it.
//
// Backward (autograd through floor_ste / ceil_ste / round_ste, 2**b, the tensor-valued clamp bound
// Q_b = 2^b - 1 and the scale), per grid b with g_b = g * weight_b:
//   dS_b   = sum g_b*(xd_b - m_b*x/s_b)                    (zero when the scale clamp is active)
//   dQ_b   = sum g_b*s_b*[level above Q_b]  +  dS_b * d(s_b)/d(Q_b)
//   d_n   += dQ_b * ln2 * 2^b      (asym: s_b = range/Q_b;  sym: s_b = absmax/H_b, H_b = 2^(b-1)-1,
//                                   d_n += dS_b * (-absmax/H_b^2) * ln2 * 2^(b-1) and only the clamp
//                                   term goes through Q_b)
//   d(blend weight)/d n cancels under the straight-through floor.
#include "miq_common.h"

namespace miq {

constexpr float kLn2m = 0.693147180559945309f;

struct MixArgs {
    const float* x;
    const float* up;
    const float* low;
    const float* n_bits;  // [rows] already clamped to [2, 8] by the caller
    int symmetric;
    int hard;
    float threshold;
};

struct Grid {
    float b, Q, H, s, s_raw, z, w, rs;  // bits, 2^b-1, 2^(b-1)-1, scale, unclamped scale, zero point, blend weight, 1/scale
};

struct MixCtx {
    ClipCtx clip;
    Grid lo, hi;
    float rng, amax;
};

__device__ __forceinline__ void make_grid(Grid& gr, float b, float w, const MixCtx& c, int symmetric) {
    gr.b = b;
    gr.w = w;
    gr.Q = exp2f(b) - 1.f;
    gr.H = exp2f(b - 1.f) - 1.f;
    if (symmetric) {
        gr.s_raw = c.amax / gr.H;
        gr.s = gr.s_raw;
        gr.z = rintf(clampf(gr.H, -kClipHi, kClipHi));
    } else {
        gr.s_raw = c.rng / gr.Q;
        gr.s = clampf(gr.s_raw, kClipLo, kClipHi);
        gr.z = rintf(clampf(-c.clip.xmin / gr.s, -kClipHi, kClipHi));
    }
    gr.rs = 1.0f / gr.s;
}

__device__ __forceinline__ MixCtx mix_ctx(const MixArgs& a, long long row, float rmin, float rmax) {
    MixCtx c;
    c.clip = lwc_clip<float>(a.up, a.low, row, rmin, rmax);
    float n = a.n_bits[row];
    if (a.hard) {
        const float t = n + a.threshold;
        n = (rintf(t) - t) + t;
    }
    const float nl = (floorf(n) - n) + n;
    const float nh = (ceilf(n) - n) + n;
    const float hp = n - nl;
    const float lp = 1.0f - hp;
    c.rng = c.clip.xmax - c.clip.xmin;
    c.amax = nmax(fabsf(c.clip.xmax), fabsf(c.clip.xmin));
    make_grid(c.lo, nl, lp, c, a.symmetric);
    // a whole number of bits (every row of the hard phase): floor and ceiling are the same grid, so the second one is
    // the first with weight 0 (field by field: assigning the struct under the branch puts it into scratch memory)
    const bool whole = hp == 0.f;
    if (!whole) make_grid(c.hi, nh, hp, c, a.symmetric);
    c.hi.b = whole ? c.lo.b : c.hi.b;
    c.hi.Q = whole ? c.lo.Q : c.hi.Q;
    c.hi.H = whole ? c.lo.H : c.hi.H;
    c.hi.s = whole ? c.lo.s : c.hi.s;
    c.hi.s_raw = whole ? c.lo.s_raw : c.hi.s_raw;
    c.hi.z = whole ? c.lo.z : c.hi.z;
    c.hi.rs = whole ? c.lo.rs : c.hi.rs;
    c.hi.w = hp;
    return c;
}

// A row whose every level is finite: finite extremes, both scales inside level_div's range, finite zero points, and
// |x / s| far below the range where the residual of the corrected product could overflow.  There the level needs
// neither level_div's guard for huge products nor NaN-preserving clamps: a median-of-three is the clamp.
__device__ __forceinline__ bool mix_plain_row(const MixCtx& c, float rmin, float rmax) {
    const float a = __builtin_fmaxf(fabsf(rmin), fabsf(rmax));
    const float bound = a * __builtin_fmaxf(c.lo.rs, c.hi.rs);
    return (rmin == rmin) && (rmax == rmax) && level_div_ok(c.lo.s) && level_div_ok(c.hi.s) && (bound <= 0x1p60f) &&
           (fabsf(c.lo.z) <= kClipHi) && (fabsf(c.hi.z) <= kClipHi) && (c.lo.w == c.lo.w) && (c.hi.w == c.hi.w);
}
__device__ __forceinline__ float plain_quotient(float xv, const Grid& gr) {
    const float q0 = xv * gr.rs;
    return __builtin_fmaf(__builtin_fmaf(-q0, gr.s, xv), gr.rs, q0);
}
__device__ __forceinline__ float plain_fq(float xv, const Grid& gr) {
    // a finite quotient: (rint(q) - q) + q, the straight-through form of the reference, is rint(q) itself (the
    // difference is exact, and so is adding q back)
    const float q = plain_quotient(xv, gr);
    const float xi = rintf(q) + gr.z;
    return (__builtin_amdgcn_fmed3f(xi, 0.f, gr.Q) - gr.z) * gr.s;
}

// LEVEL: the quotient through level_div (both scales of the row inside its range: a property of the row, tested once
// there); otherwise the IEEE division.  The level is only rounded and clamped, so the two forms give the same bits.
template <bool LEVEL>
__device__ __forceinline__ float grid_level(float xv, const Grid& gr) {
    const float q = LEVEL ? level_div(xv, gr.s, gr.rs) : xv / gr.s;
    return ((rintf(q) - q) + q) + gr.z;
}
template <bool LEVEL>
__device__ __forceinline__ float grid_fq(float xv, const Grid& gr) {
    return (clampf(grid_level<LEVEL>(xv, gr), 0.f, gr.Q) - gr.z) * gr.s;
}

// the work of one team on one quantization row
template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void mix_fwd_row(const MixArgs& a, const RowGeom& geo, long long row, int lane, float* lds,
                                            float* __restrict__ out) {
    long long base;
    int len;
    row_span(geo, row, base, len);
    RowTile<float, VEC, TEAM, NV> x;
    x.load_nt(a.x + base, len, lane);
    float rmin, rmax;
    row_minmax(x, len < geo.g, lds, rmin, rmax);
    const MixCtx c = mix_ctx(a, row, rmin, rmax);
    // instantiated per form of the level: a test inside the element loop would keep every form there
    auto quantize_row = [&](auto form_tag) {
        // 3: plain row on ONE grid (an integer bit-width, as in the hard phase: weights 1 and 0, so 1 * fq_lo + 0 * fq_hi
        // is fq_lo itself), 2: plain row, 1: level_div, 0: IEEE division
        constexpr int FORM = decltype(form_tag)::value;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<float, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = x.v[k].e[e];
                if constexpr (FORM == 3) o.e[e] = plain_fq(xv, c.lo);
                else if constexpr (FORM == 2) o.e[e] = c.lo.w * plain_fq(xv, c.lo) + c.hi.w * plain_fq(xv, c.hi);
                else o.e[e] = c.lo.w * grid_fq<FORM == 1>(xv, c.lo) + c.hi.w * grid_fq<FORM == 1>(xv, c.hi);
            }
            st_out<float, VEC>(out + base + x.offset(k), o, geo.stream_out);
        }
    };
    if (mix_plain_row(c, rmin, rmax)) {
        if (c.hi.w == 0.f) quantize_row(std::integral_constant<int, 3>{});
        else quantize_row(std::integral_constant<int, 2>{});
    }
    else if (level_div_ok(c.lo.s) && level_div_ok(c.hi.s)) quantize_row(std::integral_constant<int, 1>{});
    else quantize_row(std::integral_constant<int, 0>{});
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_mix_fwd(MixArgs a, RowGeom geo, float* __restrict__ out) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    mix_fwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, out);
}

// per-matrix bit-width vectors of a batched launch (the other tables are the uniform quantizer's)
struct MixBatchBits {
    const float* n_bits[kMaxBatch];
    float* d_nbits[kMaxBatch];  // backward only; nullptr: not wanted
};

// several matrices of one row width in one launch
template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_mix_fwd_batch(UniBatch<float> b, MixBatchBits nb, MixArgs a, RowGeom geo) {
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
    a.n_bits = nb.n_bits[m];
    mix_fwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.out[m]);
}

template <typename T, int VEC, int TEAM, int NV>
__device__ __forceinline__ void mix_bwd_row(const MixArgs& a, const RowGeom& geo, long long row, int lane, float* lds,
                                            const float* __restrict__ gout, float* __restrict__ dx, float* __restrict__ d_up,
                                            float* __restrict__ d_low, float* __restrict__ d_nbits) {
    long long base;
    int len;
    row_span(geo, row, base, len);
    RowTile<float, VEC, TEAM, NV> x, g;
    x.load_nt(a.x + base, len, lane);
    g.load_nt(gout + base, len, lane);
    const bool padded = len < geo.g;
    float rmin, rmax;
    bool at_extreme;  // one of this lane's elements equals a row extreme (rarely true in more than two lanes of a row)
    row_minmax(x, padded, lds, rmin, rmax, at_extreme);
    const MixCtx c = mix_ctx(a, row, rmin, rmax);
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // dS_lo, dS_hi, over_lo, over_hi, #max, #min
    // An integer bit-width (every row of the hard phase) puts weight 1 on the lower grid and 0 on the upper one, whose
    // terms vanish.  Such a row, when all of its levels are finite (mix_plain_row: a property of the row, tested once),
    // takes a short form on ONE grid: rint(q) is its own straight-through form, no level is NaN, (g * s) / s is g to
    // rounding, and the clamp term collects g alone (times s after the loop).
    if (c.hi.w == 0.f && mix_plain_row(c, rmin, rmax)) {
        const Grid& gr = c.lo;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = x.v[k].e[e], gv = g.v[k].e[e];
                const float q0 = xv * gr.rs;
                const float q = __builtin_fmaf(__builtin_fmaf(-q0, gr.s, xv), gr.rs, q0);
                const float xi = rintf(q) + gr.z;
                const float xc = __builtin_amdgcn_fmed3f(xi, 0.f, gr.Q);
                const bool m = xc == xi;
                acc[0] += gv * ((xc - gr.z) - (m ? q : 0.f));
                acc[2] += (xi > gr.Q) ? gv : 0.f;
                g.v[k].e[e] = m ? gv : 0.f;
            }
        }
        acc[2] *= gr.s;
    } else {
        // returns the straight-through part of dx through this grid (what autograd forms as ((g*w)*s)/s)
        auto terms = [&](float xv, float gv, const Grid& gr, float& dS, float& over) {
            const float gb = gv * gr.w;
            // backward: the corrected reciprocal product without the range test of Divisor<float> (equal to the IEEE
            // quotient for every weight a forward pass can quantize; gradients are tolerance-checked sums)
            const float q0 = xv * gr.rs;
            const float q = __builtin_fmaf(__builtin_fmaf(-q0, gr.s, xv), gr.rs, q0);
            const float xi = ((rintf(q) - q) + q) + gr.z;
            // the clamp as a median of three; the level lies inside the grid exactly when the clamp left it alone; a NaN
            // level (the median drops it) reaches the sum through xi - xi, which is 0 for every other level
            const float xc = __builtin_amdgcn_fmed3f(xi, 0.f, gr.Q);
            const bool m = xc == xi;
            const float xd = xc - gr.z;
            dS += gb * (xd - (m ? q : xi - xi));
            if (xi > gr.Q) over += gb * gr.s;
            return m ? (gb * gr.s) * gr.rs : 0.f;
        };
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xv = x.v[k].e[e], gv = g.v[k].e[e];
                float d = 0.f;
                d += terms(xv, gv, c.lo, acc[0], acc[2]);
                d += terms(xv, gv, c.hi, acc[1], acc[3]);
                g.v[k].e[e] = d;  // g is not read again: its registers carry the direct gradient to the dx pass
            }
        }
    }
    if (at_extreme) {  // the elements at the extremes are counted (and corrected, below) only by the lanes that hold them
        x.for_each([&](int, int, float xv) {
            acc[4] += (xv == rmax) ? 1.f : 0.f;
            acc[5] += (xv == rmin) ? 1.f : 0.f;
        });
    }
    team_sum<TEAM, 6>(acc, lds);
    if (padded) {
        const float npad = (float)(geo.g - len);
        if (rmax == 0.f) acc[4] += npad;
        if (rmin == 0.f) acc[5] += npad;
    }
    float d_xmax = 0.f, d_xmin = 0.f;
    if (lane == 0 || at_extreme) {  // the other lanes (most waves of a team as a whole) need none of the row's gradients
        float d_n = 0.f;
        auto live = [](const Grid& gr) { return gr.s_raw >= kClipLo && gr.s_raw <= kClipHi; };
        if (a.symmetric) {
            const float dS_l = acc[0], dS_h = acc[1];
            d_n += (dS_l * (-c.amax / (c.lo.H * c.lo.H))) * (kLn2m * exp2f(c.lo.b - 1.f));
            d_n += (dS_h * (-c.amax / (c.hi.H * c.hi.H))) * (kLn2m * exp2f(c.hi.b - 1.f));
            d_n += acc[2] * (kLn2m * exp2f(c.lo.b)) + acc[3] * (kLn2m * exp2f(c.hi.b));
            route_absmax(dS_l / c.lo.H + dS_h / c.hi.H, c.clip, d_xmax, d_xmin);
        } else {
            const float dS_l = live(c.lo) ? acc[0] : 0.f, dS_h = live(c.hi) ? acc[1] : 0.f;
            const float dQ_l = acc[2] + dS_l * (-c.rng / (c.lo.Q * c.lo.Q));
            const float dQ_h = acc[3] + dS_h * (-c.rng / (c.hi.Q * c.hi.Q));
            d_n = dQ_l * (kLn2m * exp2f(c.lo.b)) + dQ_h * (kLn2m * exp2f(c.hi.b));
            const float d_rng = dS_l / c.lo.Q + dS_h / c.hi.Q;
            d_xmax = d_rng;
            d_xmin = -d_rng;
        }
        if (lane == 0) {
            if (d_nbits) d_nbits[row] = d_n;
            if (a.up) write_lwc_grads<float>(d_up, d_low, row, rmin, rmax, d_xmax, d_xmin);
        }
    }
    if (dx) {
        if (at_extreme) {
            const float share_max = (d_xmax * c.clip.su) / acc[4];
            const float share_min = (d_xmin * c.clip.sl) / acc[5];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!x.has(k)) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float xv = x.v[k].e[e];
                    float d = g.v[k].e[e];
                    if (xv == rmax) d += share_max;
                    if (xv == rmin) d += share_min;
                    g.v[k].e[e] = d;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            st_out<float, VEC>(dx + base + x.offset(k), g.v[k], geo.stream_out);
        }
    }
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>(), (TEAM == 512 && (NV == 6 || NV == 7)) ? 4 : 1)  // the 6- and 7-pack tiles stay inside 128 registers
k_mix_bwd(MixArgs a, RowGeom geo, const float* __restrict__ gout, float* __restrict__ dx, float* __restrict__ d_up,
          float* __restrict__ d_low, float* __restrict__ d_nbits) {
    __shared__ float lds[96];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= geo.n_rows) return;
    mix_bwd_row<T, VEC, TEAM, NV>(a, geo, row, lane, lds, gout, dx, d_up, d_low, d_nbits);
}

template <typename T, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>(), (TEAM == 512 && (NV == 6 || NV == 7)) ? 4 : 1)  // the 6- and 7-pack tiles stay inside 128 registers
k_mix_bwd_batch(UniBatch<float> b, MixBatchBits nb, MixArgs a, RowGeom geo) {
    __shared__ float lds[96];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= b.row_start[b.n]) return;
    long long local;
    const int m = batch_locate(b, row, local);
    a.x = b.x[m];
    a.up = b.up[m];
    a.low = b.low[m];
    a.n_bits = nb.n_bits[m];
    mix_bwd_row<T, VEC, TEAM, NV>(a, geo, local, lane, lds, b.gout[m], b.out[m], b.a1[m], b.a2[m], nb.d_nbits[m]);
}

static int launch_mix(bool backward, const void* gout, const void* x, const void* up, const void* low, const float* n_bits,
                      void* out_or_dx, void* d_up, void* d_low, void* d_nbits, long long O, long long I, long long group,
                      int symmetric, int hard, float threshold, int flags, hipStream_t stream) {
    const char* name = backward ? "miq_rowquant_mix_bwd" : "miq_rowquant_mix_fwd";
    if (O == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(x && n_bits && O > 0 && I > 0, "%s: null input pointer or bad shape", name);
    MIQ_CHECK_ARG((up == nullptr) == (low == nullptr), "%s: up/low must both be given or both be null", name);
    MIQ_CHECK_ARG(backward ? (gout != nullptr) : (out_or_dx != nullptr), "%s: null pointer", name);
    MIQ_CHECK_ARG(!backward || !up || (d_up && d_low), "%s: d_up/d_low required when up/low are given", name);
    RowGeom geo;
    MIQ_CHECK_ARG(make_geom(O, I, group, geo), "%s: matrix too large", name);
    geo.stream_out = stream_out_of(O * I * (long long)sizeof(float));
    const bool vec_ok = (I % 4 == 0) && (geo.g % 4 == 0) && is_aligned16(x) && (!out_or_dx || is_aligned16(out_or_dx)) &&
                        (!backward || is_aligned16(gout));
    MIQ_CHECK_ARG((vec_ok ? geo.g / 4 : geo.g) <= kRowMaxPacks, "%s: row of %d elements exceeds the register tile", name, geo.g);
    MixArgs a;
    a.x = (const float*)x;
    a.up = (const float*)up;
    a.low = (const float*)low;
    a.n_bits = n_bits;
    a.symmetric = symmetric;
    a.hard = hard;
    a.threshold = threshold;
    if (!backward) MIQ_ROW_LAUNCH(k_mix_fwd, float, vec_ok, geo, stream, a, geo, (float*)out_or_dx);
    else MIQ_ROW_LAUNCH(k_mix_bwd, float, vec_ok, geo, stream, a, geo, (const float*)gout, (float*)out_or_dx, (float*)d_up,
                        (float*)d_low, (float*)d_nbits);
    MIQ_CHECK_LAUNCH(name);
    return 0;
}

// several matrices of one row width (whole rows, no groups): one launch each way
static int launch_mix_batch(bool backward, int n, const void* const* gout, const void* const* x, const void* const* up,
                            const void* const* low, const float* const* n_bits, void* const* out_or_dx, void* const* d_up,
                            void* const* d_low, void* const* d_nbits, const int64_t* O, long long I, int symmetric, int hard,
                            float threshold, int flags, hipStream_t stream) {
    const char* name = backward ? "miq_rowquant_mix_bwd_batch" : "miq_rowquant_mix_fwd_batch";
    if (n == 0 || I == 0) return 0;
    MIQ_CHECK_ARG(n > 0 && n <= kMaxBatch && x && n_bits && O && I > 0 && I < (1ll << 31), "%s: 1..%d matrices of one width expected", name,
                  kMaxBatch);
    MIQ_CHECK_ARG((up == nullptr) == (low == nullptr), "%s: up/low tables must both be given or both be null", name);
    MIQ_CHECK_ARG(backward ? (gout != nullptr) : (out_or_dx != nullptr), "%s: null pointer table", name);
    UniBatch<float> b{};
    MixBatchBits nb{};
    const bool lwc = up != nullptr && up[0] != nullptr;
    const bool want_dx = backward && out_or_dx != nullptr && out_or_dx[0] != nullptr;
    const bool want_dn = backward && d_nbits != nullptr && d_nbits[0] != nullptr;
    MIQ_CHECK_ARG(!backward || !lwc || (d_up && d_low), "%s: d_up/d_low tables required when up/low are given", name);
    bool vec_ok = (I % 4 == 0);
    long long rows = 0;
    b.n = 0;
    for (int i = 0; i < n; ++i) {
        MIQ_CHECK_ARG(O[i] >= 0, "%s: negative row count", name);
        if (O[i] == 0) continue;
        const int m = b.n++;
        b.x[m] = (const float*)x[i];
        b.up[m] = lwc ? (const float*)up[i] : nullptr;
        b.low[m] = lwc ? (const float*)low[i] : nullptr;
        b.gout[m] = backward ? (const float*)gout[i] : nullptr;
        b.out[m] = (backward && !want_dx) ? nullptr : (float*)out_or_dx[i];
        b.a1[m] = (backward && lwc) ? (float*)d_up[i] : nullptr;
        b.a2[m] = (backward && lwc) ? (float*)d_low[i] : nullptr;
        nb.n_bits[m] = n_bits[i];
        nb.d_nbits[m] = want_dn ? (float*)d_nbits[i] : nullptr;
        MIQ_CHECK_ARG(b.x[m] && nb.n_bits[m] && (!lwc || (b.up[m] && b.low[m])) && (!backward || b.gout[m]) && (backward || b.out[m]) &&
                          (!want_dx || b.out[m]) && (!(backward && lwc) || (b.a1[m] && b.a2[m])) && (!want_dn || nb.d_nbits[m]),
                      "%s: null pointer in entry %d (LWC, dx and d_nbits must be given for all matrices or for none)", name, i);
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
    geo.stream_out = stream_out_of(rows * I * (long long)sizeof(float));
    MixArgs a{};
    a.symmetric = symmetric;
    a.hard = hard;
    a.threshold = threshold;
    const int npack = vec_ok ? geo.g / 4 : geo.g;
    MIQ_CHECK_ARG(npack <= kRowMaxPacks, "%s: row of %d elements exceeds the register tile", name, geo.g);
#define LAUNCH_ONE(KERN, VECN, TEAM, NV)                                                                          \
    {                                                                                                             \
        constexpr int B = block_of<TEAM>();                                                                       \
        const long long nblk = (rows + (B / TEAM) - 1) / (B / TEAM);                                              \
        hipLaunchKernelGGL((KERN<float, VECN, TEAM, NV>), dim3((unsigned)nblk), dim3(B), 0, stream, b, nb, a, geo); \
    }
    if (!backward) {
        if (vec_ok) {
#define L(TEAM, NV) LAUNCH_ONE(k_mix_fwd_batch, 4, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        } else {
#define L(TEAM, NV) LAUNCH_ONE(k_mix_fwd_batch, 1, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        }
    } else {
        if (vec_ok) {
#define L(TEAM, NV) LAUNCH_ONE(k_mix_bwd_batch, 4, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        } else {
#define L(TEAM, NV) LAUNCH_ONE(k_mix_bwd_batch, 1, TEAM, NV)
            MIQ_ROW_DISPATCH(npack, L);
#undef L
        }
    }
#undef LAUNCH_ONE
    MIQ_CHECK_LAUNCH(name);
    return 0;
}

}  // namespace miq

extern "C" int miq_rowquant_mix_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                          const float* const* n_bits, void* const* out, const int64_t* O, int64_t I,
                                          int symmetric, int hard, float threshold, int flags, int dtype, void* stream) {
    MIQ_CHECK_ARG(dtype == MIQ_F32, "miq_rowquant_mix_fwd_batch: float32 only");
    return miq::launch_mix_batch(false, n, nullptr, x, up, low, n_bits, out, nullptr, nullptr, nullptr, O, I, symmetric, hard,
                                 threshold, flags, (hipStream_t)stream);
}

extern "C" int miq_rowquant_mix_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                          const void* const* low, const float* const* n_bits, void* const* dx,
                                          void* const* d_up, void* const* d_low, void* const* d_nbits, const int64_t* O,
                                          int64_t I, int symmetric, int hard, float threshold, int flags, int dtype,
                                          void* stream) {
    MIQ_CHECK_ARG(dtype == MIQ_F32, "miq_rowquant_mix_bwd_batch: float32 only");
    return miq::launch_mix_batch(true, n, grad_out, x, up, low, n_bits, dx, d_up, d_low, d_nbits, O, I, symmetric, hard,
                                 threshold, flags, (hipStream_t)stream);
}

extern "C" int miq_rowquant_mix_fwd(const void* x, const void* up, const void* low, const float* n_bits, void* out,
                                    int64_t O, int64_t I, int64_t group, int symmetric, int hard, float threshold,
                                    int flags, int dtype, void* stream) {
    MIQ_CHECK_ARG(dtype == MIQ_F32, "miq_rowquant_mix_fwd: float32 only");
    return miq::launch_mix(false, nullptr, x, up, low, n_bits, out, nullptr, nullptr, nullptr, O, I, group, symmetric, hard,
                           threshold, flags, (hipStream_t)stream);
}

extern "C" int miq_rowquant_mix_bwd(const void* grad_out, const void* x, const void* up, const void* low,
                                    const float* n_bits, void* dx, void* d_up, void* d_low, void* d_nbits, int64_t O,
                                    int64_t I, int64_t group, int symmetric, int hard, float threshold, int flags,
                                    int dtype, void* stream) {
    MIQ_CHECK_ARG(dtype == MIQ_F32, "miq_rowquant_mix_bwd: float32 only");
    return miq::launch_mix(true, grad_out, x, up, low, n_bits, dx, d_up, d_low, d_nbits, O, I, group, symmetric, hard,
                           threshold, flags, (hipStream_t)stream);
}
