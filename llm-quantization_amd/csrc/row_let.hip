// Learnable equivalent transformation (LET) fused into the row-wise uniform weight quantizer.
//
// In every calibration step the reference first materialises the smoothed weight
//   t = (W * cs[None, :]) / rs[:, None]   (or * rs[:, None])       models/transformation.py:24-69
// and the shift-induced bias W @ shift with separate full-matrix PyTorch ops, then fake-quantizes t
// (quantize/quantizer.py:248-273,99-121), and autograd walks the same chain backwards.  Here one
// kernel reads W once, applies the same IEEE operations in the same order in registers (so the
// quantized weight is bit-identical), and writes only the quantized weight; the backward kernel
// reads W and the incoming gradient once and produces the gradients of cs, rs, shift and of the
// clipping factors directly, without materialising the gradient of t.
//
// fp32 only (the calibration loop keeps LET / LWC parameters and weights in fp32), whole matrix rows
// as quantization rows (no groups), I a multiple of 4.
//
// Work distribution: forward and the row half of the backward use one team of threads per row (the
// tiling of the plain row quantizer); the column sums of the backward (d_cs, d_shift) come from a
// second, column-oriented streaming kernel that rebuilds the element gradients from a small per-row
// context, followed by a deterministic fold of its row-chunk partials.
#include "miq_common.h"

namespace miq {

constexpr int kLetMaxPacks = 4096;  // widest row the register tiles of these kernels hold (16384 columns)

struct LetArgs {
    const float* w;      // [O, I]
    const float* cs;     // [I] column factor or nullptr
    const float* rs;     // [O] row factor or nullptr
    int rs_div;          // 1: t = (w*cs) / rs, 0: t = (w*cs) * rs
    const float* shift;  // [I] or nullptr: wshift[o] = sum_i w[o,i] * shift[i]
    const float* up;     // [O] LWC factors or nullptr
    const float* low;
    int n_bits;
    int symmetric;
    long long O;
    int I;
};

// scale / zero point of one row: UniformAffineQuantizer.per_token_dynamic_calibration
// (quantize/quantizer.py:248-273; the asymmetric scale is not clamped there)
struct LetCtx {
    float s, rzp, qmax, su, sl, xmax, xmin, s_raw, qdiv;
};

__device__ __forceinline__ LetCtx let_ctx(const LetArgs& a, long long row, float rmin, float rmax) {
    LetCtx c;
    const float n = (float)a.n_bits;
    c.qmax = exp2f(n) - 1.0f;
    const ClipCtx cl = lwc_clip<float>(a.up, a.low, row, rmin, rmax);
    c.su = cl.su;
    c.sl = cl.sl;
    c.xmax = cl.xmax;
    c.xmin = cl.xmin;
    if (a.symmetric) {
        const float amax = nmax(fabsf(c.xmax), fabsf(c.xmin));
        c.qdiv = exp2f(n - 1.0f) - 1.0f;
        c.s_raw = amax / c.qdiv;
        c.s = clampf(c.s_raw, kClipLo, kClipHi);
        c.rzp = c.qdiv;
    } else {
        c.qdiv = c.qmax;
        c.s_raw = (c.xmax - c.xmin) / c.qmax;
        c.s = c.s_raw;
        c.rzp = rintf(clampf(-c.xmin / c.s, -kClipHi, kClipHi));
    }
    return c;
}

template <int TEAM, int NV>
struct ColVec {
    float4 v[NV];
    // packs beyond the row re-read the last one (their elements are never used); a null vector reads as `fill`
    __device__ __forceinline__ void load(const float* p, int npack, int lane, float fill) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = make_float4(fill, fill, fill, fill);
        if (!p) return;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            v[k] = *reinterpret_cast<const float4*>(p + (long long)(j < npack ? j : npack - 1) * 4);
        }
    }
    __device__ __forceinline__ float at(int k, int e) const { return reinterpret_cast<const float*>(&v[k])[e]; }
};

// t = (w * cs) [/ or *] r, in this order, every step rounded to fp32 like the PyTorch ops it replaces.  Which of the
// factors exist is a property of the launch: an absent factor is passed as 1 (x * 1 is x, bit for bit), and division
// against multiplication is a template parameter, so that the element loops carry no test of the launch's options (left
// inside, each test stays a scalar branch around one or two instructions, three per element).
template <bool RDIV>
__device__ __forceinline__ float let_apply(float w, float cs, float r) {
    const float t = w * cs;
    return RDIV ? t / r : t * r;
}
// The same for the backward kernels, with the corrected reciprocal product x * rr, rr = 1 / r, in place of the
// division (UncheckedDivisor: the IEEE quotient for every finite operand pair away from the ends of the exponent
// range).  Both backward kernels use this one function, so they agree on t bit for bit.
__device__ __forceinline__ float quot(float x, float d, float rd) {
    const float q0 = x * rd;
    return __builtin_fmaf(__builtin_fmaf(-q0, d, x), rd, q0);
}
template <bool RDIV>
__device__ __forceinline__ float let_apply_bwd(float w, float cs, float r, float rr) {
    const float t = w * cs;
    return RDIV ? quot(t, r, rr) : t * r;
}

template <typename TO, int TEAM, int NV, bool RDIV>
// (the 6-pack tile sits at 128 to 130 registers: held to 128, two of its 512-thread blocks share a CU)
__global__ void __launch_bounds__(block_of<TEAM>(), NV == 6 ? 4 : 1)
k_let_fwd(LetArgs a, TO* __restrict__ out, float* __restrict__ scale_out, float* __restrict__ rzp_out,
          float* __restrict__ wshift_out) {
    __shared__ float lds[48];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const int npack = a.I / 4;
    const bool has_r = a.rs != nullptr;
    // The column vectors (a few KB, L2-resident) do not depend on the row: requested once, before the first row's data,
    // with clamped indices instead of guards (a guarded load is issued behind its own branch, and read at its use inside
    // the element loop each one exposed an L2 round trip).  Absent vectors are 1 / 0: x * 1 is x, x * 0 adds nothing.
    ColVec<TEAM, NV> cvec, svec;
    cvec.load(a.cs, npack, lane, 1.f);
    svec.load(a.shift, npack, lane, 0.f);
    for (long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM; row < a.O; row += (long long)gridDim.x * RPB) {
        RowTile<float, 4, TEAM, NV> x;
        x.load(a.w + row * a.I, a.I, lane);
        const float r = has_r ? a.rs[row] : 1.f;
        float d[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d[0] += x.v[k].e[e] * svec.at(k, e);
                x.v[k].e[e] = let_apply<RDIV>(x.v[k].e[e], cvec.at(k, e), r);
            }
        }
        if (a.shift) {
            team_sum<TEAM, 1>(d, lds);
            if (lane == 0) wshift_out[row] = d[0];
        }
        float rmin, rmax;
        row_minmax(x, false, lds, rmin, rmax);
        const LetCtx c = let_ctx(a, row, rmin, rmax);
        if (lane == 0) {
            scale_out[row] = c.s;
            rzp_out[row] = c.rzp;
        }
        // the level only needs the quotient rounded to an integer: level_div where the scale allows it (nearly always)
        auto quantize_row = [&](const auto& dv) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!x.has(k)) continue;
                Pack<TO, 4> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xi = rint_ste(dv.div(x.v[k].e[e])) + c.rzp;
                    // TO = __half: the fp32 result rounded once more, i.e. what a cast of the fp32 tensor gives
                    o.e[e] = from_f<TO>((clampf(xi, 0.f, c.qmax) - c.rzp) * c.s);
                }
                st_pack<TO, 4>(out + row * a.I + x.offset(k), o);
            }
        };
        if (level_div_ok(c.s)) quantize_row(LevelDivisor(c.s));
        else quantize_row(Divisor<float>(c.s));
    }
}

// Backward in two streaming launches (both read w and the incoming gradient once; the second read is
// served from the Infinity Cache for matrices below ~100 MiB):
//   rows kernel   one row per team (like k_uniform_bwd): row statistics, dS, tie counts -> d_up, d_low,
//                 d_rs, and the per-row context {s, rzp, rmax, rmin, share_max, share_min} for the
//                 column kernel;
//   cols kernel   lane = 4 columns, wave = every 4th row of a row chunk: rebuilds
//                 dt = [level inside the grid] * g + shares at the row extremes, and accumulates
//                 d_cs[i] = sum_o d_t1 * w and d_shift[i] = sum_o w * g_wshift[o] per chunk; a small
//                 fold sums the chunks (deterministic).
// With dt the gradient w.r.t. t:
//   d_rs[o] = sum_i -dt * (t / r)   (division)   |   sum_i dt * (w*cs)   (multiplication)
//   d_t1    = dt / r                             |   dt * r
// A register-resident single pass (rows strided over persistent teams, column sums in registers) was
// measured at 1.2 TB/s: three dependent team reductions per row with only two rows in flight per CU.
struct RowCtx {
    float s, rzp, rmax, rmin, share_max, share_min, inv_s, inv_r;  // inv_*: 1 / s and 1 / rs[row]
};

template <typename TG, int TEAM, int NV, bool RDIV>
__global__ void __launch_bounds__(block_of<TEAM>(), 4)
k_let_bwd_rows(LetArgs a, const TG* __restrict__ gout, float* __restrict__ d_up, float* __restrict__ d_low,
               float* __restrict__ d_rs, RowCtx* __restrict__ rowctx) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= a.O) return;
    const bool has_r = a.rs != nullptr;
    RowTile<float, 4, TEAM, NV> w;
    RowTile<TG, 4, TEAM, NV> g;
    ColVec<TEAM, NV> cvec;
    w.load(a.w + row * a.I, a.I, lane);
    g.load(gout + row * a.I, a.I, lane);
    cvec.load(a.cs, a.I / 4, lane, 1.f);  // requested with the row, not at its use
    const float r = has_r ? a.rs[row] : 1.f;
    const float rr = 1.0f / r;
    // w.v <- t
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!w.has(k)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) w.v[k].e[e] = let_apply_bwd<RDIV>(w.v[k].e[e], cvec.at(k, e), r, rr);
    }
    float rmin, rmax;
    row_minmax(w, false, lds, rmin, rmax);
    const LetCtx c = let_ctx(a, row, rmin, rmax);
    const float inv_s = 1.0f / c.s;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // dS, #t == rmax, #t == rmin, sum of m*g*t (straight-through part of d_rs)
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!w.has(k)) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float t = w.v[k].e[e];
            const float gv = to_f<TG>(g.v[k].e[e]);
            const float q = quot(t, c.s, inv_s);
            const float xi = rint_ste(q) + c.rzp;
            const bool m = (xi >= 0.f) && (xi <= c.qmax);
            const float xd = clampf(xi, 0.f, c.qmax) - c.rzp;
            acc[0] += gv * (xd - (m ? q : 0.f));
            acc[1] += (t == rmax) ? 1.f : 0.f;
            acc[2] += (t == rmin) ? 1.f : 0.f;
            acc[3] += m ? gv * t : 0.f;
        }
    }
    team_sum<TEAM, 4>(acc, lds);
    float d_xmax, d_xmin;
    if (a.symmetric) {
        const bool live = c.s_raw >= kClipLo && c.s_raw <= kClipHi;
        ClipCtx cl;
        cl.su = c.su; cl.sl = c.sl; cl.xmax = c.xmax; cl.xmin = c.xmin;
        route_absmax(live ? acc[0] / c.qdiv : 0.f, cl, d_xmax, d_xmin);
    } else {
        d_xmax = acc[0] / c.qmax;
        d_xmin = -d_xmax;
    }
    if (lane != 0) return;
    if (a.up) {
        d_up[row] = d_xmax * rmax;   // w.r.t. the activated factors
        d_low[row] = d_xmin * rmin;
    }
    const float tot_max = d_xmax * c.su, tot_min = d_xmin * c.sl;  // gradient mass routed to the row max / min
    if (has_r && d_rs) {
        // sum_i dt*t with dt = m*g + shares at the extremes (t == rmax / rmin there): division by r for the
        // quotient rule, identity for the product rule (dt * t1 with t1 = t / r up to one rounding)
        const float sum_dt_t = acc[3] + tot_max * rmax + tot_min * rmin;
        d_rs[row] = RDIV ? -sum_dt_t / r : sum_dt_t / r;
    }
    if (rowctx) {
        RowCtx o;
        o.s = c.s; o.rzp = c.rzp; o.rmax = rmax; o.rmin = rmin;
        o.share_max = tot_max / acc[1];
        o.share_min = tot_min / acc[2];
        o.inv_s = inv_s;
        o.inv_r = rr;
        rowctx[row] = o;
    }
}

template <typename TG, bool RDIV>
__global__ void __launch_bounds__(256)
k_let_bwd_cols(LetArgs a, const TG* __restrict__ gout, const float* __restrict__ g_wshift, const RowCtx* __restrict__ rowctx,
               int rows_per_chunk, float* __restrict__ part_cs, float* __restrict__ part_sh) {
    __shared__ float4 sm[2][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * 4;
    const bool valid = col0 < a.I;
    const bool has_cs = a.cs != nullptr, has_r = a.rs != nullptr;
    const bool want_cs = part_cs != nullptr, want_sh = part_sh != nullptr;
    const float qmax = exp2f((float)a.n_bits) - 1.0f;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > a.O) r1 = a.O;
    float acs[4] = {0.f, 0.f, 0.f, 0.f}, ash[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        float cv[4] = {1.f, 1.f, 1.f, 1.f};
        if (has_cs) {
            const float4 c4 = *reinterpret_cast<const float4*>(a.cs + col0);
            cv[0] = c4.x; cv[1] = c4.y; cv[2] = c4.z; cv[3] = c4.w;
        }
        auto body = [&](long long row, const Pack<float, 4>& w, const Pack<TG, 4>& g) {
            const RowCtx c = rowctx[row];  // wave-uniform address
            const float r = has_r ? a.rs[row] : 1.f;
            const float gs = want_sh ? g_wshift[row] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float wv_ = w.e[e];
                const float t = let_apply_bwd<RDIV>(wv_, cv[e], r, c.inv_r);
                const float xi = rint_ste(quot(t, c.s, c.inv_s)) + c.rzp;
                float dt = ((xi >= 0.f) && (xi <= qmax)) ? to_f<TG>(g.e[e]) : 0.f;
                if (t == c.rmax) dt += c.share_max;
                if (t == c.rmin) dt += c.share_min;
                const float d_t1 = RDIV ? quot(dt, r, c.inv_r) : dt * r;  // r is 1 without a row factor
                acs[e] += d_t1 * wv_;
                ash[e] += wv_ * gs;
            }
        };
        long long row = r0 + wv;
        for (; row + 4 < r1; row += 8) {
            const long long ra = __builtin_amdgcn_readfirstlane((int)row), rb = ra + 4;
            const Pack<float, 4> w0 = ld_pack<float, 4>(a.w + ra * a.I + col0), w1 = ld_pack<float, 4>(a.w + rb * a.I + col0);
            const Pack<TG, 4> g0 = ld_pack<TG, 4>(gout + ra * a.I + col0), g1 = ld_pack<TG, 4>(gout + rb * a.I + col0);
            body(ra, w0, g0);
            body(rb, w1, g1);
        }
        for (; row < r1; row += 4) {
            const long long ra = __builtin_amdgcn_readfirstlane((int)row);
            const Pack<float, 4> w0 = ld_pack<float, 4>(a.w + ra * a.I + col0);
            const Pack<TG, 4> g0 = ld_pack<TG, 4>(gout + ra * a.I + col0);
            body(ra, w0, g0);
        }
    }
    sm[0][wv][lane] = make_float4(acs[0], acs[1], acs[2], acs[3]);
    sm[1][wv][lane] = make_float4(ash[0], ash[1], ash[2], ash[3]);
    __syncthreads();
    if (wv != 0 || !valid) return;
    auto fold = [&](int which) {
        const float4 p0 = sm[which][0][lane], p1 = sm[which][1][lane], p2 = sm[which][2][lane], p3 = sm[which][3][lane];
        return make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                           (p0.w + p1.w) + (p2.w + p3.w));
    };
    if (want_cs) *reinterpret_cast<float4*>(part_cs + (long long)blockIdx.y * a.I + col0) = fold(0);
    if (want_sh) *reinterpret_cast<float4*>(part_sh + (long long)blockIdx.y * a.I + col0) = fold(1);
}

// out[c] = sum over the nrows partial rows; block = 64 columns x 4 waves (wave w takes rows w, w+4, ...).  blockIdx.y
// picks one of up to two (partials, output) pairs, so the column-factor and the shift gradient fold in one launch.
__global__ void __launch_bounds__(256)
k_colsum(const float* __restrict__ part0, float* __restrict__ out0, const float* __restrict__ part1, float* __restrict__ out1,
         int nrows, int C) {
    __shared__ float sm[4][64];
    const float* __restrict__ part = blockIdx.y ? part1 : part0;
    float* __restrict__ out = blockIdx.y ? out1 : out0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (c < C) {
        const float* p = part + c;
        int j = w;
        for (; j + 12 < nrows; j += 16) {
            v0 += p[(long long)j * C];
            v1 += p[(long long)(j + 4) * C];
            v2 += p[(long long)(j + 8) * C];
            v3 += p[(long long)(j + 12) * C];
        }
        for (; j < nrows; j += 4) v0 += p[(long long)j * C];
    }
    sm[w][lane] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    if (w == 0 && c < C) out[c] = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
}

// 6-pack tiles for the 5120- and 11008-wide rows of Llama-2-13B / -7B: the 8-pack tile they would otherwise take holds
// 150 to 170 registers in the forward kernel (row, column factors and shifts in registers), one block per CU; with 6
// packs it is 128 and two blocks overlap (4096 x 11008 forward: 111 -> 77 us).
#define LET_DISPATCH(NPACK, L)                   \
    do {                                         \
        if ((NPACK) <= 64) { L(16, 4); }         \
        else if ((NPACK) <= 256) { L(64, 4); }   \
        else if ((NPACK) <= 1024) { L(256, 4); } \
        else if ((NPACK) <= 1536) { L(256, 6); } \
        else if ((NPACK) <= 2048) { L(256, 8); } \
        else if ((NPACK) <= 3072) { L(512, 6); } \
        else { L(512, 8); }                      \
    } while (0)

static inline int let_rpb(int npack) {
    if (npack <= 64) return 256 / 16;
    if (npack <= 256) return 256 / 64;
    return 1;
}

// row chunks of the column kernel: ~1024 blocks, at least 16 rows per chunk
static inline int let_chunks(long long O, long long I) {
    const long long tiles = (I / 4 + 63) / 64;
    long long want = 1024 / (tiles > 0 ? tiles : 1);
    const long long maxc = (O + 15) / 16;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

static int let_check(const char* name, const LetArgs& a) {
    MIQ_CHECK_ARG(a.w && a.O > 0 && a.I > 0, "%s: null weight pointer or bad shape", name);
    MIQ_CHECK_ARG(a.I % 4 == 0 && a.I / 4 <= kLetMaxPacks, "%s: I=%d must be a multiple of 4 and at most %d", name, a.I, 4 * kLetMaxPacks);
    MIQ_CHECK_ARG((a.up == nullptr) == (a.low == nullptr), "%s: up/low must both be given or both be null", name);
    MIQ_CHECK_ARG(a.n_bits >= 2 && a.n_bits <= 15, "%s: n_bits=%d outside [2,15]", name, a.n_bits);
    MIQ_CHECK_ARG(is_aligned16(a.w) && is_aligned16(a.cs) && is_aligned16(a.shift), "%s: w / cs / shift must be 16-byte aligned", name);
    return 0;
}

template <typename TO>
static void let_fwd_launch(const LetArgs& a, void* out, float* scale_out, float* rzp_out, float* wshift_out, hipStream_t stream) {
    const int npack = a.I / 4;
    const int rpb = let_rpb(npack);
    const unsigned grid = (unsigned)((a.O + rpb - 1) / rpb);
    const bool rdiv = a.rs != nullptr && a.rs_div != 0;
#define L(TEAM, NV)                                                                                                    \
    if (rdiv)                                                                                                          \
        hipLaunchKernelGGL((k_let_fwd<TO, TEAM, NV, true>), dim3(grid), dim3(block_of<TEAM>()), 0, stream, a, (TO*)out, \
                           scale_out, rzp_out, wshift_out);                                                            \
    else                                                                                                               \
        hipLaunchKernelGGL((k_let_fwd<TO, TEAM, NV, false>), dim3(grid), dim3(block_of<TEAM>()), 0, stream, a, (TO*)out, \
                           scale_out, rzp_out, wshift_out)
    LET_DISPATCH(npack, L);
#undef L
}

template <typename TG>
static void let_bwd_launch(const LetArgs& a, const void* gout, const float* g_wshift, float* d_cs, float* d_rs, float* d_shift,
                           float* d_up, float* d_low, float* ws, hipStream_t stream) {
    const int npack = a.I / 4;
    const int rpb = let_rpb(npack);
    const unsigned grid = (unsigned)((a.O + rpb - 1) / rpb);
    const bool cols = d_cs || d_shift;
    RowCtx* rowctx = cols ? reinterpret_cast<RowCtx*>(ws) : nullptr;
    const bool rdiv = a.rs != nullptr && a.rs_div != 0;
#define L(TEAM, NV)                                                                                                    \
    if (rdiv)                                                                                                          \
        hipLaunchKernelGGL((k_let_bwd_rows<TG, TEAM, NV, true>), dim3(grid), dim3(block_of<TEAM>()), 0, stream, a,     \
                           (const TG*)gout, d_up, d_low, d_rs, rowctx);                                                \
    else                                                                                                               \
        hipLaunchKernelGGL((k_let_bwd_rows<TG, TEAM, NV, false>), dim3(grid), dim3(block_of<TEAM>()), 0, stream, a,    \
                           (const TG*)gout, d_up, d_low, d_rs, rowctx)
    LET_DISPATCH(npack, L);
#undef L
    if (!cols) return;
    const int nchunk = let_chunks(a.O, a.I);
    const int rpc = (int)((a.O + nchunk - 1) / nchunk);
    float* part_cs = d_cs ? ws + 8 * a.O : nullptr;
    float* part_sh = d_shift ? ws + 8 * a.O + (d_cs ? (long long)nchunk * a.I : 0) : nullptr;
    const unsigned tiles = (unsigned)((npack + 63) / 64);
    if (rdiv)
        hipLaunchKernelGGL((k_let_bwd_cols<TG, true>), dim3(tiles, nchunk), dim3(256), 0, stream, a, (const TG*)gout, g_wshift, rowctx,
                           rpc, part_cs, part_sh);
    else
        hipLaunchKernelGGL((k_let_bwd_cols<TG, false>), dim3(tiles, nchunk), dim3(256), 0, stream, a, (const TG*)gout, g_wshift, rowctx,
                           rpc, part_cs, part_sh);
    const unsigned fg = (unsigned)((a.I + 63) / 64);
    if (d_cs && d_shift) hipLaunchKernelGGL(k_colsum, dim3(fg, 2), dim3(256), 0, stream, part_cs, d_cs, part_sh, d_shift, nchunk, a.I);
    else if (d_cs) hipLaunchKernelGGL(k_colsum, dim3(fg, 1), dim3(256), 0, stream, part_cs, d_cs, nullptr, nullptr, nchunk, a.I);
    else hipLaunchKernelGGL(k_colsum, dim3(fg, 1), dim3(256), 0, stream, part_sh, d_shift, nullptr, nullptr, nchunk, a.I);
}

}  // namespace miq

using namespace miq;

extern "C" int64_t miq_let_workspace(int64_t O, int64_t I) {
    if (O <= 0 || I <= 0 || I % 4) return 0;
    return 8 * O + 2ll * let_chunks(O, I) * I;
}

extern "C" int miq_let_rowquant_fwd(const float* w, const float* cs, const float* rs, int rs_div, const float* shift,
                                    const float* up, const float* low, void* out, float* scale_out, float* rzp_out,
                                    float* wshift_out, int64_t O, int64_t I, int n_bits, int symmetric, int flags,
                                    int out_dtype, void* stream) {
    if (O == 0 || I == 0) return 0;
    LetArgs a{w, cs, rs, rs_div, shift, up, low, n_bits, symmetric, O, (int)I};
    MIQ_CHECK_ARG(I < (1ll << 31), "miq_let_rowquant_fwd: matrix too wide");
    if (int rc = let_check("miq_let_rowquant_fwd", a)) return rc;
    MIQ_CHECK_ARG(out && scale_out && rzp_out && is_aligned16(out), "miq_let_rowquant_fwd: null or unaligned output");
    MIQ_CHECK_ARG(!shift || wshift_out, "miq_let_rowquant_fwd: wshift_out required with shift");
    MIQ_CHECK_ARG(out_dtype == MIQ_F32 || out_dtype == MIQ_F16, "miq_let_rowquant_fwd: unsupported output dtype %d", out_dtype);
    if (out_dtype == MIQ_F32) let_fwd_launch<float>(a, out, scale_out, rzp_out, wshift_out, (hipStream_t)stream);
    else let_fwd_launch<__half>(a, out, scale_out, rzp_out, wshift_out, (hipStream_t)stream);
    MIQ_CHECK_LAUNCH("miq_let_rowquant_fwd");
    return 0;
}

extern "C" int miq_let_rowquant_bwd(const void* grad_out, const float* g_wshift, const float* w, const float* cs,
                                    const float* rs, int rs_div, const float* shift, const float* up, const float* low,
                                    float* d_cs, float* d_rs, float* d_shift, float* d_up, float* d_low, float* workspace,
                                    int64_t O, int64_t I, int n_bits, int symmetric, int flags, int grad_dtype, void* stream) {
    if (O == 0 || I == 0) return 0;
    LetArgs a{w, cs, rs, rs_div, shift, up, low, n_bits, symmetric, O, (int)I};
    MIQ_CHECK_ARG(I < (1ll << 31), "miq_let_rowquant_bwd: matrix too wide");
    if (int rc = let_check("miq_let_rowquant_bwd", a)) return rc;
    MIQ_CHECK_ARG(grad_out && is_aligned16(grad_out), "miq_let_rowquant_bwd: null or unaligned grad_out");
    MIQ_CHECK_ARG(!up || (d_up && d_low), "miq_let_rowquant_bwd: d_up/d_low required when up/low are given");
    MIQ_CHECK_ARG(!d_cs || cs, "miq_let_rowquant_bwd: d_cs needs cs");
    MIQ_CHECK_ARG(!d_rs || rs, "miq_let_rowquant_bwd: d_rs needs rs");
    MIQ_CHECK_ARG(!d_shift || (shift && g_wshift), "miq_let_rowquant_bwd: d_shift needs shift and g_wshift");
    MIQ_CHECK_ARG((!d_cs && !d_shift) || (workspace && is_aligned16(workspace)), "miq_let_rowquant_bwd: workspace required");
    MIQ_CHECK_ARG(grad_dtype == MIQ_F32 || grad_dtype == MIQ_F16, "miq_let_rowquant_bwd: unsupported gradient dtype %d", grad_dtype);
    if (grad_dtype == MIQ_F32) let_bwd_launch<float>(a, grad_out, g_wshift, d_cs, d_rs, d_shift, d_up, d_low, workspace, (hipStream_t)stream);
    else let_bwd_launch<__half>(a, grad_out, g_wshift, d_cs, d_rs, d_shift, d_up, d_low, workspace, (hipStream_t)stream);
    MIQ_CHECK_LAUNCH("miq_let_rowquant_bwd");
    return 0;
}
