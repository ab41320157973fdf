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
// Work distribution: a team of threads owns one row at a time and strides over the rows, so that
// every thread keeps its columns: the column vectors cs / shift are loaded once per thread and the
// column sums of the backward (d_cs, d_shift) are accumulated in registers across the team's rows,
// written as one partial row per team and folded by a second small kernel (deterministic).
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
    int preact;
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
    const ClipCtx cl = lwc_clip<float>(a.up, a.low, a.preact, row, rmin, rmax);
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
    __device__ __forceinline__ void load(const float* p, int npack, int lane) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            v[k] = (p && j < npack) ? *reinterpret_cast<const float4*>(p + (long long)j * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    }
    __device__ __forceinline__ float at(int k, int e) const { return reinterpret_cast<const float*>(&v[k])[e]; }
    __device__ __forceinline__ float& at(int k, int e) { return reinterpret_cast<float*>(&v[k])[e]; }
};

// t = (w * cs) [/ or *] r, in this order, every step rounded to fp32 like the PyTorch ops it replaces
__device__ __forceinline__ float let_apply(float w, float cs, bool has_cs, float r, bool has_r, bool rdiv) {
    float t = has_cs ? w * cs : w;
    if (has_r) t = rdiv ? t / r : t * r;
    return t;
}

template <int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_let_fwd(LetArgs a, float* __restrict__ out, float* __restrict__ scale_out, float* __restrict__ rzp_out,
          float* __restrict__ wshift_out) {
    __shared__ float lds[48];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const int npack = a.I / 4;
    const bool has_cs = a.cs != nullptr, has_r = a.rs != nullptr, rdiv = a.rs_div != 0;
    ColVec<TEAM, NV> cs, sh;
    cs.load(a.cs, npack, lane);
    if (a.shift) sh.load(a.shift, npack, lane);
    for (long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM; row < a.O; row += (long long)gridDim.x * RPB) {
        RowTile<float, 4, TEAM, NV> x;
        x.load(a.w + row * a.I, a.I, lane);
        if (a.shift) {
            float d[1] = {0.f};
            x.for_each([&](int k, int e, float w) { d[0] += w * sh.at(k, e); });
            team_sum<TEAM, 1>(d, lds);
            if (lane == 0) wshift_out[row] = d[0];
        }
        const float r = has_r ? a.rs[row] : 1.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) x.v[k].e[e] = let_apply(x.v[k].e[e], cs.at(k, e), has_cs, r, has_r, rdiv);
        }
        float rmin, rmax;
        row_minmax(x, false, lds, rmin, rmax);
        const LetCtx c = let_ctx(a, row, rmin, rmax);
        if (lane == 0) {
            scale_out[row] = c.s;
            rzp_out[row] = c.rzp;
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<float, 4> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xi = rint_ste(x.v[k].e[e] / c.s) + c.rzp;
                o.e[e] = (clampf(xi, 0.f, c.qmax) - c.rzp) * c.s;
            }
            st_pack<float, 4>(out + row * a.I + x.offset(k), o);
        }
    }
}

// Backward.  dt = gradient w.r.t. t through the quantizer (straight-through pass + the routing of the
// scale gradient to the row extremes, as in k_uniform_bwd).  Then
//   d_rs[o]    = sum_i -dt * (t / r)     (division)      |  sum_i dt * (w*cs)   (multiplication)
//   d_t1       = dt / r                                  |  dt * r
//   d_cs[i]    = sum_o d_t1 * w
//   d_shift[i] = sum_o w * g_wshift[o]
template <int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_let_bwd(LetArgs a, const float* __restrict__ gout, const float* __restrict__ g_wshift, float* __restrict__ d_up,
          float* __restrict__ d_low, float* __restrict__ d_rs, float* __restrict__ part_cs, float* __restrict__ part_sh) {
    __shared__ float lds[64];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const int team = threadIdx.x / TEAM;
    const int npack = a.I / 4;
    const bool has_cs = a.cs != nullptr, has_r = a.rs != nullptr, rdiv = a.rs_div != 0;
    const bool want_cs = part_cs != nullptr, want_sh = part_sh != nullptr;
    ColVec<TEAM, NV> cs, acc_cs, acc_sh;
    cs.load(a.cs, npack, lane);
#pragma unroll
    for (int k = 0; k < NV; ++k) acc_cs.v[k] = acc_sh.v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long row = (long long)blockIdx.x * RPB + team; row < a.O; row += (long long)gridDim.x * RPB) {
        RowTile<float, 4, TEAM, NV> w, g;
        w.load(a.w + row * a.I, a.I, lane);
        g.load(gout + row * a.I, a.I, lane);
        const float r = has_r ? a.rs[row] : 1.f;
        if (want_sh) {
            const float gs = g_wshift[row];
            w.for_each([&](int k, int e, float wv) { acc_sh.at(k, e) += wv * gs; });
        }
        // statistics of the transformed row
        float mn = __builtin_inff(), mx = -__builtin_inff();
        w.for_each([&](int k, int e, float wv) {
            const float t = let_apply(wv, cs.at(k, e), has_cs, r, has_r, rdiv);
            mn = nmin(mn, t);
            mx = nmax(mx, t);
        });
        float mm[2] = {mn, mx};
        team_reduce<TEAM, 2>(mm, OpMinMax(), lds);
        const float rmin = mm[0], rmax = mm[1];
        const LetCtx c = let_ctx(a, row, rmin, rmax);
        float acc[3] = {0.f, 0.f, 0.f};  // dS, #t == rmax, #t == rmin
        unsigned long long inside = 0ull;
        static_assert(NV * 4 <= 64, "mask of un-clamped elements is one 64-bit word");
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!w.has(k)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = let_apply(w.v[k].e[e], cs.at(k, e), has_cs, r, has_r, rdiv);
                const float gv = g.v[k].e[e];
                const float q = t / c.s;
                const float xi = rint_ste(q) + c.rzp;
                const bool m = (xi >= 0.f) && (xi <= c.qmax);
                const float xd = clampf(xi, 0.f, c.qmax) - c.rzp;
                acc[0] += gv * (xd - (m ? q : 0.f));
                inside |= (m ? 1ull : 0ull) << (k * 4 + e);
                acc[1] += (t == rmax) ? 1.f : 0.f;
                acc[2] += (t == rmin) ? 1.f : 0.f;
            }
        }
        team_sum<TEAM, 3>(acc, lds);
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
        if (a.up && lane == 0) {
            const float ju = a.preact ? 1.f : c.su * (1.f - c.su);
            const float jl = a.preact ? 1.f : c.sl * (1.f - c.sl);
            d_up[row] = d_xmax * rmax * ju;
            d_low[row] = d_xmin * rmin * jl;
        }
        const float share_max = (d_xmax * c.su) / acc[1];
        const float share_min = (d_xmin * c.sl) / acc[2];
        float drow[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!w.has(k)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float wv = w.v[k].e[e];
                const float t1 = has_cs ? wv * cs.at(k, e) : wv;
                const float t = has_r ? (rdiv ? t1 / r : t1 * r) : t1;
                float dt = ((inside >> (k * 4 + e)) & 1ull) ? g.v[k].e[e] : 0.f;
                if (t == rmax) dt += share_max;
                if (t == rmin) dt += share_min;
                float d_t1 = dt;
                if (has_r) {
                    if (rdiv) {
                        drow[0] += -dt * (t / r);
                        d_t1 = dt / r;
                    } else {
                        drow[0] += dt * t1;
                        d_t1 = dt * r;
                    }
                }
                if (want_cs) acc_cs.at(k, e) += d_t1 * wv;
            }
        }
        if (has_r && d_rs) {
            team_sum<TEAM, 1>(drow, lds);
            if (lane == 0) d_rs[row] = drow[0];
        }
    }
    // one partial row per team (teams that had no row write zeros)
    const long long prow = (long long)blockIdx.x * RPB + team;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * TEAM + lane;
        if (j >= npack) continue;
        if (want_cs) *reinterpret_cast<float4*>(part_cs + prow * a.I + (long long)j * 4) = acc_cs.v[k];
        if (want_sh) *reinterpret_cast<float4*>(part_sh + prow * a.I + (long long)j * 4) = acc_sh.v[k];
    }
}

// out[c] = sum over the nrows partial rows; block = 64 columns x 4 waves (wave w takes rows w, w+4, ...)
__global__ void __launch_bounds__(256)
k_colsum(const float* __restrict__ part, int nrows, int C, float* __restrict__ out) {
    __shared__ float sm[4][64];
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

// number of blocks: enough to fill the chip, few enough that the partial rows stay a small fraction
// of the matrix
static inline int let_grid(long long O, int rpb) {
    long long nb = (O + rpb - 1) / rpb;
    const long long cap = 512;
    return (int)(nb < cap ? nb : cap);
}

#define LET_DISPATCH(NPACK, L)                \
    do {                                      \
        if ((NPACK) <= 64) { L(16, 4); }      \
        else if ((NPACK) <= 256) { L(64, 4); }   \
        else if ((NPACK) <= 1024) { L(256, 4); } \
        else if ((NPACK) <= 2048) { L(256, 8); } \
        else { L(512, 8); }                   \
    } while (0)

static inline int let_rpb(int npack) {
    if (npack <= 64) return 256 / 16;
    if (npack <= 256) return 256 / 64;
    return 1;
}

static int let_check(const char* name, const LetArgs& a) {
    MIQ_CHECK_ARG(a.w && a.O > 0 && a.I > 0, "%s: null weight pointer or bad shape", name);
    MIQ_CHECK_ARG(a.I % 4 == 0 && a.I / 4 <= kLetMaxPacks, "%s: I=%d must be a multiple of 4 and at most %d", name, a.I, 4 * kLetMaxPacks);
    MIQ_CHECK_ARG((a.up == nullptr) == (a.low == nullptr), "%s: up/low must both be given or both be null", name);
    MIQ_CHECK_ARG(a.n_bits >= 2 && a.n_bits <= 15, "%s: n_bits=%d outside [2,15]", name, a.n_bits);
    MIQ_CHECK_ARG(is_aligned16(a.w) && is_aligned16(a.cs) && is_aligned16(a.shift), "%s: w / cs / shift must be 16-byte aligned", name);
    return 0;
}

}  // namespace miq

using namespace miq;

extern "C" int miq_let_partial_rows(int64_t O, int64_t I) {
    if (O <= 0 || I <= 0 || I % 4) return 0;
    const int rpb = let_rpb((int)(I / 4));
    return let_grid(O, rpb) * rpb;
}

extern "C" int miq_let_rowquant_fwd(const float* w, const float* cs, const float* rs, int rs_div, const float* shift,
                                    const float* up, const float* low, float* out, float* scale_out, float* rzp_out,
                                    float* wshift_out, int64_t O, int64_t I, int n_bits, int symmetric, int flags,
                                    void* stream) {
    if (O == 0 || I == 0) return 0;
    LetArgs a{w, cs, rs, rs_div, shift, up, low, (flags >> 1) & 1, n_bits, symmetric, O, (int)I};
    MIQ_CHECK_ARG(I < (1ll << 31), "miq_let_rowquant_fwd: matrix too wide");
    if (int rc = let_check("miq_let_rowquant_fwd", a)) return rc;
    MIQ_CHECK_ARG(out && scale_out && rzp_out && is_aligned16(out), "miq_let_rowquant_fwd: null or unaligned output");
    MIQ_CHECK_ARG(!shift || wshift_out, "miq_let_rowquant_fwd: wshift_out required with shift");
    const int npack = (int)(I / 4);
    const int grid = let_grid(O, let_rpb(npack));
#define L(TEAM, NV)                                                                                              \
    hipLaunchKernelGGL((k_let_fwd<TEAM, NV>), dim3(grid), dim3(block_of<TEAM>()), 0, (hipStream_t)stream, a, out, \
                       scale_out, rzp_out, wshift_out)
    LET_DISPATCH(npack, L);
#undef L
    MIQ_CHECK_LAUNCH("miq_let_rowquant_fwd");
    return 0;
}

extern "C" int miq_let_rowquant_bwd(const float* grad_out, const float* g_wshift, const float* w, const float* cs,
                                    const float* rs, int rs_div, const float* shift, const float* up, const float* low,
                                    float* d_cs, float* d_rs, float* d_shift, float* d_up, float* d_low, float* workspace,
                                    int64_t O, int64_t I, int n_bits, int symmetric, int flags, void* stream) {
    if (O == 0 || I == 0) return 0;
    LetArgs a{w, cs, rs, rs_div, shift, up, low, (flags >> 1) & 1, n_bits, symmetric, O, (int)I};
    MIQ_CHECK_ARG(I < (1ll << 31), "miq_let_rowquant_bwd: matrix too wide");
    if (int rc = let_check("miq_let_rowquant_bwd", a)) return rc;
    MIQ_CHECK_ARG(grad_out && is_aligned16(grad_out), "miq_let_rowquant_bwd: null or unaligned grad_out");
    MIQ_CHECK_ARG(!up || (d_up && d_low), "miq_let_rowquant_bwd: d_up/d_low required when up/low are given");
    MIQ_CHECK_ARG(!d_cs || cs, "miq_let_rowquant_bwd: d_cs needs cs");
    MIQ_CHECK_ARG(!d_rs || rs, "miq_let_rowquant_bwd: d_rs needs rs");
    MIQ_CHECK_ARG(!d_shift || (shift && g_wshift), "miq_let_rowquant_bwd: d_shift needs shift and g_wshift");
    MIQ_CHECK_ARG((!d_cs && !d_shift) || (workspace && is_aligned16(workspace)), "miq_let_rowquant_bwd: workspace required");
    const int npack = (int)(I / 4);
    const int rpb = let_rpb(npack);
    const int grid = let_grid(O, rpb);
    const long long prow = (long long)grid * rpb;
    float* part_cs = d_cs ? workspace : nullptr;
    float* part_sh = d_shift ? workspace + (d_cs ? prow * I : 0) : nullptr;
#define L(TEAM, NV)                                                                                                  \
    hipLaunchKernelGGL((k_let_bwd<TEAM, NV>), dim3(grid), dim3(block_of<TEAM>()), 0, (hipStream_t)stream, a, grad_out, \
                       g_wshift, d_up, d_low, d_rs, part_cs, part_sh)
    LET_DISPATCH(npack, L);
#undef L
    const unsigned fg = (unsigned)((I + 63) / 64);
    if (d_cs) hipLaunchKernelGGL(k_colsum, dim3(fg), dim3(256), 0, (hipStream_t)stream, part_cs, (int)prow, (int)I, d_cs);
    if (d_shift) hipLaunchKernelGGL(k_colsum, dim3(fg), dim3(256), 0, (hipStream_t)stream, part_sh, (int)prow, (int)I, d_shift);
    MIQ_CHECK_LAUNCH("miq_let_rowquant_bwd");
    return 0;
}
