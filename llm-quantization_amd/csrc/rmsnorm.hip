// OmniLlamaRMSNorm (reference quantize/omni_norm.py:47-71) in one pass per direction:
//     var  = mean(x.float()^2, -1);  u = x * rsqrt(var + eps);  y = (weight * u [+ bias]).to(x.dtype)
// PyTorch runs this as eight elementwise / reduction kernels forward and about twice as many backward; the
// calibration step evaluates it twice per block with temporary (LET-smoothed) weight and bias vectors whose
// gradients reach the smoothing parameters.
//   forward   a team of threads owns one row: x stays in registers, fp32 sum of squares, y written once; rstd saved.
//   backward  rows kernel: dx = rstd * (gy*w - u * mean(gy*w*u));
//             columns kernel (lane = 4 channels, wave = every 4th row of a row chunk): partial sums of
//             d_weight[c] = sum_r gy*u and d_bias[c] = sum_r gy, then a deterministic fold of the chunks.
// All arithmetic in fp32 like the reference (every intermediate there is an fp32 tensor); the output is rounded once.
#include "miq_common.h"
#include "slot_table.h"

namespace miq {

struct RmsArgs {
    const void* x;
    const float* weight;  // [C] fp32
    const float* bias;    // [C] fp32 or nullptr
    float eps;
    long long R;
    int C;
};

// STATS: the kernel also folds the minima / maxima of the values it stores (after rounding to TX) into the slot table
// of `sink`, which is what the statistics pass of the activation quantizer would compute from `out`.  A team then walks
// kStatRows consecutive rows, keeps the running extremes of its channels in registers and publishes once (a quarter of
// the exchanges and atomics of one row per team); a team past the last row repeats the last row (same stores, same
// statistics) so that every lane takes part in the exchange.
template <typename TX, int VEC, int TEAM, int NV, bool STATS>
__global__ void __launch_bounds__(block_of<TEAM>())
k_rmsnorm_fwd(RmsArgs a, TX* __restrict__ out, float* __restrict__ rstd_out, RowStatSink sink) {
    static_assert(VEC == 4, "the weight / bias vectors are read as float4");
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    constexpr int ROWS = STATS ? kStatRows : 1;
    const int lane = threadIdx.x % TEAM;
    const long long row0 = ((long long)blockIdx.x * RPB + threadIdx.x / TEAM) * ROWS;
    unsigned tag = 0;
    if (STATS) {
        tag = (unsigned)sink.ws[0] + 1u;
        if (blockIdx.x == 0 && threadIdx.x == 0) sink.ws[1] = tag;
    } else if (row0 >= a.R) {
        return;  // TEAM > 64 implies one row per block: block-uniform where barriers follow
    }
    float lo[STATS ? NV : 1][VEC], hi[STATS ? NV : 1][VEC];
    if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { lo[k][e] = __builtin_inff(); hi[k][e] = -__builtin_inff(); }
    }
#pragma unroll 1
    for (int it = 0; it < ROWS; ++it) {
        long long row = row0 + it;
        if (row >= a.R) row = a.R - 1;  // STATS only
        RowTile<TX, VEC, TEAM, NV> x;
        x.load(reinterpret_cast<const TX*>(a.x) + row * a.C, a.C, lane);
        float ss[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float v = to_f<TX>(x.v[k].e[e]);
                ss[0] += v * v;
            }
        }
        team_sum<TEAM, 1>(ss, lds);
        const float rstd = 1.0f / sqrtf(ss[0] / (float)a.C + a.eps);
        if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            const long long c0 = x.offset(k);
            // VEC == 4: one 16-byte load per vector and pack (the entry point checks the alignment)
            const float4 w4 = *reinterpret_cast<const float4*>(a.weight + c0);
            const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
            Pack<TX, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float y = wv[e] * (to_f<TX>(x.v[k].e[e]) * rstd);
                if (a.bias) y += bv[e];
                o.e[e] = from_f<TX>(y);
                if constexpr (STATS) {
                    const float stored = to_f<TX>(o.e[e]);
                    lo[k][e] = nmin(lo[k][e], stored);
                    hi[k][e] = nmax(hi[k][e], stored);
                }
            }
            st_pack<TX, VEC>(out + row * a.C + c0, o);
        }
    }
    if constexpr (STATS)
        publish_minmax<TEAM, NV, VEC>(sink, lo, hi, lane, a.C / VEC, (int)((row0 / ROWS) % kSlotReplicas), tag, lds);
}

template <typename TX, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_rmsnorm_bwd_rows(RmsArgs a, const TX* __restrict__ gy, const float* __restrict__ rstd, TX* __restrict__ dx) {
    static_assert(VEC == 4, "the weight vector is read as float4");
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= a.R) return;
    RowTile<TX, VEC, TEAM, NV> x, g;
    x.load(reinterpret_cast<const TX*>(a.x) + row * a.C, a.C, lane);
    g.load(gy + row * a.C, a.C, lane);
    const float rs = rstd[row];
    float du[NV][VEC];
    float dot[1] = {0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!x.has(k)) continue;
        const long long c0 = x.offset(k);
        const float4 w4 = *reinterpret_cast<const float4*>(a.weight + c0);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            du[k][e] = to_f<TX>(g.v[k].e[e]) * wv[e];
            dot[0] += du[k][e] * (to_f<TX>(x.v[k].e[e]) * rs);
        }
    }
    team_sum<TEAM, 1>(dot, lds);
    const float m = dot[0] / (float)a.C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!x.has(k)) continue;
        Pack<TX, VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.e[e] = from_f<TX>(rs * (du[k][e] - (to_f<TX>(x.v[k].e[e]) * rs) * m));
        st_pack<TX, VEC>(dx + row * a.C + x.offset(k), o);
    }
}

// partial[0][chunk][c] = sum over the chunk's rows of gy*u, partial[1][chunk][c] = sum of gy
template <typename TX>
__global__ void __launch_bounds__(256)
k_rmsnorm_bwd_cols(const TX* __restrict__ x, const TX* __restrict__ gy, const float* __restrict__ rstd, long long R, int C,
                   int rows_per_chunk, float* __restrict__ part, int want_bias) {
    __shared__ float4 sm[2][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * 4;
    const bool valid = col0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        for (long long row = r0 + wv; row < r1; row += 4) {
            const long long ra = __builtin_amdgcn_readfirstlane((int)row);
            const float rs = rstd[ra];  // wave-uniform address
            const Pack<TX, 4> xv = ld_pack<TX, 4>(x + ra * C + col0), gv = ld_pack<TX, 4>(gy + ra * C + col0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = to_f<TX>(gv.e[e]);
                aw[e] += g * (to_f<TX>(xv.e[e]) * rs);
                ab[e] += g;
            }
        }
    }
    sm[0][wv][lane] = make_float4(aw[0], aw[1], aw[2], aw[3]);
    sm[1][wv][lane] = make_float4(ab[0], ab[1], ab[2], ab[3]);
    __syncthreads();
    if (wv != 0 || !valid) return;
    auto fold = [&](int which) {
        const float4 p0 = sm[which][0][lane], p1 = sm[which][1][lane], p2 = sm[which][2][lane], p3 = sm[which][3][lane];
        return make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z),
                           (p0.w + p1.w) + (p2.w + p3.w));
    };
    const long long plane = (long long)gridDim.y * C;
    *reinterpret_cast<float4*>(part + (long long)blockIdx.y * C + col0) = fold(0);
    if (want_bias) *reinterpret_cast<float4*>(part + plane + (long long)blockIdx.y * C + col0) = fold(1);
}

// out[c] = sum over the nrows partial rows (fixed order); block = 64 columns x 4 waves
__global__ void __launch_bounds__(256)
k_rmsnorm_fold(const float* __restrict__ part, int nrows, int C, float* __restrict__ out) {
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

static inline int rms_chunks(long long R, long long C) {
    const long long tiles = (C / 4 + 63) / 64;
    long long want = 1024 / (tiles > 0 ? tiles : 1);
    const long long maxc = (R + 15) / 16;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename TX>
static void rms_fwd_launch(const RmsArgs& a, void* out, float* rstd, const RowStatSink& sink, hipStream_t stream) {
    constexpr int VEC = 4;  // 16-byte fp32 / 8-byte fp16 accesses, one tiling for both
    const int npack = a.C / VEC;
#define L(TEAM, NV)                                                                                                  \
    {                                                                                                                \
        constexpr int RPB = block_of<TEAM>() / TEAM;                                                                 \
        const long long per = (long long)RPB * (sink.ws ? kStatRows : 1); /* rows per block */                       \
        const dim3 grid((unsigned)((a.R + per - 1) / per)), block(block_of<TEAM>());                                 \
        if (sink.ws) hipLaunchKernelGGL((k_rmsnorm_fwd<TX, VEC, TEAM, NV, true>), grid, block, 0, stream, a, (TX*)out, rstd, sink); \
        else hipLaunchKernelGGL((k_rmsnorm_fwd<TX, VEC, TEAM, NV, false>), grid, block, 0, stream, a, (TX*)out, rstd, sink);        \
    }
    MIQ_ROW_DISPATCH(npack, L);
#undef L
}

template <typename TX>
static void rms_bwd_launch(const RmsArgs& a, const void* gy, const float* rstd, void* dx, float* d_weight, float* d_bias,
                           float* ws, hipStream_t stream) {
    constexpr int VEC = 4;
    const int npack = a.C / VEC;
    if (dx) {
#define L(TEAM, NV)                                                                                              \
    {                                                                                                            \
        constexpr int RPB = block_of<TEAM>() / TEAM;                                                             \
        hipLaunchKernelGGL((k_rmsnorm_bwd_rows<TX, VEC, TEAM, NV>), dim3((unsigned)((a.R + RPB - 1) / RPB)),     \
                           dim3(block_of<TEAM>()), 0, stream, a, (const TX*)gy, rstd, (TX*)dx);                  \
    }
        MIQ_ROW_DISPATCH(npack, L);
#undef L
    }
    if (d_weight || d_bias) {
        const int nchunk = rms_chunks(a.R, a.C);
        const int rpc = (int)((a.R + nchunk - 1) / nchunk);
        const unsigned tiles = (unsigned)((npack + 63) / 64);
        hipLaunchKernelGGL((k_rmsnorm_bwd_cols<TX>), dim3(tiles, nchunk), dim3(256), 0, stream, (const TX*)a.x, (const TX*)gy, rstd,
                           a.R, a.C, rpc, ws, d_bias ? 1 : 0);
        const unsigned fg = (unsigned)((a.C + 63) / 64);
        if (d_weight) hipLaunchKernelGGL(k_rmsnorm_fold, dim3(fg), dim3(256), 0, stream, ws, nchunk, a.C, d_weight);
        if (d_bias) hipLaunchKernelGGL(k_rmsnorm_fold, dim3(fg), dim3(256), 0, stream, ws + (long long)nchunk * a.C, nchunk, a.C, d_bias);
    }
}

}  // namespace miq

using namespace miq;

extern "C" int64_t miq_rmsnorm_workspace(int64_t R, int64_t C) {
    if (R <= 0 || C <= 0 || C % 4) return 0;
    return 2ll * rms_chunks(R, C) * C;
}

static int rmsnorm_fwd_checked(const char* who, const void* x, const float* weight, const float* bias, void* out, float* rstd_out,
                               int64_t R, int64_t C, float eps, int dtype, const RowStatSink& sink, void* stream) {
    MIQ_CHECK_ARG(x && weight && out && R > 0 && C > 0 && C < (1ll << 31), "%s: null pointer or bad shape", who);
    MIQ_CHECK_ARG(C % 4 == 0 && C / 4 <= kRowMaxPacks, "%s: C=%lld must be a multiple of 4 and at most %d", who, (long long)C,
                  4 * kRowMaxPacks);
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "%s: unsupported dtype %d", who, dtype);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(out) && is_aligned16(weight) && is_aligned16(bias),
                  "%s: x / out / weight / bias must be 16-byte aligned", who);
    RmsArgs a{x, weight, bias, eps, R, (int)C};
    if (dtype == MIQ_F32) rms_fwd_launch<float>(a, out, rstd_out, sink, (hipStream_t)stream);
    else rms_fwd_launch<__half>(a, out, rstd_out, sink, (hipStream_t)stream);
    MIQ_CHECK_LAUNCH(who);
    return 0;
}

extern "C" int miq_rmsnorm_fwd(const void* x, const float* weight, const float* bias, void* out, float* rstd_out, int64_t R,
                               int64_t C, float eps, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    return rmsnorm_fwd_checked("miq_rmsnorm_fwd", x, weight, bias, out, rstd_out, R, C, eps, dtype, RowStatSink{nullptr, 0, nullptr}, stream);
}

// the statistics sink of the *_fwd_stats entry points, checked: equal clusters (multiples of the 4-channel pack) that
// cover every channel go by arithmetic, everything else through the channel -> slot map
int miq_row_stat_sink(const char* who, int64_t cluster_size, const int* slot_of, int n_clusters, int64_t covered, int64_t C,
                      void* stat_workspace, RowStatSink* sink) {
    MIQ_CHECK_ARG(stat_workspace && (reinterpret_cast<uintptr_t>(stat_workspace) & 127u) == 0,
                  "%s: the statistics workspace must be 128-byte aligned", who);
    MIQ_CHECK_ARG(n_clusters >= 0 && covered >= 0 && covered <= C && (cluster_size <= 0 || cluster_size * n_clusters == covered),
                  "%s: bad cluster layout", who);
    const bool by_arithmetic = cluster_size > 0 && cluster_size % 4 == 0 && covered == C;
    MIQ_CHECK_ARG(by_arithmetic || slot_of, "%s: this cluster layout needs the channel -> slot map", who);
    *sink = RowStatSink{static_cast<u64*>(stat_workspace), by_arithmetic ? (int)(cluster_size / 4) : 0, by_arithmetic ? nullptr : slot_of};
    return 0;
}

extern "C" int miq_rmsnorm_fwd_stats(const void* x, const float* weight, const float* bias, void* out, float* rstd_out, int64_t R,
                                     int64_t C, float eps, int dtype, int64_t cluster_size, const int* slot_of, int n_clusters,
                                     int64_t covered, void* stat_workspace, void* stream) {
    if (R == 0 || C == 0) return 0;
    RowStatSink sink{};
    if (const int rc = miq_row_stat_sink("miq_rmsnorm_fwd_stats", cluster_size, slot_of, n_clusters, covered, C, stat_workspace, &sink))
        return rc;
    return rmsnorm_fwd_checked("miq_rmsnorm_fwd_stats", x, weight, bias, out, rstd_out, R, C, eps, dtype, sink, stream);
}

extern "C" int miq_rmsnorm_bwd(const void* grad_out, const void* x, const float* weight, const float* rstd, void* dx,
                               float* d_weight, float* d_bias, float* workspace, int64_t R, int64_t C, int dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(grad_out && x && weight && rstd && R > 0 && R < (1ll << 31) && C > 0 && C < (1ll << 31),
                  "miq_rmsnorm_bwd: null pointer or bad shape");
    MIQ_CHECK_ARG(C % 4 == 0 && C / 4 <= kRowMaxPacks, "miq_rmsnorm_bwd: C=%lld must be a multiple of 4 and at most %d", (long long)C,
                  4 * kRowMaxPacks);
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_rmsnorm_bwd: unsupported dtype %d", dtype);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(grad_out) && is_aligned16(dx) && is_aligned16(weight),
                  "miq_rmsnorm_bwd: x / grad_out / dx / weight must be 16-byte aligned");
    MIQ_CHECK_ARG((!d_weight && !d_bias) || (workspace && is_aligned16(workspace)), "miq_rmsnorm_bwd: workspace required for d_weight / d_bias");
    RmsArgs a{x, weight, nullptr, 0.f, R, (int)C};
    if (dtype == MIQ_F32) rms_bwd_launch<float>(a, grad_out, rstd, dx, d_weight, d_bias, workspace, (hipStream_t)stream);
    else rms_bwd_launch<__half>(a, grad_out, rstd, dx, d_weight, d_bias, workspace, (hipStream_t)stream);
    MIQ_CHECK_LAUNCH("miq_rmsnorm_bwd");
    return 0;
}
