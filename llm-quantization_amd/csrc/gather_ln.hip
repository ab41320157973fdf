// RPTQ reorder gather fused into LayerNorm (forward): what OmniLayerNorm.forward does with a
// reorder index (reference quantize/omni_norm.py:29-42):
//     x = torch.index_select(x, last_dim, reorder_index);  y = layer_norm(x, weight, bias, eps)
// in ONE pass over the activation: a team of threads owns one row, the row is brought into LDS with
// coalesced 16-byte loads, read back through the index into registers (the gathered row never
// exists in memory), mean and variance are reduced in fp32 over the registers (two-pass, exact
// centring), and the normalised row is written with coalesced 16-byte stores.  Also writes the
// per-row mean and reciprocal standard deviation (fp32), which is what a LayerNorm backward needs.
//
// HBM traffic: read x once + write y once (2ns) instead of gather (2ns) + LayerNorm (2ns).
// Input / output dtypes may differ (fp16 activations normalised to fp32 under autocast).
#include "miq_common.h"
#include "slot_table.h"

namespace miq {

struct LnArgs {
    const void* x;
    const int* idx;      // [C] or nullptr (plain LayerNorm)
    const void* weight;  // [C] or nullptr
    const void* bias;    // [C] or nullptr
    int w_half;          // weight / bias are fp16 (else fp32)
    float eps;
    long long R;
    int C;
};

__device__ __forceinline__ float ld_wb(const void* p, int half, int i) {
    return half ? __half2float(reinterpret_cast<const __half*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}

// One row per team.  Two variants were measured slower and dropped: a persistent form (512 blocks, several
// rows per team with the index held in registers: the per-row barriers serialise, 1.6x slower) and loading
// the index into registers ahead of the staging loads (1.4-2x slower for 11008-wide rows).
// STATS: the kernel also folds the minima / maxima of the values it stores (after rounding to TO) into the slot table of
// `sink`, which is what the statistics pass of the activation quantizer would compute from `out`.  A team then walks
// kStatRows consecutive rows, keeps the running extremes of its channels in registers and publishes once (a quarter of
// the exchanges and atomics of one row per team); a team past the last row repeats the last row (same stores, same
// statistics) so that every lane takes part in the exchange.
template <typename TX, typename TO, int VEC, int TEAM, int NV, bool STATS>
__global__ void __launch_bounds__(block_of<TEAM>())
k_gather_ln_fwd(LnArgs a, TO* __restrict__ out, float* __restrict__ mean_out, float* __restrict__ rstd_out, RowStatSink sink) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    constexpr int ROWS = STATS ? kStatRows : 1;
    const int lane = threadIdx.x % TEAM;
    const int team = threadIdx.x / TEAM;
    const long long row0 = ((long long)blockIdx.x * RPB + team) * ROWS;
    unsigned tag = 0;
    if (STATS) {
        tag = (unsigned)sink.ws[0] + 1u;
        if (blockIdx.x == 0 && threadIdx.x == 0) sink.ws[1] = tag;
    }
    const int npack = a.C / VEC;
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
        if (STATS && row >= a.R) row = a.R - 1;
        const bool live = row < a.R;  // whole teams; all threads still reach the block barriers below
        const TX* xrow = reinterpret_cast<const TX*>(a.x) + (live ? row : 0) * a.C;
        float v[NV][VEC];
        if (a.idx) {
            TX* stage = reinterpret_cast<TX*>(smem) + (long long)team * a.C;
            if (STATS && it > 0) __syncthreads();  // the previous row has been read out of the staging row
            if (live) {
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int j = k * TEAM + lane;
                    if (j < npack) st_pack<TX, VEC>(stage + (long long)j * VEC, ld_pack<TX, VEC>(xrow + (long long)j * VEC));
                }
            }
            __syncthreads();
            if (!live) return;  // never with STATS
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = k * TEAM + lane;
                if (j >= npack) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[k][e] = to_f<TX>(stage[a.idx[j * VEC + e]]);
            }
        } else {
            if (!live) return;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = k * TEAM + lane;
                if (j >= npack) continue;
                const Pack<TX, VEC> p = ld_pack<TX, VEC>(xrow + (long long)j * VEC);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[k][e] = to_f<TX>(p.e[e]);
            }
        }
        // TEAM > 64 implies RPB == 1, so the block-wide reductions below are reached by every thread of the block
        float s[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k * TEAM + lane >= npack) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[0] += v[k][e];
        }
        team_sum<TEAM, 1>(s, lds);
        const float mean = s[0] / (float)a.C;
        float q[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (k * TEAM + lane >= npack) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float d = v[k][e] - mean;
                q[0] += d * d;
            }
        }
        team_sum<TEAM, 1>(q, lds);
        const float rstd = 1.0f / sqrtf(q[0] / (float)a.C + a.eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
        TO* orow = out + row * a.C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            if (j >= npack) continue;
            Pack<TO, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int c = j * VEC + e;
                float y = (v[k][e] - mean) * rstd;
                if (a.weight) y *= ld_wb(a.weight, a.w_half, c);
                if (a.bias) y += ld_wb(a.bias, a.w_half, c);
                o.e[e] = from_f<TO>(y);
                if constexpr (STATS) {
                    const float stored = to_f<TO>(o.e[e]);
                    lo[k][e] = nmin(lo[k][e], stored);
                    hi[k][e] = nmax(hi[k][e], stored);
                }
            }
            st_pack<TO, VEC>(orow + (long long)j * VEC, o);
        }
    }
    if constexpr (STATS)
        publish_minmax<TEAM, NV, VEC>(sink, lo, hi, lane, npack, (int)((row0 / ROWS) % kSlotReplicas), tag, lds);
}

template <typename TX, typename TO>
static int launch_ln(const LnArgs& a, void* out, float* mean_out, float* rstd_out, const RowStatSink& sink, hipStream_t stream) {
    // one pack = 4 elements for every dtype combination: 16-byte fp32 accesses, 8-byte fp16 accesses
    constexpr int VEC = 4;
    const int npack = a.C / VEC;
#define L(TEAM, NV)                                                                                                     \
    {                                                                                                                   \
        constexpr int B = block_of<TEAM>();                                                                             \
        constexpr int RPB = B / TEAM;                                                                                   \
        const long long per = (long long)RPB * (sink.ws ? kStatRows : 1); /* rows per block */                          \
        const unsigned grid = (unsigned)((a.R + per - 1) / per);                                                        \
        const size_t lds = a.idx ? (size_t)RPB * a.C * sizeof(TX) : 0;                                                  \
        if (sink.ws) {                                                                                                  \
            if (lds > 65536)                                                                                            \
                (void)hipFuncSetAttribute((const void*)k_gather_ln_fwd<TX, TO, VEC, TEAM, NV, true>,                    \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
            hipLaunchKernelGGL((k_gather_ln_fwd<TX, TO, VEC, TEAM, NV, true>), dim3(grid), dim3(B), lds, stream, a,     \
                               (TO*)out, mean_out, rstd_out, sink);                                                     \
        } else {                                                                                                        \
            if (lds > 65536)                                                                                            \
                (void)hipFuncSetAttribute((const void*)k_gather_ln_fwd<TX, TO, VEC, TEAM, NV, false>,                   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                        \
            hipLaunchKernelGGL((k_gather_ln_fwd<TX, TO, VEC, TEAM, NV, false>), dim3(grid), dim3(B), lds, stream, a,    \
                               (TO*)out, mean_out, rstd_out, sink);                                                     \
        }                                                                                                               \
    }
    MIQ_ROW_DISPATCH(npack, L);
#undef L
    MIQ_CHECK_LAUNCH("miq_gather_layernorm_fwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Backward of gather + LayerNorm.  With xg[c] = x[idx[c]], u = (xg - mean) * rstd, gw = gy * weight:
//     dxg = rstd * (gw - mean_c(gw) - u * mean_c(gw * u)),   dx[idx[c]] = dxg[c]
//     d_weight[c] = sum_rows gy * u,   d_bias[c] = sum_rows gy
// Rows kernel: one team per row as in the forward; the row of x is staged in LDS and read through the index, the
// gradient row is written back into the same LDS row AT idx[c] (the scatter happens in LDS) and leaves with coalesced
// 16-byte stores.  Columns kernel: lane = 4 channels, wave = every 4th row of a row chunk, per-chunk partial sums and
// a deterministic fold (x is read through the index: 4-byte accesses inside one row, served by the caches).
// ---------------------------------------------------------------------------------------------
struct LnBwdArgs {
    const void* x;
    const int* idx;      // [C] or nullptr
    const void* weight;  // [C] or nullptr
    int w_half;
    const float* mean;   // [R]
    const float* rstd;   // [R]
    long long R;
    int C;
};

template <typename TX, typename TG, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_gather_ln_bwd_rows(LnBwdArgs a, const TG* __restrict__ gy, TX* __restrict__ dx) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const int team = threadIdx.x / TEAM;
    const long long row = (long long)blockIdx.x * RPB + team;
    const bool live = row < a.R;  // whole teams; dead teams still reach every block barrier
    const long long rrow = live ? row : 0;
    const int npack = a.C / VEC;
    const TX* xrow = reinterpret_cast<const TX*>(a.x) + rrow * a.C;
    TX* stage = reinterpret_cast<TX*>(smem) + (long long)team * a.C;
    float v[NV][VEC], gw[NV][VEC];
    if (a.idx) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            if (j < npack) st_pack<TX, VEC>(stage + (long long)j * VEC, ld_pack<TX, VEC>(xrow + (long long)j * VEC));
        }
        __syncthreads();
    }
    const float mu = a.mean[rrow], rs = a.rstd[rrow];
    float acc[2] = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int j = k * TEAM + lane;
        if (j >= npack) continue;
        const Pack<TG, VEC> g = ld_pack<TG, VEC>(gy + rrow * a.C + (long long)j * VEC);
        Pack<TX, VEC> xp;
        if (!a.idx) xp = ld_pack<TX, VEC>(xrow + (long long)j * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = j * VEC + e;
            const float xv = a.idx ? to_f<TX>(stage[a.idx[c]]) : to_f<TX>(xp.e[e]);
            v[k][e] = (xv - mu) * rs;
            gw[k][e] = to_f<TG>(g.e[e]) * (a.weight ? ld_wb(a.weight, a.w_half, c) : 1.f);
            acc[0] += gw[k][e];
            acc[1] += gw[k][e] * v[k][e];
        }
    }
    team_sum<TEAM, 2>(acc, lds);  // TEAM > 64: one row per block, every thread is here; block barriers inside
    const float m1 = acc[0] / (float)a.C, m2 = acc[1] / (float)a.C;
    if (a.idx) {
        __syncthreads();  // every team has read its staged row
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            if (j >= npack) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) stage[a.idx[j * VEC + e]] = from_f<TX>(rs * (gw[k][e] - m1 - v[k][e] * m2));
        }
        __syncthreads();
        if (!live) return;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            if (j < npack) st_pack<TX, VEC>(dx + row * a.C + (long long)j * VEC, ld_pack<TX, VEC>(stage + (long long)j * VEC));
        }
    } else {
        if (!live) return;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int j = k * TEAM + lane;
            if (j >= npack) continue;
            Pack<TX, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.e[e] = from_f<TX>(rs * (gw[k][e] - m1 - v[k][e] * m2));
            st_pack<TX, VEC>(dx + row * a.C + (long long)j * VEC, o);
        }
    }
}

// part[0][chunk][c] = sum over the chunk's rows of gy * u, part[1][chunk][c] = sum of gy
template <typename TX, typename TG>
__global__ void __launch_bounds__(256)
k_gather_ln_bwd_cols(LnBwdArgs a, const TG* __restrict__ gy, int rows_per_chunk, float* __restrict__ part, int want_bias) {
    __shared__ float4 sm[2][4][64];
    const TX* __restrict__ x = reinterpret_cast<const TX*>(a.x);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * 4;
    const bool valid = col0 < a.C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > a.R) r1 = a.R;
    float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        int src[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) src[e] = a.idx ? a.idx[col0 + e] : (int)(col0 + e);
        for (long long row = r0 + wv; row < r1; row += 4) {
            const long long ra = __builtin_amdgcn_readfirstlane((int)row);
            const float rs = a.rstd[ra], mu = a.mean[ra];  // wave-uniform addresses
            const Pack<TG, 4> gv = ld_pack<TG, 4>(gy + ra * a.C + col0);
            float xv[4];
            if (a.idx) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = to_f<TX>(x[ra * a.C + src[e]]);
            } else {
                const Pack<TX, 4> xp = ld_pack<TX, 4>(x + ra * a.C + col0);
#pragma unroll
                for (int e = 0; e < 4; ++e) xv[e] = to_f<TX>(xp.e[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = to_f<TG>(gv.e[e]);
                aw[e] += g * ((xv[e] - mu) * rs);
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
    const long long plane = (long long)gridDim.y * a.C;
    *reinterpret_cast<float4*>(part + (long long)blockIdx.y * a.C + col0) = fold(0);
    if (want_bias) *reinterpret_cast<float4*>(part + plane + (long long)blockIdx.y * a.C + col0) = fold(1);
}

// out[c] = sum over the nrows partial rows (fixed order); block = 64 columns x 4 waves
__global__ void __launch_bounds__(256)
k_gather_ln_fold(const float* __restrict__ part, int nrows, int C, float* __restrict__ out) {
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

static inline int ln_chunks(long long R, long long C) {
    const long long tiles = (C / 4 + 63) / 64;
    long long want = 1024 / (tiles > 0 ? tiles : 1);
    const long long maxc = (R + 15) / 16;
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename TX, typename TG>
static int launch_ln_bwd(const LnBwdArgs& a, const void* gy, void* dx, float* d_weight, float* d_bias, float* ws, hipStream_t stream) {
    constexpr int VEC = 4;
    const int npack = a.C / VEC;
    if (dx) {
#define L(TEAM, NV)                                                                                                     \
    {                                                                                                                   \
        constexpr int B = block_of<TEAM>();                                                                             \
        constexpr int RPB = B / TEAM;                                                                                   \
        const unsigned grid = (unsigned)((a.R + RPB - 1) / RPB);                                                        \
        const size_t lds = a.idx ? (size_t)RPB * a.C * sizeof(TX) : 0;                                                  \
        if (lds > 65536)                                                                                                \
            (void)hipFuncSetAttribute((const void*)k_gather_ln_bwd_rows<TX, TG, VEC, TEAM, NV>,                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        hipLaunchKernelGGL((k_gather_ln_bwd_rows<TX, TG, VEC, TEAM, NV>), dim3(grid), dim3(B), lds, stream, a,          \
                           (const TG*)gy, (TX*)dx);                                                                     \
    }
        MIQ_ROW_DISPATCH(npack, L);
#undef L
    }
    if (d_weight || d_bias) {
        const int nchunk = ln_chunks(a.R, a.C);
        const int rpc = (int)((a.R + nchunk - 1) / nchunk);
        const unsigned tiles = (unsigned)((npack + 63) / 64);
        hipLaunchKernelGGL((k_gather_ln_bwd_cols<TX, TG>), dim3(tiles, nchunk), dim3(256), 0, stream, a, (const TG*)gy, rpc, ws,
                           d_bias ? 1 : 0);
        const unsigned fg = (unsigned)((a.C + 63) / 64);
        if (d_weight) hipLaunchKernelGGL(k_gather_ln_fold, dim3(fg), dim3(256), 0, stream, ws, nchunk, a.C, d_weight);
        if (d_bias) hipLaunchKernelGGL(k_gather_ln_fold, dim3(fg), dim3(256), 0, stream, ws + (long long)nchunk * a.C, nchunk, a.C, d_bias);
    }
    MIQ_CHECK_LAUNCH("miq_gather_layernorm_bwd");
    return 0;
}

}  // namespace miq

using namespace miq;

static int gather_ln_checked(const char* who, const void* x, const int* idx, const void* weight, const void* bias, void* out,
                             float* mean_out, float* rstd_out, int64_t R, int64_t C, float eps, int x_dtype, int w_dtype,
                             int out_dtype, const RowStatSink& sink, void* stream) {
    MIQ_CHECK_ARG(x && out && R > 0 && C > 0 && C < (1ll << 31), "%s: null pointer or bad shape", who);
    MIQ_CHECK_ARG(C % 4 == 0 && C / 4 <= kRowMaxPacks, "%s: C=%lld must be a multiple of 4 and at most %d", who, (long long)C,
                  4 * kRowMaxPacks);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(out), "%s: x / out must be 16-byte aligned", who);
    MIQ_CHECK_ARG((x_dtype == MIQ_F32 || x_dtype == MIQ_F16) && (w_dtype == MIQ_F32 || w_dtype == MIQ_F16) &&
                      (out_dtype == MIQ_F32 || out_dtype == MIQ_F16), "%s: unsupported dtype", who);
    MIQ_CHECK_ARG(!idx || C * (x_dtype == MIQ_F16 ? 2 : 4) <= 160 * 1024, "%s: a row does not fit in LDS", who);
    LnArgs a{x, idx, weight, bias, w_dtype == MIQ_F16, eps, R, (int)C};
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MIQ_F32 && out_dtype == MIQ_F32) return launch_ln<float, float>(a, out, mean_out, rstd_out, sink, s);
    if (x_dtype == MIQ_F16 && out_dtype == MIQ_F16) return launch_ln<__half, __half>(a, out, mean_out, rstd_out, sink, s);
    if (x_dtype == MIQ_F16 && out_dtype == MIQ_F32) return launch_ln<__half, float>(a, out, mean_out, rstd_out, sink, s);
    return launch_ln<float, __half>(a, out, mean_out, rstd_out, sink, s);
}

extern "C" int miq_gather_layernorm_fwd(const void* x, const int* idx, const void* weight, const void* bias, void* out,
                                        float* mean_out, float* rstd_out, int64_t R, int64_t C, float eps, int x_dtype,
                                        int w_dtype, int out_dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    return gather_ln_checked("miq_gather_layernorm_fwd", x, idx, weight, bias, out, mean_out, rstd_out, R, C, eps, x_dtype, w_dtype,
                             out_dtype, RowStatSink{nullptr, 0, nullptr}, stream);
}

int miq_row_stat_sink(const char* who, int64_t cluster_size, const int* slot_of, int n_clusters, int64_t covered, int64_t C,
                      void* stat_workspace, miq::RowStatSink* sink);  // rmsnorm.hip

extern "C" int miq_gather_layernorm_fwd_stats(const void* x, const int* idx, const void* weight, const void* bias, void* out,
                                              float* mean_out, float* rstd_out, int64_t R, int64_t C, float eps, int x_dtype,
                                              int w_dtype, int out_dtype, int64_t cluster_size, const int* slot_of, int n_clusters,
                                              int64_t covered, void* stat_workspace, void* stream) {
    if (R == 0 || C == 0) return 0;
    RowStatSink sink{};
    if (const int rc = miq_row_stat_sink("miq_gather_layernorm_fwd_stats", cluster_size, slot_of, n_clusters, covered, C,
                                         stat_workspace, &sink))
        return rc;
    return gather_ln_checked("miq_gather_layernorm_fwd_stats", x, idx, weight, bias, out, mean_out, rstd_out, R, C, eps, x_dtype,
                             w_dtype, out_dtype, sink, stream);
}

extern "C" int64_t miq_gather_layernorm_bwd_workspace(int64_t R, int64_t C) {
    if (R <= 0 || C <= 0 || C % 4) return 0;
    return 2ll * ln_chunks(R, C) * C;
}

extern "C" int miq_gather_layernorm_bwd(const void* grad_out, const void* x, const int* idx, const void* weight, const float* mean,
                                        const float* rstd, void* dx, float* d_weight, float* d_bias, float* workspace, int64_t R,
                                        int64_t C, int x_dtype, int w_dtype, int g_dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(grad_out && x && mean && rstd && R > 0 && R < (1ll << 31) && C > 0 && C < (1ll << 31),
                  "miq_gather_layernorm_bwd: null pointer or bad shape");
    MIQ_CHECK_ARG(C % 4 == 0 && C / 4 <= kRowMaxPacks, "miq_gather_layernorm_bwd: C=%lld must be a multiple of 4 and at most %d",
                  (long long)C, 4 * kRowMaxPacks);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(grad_out) && is_aligned16(dx), "miq_gather_layernorm_bwd: x / grad_out / dx must be 16-byte aligned");
    MIQ_CHECK_ARG((x_dtype == MIQ_F32 || x_dtype == MIQ_F16) && (w_dtype == MIQ_F32 || w_dtype == MIQ_F16) &&
                      (g_dtype == MIQ_F32 || g_dtype == MIQ_F16), "miq_gather_layernorm_bwd: unsupported dtype");
    MIQ_CHECK_ARG(!idx || C * (x_dtype == MIQ_F16 ? 2 : 4) <= 160 * 1024, "miq_gather_layernorm_bwd: a row does not fit in LDS");
    MIQ_CHECK_ARG((!d_weight && !d_bias) || (workspace && is_aligned16(workspace)), "miq_gather_layernorm_bwd: workspace required for d_weight / d_bias");
    LnBwdArgs a{x, idx, weight, w_dtype == MIQ_F16, mean, rstd, R, (int)C};
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MIQ_F32 && g_dtype == MIQ_F32) return launch_ln_bwd<float, float>(a, grad_out, dx, d_weight, d_bias, workspace, s);
    if (x_dtype == MIQ_F16 && g_dtype == MIQ_F16) return launch_ln_bwd<__half, __half>(a, grad_out, dx, d_weight, d_bias, workspace, s);
    if (x_dtype == MIQ_F16 && g_dtype == MIQ_F32) return launch_ln_bwd<__half, float>(a, grad_out, dx, d_weight, d_bias, workspace, s);
    return launch_ln_bwd<float, __half>(a, grad_out, dx, d_weight, d_bias, workspace, s);
}

