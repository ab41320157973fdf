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
template <typename TX, typename TO, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_gather_ln_fwd(LnArgs a, TO* __restrict__ out, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const int team = threadIdx.x / TEAM;
    const long long row = (long long)blockIdx.x * RPB + team;
    const bool live = row < a.R;  // whole teams; all threads still reach the block barrier below
    const int npack = a.C / VEC;
    const TX* xrow = reinterpret_cast<const TX*>(a.x) + (live ? row : 0) * a.C;
    float v[NV][VEC];
    if (a.idx) {
        TX* stage = reinterpret_cast<TX*>(smem) + (long long)team * a.C;
        if (live) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int j = k * TEAM + lane;
                if (j < npack) st_pack<TX, VEC>(stage + (long long)j * VEC, ld_pack<TX, VEC>(xrow + (long long)j * VEC));
            }
        }
        __syncthreads();
        if (!live) return;
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
        }
        st_pack<TO, VEC>(orow + (long long)j * VEC, o);
    }
}

template <typename TX, typename TO>
static int launch_ln(const LnArgs& a, void* out, float* mean_out, float* rstd_out, hipStream_t stream) {
    // one pack = 4 elements for every dtype combination: 16-byte fp32 accesses, 8-byte fp16 accesses
    constexpr int VEC = 4;
    const int npack = a.C / VEC;
#define L(TEAM, NV)                                                                                                     \
    {                                                                                                                   \
        constexpr int B = block_of<TEAM>();                                                                             \
        constexpr int RPB = B / TEAM;                                                                                   \
        const unsigned grid = (unsigned)((a.R + RPB - 1) / RPB);                                                        \
        const size_t lds = a.idx ? (size_t)RPB * a.C * sizeof(TX) : 0;                                                  \
        if (lds > 65536)                                                                                                \
            (void)hipFuncSetAttribute((const void*)k_gather_ln_fwd<TX, TO, VEC, TEAM, NV>,                              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        hipLaunchKernelGGL((k_gather_ln_fwd<TX, TO, VEC, TEAM, NV>), dim3(grid), dim3(B), lds, stream, a, (TO*)out,     \
                           mean_out, rstd_out);                                                                         \
    }
    MIQ_ROW_DISPATCH(npack, L);
#undef L
    MIQ_CHECK_LAUNCH("miq_gather_layernorm_fwd");
    return 0;
}

}  // namespace miq

using namespace miq;

extern "C" int miq_gather_layernorm_fwd(const void* x, const int* idx, const void* weight, const void* bias, void* out,
                                        float* mean_out, float* rstd_out, int64_t R, int64_t C, float eps, int x_dtype,
                                        int w_dtype, int out_dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(x && out && R > 0 && C > 0 && C < (1ll << 31), "miq_gather_layernorm_fwd: null pointer or bad shape");
    MIQ_CHECK_ARG(C % 4 == 0 && C / 4 <= kRowMaxPacks, "miq_gather_layernorm_fwd: C=%lld must be a multiple of 4 and at most %d",
                  (long long)C, 4 * kRowMaxPacks);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(out), "miq_gather_layernorm_fwd: x / out must be 16-byte aligned");
    MIQ_CHECK_ARG((x_dtype == MIQ_F32 || x_dtype == MIQ_F16) && (w_dtype == MIQ_F32 || w_dtype == MIQ_F16) &&
                      (out_dtype == MIQ_F32 || out_dtype == MIQ_F16), "miq_gather_layernorm_fwd: unsupported dtype");
    MIQ_CHECK_ARG(!idx || C * (x_dtype == MIQ_F16 ? 2 : 4) <= 160 * 1024, "miq_gather_layernorm_fwd: a row does not fit in LDS");
    LnArgs a{x, idx, weight, bias, w_dtype == MIQ_F16, eps, R, (int)C};
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == MIQ_F32 && out_dtype == MIQ_F32) return launch_ln<float, float>(a, out, mean_out, rstd_out, s);
    if (x_dtype == MIQ_F16 && out_dtype == MIQ_F16) return launch_ln<__half, __half>(a, out, mean_out, rstd_out, s);
    if (x_dtype == MIQ_F16 && out_dtype == MIQ_F32) return launch_ln<__half, float>(a, out, mean_out, rstd_out, s);
    return launch_ln<float, __half>(a, out, mean_out, rstd_out, s);
}
