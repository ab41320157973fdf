// Row statistics used by the sensitivity pass that produces the per-output-channel bit allocation
// (reference get_grads.py:201-205): acc[o] += mean_i |g[o, i]| for a row-major [O, I] gradient.
// One wave per row, 16-byte loads, fp32 accumulation; reads g once (n*s bytes).
#include "miq_common.h"

namespace miq {

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_row_absmean_accum(const T* __restrict__ g, float* __restrict__ acc, long long O, long long I) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= O) return;
    const T* p = g + row * I;
    float s0 = 0.f, s1 = 0.f;
    long long i = (long long)lane * VEC;
    for (; i + 64 * VEC < I; i += 128 * VEC) {
        Pack<T, VEC> a = ld_pack<T, VEC>(p + i), b = ld_pack<T, VEC>(p + i + 64 * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            s0 += fabsf(to_f<T>(a.e[e]));
            s1 += fabsf(to_f<T>(b.e[e]));
        }
    }
    for (; i < I; i += 64 * VEC) {
        Pack<T, VEC> a = ld_pack<T, VEC>(p + i);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s0 += fabsf(to_f<T>(a.e[e]));
    }
    float v[1] = {s0 + s1};
    team_sum<64, 1>(v, nullptr);
    if (lane == 0) acc[row] += v[0] / (float)I;
}

template <typename T>
static int absmean_t(const void* g, float* acc, long long O, long long I, hipStream_t stream) {
    constexpr int N = VecOf<T>::N;
    const unsigned grid = (unsigned)((O + 3) / 4);
    if (I % N == 0 && (reinterpret_cast<uintptr_t>(g) & 15u) == 0)
        hipLaunchKernelGGL((k_row_absmean_accum<T, N>), dim3(grid), dim3(256), 0, stream, (const T*)g, acc, O, I);
    else
        hipLaunchKernelGGL((k_row_absmean_accum<T, 1>), dim3(grid), dim3(256), 0, stream, (const T*)g, acc, O, I);
    MIQ_CHECK_LAUNCH("miq_row_absmean_accum");
    return 0;
}

}  // namespace miq

using namespace miq;

extern "C" int miq_row_absmean_accum(const void* g, float* acc, int64_t O, int64_t I, int dtype, void* stream) {
    if (O == 0) return 0;
    MIQ_CHECK_ARG(g && acc && O > 0 && I > 0 && O < (1ll << 33), "miq_row_absmean_accum: null pointer or bad shape");
    if (dtype == MIQ_F32) return absmean_t<float>(g, acc, O, I, (hipStream_t)stream);
    if (dtype == MIQ_F16) return absmean_t<__half>(g, acc, O, I, (hipStream_t)stream);
    miq_set_error("miq_row_absmean_accum: unsupported dtype %d", dtype);
    return -22;
}
