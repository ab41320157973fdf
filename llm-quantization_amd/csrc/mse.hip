// Mean squared error between the block's output and its full-precision target under fp16 autocast
// (reference quantize/omniquant.py:339, 511-513: loss_func = torch.nn.MSELoss() inside the autocast context): autocast
// casts both fp16 tensors to fp32, ATen squares the difference, a reduction takes the mean, and the backward runs
// mse_loss_backward plus a cast of the gradient back to fp16: six launches over [tokens, hidden] tensors per step.
// Here: one streaming pass + a one-block fold forward, one pass backward, reading the fp16 tensors directly.
//   forward   loss = mean((float(a) - float(b))^2)      (fp32; summation order differs from ATen's reduction)
//   backward  d = T( ((2/n) * (float(a) - float(b))) * g )   for the tensor in the `a` position, sign flipped for `b`:
//             ATen's  norm * (input - target) * grad_output  in fp32, rounded to the tensor dtype by the cast that
//             autocast's backward applies: bit-identical gradients.
#include "miq_common.h"

namespace miq {

constexpr int kMseBlocks = 1024;

template <typename T, int VEC>
__global__ void __launch_bounds__(256) k_mse_partial(const T* __restrict__ a, const T* __restrict__ b, long long n, float* __restrict__ part) {
    __shared__ float lds[16];
    const long long npack = n / VEC;
    float acc[1] = {0.f};
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npack; p += (long long)gridDim.x * 256) {
        const Pack<T, VEC> x = ld_pack_nt<T, VEC>(a + p * VEC), y = ld_pack_nt<T, VEC>(b + p * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float d = to_f<T>(x.e[e]) - to_f<T>(y.e[e]);
            acc[0] += d * d;
        }
    }
    if (blockIdx.x == 0) {  // tail elements
        for (long long i = npack * VEC + threadIdx.x; i < n; i += 256) {
            const float d = to_f<T>(a[i]) - to_f<T>(b[i]);
            acc[0] += d * d;
        }
    }
    team_sum<256, 1>(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}

__global__ void __launch_bounds__(256) k_mse_fold(const float* __restrict__ part, int nblk, float inv_n, float* __restrict__ out) {
    __shared__ float lds[16];
    float acc[1] = {0.f};
    for (int i = threadIdx.x; i < nblk; i += 256) acc[0] += part[i];
    team_sum<256, 1>(acc, lds);
    if (threadIdx.x == 0) out[0] = acc[0] * inv_n;
}

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_mse_bwd(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ g, float norm, long long n, T* __restrict__ da,
          T* __restrict__ db) {
    const float gv = g[0];
    const long long npack = n / VEC;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npack; p += (long long)gridDim.x * 256) {
        const Pack<T, VEC> x = ld_pack_nt<T, VEC>(a + p * VEC), y = ld_pack_nt<T, VEC>(b + p * VEC);
        Pack<T, VEC> oa, ob;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xa = to_f<T>(x.e[e]), xb = to_f<T>(y.e[e]);
            oa.e[e] = from_f<T>((norm * (xa - xb)) * gv);
            ob.e[e] = from_f<T>((norm * (xb - xa)) * gv);
        }
        if (da) st_pack<T, VEC>(da + p * VEC, oa);
        if (db) st_pack<T, VEC>(db + p * VEC, ob);
    }
    if (blockIdx.x == 0) {
        for (long long i = npack * VEC + threadIdx.x; i < n; i += 256) {
            const float xa = to_f<T>(a[i]), xb = to_f<T>(b[i]);
            if (da) da[i] = from_f<T>((norm * (xa - xb)) * gv);
            if (db) db[i] = from_f<T>((norm * (xb - xa)) * gv);
        }
    }
}

static inline int mse_blocks(long long n, int vec) {
    long long nb = (n / vec + 255) / 256;
    if (nb < 1) nb = 1;
    return (int)(nb > kMseBlocks ? kMseBlocks : nb);
}

}  // namespace miq

using namespace miq;

extern "C" int64_t miq_mse_workspace(void) { return kMseBlocks; }

extern "C" int miq_mse_fwd(const void* a, const void* b, float* out, float* workspace, int64_t n, int dtype, void* stream) {
    MIQ_CHECK_ARG(a && b && out && workspace && n > 0, "miq_mse_fwd: null pointer or empty input");
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_mse_fwd: unsupported dtype %d", dtype);
    MIQ_CHECK_ARG(is_aligned16(a) && is_aligned16(b), "miq_mse_fwd: inputs must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float inv_n = (float)(1.0 / (double)n);
    int nb;
    if (dtype == MIQ_F32) {
        nb = mse_blocks(n, 4);
        hipLaunchKernelGGL((k_mse_partial<float, 4>), dim3(nb), dim3(256), 0, s, (const float*)a, (const float*)b, (long long)n, workspace);
    } else {
        nb = mse_blocks(n, 8);
        hipLaunchKernelGGL((k_mse_partial<__half, 8>), dim3(nb), dim3(256), 0, s, (const __half*)a, (const __half*)b, (long long)n, workspace);
    }
    hipLaunchKernelGGL(k_mse_fold, dim3(1), dim3(256), 0, s, workspace, nb, inv_n, out);
    MIQ_CHECK_LAUNCH("miq_mse_fwd");
    return 0;
}

extern "C" int miq_mse_bwd(const void* a, const void* b, const float* grad_loss, void* da, void* db, int64_t n, int dtype, void* stream) {
    MIQ_CHECK_ARG(a && b && grad_loss && (da || db) && n > 0, "miq_mse_bwd: null pointer or empty input");
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_mse_bwd: unsupported dtype %d", dtype);
    MIQ_CHECK_ARG(is_aligned16(a) && is_aligned16(b) && is_aligned16(da) && is_aligned16(db), "miq_mse_bwd: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float norm = (float)(2.0 / (double)n);
    if (dtype == MIQ_F32)
        hipLaunchKernelGGL((k_mse_bwd<float, 4>), dim3(mse_blocks(n, 4) * 2), dim3(256), 0, s, (const float*)a, (const float*)b, grad_loss, norm,
                           (long long)n, (float*)da, (float*)db);
    else
        hipLaunchKernelGGL((k_mse_bwd<__half, 8>), dim3(mse_blocks(n, 8) * 2), dim3(256), 0, s, (const __half*)a, (const __half*)b, grad_loss,
                           norm, (long long)n, (__half*)da, (__half*)db);
    MIQ_CHECK_LAUNCH("miq_mse_bwd");
    return 0;
}
