// truncate_number of the reference (models/transformation.py:4-20, applied to every smoothing scale before each
// training step: models/int_llama_layer.py:279-285, models/int_opt_layer.py:390-396):
//     x[|x| < threshold] = sign(x) * threshold            (zeros stay zero: sign(0) = 0; NaN stays NaN)
// in place for up to 16 vectors in ONE launch.  As PyTorch ops this is abs, lt, sign, mul, where and a copy per
// vector: 24 launches per step for the four smoothing scales of a block, which is a visible share of a step that is
// bound by launches (small models).
#include "miq_common.h"

namespace miq {

constexpr int kMaxTruncate = 16;

struct TruncArgs {
    void* p[kMaxTruncate];
    long long n[kMaxTruncate];
    float threshold;
};

template <typename T>
__global__ void __launch_bounds__(256) k_truncate_inplace(TruncArgs a) {
    T* p = reinterpret_cast<T*>(a.p[blockIdx.y]);
    const long long n = a.n[blockIdx.y];
    const float thr = rnd<T>(a.threshold);  // the comparison and the product run in the tensor's dtype
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float v = to_f<T>(p[i]);
        if (fabsf(v) < thr) p[i] = from_f<T>(sgnf(v) * thr);
    }
}

}  // namespace miq

using namespace miq;

extern "C" int miq_truncate_inplace(void* const* ptrs, const int64_t* sizes, int n, float threshold, int dtype, void* stream) {
    if (n == 0) return 0;
    MIQ_CHECK_ARG(ptrs && sizes && n > 0 && n <= kMaxTruncate, "miq_truncate_inplace: 1..%d vectors per call", kMaxTruncate);
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_truncate_inplace: unsupported dtype %d", dtype);
    TruncArgs a;
    long long longest = 0;
    for (int i = 0; i < n; ++i) {
        MIQ_CHECK_ARG(sizes[i] >= 0 && (ptrs[i] || sizes[i] == 0), "miq_truncate_inplace: null pointer or negative size");
        a.p[i] = ptrs[i];
        a.n[i] = sizes[i];
        if (sizes[i] > longest) longest = sizes[i];
    }
    if (longest == 0) return 0;
    a.threshold = threshold;
    long long bx = (longest + 255) / 256;
    if (bx > 1024) bx = 1024;
    const dim3 grid((unsigned)bx, (unsigned)n);
    if (dtype == MIQ_F32) hipLaunchKernelGGL(k_truncate_inplace<float>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_truncate_inplace<__half>, grid, dim3(256), 0, (hipStream_t)stream, a);
    MIQ_CHECK_LAUNCH("miq_truncate_inplace");
    return 0;
}
