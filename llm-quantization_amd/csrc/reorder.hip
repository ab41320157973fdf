// RPTQ reorder index: inside every head, channels are ordered by ascending (max - min) range and cut
// into equal-size clusters (reference quantize/reorder_utils.py:111-147, which calls torch.argsort).
// The permutation is produced by a stable rank sort: rank(i) = #{j : r[j] < r[i] or (r[j] == r[i] and
// j < i)}, index[rank(i)] = i.  It is O(n^2) comparisons per head but n <= a few ten thousand and the
// ranges are staged in LDS, so the whole call takes microseconds; ties are broken by channel id
// (ascending), NaN ranges sort last (as torch.sort does).
#include "miq_common.h"

namespace miq {

// "a comes before b" for the ascending order with NaN last
__device__ __forceinline__ bool before(float a, int ia, float b, int ib) {
    const bool an = a != a, bn = b != b;
    if (an || bn) return (!an && bn) || (an == bn && ia < ib);
    return a < b || (a == b && ia < ib);
}

// grid: (blocks per head, n_heads); every block stages its head's ranges in LDS tiles
__global__ void __launch_bounds__(256)
k_reorder_index(const float* __restrict__ xmax, const float* __restrict__ xmin, int per_head, long long* __restrict__ index) {
    __shared__ float tile[1024];
    const int head = blockIdx.y;
    const long long off = (long long)head * per_head;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < per_head;
    const float ri = valid ? (xmax[off + i] - xmin[off + i]) : 0.f;
    int rank = 0;
    for (int t0 = 0; t0 < per_head; t0 += 1024) {
        __syncthreads();
        for (int j = threadIdx.x; j < 1024 && t0 + j < per_head; j += 256)
            tile[j] = xmax[off + t0 + j] - xmin[off + t0 + j];
        __syncthreads();
        const int n = per_head - t0 < 1024 ? per_head - t0 : 1024;
        if (valid)
            for (int j = 0; j < n; ++j) rank += before(tile[j], t0 + j, ri, i) ? 1 : 0;
    }
    if (valid) index[off + rank] = off + i;
}

}  // namespace miq

extern "C" int miq_reorder_index(const float* xmax, const float* xmin, int64_t C, int n_heads, int64_t* index_out, void* stream) {
    if (C == 0) return 0;
    MIQ_CHECK_ARG(xmax && xmin && index_out && C > 0 && n_heads >= 1 && C % n_heads == 0 && C < (1ll << 31),
                  "miq_reorder_index: null pointer, or C=%lld not divisible by n_heads=%d", (long long)C, n_heads);
    const int per_head = (int)(C / n_heads);
    dim3 grid((unsigned)((per_head + 255) / 256), (unsigned)n_heads);
    hipLaunchKernelGGL(miq::k_reorder_index, grid, dim3(256), 0, (hipStream_t)stream, xmax, xmin, per_head, (long long*)index_out);
    MIQ_CHECK_LAUNCH("miq_reorder_index");
    return 0;
}
