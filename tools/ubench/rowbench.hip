// Development micro-benchmark: what bounds the row-wise forward kernel?  Variants of one kernel on a
// working set larger than the Infinity Cache, timed with hipEvents in one process.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#include "../../llm-quantization_amd/csrc/miq_common.h"
using namespace miq;
extern "C" void miq_set_error(const char*, ...) {}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// V0: plain copy
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ in, float4* __restrict__ out, long long n) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long stride = (long long)gridDim.x * 256;
    for (; i < n; i += stride) out[i] = in[i];
}
// V0b: copy, one row tile per block like the row kernels (no loop)
template <int NV>
__global__ void __launch_bounds__(256) k_copy_rows(const float* __restrict__ in, float* __restrict__ out, int I) {
    const long long base = (long long)blockIdx.x * I;
    RowTile<float, 4, 256, NV> x;
    x.load(in + base, I, threadIdx.x);
#pragma unroll
    for (int k = 0; k < NV; ++k) if (x.has(k)) st_pack<float, 4>(out + base + x.offset(k), x.v[k]);
}
// V1: row tile + min/max reduction + multiply
template <int NV, int MODE>
__global__ void __launch_bounds__(256) k_rows(const float* __restrict__ in, float* __restrict__ out, int I) {
    __shared__ float lds[32];
    const long long base = (long long)blockIdx.x * I;
    RowTile<float, 4, 256, NV> x;
    x.load(in + base, I, threadIdx.x);
    float rmin, rmax;
    row_minmax(x, false, lds, rmin, rmax);
    const float s = (rmax - rmin) / 15.f;
    const float z = rintf(clampf(-rmin / s, -1e4f, 1e4f));
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!x.has(k)) continue;
        Pack<float, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = x.v[k].e[e];
            if (MODE == 0) o.e[e] = v * s;                      // reduction only
            else if (MODE == 1) {                               // full math, IEEE division
                const float xi = rint_ste(v / s) + z;
                o.e[e] = (clampf(xi, 0.f, 15.f) - z) * s;
            } else {                                            // full math, reciprocal multiply
                const float xi = rint_ste(v * (1.0f / s)) + z;
                o.e[e] = (clampf(xi, 0.f, 15.f) - z) * s;
            }
        }
        st_pack<float, 4>(out + base + x.offset(k), o);
    }
}

int main(int argc, char** argv) {
    const int O = argc > 1 ? atoi(argv[1]) : 4096, I = 4096, NBUF = argc > 2 ? atoi(argv[2]) : 6;
    const long long n = (long long)O * I;
    std::vector<float*> in(NBUF), out(NBUF);
    for (int b = 0; b < NBUF; ++b) { CK(hipMalloc(&in[b], n * 4)); CK(hipMalloc(&out[b], n * 4)); }
    std::vector<float> h(n);
    for (long long i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    for (int b = 0; b < NBUF; ++b) CK(hipMemcpy(in[b], h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        std::vector<float> ts;
        for (int rep = 0; rep < 7; ++rep) {
            CK(hipEventRecord(e0));
            for (int b = 0; b < NBUF; ++b) launch(in[b], out[b]);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / NBUF);
        }
        std::sort(ts.begin(), ts.end());
        printf("%-28s %8.2f us  %7.1f GB/s (min %.2f us)\n", name, ts[3] * 1e3, 2.0 * n * 4 / (ts[3] * 1e-3) / 1e9, ts[0] * 1e3);
    };
    for (int round = 0; round < 2; ++round) {
        run("copy grid-stride 2048 blk", [&](float* a, float* b) { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });
        run("copy grid-stride 8192 blk", [&](float* a, float* b) { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 4); });
        run("copy row tiles", [&](float* a, float* b) { hipLaunchKernelGGL((k_copy_rows<4>), dim3(O), dim3(256), 0, 0, a, b, I); });
        run("row minmax + mul", [&](float* a, float* b) { hipLaunchKernelGGL((k_rows<4, 0>), dim3(O), dim3(256), 0, 0, a, b, I); });
        run("row quant (IEEE div)", [&](float* a, float* b) { hipLaunchKernelGGL((k_rows<4, 1>), dim3(O), dim3(256), 0, 0, a, b, I); });
        run("row quant (rcp mul)", [&](float* a, float* b) { hipLaunchKernelGGL((k_rows<4, 2>), dim3(O), dim3(256), 0, 0, a, b, I); });
    }
    return 0;
}
