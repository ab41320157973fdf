// Development micro-benchmark: what a read-only reduction can reach on slabs of the activation sizes (33.5 / 100 MB)
// when the data is cold (buffers rotate through more than the Infinity Cache holds).  Gives the ceiling for the
// statistics kernels.  Replayed from a HIP graph, timed with hipEvents.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 readbw.hip -o readbw
#include <hip/hip_runtime.h>
#include <algorithm>
#include <functional>
#include <vector>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
static hipStream_t g_stream;

static float time_graph(const std::function<void()>& enqueue, int reps = 9) {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(g_stream, hipStreamCaptureModeGlobal));
    enqueue();
    CK(hipStreamEndCapture(g_stream, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int i = 0; i < reps + 2; ++i) {
        CK(hipEventRecord(e0, g_stream));
        CK(hipGraphLaunch(exec, g_stream));
        CK(hipEventRecord(e1, g_stream));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    return ts[ts.size() / 2];
}

// every block reads one contiguous span of `per_block` float4, DEPTH loads in flight per lane
template <int DEPTH>
__global__ __launch_bounds__(256) void k_span(const float4* __restrict__ x, long long per_block, float* out) {
    const float4* p = x + (long long)blockIdx.x * per_block;
    float lo = 3e38f, hi = -3e38f;
    for (long long i = threadIdx.x; i < per_block; i += 256ll * DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = p[min(i + 256ll * d, per_block - 1)];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            lo = fminf(lo, fminf(fminf(v[d].x, v[d].y), fminf(v[d].z, v[d].w)));
            hi = fmaxf(hi, fmaxf(fmaxf(v[d].x, v[d].y), fmaxf(v[d].z, v[d].w)));
        }
    }
    if (lo > hi) out[blockIdx.x] = lo;  // never true for real data: keeps the loads alive
}

// column-tile geometry of the statistics kernels: block = 4 waves, wave w reads rows r0+w, r0+w+4, ... of a 256-column tile
template <int DEPTH>
__global__ __launch_bounds__(256) void k_tile(const float4* __restrict__ x, int R, int C4, int rows_per_chunk, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
    float lo = 3e38f, hi = -3e38f;
    for (int r = r0 + wave; r < r1; r += 4 * DEPTH) {
        float4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = x[(long long)min(r + 4 * d, r1 - 1) * C4 + col];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            lo = fminf(lo, fminf(fminf(v[d].x, v[d].y), fminf(v[d].z, v[d].w)));
            hi = fmaxf(hi, fmaxf(fmaxf(v[d].x, v[d].y), fmaxf(v[d].z, v[d].w)));
        }
    }
    if (lo > hi) out[blockIdx.x] = lo;
}

__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ x, float4* __restrict__ y, long long n4) {
    const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    float4 v[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) v[d] = x[min(i + 256 * d, n4 - 1)];
#pragma unroll
    for (int d = 0; d < 4; ++d) if (i + 256 * d < n4) y[i + 256 * d] = v[d];
}

int main() {
    CK(hipStreamCreate(&g_stream));
    const int R = 2048, C = 4096;
    const long long n4 = (long long)R * C / 4;  // one 33.5 MB slab
    const int nbuf = 12;                        // 400 MB in rotation
    std::vector<float4*> x(nbuf);
    float* out; CK(hipMalloc(&out, 1 << 20));
    float4* y; CK(hipMalloc(&y, 3 * n4 * 16));
    for (int b = 0; b < nbuf; ++b) { CK(hipMalloc(&x[b], n4 * 16)); CK(hipMemset(x[b], 0x3c, n4 * 16)); }
    // three consecutive slabs as one 100 MB read: allocate contiguous triple buffers too
    std::vector<float4*> x3(4);
    for (int b = 0; b < 4; ++b) { CK(hipMalloc(&x3[b], 3 * n4 * 16)); CK(hipMemset(x3[b], 0x3c, 3 * n4 * 16)); }
    const double mb = n4 * 16 / 1e6;
    printf("slab %.1f MB\n", mb);
    for (int blocks : {512, 1024, 2048, 4096, 8192, 16384}) {
        {
            const float t = time_graph([&] { for (int b = 0; b < nbuf; ++b) k_span<4><<<blocks, 256, 0, g_stream>>>(x[b], n4 / blocks, out); }) / nbuf;
            printf("  span d4  %6d blocks, 1 slab   %7.2f us  %5.2f TB/s\n", blocks, t, mb / t);
        }
        {
            const float t = time_graph([&] { for (int b = 0; b < nbuf; ++b) k_span<8><<<blocks, 256, 0, g_stream>>>(x[b], n4 / blocks, out); }) / nbuf;
            printf("  span d8  %6d blocks, 1 slab   %7.2f us  %5.2f TB/s\n", blocks, t, mb / t);
        }
        {
            const float t = time_graph([&] { for (int b = 0; b < 4; ++b) k_span<4><<<blocks, 256, 0, g_stream>>>(x3[b], 3 * n4 / blocks, out); }) / 4;
            printf("  span d4  %6d blocks, 3 slabs  %7.2f us  %5.2f TB/s\n", blocks, t, 3 * mb / t);
        }
        {
            const float t = time_graph([&] { for (int b = 0; b < 4; ++b) k_span<8><<<blocks, 256, 0, g_stream>>>(x3[b], 3 * n4 / blocks, out); }) / 4;
            printf("  span d8  %6d blocks, 3 slabs  %7.2f us  %5.2f TB/s\n", blocks, t, 3 * mb / t);
        }
    }
    for (int rpc : {16, 32, 64, 128, 256}) {
        const dim3 grid(C / 256, R / rpc);
        {
            const float t = time_graph([&] { for (int b = 0; b < nbuf; ++b) k_tile<4><<<grid, 256, 0, g_stream>>>(x[b], R, C / 4, rpc, out); }) / nbuf;
            printf("  tile d4  rows/chunk %4d (%5d blocks), 1 slab   %7.2f us  %5.2f TB/s\n", rpc, grid.x * grid.y, t, mb / t);
        }
        {
            const float t = time_graph([&] { for (int b = 0; b < nbuf; ++b) k_tile<8><<<grid, 256, 0, g_stream>>>(x[b], R, C / 4, rpc, out); }) / nbuf;
            printf("  tile d8  rows/chunk %4d (%5d blocks), 1 slab   %7.2f us  %5.2f TB/s\n", rpc, grid.x * grid.y, t, mb / t);
        }
        {
            const dim3 g3(C / 256, 3 * R / rpc);
            const float t = time_graph([&] { for (int b = 0; b < 4; ++b) k_tile<4><<<g3, 256, 0, g_stream>>>(x3[b], 3 * R, C / 4, rpc, out); }) / 4;
            printf("  tile d4  rows/chunk %4d (%5d blocks), 3 slabs  %7.2f us  %5.2f TB/s\n", rpc, g3.x * g3.y, t, 3 * mb / t);
        }
    }
    {
        const float t = time_graph([&] { for (int b = 0; b < 4; ++b) k_copy<<<(3 * n4 + 1023) / 1024, 256, 0, g_stream>>>(x3[b], y, 3 * n4); }) / 4;
        printf("  copy 3 slabs %7.2f us  %5.2f TB/s (read+write)\n", t, 6 * mb / t);
    }
    // one empty-ish launch: fixed cost of a graph-replayed kernel
    {
        const float t = time_graph([&] { for (int b = 0; b < 16; ++b) k_span<4><<<256, 256, 0, g_stream>>>(x[0], 256, out); }) / 16;
        printf("  tiny kernel %7.2f us per launch in a chain of 16\n", t);
    }
    printf("done\n");
    return 0;
}
