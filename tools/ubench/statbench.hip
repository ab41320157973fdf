// Development micro-benchmark: where the statistics kernel of the activation quantizer loses time against a pure
// read of the same geometry.  MODE 0 = the kernel's first form, 1 = call counter requested first, 2 = no publish tail
// (upper bound), 3 = counter first + uniform-slot tail.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../llm-quantization_amd/csrc statbench.hip -o statbench
#include <algorithm>
#include <functional>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include "../../llm-quantization_amd/csrc/colquant.hip"

static thread_local char g_err[512];
extern "C" void miq_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
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

namespace miq {
template <typename T, int VEC, int DEPTH, int MODE>
__global__ void __launch_bounds__(256)
k_stat_x(SlabPtrs xs, long long R, int C, int rows_per_chunk, SlotMap sm, u64* ws_base, long long ws_stride, int nrep) {
    __shared__ float smin[3][64 * VEC];
    __shared__ float smax[3][64 * VEC];
    const T* __restrict__ x = static_cast<const T*>(xs.p[blockIdx.z]);
    u64* ws = ws_base;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    const bool valid = col0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    unsigned tag = 0;
    if (MODE >= 1 && w == 0) tag = (unsigned)__builtin_nontemporal_load(ws) + 1u;
    float mn[VEC], mx[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mn[e] = __builtin_inff(); mx[e] = -__builtin_inff(); }
    if (valid) {
        const T* p = x + col0;
        long long r = r0 + w;
        for (; r + (DEPTH - 1) * 4 < r1; r += DEPTH * 4) {
            Pack<T, VEC> a[DEPTH];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) a[u] = ld_pack<T, VEC>(p + (r + u * 4) * C);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float f = to_f<T>(a[u].e[e]);
                    mn[e] = nmin(mn[e], f);
                    mx[e] = nmax(mx[e], f);
                }
        }
        for (; r < r1; r += 4) {
            Pack<T, VEC> a = ld_pack<T, VEC>(p + r * C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float f = to_f<T>(a.e[e]);
                mn[e] = nmin(mn[e], f);
                mx[e] = nmax(mx[e], f);
            }
        }
    }
    if (MODE == 0 && w == 0) tag = (unsigned)ws[0] + 1u;
    if (MODE == 3) {
        // uniform slots: reduce the pack to one pair before the exchange
        float lo_v = mn[0], hi_v = mx[0];
#pragma unroll
        for (int e = 1; e < VEC; ++e) { lo_v = nmin(lo_v, mn[e]); hi_v = nmax(hi_v, mx[e]); }
        if (w > 0) { smin[w - 1][lane] = lo_v; smax[w - 1][lane] = hi_v; }
        __syncthreads();
        if (w != 0) return;
#pragma unroll
        for (int u = 0; u < 3; ++u) { lo_v = nmin(lo_v, smin[u][lane]); hi_v = nmax(hi_v, smax[u][lane]); }
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) ws[1] = tag;
        u64* const table = ws + kSlotStride + blockIdx.z * ws_stride;
        const int rep = blockIdx.y % nrep;
        const u64 hi = (u64)tag << 32;
        const int key = valid ? (int)(col0 / sm.csize) : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int k2 = __shfl_up(key, o);
            const float a = __shfl_up(lo_v, o), b = __shfl_up(hi_v, o);
            if (lane >= o && k2 == key) { lo_v = nmin(lo_v, a); hi_v = nmax(hi_v, b); }
        }
        const int knext = __shfl_down(key, 1);
        if (valid && (lane == 63 || knext != key)) {
            u64* q = table + ((long long)key * nrep + rep) * kSlotStride;
            atomicMax(q, hi | min_key(lo_v));
            atomicMax(q + 1, hi | max_key(hi_v));
        }
        return;
    }
    if (w > 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            smin[w - 1][lane * VEC + e] = mn[e];
            smax[w - 1][lane * VEC + e] = mx[e];
        }
    }
    __syncthreads();
    if (w != 0) return;
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            mn[e] = nmin(mn[e], smin[u][lane * VEC + e]);
            mx[e] = nmax(mx[e], smax[u][lane * VEC + e]);
        }
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) ws[1] = tag;
    u64* const table = ws + kSlotStride + blockIdx.z * ws_stride;
    const int rep = blockIdx.y % nrep;
    const u64 hi = (u64)tag << 32;
    auto publish = [&](int slot, float lo_v, float hi_v) {
        u64* q = table + ((long long)slot * nrep + rep) * kSlotStride;
        atomicMax(q, hi | min_key(lo_v));
        atomicMax(q + 1, hi | max_key(hi_v));
    };
    if (MODE == 2) {
        float lo_v = mn[0], hi_v = mx[0];
#pragma unroll
        for (int e = 1; e < VEC; ++e) { lo_v = nmin(lo_v, mn[e]); hi_v = nmax(hi_v, mx[e]); }
        if (lo_v > hi_v) publish(lane, lo_v, hi_v);  // never for real data
        return;
    }
    int sl[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) sl[e] = valid ? sm(col0 + e) : -1;
    bool uniform = valid;
#pragma unroll
    for (int e = 1; e < VEC; ++e) uniform = uniform && (sl[e] == sl[0]);
    float lo_v = mn[0], hi_v = mx[0];
#pragma unroll
    for (int e = 1; e < VEC; ++e) { lo_v = nmin(lo_v, mn[e]); hi_v = nmax(hi_v, mx[e]); }
    const int key = uniform ? sl[0] : -2 - lane;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int k2 = __shfl_up(key, o);
        const float a = __shfl_up(lo_v, o), b = __shfl_up(hi_v, o);
        if (lane >= o && k2 == key) { lo_v = nmin(lo_v, a); hi_v = nmax(hi_v, b); }
    }
    const int knext = __shfl_down(key, 1);
    if (uniform) {
        if (lane == 63 || knext != key) publish(key, lo_v, hi_v);
    } else if (valid) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) publish(sl[e], mn[e], mx[e]);
    }
}
}  // namespace miq

template <typename T>
static void run_shape(const char* label, long long R, int C, int cluster, int nslab, int nbuf) {
    constexpr int N = VecOf<T>::N;
    const long long n = R * C;
    const double mb = n * sizeof(T) / 1e6;
    printf("\n=== %s [%lld, %d] %s, clusters of %d, %d slab(s) per launch (%.1f MB each) ===\n", label, R, C,
           sizeof(T) == 4 ? "fp32" : "fp16", cluster, nslab, mb);
    std::vector<T*> x(nbuf);
    for (int b = 0; b < nbuf; ++b) { CK(hipMalloc(&x[b], n * sizeof(T))); CK(hipMemset(x[b], 0x3c, n * sizeof(T))); }
    const int ncl = C / cluster, covered = ncl * cluster;
    const size_t wsb = (size_t)miq_act_fresh_workspace(8, C, ncl, covered);
    u64* ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
    const int tiles = (C / N + 63) / 64;
    const int nchunk = pick_chunks(R, tiles);
    const int rpc = (int)((R + nchunk - 1) / nchunk);
    const int nrep = nchunk > 16 ? kSlotReplicas : 1;
    const long long ws_stride = fresh_table_words(fresh_slots(C, ncl, covered));
    const SlotMap sm{nullptr, cluster, covered, ncl};
    const dim3 grid(tiles, nchunk, nslab);
    const int groups = nbuf / nslab;
    auto slabs = [&](int g) { SlabPtrs p{}; for (int i = 0; i < nslab; ++i) p.p[i] = x[(g * nslab + i) % nbuf]; return p; };
    printf("  grid %d x %d x %d, rows/chunk %d, replicas %d\n", tiles, nchunk, nslab, rpc, nrep);
#define RUN(MODE, DEPTH, name) { \
        const float t = time_graph([&] { for (int g = 0; g < groups; ++g) hipLaunchKernelGGL((k_stat_x<T, N, DEPTH, MODE>), grid, dim3(256), 0, g_stream, slabs(g), R, C, rpc, sm, ws, ws_stride, nrep); }) / groups; \
        printf("  %-34s %7.2f us  %5.2f TB/s\n", name, t, nslab * mb / t); }
    {   // what the library launches for this shape (act_fresh_t): uniform slots, 8 rows in flight in fp32, 4 in fp16
        constexpr int D = sizeof(T) == 4 ? 8 : 4;
        const float t = time_graph([&] { for (int g = 0; g < groups; ++g) hipLaunchKernelGGL((k_colstat_slots<T, N, D, true>), grid, dim3(256), 0, g_stream, slabs(g), (long long)R, C, rpc, sm, ws, ws_stride, nrep); }) / groups;
        printf("  %-34s %7.2f us  %5.2f TB/s\n", "k_colstat_slots as shipped", t, nslab * mb / t);
    }
    RUN(0, 4, "first form (per-channel tail)");
    RUN(1, 4, "counter first");
    RUN(2, 4, "counter first, no publish");
    RUN(3, 4, "counter first, uniform tail");
    RUN(3, 8, "counter first, uniform tail, d8");
    for (int b = 0; b < nbuf; ++b) CK(hipFree(x[b]));
    CK(hipFree(ws));
}

int main() {
    CK(hipStreamCreate(&g_stream));
    run_shape<float>("q/k/v input", 2048, 4096, 128, 3, 12);
    run_shape<float>("gate/up input", 2048, 4096, 128, 2, 12);
    run_shape<__half>("q/k/v output", 2048, 4096, 32, 3, 24);
    run_shape<__half>("o input", 2048, 4096, 32, 1, 24);
    run_shape<__half>("down input", 2048, 11008, 344, 1, 10);
    run_shape<float>("per-tensor", 2048, 4096, 4096, 3, 12);
    printf("done\n");
    return 0;
}
