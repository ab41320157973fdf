// Development micro-benchmark of the activation-quantizer forward kernels: statistics-kernel variants (loads in
// flight, waves per block, rows per chunk), the slot-table pipeline against the four-kernel pipeline, on a working
// set larger than the Infinity Cache, each variant replayed from a HIP graph and timed with hipEvents.
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I../../llm-quantization_amd/csrc actbench.hip -o actbench
#include <algorithm>
#include <functional>
#include <string>
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

template <typename T> struct Host;
template <> struct Host<float> { static float make(float v) { return v; } };
template <> struct Host<__half> { static __half make(float v) { return __float2half(v); } };

static hipStream_t g_stream;

// replay `enqueue` (which launches on g_stream) from a graph; median time per replay in microseconds
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

template <typename T>
static void run_shape(const char* label, long long R, int C, int cluster, int nbuf) {
    constexpr int N = VecOf<T>::N;
    const long long n = R * C;
    printf("\n=== %s [%lld, %d] %s, clusters of %d channels, %d buffers of %.1f MB ===\n", label, R, C,
           sizeof(T) == 4 ? "fp32" : "fp16", cluster, nbuf, n * sizeof(T) / 1e6);
    std::vector<T*> x(nbuf), out(nbuf), out2(nbuf);
    std::vector<T> h(n);
    unsigned st = 12345;
    for (long long i = 0; i < n; ++i) {
        st = st * 1664525u + 1013904223u;
        const float u = ((st >> 8) & 0xFFFF) / 65536.f - 0.5f;
        h[i] = Host<T>::make(u * (1.f + (i % C) * 0.01f));
    }
    for (int b = 0; b < nbuf; ++b) {
        CK(hipMalloc(&x[b], n * sizeof(T))); CK(hipMalloc(&out[b], n * sizeof(T))); CK(hipMalloc(&out2[b], n * sizeof(T)));
        CK(hipMemcpy(x[b], h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    }
    // cluster layout
    const int ncl = C / cluster;
    std::vector<int> starts(ncl + 1), slot(C);
    for (int k = 0; k <= ncl; ++k) starts[k] = k * cluster;
    const int covered = ncl * cluster;
    for (int c = 0; c < C; ++c) slot[c] = c < covered ? c / cluster : ncl + (c - covered);
    const int nslot = ncl + (C - covered);
    int *d_starts, *d_slot;
    CK(hipMalloc(&d_starts, (ncl + 1) * 4)); CK(hipMalloc(&d_slot, C * 4));
    CK(hipMemcpy(d_starts, starts.data(), (ncl + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_slot, slot.data(), C * 4, hipMemcpyHostToDevice));
    const int nchunk0 = miq_colstat_chunks(R, C, sizeof(T) == 4 ? MIQ_F32 : MIQ_F16);
    float* wsf; CK(hipMalloc(&wsf, 2ll * (nchunk0 + 1) * C * 4));
    const int NREPMAX = 16;
    const size_t wsb = (size_t)miq_act_fresh_workspace(8, C, ncl, covered); (void)NREPMAX; (void)nslot;
    u64* ws; CK(hipMalloc(&ws, wsb)); CK(hipMemset(ws, 0, wsb));
    T *v5a, *v5b; CK(hipMalloc(&v5a, 5ll * C * sizeof(T))); CK(hipMalloc(&v5b, 5ll * C * sizeof(T)));
    const int dt = sizeof(T) == 4 ? MIQ_F32 : MIQ_F16;
    const int tiles = (C / N + 63) / 64;
    const float qmax = 15.f;
    const double mb = n * sizeof(T) / 1e6;

    // ---- statistics kernel alone
    {
        const float t = time_graph([&] { for (int b = 0; b < nbuf; ++b) miq_colstat_minmax(x[b], R, C, wsf, wsf + (long long)(nchunk0 + 1) * C, nchunk0, dt, g_stream); }) / nbuf;
        printf("  stat  four-kernel form (4 waves, 4 loads, %4d chunks)   %7.2f us  %6.2f TB/s\n", nchunk0, t, mb / t);
    }
    // ---- whole forward: four-kernel form against the slot-table form (miq_act_fresh_quant_fwd), 1..3 slabs per launch
    auto legacy = [&](int b) {
        miq_act_calib_quant_fwd(x[b], nullptr, nullptr, 1, d_starts, ncl, covered, 4, wsf, out[b], nullptr, v5a, v5a + C, v5a + 2 * C,
                                v5a + 3 * C, v5a + 4 * C, R, C, dt, g_stream);
    };
    const float t_old = time_graph([&] { for (int b = 0; b < nbuf; ++b) legacy(b); }) / nbuf;
    printf("  fwd   four kernels                          %7.2f us  %6.2f TB/s (3ns)\n", t_old, 3 * mb / t_old);
    {
        const float t = time_graph([&] { for (int b = 0; b < nbuf; ++b) miq_colquant_fwd(x[b], v5a + 2 * C, v5a + 4 * C, out[b], nullptr, R, C, 4, dt, g_stream); }) / nbuf;
        printf("  quant four-kernel form alone                %7.2f us  %6.2f TB/s (2ns)\n", t, 2 * mb / t);
    }
    for (int use_map = 0; use_map < 2; ++use_map)
        for (int nb : {1, 2, 3}) {
            auto fresh = [&](int b0, std::vector<T*>& o) {
                const void* xp[8]; void* op[8]; void* vp[8];
                for (int i = 0; i < nb; ++i) { xp[i] = x[(b0 + i) % nbuf]; op[i] = o[(b0 + i) % nbuf]; vp[i] = v5b; }
                const int rc = miq_act_fresh_quant_fwd(nb, xp, use_map ? 0 : cluster, d_slot, ncl, covered, 4, ws, op, nullptr, vp, R, C, dt, g_stream);
                if (rc) { printf("fresh failed: %s\n", g_err); exit(1); }
            };
            const int groups = nbuf / nb;
            const float t = time_graph([&] { for (int g = 0; g < groups; ++g) fresh(g * nb, out2); }) / (groups * nb);
            printf("  fwd   slots %s, %d slab(s) per launch          %7.2f us  %6.2f TB/s (3ns)\n", use_map ? "map" : "uni", nb, t, 3 * mb / t);
            legacy(0); fresh(0, out2);
            CK(hipStreamSynchronize(g_stream));
            std::vector<T> a(n), b2(n), va(5ll * C), vb(5ll * C);
            for (int i = 0; i < nb; ++i) {
                legacy(i); CK(hipStreamSynchronize(g_stream));
                CK(hipMemcpy(a.data(), out[i], n * sizeof(T), hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), out2[i], n * sizeof(T), hipMemcpyDeviceToHost));
                if (memcmp(a.data(), b2.data(), n * sizeof(T)) != 0) printf("        *** MISMATCH (out, slab %d) against the four-kernel form ***\n", i);
            }
            CK(hipMemcpy(va.data(), v5a, 5ll * C * sizeof(T), hipMemcpyDeviceToHost)); CK(hipMemcpy(vb.data(), v5b, 5ll * C * sizeof(T), hipMemcpyDeviceToHost));
            if (memcmp(va.data(), vb.data(), 5ll * C * sizeof(T)) != 0) printf("        *** MISMATCH (vectors) against the four-kernel form ***\n");
        }

    for (int b = 0; b < nbuf; ++b) { CK(hipFree(x[b])); CK(hipFree(out[b])); CK(hipFree(out2[b])); }
    CK(hipFree(d_starts)); CK(hipFree(d_slot)); CK(hipFree(wsf)); CK(hipFree(ws)); CK(hipFree(v5a)); CK(hipFree(v5b));
}

int main() {
    CK(hipStreamCreate(&g_stream));
    run_shape<float>("q/k/v input", 2048, 4096, 128, 10);
    run_shape<__half>("q/k/v output", 2048, 4096, 32, 18);
    run_shape<__half>("down input", 2048, 11008, 344, 8);
    run_shape<float>("per-tensor", 2048, 4096, 4096, 10);
    run_shape<__half>("70B hidden", 2048, 8192, 256, 10);
    printf("done\n");
    return 0;
}
