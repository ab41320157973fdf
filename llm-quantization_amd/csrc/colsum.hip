// Column sums of a row-major [R, C] matrix: out[c] = sum_r x[r, c] in fp32.
// This is the bias gradient of QuantLinear's F.linear (reference quantize/int_linear.py:86) in the calibration step,
// where every linear carries a temporary (LET-shifted) bias: six [tokens, out_features] reductions per Llama block
// step.  ATen's reduction over the slow dimension reaches ~0.4 TB/s on [2048, 11008] fp16 (112 us); this one streams
// the matrix once with the activation kernels' tiling: lane = a 16-byte pack of channels, wave = every 4th row of a
// row chunk (each wave-load is one contiguous 1 KiB segment), (column tiles x row chunks) ~ 1024 blocks, fp32
// partials per chunk and a deterministic fold.
#include "miq_common.h"

namespace miq {

template <typename T, int VEC>
__global__ void __launch_bounds__(256)
k_colsum_partial(const T* __restrict__ x, long long R, int C, int rows_per_chunk, float* __restrict__ part, int round_half) {
    __shared__ float sm[4][64][VEC + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long col0 = ((long long)blockIdx.x * 64 + lane) * VEC;
    const bool valid = col0 < C;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (valid) {
        const T* p = x + col0;
        long long row = r0 + wv;
        for (; row + 12 < r1; row += 16) {  // four loads in flight
            Pack<T, VEC> v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = ld_pack_nt<T, VEC>(p + (row + 4 * u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += to_f<T>(v[u].e[e]);
            }
        }
        for (; row < r1; row += 4) {
            const Pack<T, VEC> v = ld_pack_nt<T, VEC>(p + row * C);
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += to_f<T>(v.e[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) sm[wv][lane][e] = acc[e];
    __syncthreads();
    if (wv != 0 || !valid) return;
    float* o = part + (long long)blockIdx.y * C + col0;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float v = (sm[0][lane][e] + sm[1][lane][e]) + (sm[2][lane][e] + sm[3][lane][e]);
        o[e] = round_half ? rnd<__half>(v) : v;  // only set when this launch writes the final sums (one chunk)
    }
}

// out[c] = sum over the nrows partial rows in a fixed order; block = 64 columns x 4 waves
__global__ void __launch_bounds__(256)
k_colsum_fold(const float* __restrict__ part, int nrows, int C, float* __restrict__ out, int round_half) {
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
    if (w == 0 && c < C) {
        const float v = (sm[0][lane] + sm[1][lane]) + (sm[2][lane] + sm[3][lane]);
        out[c] = round_half ? rnd<__half>(v) : v;
    }
}

static inline int colsum_vec(const void* x, long long C, int dtype) {
    const int vec = dtype == MIQ_F16 ? 8 : 4;
    return (C % vec == 0 && is_aligned16(x)) ? vec : 1;
}

static inline int colsum_chunks(long long R, long long C, int vec) {
    const long long tiles = ((C + vec - 1) / vec + 63) / 64;
    long long want = (1024 + tiles - 1) / tiles;
    const long long maxc = (R + 31) / 32;  // at least 32 rows per chunk: 8 per wave
    if (want > maxc) want = maxc;
    if (want < 1) want = 1;
    return (int)want;
}

template <typename T, int VEC>
static void colsum_launch(const void* x, long long R, int C, float* out, float* ws, int round_half, hipStream_t stream) {
    const int nchunk = colsum_chunks(R, C, VEC);
    const int rpc = (int)((R + nchunk - 1) / nchunk);
    const unsigned tiles = (unsigned)(((C + VEC - 1) / VEC + 63) / 64);
    float* part = nchunk == 1 ? out : ws;
    hipLaunchKernelGGL((k_colsum_partial<T, VEC>), dim3(tiles, nchunk), dim3(256), 0, stream, (const T*)x, R, C, rpc, part,
                       nchunk == 1 ? round_half : 0);
    if (nchunk > 1)
        hipLaunchKernelGGL(k_colsum_fold, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, stream, ws, nchunk, C, out, round_half);
}

}  // namespace miq

using namespace miq;

// floats of workspace for the worst case over both vector widths
extern "C" int64_t miq_col_sum_workspace(int64_t R, int64_t C) {
    if (R <= 0 || C <= 0) return 0;
    int n = colsum_chunks(R, C, 1);
    for (int vec = 4; vec <= 8; vec *= 2) {
        const int m = colsum_chunks(R, C, vec);
        if (m > n) n = m;
    }
    return (int64_t)n * C;
}

extern "C" int miq_col_sum(const void* x, float* out, float* workspace, int64_t R, int64_t C, int dtype, int round_dtype,
                           void* stream) {
    if (C == 0) return 0;
    MIQ_CHECK_ARG(out && C > 0 && C < (1ll << 31) && R >= 0 && R < (1ll << 40), "miq_col_sum: null pointer or bad shape");
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_col_sum: unsupported dtype %d", dtype);
    MIQ_CHECK_ARG(round_dtype == MIQ_F32 || round_dtype == MIQ_F16, "miq_col_sum: unsupported rounding dtype %d", round_dtype);
    const int rh = round_dtype == MIQ_F16;
    hipStream_t s = (hipStream_t)stream;
    if (R == 0) {
        if (hipMemsetAsync(out, 0, sizeof(float) * C, s) != hipSuccess) {
            miq_set_error("miq_col_sum: memset failed");
            return -5;
        }
        return 0;
    }
    MIQ_CHECK_ARG(x, "miq_col_sum: null input");
    const int vec = colsum_vec(x, C, dtype);
    MIQ_CHECK_ARG(colsum_chunks(R, C, vec) == 1 || workspace, "miq_col_sum: workspace of miq_col_sum_workspace(R, C) floats required");
    if (dtype == MIQ_F32) {
        if (vec == 4) colsum_launch<float, 4>(x, R, (int)C, out, workspace, rh, s);
        else colsum_launch<float, 1>(x, R, (int)C, out, workspace, rh, s);
    } else {
        if (vec == 8) colsum_launch<__half, 8>(x, R, (int)C, out, workspace, rh, s);
        else colsum_launch<__half, 1>(x, R, (int)C, out, workspace, rh, s);
    }
    MIQ_CHECK_LAUNCH("miq_col_sum");
    return 0;
}
