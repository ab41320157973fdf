// Rotary position embedding of the Llama block's query / key states (reference models/int_llama_layer.py:124-125,
// transformers' apply_rotary_pos_emb):   y = x * cos + rotate_half(x) * sin,   rotate_half(x) = cat(-x_hi, x_lo)
// between the q/k projections and the quantizers of the q.k^T matmul.  PyTorch runs it as five elementwise kernels
// per tensor forward (neg, cat, two products, sum) and about eight backward, on a transposed view of the projection
// output that the next operator then has to copy into a contiguous tensor.  Here one kernel per tensor and
// direction reads the strided [B, T, H, D] projection layout and writes the contiguous [B, H, T, D] result (and the
// reverse for the gradient), one thread per 16-byte pack of the low half together with its partner in the high half.
//
// Bit-exact with the PyTorch chain: every product and the sum are rounded to the tensor dtype like the
// materialised intermediates (the translation unit is built without FMA contraction).  Backward = what autograd
// derives: dx = g * cos + rotate_half^T(g * sin), i.e. dx_lo = g_lo cos_lo + g_hi sin_hi, dx_hi = g_hi cos_hi - g_lo sin_lo.
#include "miq_common.h"

namespace miq {

struct RotArgs {
    const void* x;
    const void* cos;  // [table_batches, T, D]
    const void* sin;
    void* out;
    long long H, T;
    int D;
    long long xs_b, xs_h, xs_t;  // element strides of x  (logical index [b, h, t, :], last stride 1)
    long long os_b, os_h, os_t;  // element strides of out
    long long tab_bs;            // table stride between batch entries (0: one table for all)
    long long n_threads;         // B * H * T * (D / 2 / VEC)
};

template <typename T, int VEC, bool BWD>
__global__ void __launch_bounds__(256) k_rotary(RotArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n_threads) return;
    const int half = a.D / 2;
    const int ppr = half / VEC;  // threads per (b, h, t) row
    const long long row = i / ppr;
    const int j = (int)(i - row * ppr) * VEC;
    const long long t = row % a.T;
    const long long bh = row / a.T;
    const long long h = bh % a.H, b = bh / a.H;
    const T* xp = reinterpret_cast<const T*>(a.x) + b * a.xs_b + h * a.xs_h + t * a.xs_t + j;
    const long long tab = b * a.tab_bs + t * a.D + j;
    const T* cp = reinterpret_cast<const T*>(a.cos) + tab;
    const T* sp = reinterpret_cast<const T*>(a.sin) + tab;
    const Pack<T, VEC> xl = ld_pack_nt<T, VEC>(xp), xh = ld_pack_nt<T, VEC>(xp + half);
    const Pack<T, VEC> cl = ld_pack<T, VEC>(cp), ch = ld_pack<T, VEC>(cp + half);
    const Pack<T, VEC> sl = ld_pack<T, VEC>(sp), sh = ld_pack<T, VEC>(sp + half);
    Pack<T, VEC> ol, oh;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const float vl = to_f<T>(xl.e[e]), vh = to_f<T>(xh.e[e]);
        const float pl = rnd<T>(vl * to_f<T>(cl.e[e])), ph = rnd<T>(vh * to_f<T>(ch.e[e]));
        float lo, hi;
        if (BWD) {
            lo = pl + rnd<T>(vh * to_f<T>(sh.e[e]));
            hi = ph - rnd<T>(vl * to_f<T>(sl.e[e]));
        } else {
            lo = pl - rnd<T>(vh * to_f<T>(sl.e[e]));
            hi = ph + rnd<T>(vl * to_f<T>(sh.e[e]));
        }
        ol.e[e] = from_f<T>(lo);
        oh.e[e] = from_f<T>(hi);
    }
    T* op = reinterpret_cast<T*>(a.out) + b * a.os_b + h * a.os_h + t * a.os_t + j;
    st_pack<T, VEC>(op, ol);
    st_pack<T, VEC>(op + half, oh);
}

template <typename T, int VEC>
static void rotary_launch(RotArgs a, long long B, int backward, hipStream_t stream) {
    a.n_threads = B * a.H * a.T * (a.D / 2 / VEC);
    const unsigned grid = (unsigned)((a.n_threads + 255) / 256);
    if (backward) hipLaunchKernelGGL((k_rotary<T, VEC, true>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((k_rotary<T, VEC, false>), dim3(grid), dim3(256), 0, stream, a);
}

}  // namespace miq

using namespace miq;

extern "C" int miq_rotary(const void* x, const void* cos, const void* sin, void* out, int64_t B, int64_t H, int64_t T, int64_t D,
                          int64_t xs_b, int64_t xs_h, int64_t xs_t, int64_t os_b, int64_t os_h, int64_t os_t,
                          int64_t table_batch_stride, int backward, int dtype, void* stream) {
    if (B == 0 || H == 0 || T == 0 || D == 0) return 0;
    MIQ_CHECK_ARG(x && cos && sin && out && B > 0 && H > 0 && T > 0 && D > 0, "miq_rotary: null pointer or bad shape");
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_rotary: unsupported dtype %d", dtype);
    const int vec = dtype == MIQ_F16 ? 8 : 4;
    MIQ_CHECK_ARG(D % (2 * vec) == 0 && D <= 65536, "miq_rotary: head dimension %lld must be a multiple of %d", (long long)D, 2 * vec);
    MIQ_CHECK_ARG(xs_b % vec == 0 && xs_h % vec == 0 && xs_t % vec == 0 && os_b % vec == 0 && os_h % vec == 0 && os_t % vec == 0 &&
                      table_batch_stride % vec == 0 && xs_b >= 0 && xs_h >= 0 && xs_t >= 0 && os_b >= 0 && os_h > 0 && os_t > 0,
                  "miq_rotary: strides must be non-negative multiples of %d elements", vec);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(cos) && is_aligned16(sin) && is_aligned16(out), "miq_rotary: pointers must be 16-byte aligned");
    MIQ_CHECK_ARG(B * H * T < (1ll << 40), "miq_rotary: tensor too large");
    RotArgs a;
    a.x = x; a.cos = cos; a.sin = sin; a.out = out;
    a.H = H; a.T = T; a.D = (int)D;
    a.xs_b = xs_b; a.xs_h = xs_h; a.xs_t = xs_t;
    a.os_b = os_b; a.os_h = os_h; a.os_t = os_t;
    a.tab_bs = table_batch_stride;
    a.n_threads = 0;
    if (dtype == MIQ_F32) rotary_launch<float, 4>(a, B, backward, (hipStream_t)stream);
    else rotary_launch<__half, 8>(a, B, backward, (hipStream_t)stream);
    MIQ_CHECK_LAUNCH("miq_rotary");
    return 0;
}
