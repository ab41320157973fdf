// Attention-score chain of the decoder blocks between the two quantized matmuls, in one pass per direction:
//     w = scores / sqrt(head_dim)        (Llama only; reference models/int_llama_layer.py:139)
//     w = w + attention_mask             (broadcast over heads)
//     w = max(w, finfo.min)              (int_llama_layer.py:149-151, int_opt_layer.py:207-210)
//     p = softmax(w, dim=-1, dtype=float32).to(scores.dtype)
// PyTorch runs these as five elementwise / reduction kernels over the [heads*T, T] score matrix (134 M elements
// for a Llama-2-7B block: ~4.8 GB of traffic forward, ~4 GB backward); here a team of threads owns one row, keeps
// it in registers, and the row crosses HBM once in each direction.
//
// Arithmetic: every step in fp32 on the values PyTorch would see (the fp16 quotient by sqrt(head_dim) is rounded
// to fp16 first, and formed as a product with the fp32 reciprocal like ATen's tensor / scalar on a GPU); max / sum
// reductions in fp32.  The summation order of
// the softmax differs from ATen's, so probabilities agree to fp32 rounding, not bit for bit (these operators are
// not quantizer arithmetic; the blocks' tests compare them with the PyTorch chain).
// Backward from the probabilities p and the incoming gradient g:
//     dw = p * (g - sum_j g_j p_j) / sqrt(head_dim)
// (the softmax gradient is rounded to the scores' dtype before the division, as the reference's cast precedes its
// division node)
// (positions cut by the max(., finfo.min) have p == 0, hence no gradient, like the reference's clamp mask; a row whose
// every key is masked has uniform p, and there ATen's maximum halves the gradient at w == finfo.min: the backward from
// the scores reproduces that, the one from the probabilities cannot see it).
// PyTorch keeps the fp32 softmax output for this, not the rounded copy it returns.  For fp32 scores the saved
// output is that tensor.  For fp16 scores the forward also stores each row's fp32 maximum and reciprocal sum
// (8 bytes a row), and the backward rebuilds the fp32 probabilities from the scores and the mask with the same
// instructions as the forward (miq_masked_softmax_bwd_scores): the same traffic as reading the rounded
// probabilities, without their 2^-11 relative error in every gradient element.
#include "miq_common.h"

namespace miq {

struct SmArgs {
    const void* x;       // [R, C] scores
    const void* mask;    // [mask_batches * q_len, C] or nullptr
    int mask_half;       // mask is fp16 (else fp32)
    int sum_half;        // scores + mask is an fp16 tensor (both operands fp16): round the sum to fp16
    float inv_div;       // reciprocal of the divisor applied first (1 = none): ATen multiplies by 1/scalar on the GPU
    float floor_;        // the lower clamp (finfo.min of the sum's dtype)
    long long R;
    int C;
    long long rows_per_batch;  // heads * q_len score rows share one batch entry of the mask
    int q_len;
    int head_block;      // heads a block walks per query row (set by the launcher, with the two counts below)
    unsigned qblocks;           // q_len / rows per block
    unsigned blocks_per_batch;  // (heads / head_block) * qblocks
};

struct OpMaxNan {
    __device__ float operator()(int, float a, float b) const { return nmax(a, b); }
};

__device__ __forceinline__ long long mask_row(const SmArgs& a, long long row) {
    return (row / a.rows_per_batch) * a.q_len + row % a.q_len;
}

// The mask values of one score row in registers, in the row tile's geometry.  Every head of a batch entry adds the same
// [q_len, C] mask, so a block that walks several heads of one query row loads them once (a 2048 x 2048 fp32 mask re-read
// by each of 32 heads is twice the traffic of the fp16 scores themselves, and it shows: 31 us of a 121 us forward).
template <int VEC, int TEAM, int NV>
struct MaskTile {
    float m[NV][VEC];
    template <typename TX>
    __device__ __forceinline__ void load(const SmArgs& a, long long mrow, const RowTile<TX, VEC, TEAM, NV>& geo) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int e = 0; e < VEC; ++e) m[k][e] = 0.f;  // no mask: w + 0 is w
        if (!a.mask) return;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!geo.has(k)) continue;
            if (a.mask_half) {
                const Pack<__half, VEC> p = ld_pack<__half, VEC>(reinterpret_cast<const __half*>(a.mask) + mrow * a.C + geo.offset(k));
#pragma unroll
                for (int e = 0; e < VEC; ++e) m[k][e] = __half2float(p.e[e]);
            } else {
                const float* mp = reinterpret_cast<const float*>(a.mask) + mrow * a.C + geo.offset(k);
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) {  // 16-byte loads: VEC is 4 or 8
                    const Pack<float, 4> p = ld_pack<float, 4>(mp + 4 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) m[k][4 * q + e] = p.e[e];
                }
            }
        }
    }
};

// v = x / div + mask of one row, with the roundings of the PyTorch chain; the caller applies the floor.
// No test of the launch's options per element (the compiler keeps each as a scalar branch around one or two
// instructions, a hundred of them per row tile): a divisor of 1 multiplies exactly, an absent mask adds 0, and the
// rounding of an fp16 sum is a select.
template <typename TX, int VEC, int TEAM, int NV>
__device__ __forceinline__ void masked_scores_pack(const SmArgs& a, const RowTile<TX, VEC, TEAM, NV>& x,
                                                   const MaskTile<VEC, TEAM, NV>& mt, int k, float (&v)[VEC]) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        float t = rnd<TX>(to_f<TX>(x.v[k].e[e]) * a.inv_div);
        t = t + mt.m[k][e];
        if constexpr (sizeof(TX) == 2) {
            const float th = rnd<__half>(t);
            t = a.sum_half ? th : t;
        }
        v[e] = t;
    }
}
template <typename TX, int VEC, int TEAM, int NV>
__device__ __forceinline__ void masked_scores(const SmArgs& a, const RowTile<TX, VEC, TEAM, NV>& x, const MaskTile<VEC, TEAM, NV>& mt,
                                              float (&v)[NV][VEC]) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!x.has(k)) continue;
        masked_scores_pack(a, x, mt, k, v[k]);
    }
}

// Rows of a block.  Plain order: block b holds rows b * RPB ... (one per team).  With a.head_block = HB > 1 (the host sets
// it when a mask is present, q_len is a multiple of RPB and the head count a multiple of HB) a block holds RPB consecutive
// query rows of HB consecutive heads: its teams walk the heads one after the other with the mask values in registers.
// Consecutive blocks still walk consecutive query rows of one head group, so the score traffic keeps its order.
// The position is decomposed from the block index with 32-bit, block-uniform arithmetic; the mask row follows from the
// parts (a 64-bit division and remainder per thread, as mask_row needs them, are some 300 instructions).
template <int RPB>
__device__ __forceinline__ void block_rows(const SmArgs& a, int team, long long& row, long long& mrow) {
    if (a.head_block > 1) {
        const unsigned b = blockIdx.x;
        const unsigned batch = b / a.blocks_per_batch, r = b % a.blocks_per_batch;
        const unsigned group = r / a.qblocks, qb = r % a.qblocks;
        const unsigned q = qb * RPB + team;
        row = (long long)batch * a.rows_per_batch + (long long)group * a.head_block * a.q_len + q;
        mrow = (long long)batch * a.q_len + q;
    } else {
        row = (long long)blockIdx.x * RPB + team;
        mrow = (a.mask && row < a.R) ? mask_row(a, row) : 0;
    }
}

template <typename TX, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_masked_softmax_fwd(SmArgs a, TX* __restrict__ out, float2* __restrict__ row_stats) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    long long row, mrow;
    block_rows<RPB>(a, threadIdx.x / TEAM, row, mrow);
    // TEAM > 64 implies RPB == 1: a block is one row, so this exit is block-uniform where barriers follow
    if (row >= a.R) return;
    RowTile<TX, VEC, TEAM, NV> x;
    x.load_nt(reinterpret_cast<const TX*>(a.x) + row * a.C, a.C, lane);
    MaskTile<VEC, TEAM, NV> mt;
    mt.load(a, mrow, x);
    for (int hh = 0;;) {
        float v[NV][VEC];
        masked_scores(a, x, mt, v);
        float mx[1] = {-__builtin_inff()};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[k][e] = (v[k][e] < a.floor_) ? a.floor_ : v[k][e];  // torch.max(w, finfo.min): NaN stays NaN
                mx[0] = nmax(mx[0], v[k][e]);
            }
        }
        const long long here = row;
        const bool more = ++hh < a.head_block;
        if (more) {  // the next head's scores travel while this one is reduced
            row += a.q_len;
            x.load_nt(reinterpret_cast<const TX*>(a.x) + row * a.C, a.C, lane);
        }
        team_reduce<TEAM, 1>(mx, OpMaxNan(), lds);
        float sum[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                v[k][e] = __expf(v[k][e] - mx[0]);
                sum[0] += v[k][e];
            }
        }
        team_sum<TEAM, 1>(sum, lds);
        const float inv = 1.0f / sum[0];
        if (row_stats && lane == 0) row_stats[here] = make_float2(mx[0], inv);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<TX, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.e[e] = from_f<TX>(v[k][e] * inv);
            st_pack<TX, VEC>(out + here * a.C + x.offset(k), o);
        }
        if (!more) break;
    }
}

template <typename TX, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_masked_softmax_bwd(const TX* __restrict__ p, const TX* __restrict__ g, TX* __restrict__ dx, long long R, int C, float div) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    const long long row = (long long)blockIdx.x * RPB + threadIdx.x / TEAM;
    if (row >= R) return;
    RowTile<TX, VEC, TEAM, NV> pp, gg;
    pp.load_nt(p + row * C, C, lane);
    gg.load_nt(g + row * C, C, lane);
    float dot[1] = {0.f};
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!pp.has(k)) continue;
#pragma unroll
        for (int e = 0; e < VEC; ++e) dot[0] += to_f<TX>(pp.v[k].e[e]) * to_f<TX>(gg.v[k].e[e]);
    }
    team_sum<TEAM, 1>(dot, lds);
    const float inv_div = 1.0f / div;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (!pp.has(k)) continue;
        Pack<TX, VEC> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float pv = to_f<TX>(pp.v[k].e[e]);
            o.e[e] = from_f<TX>(rnd<TX>(pv * (to_f<TX>(gg.v[k].e[e]) - dot[0])) * inv_div);
        }
        st_pack<TX, VEC>(dx + row * C + pp.offset(k), o);
    }
}

// The same from the scores: p = exp(w - max) * (1 / sum) in fp32 exactly as the forward formed it.
// Gradient of torch.max(w, floor) with respect to w: 1 above the floor, 0 below, 1/2 on a tie (ATen's rule for maximum).
// It differs from 1 only where w <= floor, and there p = exp(floor - max) is exactly 0 (hence the gradient 0 whatever the
// factor) unless the row's maximum lies within ~104 of the floor (a row whose every key is masked): such rows take the
// branch that applies the factor, decided once per row.
template <typename TX, int VEC, int TEAM, int NV>
__global__ void __launch_bounds__(block_of<TEAM>())
k_masked_softmax_bwd_scores(SmArgs a, const float2* __restrict__ row_stats, const TX* __restrict__ g, TX* __restrict__ dx) {
    __shared__ float lds[32];
    constexpr int RPB = block_of<TEAM>() / TEAM;
    const int lane = threadIdx.x % TEAM;
    long long row, mrow;
    block_rows<RPB>(a, threadIdx.x / TEAM, row, mrow);
    if (row >= a.R) return;
    RowTile<TX, VEC, TEAM, NV> x, gg;
    x.load_nt(reinterpret_cast<const TX*>(a.x) + row * a.C, a.C, lane);
    gg.load_nt(g + row * a.C, a.C, lane);
    MaskTile<VEC, TEAM, NV> mt;
    mt.load(a, mrow, x);
    for (int hh = 0;;) {
        const float2 st = row_stats[row];
        const bool near_floor = !(st.x - a.floor_ > 128.f);  // also a NaN maximum
        float v[NV][VEC];
        masked_scores(a, x, mt, v);
        float dot[1] = {0.f};
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float w = v[k][e];
                v[k][e] = __expf(((w < a.floor_) ? a.floor_ : w) - st.x) * st.y;
                dot[0] += v[k][e] * to_f<TX>(gg.v[k].e[e]);
            }
        }
        team_sum<TEAM, 1>(dot, lds);
        if (near_floor) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!x.has(k)) continue;
                float w[VEC];
                masked_scores_pack(a, x, mt, k, w);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float pass = (w[e] < a.floor_) ? 0.f : (w[e] == a.floor_ ? 0.5f : 1.f);
                    const float dw = rnd<TX>(v[k][e] * (to_f<TX>(gg.v[k].e[e]) - dot[0]));
                    v[k][e] = rnd<TX>(dw * pass);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (!x.has(k)) continue;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    v[k][e] = rnd<TX>(v[k][e] * (to_f<TX>(gg.v[k].e[e]) - dot[0]));  // rounded by the softmax's cast
            }
        }
        const long long here = row;
        const bool more = ++hh < a.head_block;
        if (more) {  // the next head's operands travel while this one is stored
            row += a.q_len;
            x.load_nt(reinterpret_cast<const TX*>(a.x) + row * a.C, a.C, lane);
            gg.load_nt(g + row * a.C, a.C, lane);
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (!x.has(k)) continue;
            Pack<TX, VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.e[e] = from_f<TX>(v[k][e] * a.inv_div);
            st_pack<TX, VEC>(dx + here * a.C + x.offset(k), o);
        }
        if (!more) break;
    }
}

// grid of the head-blocked kernels and the head block they use
template <int RPB>
static inline unsigned sm_grid(SmArgs& a) {
    constexpr int HB = 4;
    const long long heads = a.rows_per_batch / a.q_len;
    const long long plain = (a.R + RPB - 1) / RPB;
    a.head_block = 1; a.qblocks = a.blocks_per_batch = 1;
    a.qblocks = a.blocks_per_batch = 1;
    if (a.mask && a.q_len % RPB == 0 && heads % HB == 0 && plain < (1ll << 32)) {
        a.head_block = HB;
        a.qblocks = (unsigned)(a.q_len / RPB);
        a.blocks_per_batch = (unsigned)(heads / HB) * a.qblocks;
        return (unsigned)(a.R / ((long long)RPB * HB));
    }
    return (unsigned)plain;
}

template <typename TX>
static int launch_sm_fwd(const SmArgs& a, void* out, float* row_stats, hipStream_t stream) {
    constexpr int VEC = VecOf<TX>::N;
    const int npack = a.C / VEC;
#define L(TEAM, NV)                                                                                               \
    {                                                                                                             \
        SmArgs b = a;                                                                                             \
        const unsigned grid = sm_grid<block_of<TEAM>() / TEAM>(b);                                                \
        hipLaunchKernelGGL((k_masked_softmax_fwd<TX, VEC, TEAM, NV>), dim3(grid), dim3(block_of<TEAM>()), 0,      \
                           stream, b, (TX*)out, (float2*)row_stats);                                              \
    }
    MIQ_ROW_DISPATCH(npack, L);
#undef L
    MIQ_CHECK_LAUNCH("miq_masked_softmax_fwd");
    return 0;
}

template <typename TX>
static int launch_sm_bwd(const void* p, const void* g, void* dx, long long R, int C, float div, hipStream_t stream) {
    constexpr int VEC = VecOf<TX>::N;
    const int npack = C / VEC;
#define L(TEAM, NV)                                                                                               \
    {                                                                                                             \
        constexpr int RPB = block_of<TEAM>() / TEAM;                                                              \
        hipLaunchKernelGGL((k_masked_softmax_bwd<TX, VEC, TEAM, NV>), dim3((unsigned)((R + RPB - 1) / RPB)),      \
                           dim3(block_of<TEAM>()), 0, stream, (const TX*)p, (const TX*)g, (TX*)dx, R, C, div);    \
    }
    MIQ_ROW_DISPATCH(npack, L);
#undef L
    MIQ_CHECK_LAUNCH("miq_masked_softmax_bwd");
    return 0;
}

template <typename TX>
static int launch_sm_bwd_scores(const SmArgs& a, const float* row_stats, const void* g, void* dx, hipStream_t stream) {
    constexpr int VEC = VecOf<TX>::N;
    const int npack = a.C / VEC;
#define L(TEAM, NV)                                                                                                  \
    {                                                                                                                \
        SmArgs b = a;                                                                                                \
        const unsigned grid = sm_grid<block_of<TEAM>() / TEAM>(b);                                                   \
        hipLaunchKernelGGL((k_masked_softmax_bwd_scores<TX, VEC, TEAM, NV>), dim3(grid), dim3(block_of<TEAM>()), 0,  \
                           stream, b, (const float2*)row_stats, (const TX*)g, (TX*)dx);                              \
    }
    MIQ_ROW_DISPATCH(npack, L);
#undef L
    MIQ_CHECK_LAUNCH("miq_masked_softmax_bwd_scores");
    return 0;
}

}  // namespace miq

using namespace miq;

// argument checks and the kernel argument block shared by the forward and the backward from the scores
static int sm_args(const char* who, const void* x, const void* mask, int64_t R, int64_t C, int64_t rows_per_batch, int64_t q_len,
                   float div, int x_dtype, int mask_dtype, SmArgs& a) {
    MIQ_CHECK_ARG(x && R > 0 && C > 0, "%s: null pointer or bad shape", who);
    MIQ_CHECK_ARG(x_dtype == MIQ_F32 || x_dtype == MIQ_F16, "%s: unsupported dtype %d", who, x_dtype);
    const int vec = x_dtype == MIQ_F16 ? 8 : 4;
    MIQ_CHECK_ARG(C % vec == 0 && C / vec <= kRowMaxPacks && C < (1ll << 31), "%s: C=%lld must be a multiple of %d and at most %d", who,
                  (long long)C, vec, vec * kRowMaxPacks);
    MIQ_CHECK_ARG(is_aligned16(x) && is_aligned16(mask), "%s: pointers must be 16-byte aligned", who);
    MIQ_CHECK_ARG(!mask || ((mask_dtype == MIQ_F32 || mask_dtype == MIQ_F16) && q_len > 0 && rows_per_batch > 0 &&
                            rows_per_batch % q_len == 0 && R % rows_per_batch == 0),
                  "%s: mask needs q_len > 0 and rows_per_batch (a multiple of q_len) dividing R", who);
    MIQ_CHECK_ARG(div > 0.f, "%s: divisor must be positive", who);
    a.x = x; a.mask = mask; a.mask_half = mask_dtype == MIQ_F16;
    a.sum_half = (x_dtype == MIQ_F16 && mask_dtype == MIQ_F16);
    a.inv_div = 1.0f / div;
    a.floor_ = !mask ? -__builtin_inff() : (a.sum_half ? -65504.0f : -3.4028234663852886e38f);
    a.R = R; a.C = (int)C; a.rows_per_batch = mask ? rows_per_batch : 1; a.q_len = mask ? (int)q_len : 1;
    a.head_block = 1;
    return 0;
}

extern "C" int miq_masked_softmax_fwd(const void* x, const void* mask, void* out, float* row_stats_out, int64_t R, int64_t C,
                                      int64_t rows_per_batch, int64_t q_len, float div, int x_dtype, int mask_dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    SmArgs a;
    if (const int rc = sm_args("miq_masked_softmax_fwd", x, mask, R, C, rows_per_batch, q_len, div, x_dtype, mask_dtype, a)) return rc;
    MIQ_CHECK_ARG(out && is_aligned16(out) && (reinterpret_cast<uintptr_t>(row_stats_out) & 7) == 0,
                  "miq_masked_softmax_fwd: out must be 16-byte aligned, row_stats_out 8-byte aligned");
    if (x_dtype == MIQ_F32) return launch_sm_fwd<float>(a, out, row_stats_out, (hipStream_t)stream);
    return launch_sm_fwd<__half>(a, out, row_stats_out, (hipStream_t)stream);
}

extern "C" int miq_masked_softmax_bwd_scores(const void* x, const void* mask, const float* row_stats, const void* grad_out, void* dx,
                                             int64_t R, int64_t C, int64_t rows_per_batch, int64_t q_len, float div, int x_dtype,
                                             int mask_dtype, void* stream) {
    if (R == 0 || C == 0) return 0;
    SmArgs a;
    if (const int rc = sm_args("miq_masked_softmax_bwd_scores", x, mask, R, C, rows_per_batch, q_len, div, x_dtype, mask_dtype, a)) return rc;
    MIQ_CHECK_ARG(row_stats && grad_out && dx && is_aligned16(grad_out) && is_aligned16(dx) && (reinterpret_cast<uintptr_t>(row_stats) & 7) == 0,
                  "miq_masked_softmax_bwd_scores: null pointer, or grad_out / dx not 16-byte aligned, or row_stats not 8-byte aligned");
    if (x_dtype == MIQ_F32) return launch_sm_bwd_scores<float>(a, row_stats, grad_out, dx, (hipStream_t)stream);
    return launch_sm_bwd_scores<__half>(a, row_stats, grad_out, dx, (hipStream_t)stream);
}

extern "C" int miq_masked_softmax_bwd(const void* p, const void* grad_out, void* dx, int64_t R, int64_t C, float div, int dtype,
                                      void* stream) {
    if (R == 0 || C == 0) return 0;
    MIQ_CHECK_ARG(p && grad_out && dx && R > 0 && C > 0, "miq_masked_softmax_bwd: null pointer or bad shape");
    MIQ_CHECK_ARG(dtype == MIQ_F32 || dtype == MIQ_F16, "miq_masked_softmax_bwd: unsupported dtype %d", dtype);
    const int vec = dtype == MIQ_F16 ? 8 : 4;
    MIQ_CHECK_ARG(C % vec == 0 && C / vec <= kRowMaxPacks && C < (1ll << 31),
                  "miq_masked_softmax_bwd: C=%lld must be a multiple of %d and at most %d", (long long)C, vec, vec * kRowMaxPacks);
    MIQ_CHECK_ARG(is_aligned16(p) && is_aligned16(grad_out) && is_aligned16(dx), "miq_masked_softmax_bwd: pointers must be 16-byte aligned");
    MIQ_CHECK_ARG(div > 0.f, "miq_masked_softmax_bwd: divisor must be positive");
    if (dtype == MIQ_F32) return launch_sm_bwd<float>(p, grad_out, dx, R, (int)C, div, (hipStream_t)stream);
    return launch_sm_bwd<__half>(p, grad_out, dx, R, (int)C, div, (hipStream_t)stream);
}
