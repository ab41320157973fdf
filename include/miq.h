/*
 * libmiq — C ABI of the MI355X (gfx950) fake-quantization kernels.
 *
 * The reference has no FFI layer: its hot path is a set of torch.nn.Module classes that issue
 * dozens of eager PyTorch ops per call.  Each entry point below replaces the op sequence of one
 * reference method (cited per function, paths relative to the reference tree) with one or two
 * fused HIP kernels.  A binding needs nothing but device pointers, sizes and a stream:
 *
 *   - all pointers are DEVICE pointers owned by the caller (inputs are only read, outputs are
 *     fully overwritten); nothing is allocated, freed or retained by the library;
 *   - `dtype` is MIQ_F32 or MIQ_F16 and applies to every `void*` tensor of the call; fp16 tensors
 *     are computed in fp32 with a rounding to fp16 after every elementwise step, which reproduces
 *     PyTorch's per-op materialisation bit for bit;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work and never
 *     synchronise;
 *   - return value: 0 on success, a negative errno-style code otherwise (-22 bad argument,
 *     -5 launch failure); miq_last_error() returns the text for the calling thread.
 *
 * Row geometry shared by the row-wise ("per output channel") quantizers: the tensor is a row-major
 * [O, I] matrix.  With `group` == 0 every matrix row is one quantization row.  With `group` > 0
 * every matrix row is cut into ceil(I/group) groups; a short last group is zero-padded for the
 * min/max statistics only (reference quantizer.py:249-255, the padding never reaches the output).
 * Per-row vectors (up, low, scale, rzp, ...) have O*ceil(I/group) entries.
 */
#ifndef MIQ_H_
#define MIQ_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIQ_F32 0
#define MIQ_F16 1

/* flags of miq_rowquant_uniform_* */
#define MIQ_CLAMP_ASYM_SCALE 1 /* clamp the asymmetric scale to [1e-5, 1e4] (quantizer_mix_v2.py:140) */
/* Learnable weight clipping, every row-wise entry point: `up` / `low` hold the ACTIVATED clip factors
 * sigmoid(upbound_factor), sigmoid(lowbound_factor) in (0, 1) -- the tensors the reference forms at
 * quantize/quantizer.py:259-261 before it multiplies them with the row extremes -- and `d_up` / `d_low` receive the
 * gradients with respect to those activated values (the caller chains sigmoid' = s*(1-s), as autograd does through
 * torch.sigmoid).  Every result of the library is then a pure function of IEEE add / mul / div / round and bit-identical
 * to the reference's op sequence; a sigmoid evaluated inside the kernels would depend on the device math library's exp
 * in its last bit, and one bit of a clip factor moves the scale of a whole row.  (Interface versions below 300 took the
 * raw factors and a flag bit 1 for this form; the bit is ignored now.) */

/* major*10000 + minor*100 + patch of this interface; it grows when an entry point is added or a signature changes
 * (303: miq_rowquant_exp1_fwd_batch / _bwd_batch; 302: miq_act_static_init; 301: miq_gather_lastdim_batch; 300: up / low are the activated clip factors in every row-wise entry point; 200: row statistics output of miq_masked_softmax_fwd, miq_masked_softmax_bwd_scores, the *_fwd_stats norm kernels,
 * miq_act_quant_from_table, miq_gather_layernorm_bwd, miq_act_calib_quant_fwd_static) */
int miq_version(void);
const char* miq_last_error(void);

/*
 * Uniform affine fake quantization of matrix rows with optional learnable weight clipping.
 * Replaces UniformAffineQuantizer.per_token_dynamic_calibration + fake_quant
 * (quantize/quantizer.py:248-273 and :99-121); with `row_bits` != NULL it is
 * Mix2AffineQuantizer.forward (quantize/quantizer_mix_v2.py:116-146).
 *   x         [O, I]            input
 *   up, low   [rows] or NULL    activated LWC factors sigmoid(upbound_factor), sigmoid(lowbound_factor); both or neither
 *   row_bits  [rows] f32|NULL   per-row integer bit-widths; overrides n_bits
 *   out       [O, I]            dequantized result
 *   scale_out, rzp_out [rows]   scale and rounded zero point that were used
 *   codes_out [O, I] u8 | NULL  clamped integer levels (n_bits <= 8); 0 where the level is NaN
 */
int miq_rowquant_uniform_fwd(const void* x, const void* up, const void* low, const float* row_bits,
                             void* out, void* scale_out, void* rzp_out, void* codes_out,
                             int64_t O, int64_t I, int64_t group, int n_bits, int symmetric,
                             int flags, int dtype, void* stream);

/*
 * Backward of the above: what autograd derives from round_ste (quantize/quantizer.py:13-17)
 * through amin/amax, the clip product and the scale.  Statistics are recomputed from x.
 *   grad_out [O, I]; dx [O, I] or NULL (skip the input gradient); d_up, d_low [rows] (required
 *   iff up/low are given): gradients of the activated factors.
 */
int miq_rowquant_uniform_bwd(const void* grad_out, const void* x, const void* up, const void* low,
                             const float* row_bits, void* dx, void* d_up, void* d_low,
                             int64_t O, int64_t I, int64_t group, int n_bits, int symmetric,
                             int flags, int dtype, void* stream);

/*
 * The two entry points above for up to 16 matrices of ONE row width in one launch (whole rows, no
 * groups, no per-row bits, no codes).  The weight matrices of a block are quantized back to back
 * and are independent (QuantOPTDecoderLayer.smooth_and_quant_temporary, models/int_opt_layer.py:436-444):
 * batching them removes the ramp-up / tail of six short launches (5.0 -> 5.7 TB/s on Llama-2-7B).
 * x, up, low, out, ... are HOST arrays of n device pointers; O[i] = rows of matrix i.  up/low: NULL
 * table (or NULL first entry) = no clipping for any matrix.  Backward: dx NULL table (or NULL first
 * entry) = skip every input gradient.
 */
int miq_rowquant_uniform_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                   void* const* out, void* const* scale_out, void* const* rzp_out,
                                   const int64_t* O, int64_t I, int n_bits, int symmetric, int flags, int dtype,
                                   void* stream);
int miq_rowquant_uniform_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                   const void* const* low, void* const* dx, void* const* d_up, void* const* d_low,
                                   const int64_t* O, int64_t I, int n_bits, int symmetric, int flags, int dtype,
                                   void* stream);
/* The batched launches with per-row integer bit-widths: row_bits[i] holds O_i floats, as the row_bits argument of
 * miq_rowquant_uniform_fwd (Mix2AffineQuantizer.forward, quantize/quantizer_mix_v2.py:116-146, which clamps the scale:
 * pass flags bit 0).  Results are those of the single-matrix entry points bit for bit. */
int miq_rowquant_mix2_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                                const float* const* row_bits, void* const* out, void* const* scale_out,
                                void* const* rzp_out, const int64_t* O, int64_t I, int symmetric, int flags, int dtype,
                                void* stream);
int miq_rowquant_mix2_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                const void* const* low, const float* const* row_bits, void* const* dx, void* const* d_up,
                                void* const* d_low, const int64_t* O, int64_t I, int symmetric, int flags, int dtype,
                                void* stream);

/*
 * Static activation quantisation over the channel (last) axis of a row-major [R, C] slab.
 * Replaces UniformAffineQuantizer.per_cluster_minmax_calibration / per_tensor_minmax_calibration /
 * calibration (quantize/quantizer.py:157-238): one streaming pass for the statistics instead of
 * two reductions plus a Python loop of four kernels per cluster.
 *
 * miq_colstat_chunks   number of row chunks the statistics pass will use for this shape
 * miq_colstat_minmax   per-channel min/max of every row chunk -> part_min/part_max [nchunk, C] f32
 * miq_cluster_scale    fold the chunks (taking column idx[c] for output channel c when idx is
 *                      given), combine with the cached statistics (metric 0: running min/max,
 *                      1: 0.99/0.01 EMA; cached_* may be NULL), merge channels into the contiguous
 *                      clusters described by starts[0..n_clusters] (prefix sums; channels from
 *                      `covered` on keep per-channel statistics), then
 *                      scale = clamp((max-min)/(2^n-1), 1e-5, 1e4), zp = clamp(-min/scale, +-1e4),
 *                      rzp = round(zp).  part_min/part_max must have (nchunk+1)*C floats: row
 *                      `nchunk` receives the per-channel statistics before the merge (kept by the
 *                      caller for the backward pass).  Outputs are [C] vectors of `dtype`.
 *                      idx entries must lie in [0, C).
 */
int miq_colstat_chunks(int64_t R, int64_t C, int dtype);
int miq_colstat_minmax(const void* x, int64_t R, int64_t C, float* part_min, float* part_max,
                       int nchunk, int dtype, void* stream);
int miq_cluster_scale(float* part_min, float* part_max, int nchunk, int64_t C, const int* idx,
                      const void* cached_min, const void* cached_max, int metric, const int* starts,
                      int n_clusters, int64_t covered, int n_bits, void* cmin_out, void* cmax_out,
                      void* scale_out, void* zp_out, void* rzp_out, int dtype, void* stream);

/*
 * Quantize-dequantize with per-channel scale / rounded zero point vectors [C]
 * (UniformAffineQuantizer.fake_quant with the calibrated buffers, quantize/quantizer.py:99-121,140).
 * codes_out: optional u8 [R, C] clamped levels (n_bits <= 8).
 */
int miq_colquant_fwd(const void* x, const void* scale, const void* rzp, void* out, void* codes_out,
                     int64_t R, int64_t C, int n_bits, int dtype, void* stream);

/*
 * Backward of calibration + fake_quant as autograd derives it from round_ste
 * (quantize/quantizer.py:13-17).  With chan_min/chan_max == NULL the scales are constants (eval
 * mode) and dx = grad_out * [level inside the grid].  Otherwise the scale gradient is summed per
 * cluster and routed to the extreme elements: chan_min/chan_max are the per-channel statistics
 * (row `nchunk` of miq_cluster_scale's partial buffers), kmin/kmax the merged [C] statistics,
 * workspace has (3*nchunk + 5)*C floats with nchunk = miq_colstat_chunks(R, C, dtype).
 * x and grad_out are read once (dx receives the direct path in the same pass); the routed shares
 * are then added in place to the few elements that attain a cluster extreme.
 * flat_ties: how tied extremes share the scale gradient.  0: per channel, then per row (the statistics of a cluster are
 * an amax over channels of a per-channel amax over rows, quantizer.py:190-210); 1: evenly among all tied elements of the
 * cluster (per-tensor statistics are one reduction over all elements: torch.min(x) / torch.max(x), quantizer.py:222-223,
 * and amin / amax over every axis when no cluster axis is set).
 * phase 0 runs everything.  Data-parallel callers that quantize with GLOBAL-batch statistics split
 * the call: phase 1 writes the direct path into dx and leaves the per-channel sums [dS, #max, #min]
 * in workspace[3*nchunk*C : (3*nchunk + 3)*C] (to be summed over ranks by the caller), phase 2
 * derives the shares from them and corrects dx.
 */
int miq_colquant_bwd(const void* grad_out, const void* x, const void* scale, const void* rzp,
                     const float* chan_min, const float* chan_max, const void* kmin, const void* kmax,
                     const int* starts, int n_clusters, int64_t covered, float* workspace, int nchunk,
                     void* dx, int64_t R, int64_t C, int n_bits, int phase, int flat_ties, int dtype, void* stream);

/*
 * The calibrating static quantizer in one call: the whole of UniformAffineQuantizer.forward in
 * calibration mode (quantize/quantizer.py:124-142 -> :157-238 -> :99-121).  Exactly
 *   miq_colstat_minmax -> miq_cluster_scale (idx = NULL) -> miq_colquant_fwd
 * with nchunk = miq_colstat_chunks(R, C, dtype); arguments as documented there.
 * workspace: miq_act_workspace(R, C, dtype) floats = two planes of [nchunk + 1, C]; on return row
 * nchunk of plane 0 / plane 1 holds the per-channel min / max before the cluster merge (the chan_min /
 * chan_max inputs of miq_colquant_bwd).
 */
int64_t miq_act_workspace(int64_t R, int64_t C, int dtype);
int miq_act_calib_quant_fwd(const void* x, const void* cached_min, const void* cached_max, int metric,
                            const int* starts, int n_clusters, int64_t covered, int n_bits, float* workspace,
                            void* out, void* codes_out, void* cmin_out, void* cmax_out, void* scale_out,
                            void* zp_out, void* rzp_out, int64_t R, int64_t C, int dtype, void* stream);

/*
 * The calibrating static quantizer with FRESH statistics (no cached values: what every forward of a training step
 * sees, because omniquant() frees the cached statistics after each optimizer step, quantize/omniquant.py:530-535) for
 * up to 8 slabs of one shape, dtype and cluster layout per call — the q/k/v projections' inputs and outputs and the
 * gate/up inputs of a decoder block are quantized side by side (models/int_llama_layer.py:110-116, :139-154).
 * Forward: two launches (statistics, quantize) instead of four.  The row-chunk blocks of the statistics kernel merge
 * their per-cluster minima / maxima with integer atomic maxima on order-preserving keys (exact, order independent), the
 * quantize kernel derives scale / zero point from that table itself.  Results are bit-identical to
 * miq_act_calib_quant_fwd with cached_min == NULL.
 *   x, out        host arrays of n device pointers to [R, C] slabs; codes_out: NULL or n pointers (entries may be NULL)
 *   stats_out     n device pointers to [5, C] blocks of `dtype`: merged min, merged max, scale, clamped zero point,
 *                 rounded zero point (rows 0 / 1 are what the quantizer caches, rows 2 / 4 its scale / round_zero_point)
 *   cluster_size  > 0: n_clusters clusters of that many channels each (cluster_size * n_clusters == covered);
 *                 0: unequal clusters, described by slot_of [C] int32 on the device: cluster id for a channel inside a
 *                 cluster, n_clusters + (c - covered) for a channel c >= covered (those keep per-channel statistics)
 *   workspace     miq_act_fresh_workspace(n, C, n_clusters, covered) BYTES, 128-byte aligned, zero-filled once before
 *                 its first use and afterwards only passed to this function (it carries a call counter; 2^32 calls).
 *                 Calls that share a workspace must be ordered on one stream.
 * Backward (what autograd derives, as miq_colquant_bwd): stats = the [5, C] blocks of the forward; the extreme elements
 * are found by comparing with the merged statistics.  With equal clusters of at most 1024 channels that cover every
 * channel the chunk fold, the shares and the (usually sparse) correction run in one launch: 2 launches in all when a
 * cluster's slab R * cluster_size has at most 2^20 elements (its block then also handles mass ties), else 3, the last
 * one a no-op unless a cluster has mass ties; otherwise the four launches of miq_colquant_bwd; same bits either way.
 *   starts        [n_clusters + 1] int32 prefix sums of the cluster sizes on the device
 *   flat_ties     as for miq_colquant_bwd
 *   workspace     miq_act_fresh_bwd_workspace(n, R, C, dtype) BYTES, 16-byte aligned, no initialisation needed
 */
int64_t miq_act_fresh_workspace(int n, int64_t C, int n_clusters, int64_t covered);
int miq_act_fresh_quant_fwd(int n, const void* const* x, int64_t cluster_size, const int* slot_of, int n_clusters,
                            int64_t covered, int n_bits, void* workspace, void* const* out, void* const* codes_out,
                            void* const* stats_out, int64_t R, int64_t C, int dtype, void* stream);
int64_t miq_act_fresh_bwd_workspace(int n, int64_t R, int64_t C, int dtype);
int miq_act_fresh_quant_bwd(int n, const void* const* grad_out, const void* const* x, const void* const* stats,
                            int64_t cluster_size, const int* starts, int n_clusters, int64_t covered, int n_bits,
                            void* workspace, void* const* dx, int64_t R, int64_t C, int flat_ties, int dtype, void* stream);

/*
 * miq_act_calib_quant_fwd for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same kernels and
 * results as miq_act_calib_quant_fwd; the partial statistics live in a device array inside the library
 * (miq_act_static_capacity() floats = 8 MiB on every GPU the library runs on, held until the process ends)
 * instead of a caller workspace, and the call makes no hipMalloc / hipFree.  Slabs whose
 * miq_act_workspace(R, C, dtype) exceeds the capacity are rejected with -22.  chan_min_out / chan_max_out:
 * optional fp32 [C], receive the per-channel statistics before the cluster merge (the chan_min / chan_max inputs
 * of miq_colquant_bwd).  The tensors' device must be the current device.  All calls on one GPU share that GPU's
 * array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream are safe.
 * The device address of the array is looked up once per GPU (hipGetSymbolAddress: no allocation, but not a stream
 * operation either): by miq_act_static_init() with that GPU current, at set-up, or else by the first call there, which
 * must then not run under stream capture.  miq_act_static_init returns 0 or -5.
 */
int64_t miq_act_static_capacity(void);
int miq_act_static_init(void);
int miq_act_calib_quant_fwd_static(const void* x, const void* cached_min, const void* cached_max, int metric,
                                   const int* starts, int n_clusters, int64_t covered, int n_bits, void* out,
                                   void* codes_out, void* cmin_out, void* cmax_out, void* scale_out, void* zp_out,
                                   void* rzp_out, float* chan_min_out, float* chan_max_out, int64_t R, int64_t C,
                                   int dtype, void* stream);

/*
 * RPTQ channel reorder: out[r, c] = x[r, idx[c]] (torch.index_select on the last axis, as in
 * OmniLayerNorm.forward, quantize/omni_norm.py:36-40).  Rows are staged through LDS so that both
 * the global loads and the global stores stay coalesced.  idx: int32 [C], entries in [0, C).
 */
int miq_gather_lastdim(const void* x, const int* idx, void* out, int64_t R, int64_t C, int dtype,
                       void* stream);
/* The same for up to 8 tensors of one shape and dtype in one launch, each with its own index (x, idx, out: HOST arrays
 * of n device pointers): the two reordered layer norms of a block gather tensors of one shape
 * (R1 / R4 of quantize/reorder_utils.py:6-58), and a 16.8 MB gather alone is bound by the launch, not by memory. */
int miq_gather_lastdim_batch(int n, const void* const* x, const int* const* idx, void* const* out, int64_t R, int64_t C,
                             int dtype, void* stream);

/*
 * RPTQ reorder gather fused into LayerNorm, forward (OmniLayerNorm.forward with a reorder index,
 * quantize/omni_norm.py:29-42: index_select on the last axis, then layer_norm):
 *   out[r, c] = (x[r, idx[c]] - mean_r) * rstd_r * weight[c] + bias[c]
 * with mean / biased variance over the C channels of row r in fp32 and rstd = 1/sqrt(var + eps).
 * idx NULL = plain LayerNorm; weight / bias may be NULL.  x, out: [R, C] of x_dtype / out_dtype
 * (fp16 in -> fp32 out is what autocast produces); weight, bias: [C] of w_dtype; mean_out, rstd_out:
 * optional fp32 [R] (the saved statistics of a LayerNorm backward).  C % 4 == 0.
 */
int miq_gather_layernorm_fwd(const void* x, const int* idx, const void* weight, const void* bias, void* out,
                             float* mean_out, float* rstd_out, int64_t R, int64_t C, float eps, int x_dtype,
                             int w_dtype, int out_dtype, void* stream);

/*
 * The norm in front of a static activation quantizer that calibrates on fresh statistics (the layer norms feeding the
 * q/k/v and fc1 / gate / up projections: quantize/omni_norm.py:29-42,60-71 followed by
 * UniformAffineQuantizer.forward in calibration mode, quantize/quantizer.py:124-142,170-214), without the statistics
 * pass over the norm's output:
 *   miq_gather_layernorm_fwd_stats / miq_rmsnorm_fwd_stats   the norm as above; every row additionally folds the minima
 *       and maxima of the values it STORES (after rounding to the output dtype) into the statistics table that
 *       miq_act_fresh_quant_fwd keeps in `stat_workspace` (slab 0), i.e. exactly what that function's statistics kernel
 *       would compute from `out`;
 *   miq_act_quant_from_table   the quantize half of miq_act_fresh_quant_fwd for one slab whose statistics are already
 *       in the table: out / codes_out / stats_out as there.
 * The two calls must follow each other on one stream with nothing else using the workspace in between.  Cluster
 * layout arguments (cluster_size, slot_of, n_clusters, covered) as for miq_act_fresh_quant_fwd (one cluster of C
 * channels = per-tensor statistics); stat_workspace: miq_act_fresh_workspace(1, C, n_clusters, covered) bytes, as there.
 *
 * miq_gather_layernorm_bwd: backward of miq_gather_layernorm_fwd (autograd of index_select + layer_norm) from the saved
 * mean / rstd.  With xg[c] = x[idx[c]], u = (xg - mean) * rstd, gw = grad_out * weight:
 *   dx[r, idx[c]] = rstd * (gw - mean_c(gw) - u * mean_c(gw * u));  d_weight[c] = sum_r grad_out * u;  d_bias[c] = sum_r grad_out
 * grad_out: [R, C] of g_dtype; dx: [R, C] of x_dtype; d_weight, d_bias: fp32 [C]; each of dx / d_weight / d_bias may be
 * NULL.  workspace: miq_gather_layernorm_bwd_workspace(R, C) floats, needed for d_weight / d_bias.
 */
int miq_gather_layernorm_fwd_stats(const void* x, const int* idx, const void* weight, const void* bias, void* out,
                                   float* mean_out, float* rstd_out, int64_t R, int64_t C, float eps, int x_dtype,
                                   int w_dtype, int out_dtype, int64_t cluster_size, const int* slot_of, int n_clusters,
                                   int64_t covered, void* stat_workspace, void* stream);
int miq_rmsnorm_fwd_stats(const void* x, const float* weight, const float* bias, void* out, float* rstd_out, int64_t R,
                          int64_t C, float eps, int dtype, int64_t cluster_size, const int* slot_of, int n_clusters,
                          int64_t covered, void* stat_workspace, void* stream);
int miq_act_quant_from_table(const void* x, int64_t cluster_size, const int* slot_of, int n_clusters, int64_t covered,
                             int n_bits, void* workspace, void* out, void* codes_out, void* stats_out, int64_t R, int64_t C,
                             int dtype, void* stream);
int64_t miq_gather_layernorm_bwd_workspace(int64_t R, int64_t C);
int miq_gather_layernorm_bwd(const void* grad_out, const void* x, const int* idx, const void* weight, const float* mean,
                             const float* rstd, void* dx, float* d_weight, float* d_bias, float* workspace, int64_t R,
                             int64_t C, int x_dtype, int w_dtype, int g_dtype, void* stream);

/*
 * Exponential/log weight quantizer v2 (levels 0, +-1, +-2, ..., +-2^(2^(n-1)-2)), n_bits in [2, 4].
 * Replaces ExpAffineQuantizer.per_channel_dynamic_calibration + FakeQuant.forward / backward
 * (quantize/quantizer_exp_v2.py:186-204, :21-62, :65-105 — the backward is the reference's own
 * hand-written rule dx = g * level/|x/s|, not the true derivative).  flags: reserved, pass 0.
 */
int miq_rowquant_exp2_fwd(const void* x, const void* up, const void* low, void* out, void* scale_out,
                          int64_t O, int64_t I, int64_t group, int n_bits, int flags, int dtype, void* stream);
int miq_rowquant_exp2_bwd(const void* grad_out, const void* x, const void* up, const void* low, void* dx,
                          void* d_up, void* d_low, int64_t O, int64_t I, int64_t group, int n_bits,
                          int flags, int dtype, void* stream);
/* The same over n (1..16) matrices [O_i, I] of one row width and dtype in one launch each way (whole rows, no groups;
 * pointer tables like miq_rowquant_uniform_fwd_batch; up / low, dx and d_up / d_low for all matrices or for none).
 * Results are those of the single-matrix entry points bit for bit.  No counterpart in the reference, which quantizes
 * the weight matrices of a block one after the other (quantize/int_linear.py:72-87). */
int miq_rowquant_exp2_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low, void* const* out,
                                void* const* scale_out, const int64_t* O, int64_t I, int n_bits, int flags, int dtype,
                                void* stream);
int miq_rowquant_exp2_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                const void* const* low, void* const* dx, void* const* d_up, void* const* d_low,
                                const int64_t* O, int64_t I, int n_bits, int flags, int dtype, void* stream);

/*
 * Exponential/log weight quantizer v1 (levels +-2^e, e in [-(2^(n-1)-1), 0], no zero level).
 * Replaces ExpAffineQuantizer.per_channel_dynamic_calibration + fake_quant and their autograd
 * (quantize/quantizer_exp.py:143-166, :92-124).  symmetric=0 quantizes around the row midpoint,
 * which is returned in mid_out [rows] (may be NULL when symmetric).
 */
int miq_rowquant_exp1_fwd(const void* x, const void* up, const void* low, void* out, void* scale_out,
                          void* mid_out, int64_t O, int64_t I, int64_t group, int n_bits, int symmetric,
                          int flags, int dtype, void* stream);
int miq_rowquant_exp1_bwd(const void* grad_out, const void* x, const void* up, const void* low, void* dx,
                          void* d_up, void* d_low, int64_t O, int64_t I, int64_t group, int n_bits,
                          int symmetric, int flags, int dtype, void* stream);
/* The same over n (1..16) matrices [O_i, I] of one row width and dtype in one launch each way (whole rows, no groups;
 * pointer tables like miq_rowquant_exp2_fwd_batch; mid_out may be NULL when symmetric).  Results are those of the
 * single-matrix entry points bit for bit.  No counterpart in the reference, which quantizes the weight matrices of a
 * block one after the other (quantize/int_linear.py:72-87). */
int miq_rowquant_exp1_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low, void* const* out,
                                void* const* scale_out, void* const* mid_out, const int64_t* O, int64_t I, int n_bits,
                                int symmetric, int flags, int dtype, void* stream);
int miq_rowquant_exp1_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                                const void* const* low, void* const* dx, void* const* d_up, void* const* d_low,
                                const int64_t* O, int64_t I, int n_bits, int symmetric, int flags, int dtype, void* stream);


/*
 * Learnable per-row bit-width quantizer: blend of the floor(n)- and ceil(n)-bit uniform grids.
 * Replaces MixAffineQuantizer.forward / fake_quant and their autograd
 * (quantize/quantizer_mix.py:143-191, :118-140).  float32 only.  n_bits [rows] must already be
 * clamped to [2, 8] (the reference clamps the parameter in place before use, :145);
 * hard != 0 rounds n + threshold first (:147-148).  d_nbits [rows] may be NULL.
 */
int miq_rowquant_mix_fwd(const void* x, const void* up, const void* low, const float* n_bits, void* out,
                         int64_t O, int64_t I, int64_t group, int symmetric, int hard, float threshold,
                         int flags, int dtype, void* stream);
int miq_rowquant_mix_bwd(const void* grad_out, const void* x, const void* up, const void* low,
                         const float* n_bits, void* dx, void* d_up, void* d_low, void* d_nbits, int64_t O,
                         int64_t I, int64_t group, int symmetric, int hard, float threshold, int flags,
                         int dtype, void* stream);
/* The same over n (1..16) matrices [O_i, I] of one row width in one launch each way (whole rows, no groups; pointer
 * tables like miq_rowquant_uniform_fwd_batch, n_bits[i] / d_nbits[i] holding O_i entries; up / low, dx and d_nbits for
 * all matrices or for none).  Results are those of the single-matrix entry points bit for bit.  No counterpart in
 * the reference, which quantizes the weight matrices of a layer one after the other (quantize/int_linear.py:72-87). */
int miq_rowquant_mix_fwd_batch(int n, const void* const* x, const void* const* up, const void* const* low,
                               const float* const* n_bits, void* const* out, const int64_t* O, int64_t I, int symmetric,
                               int hard, float threshold, int flags, int dtype, void* stream);
int miq_rowquant_mix_bwd_batch(int n, const void* const* grad_out, const void* const* x, const void* const* up,
                               const void* const* low, const float* const* n_bits, void* const* dx, void* const* d_up,
                               void* const* d_low, void* const* d_nbits, const int64_t* O, int64_t I, int symmetric,
                               int hard, float threshold, int flags, int dtype, void* stream);

/*
 * RPTQ reorder index: per head, ascending stable argsort of (xmax - xmin), offset by the head start
 * (peg_tensor_calc_reorder_index, quantize/reorder_utils.py:111-147).  xmax/xmin: f32 [C] on the
 * device, index_out: int64 [C].  Ties are ordered by channel id; NaN ranges sort last.
 */
int miq_reorder_index(const float* xmax, const float* xmin, int64_t C, int n_heads, int64_t* index_out,
                      void* stream);

/*
 * LET-fused row-wise uniform weight quantizer (fp32 weights, whole rows, I % 4 == 0, I <= 16384).
 * Forward: out = fake_quant_rows(t) with t = (w * cs[None,:]) / rs[:,None]  (rs_div = 1) or
 * (w * cs[None,:]) * rs[:,None] (rs_div = 0), the smoothed weight that smooth_ln_fcs_temporary /
 * smooth_fc_fc_temporary / smooth_q_k_temporary (models/transformation.py:24-69) materialise
 * before UniformAffineQuantizer.forward (quantize/quantizer.py:124-142); same IEEE operations in the
 * same order, so `out`, `scale_out`, `rzp_out` are bit-identical to that chain (up / low: activated
 * clip factors, as everywhere).  cs, rs, shift, up/low may each be NULL.
 * out_dtype MIQ_F16 stores the fp32 result rounded to fp16 (what the autocast of the consuming
 * linear would produce from the fp32 tensor).
 * With `shift`, wshift_out[o] = sum_i w[o,i] * shift[i] (the `fc.weight @ shifts` term of the
 * smoothed bias; fp32 accumulation in kernel order).
 * Backward: given grad_out = dL/d out (fp32 or fp16, grad_dtype) and g_wshift = dL/d wshift, returns
 * d_cs [I], d_rs [O], d_shift [I], d_up / d_low [O] (each optional: pass NULL to skip) — what
 * autograd derives through the same chain.  workspace: miq_let_workspace(O, I) floats, needed when
 * d_cs or d_shift is requested.
 */
int64_t miq_let_workspace(int64_t O, int64_t I);
int miq_let_rowquant_fwd(const float* w, const float* cs, const float* rs, int rs_div, const float* shift,
                         const float* up, const float* low, void* out, float* scale_out, float* rzp_out,
                         float* wshift_out, int64_t O, int64_t I, int n_bits, int symmetric, int flags,
                         int out_dtype, void* stream);
int miq_let_rowquant_bwd(const void* grad_out, const float* g_wshift, const float* w, const float* cs,
                         const float* rs, int rs_div, const float* shift, const float* up, const float* low,
                         float* d_cs, float* d_rs, float* d_shift, float* d_up, float* d_low, float* workspace,
                         int64_t O, int64_t I, int n_bits, int symmetric, int flags, int grad_dtype, void* stream);

/*
 * Sensitivity statistic behind the per-output-channel bit allocation (get_grads.py:201-205):
 * acc[o] += mean over i of |g[o, i]| for a row-major [O, I] weight gradient; acc is fp32 [O] and is
 * accumulated over calls (the caller zeroes it once per layer).
 */
int miq_row_absmean_accum(const void* g, float* acc, int64_t O, int64_t I, int dtype, void* stream);

/*
 * The attention-score chain of the decoder blocks between the quantized q·k^T and p·v matmuls
 * (models/int_llama_layer.py:139-154, models/int_opt_layer.py:199-221):
 *   p = softmax(max(x / div + mask, finfo.min), dim=-1, dtype=float32).to(x dtype)
 * in one pass.  x, out: [R, C] scores / probabilities (fp32 or fp16), R = batch*heads*q_len rows of C keys;
 * mask: [R / rows_per_batch * q_len, C] fp32 or fp16 (the [B, 1, q_len, C] attention mask: the
 * rows_per_batch = heads*q_len score rows of one batch entry share its q_len mask rows) or NULL (then no
 * floor either); div = sqrt(head_dim) for Llama, 1 for OPT (which scales q before the matmul).  x / div is
 * formed as x * (1/div) like ATen's tensor / scalar on a GPU and rounded to x's dtype.  C a multiple of 4 (fp32)
 * or 8 (fp16).
 * row_stats_out: NULL, or fp32 [R, 2] receiving each row's maximum and reciprocal sum of exponentials.
 * Backward: dx = p * (grad_out - sum_j grad_out_j p_j) / div.  PyTorch's softmax(dtype=float32).to(fp16) keeps the
 * fp32 probabilities for this, so there are two forms: miq_masked_softmax_bwd takes probabilities (exact for fp32
 * scores, where `out` is that tensor); miq_masked_softmax_bwd_scores takes the forward's x, mask and row_stats
 * and rebuilds the fp32 probabilities with the forward's own instructions (the form for fp16 scores: same
 * traffic, no 2^-11 rounding of p in every gradient element).
 */
int miq_masked_softmax_fwd(const void* x, const void* mask, void* out, float* row_stats_out, int64_t R, int64_t C,
                           int64_t rows_per_batch, int64_t q_len, float div, int x_dtype, int mask_dtype, void* stream);
int miq_masked_softmax_bwd(const void* p, const void* grad_out, void* dx, int64_t R, int64_t C, float div, int dtype,
                           void* stream);
int miq_masked_softmax_bwd_scores(const void* x, const void* mask, const float* row_stats, const void* grad_out, void* dx,
                                  int64_t R, int64_t C, int64_t rows_per_batch, int64_t q_len, float div, int x_dtype,
                                  int mask_dtype, void* stream);

/*
 * OmniLlamaRMSNorm (quantize/omni_norm.py:60-71), the norm in front of the quantized q/k/v and up/gate
 * projections of the Llama block, in one pass per direction:
 *   rstd = rsqrt(mean(x.float()^2, -1) + eps);  out = (weight * (x * rstd) [+ bias]).to(x dtype)
 * x, out: [R, C] fp32 or fp16; weight, bias (may be NULL): fp32 [C] (the caller converts fp16 vectors, which is
 * what type promotion does in the reference); rstd_out: fp32 [R] saved for the backward, may be NULL.
 * C a multiple of 4.
 * Backward: dx = rstd * (g*w - u * mean(g*w*u)) with u = x*rstd; d_weight[c] = sum_r g*u, d_bias[c] = sum_r g
 * (fp32 [C]).  dx, d_weight, d_bias may each be NULL; workspace: miq_rmsnorm_workspace(R, C) floats, needed
 * when d_weight or d_bias is asked for.
 */
int64_t miq_rmsnorm_workspace(int64_t R, int64_t C);
int miq_rmsnorm_fwd(const void* x, const float* weight, const float* bias, void* out, float* rstd_out, int64_t R,
                    int64_t C, float eps, int dtype, void* stream);
int miq_rmsnorm_bwd(const void* grad_out, const void* x, const float* weight, const float* rstd, void* dx,
                    float* d_weight, float* d_bias, float* workspace, int64_t R, int64_t C, int dtype, void* stream);

/*
 * Column sums out[c] = sum_r x[r, c] (fp32 [C]) of a row-major [R, C] fp32 / fp16 matrix: the bias gradient of the
 * F.linear inside QuantLinear.forward (quantize/int_linear.py:86), which autograd computes with a reduction over
 * the slow dimension.  workspace: miq_col_sum_workspace(R, C) floats (may be NULL when that is <= C, i.e. one row
 * chunk).  Deterministic summation order.  round_dtype = MIQ_F16 rounds every sum to fp16 before it is stored as fp32
 * (what an fp16 reduction returns, overflow to inf included: the loss scaler relies on it); MIQ_F32 stores the sum.
 */
int64_t miq_col_sum_workspace(int64_t R, int64_t C);
int miq_col_sum(const void* x, float* out, float* workspace, int64_t R, int64_t C, int dtype, int round_dtype,
                void* stream);

/*
 * Rotary position embedding of the Llama query / key states in front of the q.k^T quantizers
 * (models/int_llama_layer.py:124-125, transformers' apply_rotary_pos_emb):
 *   out = x * cos + rotate_half(x) * sin,  rotate_half(x) = cat(-x[..., D/2:], x[..., :D/2])
 * for a logical [B, H, T, D] tensor given by element strides (last stride 1), so the transposed view of the
 * projection output is read in place and `out` can be written contiguous; cos / sin: [table batches, T, D] of the
 * same dtype (table_batch_stride = 0 shares one table).  Every product and the sum are rounded to the dtype like
 * PyTorch's intermediates (bit-exact).  backward != 0 applies the transposed rotation to a gradient:
 *   out_lo = x_lo cos_lo + x_hi sin_hi,  out_hi = x_hi cos_hi - x_lo sin_lo.
 * D a multiple of 8 (fp32) or 16 (fp16); strides multiples of 4 / 8 elements.
 */
int miq_rotary(const void* x, const void* cos, const void* sin, void* out, int64_t B, int64_t H, int64_t T, int64_t D,
               int64_t xs_b, int64_t xs_h, int64_t xs_t, int64_t os_b, int64_t os_h, int64_t os_t,
               int64_t table_batch_stride, int backward, int dtype, void* stream);

/*
 * truncate_number (models/transformation.py:4-20) applied in place to up to 16 vectors in one launch:
 * x[|x| < threshold] = sign(x) * threshold (zeros stay zero, NaN stays NaN).  The decoder layers call it on every
 * smoothing scale before each training step (models/int_llama_layer.py:279-285, models/int_opt_layer.py:390-396).
 * ptrs / sizes are HOST arrays of n device pointers / element counts, all of `dtype`.
 */
int miq_truncate_inplace(void* const* ptrs, const int64_t* sizes, int n, float threshold, int dtype, void* stream);

/*
 * The calibration loss loss_func(fp_out, quant_out) = torch.nn.MSELoss() (quantize/omniquant.py:339, 511-513) on
 * tensors of `dtype` without the fp32 copies autocast makes of fp16 operands:
 *   miq_mse_fwd  out[0] = mean((float(a) - float(b))^2); workspace: miq_mse_workspace() floats
 *   miq_mse_bwd  da = dtype(((2/n) * (a - b)) * grad_loss[0]), db = the same with (b - a): ATen's
 *                mse_loss_backward in fp32 followed by the cast back that autocast's backward applies
 *                (bit-identical gradients); grad_loss is a DEVICE scalar; da / db may each be NULL.
 * a, b contiguous with n elements, 16-byte aligned.
 */
int64_t miq_mse_workspace(void);
int miq_mse_fwd(const void* a, const void* b, float* out, float* workspace, int64_t n, int dtype, void* stream);
int miq_mse_bwd(const void* a, const void* b, const float* grad_loss, void* da, void* db, int64_t n, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MIQ_H_ */
