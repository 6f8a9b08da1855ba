/*
 * dsrl_hip.h - C ABI of libdsrl_hip.so: the MI355X (gfx950) arithmetic of the DSRL stage-3 training hot path.
 *
 * The reference (sanje2v/DualSuperResLearningForSemSeg) has no FFI of its own: every operation below
 * replaces the torch.nn call the reference makes at the cited file:line (paths under /root/reference).
 * The host side (dualsuperreslearningforsemseg_amd/functional.py) binds these with ctypes and keeps the
 * reference's nn.Module surface above them; INTEGRATION.md shows the binding a reference maintainer adds.
 *
 * Conventions
 *   - fp32 only. Activations are "pixel-major": element (n,h,w,c) of a logical NCHW tensor lives at
 *     ((n*H + h)*W + w)*ld + c   (= torch.channels_last); `ld` >= C is the pixel stride in floats, so a
 *     channel slice of a concatenation buffer is addressed by (ptr + first_channel, ld = total channels).
 *   - Conv weights are [K][R][S][C] (= torch (K,C,R,S) in channels_last), ConvTranspose weights are the
 *     plain torch layout (Cin,Cout,2,2).
 *   - Every function enqueues on `stream` (a hipStream_t) and returns immediately; no allocation, no host
 *     synchronisation (the two read-out functions dsrl_prof_read and dsrl_bn_fused_barrier_timeouts excepted); re-entrant from
 *     the autograd worker thread. Process-wide state is limited to four settings (dsrl_conv_precision, dsrl_bn_fused_max_blocks,
 *     dsrl_prof_enable, dsrl_rng_bind_device_key) and the self-resetting arrival counter of the fused BatchNorm kernels'
 *     device-wide barrier. The fused BatchNorm launches of one device share that counter, so they must not overlap each other: the
 *     first one pins its stream, launches on any other stream (outside graph capture) take the three-kernel path.
 *   - The caller owns all buffers including `ws` (workspace, >= the matching *_workspace_bytes()).
 *   - Return 0 on success, a negative DSRL_E_* otherwise; dsrl_last_error() has the message (thread-local).
 */
#ifndef DSRL_HIP_H
#define DSRL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSRL_ABI_VERSION 1

#define DSRL_OK 0
#define DSRL_E_BADARG (-1)      /* inconsistent shape / null pointer / alignment                      */
#define DSRL_E_UNSUPPORTED (-2) /* shape outside what the kernels implement                            */
#define DSRL_E_WORKSPACE (-3)   /* workspace too small                                                 */
#define DSRL_E_LAUNCH (-4)      /* hipGetLastError() after launch                                      */

typedef void* dsrl_stream_t; /* hipStream_t */

int dsrl_version(void);
const char* dsrl_last_error(void);
/* fills cu_count; returns 0 only on a gfx950 device */
int dsrl_device_check(int* cu_count);

/* ------------------------------------------------------------------------------------------------
 * conv2d: implicit GEMM on the matrix cores; fp32 in / out / accumulate, products formed as dsrl_conv_precision selects
 * (default: "f16x3" split-precision fp16 MFMAs = fp32-equivalent in every pass; mode 0: v_mfma_f32_32x32x2_f32).
 * replaces nn.Conv2d forward/backward at ASPP.py:10-15,19; DSRL.py:19-23,34-38,42-46,50,78-83 and the
 * ResNet101.py convolutions; x (N,H,W,C) -> y (N,Ho,Wo,K).
 * ---------------------------------------------------------------------------------------------- */
size_t dsrl_conv2d_fwd_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int dsrl_conv2d_fwd(const float* x, int ldx, const float* w, const float* bias /*nullable*/, float* y, int ldy,
                    int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                    void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* The same conv that additionally leaves BatchNorm partials of its output in `stats`: [3][parts][K] floats = (count, mean, centred
 * second moment) per (block of output rows, output channel), parts = dsrl_conv2d_fwd_stats_parts(shape) for the current arithmetic mode
 * (0: this launch cannot provide them - exact-fp32 kernels, some split-K plans, more than 4096 row blocks). The BatchNorm that follows
 * (ASPP.py:19-20, DSRL.py:22-24,36-40,44-48, every ResNet block) then skips its own statistics pass: dsrl_bn_train_fwd_from_stats. */
int dsrl_conv2d_fwd_stats_parts(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int dsrl_conv2d_fwd_stats(const float* x, int ldx, const float* w, const float* bias /*nullable*/, float* y, int ldy,
                          int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                          void* ws, size_t ws_bytes, float* stats, int stats_parts, dsrl_stream_t stream);
/* dx (N,H,W,C) from dy (N,Ho,Wo,K). The kernel reads the filter as wt[c][tap][k] (k padded to a multiple of 4): pass `wt` from
 * dsrl_conv2d_transpose_filter (e.g. built during the forward pass on another stream) or NULL to have it built here in `ws`. */
size_t dsrl_conv2d_transposed_filter_floats(int C, int K, int R, int S);
int dsrl_conv2d_transpose_filter(const float* w, float* wt, int C, int K, int R, int S, dsrl_stream_t stream);
/* The same for n filters in one launch (once per training step instead of once per layer): table is a DEVICE array of n rows of ten
 * int64 {w pointer, wt pointer, K, Kp = K rounded up to 4, R*S, C, index of the row's first 32x32 tile, ceil(C/32), amax pointer, 0}, rows
 * ordered by first tile; total_tiles = sum over rows of R*S * ceil(C/32) * ceil(Kp/32). A non-zero amax pointer names an amax record
 * (DSRL_AMAX_WORDS uint32, zeroed by the caller before the launch) into which the launch maxes the bit pattern of max |w| of that filter:
 * the filter's operand magnitude for the "f16x3" arithmetic (dsrl_amax, dsrl_conv2d_*_amax below). */
int dsrl_conv2d_transpose_filters_batched(const int64_t* table, int n, int64_t total_tiles, dsrl_stream_t stream);
/* The amax records of all filters in one streaming launch (what the transpose above leaves as a by-product, without the transposes: the per-step filter
 * pass of the "f16x3" arithmetic needs only the records and the split forms below).  table: nseg rows of 3 int64 {pointer into a filter's contiguous
 * [K][R][S][C] storage, number of floats of this segment (<= dsrl_conv2d_filters_amax_segment_floats()), amax record of that filter}; the caller
 * zeroes the records; a filter is cut into as many segments as it needs. */
int dsrl_conv2d_filters_amax_segment_floats(void);
int dsrl_conv2d_filters_amax_batched(const int64_t* table, int64_t nseg, dsrl_stream_t stream);

/* Filters pre-split for the "f16x3" arithmetic, all filters in one launch behind the transpose above (which leaves the amax records this
 * launch scales by): table rows {w, wt_split, K, Kp, R*S, C, first tile, ceil(C/32), amax record, w_split}; w_split [K][R][S][C] and
 * wt_split [C][R][S][Kp] (either may be 0) hold, per 4 consecutive elements of the last dimension, the 4 fp16 first terms of v * 2^e
 * followed by the 4 second terms: 16 bytes for 16 bytes of fp32, indexed like w / wt. dsrl_conv2d_fwd_amax / _dgrad_amax take them as
 * w_split / wt_split together with the SAME amax record and then stage the filter operand without splitting it in every row tile. */
int dsrl_conv2d_split_filters_batched(const int64_t* table, int n, int64_t total_tiles, dsrl_stream_t stream);
size_t dsrl_conv2d_dgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int dsrl_conv2d_dgrad(const float* dy, int lddy, const float* w, const float* wt /*nullable*/, float* dx, int lddx,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* dgrad of a conv whose input was y = relu(bn(x)) (or bn(x)): besides dx = the gradient w.r.t. y, the launch leaves the BatchNorm-backward
 * partial sums of g = dx * [y > 0] and g * xhat per (block of rows, channel) in bstats [2][parts][C], parts =
 * dsrl_conv2d_dgrad_stats_parts(shape) (0: this launch cannot - exact-fp32 kernels, split-K slabs, more than 4096 row blocks). The
 * BatchNorm backward then needs no reduction of its own: dsrl_bn_bwd_from_stats. bn_y may be null when bn_relu = 0. */
int dsrl_conv2d_dgrad_stats_parts(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int dsrl_conv2d_dgrad_bnstats(const float* dy, int lddy, const float* w, const float* wt /*nullable*/, float* dx, int lddx,
                              int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                              void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y /*nullable*/, int bn_ldy,
                              const float* bn_mean, const float* bn_invstd, int bn_relu, float* bstats, int stats_parts,
                              int accumulate /* dx += ..., the sums are taken of the accumulated values */, dsrl_stream_t stream);
/* dx += dgrad(dy, w): the same computation accumulated onto the existing contents of dx (a tensor that feeds two branches receives
 * both gradient contributions in one buffer, e.g. the input of a ResNet bottleneck: ResNet101.py residual add + conv1). */
int dsrl_conv2d_dgrad_accumulate(const float* dy, int lddy, const float* w, const float* wt /*nullable*/, float* dx, int lddx,
                                 int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                 void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* dw [K][R][S][C] from x and dy */
size_t dsrl_conv2d_wgrad_workspace_bytes(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);
int dsrl_conv2d_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw,
                      int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                      void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* Grouped weight gradients: every dsrl_conv2d_wgrad of a backward pass as a few grids (one per tile configuration, blocks ordered
 * longest first) plus one slab reduce, instead of one launch + reduce per layer. The weight gradients of a pass depend on nothing
 * but (x, dy) of their layer and are read by the optimiser only, so the caller may collect the problems while backward runs and
 * launch them once at its end (ddp.FlatParams does). Split-precision arithmetics only (dsrl_conv_precision 1..4).
 *   1. ws  = dsrl_conv2d_wgrad_group_workspace_bytes(problems, n); table = dsrl_conv2d_wgrad_group_table_bytes(n)
 *   2. dsrl_conv2d_wgrad_group_plan(problems, n, host_table, table, dev_table, ws_ptr, ws)  - fills host_table (host memory) with
 *      device-side descriptors that hold absolute device pointers (x, dy, dw, slabs inside ws_ptr) and offsets valid for dev_table
 *   3. the caller copies host_table -> dev_table (table bytes, stream-ordered before step 4; from pinned memory if asynchronous)
 *   4. dsrl_conv2d_wgrad_group_launch(host_table, dev_table, stream)
 * Same results as n calls of dsrl_conv2d_wgrad up to the summation order over pixel ranges. */
typedef struct dsrl_wgrad_problem {
    const float* x; const float* dy; float* dw;
    int32_t ldx, lddy, N, H, W, C, K, R, S, stride, pad, dil;
    const uint32_t* x_amax; const uint32_t* dy_amax;      /* "f16x3" arithmetic: operand magnitudes of x and dy (required there, else unused) */
} dsrl_wgrad_problem;
size_t dsrl_conv2d_wgrad_group_table_bytes(int n);
size_t dsrl_conv2d_wgrad_group_workspace_bytes(const dsrl_wgrad_problem* problems, int n);
int dsrl_conv2d_wgrad_group_plan(const dsrl_wgrad_problem* problems, int n, void* host_table, size_t table_bytes, const void* dev_table,
                                 void* ws, size_t ws_bytes);
int dsrl_conv2d_wgrad_group_launch(const void* host_table, const void* dev_table, dsrl_stream_t stream);
/* Operand magnitudes for the "f16x3" arithmetic (dsrl_conv_precision 4). That arithmetic carries every operand as two fp16 terms of
 * x * 2^e, with e chosen per TENSOR so that the tensor's largest magnitude lands in [2^14, 2^15); it therefore needs max |x| of both
 * operands of a launch, as an "amax record": DSRL_AMAX_WORDS uint32 device words (1 KiB, 64-byte aligned) of which every 16th holds the
 * bit pattern of a partial maximum of |x| (the writers' atomics are spread over 16 cache lines; the maximum of the record is max |x|).
 * Producers can leave the record while they write the tensor (the y_amax / dx_amax arguments of the BatchNorm kernels, the batched filter
 * transpose above); dsrl_amax measures any pixel-major tensor: it maxes into the record atomically, the caller zeroes the record first
 * (several calls may share a record).
 * The *_amax entry points below are dsrl_conv2d_fwd(_stats) / _dgrad(_bnstats, _accumulate) / _wgrad with the two records passed in; a
 * null record - and every call through the plain entry points - is measured by the call itself (one extra pass over that operand, in
 * the last 2 KiB of the workspace, which the *_workspace_bytes queries include). The other arithmetics ignore the records.
 * dsrl_conv2d_fwd_amax: stats may be null (no BatchNorm partials); dsrl_conv2d_dgrad_amax: bstats may be null (no BatchNorm sums; the
 * bn_* arguments are then unused), accumulate as in dsrl_conv2d_dgrad_accumulate. */
#define DSRL_AMAX_WORDS 256
int dsrl_amax(const float* x, int ld, int64_t P, int C, uint32_t* amax, dsrl_stream_t stream);
int dsrl_conv2d_fwd_amax(const float* x, int ldx, const uint32_t* x_amax, const float* w, const uint32_t* w_amax, const void* w_split /*nullable*/,
                         const float* bias /*nullable*/, float* y, int ldy, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                         void* ws, size_t ws_bytes, float* stats /*nullable*/, int stats_parts, dsrl_stream_t stream);
int dsrl_conv2d_dgrad_amax(const float* dy, int lddy, const uint32_t* dy_amax, const float* w, const float* wt /*nullable*/, const uint32_t* w_amax,
                           const void* wt_split /*nullable*/, float* dx, int lddx, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                           const float* bn_mean, const float* bn_invstd, int bn_relu, float* bstats /*nullable*/, int stats_parts, int accumulate,
                           dsrl_stream_t stream);
/* fp16 PLANES (round 4): an operand tensor of the "f16x3" arithmetic stored as its two terms, hi = f16(v * 2^e) and lo = f16(v * 2^e - hi) with e from
 * the tensor's amax record, each plane with the tensor's own indexing ([P][ld] fp16 for a pixel-major activation, [K][R][S][C] / [C][R][S][K] for a filter
 * and its transpose); the second plane starts dsrl_planes_lo_offset(elements of one plane) bytes behind the first. The conv kernels then stage operands by
 * LDS-DMA (buffer_load ... lds) without conversion. dsrl_planes_bytes: size of a plane buffer (nplanes 1 or 2).
 * dsrl_split_planes: planes of a pixel-major fp32 tensor (C and ld multiples of 8, 16-byte aligned), scaled by `amax` (a measured record, see dsrl_amax).
 * dsrl_conv2d_filter_planes_batched: all filters of a model in one launch; table rows of 10 int64 {w, wt_planes, K, unused, R*S, C, first tile,
 * ceil(C/32), amax record, w_planes} with tiles counted as R*S * ceil(C/32) * ceil(K/32) per filter; K and C multiples of 8; either pointer may be 0.
 * dsrl_conv2d_fwd_planes / _dgrad_planes: dsrl_conv2d_fwd_amax / _dgrad_amax with the plane forms passed as well (nullable); a launch that cannot
 * use them runs exactly as the _amax call; with the same tile plan the results are bit-identical to it. */
size_t dsrl_planes_lo_offset(int64_t elems);
size_t dsrl_planes_bytes(int64_t elems, int nplanes);
int dsrl_split_planes(const float* x, int ld, int64_t P, int C, const uint32_t* amax, void* planes, int nplanes, dsrl_stream_t stream);
int dsrl_conv2d_filter_planes_batched(const int64_t* table, int n, int64_t total_tiles, dsrl_stream_t stream);
int dsrl_conv2d_fwd_planes(const float* x, int ldx, const uint32_t* x_amax, const void* x_planes /*nullable*/, const float* w, const uint32_t* w_amax,
                           const void* w_split /*nullable*/, const void* w_planes /*nullable*/, const float* bias /*nullable*/, float* y, int ldy,
                           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* ws, size_t ws_bytes, float* stats /*nullable*/, int stats_parts, dsrl_stream_t stream);
int dsrl_conv2d_dgrad_planes(const float* dy, int lddy, const uint32_t* dy_amax, const void* dy_planes /*nullable*/, const float* w, const float* wt /*nullable*/,
                             const uint32_t* w_amax, const void* wt_split /*nullable*/, const void* wt_planes /*nullable*/, float* dx, int lddx,
                             int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                             void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                             const float* bn_mean, const float* bn_invstd, int bn_relu, float* bstats /*nullable*/, int stats_parts, int accumulate,
                             dsrl_stream_t stream);
/* the same with a Dropout(p) between the BatchNorm's ReLU and this conv (DSRL.py:38-41,46-49: cat_conv): bn_y is the tensor BEHIND the dropout, whose zeros are the
 * combined mask, and the sums are taken of g = dy / (1 - p) where bn_y > 0 (round 5) */
int dsrl_conv2d_dgrad_planes_drop(const float* dy, int lddy, const uint32_t* dy_amax, const void* dy_planes /*nullable*/, const float* w, const float* wt /*nullable*/,
                             const uint32_t* w_amax, const void* wt_split /*nullable*/, const void* wt_planes /*nullable*/, float* dx, int lddx,
                             int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                             void* ws, size_t ws_bytes, const float* bn_x, int bn_ldx, const float* bn_y, int bn_ldy,
                             const float* bn_mean, const float* bn_invstd, int bn_relu, float bn_drop_p, float* bstats /*nullable*/, int stats_parts, int accumulate,
                             dsrl_stream_t stream);
int dsrl_conv2d_wgrad_amax(const float* x, int ldx, const uint32_t* x_amax, const float* dy, int lddy, const uint32_t* dy_amax, float* dw,
                           int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                           void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* Arithmetic of the implicit-GEMM conv kernels (forward, dgrad, wgrad, row-folded stem), process-wide. fp32 in, fp32 out and fp32
 * accumulation in every mode; the modes differ in how the products are formed on the matrix cores:
 *   0  v_mfma_f32_32x32x2_f32 (exact fp32 products)
 *   1  "bf16x3": operand = 2 bf16 terms (16 mantissa bits), 3 bf16 MFMAs per product; ~5e-6 relative error per conv
 *   2  "bf16x6": operand = 3 bf16 terms (24 mantissa bits), 6 bf16 MFMAs per product; error vs fp64 equal to mode 0
 *   3  "mixed": forward bf16x6, dgrad / wgrad bf16x3 (reduced-precision gradients, ~5e-6)
 *   4  "f16x3": operand = 2 fp16 terms of the per-tensor scaled value (22 mantissa bits), 3 fp16 MFMAs per product; error vs fp64 equal
 *      to modes 0 and 2 at half the matrix work of mode 2 (operand magnitudes: see dsrl_amax above); the default
 *   5  "f16x1": operand = ONE fp16 term of the per-tensor scaled value (11 significand bits), one fp16 MFMA per product, fp32 accumulation:
 *      the arithmetic of the reference's apex O1 / O2 runs (train_or_resume.py:68-72) - reduced precision (~5e-4 of the output range per conv);
 *      the operand scales of mode 4 stand in for loss scaling
 *  -1  follow the environment variable DSRL_CONV_PRECISION (unset = 4)
 * Any other value changes nothing (query). Returns the previous setting. */
int dsrl_conv_precision(int mode);
/* in-bounds multiply-accumulates of one forward conv (zero-padding taps excluded): the roofline numerator */
int64_t dsrl_conv2d_inbounds_macs(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil);

/* Row-folded conv for few-channel inputs (the 7x7 stride-2 RGB stem, ResNet101.py:28): the caller pre-pads the image
 * physically (dsrl_pad_image_nhwc) and folds the S horizontal taps into the channel axis, so that one filter row is ONE
 * contiguous run of Cfold = S*ldx floats per output pixel: y[n,ho,wo,k] = sum_{r<R} sum_{c<Cfold} x[n, ho*stride + r, wo*stride, c] * w[k][r][c]
 * (c runs over neighbouring pixels). Requires (Ho-1)*stride + R <= H and (Wo-1)*stride*ldx + Cfold <= W*ldx.
 * algorithmic_macs is only bookkeeping for the launch timer (the true conv's in-bounds MACs). */
size_t dsrl_conv2d_rowfold_fwd_workspace_bytes(int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo);
int dsrl_conv2d_rowfold_fwd(const float* x, int ldx, const float* w, const float* bias /*nullable*/, float* y, int ldy,
                            int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo, int64_t algorithmic_macs,
                            void* ws, size_t ws_bytes, dsrl_stream_t stream);
size_t dsrl_conv2d_rowfold_wgrad_workspace_bytes(int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo);
/* dw [K][R][Cfold] */
int dsrl_conv2d_rowfold_wgrad(const float* x, int ldx, const float* dy, int lddy, float* dw,
                              int N, int H, int W, int Cfold, int K, int R, int stride, int Ho, int Wo, int64_t algorithmic_macs,
                              void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* y (N,Hp,Wp,Cp) pixel-major, zero except y[n, top+h, left+w, c] = x[n*sn + c*sc + h*sh + w*sw] for c < C */
int dsrl_pad_image_nhwc(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y,
                        int N, int C, int H, int W, int Cp, int top, int left, int Hp, int Wp, dsrl_stream_t stream);

/* column sums: out[c] = sum_p x[p*ld + c]  (conv / convT bias gradients, DSRL.py:50,64-69,78-83) */
size_t dsrl_colsum_workspace_bytes(int64_t P, int C);
int dsrl_colsum(const float* x, int ld, int64_t P, int C, float* out, void* ws, size_t ws_bytes, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (+ optional residual add, ReLU, Dropout) - nn.BatchNorm2d/ReLU/Dropout at ASPP.py:20-21,
 * DSRL.py:24-25,39-41,47-49,61-63,94-95 and the ResNet bottlenecks.
 * ---------------------------------------------------------------------------------------------- */
size_t dsrl_bn_workspace_bytes(int64_t P, int C);
/* training statistics: mean[c], invstd[c] = 1/sqrt(biased var + eps); running stats updated in place
 * (unbiased var, momentum) when non-null */
int dsrl_bn_stats(const float* x, int ldx, int64_t P, int C, float eps, float momentum,
                  float* mean, float* invstd, float* running_mean /*nullable*/, float* running_var /*nullable*/,
                  void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* invstd[c] = 1/sqrt(running_var[c] + eps) (eval mode / frozen BN, train_or_resume.py:379-382) */
int dsrl_bn_invstd_from_var(const float* running_var, int C, float eps, float* invstd, dsrl_stream_t stream);
/* y = dropout(relu((x-mean)*invstd*gamma + beta + residual)); flags: bit0 relu; dropout when p > 0.
 * y_amax / dx_amax (here and in the four functions below; nullable): an amax record (dsrl_amax), zeroed by the caller, into which the
 * launch maxes the bit pattern of max |.| of the tensor it writes - the operand magnitude of the "f16x3" conv arithmetic, taken while
 * the tensor is written instead of by a pass of its own. */
int dsrl_bn_apply(const float* x, int ldx, float* y, int ldy, int64_t P, int C,
                  const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* residual /*nullable*/, int ldr, int relu, float drop_p, uint64_t seed, uint32_t rng_stream,
                  uint32_t* y_amax /*nullable*/, dsrl_stream_t stream);
/* dsrl_bn_train_fwd with the batch statistics taken from the partials a preceding dsrl_conv2d_fwd_stats left behind (C a multiple of 32):
 * one streaming kernel, no statistics pass over x, no device-wide barrier. Same outputs as dsrl_bn_train_fwd. More than 256 row blocks of
 * partials (up to 4096: the 65536-pixel layers) are first reduced to 32 by a small launch, into the room behind the partials: a partials
 * buffer holds dsrl_bn_stats_floats(rows = 3 here / 2 for dsrl_bn_bwd_from_stats, parts, C) floats. */
size_t dsrl_bn_stats_floats(int rows, int parts, int C);
int dsrl_bn_train_fwd_from_stats(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float eps, float momentum, float* mean, float* invstd,
                                 float* running_mean /*nullable*/, float* running_var /*nullable*/, const float* gamma, const float* beta,
                                 const float* residual /*nullable*/, int ldr, int relu, float drop_p, uint64_t seed, uint32_t rng_stream,
                                 float* stats, int stats_parts, uint32_t* y_amax /*nullable*/, dsrl_stream_t stream);
/* The fused small-tensor BN kernels (dsrl_bn_train_fwd / dsrl_bn_bwd) cross a device-wide barrier: all blocks of a launch (128, or
 * 256 for tensors of 4.2-8.4 M elements; one 512-thread block per CU) must become resident together, so they assume that the process
 * has the GPU to itself apart from its own streams. dsrl_bn_fused_max_blocks: 0 = never use them, 128 = the 128-block variant only (what
 * ddp.FlatParams selects when RCCL kernels share the device), 256 = both, -1 = follow DSRL_BN_FUSED / DSRL_BN_FUSED_BIG; other values
 * change nothing; returns the previous setting. A block that waits for seconds gives up, poisons its outputs with NaN and counts in
 * dsrl_bn_fused_barrier_timeouts (synchronous read). */
int dsrl_bn_fused_max_blocks(int max_blocks);
int dsrl_bn_fused_barrier_timeouts(int64_t* count);
/* Training-mode BatchNorm2d forward in one call: batch statistics (as dsrl_bn_stats: mean / invstd out, running stats updated in place)
 * and y = act(gamma * xhat + beta (+ residual)) (as dsrl_bn_apply). Small tensors (C a power-of-two multiple of 32, P*C <= 4.2 M) take a
 * single fused kernel that reads x once; everything else runs the statistics and apply kernels. nn.BatchNorm2d (+ReLU, +Dropout) of
 * ASPP.py:20-21, DSRL.py:24-25,39-41,47-49 and of every backbone block (ResNet101.py). Workspace: dsrl_bn_workspace_bytes(P, C). */
int dsrl_bn_train_fwd(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float eps, float momentum, float* mean, float* invstd,
                      float* running_mean /*nullable*/, float* running_var /*nullable*/, const float* gamma, const float* beta,
                      const float* residual /*nullable*/, int ldr, int relu, float drop_p, uint64_t seed, uint32_t rng_stream,
                      void* ws, size_t ws_bytes, uint32_t* y_amax /*nullable*/, dsrl_stream_t stream);
/* backward of dsrl_bn_apply. y is the forward output (mask = y > 0 covers relu and dropout).
 * training != 0: batch-statistics gradient; == 0: statistics are constants. dresidual nullable. */
int dsrl_bn_bwd(const float* x, int ldx, const float* y, int ldy, const float* dy, int lddy,
                float* dx, int lddx, float* dresidual /*nullable*/, int lddr, int64_t P, int C,
                const float* mean, const float* invstd, const float* gamma,
                float* dgamma, float* dbeta, int relu, float drop_p, int training,
                void* ws, size_t ws_bytes, uint32_t* dx_amax /*nullable*/, dsrl_stream_t stream);
/* dsrl_bn_bwd (without dropout) with the two per-channel sums taken from the partials of dsrl_conv2d_dgrad_bnstats: one streaming kernel,
 * no reduction pass, no device-wide barrier (C a multiple of 32). */
int dsrl_bn_bwd_from_stats(const float* x, int ldx, const float* y /*nullable*/, int ldy, const float* dy, int lddy, float* dx, int lddx,
                           float* dresidual /*nullable*/, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                           float* dgamma /*nullable*/, float* dbeta /*nullable*/, int relu, int training, float* stats, int stats_parts,
                           uint32_t* dx_amax /*nullable*/, dsrl_stream_t stream);
/* ... with a Dropout(p) behind the ReLU: y is the tensor behind the dropout (mask y > 0, factor 1 / (1 - p)) */
int dsrl_bn_bwd_from_stats_drop(const float* x, int ldx, const float* y /*nullable*/, int ldy, const float* dy, int lddy, float* dx, int lddx,
                           float* dresidual /*nullable*/, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                           float* dgamma /*nullable*/, float* dbeta /*nullable*/, int relu, float drop_p, int training, float* stats, int stats_parts,
                           uint32_t* dx_amax /*nullable*/, dsrl_stream_t stream);
/* ... when the residual added before the ReLU is itself the output of a BatchNorm without ReLU - the downsample branch of a layer's first bottleneck,
 * out = relu(bn3(.) + bn_ds(conv_ds(x))), models/modules/backbone/ResNet101.py:67-89 (torchvision Bottleneck.forward) - its output gradient is the masked
 * gradient this launch writes to dresidual (required).  Given that BatchNorm's input res_x (pixel stride res_ldx) and batch statistics, the launch also
 * leaves ITS two backward sums per (row block, channel) in res_stats [2][res_parts][C] (room per dsrl_bn_stats_floats(2, res_parts, C)), res_parts =
 * dsrl_bn_bwd_from_stats_res_parts(P, C, stats_parts): the downsample BatchNorm's backward then is dsrl_bn_bwd_from_stats on dresidual - one streaming
 * launch instead of a reduction pass + apply pass (or the device-wide-barrier kernel) over the same tensors. */
int dsrl_bn_bwd_from_stats_res_parts(int64_t P, int C, int stats_parts);
int dsrl_bn_bwd_from_stats_res(const float* x, int ldx, const float* y /*nullable*/, int ldy, const float* dy, int lddy, float* dx, int lddx,
                           float* dresidual, int lddr, int64_t P, int C, const float* mean, const float* invstd, const float* gamma,
                           float* dgamma /*nullable*/, float* dbeta /*nullable*/, int relu, float drop_p, int training, float* stats, int stats_parts,
                           uint32_t* dx_amax /*nullable*/, const float* res_x, int res_ldx, const float* res_mean, const float* res_invstd,
                           float* res_stats, int res_parts, dsrl_stream_t stream);

/* Device-resident dropout key. By default every dropout-bearing launch (dsrl_bn_apply, dsrl_bn_train_fwd*, dsrl_dropout_*) bakes its
 * `seed` argument into the launch. After dsrl_rng_bind_device_key(ptr) the kernels of the CURRENT device ignore that argument and read
 * the 64-bit key at `ptr` when they run, so launches captured in a hipGraph draw fresh masks on every replay; NULL restores the
 * default. dsrl_rng_advance_key(state) enqueues the per-step key derivation on three device words {key, step, base}:
 * step += 1; key = base * 1000003 + step (mod 2^64) - the derivation functional.begin_forward() performs on the host. */
int dsrl_rng_bind_device_key(const uint64_t* dev_key /*nullable*/);
int dsrl_rng_advance_key(uint64_t* state /*device: key, step, base*/, dsrl_stream_t stream);

/* standalone Dropout (DSRL.py:54): Philox4x32-10 keyed by (seed, rng_stream), element index = p*C + c */
int dsrl_dropout_fwd(const float* x, int ldx, float* y, int ldy, int64_t P, int C, float p, uint64_t seed, uint32_t rng_stream, dsrl_stream_t stream);
int dsrl_dropout_bwd(const float* dy, int lddy, float* dx, int lddx, int64_t P, int C, float p, uint64_t seed, uint32_t rng_stream, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * bilinear resize, align_corners=True (ASPP.py:41, DSRL.py:53, DSRL.py:163) and pools
 * ---------------------------------------------------------------------------------------------- */
int dsrl_bilinear_ac_fwd(const float* x, int ldx, float* y, int ldy, int N, int H, int W, int C, int Ho, int Wo, dsrl_stream_t stream);
size_t dsrl_bilinear_ac_bwd_workspace_bytes(int N, int H, int W, int C, int Ho, int Wo);
int dsrl_bilinear_ac_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int H, int W, int C, int Ho, int Wo,
                         void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* nn.AdaptiveAvgPool2d((1,1)) (ASPP.py:22,38) */
int dsrl_global_avgpool_fwd(const float* x, int ldx, float* y, int N, int HW, int C, dsrl_stream_t stream);
int dsrl_global_avgpool_bwd(const float* dy, float* dx, int lddx, int N, int HW, int C, dsrl_stream_t stream);
/* nn.MaxPool2d(3, stride 2, pad 1) (ResNet101.py:32) */
/* argmax (nullable in fwd): per output element the winning tap r*3+s (first maximum in scan order), consumed by the backward */
int dsrl_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* argmax, int N, int H, int W, int C, dsrl_stream_t stream);
int dsrl_maxpool3x3s2_bwd(const uint8_t* argmax, const float* dy, float* dx, int N, int H, int W, int C, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ConvTranspose2d(kernel 2, stride 2, pad 0) (DSRL.py:55-60, 64-69); w is (Cin,Cout,2,2) contiguous
 * ---------------------------------------------------------------------------------------------- */
int dsrl_convt2x2_fwd(const float* x, const float* w, const float* bias /*nullable*/, float* y, int N, int H, int W, int Cin, int Cout, dsrl_stream_t stream);
size_t dsrl_convt2x2_bwd_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int dsrl_convt2x2_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias /*nullable*/,
                      int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, dsrl_stream_t stream);

/* dsrl_convt2x2_fwd that also evaluates nn.CrossEntropyLoss(ignore_index), mean reduction, of its output against `target` (N,2H,2W bytes) while the
 * output tile is in LDS (DSRL.py:69 -> train_or_resume.py:435): loss_out[0] = the loss, loss_out[1] = the number of pixels that count, as
 * dsrl_ce_fused writes them; nan_flag (nullable) bit 0 = NaN logit, bit 1 = label outside [0, Cout).  19 -> 19 channels, W % 4 == 0. */
int dsrl_convt2x2_fwd_ce_supported(const float* x, const float* y, int N, int H, int W, int Cin, int Cout);
size_t dsrl_convt2x2_fwd_ce_workspace_bytes(int N, int H, int W);
int dsrl_convt2x2_fwd_ce(const float* x, const float* w, const float* bias /*nullable*/, float* y, int N, int H, int W, int Cin, int Cout,
                         const uint8_t* target, int ignore_index, float* loss_out, int* nan_flag /*nullable*/, void* ws, size_t ws_bytes, dsrl_stream_t stream);

/* The same backward when y IS the logits of nn.CrossEntropyLoss(ignore_index) with mean reduction (DSRL.py:69 feeds train_or_resume.py:435) and the
 * loss is the root of the backward pass with unit gradient: dy = d(CE)/d(logits) is formed inside the kernel from `logits` (the forward output y),
 * `target` (N,2H,2W bytes) and `ce_count` (the number of pixels that are not ignore_index, dsrl_ce_fused's loss_out[1]), in the arithmetic of
 * dsrl_ce_fused, and never written to memory.  ft_g (nullable): the incoming gradient (N, ceil(2H/s), ceil(2W/s)) of the stride-s single-channel 1x1
 * conv that also reads the logits (feature transformer, DSRL.py:88-93) and ft_w its Cout weights: g * w_c is added on the stride grid, as
 * dsrl_pointwise_strided_bwd(accumulate = 1) would.  Results are bit-identical to dsrl_ce_fused -> dsrl_pointwise_strided_bwd -> dsrl_convt2x2_bwd.
 * _supported: 1 when the shape can take this path (19 -> 19 channels, W % 128 == 0, 16-byte aligned tensors); otherwise use the three calls. */
int dsrl_convt2x2_bwd_ce_supported(const float* x, const float* logits, const uint8_t* target, int N, int H, int W, int Cin, int Cout);
int dsrl_convt2x2_bwd_ce(const float* x, const float* w, const float* logits, const uint8_t* target, int ignore_index, const float* ce_count,
                         const float* ft_g /*nullable*/, const float* ft_w /*nullable*/, int ft_stride, float* dx, float* dw, float* dbias /*nullable*/,
                         int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, dsrl_stream_t stream);

/* nn.PixelShuffle(r) (DSRL.py:84): x (N,H,W,c*r*r) -> y (N,H*r,W*r,c) */
int dsrl_pixel_shuffle_fwd(const float* x, float* y, int N, int H, int W, int c, int r, dsrl_stream_t stream);
int dsrl_pixel_shuffle_bwd(const float* dy, float* dx, int N, int H, int W, int c, int r, dsrl_stream_t stream);

/* 1x1 stride-s conv with a single output channel, no bias (feature transformers, DSRL.py:88-93) */
int dsrl_pointwise_strided_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int stride, dsrl_stream_t stream);
size_t dsrl_pointwise_strided_bwd_workspace_bytes(int N, int H, int W, int C, int stride);
/* dx is fully written (zeros off the stride grid) when accumulate == 0, dx += on the stride grid only when 1; 2: dw only (dx may be null) */
int dsrl_pointwise_strided_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, int accumulate,
                               int N, int H, int W, int C, int stride, void* ws, size_t ws_bytes, dsrl_stream_t stream);

/* channel concatenation (ASPP.py:44, DSRL.py:165): dst[p*ld_dst + c] = src[p*ld_src + c], c < C (strided 2-D copy) */
int dsrl_copy2d(const float* src, int ld_src, float* dst, int ld_dst, int64_t P, int C, dsrl_stream_t stream);
/* the whole concatenation as one launch (round 5): sources srcs[i] (HOST array of n <= 8 device pointers; pixel strides lds[i], widths cs[i], all multiples of
 * 4, 16-byte aligned) written side by side into dst (pixel stride ld_dst), and max |value| of everything written left in the amax record `amax`
 * (nullable) - the consumers of a concatenated buffer (ASPP.py:44 -> the 1280 -> 256 projection; DSRL.py:165 -> cat_conv.0 and the SISR conv) need that
 * magnitude for their operand scale and used to measure it with a pass of its own.  dsrl_cat_channels_supported: 1 when the arguments qualify (the
 * kernel indexes float4 with 32 bits: P * stride / 4 < 2^31 for every tensor). */
int dsrl_cat_channels_supported(const float* const* srcs, const int* lds, const int* cs, int n, const float* dst, int ld_dst, int64_t P);
int dsrl_cat_channels(const float* const* srcs, const int* lds, const int* cs, int n, float* dst, int ld_dst, int64_t P, uint32_t* amax, dsrl_stream_t stream);

/* layout converters at the boundary (NCHW image in, ResNet101.py:92): y has Cpad >= C channels, extra = 0 */
int dsrl_nchw_to_nhwc(const float* x, float* y, int N, int C, int H, int W, int Cpad, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * losses (train_or_resume.py:116-119, 435-438)
 * ---------------------------------------------------------------------------------------------- */
size_t dsrl_ce_workspace_bytes(int64_t P);
/* nn.CrossEntropyLoss(ignore_index): logits [P][C] pixel-major, target uint8 [P]; loss_out[0] = mean NLL over
 * valid pixels, loss_out[1] = number of valid pixels */
int dsrl_ce_fwd(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index,
                float* loss_out, void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* dlogits = (softmax - onehot) * grad_out[0] / n_valid ; loss_out is the pair written by dsrl_ce_fwd */
int dsrl_ce_bwd(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index,
                const float* loss_out, const float* grad_out, float* dlogits, int lddl, dsrl_stream_t stream);
size_t dsrl_mse_workspace_bytes(int64_t n);
int dsrl_mse_fwd(const float* a, const float* b, int64_t n, float* loss_out, void* ws, size_t ws_bytes, dsrl_stream_t stream);
int dsrl_mse_bwd(const float* a, const float* b, int64_t n, const float* grad_out, float* da, dsrl_stream_t stream);
/* Fused loss pass (train_or_resume.py:116-117, 426-438): ONE pass over the logits computes CrossEntropyLoss(ignore_index,
 * mean) into loss_out[0] (loss_out[1] = pixels counted) AND writes d(loss)/d(logits) for an upstream gradient of 1 into dlogits
 * (nullable: forward only), and ORs 1 into nan_flag (nullable) if a logit is NaN. dsrl_mse_fused does the same for MSELoss:
 * da = grad_scale * d(mse)/da. dsrl_loss_mix writes {CE, w1*MSE, w2*FA, sum, (float)*nan_flag} to vals[5] (mse / fa nullable: stages 1, 2). */
size_t dsrl_ce_fused_workspace_bytes(int64_t P);
int dsrl_ce_fused(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index, float* dlogits /*nullable*/, int lddl,
                  float* loss_out /*[2]*/, int* nan_flag /*nullable*/, void* ws, size_t ws_bytes, dsrl_stream_t stream);
int dsrl_mse_fused(const float* a, const float* b, int64_t n, float grad_scale, float* da /*nullable*/, float* loss_out, int* nan_flag /*nullable*/,
                   void* ws, size_t ws_bytes, dsrl_stream_t stream);
int dsrl_loss_mix(const float* ce, const float* mse /*nullable*/, const float* fa /*nullable*/, float w1, float w2, const int* nan_flag /*nullable*/,
                  float* vals /*[5]*/, dsrl_stream_t stream);

/* FALoss (models/losses/FALoss.py:8-34). fm1/fm2 are (B,C,H,W) with element strides (sb,sc,sh,sw).
 * reduction: 0 mean, 1 sum, 2 none (out has B*C*n*n floats, n = (W/k)^2).
 * `saved` (>= dsrl_fa_saved_floats) carries S1,S2,sigma,u1,v1 to the backward. */
size_t dsrl_fa_saved_floats(int B, int C, int H, int W, int k);
size_t dsrl_fa_workspace_bytes(int B, int C, int H, int W, int k);
int dsrl_fa_fwd(const float* fm1, const float* fm2, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                int k, int reduction, float* out, float* saved, void* ws, size_t ws_bytes, dsrl_stream_t stream);
/* mean/sum: grad_out is 1 float; d1/d2 are (B,C,H,W) contiguous */
int dsrl_fa_bwd(const float* fm1, const float* fm2, int B, int C, int H, int W, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                int k, int reduction, const float* grad_out, const float* saved, float* d1, float* d2,
                void* ws, size_t ws_bytes, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * callers / data formats either side of the path
 * ---------------------------------------------------------------------------------------------- */
/* Validation metrics (metrices/mIoU.py:15-41, metrices/Accuracy.py:13-30) straight from the logits: pred = argmax_c logits,
 * counts[0..C) += area_pred, [C..2C) += area_inter, [2C..3C) += area_target, counts[3C] += correct, counts[3C+1] += valid
 * (pixels whose target is the ignore label are excluded everywhere). 64-bit integer atomics: exact and order independent. */
int dsrl_seg_metrics(const float* logits, int ld, const uint8_t* target, int64_t P, int C, int ignore_index,
                     unsigned long long* counts, dsrl_stream_t stream);
/* Deterministic tail of the training input pipeline (JointImageAndLabelTensor.py:9-16 label remap, JointNormalize.py:11,
 * JointScaledImage.py:27-32): from a decoded RGB uint8 crop (N,Hs,Ws,3) and its label-id map (N,Hs,Ws):
 *   img_in  (N,H,W,4)   = align-corners bilinear resize of ((u8/255 - mean)/std) to the model input size, 4th channel 0
 *                         (pixel-major, padded to the 4 channels the stem kernel wants), img_org (N,2H,2W,3) the same at the output size,
 *   target  (N,2H,2W)   = nearest resize of lut[label]                                                                        */
int dsrl_prepare_batch(const uint8_t* rgb, const uint8_t* labels, const uint8_t* lut /*256*/, const float* mean /*3*/, const float* std /*3*/,
                       float* img_in, float* img_org, uint8_t* target, int N, int Hs, int Ws, int H, int W, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * optimiser + bookkeeping on the flat parameter arena (train_or_resume.py:63-66, 445, 426-433)
 * ---------------------------------------------------------------------------------------------- */
/* torch.optim.SGD(momentum, weight_decay): d = g*grad_scale + wd*p; buf = mom*buf + d; p -= lr*buf */
int dsrl_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float weight_decay,
                  float grad_scale, dsrl_stream_t stream);
/* the same update with {lr, momentum, weight_decay, grad_scale} read from four floats in device memory when the kernel runs: a launch
 * captured in a hipGraph follows the per-epoch LR schedule (train_or_resume.py:109-113, 349) without being re-captured */
int dsrl_sgd_step_dev(float* p, const float* g, float* buf, int64_t n, const float* hyper /*device, 4 floats*/, dsrl_stream_t stream);
/* the same update driven by a device table of nseg rows {first float, floats, address of an amax record or 0} (int64 each) that covers the arena with segments of
 * whole float4 (first float a multiple of 4, count a multiple of 4, each segment inside ONE parameter; one block per segment, <= 32768 floats recommended): for
 * rows with a record the launch also leaves max |p| of the UPDATED values there (same bit patterns as dsrl_conv2d_filters_amax_batched; the records must be zero
 * when the launch starts) - the conv filters' operand magnitudes for the next step come out of the optimiser pass instead of a sweep of their own */
int dsrl_sgd_step_dev_segments(float* p, const float* g, float* buf, const int64_t* table, int64_t nseg, const float* hyper /*device, 4 floats*/, dsrl_stream_t stream);
/* flag[0] |= 1 if any element is NaN (the reference's per-output NaN asserts folded into one readback) */
int dsrl_nan_check(const float* x, int64_t n, int* flag, dsrl_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * in-library launch timing (bench.py roofline): dsrl_prof_enable(n) brackets every n-th MFMA conv launch (n = 1: all of them,
 * 0: off) with HIP events on its own stream - an event pair costs ~3 us of device time, so a timed region samples. dsrl_prof_read() synchronises those events and reports per kernel family the launch count,
 * summed milliseconds and summed in-bounds FLOPs. family = 3 * arithmetic + pass, arithmetic 0 fp32 / 1 bf16x3 / 2 bf16x6
 * (see dsrl_conv_precision), pass 0 forward / 1 wgrad / 2 dgrad; dsrl_prof_kernel_name() names the kernel of a family.
 * ---------------------------------------------------------------------------------------------- */
int dsrl_prof_enable(int on);
int dsrl_prof_read(int family, int64_t* launches, double* total_ms, double* total_flops);
/* summed ALGORITHMIC bytes of the recorded launches of a family: every operand (input, filter, output) of the conv once, fp32 */
int dsrl_prof_read_bytes(int family, double* total_bytes);
const char* dsrl_prof_kernel_name(int family);

#ifdef __cplusplus
}
#endif
#endif /* DSRL_HIP_H */
