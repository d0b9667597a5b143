/*
 * kgdet_hip.h -- C ABI of libkgdet_hip.so, the MI355X (gfx950) implementation of KGDet's hot
 * operators.  Plain pointers and sizes only; every pointer named "device" is HBM memory on the
 * current HIP device, `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *
 * Each entry point replaces one native function that the reference's Python wrappers bind
 * through pybind11 (R = mmdetection/mmdet/ops in the reference tree); the reference interface
 * it stands in for is cited next to it.  Conventions kept from the reference: the caller
 * allocates outputs and gradient buffers, the library writes in place; float32, dense NCHW.
 * Differences (not observable by callers): kernels run on the given stream instead of the
 * legacy default stream, scratch comes from a caller-provided workspace instead of per-call
 * at::zeros, and failures are reported (status code + kgdet_last_error()) instead of printf.
 *
 * Return value: 0 on success, non-zero on error (see KGDET_E_*).
 */
#ifndef KGDET_HIP_H_
#define KGDET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KGDET_OK 0
#define KGDET_E_SHAPE 1     /* shape/argument check failed (reference: AT_CHECK -> RuntimeError) */
#define KGDET_E_WORKSPACE 2 /* workspace too small */
#define KGDET_E_HIP 3       /* 1x1 convolution, stride 1, float32 NCHW, as bf16 hi/lo-split MFMA GEMMs (fp32-level accuracy) -- the conv1 / conv3 /
 * stride-1 downsample convolutions of the bottleneck, mmdet/models/backbones/resnet.py:142-186, which the reference
 * runs through cuDNN / MIOpen.  No reference native entry point.
 *   taps = 1: 1x1;  taps = 9: 3x3 with stride 1, padding 1, dilation 1 (conv2 of the bottleneck) as an implicit GEMM.
 *   kgdet_conv_pack: weight [O, C, taps] -> operand image; transpose = 0 for the forward (y = W * x), 1 for
 *                       grad_input (gx = W^T * gy, taps mirrored); the reduction length (C resp. O) a multiple of 16.
 *   kgdet_conv_apply: y[b] (M x HoWo) = A (M x K*taps) . patches(x[b]) with A = the packed image; x [B, K, H, W],
 *                       y [B, M, ceil(H / stride), ceil(W / stride)]; stride 1 or 2 (forward only: the transposed
 *                       image gives grad_input for stride 1).
 *   kgdet_conv1x1_grad_weight: grad_w [O, C] = sum_b grad_y[b] (O x HW) . x[b]^T (HW x C); HW even; deterministic
 *                       (per-chunk partial tiles added in fixed order); workspace from the _workspace_bytes query. */
size_t kgdet_conv_packed_bytes(int32_t M, int32_t K, int32_t taps);
int kgdet_conv_pack(const float *w, int32_t O, int32_t C, int32_t taps, int32_t transpose, void *packed, void *stream);
/* operand_format of the *_fmt entry points: 0 = two bf16 parts per fp32 value (16 mantissa bits; every gradient operand),
 * 1 = two fp16 parts (22 bits, fp32-class results at the same MFMA rate) for FORWARD operands.  Envelope of format 1: weights up
 * to 255 in magnitude (the image stores them scaled by 2^8 so that their lo parts stay normal; the kernels scale the
 * accumulators back), activations of magnitude ~1e-2 .. 6e4 at full accuracy (<= 1e-6 of the output scale); beyond 65504 they
 * saturate, below ~1e-3 the lo part becomes an fp16 subnormal and the error has an ABSOLUTE floor of ~3e-8 per activation
 * (1e-4 of the output scale at 1e-3) -- the outputs of BatchNorm / GroupNorm-normalised layers sit inside the envelope;
 * format 0 has no floor and 16 bits.  An image packed with format f must be applied with format f; only
 * forward images (transpose = 0) take format 1.  kgdet_conv_pack_multi: bit 62 of a descriptor's last word selects
 * format 1 for that row's forward image.  The plain entry points are format 0. */
int kgdet_conv_pack_fmt(const float *w, int32_t O, int32_t C, int32_t taps, int32_t transpose, void *packed,
                        int32_t operand_format, void *stream);
int kgdet_conv_pack_both_fmt(const float *w, int32_t O, int32_t C, int32_t taps, void *packed, void *packed_t,
                             int32_t forward_format, void *stream);
int kgdet_conv_apply_epilogue_fmt(const void *packed, const float *x, float *y, const float *bias, const float *residual,
                                  int32_t relu, int64_t B, int32_t M, int32_t K, int32_t H, int32_t W, int32_t taps,
                                  int32_t stride, int32_t operand_format, void *workspace, size_t workspace_bytes,
                                  void *stream);
/* ... with gate [B, M, Ho, Wo] (nullable): y = [gate > 0] * ([relu](conv + bias [+ residual])).  The backward of the ReLU in
 * front of a convolution (mmdet/models/backbones/resnet.py:240-262: relu(norm(conv(.))) feeding the next conv) applied in the
 * store of that convolution's grad_input kernel, gate = its forward input: the producer of the input then receives its
 * gradient already masked (kgdet_amd/backbone.py _GateLink). */
int kgdet_conv_apply_gated_fmt(const void *packed, const float *x, float *y, const float *bias, const float *residual,
                               int32_t relu, const float *gate, int64_t B, int32_t M, int32_t K, int32_t H, int32_t W,
                               int32_t taps, int32_t stride, int32_t operand_format, void *workspace, size_t workspace_bytes,
                               void *stream);
int kgdet_stem_conv7x7_s2_fmt(const void *packed, const float *x, float *y, int64_t B, int32_t H, int32_t W,
                              int32_t operand_format, void *stream);
/* forward and grad_input images of one weight in one launch (O and C multiples of 16) */
int kgdet_conv_pack_both(const float *w, int32_t O, int32_t C, int32_t taps, void *packed, void *packed_t, void *stream);
/* Both images of n weights in ONE launch (training re-packs every weight every step).  desc_dev: device table of n x 6
 * int64 {weight ptr, image ptr, transposed-image ptr, (O << 32) | C, (taps << 32) | first block, scale ptr or 0}, first
 * block = running sum of kgdet_conv_pack_blocks(O, C, taps) over the preceding entries; total_blocks = the sum over all
 * entries.  scale: float [O]; the images are those of w[o] * scale[o] (a frozen-statistics BatchNorm folded into the weight). */
int64_t kgdet_conv_pack_blocks(int32_t O, int32_t C, int32_t taps);
int kgdet_conv_pack_multi(const int64_t *desc_dev, int32_t n, int64_t total_blocks, void *stream);
size_t kgdet_conv_apply_workspace_bytes(int64_t B, int32_t M, int32_t K, int32_t H, int32_t W, int32_t taps,
                                        int32_t stride); /* mostly 0 */
int kgdet_conv_apply(const void *packed, const float *x, float *y, int64_t B, int32_t M, int32_t K, int32_t H, int32_t W,
                     int32_t taps, int32_t stride, void *workspace, size_t workspace_bytes, void *stream);
/* the same with the inference epilogue fused into the store: y = [relu](conv + bias[m] [+ residual]); bias [M] and
 * residual [B, M, Ho, Wo] nullable (the folded-BatchNorm bottleneck at inference, kgdet_amd/backbone.py conv_bn). */
int kgdet_conv_apply_epilogue(const void *packed, const float *x, float *y, const float *bias, const float *residual,
                              int32_t relu, int64_t B, int32_t M, int32_t K, int32_t H, int32_t W, int32_t taps,
                              int32_t stride, void *workspace, size_t workspace_bytes, void *stream);
/* The stem convolution conv1 = Conv2d(3, 64, 7, stride 2, padding 3, bias none) (mmdet/models/backbones/resnet.py:487-488):
 * x [B, 3, H, W] -> y [B, 64, (H-1)/2+1, (W-1)/2+1] with the split-bf16 arithmetic of the other convolutions.  packed: the
 * image kgdet_conv_pack(w160, 64, 160, 1, 0, ...) of the weight flattened to [64, 147] and zero-padded to [64, 160].
 * Forward only (the stem is frozen: frozen_stages >= 0). */
int kgdet_stem_conv7x7_s2(const void *packed, const float *x, float *y, int64_t B, int32_t H, int32_t W, void *stream);
/* grad_x [B, C, Hin, Win] of the 3x3 stride-2 padding-1 convolution (the bottleneck's conv2 at the head of layers 2-4,
 * mmdet/models/backbones/resnet.py:142-186 with stride 2) from grad_y [B, O, ceil(Hin/2), ceil(Win/2)] and the TRANSPOSED image
 * of its weight (kgdet_conv_pack(w, O, C, 9, 1, ...)); O % 16 == 0.  Replaces ATen's convolution_backward (MIOpen). */
int kgdet_conv3x3_s2_grad_input(const void *packed_t, const float *grad_y, float *grad_x, int64_t B, int32_t C, int32_t O,
                                int32_t Hin, int32_t Win, void *stream);
/* grad_w [O, C, 3, 3] of the same stride-2 convolution: x [B, C, H, W], grad_y [B, O, ceil(H/2), ceil(W/2)].  The nine strided views of x are
 * gathered into the workspace once, the product over the pixels runs on the 1x1 weight-gradient kernels (deterministic).  O * C even.
 * Replaces ATen's convolution_backward (MIOpen igemm_wrw + layout transposes), mmdet/models/backbones/resnet.py:142-186. */
size_t kgdet_conv3x3_s2_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int32_t H, int32_t W);
int kgdet_conv3x3_s2_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C, int32_t H,
                                 int32_t W, void *workspace, size_t workspace_bytes, void *stream);
size_t kgdet_conv1x1_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int64_t HW);
int kgdet_conv1x1_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                              int64_t HW, void *workspace, size_t workspace_bytes, void *stream);
/* 3x3 (stride 1, padding 1): grad_w [O, C, 3, 3]; needs C % 128 == 0 (else KGDET_E_UNSUPPORTED).  W % 4 != 0: both operands are
 * copied into the workspace with zero columns up to a multiple of 4 (one launch) first. */
size_t kgdet_conv3x3_grad_weight_workspace_bytes(int64_t B, int32_t O, int32_t C, int32_t H, int32_t W);
int kgdet_conv3x3_grad_weight(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                              int32_t H, int32_t W, void *workspace, size_t workspace_bytes, void *stream);

/* Frozen-statistics BatchNorm (+ residual add) (+ ReLU) in one pass -- the norm_eval=True training path of
 * mmdet/models/backbones/resnet.py:240-262,518-525.  float32, NCHW: x, residual, y [N, C, HW]; gamma, beta (nullable:
 * 1 / 0), mean, var [C].     y = [relu](x * s + t [+ residual]),  s = gamma / sqrt(var + eps),  t = beta - mean * s.
 * y may alias x when x is not needed by a later backward. */
int kgdet_bn_act_forward(const float *x, const float *gamma, const float *beta, const float *mean, const float *var,
                         float eps, const float *residual, float *y, int64_t N, int32_t C, int64_t HW, int32_t relu,
                         void *stream);
/* Backward of the above.  g' = grad_y * [y > 0] (relu) else grad_y;  grad_x = g' * s (grad_x nullable);
 * grad_residual = g' is WRITTEN only when has_residual && relu (otherwise it equals grad_y and the caller reuses
 * that tensor); y is read only in that case.  partial: [2][C][P] with P = kgdet_bn_act_partials(N, C, HW):
 * partial[0][c][:] sums to grad_beta[c], partial[1][c][:] to grad_gamma[c] (per-workgroup partials in a fixed order:
 * deterministic, no atomics).  sums (nullable): [2][C] = the partials added in slot order by a second tiny launch --
 * sums[0] = grad_beta, sums[1] = grad_gamma; NULL: the caller adds them. */
int32_t kgdet_bn_act_partials(int64_t N, int32_t C, int64_t HW);
int kgdet_bn_act_backward(const float *grad_y, const float *x, const float *y, const float *gamma, const float *beta,
                          const float *mean, const float *var, float eps, int32_t has_residual, int32_t relu,
                          float *grad_x, float *grad_residual, float *partial, float *sums, int64_t N, int32_t C,
                          int64_t HW, void *stream);
/* Frozen-statistics BatchNorm folded into the convolution in front of it (mmdet resnet.py:518-525 norm_eval; replaces the
 * separate BatchNorm pass of resnet.py:240-262): the forward is kgdet_conv_apply_epilogue* with the image of w * s and bias t.
 * Backward: g = grad_z * [z > 0] (relu; without relu only the channel sums are formed and g is not written) + per-workgroup
 * partial channel sums [C][P], P = kgdet_bn_act_partials(N, C, HW); then per convolution kgdet_bn_fold_finish:
 * grad_beta = sum of partials, grad_gamma = (<w[o], G[o]> - mean * grad_beta) / sqrt(var + eps) with G the weight gradient
 * of conv(x, .) against g, and G scaled by s in place (= grad_w).  Deterministic. */
/* Inference, bf16 channels-last: conv3 + folded bn3 + identity add + ReLU of a bottleneck (mmdet/models/backbones/resnet.py:
 * 240-262) as ONE kernel: out[m][n] = [relu](bf16(sum_k x[m][k] weight[n][k]) + bias[n] + residual[m][n]); x [M, K], weight
 * [N, K], residual / out [M, N] bf16 (M = B*H*W), bias fp32 [N]; K % 16 == 0, K <= 512, N % 128 == 0 or N == 64.
 * residual == NULL: conv1 + folded bn1 + ReLU (no identity add) on the same kernel. */
int kgdet_conv1x1_nhwc_residual(const void *x, const void *weight, const float *bias, const void *residual, void *out,
                                int64_t M, int32_t K, int32_t N, int32_t relu, void *stream);
/* the same with x = the RAW output of the convolution in front (conv2 without its epilogue): relu(x + in_bias[k]) (rounded to
 * bf16, as the separate pass stores it) is applied to the activations as they are loaded; in_bias fp32 [K] */
int kgdet_conv1x1_nhwc_residual_in(const void *x, const float *in_bias, const void *weight, const float *bias,
                                   const void *residual, void *out, int64_t M, int32_t K, int32_t N, int32_t relu,
                                   void *stream);
int kgdet_bn_fold_backward(const float *grad_z, const float *z, int32_t relu, float *g, float *partial, int64_t N, int32_t C,
                           int64_t HW, void *stream);
int kgdet_bn_fold_finish(const float *partial, int32_t P, const float *w, float *G /*nullable*/, const float *s,
                         const float *mean, const float *var, float eps, float *grad_beta /*nullable*/,
                         float *grad_gamma /*nullable*/, int32_t O, int32_t CK, void *stream);
/* kgdet_conv1x1_grad_weight / kgdet_conv3x3_grad_weight with kgdet_bn_fold_finish as the epilogue of their split sum (one
 * launch less per convolution): grad_w = s * G, grad_beta, grad_gamma as above; bn_partial / P from kgdet_bn_fold_backward --
 * or bn_partial == NULL with P == 0: grad_y is final (no ReLU mask to apply, or applied already by kgdet_conv_apply_gated_fmt) and
 * its per-channel sums are formed inside the weight-gradient kernel, from the operand values it loads anyway (no pass of
 * kgdet_bn_fold_backward over grad_y at all). */
int kgdet_conv1x1_grad_weight_fold(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                                   int64_t HW, void *workspace, size_t workspace_bytes, const float *w, const float *s,
                                   const float *mean, const float *var, float eps, const float *bn_partial, int32_t P,
                                   float *grad_beta /*nullable*/, float *grad_gamma /*nullable*/, void *stream);
int kgdet_conv3x3_grad_weight_fold(const float *grad_y, const float *x, float *grad_w, int64_t B, int32_t O, int32_t C,
                                   int32_t H, int32_t W, void *workspace, size_t workspace_bytes, const float *w,
                                   const float *s, const float *mean, const float *var, float eps, const float *bn_partial,
                                   int32_t P, float *grad_beta /*nullable*/, float *grad_gamma /*nullable*/, void *stream);
/* The frozen stem: y = maxpool3x3/s2/p1(relu(batch_norm_eval(x))) in one pass (mmdet/models/backbones/resnet.py:487-491, 528);
 * x [N, C, H, W] -> y [N, C, (H-1)/2+1, (W-1)/2+1].  Forward only (conv1 / norm1 are frozen: frozen_stages >= 0). */
int kgdet_bn_relu_maxpool(const float *x, const float *gamma, const float *beta, const float *mean, const float *var,
                          float eps, float *y, int64_t N, int32_t C, int32_t H, int32_t W, void *stream);

/* Inference epilogue of a convolution with folded BatchNorm: x = [relu](x + bias[c] [+ residual]) in place.
 * x, residual: [N, C, HW] contiguous, or [N, HW, C] when channels_last != 0 (then C must be a multiple of 4
 * (float32) / 8 (bfloat16), else KGDET_E_UNSUPPORTED); dtype 0 = float32, 1 = bfloat16; bias [C] float32
 * (nullable); residual nullable.  No reference counterpart: it fuses the BatchNorm / residual add / ReLU passes of
 * mmdet/models/backbones/resnet.py:231-262 once the frozen statistics are folded into the weights. */
int kgdet_bias_act(void *x, const float *bias, const void *residual, int64_t N, int32_t C, int64_t HW, int32_t dtype,
                   int32_t relu, int32_t channels_last, void *stream);
/* The stem at inference with the frozen BatchNorm folded into conv1: y = maxpool3x3/s2/p1(relu(x + bias[c])) in one pass on
 * channels-last tensors (mmdet/models/backbones/resnet.py:487-491, 528); x [N, H, W, C] -> y [N, (H-1)/2+1, (W-1)/2+1, C];
 * dtype 0 = float32 (C % 4 == 0), 1 = bfloat16 (C % 8 == 0), else KGDET_E_UNSUPPORTED; bias [C] float32, nullable.  Same
 * values as kgdet_bias_act + max pooling (each element is rounded to the storage type before the maximum). */
int kgdet_bias_relu_maxpool_nhwc(const void *x, const float *bias, void *y, int64_t N, int32_t C, int32_t H, int32_t W,
                                 int32_t dtype, void *stream);

/* HIP runtime error (launch failure etc.) */
#define KGDET_E_UNSUPPORTED 4
#define KGDET_E_PARTIAL 5     /* a measurement switch (KGDET_OPT_BWD_PHASE) was set: only part of the call's outputs were written */

const char *kgdet_last_error(void); /* thread-local message for the last non-zero status */
int kgdet_version(void);            /* ABI version, currently 1 */
int kgdet_device_cu_count(void);    /* compute units of the current device (0 if none) */

/* Process-wide switches (diagnostics / accuracy studies; defaults 0).
 * KGDET_OPT_EXACT_BACKWARD != 0: kgdet_deform_conv_backward_input takes the exact-fp32 kernels (f32-input MFMA)
 * instead of the split-bf16 plane kernels it prefers for v1 problems -- the arithmetic of the reference's fp32
 * col2im path (deform_conv_cuda.cpp:260-371), used to measure what the hi/lo split costs over a training step. */
#define KGDET_OPT_EXACT_BACKWARD 0
/* (slot 1: KGDET_OPT_TAP_PAIRS in rounds 2-4, the column-wave forward in round 5 -- both kernels left the library, tools/experiments/;
 * setting it has no effect) */
#define KGDET_OPT_FWD_COLUMN_WAVE 1
/* KGDET_OPT_WGRAD_STREAMK != 0: kgdet_deform_conv_grad_weight_grouped keeps rounds 1-3's schedule -- 256 x 128 tiles, (tile,
 * stage) units dealt stream-K, partial tiles + fix-up -- where it would otherwise run the output-stationary kernel (round 4:
 * one 256 x 208 tile per workgroup for the whole reduction, csrc/dcn_backward_weight_os.hip).  Same results to round-off
 * (another summation order); for A/B measurements and so that both kernels stay under test. */
#define KGDET_OPT_WGRAD_STREAMK 2
/* KGDET_OPT_BWD_PHASE (measurement switch, bench.py `roofline.backward`): 1 = kgdet_deform_conv_backward_input_grouped launches
 * only its grad_input phase, 2 = only its grad_offset phase (the other product's outputs are left untouched), 0 = both. */
#define KGDET_OPT_BWD_PHASE 3
#define KGDET_OPT_COUNT 4
int kgdet_set_option(int32_t option, int32_t value);

/* ------------------------------------------------------------------------------------------
 * Deformable convolution v1 / v2
 * shape of one call; weight is [O, C/groups, kh, kw], offset [N, dg*2*kh*kw, Ho, Wo] with
 * (dy,dx) interleaved per tap, taps row-major; mask (v2) [N, dg*kh*kw, Ho, Wo].
 * ------------------------------------------------------------------------------------------ */
typedef struct kgdet_dcn_shape {
  int32_t N, C, H, W;
  int32_t O, kh, kw;
  int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w;
  int32_t groups, deformable_groups;
  /* Optional: the output (forward) / grad_output (backward) tensor is channels
   * [out_channel_offset, out_channel_offset + O) of a wider [N, out_channels_total, Ho, Wo] buffer, so
   * several convolutions can write straight into one concatenated feature map (KGDet concatenates
   * the 3x3 / 5x5 / 7x7 branches, R/../anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py:151-153).
   * Zero for both means a dense [N, O, Ho, Wo] tensor. */
  int32_t out_channel_offset, out_channels_total;
} kgdet_dcn_shape;

/* flags for the fused epilogue */
#define KGDET_DCN_RELU 1u /* out = max(out, 0) */
/* Forward arithmetic.  Default: bf16 MFMA on a hi/lo split of both fp32 operands (3 products, fp32
 * accumulate; <= 2^-16 relative error per product, i.e. fp32-accurate to ~1e-6 of the output scale).
 * KGDET_DCN_BF16: operands rounded to bf16 once (autocast inference).  KGDET_DCN_EXACT_FP32: the
 * v_mfma_f32_32x32x2_f32 kernel, bit-exact fp32 products (also used when a 16-channel slice of one input
 * image does not fit in LDS: H*W > 1344). */
#define KGDET_DCN_BF16 2u
#define KGDET_DCN_EXACT_FP32 4u

/* output spatial size, R/dcn/deform_conv.py:96-110; returns KGDET_E_SHAPE if it is < 1 */
int kgdet_dcn_output_size(const kgdet_dcn_shape *s, int32_t *Ho, int32_t *Wo);

/* bytes of the MFMA-friendly weight image [groups][kh*kw][C/groups (pad 16)][O/groups (pad 256)] */
size_t kgdet_dcn_packed_weight_bytes(const kgdet_dcn_shape *s);
/* bytes of scratch (split-K partial tiles) forward / backward need besides the packed weight */
size_t kgdet_dcn_workspace_bytes(const kgdet_dcn_shape *s);

/* weight [O, C/groups, kh, kw] (device) -> packed (device).  Cache it while the weight is unchanged. */
int kgdet_dcn_pack_weight(const kgdet_dcn_shape *s, const float *weight, float *packed, void *stream);
/* n weights at once (training re-packs every DeformConv weight every step: the six of a KGDet head stage in one launch);
 * same images as n kgdet_dcn_pack_weight calls. */
int kgdet_dcn_pack_weight_multi(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *weights,
                                float *const *packeds, void *stream);
/* The same with a choice of images (training re-packs every weight every step: 170 MB written per KGDet head stage for all four
 * images): bit 0 of `images` = the two fp32 images (exact-fp32 forward, fallback backward kernels), bit 1 = the two split (bf16
 * hi/lo) images of the default kernels.  kgdet_dcn_split_path_complete(shape) = 1 when every product of a convolution of that
 * shape has a split-operand kernel, i.e. nothing reads the fp32 images unless an exact-arithmetic switch is set.  The packed
 * buffer's layout and size do not change; images that are not packed are left as they are. */
int kgdet_dcn_pack_weight_images(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *weights,
                                 float *const *packeds, uint32_t images, void *stream);
int32_t kgdet_dcn_split_path_complete(const kgdet_dcn_shape *shape);
/* packed gradient image -> grad_weight [O, C/groups, kh, kw]; accumulate != 0 adds into grad_weight */
int kgdet_dcn_unpack_weight_grad(const kgdet_dcn_shape *s, const float *packed, float *grad_weight,
                                 int accumulate, void *stream);

/*
 * Forward.  Replaces deform_conv_forward_cuda (R/dcn/src/deform_conv_cuda.cpp:151-156) when
 * mask == NULL && bias == NULL, and modulated_deform_conv_cuda_forward (:486-492) otherwise.
 * `packed_weight` comes from kgdet_dcn_pack_weight.  output [N, O, Ho, Wo] is overwritten.
 * No column matrix is materialised: samples are gathered into LDS and contracted with MFMA.
 */
int kgdet_deform_conv_forward(const kgdet_dcn_shape *s, const float *input, const float *offset,
                              const float *mask /*nullable*/, const float *packed_weight,
                              const float *bias /*nullable*/, float *output, uint32_t flags,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Scratch for kgdet_deform_conv_forward_grouped: split-K slabs plus one table of sampling records per problem
 * (also >= kgdet_dcn_workspace_bytes of every member, so the same buffer serves their backward calls). */
size_t kgdet_dcn_group_workspace_bytes(int32_t n, const kgdet_dcn_shape *const *shapes);

/* n independent forward problems in ONE launch: same results as n calls of kgdet_deform_conv_forward
 * with the same flags.  The KGDet head runs a 3x3, a 5x5 and a 7x7 deformable conv on each of two feature
 * maps per stage (reppoints_head_kp3rep_cas_1_assign_once.py:145-163); grouping them shares the split-K
 * partial tiles, the fix-up pass and the launch overhead.  All arrays have n entries; `masks` / `biases`
 * (or single entries) may be NULL.  workspace >= kgdet_dcn_group_workspace_bytes(n, shapes). */
int kgdet_deform_conv_forward_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                      const float *const *offsets, const float *const *masks /*nullable*/,
                                      const float *const *packed_weights, const float *const *biases /*nullable*/,
                                      float *const *outputs, uint32_t flags, void *workspace,
                                      size_t workspace_bytes, void *stream);

/* grad_input only, on the plane kernel (transposed sampling; atomic-free, deterministic, overwrites grad_input).
 * Same quantity as the grad_input of kgdet_deform_conv_backward_input; weight groups and deformable groups run as
 * channel runs (channels that share both; at most 8 runs, each starting on a 256-channel boundary of its weight group or
 * ending before the next one); needs H*W, Ho*Wo <= 1344 (KGDET_E_UNSUPPORTED otherwise).  flags: KGDET_DCN_BF16 or 0
 * (hi/lo split).
 * workspace >= kgdet_dcn_workspace_bytes(s). */
int kgdet_deform_conv_grad_input(const kgdet_dcn_shape *s, const float *offset, const float *mask /*nullable*/,
                                 const float *packed_weight, const float *grad_output, float *grad_input,
                                 uint32_t flags, void *workspace, size_t workspace_bytes, void *stream);

/* grad_offset only (v1), with the column gradient W^T grad_out kept in registers and the feature plane in LDS.
 * Same quantity as the grad_offset of kgdet_deform_conv_backward_input; needs every deformable group inside ONE weight
 * group (groups == 1 with any number of deformable groups of whole 16-channel chunks, or deformable groups that subdivide
 * the weight groups), O / groups <= 256, H*W <= 1344 (KGDET_E_UNSUPPORTED otherwise).  Overwrites grad_offset.
 * workspace >= kgdet_dcn_workspace_bytes(s). */
int kgdet_deform_conv_grad_offset(const kgdet_dcn_shape *s, const float *input, const float *offset,
                                  const float *packed_weight, const float *grad_output, float *grad_offset,
                                  uint32_t flags, void *workspace, size_t workspace_bytes, void *stream);

/* grad_input and grad_offset of n v1 problems in two grouped launches (one per quantity); same results as
 * kgdet_deform_conv_grad_input + kgdet_deform_conv_grad_offset per problem.  n <= 8; every problem must be
 * eligible for both kernels and have groups == 1, deformable_groups == 1 (KGDET_E_UNSUPPORTED otherwise -- call the
 * single entry points).
 * ALIASED OUTPUTS ARE SUMMED (round 6): problems that pass the SAME grad_input pointer (the 3x3 / 5x5 / 7x7 convolutions of one
 * feature map) or the same grad_offset pointer (the two maps of a Kp3RepBlock, which share their offsets) get the SUM of their
 * results in that tensor -- added by the fix-up kernels in fixed order (problems ascending, reduction parts ascending), what
 * autograd's accumulation of the reference's per-call gradients (deform_conv.py:62-93) produces.  Up to four problems per pointer;
 * they must tile alike (same N, pixels, channel rows; grad_offset: same kernel size) and the launch must fit the static schedule,
 * else KGDET_E_UNSUPPORTED before anything is launched for that quantity (pass one tensor per problem and add them yourself).
 * workspace >= kgdet_dcn_group_workspace_bytes(n, shapes). */
int kgdet_deform_conv_backward_input_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                             const float *const *offsets, const float *const *packed_weights,
                                             const float *const *grad_outputs, float *const *grad_inputs,
                                             float *const *grad_offsets, void *workspace, size_t workspace_bytes,
                                             void *stream);

/* grad_weight of n v1 problems (H*W <= 1344; weight groups and deformable groups as channel runs, at most 8 runs over
 * all problems) in one launch: a GEMM with the
 * reduction over pixels, sampled B stages from an LDS-resident plane, bf16 hi/lo split MFMA; grad_weights[i]
 * [O, C, kh, kw] is OVERWRITTEN.  KGDET_E_UNSUPPORTED for other shapes (use kgdet_deform_conv_backward_weight).
 * workspace >= kgdet_dcn_group_workspace_bytes(n, shapes). */
int kgdet_deform_conv_grad_weight_grouped(int32_t n, const kgdet_dcn_shape *const *shapes, const float *const *inputs,
                                          const float *const *offsets, const float *const *grad_outputs,
                                          float *const *grad_weights, void *workspace, size_t workspace_bytes,
                                          void *stream);

/*
 * Backward w.r.t. input and offset (and mask for v2).  Replaces
 * deform_conv_backward_input_cuda (deform_conv_cuda.cpp:260-266) and the input/offset/mask part
 * of modulated_deform_conv_cuda_backward (:566-573).  grad_offset / grad_mask are overwritten.
 * grad_input: pass it zero-filled, as R/dcn/deform_conv.py:73 does.  Maps whose 32-channel plane set
 * fits LDS (H*W <= 1088) take the atomic-free gather path, which overwrites it deterministically;
 * larger maps accumulate into it with float atomics like the reference.
 */
int kgdet_deform_conv_backward_input(const kgdet_dcn_shape *s, const float *input, const float *offset,
                                     const float *mask /*nullable*/, const float *packed_weight,
                                     const float *grad_output, float *grad_input, float *grad_offset,
                                     float *grad_mask /*nullable*/, void *workspace,
                                     size_t workspace_bytes, void *stream);

/*
 * Backward w.r.t. weight (and bias for v2).  Replaces deform_conv_backward_parameters_cuda
 * (deform_conv_cuda.cpp:373-378, scale = 1) and the weight/bias part of
 * modulated_deform_conv_cuda_backward.  grad_weight [O, C/groups, kh, kw] is overwritten
 * (accumulate == 0) or added to (accumulate != 0); grad_bias [O] likewise when not NULL.
 * accumulate == 0: bf16 hi/lo split MFMA kernels with natural-layout output -- an LDS-resident plane for maps up to 1344
 * pixels, corners gathered from a pixel-major copy of the input (in the workspace) beyond; any weight groups /
 * deformable groups of 16-channel multiples.  accumulate != 0 or KGDET_OPT_EXACT_BACKWARD: the fp32 MFMA kernel.
 */
int kgdet_deform_conv_backward_weight(const kgdet_dcn_shape *s, const float *input, const float *offset,
                                      const float *mask /*nullable*/, const float *grad_output,
                                      float *grad_weight, float *grad_bias /*nullable*/, int accumulate,
                                      void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Deformable PS-RoI pooling.  Replaces deform_psroi_pooling_cuda_forward / _backward
 * (R/dcn/src/deform_pool_cuda.cpp:30-34, 54-59; kernels deform_pool_cuda_kernel.cu:53-140, 143-263).
 * rois [R,5] = (batch_idx,x1,y1,x2,y2); trans [R, 2*num_classes, part, part] (NULL when no_trans);
 * out/count [R, out_dim, P, P].
 * Backward OVERWRITES grad_input [B,C,H,W] and grad_trans (every element written exactly once: no pre-zeroing, no
 * atomics, bit-repeatable; the reference accumulates with atomicAdd into zero-filled tensors).  It needs
 * kgdet_deform_psroi_backward_workspace_bytes(s) bytes of caller-owned scratch (per-RoI bounding boxes + the
 * transposed grad_out / count quotient + the per-pixel contribution lists grad_data is summed from).
 * Envelope (LDS record table): pooled_size^2 * sample_per_part^2 <= 2048, pooled_size <= 16, group_size <= 16 --
 * KGDET_E_SHAPE beyond (the reference's configs use 7 / 4 / 1 or 7).
 * ------------------------------------------------------------------------------------------ */
typedef struct kgdet_psroi_shape {
  int32_t B, C, H, W;  /* data */
  int32_t R;           /* number of rois */
  int32_t out_dim, group_size, pooled_size, part_size, sample_per_part;
  int32_t no_trans, num_classes; /* num_classes = no_trans ? 1 : trans.size(1)/2 */
  float spatial_scale, trans_std;
} kgdet_psroi_shape;

size_t kgdet_deform_psroi_forward_workspace_bytes(const kgdet_psroi_shape *s);
int kgdet_deform_psroi_forward(const kgdet_psroi_shape *s, const float *data, const float *rois,
                               const float *trans, float *out, float *count, void *workspace, size_t workspace_bytes,
                               void *stream);
size_t kgdet_deform_psroi_backward_workspace_bytes(const kgdet_psroi_shape *s);
int kgdet_deform_psroi_backward(const kgdet_psroi_shape *s, const float *grad_out, const float *count,
                                const float *data, const float *rois, const float *trans,
                                float *grad_data, float *grad_trans, void *workspace, size_t workspace_bytes,
                                void *stream);

/* ------------------------------------------------------------------------------------------
 * Sigmoid focal loss.  Replaces sigmoid_focal_loss_cuda.forward / .backward
 * (R/sigmoid_focal_loss/src/sigmoid_focal_loss.cpp:40-45).  logits [num, C]; targets [num]
 * int64 (0 = background, 1..C = class); losses / d_logits [num, C].
 * ------------------------------------------------------------------------------------------ */
int kgdet_sigmoid_focal_loss_forward(const float *logits, const int64_t *targets, int64_t num,
                                     int32_t num_classes, float gamma, float alpha, float *losses,
                                     void *stream);
int kgdet_sigmoid_focal_loss_backward(const float *logits, const int64_t *targets, const float *d_losses,
                                      int64_t num, int32_t num_classes, float gamma, float alpha,
                                      float *d_logits, void *stream);

/* ------------------------------------------------------------------------------------------
 * Target assignment + the nine losses of the KGDet head (one pyramid level) without materialised targets.
 * Replaces, for RepPointsHeadKp3RepCas1AssignOnce.loss (reppoints_head_kp3rep_cas_1_assign_once.py:581-665):
 * PointAssigner.assign (core/bbox/assigners/point_assigner.py:23-121), point_target_kp (core/anchor/
 * point_target_kp.py:7-169), offset_to_pts (:537-579), FocalLoss / SmoothL1Loss with weight and avg_factor
 * (models/losses/focal_loss.py:28-82, smooth_l1_loss.py:8-45, utils.py:7-52; sigmoid_focal_loss_cuda.cu:24-97).
 * Prediction maps are the head's NCHW outputs: cls [B, num_classes, H, W], bbox [B, 4, H, W] (x1, y1, x2, y2
 * offsets), kpt [B, 2 * num_keypoints, H, W] ((y, x) offset pairs), three stages each.  Points are the level's grid
 * (x = column * stride, y = row * stride).  losses[9] = cls 1-3, bbox 1-3, kpt 1-3; num_total = sum over images of
 * max(positives, 1) (a device scalar; nothing is read by the host).  The backward call takes the forward call's
 * workspace (it holds the per-gt selections) and writes every element of the nine gradient maps.
 * Limits: B <= 16, 1..64 ground truths per image, H * W <= 4096.
 * ------------------------------------------------------------------------------------------ */
#define KGDET_HEAD_MAX_IMAGES 16
typedef struct kgdet_head_targets {
  int32_t B, H, W, num_classes, num_keypoints;
  float stride;
  int32_t num_gt[KGDET_HEAD_MAX_IMAGES];
  const float *gt_bboxes[KGDET_HEAD_MAX_IMAGES];     /* [num_gt, 4] */
  const int64_t *gt_labels[KGDET_HEAD_MAX_IMAGES];   /* [num_gt], NULL: every label 1 */
  const float *gt_keypoints[KGDET_HEAD_MAX_IMAGES];  /* [num_gt, num_keypoints, 3] (x, y, visibility) */
  /* valid extent of image b on the grid: rows < valid_h[b] and columns < valid_w[b] (ceil(pad_shape / stride), the flags of
   * reppoints_head_kp3rep_cas_1_assign_once.py:524-535); 0 = the whole grid.  Points beyond it are no candidates of the
   * assigner and carry label weight 0 (point_target_kp.py:107-161 with unmap): no loss, zero gradient. */
  int32_t valid_h[KGDET_HEAD_MAX_IMAGES], valid_w[KGDET_HEAD_MAX_IMAGES];
} kgdet_head_targets;
typedef struct kgdet_head_loss_cfg {
  int32_t pos_num;               /* PointAssigner.pos_num */
  float pos_weight;              /* label weight of positives (train_cfg.pos_weight <= 0 -> 1) */
  float normalize_term;          /* point_base_scale * stride */
  float gamma[3], alpha[3];      /* FocalLoss of stage 1-3 */
  float beta[6];                 /* SmoothL1Loss: bbox 1-3, kpt 1-3 */
  float loss_weight[9];          /* cls 1-3, bbox 1-3, kpt 1-3 */
} kgdet_head_loss_cfg;
typedef struct kgdet_head_maps {
  float *cls[3], *bbox[3], *kpt[3];
} kgdet_head_maps;
size_t kgdet_head_loss_workspace_bytes(const kgdet_head_targets *t);
int kgdet_head_loss_forward(const kgdet_head_targets *t, const kgdet_head_loss_cfg *cfg, const kgdet_head_maps *maps,
                            float *losses, float *num_total, void *workspace, size_t workspace_bytes, void *stream);
int kgdet_head_loss_backward(const kgdet_head_targets *t, const kgdet_head_loss_cfg *cfg, const kgdet_head_maps *maps,
                             const float *grad_losses, const float *num_total, const kgdet_head_maps *grads,
                             const void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Element-wise glue of the KGDet step as single passes (no native function in the reference).
 * kgdet_reppts_offsets_*: the deformable offsets of a Kp3RepBlock from the previous stage's reppoints,
 *   reppoints_head_kp3rep_cas_1_assign_once.py:131-143: for the 2 k^2-channel slices (k = kernel_sizes[0..2], consecutive)
 *   of reppts [B, C, HW]: out_k = (gm * part + (1 - gm) * part) - regular_grid_k; backward: grad_reppts = gm * grad_k on the
 *   slices (NULL grad_k: zeros), 0 on the channels beyond them.
 * kgdet_subsample2_*: y = x[:, :, ::2, ::2] of [planes, H, W] and grad_x = zero-stuffed grad_y (+ other, nullable) -- the
 *   stride-2 1x1 downsample branch of resnet.py:180-186 as a 1x1 convolution of the subsampled input; backward needs W % 4 == 0.
 * kgdet_pts_from_offsets_*: a level's predicted offsets pred [B, C = 2n, HW] as image coordinates pts [B, HW, 2n] =
 *   offset * stride + centre with (x, y) interleaved -- offset_to_pts of reppoints_head_kp_serial.py:400-421 (permute, flip when
 *   the channels are (y, x) pairs: y_first, multiply, add) in one pass; centres [B, HW, 2] = (x, y).  Values equal the torch
 *   chain's bit for bit (separate multiply and add).  Backward: grad_pred = transposed (pair-swapped) grad_pts * stride.
 * ------------------------------------------------------------------------------------------ */
int kgdet_reppts_offsets_forward(const float *reppts, int32_t B, int32_t C, int32_t HW, const int32_t *kernel_sizes, float gm,
                                 float *out0, float *out1, float *out2, void *stream);
int kgdet_reppts_offsets_backward(const float *g0, const float *g1, const float *g2, int32_t B, int32_t C, int32_t HW,
                                  const int32_t *kernel_sizes, float gm, float *grad_reppts, void *stream);
int kgdet_subsample2_forward(const float *x, float *y, int64_t planes, int32_t H, int32_t W, void *stream);
int kgdet_subsample2_backward(const float *grad_y, const float *other /*nullable*/, float *grad_x, int64_t planes, int32_t H,
                              int32_t W, void *stream);
int kgdet_pts_from_offsets_forward(const float *pred, const float *centres, float *pts, int64_t B, int32_t C, int64_t HW,
                                   float stride, int32_t y_first, void *stream);
int kgdet_pts_from_offsets_backward(const float *grad_pts, float *grad_pred, int64_t B, int32_t C, int64_t HW, float stride,
                                    int32_t y_first, void *stream);

/*
 * Gradient clipping + Adam over all parameters as multi-tensor passes: what OptimizerHook.after_train_iter does with
 * clip_grad_norm_(params, max_norm, 2) followed by torch.optim.Adam.step() (mmdet/core/utils/dist_utils.py:44-58; torch/optim/
 * adam.py).  table_dev: n rows of six int64 on the device -- {param, grad, exp_avg, exp_avg_sq (float32 pointers), numel,
 * first block}, a tensor taking ceil(numel / kgdet_optim_chunk()) blocks, rows in block order; total_blocks = their sum.
 * kgdet_multi_grad_norm: partial [total_blocks] scratch, norm_out[0] = ||all grads||_2.  kgdet_multi_clip_adam: gradients
 * are scaled by min(1, max_norm / (norm[0] + 1e-6)) in place (max_norm <= 0: no clipping), then the Adam update with
 * bias_correction1 = 1 - beta1^step and bias_correction2_sqrt = sqrt(1 - beta2^step) of the step being taken; weight decay
 * is added to the gradient (Adam, not AdamW); the betas come as doubles so that 1 - beta is rounded once, as torch does.  No amsgrad / maximize.  Deterministic; no host synchronisation.
 */
int32_t kgdet_optim_chunk(void);
int kgdet_multi_grad_norm(const int64_t *table_dev, int32_t n, int64_t total_blocks, float *partial, float *norm_out, void *stream);
int kgdet_multi_clip_adam(const int64_t *table_dev, int32_t n, int64_t total_blocks, const float *norm, float max_norm, float lr,
                          double beta1, double beta2, float eps, float weight_decay, float bias_correction1,
                          float bias_correction2_sqrt, void *stream);
/*
 * The same clip + Adam update with the step's schedule in DEVICE memory, for a training step captured as one HIP graph (kernel
 * arguments are frozen at capture; the reference's optimizer.step() -- dist_utils.py:57 -- takes them from host state every call):
 * sched[4] = {learning rate, 1 - beta1^t, sqrt(1 - beta2^t), t}.  The call first advances the schedule (t <- t + 1, corrections
 * recomputed in double precision; learning rate <- lr_ring[t % ring] when lr_ring != NULL: `ring` floats of page-locked host memory
 * the device reads in place, slot k written by the host before it launches step k), then applies the update with it.
 */
int kgdet_multi_clip_adam_dev(const int64_t *table_dev, int32_t n, int64_t total_blocks, const float *norm, float max_norm,
                              float *sched, const float *lr_ring /*nullable*/, int32_t ring, double beta1, double beta2, float eps,
                              float weight_decay, void *stream);

/*
 * GroupNorm (+ ReLU) of the ConvModules of the head towers and the neck as one pass each way -- ATen runs it as ten kernels
 * per layer (mmdet/models/utils/conv_module.py:142-165: norm, then activate; mmdet/models/utils/norm.py GN = nn.GroupNorm).
 * x, y, grad_* [N, C, HW] float32 contiguous; gamma, beta [C] nullable; mean, rstd [N * groups] (saved for the backward);
 * C / groups <= 64.  Backward: grad_x nullable; dgamma_dbeta [2][N][C] = per-image rows of dgamma, then of dbeta (the caller
 * adds the N rows); relu != 0 masks grad_y where y <= 0.  Deterministic.
 */
int kgdet_gn_act_forward(const float *x, const float *gamma, const float *beta, int32_t groups, float eps, int32_t relu,
                         float *y, float *mean, float *rstd, int64_t N, int32_t C, int64_t HW, void *stream);
int kgdet_gn_act_backward(const float *grad_y, const float *x, const float *y, const float *gamma, const float *mean,
                          const float *rstd, int32_t groups, int32_t relu, float *grad_x, float *dgamma_dbeta, int64_t N,
                          int32_t C, int64_t HW, void *stream);
/*
 * The same for groups of any size: beyond 65536 elements per (image, group) -- the 100 x 168 level of the five-level heads:
 * 8 channels x 16800 pixels, 64 (image, group) pairs -- the pixels of a group are cut into kgdet_gn_act_slices(...) slices with
 * a workgroup each (slice moments combined with the parallel-variance formula; per-channel gradient sums added in slice
 * order: deterministic), through `scratch` of kgdet_gn_act_scratch_floats(...) floats (NULL when that is 0; the forward's and the
 * backward's scratch need not be the same buffer).  Small groups run the kernels above.
 */
int32_t kgdet_gn_act_slices(int64_t N, int32_t C, int32_t groups, int64_t HW);
size_t kgdet_gn_act_scratch_floats(int64_t N, int32_t C, int32_t groups, int64_t HW);
int kgdet_gn_act_forward_split(const float *x, const float *gamma, const float *beta, int32_t groups, float eps, int32_t relu,
                               float *y, float *mean, float *rstd, float *scratch, int64_t N, int32_t C, int64_t HW,
                               void *stream);
int kgdet_gn_act_backward_split(const float *grad_y, const float *x, const float *y, const float *gamma, const float *mean,
                                const float *rstd, int32_t groups, int32_t relu, float *grad_x, float *dgamma_dbeta,
                                float *scratch, int64_t N, int32_t C, int64_t HW, void *stream);
/* Inference under autocast: x, y bfloat16, fp32 arithmetic, no saved statistics; at most 65536 elements per group.
 * x and y [N, C, HW], or both channels-last [N, HW, C] when x_channels_last != 0. */
int kgdet_gn_act_forward_bf16(const void *x, int32_t x_channels_last, const float *gamma, const float *beta, int32_t groups,
                              float eps, int32_t relu, void *y, int64_t N, int32_t C, int64_t HW, void *stream);
/* the same for groups of any size (the 100 x 168 level of a five-level head: 8 channels x 16800 pixels per group): beyond 65536
 * elements the split kernels, `scratch` of kgdet_gn_act_scratch_floats(...) floats (NULL when that is 0) */
int kgdet_gn_act_forward_bf16_split(const void *x, int32_t x_channels_last, const float *gamma, const float *beta, int32_t groups,
                                    float eps, int32_t relu, void *y, float *scratch, int64_t N, int32_t C, int64_t HW,
                                    void *stream);

/*
 * Weighted smooth-L1 sum of the head's box / keypoint losses in one pass each way: what the reference computes as a chain of
 * element-wise torch ops (mmdet/models/losses/smooth_l1_loss.py:8-45, utils.py:7-52 called from KP3:621-665 with
 * pred / d and target / d):   sum_out[0] = sum_i weight[i] * l(|pred[i] / d - target[i] / d|),
 * l(x) = x < beta ? 0.5 x^2 / beta : x - 0.5 beta;   grad_pred[i] = grad_sum[0] * weight[i] * l'(.) / d.
 * n elements, weight nullable (all ones); partial: kgdet_smooth_l1_partials() floats of scratch; grad_sum: DEVICE scalar (no
 * host round trip).  The caller applies loss_weight and 1 / avg_factor to the sum.  Deterministic.
 */
int32_t kgdet_smooth_l1_partials(void);
int kgdet_smooth_l1_sum_forward(const float *pred, const float *target, const float *weight, int64_t n, float beta,
                                float divisor, float *partial, float *sum_out, void *stream);
int kgdet_smooth_l1_sum_backward(const float *pred, const float *target, const float *weight, const float *grad_sum, int64_t n,
                                 float beta, float divisor, float *grad_pred, void *stream);

/* ------------------------------------------------------------------------------------------
 * NMS.  kgdet_nms replaces nms_cpu.nms / nms_cuda.nms (R/nms/src/nms_cpu.cpp:62-68,
 * nms_kernel.cu:70-131) with the CPU semantics the project pins: +1 areas, suppress when
 * IoU >= thr, visiting order = score descending (ties: lower index first), result = kept
 * indices in ascending index order.  The greedy sweep runs on the device; nothing is copied
 * to the host.  dets [n,5] float32; keep [n] int64; *num_keep (device int64) receives the count.
 *
 * kgdet_nms_batched runs many independent problems (one per (image, class) group) in ONE
 * launch: segment i covers dets rows [seg_offsets[i], seg_offsets[i+1]); kept indices are
 * written segment-relative to keep + seg_offsets[i], counts to num_keep[i].
 * workspace: kgdet_nms_workspace_bytes(total_n, num_segments).
 *
 * kgdet_soft_nms replaces soft_nms_cpu (R/nms/src/soft_nms_cpu.pyx:22-127): method 1 linear,
 * 2 gaussian; out_dets [n,5], out_inds [n] int64, *num_out device int64.
 * ------------------------------------------------------------------------------------------ */
/* Segments of up to 4096 boxes are processed entirely in LDS; longer ones (nms_wrapper.py:8-49 takes any N) run the same
 * algorithm with their arrays in `workspace` (size from kgdet_nms_workspace_bytes; 16 bytes suffice when no segment
 * exceeds 4096 boxes).  Kept indices are bit-identical to nms_cpu.cpp either way. */
size_t kgdet_nms_workspace_bytes(int64_t total_n, int32_t num_segments);
/* multiclass_nms_kp for a whole batch (mmdet/core/post_processing/bbox_nms_kp.py:6-75 with type='nms'), two launches,
 * nothing read by the host: per (image, class) the candidates with score > score_thr are suppressed as kgdet_nms does;
 * per image the classes' survivors are concatenated (class order, ascending candidate row) and, beyond max_num, the
 * max_num highest scores are kept (ties: earlier first).  boxes [B, N, 4]; scores [B, N, score_stride] with class c in
 * column score_col0 + c (C <= 64, N <= 4096, N*C <= 16384); out_det [B, max_num, 5]; out_label (0-based class) and
 * out_src (candidate row, for gathering the landmarks) [B, max_num] int64; out_count [B] int64; rows past the count
 * are zero.  workspace: kgdet_multiclass_nms_workspace_bytes(B, N, C). */
size_t kgdet_multiclass_nms_workspace_bytes(int32_t B, int32_t N, int32_t C);
int kgdet_multiclass_nms(const float *boxes, const float *scores, int32_t B, int32_t N, int32_t C,
                         int32_t score_stride, int32_t score_col0, float score_thr, float iou_thr, int32_t max_num,
                         float *out_det, int64_t *out_label, int64_t *out_src, int64_t *out_count, void *workspace,
                         size_t workspace_bytes, void *stream);
/* multiclass_nms_kp with test_cfg.nms.type = 'soft_nms' for a whole batch (bbox_nms_kp.py:25-50 ->
 * R/nms/nms_wrapper.py:52-78 -> R/nms/src/soft_nms_cpu.pyx:22-127), two launches, nothing read by the host (round 4: the
 * per-class host loop over kgdet_soft_nms kept config-5 inference eager): per (image, class) the candidates with
 * score > score_thr go through the reference's selection-sort loop in LDS, in ascending candidate row order as
 * `multi_bboxes[cls_inds]` hands them over (method 0 hard / 1 linear / 2 gaussian, `sigma`, `min_score` as soft_nms_cpu);
 * per image the survivors are concatenated class by class in the loop's own order and, beyond max_num, the max_num
 * highest DECAYED scores are kept (ties: earlier first).  Same array shapes as kgdet_multiclass_nms; out_det's fifth
 * column is the decayed score.  Limits: N * 36 bytes of LDS (N <= 4544), C <= 64, C * max_num <= 16384.
 * workspace: kgdet_multiclass_soft_nms_workspace_bytes(B, N, C). */
size_t kgdet_multiclass_soft_nms_workspace_bytes(int32_t B, int32_t N, int32_t C);
/* 1 when kgdet_multiclass_soft_nms serves this problem on chip (N * 36 bytes of LDS per segment, C <= 64, C * max_num keys <=
 * 16384), 0 when it returns KGDET_E_UNSUPPORTED: the one statement of the limits (callers deciding before a graph capture). */
int kgdet_multiclass_soft_nms_supported(int32_t B, int32_t N, int32_t C, int32_t max_num);
int kgdet_multiclass_soft_nms(const float *boxes, const float *scores, int32_t B, int32_t N, int32_t C,
                              int32_t score_stride, int32_t score_col0, float score_thr, float iou_thr,
                              int32_t method, float sigma, float min_score, int32_t max_num, float *out_det,
                              int64_t *out_label, int64_t *out_src, int64_t *out_count, void *workspace,
                              size_t workspace_bytes, void *stream);
int kgdet_nms(const float *dets, int64_t n, float iou_thr, int64_t *keep, int64_t *num_keep,
              void *workspace, size_t workspace_bytes, void *stream);
int kgdet_nms_batched(const float *dets, const int64_t *seg_offsets, int32_t num_segments,
                      int64_t total_n, int64_t max_seg_len, float iou_thr, int64_t *keep,
                      int64_t *num_keep, void *workspace, size_t workspace_bytes, void *stream);
int kgdet_soft_nms(const float *dets, int64_t n, float iou_thr, int32_t method, float sigma,
                   float min_score, float *out_dets, int64_t *out_inds, int64_t *num_out,
                   void *stream);

/* ------------------------------------------------------------------------------------------
 * Moment bounding box ("points2bbox", transform_method='moment';
 * R/../models/anchor_heads/reppoints_head_kp3rep_cas_1_assign_once.py:342-391).
 * pts [B, 2*n_pts, H, W] with (y,x) interleaved per point (y_first != 0) or (x,y); moment_transfer [2] (already
 * blended with its detached copy by the caller); bbox [B, 4, H, W] = (x1,y1,x2,y2).
 * Mean and UNBIASED std (n-1) over the points; half extent = std * exp(transfer).
 * Backward returns grad_pts and grad_transfer[2] (accumulated into a zero-filled buffer).
 * ------------------------------------------------------------------------------------------ */
int kgdet_moment_bbox_forward(const float *pts, const float *moment_transfer, int32_t B, int32_t n_pts,
                              int32_t HW, int32_t y_first, float *bbox, void *stream);
/* backward: grad_pts and grad_transfer [2] are OVERWRITTEN; the two transfer sums leave the blocks as partials in the
 * caller's workspace and are added in block order (no float atomics: bit-repeatable). */
size_t kgdet_moment_bbox_backward_workspace_bytes(int32_t B, int32_t HW);
int kgdet_moment_bbox_backward(const float *pts, const float *moment_transfer, const float *grad_bbox,
                               int32_t B, int32_t n_pts, int32_t HW, int32_t y_first, float *grad_pts,
                               float *grad_transfer, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KGDET_HIP_H_ */
