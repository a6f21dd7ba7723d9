/*
 * advhip.h -- C ABI of libadvhip.so: hand-written gfx950 (MI355X / CDNA4) HIP kernels for the
 * I3D-ResNet50 feature-extraction + MGFN MIL-scoring hot path of
 * jinmang2/anomaly_detection_on_video.
 *
 * The reference has no FFI: its hot path calls torch.nn modules.  Each entry point below names
 * the reference call site(s) it replaces (paths relative to /root/reference).  INTEGRATION.md
 * shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, channels-first (NCDHW / NCT) device pointers owned by
 *     the caller; nothing is allocated, freed or synchronised inside a call (graph-capturable)
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); launches are asynchronous
 *   - return value: 0 on success, a negative ADVHIP_E* code otherwise; no exceptions cross the
 *     ABI; advhip_last_error() returns a thread-local message for the last failure
 *   - no global mutable state besides that message: thread-compatible, one process per GPU
 */
#ifndef ADVHIP_H
#define ADVHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADVHIP_ABI_VERSION 2

#define ADVHIP_OK 0
#define ADVHIP_EINVAL (-1)   /* bad shape / null pointer / unsupported configuration */
#define ADVHIP_ELAUNCH (-2)  /* hipLaunch failed (message has hipGetErrorString) */
#define ADVHIP_ERANGE (-3)   /* tensor too large for 32-bit element indexing */

/* conv algorithm selector (advhip_conv3d_desc.algo) */
#define ADVHIP_ALGO_AUTO 0
#define ADVHIP_ALGO_IGEMM_128x128 1 /* generic implicit GEMM, 128(M) x 128(N) block tile */
#define ADVHIP_ALGO_IGEMM_128x64 2
#define ADVHIP_ALGO_IGEMM_64x64 3
#define ADVHIP_ALGO_IGEMM_64x128 4
#define ADVHIP_ALGO_IGEMM_128x128x32 5 /* same tiles, 32-deep k-tile (longer MFMA run per barrier) */
#define ADVHIP_ALGO_IGEMM_128x64x32 6
#define ADVHIP_ALGO_IGEMM_64x64x32 7
#define ADVHIP_ALGO_IGEMM_64x128x32 8
#define ADVHIP_ALGO_IGEMM_256x64 9 /* 256(M) x 64(N) x 16: four waves stacked along M (64 x 64 wave tiles); ADVHIP_ALGO_DMA2_BASE only */
/* + tile id 1..8: same tiles, gather arithmetic hoisted to scalar offsets + coordinate bit-mask (kernel <= 10^3) */
#define ADVHIP_ALGO_FAST_BASE 32
/* + tile id: the fast gather with LDS-DMA operand staging into a 3-deep ring (no 128x128x32) */
#define ADVHIP_ALGO_DMA_BASE 64
/* + tile id 2..4: the same with a 4-deep ring (three k-tiles in flight) */
#define ADVHIP_ALGO_DMA4_BASE 96
/* + tile id 5 (128x128x32) or 6 (128x64x32): OPT-IN split-bf16 arithmetic -- every fp32 operand as two bf16 terms,
 * three bf16 MFMAs (hi*hi + hi*lo + lo*hi) accumulated in fp32: 3/16 of the fp32 MFMA cycles, agreement with the
 * fp32 kernels ~1e-5 relative (inside the 1e-3 contract, not bit-comparable).  Takes the weights from
 * advhip_conv3d_pack_weight_bf16x3 instead of advhip_conv3d_pack_weight_f32.  Never chosen by ADVHIP_ALGO_AUTO. */
#define ADVHIP_ALGO_BF16X3_BASE 128
/* + tile id 1..4, 6..9: the LDS-DMA kernel with a 2-deep ring (one k-tile in flight): 16 KiB of LDS per 64x64x16
 * workgroup instead of 24 -> 8 resident workgroups per CU instead of 6 */
#define ADVHIP_ALGO_DMA2_BASE 160
/* the 2-deep LDS-DMA kernel, 128x64x16 tile, with m-tiles that span T (2 or 4 frames x 64 / 32 flattened spatial positions)
 * for (kt,1,1) stride-1 "same" convs: the temporal taps of a tile re-read the same activation rows from L1/L2 instead of
 * three far-apart tiles fetching them from HBM.  Unsplit; T even. */
#define ADVHIP_ALGO_TSPAN_128x64 192
/* the 2-deep LDS-DMA kernel for unsplit 1x1x1 stride-1 convs on 16-byte aligned rows, 128x64x16 tiles, with the last m-tile rows cut
 * into 64x64 tiles when the 128x64 tiles number a little more than a whole number of rounds of resident workgroups (6 per compute
 * unit): every workgroup of the last, partial round is then a short one.  Same K order per output as the plain tiles: bit-identical
 * results.  Launches below one round / on a round boundary are the plain 128x64 launch. */
#define ADVHIP_ALGO_MIXED_128x64 200
/* + tile id (ADVHIP_ALGO_IGEMM_128x64 or _64x64) + 8 * (W - 1), W = 1..3 workgroups per compute unit: a PERSISTENT, wave-specialised
 * kernel for unsplit 1x1x1 stride-1 convs on 16-byte aligned rows with K >= 64 (the `conv3` + residual launches,
 * src/i3d.py:85-89, 108-121): eight-wave workgroups that stay for the whole launch and walk a share of the output tiles; four waves
 * only multiply, four fetch the operands (plain loads -> registers -> a two-stage LDS ring, four k-tiles ahead, across tile
 * boundaries) and run the epilogue of the previous tile beside the next tile's MFMAs.  Same k order and accumulation chain as the
 * other families: bit-identical results.  No LayerNorm fold / GELU-backward operand.  OPT-IN and never chosen by ADVHIP_ALGO_AUTO
 * or the tuned table: measured slower than the one-tile kernels on the B = 32 plan (profiles/r04_persist_kernel_study.md). */
#define ADVHIP_ALGO_PERSIST_BASE 224

typedef struct advhip_conv3d_desc {
  int32_t B, Cin, T, H, W;    /* input  (B, Cin, T, H, W) */
  int32_t Cout, kt, kh, kw;   /* weight (Cout, Cin, kt, kh, kw), bias-free */
  int32_t st, sh, sw;         /* stride */
  int32_t pt, ph, pw;         /* zero padding */
  int32_t relu;               /* activation applied last: 0 none, 1 ReLU max(0,.), 2 GELU (erf form, nn.GELU());
                               * advhip_conv3d_bn_act_ex_f32 only: 3 = GELU, with GELU'(pre-activation) -- not the pre-activation -- written to
                               * the epilogue's y_preact (what the backward pass multiplies by); 4 = no activation, the result multiplied by the
                               * epilogue's dact_z tensor AS IS (a tensor saved by a code-3 forward: the GELU backward without erf / exp) */
  int32_t algo;               /* ADVHIP_ALGO_* */
  int32_t splits;             /* split-K factor: 0 = heuristic, 1 = none, n = n K-slices + reduce pass */
} advhip_conv3d_desc;

/* --- library ------------------------------------------------------------------------------ */
int advhip_abi_version(void);
const char* advhip_last_error(void);
/* "gfx950" -- the only code object in the library */
const char* advhip_target_arch(void);

/* --- I3D backbone ---------------------------------------------------------------------------
 * Output extents of a conv / pool (floor mode), the torch formula. */
int advhip_conv3d_out_dims(const advhip_conv3d_desc* d, int32_t* To, int32_t* Ho, int32_t* Wo);

/* Rows of the packed weight matrix: K = Cin*kt*kh*kw rounded up to a multiple of 32. */
int advhip_conv3d_packed_rows(const advhip_conv3d_desc* d);

/* Pack torch-layout weights w[Cout][Cin][kt][kh][kw] into the kernels' [Kpad][Cout] layout
 * (row k = ((ci*kt+dt)*kh+dh)*kw+dw, zero rows above K).  Done once per layer at load time: the
 * load-time analogue of model.load_state_dict (src/i3d.py:356-359). */
int advhip_conv3d_pack_weight_f32(const advhip_conv3d_desc* d, const float* w, float* w_packed,
                                  void* stream);

/* Split torch-layout fp32 weights into the bf16 images of the ADVHIP_ALGO_BF16X3_* kernels:
 * w_split = uint16[2][Cout][Kpad] (hi image, then lo image; k contiguous, zero above K); 4*Cout*Kpad bytes. */
int advhip_conv3d_pack_weight_bf16x3(const advhip_conv3d_desc* d, const float* w, void* w_split, void* stream);

/* Per-row gather table for one input size (d->T, d->H, d->W):
 * ktab[k] = {input element offset of tap k relative to the window origin, dt, dh, dw}; rows
 * above K are marked out-of-range so they contribute exact zeros, followed by the fast kernels'
 * compact table {byte offset, tap bit}.  int32[Kpad][4] + int32[Kpad][2] = 6*Kpad int32. */
int advhip_conv3d_build_ktab(const advhip_conv3d_desc* d, int32_t* ktab, void* stream);

/* Fold eval-mode BatchNorm into per-channel scale/shift:
 *   scale = gamma / sqrt(var + eps), shift = beta - mean * scale.
 * Replaces nn.BatchNorm3d in eval mode (src/i3d.py:75,84,88,210,271). */
int advhip_bn_fold_f32(const float* gamma, const float* beta, const float* mean, const float* var,
                       float eps, int32_t C, float* scale, float* shift, void* stream);

/* y = act( conv3d(x, w) * scale[c] + shift[c] (+ residual) )  -- one fused launch.
 * Replaces the nn.Conv3d -> nn.BatchNorm3d -> (+=residual) -> nn.ReLU module sequences of
 * Bottleneck.forward (src/i3d.py:98-121), the stem (src/i3d.py:303-305) and the downsample
 * branch (src/i3d.py:262-272).  `residual` may be NULL; otherwise it has y's shape.
 * fp32 MFMA (v_mfma_f32_16x16x4_f32): exact fp32 products, fp32 accumulation.
 * `workspace`: caller-owned device scratch of at least advhip_conv3d_workspace_bytes(d) bytes
 * (0 unless the (pinned or heuristic) configuration uses split-K: partial-sum slabs that a second
 * launch reduces in a fixed order, so results stay run-to-run bit-identical). */
int64_t advhip_conv3d_workspace_bytes(const advhip_conv3d_desc* d);
int advhip_conv3d_bn_act_f32(const advhip_conv3d_desc* d, const float* x, const float* w_packed,
                             const int32_t* ktab, const float* scale, const float* shift,
                             const float* residual, float* y, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* The same with explicit batch strides (in elements; 0 = dense): x and/or y may be a channel slice of a wider NCDHW
 * tensor, e.g. one half of a concatenation buffer -- how the layer-1 downsample branch is folded into conv3 (one conv
 * over [x ; h] written by their producers into one buffer, no torch.cat, no residual round trip).  The residual and the
 * split-K slabs stay dense; a strided y runs unsplit. */
int advhip_conv3d_bn_act_strided_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                     const float* w_packed, const int32_t* ktab, const float* scale, const float* shift,
                                     const float* residual, float* y, int64_t y_batch_stride, void* workspace,
                                     int64_t workspace_bytes, void* stream);

/* The same with the extra epilogue operands the MGFN scorer's GEMM-shaped layers need, all optional (a 1x1 Conv1d over a
 * (C, B*T) activation is this conv on a (1, C, 1, 1, B*T) tensor; src/models/mgfn/modeling_mgfn.py:49-64, 150-205):
 *   y_preact (nullable, y's shape / batch stride): receives the value BEFORE the activation -- z of h = GELU(z), kept for
 *            the backward pass while y gets h (MGFNFeedForward in_conv -> GELU, :53-56);
 *   dact_z   (nullable, y's shape, dense): the result is multiplied by GELU'(dact_z) -- the backward of that GELU fused into
 *            the GEMM that produces dL/dh (dL/dz = (W_out^T . dL/dy) * GELU'(z)).
 *   ln_u / ln_mu / ln_rs (nullable, together): the channel-LayerNorm fold for a 1x1x1 conv -- with w_packed holding
 *            W.diag(g) and ln_u[n] its row sums, v = acc * ln_rs[m] - ln_u[n] * ln_mu[m] * ln_rs[m] turns the conv of the RAW
 *            activation into W.LN(x) minus the W.b term (put W.b + bias in `shift`): MGFNLayerNorm (:36-46) never
 *            materialises; ln_mu / ln_rs are per position m (advhip_chan_stats_f32).
 * A zero-filled struct = advhip_conv3d_bn_act_strided_f32. */
typedef struct advhip_conv3d_epilogue {
  float* y_preact;
  const float* dact_z;
  const float* ln_u;
  const float* ln_mu;
  const float* ln_rs;
  /* ABI 2.  Nullable: a caller-owned block of arrival counters for the in-launch split-K reduction, 4 bytes per output tile
   * (65 536 bytes cover every shape of the I3D / MGFN plans).  Contract: ZERO when first handed over, used by one stream at a
   * time, never written by the caller afterwards -- each launch leaves it zero (the last arriver of a tile resets its word), so
   * no memset node runs ahead of the launch.  Without it (or if it is too small) the counters live at the head of
   * `workspace` and are cleared by a memset before every launch. */
  void* splitk_counters;
  int64_t splitk_counter_bytes;
  /* ABI 2.  Nullable: (B, Cout) -- the launch writes the mean of act(conv * scale + shift (+ residual)) over each sample's
   * positions here INSTEAD of y (which may then be null): layer4's last conv3 + bn3 + residual + ReLU and the
   * nn.AdaptiveAvgPool3d((1,1,1)) behind it (src/i3d.py:111-121, 314) in one launch.  1x1x1 stride-1 convs on <= 128
   * positions per sample, Cout % 64 == 0, x 16-byte aligned, none of the operands above; bit for bit
   * advhip_global_avgpool_f32 of the conv's own output. */
  float* avgpool_out;
} advhip_conv3d_epilogue;
int advhip_conv3d_bn_act_ex_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride, const float* w_packed,
                                const int32_t* ktab, const float* scale, const float* shift, const float* residual, float* y,
                                int64_t y_batch_stride, const advhip_conv3d_epilogue* ep, void* workspace,
                                int64_t workspace_bytes, void* stream);

/* mu[n] = mean over channels of x[c, n], rs[n] = 1 / (sqrt(biased variance over channels) + eps) for a (C, N) activation:
 * the statistics of MGFNLayerNorm (modeling_mgfn.py:43-46: division by std + eps, not sqrt(var + eps)). */
int advhip_chan_stats_f32(const float* x, float* mu, float* rs, int32_t C, int64_t N, float eps, void* stream);

/* --- conv + max-pool fused (the two nn.MaxPool3d of I3Res50.forward_single, src/i3d.py:303-309) -------------------
 * Pooled extents of conv(d) followed by a floor-mode, padding-0 max-pool with window pk* and stride ps*. */
int advhip_conv3d_pool_out_dims(const advhip_conv3d_desc* d, int32_t pkt, int32_t pkh, int32_t pkw, int32_t pst,
                                int32_t psh, int32_t psw, int32_t* Tp, int32_t* Hp, int32_t* Wp);

/* y = MaxPool3d(k=(2,3,3), s=(2,2,2), p=0)( relu( conv3d(x, w) * scale + shift ) ): the stem conv1 -> bn1 -> relu ->
 * maxpool1 of src/i3d.py:303-306 without writing the un-pooled activation (822 MB at B = 32) to HBM.  The conv runs on
 * m-tiles that are 2(t) x 4(h) x 16(w) bricks of output positions; each brick writes the maxima of the pooling windows
 * it touches (27 per channel) to `workspace`, and a small second launch maxes the 1, 2 or 4 partials of every pooled
 * output: the same fp32 conv values as the unfused pair, so the result is bit-identical to advhip_conv3d_bn_act_f32
 * (relu) + advhip_maxpool3d_f32.  x / y may be channel slices of wider tensors (batch strides in elements, 0 = dense).
 * d->relu and d->algo / d->splits are ignored (ReLU is part of the op; one tile configuration). */
int64_t advhip_conv3d_relu_maxpool233_workspace_bytes(const advhip_conv3d_desc* d);
int advhip_conv3d_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                         const float* w_packed, const int32_t* ktab, const float* scale, const float* shift,
                                         float* y, int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                         void* stream);

/* --- the same stem with 16-byte gather pieces: column-parity planes of the input ------------------------------------------------
 * The stem's stride-2 window along w makes its A gather one 4-byte LDS-DMA per (tap, position), and the issue slots those
 * instructions share with the MFMAs are what bounds the kernel.  With the input rewritten as column-parity planes
 *   xs[b, c, t, h, par, 2 + j] = x[b, c, t, h, 2 j + par]   (advhip_split_w_f32; WP = advhip_split_w_plane_floats(W) = W/2 + 4
 *   floats per plane, zero columns either side)
 * tap dw of four consecutive output columns reads four CONSECUTIVE floats of plane (dw - pw) & 1: one 16-byte piece per lane,
 * two whole k-rows of the tile per wave-instruction (4x fewer A instructions).  Same K order, operands and accumulation as
 * advhip_conv3d_bn_relu_maxpool233_f32: bit-identical results.  Needs stride 2, an odd kernel <= 9 with padding kw/2 along w,
 * W even and an output width that is a multiple of 4 (the I3D stem: 7 / 2 / 3, 224 -> 112).  ktab_s2w: int32[2 * Kpad]. */
int32_t advhip_split_w_plane_floats(int32_t W);
int advhip_split_w_f32(const float* x, float* xs, int64_t rows, int32_t W, void* stream);
int advhip_conv3d_s2w_build_ktab(const advhip_conv3d_desc* d, int32_t* ktab_s2w, void* stream);
int advhip_conv3d_s2w_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const float* xs, int64_t xs_batch_stride,
                                             const float* w_packed, const int32_t* ktab_s2w, const float* scale, const float* shift,
                                             float* y, int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                             void* stream);

/* TenCrop + float + normalise + the layout permutes (advhip_tencrop_normalize_u8 below) writing those planes directly for
 * crop-clips [first_crop_clip, first_crop_clip + count) of a video: xs (count, C, frames_per_clip, crop, 2, crop/2 + 4).  The
 * real-data path in front of advhip_conv3d_s2w_bn_relu_maxpool233_f32: the planes cost the pass nothing extra, and the values
 * are those of advhip_tencrop_normalize_u8 (same arithmetic per pixel), so the features equal the fp32 pipeline's bit for bit. */
int advhip_tencrop_normalize_planes_u8(const uint8_t* frames, float* xs, int32_t F, int32_t H, int32_t W, int32_t C,
                                       int32_t frames_per_clip, int32_t crop, int64_t first_crop_clip, int64_t count, float mean,
                                       float stdv, void* stream);

/* --- the same stem, fed by resized uint8 frames (src/gtransforms.py:29-38,57-73 + extract_features.py:83-89 in the load stage)
 * frames: uint8 (F, FH, FW, Cin) -- what the decoder + GroupResize hand over -- F a whole number of clips of d->T frames.
 * Sample b of the launch (d->B of them) is crop-clip first_crop_clip + b = clip * 10 + crop in torchvision's TenCrop order
 * (top-left, top-right, bottom-left, bottom-right, centre, then the same five mirrored along w); d->H / d->W are the crop size.
 * The gather reads the crop's pixels as bytes (`buffer_load_ubyte ... lds`), the operand of the matrix pipe is the pixel value
 * (exact in fp32), sum w (pixel - mean) = acc - mean * (sum of the weights of the taps inside the clip) with that sum tabulated
 * per border class, and 1/std is folded into the BN scale: neither the fp32 ten-crop tensor (385 MB per 40 crop-clips) nor
 * the un-pooled stem output exists.
 *   advhip_conv3d_u8_table_sizes: element counts of the two tables below (int32 / float)
 *   advhip_conv3d_u8_build_tables: ktab_u8 = gather offsets (as stored + mirrored); corr = -mean * (sum of the weights of the
 *     taps inside the clip) per border class (per dimension: taps before the clip * (p + 1) + taps past its end) and channel;
 *     once per (weights, FH, FW, clip dims)
 * Result: within fp32 rounding of advhip_tencrop_normalize_u8 + advhip_conv3d_bn_relu_maxpool233_f32 (tests: 2e-5). */
int advhip_conv3d_u8_table_sizes(const advhip_conv3d_desc* d, int64_t* ktab_ints, int64_t* corr_floats);
int advhip_conv3d_u8_build_tables(const advhip_conv3d_desc* d, int32_t FH, int32_t FW, const float* w_packed, float mean,
                                  int32_t* ktab_u8, float* corr, void* stream);
int advhip_conv3d_u8_tencrop_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const uint8_t* frames, int64_t F, int32_t FH,
                                                    int32_t FW, int64_t first_crop_clip, const float* w_packed,
                                                    const int32_t* ktab_u8, const float* corr,
                                                    const float* scale, const float* shift, float stdv, float* y,
                                                    int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                                    void* stream);

/* The same stem from WHOLE PIXELS: K runs tap-major (k' = tap * 3 + c), one 4-byte LDS-DMA per (tap, position) fetches the
 * pixel's three channel bytes at the pixel's byte address (a third of the gather instructions of the byte form for the same
 * K; LDS-DMA and ds_read take byte-granular addresses on gfx950), the byte select is part of the LDS read address.  Cin = 3,
 * Cout = 64.  `readable_bytes`: bytes readable from `frames` on; must be >= F*FH*FW*3 + 1 (the last pixel's 4-byte piece).
 *   w_taps: the weights re-ordered tap-major (built from w_packed), ktab_taps: pixel offsets per tap (as stored + mirrored). */
int advhip_conv3d_u8_taps_table_sizes(const advhip_conv3d_desc* d, int64_t* ktab_ints, int64_t* corr_floats, int64_t* w_taps_floats);
int advhip_conv3d_u8_taps_build_tables(const advhip_conv3d_desc* d, int32_t FH, int32_t FW, const float* w_packed, float mean,
                                       int32_t* ktab_taps, float* corr, float* w_taps, void* stream);
int advhip_conv3d_u8_taps_tencrop_bn_relu_maxpool233_f32(const advhip_conv3d_desc* d, const uint8_t* frames, int64_t F, int32_t FH,
                                                         int32_t FW, int64_t readable_bytes, int64_t first_crop_clip,
                                                         const float* w_taps, const int32_t* ktab_taps, const float* corr,
                                                         const float* scale, const float* shift, float stdv, float* y,
                                                         int64_t y_batch_stride, void* workspace, int64_t workspace_bytes,
                                                         void* stream);

/* y = MaxPool3d(k=(2,1,1), s=(2,1,1))( act( conv3d(x, w) * scale + shift (+ residual) ) ) for a 1x1x1 stride-1 conv
 * (Cin a multiple of 32) in ONE launch: the last Bottleneck of layer1 followed by maxpool2 (src/i3d.py:111-121, 309).
 * Each m-tile holds both frames of a pooling pair, so the pooling is exact inside the epilogue: the un-pooled
 * activation is neither written nor read back.  `residual` (nullable) has the UN-pooled conv output's shape, dense. */
int advhip_conv3d_bn_act_maxpool211_f32(const advhip_conv3d_desc* d, const float* x, int64_t x_batch_stride,
                                        const float* w_packed, const int32_t* ktab, const float* scale, const float* shift,
                                        const float* residual, float* y, int64_t y_batch_stride, void* stream);

/* nn.MaxPool3d with zero padding=0, floor mode (src/i3d.py:212-217, 306, 309). */
int advhip_maxpool3d_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t H,
                         int32_t W, int32_t kt, int32_t kh, int32_t kw, int32_t st, int32_t sh,
                         int32_t sw, void* stream);

/* nn.MaxPool3d with padding (implicit -inf, at most half the window; floor mode): the stem pool k(1,3,3) s(1,2,2) p(0,1,1)
 * of the pytorchvideo ResNet that `i3d_8x8_r50` names (src/i3d.py:339-350).  Dense x and y. */
int advhip_maxpool3d_padded_f32(const float* x, float* y, int32_t B, int32_t C, int32_t T, int32_t H, int32_t W, int32_t kt,
                                int32_t kh, int32_t kw, int32_t st, int32_t sh, int32_t sw, int32_t pt, int32_t ph, int32_t pw,
                                void* stream);

/* The same, y being a channel slice of a wider tensor (y_batch_stride in elements, 0 = dense). */
int advhip_maxpool3d_strided_f32(const float* x, float* y, int64_t y_batch_stride, int32_t B, int32_t C, int32_t T, int32_t H,
                                 int32_t W, int32_t kt, int32_t kh, int32_t kw, int32_t st, int32_t sh, int32_t sw, void* stream);

/* nn.AdaptiveAvgPool3d((1,1,1)) (src/i3d.py:244, 314): x (rows, n) -> y (rows), mean over n. */
int advhip_global_avgpool_f32(const float* x, float* y, int64_t rows, int32_t n, void* stream);

/* --- batched strided GEMM (fp32 MFMA) ---------------------------------------------------------------------------------
 * C[b,m,n] = epi( alpha * sum_k A[b,m,k] * B[b,k,n] ), element (b,m,k) of A at A + b*sAb + m*sAm + k*sAk (strides in
 * elements; one of sAm / sAk and one of sBk / sBn must be 1), likewise B and C.  Replaces the torch.bmm pair of
 * NonLocalBlock.forward (src/i3d.py:171-178) and the Conv1d / Linear / einsum contractions of the MGFN scorer and their
 * backward products (src/models/mgfn/modeling_mgfn.py:49-64, 96-123, 150-205).  Epilogue, in this order (every pointer
 * nullable): LayerNorm fold  v = v*ln_rs[b,n] - ln_u[m]*ln_mu[b,n]*ln_rs[b,n]  (W.diag(g) applied to RAW columns x equals
 * W.LN(x) minus the bias term when ln_u = W.g row sums: MGFNLayerNorm over channels, modeling_mgfn.py:36-46);
 * + bias_m[m] + bias_n[n];  act (0 none, 1 ReLU, 2 GELU-erf);  + beta * residual (C's strides). */
typedef struct advhip_gemm_desc {
  int32_t M, N, K, batch;
  int64_t sAb, sAm, sAk;
  int64_t sBb, sBk, sBn;
  int64_t sCb, sCm, sCn;
  float alpha, beta;
  int32_t act;
  const float* bias_m;
  const float* bias_n;
  const float* ln_u;
  const float* ln_mu;
  const float* ln_rs;
  const float* residual;
} advhip_gemm_desc;
int advhip_bgemm_f32(const advhip_gemm_desc* d, const float* A, const float* B, float* C, void* stream);

/* C[s][m][n] = sum over K slice s of A[m][k] * B[n][k]: both operands k-contiguous with row pitches lda / ldb (elements, any
 * value >= K: a 16-byte LDS-DMA piece takes any 4-byte aligned address on gfx950; K a multiple of 16; 4-byte aligned bases).
 * The weight gradient dW[o][c] = sum_n dY[o][n] X[c][n] of the MGFN scorer's GEMM-shaped layers, whose activations are stored
 * (channel, position): no transposed copies; and the token conv's tap products on the scorer's input rows as stored (2 049-float
 * pitch, modeling_mgfn.py:81-93).  LDS-DMA row copies, fp32 MFMA.  splits > 1 cuts K into slices written to C + s * slab_stride (the caller sums them). */
int advhip_gemm_nt_f32(const float* A, const float* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb,
                       int64_t ldc, int32_t splits, int64_t slab_stride, void* stream);

/* The same product with the row sums of A beside it: rowsum_a[s][m] = sum over K slice s of A[m][k] (nullable) -- the bias
 * gradient db[o] = sum_n dY[o][n] of the same layer (autograd of nn.Conv1d's bias, modeling_mgfn.py:53-56, 101, 155, 167) out of
 * the fragments the n-tile-0 workgroups hold anyway, instead of a second pass over dY.  `tile`: 0 = heuristic, 1 = 64 x 64,
 * 2 = 128 x 64, 3 = 128 x 128 output tile per workgroup. */
int advhip_gemm_nt_rowsum_f32(const float* A, const float* B, float* C, float* rowsum_a, int32_t M, int32_t N, int32_t K,
                              int64_t lda, int64_t ldb, int64_t ldc, int32_t splits, int64_t slab_stride, int32_t tile,
                              void* stream);

/* The same with the K slices summed INSIDE the launch: every (tile, slice) workgroup publishes its partial tile (write-through
 * stores) into `workspace`, the workgroup that arrives last at the tile's counter sums the slices in slice order (run-to-run
 * bit-identical) and writes C (M x N, pitch ldc) and rowsum_a (M, nullable) once -- no slabs, no second pass.
 * `workspace`: 256-byte aligned, advhip_gemm_nt_workspace_bytes(M, N, splits, tile) bytes, whose first 64 KiB (the arrival
 * counters: at most 16 384 output tiles) are ZERO on entry; the launch leaves them zero, so a workspace zero-filled once
 * serves every later launch -- of any shape -- on the same stream. */
int64_t advhip_gemm_nt_workspace_bytes(int32_t M, int32_t N, int32_t splits, int32_t tile);
int advhip_gemm_nt_reduced_f32(const float* A, const float* B, float* C, float* rowsum_a, int32_t M, int32_t N, int32_t K,
                               int64_t lda, int64_t ldb, int64_t ldc, int32_t splits, int32_t tile, void* workspace,
                               int64_t workspace_bytes, void* stream);

/* Few output tiles (the 64- and 128-channel layers' weight gradients: M x N of a few 64 x 64 tiles, K = all positions): there
 * the last arriver's serial sum is the whole tail of the launch, so the slices go to `slabs` = [splits][M*N (+ M row sums when
 * with_rowsum)] and advhip_sum_slabs_f32 (dst[i] = sum in slice order of src[s*stride + i]) reduces product AND row sums
 * in one further launch. */
int advhip_gemm_nt_slabs_f32(const float* A, const float* B, float* slabs, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb,
                             int32_t splits, int32_t tile, int32_t with_rowsum, void* stream);
int advhip_sum_slabs_f32(const float* slabs, float* out, int64_t n, int32_t splits, int64_t stride, void* stream);
/* Many such products in ONE launch: item i writes slice s of A_i B_i^T (64 x 64 tiles, as advhip_gemm_nt_slabs_f32 with tile 1) to
 * C_i + s * slab_stride (row pitch N_i) and, when rowsum_i is given, the slice's row sums of A_i to rowsum_i + s * slab_stride;
 * all items contract over the same K in the same number of slices.  With C_i / rowsum_i laid out back to back inside one
 * [splits][slab_stride] matrix, ONE advhip_sum_slabs_f32 reduces every product and every row sum.  `items` is a HOST array (it
 * travels in the kernel arguments, 32 items per launch): the weight and bias gradients of all narrow layers of an MGFN training
 * step (autograd of nn.Conv1d, modeling_mgfn.py:49-64,101-108,150-193) at the end of the backward pass. */
typedef struct advhip_nt_item {
  const float* A;  /* (M, K), row pitch lda */
  const float* B;  /* (N, K), row pitch ldb */
  float* C;
  float* rowsum;   /* nullable */
  int32_t M, N;
  int64_t lda, ldb;
} advhip_nt_item;
int advhip_gemm_nt_group_slabs_f32(const advhip_nt_item* items, int32_t n_items, int32_t K, int32_t splits, int64_t slab_stride, void* stream);

/* y[r, :] = softmax(x[r, :] * scale) over rows of n contiguous floats (F.softmax(theta_phi * dim_inner**-0.5, dim=-1),
 * src/i3d.py:174-175; the attention softmax of GlanceAttention, modeling_mgfn.py:115-120).  x == y allowed. */
int advhip_softmax_rows_f32(const float* x, float* y, int64_t rows, int32_t n, float scale, void* stream);

/* --- MGFN body: fused element-wise / reduction kernels on (C, N) activations (channels outermost) --------------------
 * MGFNLayerNorm over the channel dim (modeling_mgfn.py:36-46): y = (x - mean_c) / (sqrt(var_biased_c) + eps) * g[c] + b[c]
 * per position n; mu / rs (= 1 / (std + eps)) per position are returned for the backward pass. */
int advhip_chan_layernorm_fwd_f32(const float* x, const float* g, const float* b, float* y, float* mu, float* rs, int32_t C,
                                  int64_t N, float eps, void* stream);
/* Its backward: dx (C, N), and per-block partial sums of dg / db as [advhip_chan_layernorm_bwd_partial_rows(N)][C] matrices
 * (the caller sums the rows: fixed order, no atomics). */
int64_t advhip_chan_layernorm_bwd_partial_rows(int64_t N);
int advhip_chan_layernorm_bwd_f32(const float* dy, const float* x, const float* g, const float* mu, const float* rs, float* dx,
                                  float* dg_partial, float* db_partial, int32_t C, int64_t N, float eps, void* stream);

/* nn.BatchNorm1d in training mode on a (C, N) activation (FocusAttention.norm, modeling_mgfn.py:162, 174): batch statistics
 * per channel row, y = (x - mean) * rsqrt(var_biased + eps) * gamma + beta; mean / var are returned (running-statistics
 * update by the caller, backward).  Backward: dx, dgamma, dbeta. */
int advhip_bn_rows_fwd_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var, int32_t C,
                           int64_t N, float eps, void* stream);
int advhip_bn_rows_bwd_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* var, float* dx,
                           float* dgamma, float* dbeta, int32_t C, int64_t N, float eps, void* stream);

/* The same three with the fusions a training step's launch count asks for (bit-identical arithmetic):
 *   advhip_bn_rows_fwd_running_f32: also updates nn.BatchNorm1d's running_mean / running_var in place (nullable, together:
 *     running = (1 - momentum) * running + momentum * batch, the variance unbiased) -- torch's five small launches;
 *   advhip_bn_rows_bwd_add_f32 / advhip_chan_layernorm_bwd_add_f32: dx = backward + add (nullable, dx's shape) -- the
 *     gradient of the block's skip connection (y = f(norm(x)) + x) without autograd's separate add; the LayerNorm form writes
 *     its per-block partial sums as ONE [rows][2C] matrix (dg in columns [0, C), db in [C, 2C)): one reduction for both. */
int advhip_bn_rows_fwd_running_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* var,
                                   float* running_mean, float* running_var, float momentum, int32_t C, int64_t N, float eps,
                                   void* stream);
int advhip_bn_rows_bwd_add_f32(const float* dy, const float* x, const float* gamma, const float* mean, const float* var,
                               const float* add, float* dx, float* dgamma, float* dbeta, int32_t C, int64_t N, float eps, void* stream);
int advhip_chan_layernorm_bwd_add_f32(const float* dy, const float* x, const float* g, const float* mu, const float* rs,
                                      const float* add, float* dx, float* dgb_partial, int32_t C, int64_t N, float eps, void* stream);

/* A whole `x = x + FFN(LN(x))` step of a NARROW block in one launch (MGFNLayerNorm -> Conv1d(C, 4C, 1) -> GELU -> Conv1d(4C, C, 1) -> + x,
 * modeling_mgfn.py:36-64, 147, 205), C = 64 or 128 channels, N a multiple of 64 positions, activations (C, N) with the positions contiguous:
 *   forward : w1 = in_conv.weight (4C, C), w2 = out_conv.weight (C, 4C) AS STORED; writes y and what the backward pass keeps: xh = LN(x),
 *             mu / rs (N), h = GELU(W1 xh + b1) and z = GELU'(W1 xh + b1), both (4C, N);
 *   backward: w2_packed / w1_packed = the forward GEMMs' packed operands [4C][C] / [C][4C] (advhip_conv3d_pack_weight_f32); writes
 *             dz = (W2^T dy) * z (4C, N) -- what the weight gradients dW1 = dz xh^T, db1 = rowsum(dz) contract with (dW2 = dy h^T) --,
 *             dx = LayerNorm backward of W1^T dz + dy (the skip connection), and dgb_partial [advhip_ffn_block_partial_rows(N)][2C]: per-workgroup
 *             partial sums of dg (columns [0, C)) and db ([C, 2C)), to be column-summed (advhip_colsum_f32 / _group).
 * The same arithmetic as advhip_chan_layernorm_fwd_f32 + two advhip_conv3d_bn_act_ex_f32 launches (activation code 3, then bias + residual) and
 * their three backward launches; sums inside a dot product run in another order (agreement ~1e-6). */
int64_t advhip_ffn_block_partial_rows(int64_t N);
int advhip_ffn_block_fwd_f32(const float* x, const float* ln_g, const float* ln_b, float eps, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* xh, float* mu, float* rs, float* h, float* z, float* y, int32_t C,
                             int64_t N, void* stream);
int advhip_ffn_block_bwd_f32(const float* dy, const float* x, const float* ln_g, const float* mu, const float* rs, float eps, const float* z,
                             const float* w2_packed, const float* w1_packed, float* dz, float* dx, float* dgb_partial, int32_t C, int64_t N,
                             void* stream);

/* The packed operand of a Conv1d's transposed conv (its input gradient dX = conv1d(dY; W'), W'[c][o][j] = W[o][c][k-1-j];
 * autograd of nn.Conv1d, modeling_mgfn.py:101,155) straight from the parameter w (Cout, Cin, k):
 * w_packed[(o*k + j)][c] = w[o][c][k-1-j], rows padded with zeros to a multiple of 32 -- one launch instead of flip +
 * transpose + pack. */
int advhip_conv1d_pack_weight_dx_f32(const float* w, float* w_packed, int32_t Cout, int32_t Cin, int32_t k, void* stream);

/* Every packed operand a training step needs, in one launch: item i packs the Conv1d parameter `src` (Cout, Cin, k) into `dst`,
 * mode 0 = advhip_conv3d_pack_weight_f32's [roundup32(Cin*k)][Cout] (the forward GEMM's operand), mode 1 =
 * advhip_conv1d_pack_weight_dx_f32's [roundup32(Cout*k)][Cin] (the input gradient's).  `items_dev` is an array in DEVICE memory;
 * tile_begin = the exclusive prefix sum of advhip_pack_item_tiles over the items, n_tiles their total.  Replaces the per-layer
 * weight re-packs autograd's forward of nn.Conv1d hides inside MIOpen (modeling_mgfn.py:101-108,155,183-193). */
typedef struct advhip_pack_item {
  const float* src;
  float* dst;
  int32_t Cout, Cin, k;
  int32_t mode;
  int32_t tile_begin;
  int32_t reserved;
} advhip_pack_item;
int64_t advhip_pack_item_tiles(int32_t Cout, int32_t Cin, int32_t k, int32_t mode);
int advhip_pack_weights_multi_f32(const advhip_pack_item* items_dev, int32_t n_items, int32_t n_tiles, void* stream);

/* MGFNFeatureAmplifier (modeling_mgfn.py:81-93) after the GEMM of the stacked tap matrices: y[o, r, t] = sum_j z[j, o, r, t + j - 1]
 * (zero outside [0, T)) + bias[o] + ratio * (sum_j wm[o][j] * mag[r, t + j - 1] + bm[o]) for z (3, O, rows, T), mag = the magnitude
 * channel read in place: element (r, t) at mag[(r * T + t) * mag_stride].  Backward: dz (3, O, rows, T), d_bias (O), d_wm (O, 3),
 * d_bm (O) -- the magnitude itself gets no gradient (it is input data). */
int advhip_amp_combine_fwd_f32(const float* z, const float* bias, const float* mag, int64_t mag_stride, const float* wm, const float* bm,
                               float ratio, float* y, int32_t O, int64_t rows, int32_t T, void* stream);
int advhip_amp_combine_bwd_f32(const float* dy, const float* mag, int64_t mag_stride, float ratio, float* dz, float* d_bias, float* d_wm,
                               float* d_bm, int32_t O, int64_t rows, int32_t T, void* stream);

/* dst[c] = sum over r, in row order, of src[r][c]: the per-block partial sums the backward kernels above leave to the caller
 * (rows = a few hundred blocks). */
int advhip_colsum_f32(const float* src, float* dst, int64_t rows, int32_t cols, void* stream);
/* The same for many matrices in one launch (`items`: a HOST array, 64 per launch, carried in the kernel arguments): all partial-sum
 * matrices of a backward pass at its end.  period > 0: the columns are [cols / period groups][period] and leave de-interleaved --
 * element j < period - 1 of group h at dst[h * (period - 1) + j], the last element of every group at
 * dst[(cols / period) * (period - 1) + h]: advhip_dwconv_t_bwd_f32's per-head (K filter taps | bias) sums as the filter gradient
 * followed by the bias gradient (period = K + 1). */
typedef struct advhip_colsum_item {
  const float* src;
  float* dst;
  int64_t rows;
  int32_t cols;
  int32_t period;
} advhip_colsum_item;
int advhip_colsum_group_f32(const advhip_colsum_item* items, int32_t n_items, void* stream);

/* torch.optim.Adam's update (L2 weight decay added to the gradient, no amsgrad; /root/reference/src/runner.py:53-59) for many fp32
 * tensors in one launch per 80 of them (`items`: a HOST array, carried in the kernel arguments -- graph-capturable as is):
 *   g' = g + wd p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g'^2;  p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
 * (fp32 arithmetic; 1 - beta and lr / (1 - beta^t) formed in double and rounded once, as torch's Python code does)
 * with t = *step: a device-side fp32 counter (torch's capturable Adam keeps one per parameter) that the caller has already
 * incremented for this step. */
typedef struct advhip_adam_item {
  float* param;
  const float* grad;
  float* exp_avg;
  float* exp_avg_sq;
  const float* step;
  int64_t n;
} advhip_adam_item;
int advhip_adam_multi_f32(const advhip_adam_item* items, int32_t n_items, double lr, double beta1, double beta2, double eps,
                          double weight_decay, void* stream);

/* The operand of a k = 3, padding 1 Conv1d's weight gradient (autograd of nn.Conv1d, modeling_mgfn.py:101,155):
 * u[(c*3 + j), r, t] = x[c, r, t + j - 1] (zero outside [0, T)), x (C, rows, T) -> u (3C, rows, T); dW = dY . u^T by
 * advhip_gemm_nt_f32.  T a multiple of 4. */
int advhip_unfold3_f32(const float* x, float* u, int32_t C, int64_t rows, int32_t T, void* stream);

/* FocusAttention.rel_pos (modeling_mgfn.py:169-171, 176-178): depth-wise temporal conv, one K-tap filter per head, on a
 * (C, rows, T) activation whose channel c belongs to head c % H: out[c,r,t] = bias[h] + sum_j w[h][j] * v[c,r,t+j-K/2]
 * (zero padding).  K in {3, 5}. */
int advhip_dwconv_t_fwd_f32(const float* v, const float* w, const float* bias, float* out, int32_t C, int32_t H, int64_t rows,
                            int32_t T, int32_t K, void* stream);
/* Backward: dv, and partial[(C / H) * chunks][H][K + 1] = per-block sums of (dout * v shifted by tap j, j < K; dout) of channel
 * c = c_idx * H + h -- the caller adds the (C / H) * chunks rows (advhip_colsum_f32; chunks = advhip_dwconv_t_bwd_chunks(C, rows)). */
int32_t advhip_dwconv_t_bwd_chunks(int32_t C, int64_t rows);
int advhip_dwconv_t_bwd_f32(const float* dout, const float* v, const float* w, float* dv, float* partial, int32_t C, int32_t H,
                            int64_t rows, int32_t T, int32_t K, void* stream);

/* GlanceAttention's core on (C, B, T) activations (modeling_mgfn.py:113-122: q * scale, sim = q^T k, softmax over the keys, out =
 * v attn^T, "b h n d -> b (h d) n"), T = 32 clips, dim_head = 64: qkv (3 * heads * 64, B, T) -- the to_qkv conv's output,
 * q rows then k rows then v rows -- -> out (heads * 64, B, T), the softmax p (B, heads, T, T) kept for the backward pass; one
 * workgroup per (sequence, head).  Backward: dqkv from dout, qkv and p.  torch: scale + bmm + softmax + bmm + layout copy
 * forward, ten launches backward. */
int advhip_glance_attention_fwd_f32(const float* qkv, float* out, float* p, int32_t heads, int64_t B, int32_t T, int32_t dim_head,
                                    float scale, void* stream);
int advhip_glance_attention_bwd_f32(const float* dout, const float* qkv, const float* p, float* dqkv, int32_t heads, int64_t B,
                                    int32_t T, int32_t dim_head, float scale, void* stream);

/* The same core for ANY T (dim_head = 64) -- the validation pass scores a whole video, T = n_clips
 * (/root/reference/src/runner.py:42-50 -> modeling_mgfn.py:107-123): one workgroup per (32-query tile, sequence, head), key / value
 * tiles of 32 clips through LDS with an online softmax; the T x T attention matrix never exists in memory.  lse (B, heads, T) =
 * log-sum-exp of every row of scale * q^T k (nullable: inference) replaces p for the backward pass, which recomputes the softmax
 * tile by tile from qkv, out and lse (dq per query tile, dk / dv per key tile: every output written once, sums in tile order). */
int advhip_glance_attention_fwd_anyt_f32(const float* qkv, float* out, float* lse, int32_t heads, int64_t B, int32_t T, int32_t dim_head,
                                         float scale, void* stream);
int advhip_glance_attention_bwd_anyt_f32(const float* dout, const float* qkv, const float* out, const float* lse, float* dqkv, int32_t heads,
                                         int64_t B, int32_t T, int32_t dim_head, float scale, void* stream);

/* A per-input-channel affine map x -> mul[c] x[c] + add[c] folded into the 1x1 layer W (O, C) (+ bias, nullable) that follows it:
 *   Wf[o][c] = W[o][c] mul[c];  bias_f[o] = bias[o] + sum_c W[o][c] add[c] (add nullable: 0);  rowsum[o] = sum_c Wf[o][c] (nullable).
 * Eval-mode nn.BatchNorm1d in front of FocusAttention.to_v (modeling_mgfn.py:162, 173-174; mul / add = advhip_bn_fold_f32's scale / shift)
 * and MGFNLayerNorm's (g, b) in front of MGFNFeedForward.in_conv at inference (modeling_mgfn.py:36-64): operand-build time, once per
 * set of weights -- the scoring pass itself then has no normalisation launch and no torch arithmetic for these layers. */
int advhip_fold_affine_f32(const float* W, const float* mul, const float* add, const float* bias, float* Wf, float* bias_f, float* rowsum,
                           int32_t O, int32_t C, void* stream);

/* The scorer's head on the body's (C, N) layout (modeling_mgfn.py:387-389: permute -> nn.LayerNorm(C) -> nn.Linear(C, 1) -> sigmoid):
 * xn (N, C) = LayerNorm over C of y (C, N) -- transposed through LDS, so that the MIL head reads rows --, score[n] =
 * sigmoid(xn[n, :] . fc_w + fc_b[0]); mean / rstd (N) kept for the backward pass.  Backward: dy (C, N) from d_xn (N, C) and
 * d_score (N) (either nullable), and partial[advhip_head_ln_fc_partial_rows(N)][3 C + 1] = per-block sums of
 * (d ln_g | d ln_b | d fc_w | d fc_b) for the caller to add up. */
int64_t advhip_head_ln_fc_partial_rows(int64_t N);
int advhip_head_ln_fc_fwd_f32(const float* y, const float* ln_g, const float* ln_b, const float* fc_w, const float* fc_b, float* xn,
                              float* mean, float* rstd, float* score, int32_t C, int64_t N, float eps, void* stream);
int advhip_head_ln_fc_bwd_f32(const float* d_xn, const float* d_score, const float* y, const float* ln_g, const float* ln_b,
                              const float* fc_w, const float* mean, const float* rstd, const float* score, float* dy, float* partial,
                              int32_t C, int64_t N, void* stream);

/* --- MIL scorer (MGFN head) -----------------------------------------------------------------
 * Fused magnitude / score reduction of magnitude_selection_and_score_prediction
 * (src/models/mgfn/modeling_mgfn.py:314-319): for features (bs*ncrops, T, F) and per-crop scores
 * (bs*ncrops, T): mag[b,t] = mean_c ||features[b*ncrops+c, t, :]||_2, sc[b,t] = mean_c scores. */
int advhip_mil_magnitude_f32(const float* features, const float* scores, float* mag, float* sc,
                             int32_t bs, int32_t ncrops, int32_t T, int32_t F, void* stream);

/* Top-k over T of mag*keep (ties -> lowest index first, as torch.topk on CPU/ROCm for distinct
 * values; modeling_mgfn.py:345-346), gather of the k selected feature rows for every crop in
 * crop-major order (modeling_mgfn.py:349-355) and mean of the k selected scores (:359-362).
 *   mag, keep, sc : (n, T)         features : (n*ncrops, T, F)   [row = video*ncrops + crop]
 *   idx  : (n, k) int64            sel      : (ncrops*n, k, F)   [row = crop*n + video]
 *   score: (n)                     keep may be NULL (all ones).   k <= min(16, T); T has no bound but int32 (n * T elements addressed as size_t). */
int advhip_mil_topk_select_f32(const float* mag, const float* keep, const float* sc,
                               const float* features, int64_t* idx, float* sel, float* score,
                               int32_t n, int32_t ncrops, int32_t T, int32_t F, int32_t k,
                               void* stream);

/* Backward of the gather + score mean: scatter-add of d_sel into d_features (must be
 * zero-initialised by the caller) and of d_score/k into d_sc. */
int advhip_mil_topk_select_bwd_f32(const int64_t* idx, const float* d_sel, const float* d_score,
                                   float* d_features, float* d_sc, int32_t n, int32_t ncrops,
                                   int32_t T, int32_t F, int32_t k, void* stream);

/* Backward of advhip_mil_magnitude_f32 w.r.t. features and scores:
 *   d_features[r,t,:] += d_mag[b,t]/ncrops * features[r,t,:]/||features[r,t,:]||,
 *   d_scores[r,t]     += d_sc[b,t]/ncrops                      (r = b*ncrops + c). */
int advhip_mil_magnitude_bwd_f32(const float* features, const float* d_mag, const float* d_sc,
                                 float* d_features, float* d_scores, int32_t bs, int32_t ncrops,
                                 int32_t T, int32_t F, void* stream);

/* Fused loss reductions (src/loss/base.py:7-48, src/loss/mgfn.py:7-47, modeling_mgfn.py:406-418).
 * Inputs: scores (bs,T) video-level scores; abn/nor_score (n) top-k mean scores (n = bs/2);
 *         a_feat/n_feat (ncrops*n, k, F) selected features; labels (n) each.
 * out[0..7] = {total, bce, con, con_a, con_n, smooth, sparse, mgfn}.
 * `ws` is caller scratch of advhip_mgfn_loss_ws_floats(...) floats (holds the L1 norms). */
int64_t advhip_mgfn_loss_ws_floats(int32_t n, int32_t ncrops, int32_t k);
int advhip_mgfn_loss_fwd_f32(const float* scores, const float* abn_score, const float* nor_score,
                             const float* a_feat, const float* n_feat, const float* abn_labels,
                             const float* nor_labels, float* ws, float* out, int32_t bs, int32_t T,
                             int32_t ncrops, int32_t k, int32_t F, void* stream);

/* Gradients of out[0] (times *d_loss) w.r.t. scores, abn/nor_score, a_feat, n_feat. */
int advhip_mgfn_loss_bwd_f32(const float* d_loss, const float* scores, const float* abn_score,
                             const float* nor_score, const float* a_feat, const float* n_feat,
                             const float* abn_labels, const float* nor_labels, const float* ws,
                             float* d_scores, float* d_abn_score, float* d_nor_score,
                             float* d_a_feat, float* d_n_feat, int32_t bs, int32_t T,
                             int32_t ncrops, int32_t k, int32_t F, void* stream);

/* --- feature post-processing (on-device analogues of host numpy code) -------------------------
 * segment(): (n_clips, ncrops, C) -> (ncrops, seg, C) linspace-bucket means
 * (extract_features.py:171-183). */
int advhip_segment_features_f32(const float* feats, float* out, int32_t n_clips, int32_t ncrops,
                                int32_t C, int32_t seg, void* stream);

/* FeatureDataset.add_magnitude (src/dataset.py:121-124): (rows, C) -> (rows, C+1), last column
 * = L2 norm of the row. */
int advhip_add_magnitude_f32(const float* feats, float* out, int64_t rows, int32_t C, void* stream);

/* Clip pre-processing on the device: uint8 frames (N, T, C, H, W) -> fp32 (N, C, T, H, W) with
 * y = (x - mean) / std.  Replaces PILToTensor().float() + GroupNormalize(114.75, 57.375)
 * (src/dataset.py:175-183) and the permute of extract_features.py:83, so that only uint8 pixels
 * cross PCIe.  H*W must be a multiple of 4. */
int advhip_normalize_permute_u8(const uint8_t* x, float* y, int64_t N, int32_t T, int32_t C,
                                int32_t H, int32_t W, float mean, float stdv, void* stream);

/* The whole clip pre-processing after the resize, on the device: TenCrop (four corners + centre, then the same five of
 * the horizontally flipped frame: src/gtransforms.py:20-26 -> torchvision.transforms.TenCrop), PILToTensor().float(),
 * (x - mean) / std (src/gtransforms.py:57-73), LoopPad(frames_per_clip) (src/gtransforms.py:115-132) and the permutes
 * of src/dataset.py:195 + extract_features.py:83.
 *   frames: uint8 (F, H, W, C), the resized frames as decoded (HWC);  y: fp32 (ceil(F / fpc) * 10, C, fpc, crop, crop),
 *   row = clip * 10 + crop index.  Only the resized uint8 frames cross PCIe: 1/23 of the fp32 ten-crop bytes. */
int advhip_tencrop_normalize_u8(const uint8_t* frames, float* y, int32_t F, int32_t H, int32_t W, int32_t C,
                                int32_t frames_per_clip, int32_t crop, float mean, float stdv, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADVHIP_H */
