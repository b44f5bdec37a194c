/*
 * sm3det_hip.h -- C ABI of libsm3det_hip.so (MI355X / gfx950 hand-written HIP kernels).
 *
 * This is the drop-in boundary for the reference's native-operator surface (`mmcv._ext`):
 * every entry point below replaces one pybind `m.def` of the reference and takes ONLY plain device
 * pointers, sizes and a stream -- no torch types.  The Python host layer (sm3det_amd/mmcv_ext.py) owns the
 * tensors, checks dtype/contiguity/device and maps the int return code to RuntimeError, the way the
 * reference's pybind layer raises from TORCH_CHECK.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); kernels are enqueued on it and
 *     the call returns without synchronising;
 *   - return 0 on success, <0 on error (SM3_ERR_*); no exceptions, no hidden allocations: scratch memory is
 *     passed in as (workspace, workspace_bytes) and sized by the matching *_workspace_bytes() query;
 *   - float32 I/O; indices int64 (reference returns at::kLong).
 *
 * Reference paths are relative to /root/reference/mmcv/mmcv/ops/csrc/.
 */
#ifndef SM3DET_HIP_H
#define SM3DET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sm3_stream_t;

#define SM3_OK 0
#define SM3_ERR_INVALID_ARG (-1)
#define SM3_ERR_WORKSPACE (-2)
#define SM3_ERR_LAUNCH (-3)
#define SM3_ERR_UNSUPPORTED (-4)

/* library / device info ------------------------------------------------------------------------------- */
const char* sm3_version(void);              /* "sm3det_hip <semver> gfx950" */
const char* sm3_error_string(int code);
/* replaces get_compiler_version / get_compiling_cuda_version (pytorch/pybind.cpp:7-9,486-487) */
const char* sm3_compiler_version(void);

/* ---------------------------------------------------------------------------------------------------------
 * box_iou_rotated  -- replaces `box_iou_rotated(boxes1, boxes2, ious, mode_flag, aligned)`
 *   pybind: pytorch/pybind.cpp:308-309,746-748; CPU semantics followed: pytorch/cpu/box_iou_rotated.cpp:8-30,
 *   common/box_iou_rotated_utils.hpp:344-378.
 *   boxes1 (n1,5), boxes2 (n2,5) contiguous (cx,cy,w,h,theta_rad); ious (n1*n2) or (n1) when aligned
 *   (then n2 must equal n1). mode_flag 0 = IoU, 1 = IoF. */
int sm3_box_iou_rotated(const float* boxes1, const float* boxes2, float* ious, int n1, int n2,
                        int mode_flag, int aligned, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * argsort helper used by both NMS entry points when the caller passes order == NULL:
 * stable descending argsort of float scores (ties -> lower original index first). */
size_t sm3_argsort_desc_workspace_bytes(int n);
int sm3_argsort_desc_f32(const float* scores, int n, int64_t* order, void* workspace,
                         size_t workspace_bytes, sm3_stream_t stream);

/* The k (<= 2048) best scores in descending order: order[0..min(k,n)) = exactly the first entries sm3_argsort_desc_f32
 * would produce (ties by lower index), without sorting the rest -- `scores.topk(nms_pre)` of the proposal stage
 * (mmrotate/models/dense_heads/oriented_rpn_head.py:239-244).  k > 2048: SM3_ERR_UNSUPPORTED (use the argsort). */
size_t sm3_topk_desc_workspace_bytes(int n);
int sm3_topk_desc_f32(const float* scores, int n, int k, int64_t* order, void* workspace, size_t workspace_bytes,
                      sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * nms  -- replaces `nms(boxes, scores, iou_threshold, offset) -> Tensor[int64]`
 *   pybind: pytorch/pybind.cpp:185,634-635; CPU semantics followed: pytorch/cpu/nms.cpp:5-54
 *   (suppress when inter/(a_i+a_j-inter) > thr, division form; offset in {0,1}).
 *   boxes (n,4) x1,y1,x2,y2; scores (n). `order` = descending-score permutation (int64, n) or NULL to have
 *   the library sort. Outputs: keep (n int64, first *num_keep valid, original indices in score order),
 *   num_keep (1 int32, device). */
size_t sm3_nms_workspace_bytes(int n);
int sm3_nms(const float* boxes, const float* scores, const int64_t* order, int n, float iou_threshold,
            int offset, int64_t* keep, int32_t* num_keep, void* workspace, size_t workspace_bytes,
            sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * nms_rotated  -- replaces `nms_rotated(dets, scores, order, dets_sorted, iou_threshold, multi_label)`
 *   pybind: pytorch/pybind.cpp:311-313,749-751; CPU semantics followed: pytorch/cpu/nms_rotated.cpp:7-57
 *   (suppress when IoU >= thr -- line 51; the label column and `multi_label` are IGNORED by the CPU path,
 *   pytorch/nms_rotated.cpp:31).  dets (n, dets_stride>=5) rows (cx,cy,w,h,theta[,label]).
 *   multi_label != 0 additionally skips pairs with different labels (the reference's CUDA behaviour,
 *   common/cuda/nms_rotated_cuda.cuh:61-74 with `>`->`>=` kept CPU-style); the host mirror passes 0. */
size_t sm3_nms_rotated_workspace_bytes(int n);
int sm3_nms_rotated(const float* dets, int dets_stride, const float* scores, const int64_t* order, int n,
                    float iou_threshold, int multi_label, int64_t* keep, int32_t* num_keep,
                    void* workspace, size_t workspace_bytes, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * roi_align_rotated_{forward,backward}
 *   pybind: pytorch/pybind.cpp:323-326,761-770; CPU semantics followed:
 *   pytorch/cpu/roi_align_rotated.cpp:115-212 (fwd), :272-372 (bwd).
 *   input/grad_input (batch,channels,height,width); layout 0 = NCHW contiguous, 1 = NHWC (torch
 *   channels_last) memory.  rois (n_rois,6) [batch_idx,cx,cy,w,h,theta]; output/grad_output
 *   (n_rois,channels,pooled_h,pooled_w) NCHW contiguous.  sm3_roi_align_rotated_backward is the SCATTER form: grad_input
 *   must arrive zero-filled (mmcv/ops/roi_align_rotated.py:92) and is accumulated with fp32 atomics.  It is the fallback
 *   only (adaptive sampling grids, channels % 4 != 0, NCHW maps too small to pay for an NHWC pass): since round 4 the
 *   default path of the `mmcv._ext` mirror is the sorted GATHER form sm3_roi_align_rotated_backward_tiled below (no atomic
 *   accumulation; optional overwrite mode that needs no zero fill). */
int sm3_roi_align_rotated_forward(const float* input, const float* rois, float* output, int n_rois,
                                  int batch, int channels, int height, int width, int pooled_h,
                                  int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                  int clockwise, int layout, sm3_stream_t stream);
int sm3_roi_align_rotated_backward(const float* grad_output, const float* rois, float* grad_input,
                                   int n_rois, int batch, int channels, int height, int width,
                                   int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                   int aligned, int clockwise, int layout, sm3_stream_t stream);

/* Multi-level RoIAlignRotated = RotatedSingleRoIExtractor.forward (mmrotate/models/roi_heads/roi_extractors/
 * rotate_single_level_roi_extractor.py:103-140) in ONE launch: RoI n reads level
 * clamp(floor(log2(sqrt(w*h) / finest_scale + 1e-6)), 0, num_levels-1) (map_roi_levels :66-84) with that level's
 * spatial scale; no nonzero()/gather/scatter per level, no host sync.  inputs / grad_inputs / heights / widths /
 * scales are HOST arrays of num_levels (<= 8) entries (device pointers, ints, floats); all levels share
 * `channels` and the layout flag.  levels_out (n_rois int32, may be NULL) receives the chosen levels.
 * The multilevel backward declared here is the scatter form (fp32 atomics into zero-filled grad_inputs[l]); the fused
 * extractor's default backward is sm3_roi_align_rotated_backward_tiled (sorted gather, below). */
int sm3_roi_align_rotated_multilevel_forward(const float* const* inputs, const int* heights, const int* widths,
                                             const float* scales, int num_levels, float finest_scale,
                                             const float* rois, float* output, int32_t* levels_out, int n_rois,
                                             int channels, int pooled_h, int pooled_w, int sampling_ratio,
                                             int aligned, int clockwise, int layout, sm3_stream_t stream);
int sm3_roi_align_rotated_multilevel_backward(const float* grad_output, const float* rois,
                                              float* const* grad_inputs, const int* heights, const int* widths,
                                              const float* scales, int num_levels, float finest_scale, int n_rois,
                                              int channels, int pooled_h, int pooled_w, int sampling_ratio,
                                              int aligned, int clockwise, int layout, sm3_stream_t stream);

/* RoIAlignRotated backward as a GATHER (round 4; NHWC maps, sampling_ratio > 0, channels % 4 == 0): a counting sort files
 * every corner contribution under the pixel it lands on, and one wave per 16 pixels of an 8 x 8-pixel tile sums a pixel's
 * contributions in registers and adds them to grad_inputs[l] once -- no atomic accumulation.  Same semantics as
 * sm3_roi_align_rotated{,_multilevel}_backward (`grad_input +=`, roi_align_rotated_cuda_kernel.cuh:129-200 /
 * cpu/roi_align_rotated.cpp:272-372) with overwrite == 0; overwrite != 0: the maps need NOT be zero-filled, every pixel is
 * written (the result of the reference's `new_zeros` + kernel, mmcv/ops/roi_align_rotated.py:88-103, without the fill
 * pass).  num_levels == 1: a single map, the level rule is not evaluated.  `batch` = images per map.  workspace: pixel
 * counters / offsets, the entries, the transposed grad_output. */
size_t sm3_roi_align_rotated_backward_tiled_workspace_bytes(int n_rois, int batch, int channels, int pooled_h, int pooled_w,
                                                            int sampling_ratio, const int* heights, const int* widths,
                                                            int num_levels);
int sm3_roi_align_rotated_backward_tiled(const float* grad_output, const float* rois, float* const* grad_inputs,
                                         const int* heights, const int* widths, const float* scales, int num_levels,
                                         float finest_scale, int n_rois, int batch, int channels, int pooled_h,
                                         int pooled_w, int sampling_ratio, int aligned, int clockwise, int overwrite,
                                         void* workspace, size_t workspace_bytes, sm3_stream_t stream);

/* =========================================================================================================
 * Backbone hot path (a): grid-level sparse-MoE ConvNeXt.  Reference: mmrotate/models/backbones/convnext_moe.py.
 * Activations are token-major (T, C) float32 = NHWC; T = B*H*W.
 * ========================================================================================================= */

/* ---------------------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM family (v_mfma_f32_32x32x2_f32, exact f32).  Covers FFN.forward (convnext_moe.py:397-405),
 * the expert loop (:244) as a grouped GEMM, the cosine-gate projection (:101), stem/downsample patch GEMMs
 * (:533-558,:783-791) and all their gradients.
 *   mode 0 NT: C[M,N] = A[M,K] . B[N,K]^T      mode 1 NN: C[M,N] = A[M,K] . B[K,N]
 *   mode 2 TN: C[M,N] = A[Kt,M]^T . B[Kt,N]    (split-K over the reduction rows, needs workspace)
 * Grouping (MoE): `group_offsets` = DEVICE int32[num_groups+1] prefix of rows in expert-major slot order; group g
 * uses B + g*stride_b, bias + g*stride_bias; TN writes C + g*M*N.  NULL => a single group.
 * Requirements: lda, ldb, N multiples of 4; K multiple of 32 for NT/NN; TN: M multiple of 4. */
#define SM3_GEMM_NT 0
#define SM3_GEMM_NN 1
#define SM3_GEMM_TN 2
#define SM3_EPI_NONE 0           /* C = acc */
#define SM3_EPI_BIAS 1           /* C = acc + bias[n] */
#define SM3_EPI_BIAS_GELU 2      /* h = acc + bias ; C = gelu_erf(h) ; aux_out = gelu_erf'(h) (FFN first linear) */
#define SM3_EPI_BIAS_SCALE_RES 3 /* aux_out = y = acc + bias ; C = aux_in + gamma[n]*rowscale[m/rows_per_scale]*y */
#define SM3_EPI_GELU_BWD 4       /* C = acc * aux_in (aux_in = saved gelu'); optional colsum_out[g][n] = column sums of C
                                    per group = the first linear's bias gradient (needs workspace)               */
#define SM3_EPI_BIAS_RELU 5      /* NT only: C = max(acc + bias, 0) (nn.Linear + ReLU: convfc_rbbox_head.py:176-177)  */
typedef struct sm3_gemm_desc {
  int32_t mode, epilogue;
  const float* A;
  const float* B;
  float* C;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  const int32_t* group_offsets;
  int32_t num_groups;
  int32_t splits; /* split-K slices: TN per group (0 = automatic); NT/NN 0 = automatic (few output tiles, long K), 1 = off */
  int64_t stride_b, stride_bias;
  const float* bias;
  const float* aux_in;
  float* aux_out;
  const float* gamma;
  const float* rowscale;
  int32_t rows_per_scale;
  int32_t ld_aux;
  float* colsum_out; /* NN + EPI_GELU_BWD: column sums of C per group (G, N); TN (fp32): column sums of A over each group's
                        reduction rows (G, M) = the bias gradient next to the weight gradient dY^T X; may be NULL */
  int32_t* counters; /* ticket counters of the in-kernel split-K fix-up: sm3_gemm_f32_counter_slots() int32, ZERO on entry,
                        left zero on exit; one array per stream that may run GEMMs concurrently.  NULL: no in-kernel
                        fix-up (TN slices are then reduced by a second pass, NT/NN never slice K) */
  int32_t tuning;    /* 0 in production.  Benchmarking override: bits 0-3 tile+1 (0 128x128, 1 128x96, 2 96x128,
                        3 128x192, 4 192x128, 5 64x128), bits 4-7 k-step (1 = 16, 2 = 32), bits 8-15 slices,
                        bit 16 TN: slices summed by the in-kernel fix-up instead of the second pass, bits 20-23: slices / 256 */
  int32_t compute;   /* 0: fp32 operands (v_mfma_f32_32x32x2_f32, exact).  1: operands rounded to fp16 on the fly, fp32
                        accumulation (v_mfma_f32_32x32x16_f16) -- the arithmetic autocast gives nn.Linear in the reference's
                        AMP configs (fp16 = dict(loss_scale='dynamic')).  2: fp32 tensors, fp32-equivalent arithmetic on the
                        bf16 matrix pipe -- each operand element is split exactly into three bf16 pieces (x = x0+x1+x2,
                        round-to-nearest each) in the loader and the six products a_i.b_j with i+j <= 2 are accumulated in
                        fp32 by v_mfma_f32_32x32x16_bf16 (dropped terms <= 2^-25 |a||b|; non-finite inputs give NaN);
                        K % 16 == 0, tiles 128x128 / 128x96 / 64x128 (TN: 96x128), k-step 16 */
  int32_t io;        /* compute == 1 only: which tensors are STORED as fp16 (the AMP data path; 0 = everything fp32 in
                        memory, rounded in the loader).  Bits: 1 A, 2 B, 4 C, 8 aux_in / aux_out.  Supported: NT {1 with
                        epilogue none / bias / bias+scale+residual, 1|4|8 with bias+GELU}; NN {1 with none, 4|8 with
                        GELU'}; TN {2, 1|2}.  Flagged pointers address _Float16 elements; leading dimensions stay in
                        elements; weights, biases, residual and colsum outputs are always fp32 */
  /* compute == 2: io bit 16 (A) / 32 (B) = the operand arrives as bf16x3 PLANES (sm3_split_planes_f32 below) instead of
     fp32: NT / NN only.  A: the planes of A[M][K], lda = rows per k-octet block of the planes tensor.  B: the planes of the
     k-contiguous form of B -- B[N][K] in NT, B^T (made with transpose = 1) in NN -- ldb = rows per octet block, stride_b =
     rows between groups.  Results are bit-identical to the fp32-operand launch. */
} sm3_gemm_desc;
/* bf16x3 operand planes of an fp32 matrix x[rows][cols] (leading dimension ld): planes[p][o][r][j] = piece p (of the exact
 * three-way bf16 split, round-to-nearest-even each) of element (r, 8 o + j) -- 3 planes x (K / 8) octet blocks x Rp rows x 8
 * bf16, i.e. one 16-byte granule per plane, k-octet and row.  transpose != 0: the planes of x^T (K = rows of x, plane rows =
 * columns of x).  The matrix occupies plane rows [row_off, row_off + R) so that several matrices of equal K (the experts of a
 * layer) share one planes tensor.  What it replaces: the per-tile operand split of the bf16x3 GEMM loader for operands that
 * many tiles / launches read (weights: once per optimizer step). */
int sm3_split_planes_f32(const float* x, int rows, int cols, long ld, void* planes, long Rp, long row_off, int transpose,
                         sm3_stream_t stream);
int sm3_gemm_f32_counter_slots(void);
size_t sm3_gemm_f32_workspace_bytes(const sm3_gemm_desc* desc);
int sm3_gemm_f32(const sm3_gemm_desc* desc, void* workspace, size_t workspace_bytes, sm3_stream_t stream);
/* out[g][n] = sum over rows of group g of x[r][n]  (bias gradients); out is overwritten */
int sm3_colsum_f32(const float* x, int ld, int m, int n, const int32_t* group_offsets, int num_groups, float* out,
                   sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * FPN: MultitaskFPN.forward, mmrotate/models/necks/Multitask_FPN.py:113-162, on NHWC tokens.
 * 3x3 convolution, padding 1, stride 1 or 2 (ConvModule(out, out, 3, padding=1) :75-83 and the stride-2 extra levels
 * :97-106; conv/norm/act cfg None -> nn.Conv2d + bias) as implicit GEMMs -- no im2col buffer.  x (B,H,W,Cin),
 * w (Cout,3,3,Cin) [= Conv2d.weight.permute(0,2,3,1)], y (B,Ho,Wo,Cout), Ho = (H-1)/stride + 1.
 * fwd needs Cin % 32 == 0, bwd_input Cout % 32 == 0, bwd_weight Cin % 128 == 0; channel counts multiples of 4. */
/* workspace of fwd (backward_input = 0) / bwd_input (= 1): split-K slabs when the level has few output tiles, else 0 */
size_t sm3_conv3x3_nhwc_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride, int backward_input);
/* relu != 0: y = max(conv + bias, 0) (the F.relu after rpn_conv, rotated_rpn_head.py:45-46), needs bias */
int sm3_conv3x3_nhwc_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                         int Cout, int stride, int relu, void* workspace, size_t workspace_bytes,
                         sm3_stream_t stream);
int sm3_conv3x3_nhwc_bwd_input(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Cout,
                               int stride, void* workspace, size_t workspace_bytes, sm3_stream_t stream);
size_t sm3_conv3x3_nhwc_bwd_weight_workspace_bytes(int B, int H, int W, int Cin, int Cout, int stride);
int sm3_conv3x3_nhwc_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                int stride, void* workspace, size_t workspace_bytes, sm3_stream_t stream);
/* top-down merge (:123-135): out = fine + nearest_upsample_2x(coarse); fine/out (B,H,W,C), coarse (B,H/2,W/2,C).
 * gradient: dcoarse = base (may be NULL) + 2x2 sum-pool of dfine; dfine (B,2Hc,2Wc,C). */
int sm3_upsample2x_add(const float* fine, const float* coarse, float* out, int B, int H, int W, int C,
                       sm3_stream_t stream);
int sm3_sumpool2x_add(const float* dfine, const float* base, float* dcoarse, int B, int Hc, int Wc, int C,
                      sm3_stream_t stream);

/* dst[b][c][r] = src[b][r][c] (batched 2-D transpose; NHWC <-> NCHW of a feature map with rows = H*W, cols = C).  Used
 * where the mmcv._ext boundary hands NCHW tensors to a kernel whose scatter pattern wants NHWC (RoIAlignRotated
 * backward: 4.1 ms on NCHW vs 0.41 ms on NHWC for 512 RoIs on a 256x256x256 level). */
int sm3_transpose_f32(const float* src, float* dst, int batch, int rows, int cols, sm3_stream_t stream);
/* dst[b][c][r] += src[b][r][c]: the accumulate form (the reference kernel atomically ADDS into grad_input,
 * roi_align_rotated_cuda_kernel.cuh:129-200, so a caller-provided non-zero grad_input must survive). */
int sm3_transpose_add_f32(const float* src, float* dst, int batch, int rows, int cols, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * MaxIoU target assignment: mmdet MaxIoUAssigner.assign_wrt_overlaps as configured by
 * local_configs/main_SM3Det.py:165-196 (rpn: BboxOverlaps2D on (x1,y1,x2,y2) anchors; rcnn: RBboxOverlaps2D =
 * rbbox_overlaps, mmrotate/core/bbox/iou_calculators/rotate_iou2d_calculator.py:52-87), called from
 * oriented_rpn_head.py:76-78 / oriented_standard_roi_head.py:68-70.  No (k x n) overlap matrix is materialised.
 * gt_inds[j] = -1 ignore / 0 background / i+1 matched gt; max_overlaps[j]; labels[j] = gt_labels[i] or -1 (labels and
 * gt_labels may be NULL).  gt_max_assign_all semantics (every box tying a gt's best IoU >= min_pos_iou is assigned to
 * it; later gts override earlier ones).  rotated: boxes (cx,cy,w,h,a) else (x1,y1,x2,y2); strides in floats. */
size_t sm3_max_iou_assign_workspace_bytes(int n, int k);
int sm3_max_iou_assign(const float* boxes, int box_stride, int n, const float* gts, int gt_stride, int k, int rotated,
                       float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                       const int64_t* gt_labels, int64_t* gt_inds, float* max_overlaps, int64_t* labels,
                       void* workspace, size_t workspace_bytes, sm3_stream_t stream);
/* The same with a per-box validity mask (uint8, NULL = all valid): boxes with box_flags[j] == 0 take no part in the
 * assignment -- gt_inds -1, max_overlaps 0, and they do not count towards a gt's best IoU.  This is the reference's
 * `anchors = flat_anchors[inside_flags]` followed by `unmap` (oriented_rpn_head.py:67-76, :124-131, mmdet
 * anchor_inside_flags with train_cfg.allowed_border) without the compaction: no data-dependent shape, no host sync. */
int sm3_max_iou_assign_masked(const float* boxes, int box_stride, int n, const uint8_t* box_flags, const float* gts,
                              int gt_stride, int k, int rotated, float pos_iou_thr, float neg_iou_thr,
                              float min_pos_iou, int match_low_quality, const int64_t* gt_labels, int64_t* gt_inds,
                              float* max_overlaps, int64_t* labels, void* workspace, size_t workspace_bytes,
                              sm3_stream_t stream);

/* obb2xyxy(., 'le90') (mmrotate/core/bbox/transforms.py:685-702): (n, stride >= 5) cx,cy,w,h,a -> (n,4) x1,y1,x2,y2. */
int sm3_obb2xyxy_le90(const float* obb, int n, int stride, float* out, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Targets + losses of the two-stage branch on FIXED-SIZE sample blocks, sync-free.
 *
 * RPN: OrientedRPNHead._get_targets_single + RotatedRPNHead.get_targets / loss + loss_single
 * (mmrotate/models/dense_heads/oriented_rpn_head.py:26-187, rotated_rpn_head.py:152-372) for use_sigmoid_cls,
 * reg_decoded_bbox = False: only the sampled anchors carry a weight, so only they are visited.  Per image b the sampler
 * hands `samples` slots: idx (flat anchor index over the concatenated levels, position-major then anchor), is_pos, valid;
 * gt_inds (batch, total_anchors) is the assigner's output (i + 1 for a positive), gts (batch, max_gts, 5) the oriented
 * ground truth (cx,cy,w,h,a), n_pos / n_neg (batch,) the sampled counts (avg_factor = sum_b max(n_pos,1) + max(n_neg,1)).
 * A level's prediction element (image b, position p, channel ch) lives at ptr[b*stride[0] + p*stride[1] + ch*stride[2]]
 * (floats; cls channel = anchor a, reg channel = 6 a + c): NHWC or NCHW maps, or views of a fused head output.
 * forward: loss_cls[l], loss_bbox[l] per level (mmdet CrossEntropyLoss(use_sigmoid=True) / SmoothL1Loss(beta), each
 * loss_weight * sum / (avg_factor + eps)).  backward: writes d loss / d prediction, times dloss_*[l], at the sampled
 * anchors of the dcls / dreg maps, which the caller zero-filled. */
#define SM3_RPN_MAX_LEVELS 8
typedef struct {
  const float* cls;
  const float* reg;
  long cls_stride[3], reg_stride[3];
  float* dcls; /* backward only */
  float* dreg;
  long dcls_stride[3], dreg_stride[3];
  long num_anchors; /* H * W * anchors_per_pos of this level */
} sm3_rpn_loss_level;
typedef struct {
  sm3_rpn_loss_level level[SM3_RPN_MAX_LEVELS];
  int num_levels, anchors_per_pos;
  const float* anchors;   /* (total_anchors, 4) x1,y1,x2,y2 */
  const int64_t* idx;     /* (batch, samples) */
  const uint8_t* is_pos;  /* (batch, samples) */
  const uint8_t* valid;   /* (batch, samples) */
  const int64_t* gt_inds; /* (batch, total_anchors) */
  const float* gts;       /* (batch, max_gts, 5) */
  int batch, samples, total_anchors, max_gts;
  const int64_t* n_pos;   /* (batch,) */
  const int64_t* n_neg;
  float means[6], stds[6]; /* MidpointOffsetCoder target_means / target_stds */
  float beta, loss_weight_cls, loss_weight_bbox, pos_weight; /* pos_weight <= 0: 1 (train_cfg.pos_weight = -1) */
} sm3_rpn_loss_desc;
int sm3_rpn_loss_forward(const sm3_rpn_loss_desc* d, float* loss_cls, float* loss_bbox, sm3_stream_t stream);
int sm3_rpn_loss_backward(const sm3_rpn_loss_desc* d, const float* dloss_cls, const float* dloss_bbox,
                          sm3_stream_t stream);

/* RCNN: RotatedBBoxHead._get_target_single / get_targets / loss (mmrotate/models/roi_heads/bbox_heads/
 * rotated_bbox_head.py:141-356) for reg_class_agnostic, reg_decoded_bbox = False, DeltaXYWHAOBBoxCoder 'le90'.  Row i is
 * one sampled RoI: valid[i]; labels[i] in [0, num_classes) = positive matched to gts[i], num_classes = background.
 * forward: out3 = [loss_cls (softmax CE, avg_factor = max(#rows with weight > 0, 1)), loss_bbox (SmoothL1 over the
 * positives / #valid rows), acc (top-1, per cent)]; counts2 = the two avg factors (kept for backward).
 * backward: d cls_score (num_rois, ld_cls) and d bbox_pred (num_rois, ld_reg), every row written.
 * With label_weights / bbox_targets given, the same kernels evaluate the reference's loss(cls_score, bbox_pred, rois,
 * labels, label_weights, bbox_targets, bbox_weights) on targets a caller built with get_targets(). */
typedef struct {
  const float* cls_score;
  const float* bbox_pred;
  float* dcls_score; /* backward only */
  float* dbbox_pred;
  int ld_cls, ld_reg, num_classes, num_rois;
  const int64_t* labels;
  const uint8_t* valid;
  const float* rois; /* (num_rois, 5) cx,cy,w,h,a */
  const float* gts;  /* (num_rois, 5) */
  const float* label_weights; /* optional (num_rois): the reference API's precomputed weights (valid may then be NULL) */
  const float* bbox_targets;  /* optional (num_rois, 5): precomputed targets (rois / gts may then be NULL) */
  float means[5], stds[5];
  float norm_factor; /* 0 = None */
  int edge_swap, proj_xy;
  float beta, loss_weight_cls, loss_weight_bbox, pos_weight;
} sm3_rcnn_loss_desc;
int sm3_rcnn_loss_forward(const sm3_rcnn_loss_desc* d, float* out3, float* counts2, sm3_stream_t stream);
int sm3_rcnn_loss_backward(const sm3_rcnn_loss_desc* d, const float* counts2, const float* dloss_cls,
                           const float* dloss_bbox, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * GroupNorm (+ ReLU) on NHWC tokens: the normalisation of the GFL head's conv towers (local_configs/main_SM3Det.py:29-48
 * `type='GFLHead', stacked_convs=4`: ConvModule(conv3x3, GN(32), ReLU); the class is mmdet 2.x code the reference does
 * not vendor -- torch.nn.GroupNorm semantics).  x, y, dy, dx: (B, P, C) with P = H*W; stats (B, G, 2) = (mean, rstd);
 * C <= 1024, (C/G) % 4 == 0.  bwd leaves (B * sm3_groupnorm_blocks(P,C)) x 2C partial rows [d gamma | d beta] for
 * sm3_row_partials_reduce.  relu != 0: y = max(0, gn(x)), the backward masks dy with y > 0. */
int sm3_groupnorm_blocks(long P, int C);
size_t sm3_groupnorm_workspace_bytes(int B, long P, int C, int G);
int sm3_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int relu, float* y,
                      float* stats, int B, long P, int C, int G, void* workspace, size_t workspace_bytes,
                      sm3_stream_t stream);
int sm3_groupnorm_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* stats,
                      int relu, float* dx, float* dgamma_dbeta_part, int B, long P, int C, int G, void* workspace,
                      size_t workspace_bytes, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Oriented-RPN proposal glue.
 * relu_bwd: dx = y > 0 ? dy : 0 (n multiple of 4).  sigmoid: y = 1/(1+exp(-x)) (oriented_rpn_head.py:236-238).
 * rpn_decode_le90: for i < n, src = order ? order[i] : i (order = descending-score permutation: the top-k gather of
 * :248-254); proposals[i] (cx,cy,w,h,a) = MidpointOffsetCoder.decode(anchors[src] (x1,y1,x2,y2), deltas[src] (6))
 * (delta_midpointoffset_rbbox_coder.py:150-238, angle version 'le90'), hboxes[i] = obb2xyxy_le90(proposals[i])
 * (transforms.py:685-702), scores_out[i] = scores[src] (both may be NULL).  means6 / stds6 are HOST arrays. */
int sm3_relu_bwd(const float* dy, const float* y, float* dx, long n, sm3_stream_t stream);
int sm3_sigmoid_f32(const float* x, float* y, long n, sm3_stream_t stream);
int sm3_rpn_decode_le90(const float* anchors, const float* deltas, const float* scores, const int64_t* order, int n,
                        const float* means6, const float* stds6, float wh_ratio_clip, float* proposals,
                        float* hboxes, float* scores_out, sm3_stream_t stream);

/* Box coders of the 2-stage branch (angle version 'le90'), one thread per box, reference operation order; means /
 * stds are HOST arrays.
 * midpoint_offset_encode: MidpointOffsetCoder.encode = bbox2delta (delta_midpointoffset_rbbox_coder.py:87-148):
 *   proposals (n,4) x1,y1,x2,y2 + gt (n,5) -> deltas (n,6).
 * delta_xywha_decode / _encode: DeltaXYWHAOBBoxCoder (delta_xywha_rbbox_coder.py:112-176, :180-283), class-agnostic
 *   (n,5) deltas, add_ctr_clamp = False; norm_factor 0 = None; max_h/max_w 0 = no max_shape clamp. */
int sm3_midpoint_offset_encode_le90(const float* proposals, const float* gt, int n, const float* means6,
                                    const float* stds6, float* deltas, sm3_stream_t stream);
int sm3_delta_xywha_decode_le90(const float* rois, const float* deltas, int n, const float* means5,
                                const float* stds5, float wh_ratio_clip, float norm_factor, int edge_swap,
                                int proj_xy, int max_h, int max_w, float* out, sm3_stream_t stream);
int sm3_delta_xywha_encode_le90(const float* proposals, const float* gt, int n, const float* means5,
                                const float* stds5, float norm_factor, int edge_swap, int proj_xy, float* deltas,
                                sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Stem: Conv2d(3,C0,k=4,s=4) (convnext_moe.py:783-791) as patchify + NT GEMM.  x (B,3,H,W) NCHW ->
 * a (B*H/4*W/4, 64), columns c*16+kh*4+kw (= weight.view(C0,48) order), columns 48..63 zero. */
int sm3_stem_patchify(const float* x, float* a, int B, int H, int W, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * LayerNorm over the channel dim of token rows (LayerNorm2d, convnext_moe.py:30-47; block norm :351).
 * out_mode 0: y[t] ; out_mode 1: patch-major rows feeding the 2x2/s2 downsample conv (:549-556) as one GEMM;
 * out_mode 2 (forward only): y[t] stored as fp16 (_Float16* passed as float*): the AMP data path's GEMM operand.
 * mean/rstd (T) are saved for the backward (may be NULL).  bwd: dwdb = [dw (C) | db (C)], overwritten;
 * accumulate_dx != 0 adds into dx. */
/* out_mode | SM3_LN_X_F16 (forward and backward): the rows of `x` are stored as fp16 (the depthwise output of the AMP data
 * path, what autocast makes of ConvNeXtBlock.depthwise_conv: convnext_moe.py:347); statistics and arithmetic stay fp32. */
#define SM3_LN_X_F16 16
int sm3_layernorm_fwd(const float* x, const float* w, const float* b, float eps, float* y, float* mean, float* rstd,
                      long T, int C, int out_mode, int H, int W, sm3_stream_t stream);
/* dst (n halves) = round-to-nearest-even fp16 of src (n floats, n % 4 == 0): the fp16 shadow of a weight tensor that the
 * fp16-operand GEMMs read as their B operand (io bit 2 on NT / NN) -- the half model `wrap_fp16_model` keeps next to the
 * fp32 master weights (mmcv/mmcv/runner/fp16_utils.py, hooks/optimizer.py:245-261 copy_params_to_fp16). */
int sm3_cast_f32_f16(const float* src, void* dst, long n, sm3_stream_t stream);
/* scratch for the column reductions of layernorm_bwd / scale_bwd_prep / moe_combine_bwd (per-block partials) */
size_t sm3_row_reduce_workspace_bytes(int C);
/* The three kernels reduce their partials themselves unless their reduction output (dwdb / dgamma_db / dgamma) is NULL;
 * then the caller finishes with sm3_row_partials_reduce(workspace, sm3_row_partial_blocks(T, C), ncols, out) -- e.g. on a
 * side stream, since the results are parameter gradients (ncols = 2C, 2C, C respectively). */
int sm3_row_partial_blocks(long T, int C);
int sm3_row_partials_reduce(const float* partials, int nblocks, int ncols, float* out, sm3_stream_t stream);
/* n such reductions in one launch (chunks of 64): partials / outs / nblocks / ncols are HOST arrays of n entries (device
 * pointers and sizes as for sm3_row_partials_reduce); the table travels in the kernel arguments, so the call is capturable.
 * Used to run the parameter-gradient reductions of a whole backward pass together when the pass ends. */
int sm3_row_partials_reduce_multi(const float* const* partials, float* const* outs, const int* nblocks, const int* ncols,
                                  int n, sm3_stream_t stream);
int sm3_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                      float* dx, float* dwdb, long T, int C, int out_mode, int H, int W, int accumulate_dx,
                      void* workspace, size_t workspace_bytes, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Depthwise 7x7, padding 3 (ConvNeXtBlock.depthwise_conv, convnext_moe.py:311-312,347) on NHWC tokens.
 * w49 (49,C) = weight (C,1,7,7) permuted to tap-major; y = conv(x) + bias (+ addend).  flip != 0 reads the taps
 * reversed (w49[48 - t]): the input gradient is this kernel with flip = 1 and addend = the residual branch gradient.
 * bwd_weight: dw49 (49,C) and dbias (C) overwritten (one fill when dbias == dw49 + 49*C, i.e. a (50,C) buffer). */
/* flip | SM3_DW_OUT_F16: y is stored as fp16 (AMP data path; the LDS-tiled kernels only -- C % 32 == 0, H and W multiples
 * of 16, one image below 2 GiB -- otherwise SM3_ERR_UNSUPPORTED). */
#define SM3_DW_OUT_F16 32
int sm3_dwconv7_fwd(const float* x, const float* w49, const float* bias, const float* addend, float* y, int B, int H,
                    int W, int C, int flip, sm3_stream_t stream);
int sm3_dwconv7_bwd_weight(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                           sm3_stream_t stream);
/* ... _acc: the same sums ADDED to dw49 / dbias (no fill): the caller provides zeros -- e.g. slices of ONE zero-filled
 * arena for all blocks of a backward pass (18 fills -> 1) -- or a running sum */
int sm3_dwconv7_bwd_weight_acc(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                               sm3_stream_t stream);

/* layer-scale / stochastic-depth backward of a dense block (:368-370): dy = gamma*rs[b]*dout ;
 * dgamma_db (2C, overwritten) = [ dgamma[c] = sum_t rs[b]*dout[t,c]*y[t,c] | db2[c] = sum_t dy[t,c] ] */
int sm3_scale_bwd_prep(const float* dout, const float* y, const float* gamma, const float* rowscale,
                       int rows_per_scale, float* dy, float* dgamma_db, long T, int C, void* workspace,
                       size_t workspace_bytes, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * MoE router (CosineTopKGate.forward :99-106 + noisy_top_k_gating :194-223 + _prob_in_top_k :152-174).
 * hcat (T, ldh) rows = [h = x.Wp^T+bp (P) | raw = x.Wn (E) | pad]; snorm (P,E) = F.normalize(sim_matrix, dim=0);
 * scale = device scalar exp(min(temperature, ln 100)); noise (T,E) ~ N(0,1) (train) or NULL.
 * Outputs: top_idx/top_val (T, m=min(k+1,E)) descending; gates (T,k) = softmax(top k); clean (T,E); sigma (T,E)
 * (train); hnorm (T); partials (sm3_moe_router_partial_rows(T), 2E) = per-workgroup [importance | load] sums
 * (sum over dim 0 = totals, deterministic).  forced_topk (T,k) int32 or NULL (production): teacher-forced routing for
 * precision tests -- the given experts win the top-k selection, all values still come from this run's logits. */
int sm3_moe_router_partial_rows(int T);
int sm3_moe_router_fwd(const float* hcat, int ldh, int P, const float* snorm, const float* scale, const float* noise,
                       int T, int E, int k, int train, int32_t* top_idx, float* top_val, float* gates, float* clean,
                       float* sigma, float* hnorm, float* partials, const int32_t* forced_topk, sm3_stream_t stream);
/* backward: dgate (T,k) from the combine, dimp/dload (E) from the aux loss.  Writes dhcat (T,ldh) = [dh | draw | 0],
 * dcn (T,E) = dclean/max(|h|,eps) and ds_part (sm3_moe_router_partial_rows(T), DOUBLE) partial sums of d(scale); ds_sq (same
 * size, may be NULL): per workgroup the sum of the SQUARED per-token terms of that sum (its conditioning, for tests). */
int sm3_moe_router_bwd(const float* hcat, int ldh, int P, const float* snorm, const float* scale, const float* noise,
                       int T, int E, int k, int train, const int32_t* top_idx, const float* top_val,
                       const float* gates, const float* clean, const float* sigma, const float* hnorm,
                       const float* dgate, const float* dimp, const float* dload, float* dhcat, float* dcn,
                       double* ds_part, double* ds_sq, sm3_stream_t stream);

/* Gate parameter preparation, one launch (CosineTopKGate.forward :96-105 + the `x @ w_noise` operand :199-201):
 * wcat (PC,C) = [cosine_projector.weight (P,C); w_noise^T (E,C); 0], bcat (PC) = [cosine_projector.bias; 0],
 * snorm (P,E) = F.normalize(sim_matrix, dim=0), scale (1) = exp(min(temperature, clamp_max)). */
int sm3_moe_gate_prep_fwd(const float* wp, const float* bp, const float* wn, const float* sim,
                          const float* temperature, float clamp_max, int P, int C, int E, int PC, float* wcat,
                          float* bcat, float* snorm, float* scale, sm3_stream_t stream);
/* backward: dwcat (PC,C) / dbcat (PC) from the gate GEMM, dsn (P,E) = h^T.dcn (gradient w.r.t. snorm*scale),
 * ds_part (n_part, double) partial sums of d(scale) from sm3_moe_router_bwd -> gradients of the five reference parameters. */
int sm3_moe_gate_prep_bwd(const float* dwcat, const float* dbcat, const float* dsn, const double* ds_part, int n_part,
                          const float* sim, const float* temperature, float clamp_max, int P, int C, int E,
                          float* dwp, float* dbp, float* dwn, float* dsim, float* dtemp, sm3_stream_t stream);
/* Auxiliary load-balancing loss (:140-147, :234-238): tot (2E) = column sums of the router partials =
 * [importance | load]; loss (1) = coef * (cv_squared(importance) + cv_squared(load)).  Backward: dimp/dload (E). */
int sm3_moe_aux_loss_fwd(const float* partials, int nblk, int E, float coef, float* tot, float* loss,
                         sm3_stream_t stream);
int sm3_moe_aux_loss_bwd(const float* tot, const float* dloss, int E, float coef, float* dimp, float* dload,
                         sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * SparseDispatcher (:250-293) without the host sync: expert-major slot tables from the top-k indices.
 * offsets (E+1) prefix of slots per expert; slot_token (T*k) token of each slot; token_slot (T,k) slot of each
 * (token, j).  Slot order inside an expert = (token, j) order (deterministic). */
size_t sm3_moe_plan_workspace_bytes(int T, int E);
int sm3_moe_plan(const int32_t* top_idx, int m, int T, int E, int k, int32_t* offsets, int32_t* slot_token,
                 int32_t* token_slot, void* workspace, size_t workspace_bytes, sm3_stream_t stream);
/* dispatch (:264-266): xslot[s] = x[slot_token[s]] */
int sm3_moe_dispatch(const float* x, const int32_t* slot_token, float* xslot, long S, int C, sm3_stream_t stream);
/* combine (:269-284) fused with layer scale + residual (:368-370):
 * out[t] = shortcut[t] + gamma*rs[b]*sum_j gates[t,j]*yslot[token_slot[t,j]] */
int sm3_moe_combine_fwd(const float* yslot, const int32_t* token_slot, const float* gates, const float* shortcut,
                        const float* gamma, const float* rowscale, int rows_per_scale, float* out, long T, int C,
                        int k, sm3_stream_t stream);
int sm3_moe_combine_bwd(const float* dout, const float* yslot, const int32_t* token_slot, const float* gates,
                        const float* gamma, const float* rowscale, int rows_per_scale, float* dyslot, float* dgate,
                        float* dgamma, long T, int C, int k, void* workspace, size_t workspace_bytes,
                        sm3_stream_t stream);
/* dx[t] (+)= sum_j dxslot[token_slot[t,j]]  (backward of dispatch) */
int sm3_moe_gather_add(const float* dxslot, const int32_t* token_slot, float* dx, long T, int C, int k,
                       int accumulate, sm3_stream_t stream);

/* DeformConv2d forward WITHOUT the column matrix (round 4): the bilinear sampling of deformable_im2col
 * (common/cuda/deform_conv_cuda_kernel.cuh:190-241) runs inside the A-operand producer of an MFMA GEMM whose column tile
 * lives in LDS only.  x_nhwc (B,H,W,Cin); offset (B, 2*kh*kw, Ho, Wo) as the reference passes it; w_t = the weight
 * (Cout,Cin,kh,kw) re-laid out as (kh*kw*Cin, Cout); out (B,Cout,Ho,Wo) -- replaces deform_conv_forward's im2col + addmm
 * loop (pytorch/deform_conv.cpp:140-258) for groups = deformable_groups = 1 and Cin % 16 == 0 (`_supported` says so). */
int sm3_deform_conv_fwd_fused_supported(int channels, int out_channels, int kh, int kw, int group, int deformable_group);
int sm3_deform_conv_fwd_fused(const float* x_nhwc, const float* offset, const float* w_t, float* out, int batch, int channels,
                              int height, int width, int out_channels, int kh, int kw, int pad_h, int pad_w, int stride_h,
                              int stride_w, int dil_h, int dil_w, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * DeformConv2d sampling kernels (the device half of deform_conv_{forward,backward_input,backward_parameters},
 * pybind.cpp:38-57,501-522; semantics of pytorch/cpu/deform_conv.cpp:114-290).  Reference layouts: im
 * (imgs,C,H,W), offset (imgs, dg*2*kh*kw, Ho, Wo), col (C*kh*kw, ld_col) with ld_col >= imgs*Ho*Wo.  The
 * per-group GEMMs around them use sm3_gemm_f32 (host: sm3det_amd/deform_conv_host.py). col2im accumulates into
 * grad_im with fp32 atomics (grad_im zero-filled by the caller, mmcv/ops/deform_conv.py:127). */
int sm3_deform_im2col(const float* im, const float* offset, float* col, int channels, int height, int width, int kh,
                      int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int imgs,
                      int deformable_group, long ld_col, sm3_stream_t stream);
int sm3_deform_col2im(const float* col, const float* offset, float* grad_im, int channels, int height, int width,
                      int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int imgs,
                      int deformable_group, long ld_col, sm3_stream_t stream);
/* The input-gradient scatter onto an NHWC map (imgs, H, W, C) that the caller zero-filled (a wave = one sampling
 * position x 64 channels: coalesced 256-byte atomics instead of 64 scattered ones); the caller then adds the map into
 * the NCHW gradInput with sm3_transpose_add_f32. */
int sm3_deform_col2im_nhwc(const float* col, const float* offset, float* grad_im_nhwc, int channels, int height,
                           int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                           int dil_w, int imgs, int deformable_group, long ld_col, sm3_stream_t stream);
/* deformable_col2im + deformable_col2im_coord in ONE pass over the columns, NHWC on both sides: im_nhwc (imgs,H,W,C) is the
 * input transposed by the caller, grad_im_nhwc (imgs,H,W,C) zero-filled by the caller receives the input gradient
 * (coalesced channel-vector atomics), grad_offset (imgs, dg*2*kh*kw, Ho, Wo) is fully written. */
int sm3_deform_bwd_input_fused(const float* col, const float* im_nhwc, const float* offset, float* grad_im_nhwc,
                               float* grad_offset, int channels, int height, int width, int kh, int kw, int pad_h,
                               int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int imgs,
                               int deformable_group, long ld_col, sm3_stream_t stream);
int sm3_deform_col2im_coord(const float* col, const float* im, const float* offset, float* grad_offset, int channels,
                            int height, int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                            int dil_h, int dil_w, int imgs, int deformable_group, long ld_col, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * RandomSampler / RRandomSampler as a fixed-size selection (mmdet `RandomSampler._sample_pos /
 * _sample_neg` as `train_cfg.{rpn,rcnn}.sampler` of local_configs/main_SM3Det.py:98-107,128-133 configure it,
 * mmrotate/core/bbox/samplers/rotate_random_sampler.py:10).  gt_inds (n) int64 from the assigner (> 0 positive, 0 negative,
 * < 0 ignored), key (n) uniform random floats in [0, 1).  Slots [0, n_pos) receive the min(exp_pos, #positives) positives
 * with the smallest keys in key order, slots [n_pos, n_pos + n_neg) the negatives with the smallest keys, n_neg =
 * min(#negatives, num - n_pos[, neg_pos_ub * max(n_pos, 1) when neg_pos_ub >= 0]); the other slots get valid = 0.
 * num <= 2048.  workspace: sm3_random_sample_workspace_bytes() bytes, the first 16 ZERO on entry (left zero on exit). */
/* The sampled rows of one image for the RCNN stage: RoI (batch index | box; gts count as proposals when `prepended`),
 * class label (num_classes = background for negatives) and matched gt of every slot
 * (oriented_standard_roi_head.py:66-92 `rbbox2roi` of the sampling results, rotated_bbox_head.py:131-204 labels / gts per
 * sample), written at slot `out0` of batch-wide blocks; unused slots get a unit box, valid_out = 0. */
int sm3_rcnn_gather_samples(const float* gts, const int64_t* gt_labels, int k, int prepended, const float* props,
                            int ld_props, const int64_t* gt_inds_all, const int64_t* labels, const int64_t* idx,
                            const uint8_t* is_pos, const uint8_t* valid, int S, int num_classes, float batch_index, long out0,
                            float* rois_out, int64_t* labels_out, float* gts_out, uint8_t* valid_out, sm3_stream_t stream);
size_t sm3_random_sample_workspace_bytes(void);
int sm3_random_sample_fixed(const int64_t* gt_inds, const float* key, int n, int num, int exp_pos, float neg_pos_ub,
                            int64_t* idx_out, uint8_t* is_pos_out, uint8_t* valid_out, int64_t* n_pos_out,
                            int64_t* n_neg_out, void* workspace, size_t workspace_bytes, sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Optimizer step: clip_grad_norm_(max_norm) + AdamW over all tensors, one param group per
 * tensor (mmcv/mmcv/runner/hooks/optimizer.py:55-73, optimizer/default_constructor.py:180-227; per-group lr written by
 * mmrotate/core/hook/dynamic_lr.py:197-218).  Device tables: *_ptrs[t] = float* of tensor t (param, grad, exp_avg,
 * exp_avg_sq), numel[t]; chunk_tab[c] = (tensor id, chunk index) with sm3_optim_chunk_elems() elements per chunk;
 * lr[t], wd[t] device vectors.  step / clip_coef / grad_norm are device scalars (step is incremented here;
 * max_grad_norm <= 0 disables clipping; partials = n_chunks floats of scratch).  torch.optim.AdamW arithmetic.
 * scaler (may be NULL) = device [loss scale, growth tracker, found_inf]: torch.cuda.amp.GradScaler semantics of the
 * reference's Fp16OptimizerHook (mmcv/mmcv/runner/hooks/optimizer.py:283-300) without a host sync -- gradients are
 * unscaled inside the update, a non-finite gradient norm skips the step (nothing is written, `step` does not advance)
 * and multiplies the scale by backoff_factor, growth_interval clean steps multiply it by growth_factor;
 * growth_interval <= 0 keeps the scale static.
 * h_ptrs (may be NULL; entries may be 0): _Float16* of a tensor's fp16 SHADOW -- the operand copy the AMP GEMMs read
 * (what the half model of wrap_fp16_model holds, mmcv/mmcv/runner/fp16_utils.py:71-149): the update writes the rounded
 * new value next to the fp32 master, so no cast pass runs in the forward. */
int sm3_optim_chunk_elems(void);
int sm3_adamw_multi(const uint64_t* p_ptrs, const uint64_t* g_ptrs, const uint64_t* m_ptrs, const uint64_t* v_ptrs,
                    const uint64_t* h_ptrs, const int64_t* numel, const int32_t* chunk_tab, int n_chunks, const float* lr, const float* wd,
                    float beta1, float beta2, float eps, float max_grad_norm, float* step, float* clip_coef,
                    float* grad_norm, float* partials, float* scaler, float growth_factor, float backoff_factor,
                    int growth_interval, sm3_stream_t stream);

/* Arithmetic of the implicit-GEMM 3x3 convolutions below (sm3_conv3x3_nhwc_*): 0 = native fp32 MFMA (default of the
 * library), 2 = the bf16x3 form of sm3_gemm_desc.compute == 2 (fp32-equivalent, k-step 16; the weight gradient uses it when
 * the output width is a multiple of 16, otherwise the native form).  Process-wide; set it before querying workspace sizes. */
int sm3_conv3x3_set_arith(int compute);

/* DynamicLrUpdaterHook (mmrotate/core/hook/dynamic_lr.py:107-217; local_configs/main_SM3Det.py:291-300 `policy='dynamic'`)
 * on the device: losses[n] = this iteration's loss scalars (the keys of `reweight_losses` present in the step, in log_vars
 * order), loss_subnet[n] = index of the sub-network each belongs to, param_subnet[n_params] = sub-network of every optimizer
 * tensor or -1 for shared (backbone / neck) tensors, base_lr[n_params] = the groups' initial lr, sched[1] = the step-decay
 * factor gamma^exp of get_lr.  state = n + 2 doubles, zero before the first call: loss EMAs, number of EMA updates,
 * iteration index (all advanced here).  Writes lr[n_params] -- the vector sm3_adamw_multi reads -- without any host read:
 * warm-up iterations (warmup_iters > 0, iteration < warmup_iters) get base_lr * (1 - (1 - it / warmup_iters) * (1 -
 * warmup_ratio)) while the EMAs update -- warmup_ratio = 1 reproduces what a reference run does (the hook never installs
 * `regular_lr` under IterBasedRunner, so the lr stays the initial lr during the warm-up: sm3det_amd/optim.py), a ratio < 1 is
 * mmcv's documented linear ramp; later iterations get base_lr * sched * (sub-network weight | backbone-policy weight).
 * warmup_iters < 0: no warm-up, but the hook's head-weight gate `EMA updates < warmup_iters` (:124) reads |warmup_iters|.
 * head_policy 0 normal / 1 reverse / 2 'None'; backbone_policy 0 min / 1 avg / 2 max / 3 kl / 4 sigmoid_kl / 5 none.
 * n <= 32, n_subnets <= 63. */
int sm3_dla_lr(const float* losses, int n, const int32_t* loss_subnet, int n_subnets, const int32_t* param_subnet,
               const float* base_lr, int n_params, const float* sched, double* state, int head_policy,
               int backbone_policy, int warmup_iters, float warmup_ratio, float T, float b, float ema_beta, float* lr,
               sm3_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * SAR branch, loss side of mmdet's GFLHead (local_configs/main_SM3Det.py:29-48,145-149; consumed at
 * mmrotate/models/detectors/trisource_H1stage_R2stage_detector.py:235-369).  mmdet 2.x is not vendored by the reference:
 * semantics restated from atss_assigner.py:47-201, gfl_head.py:16-50,210-330, gfocal_loss.py:12-52,95-118,
 * iou_loss.py:120-135 ("parity unpinned"; pinned on this package's two independent restatements).
 * Anchors (A,4) are level-major, level l = [level_off[l], level_off[l+1]) (HOST int[num_levels+1]), level_stride HOST floats.
 * sm3_atss_assign (one image): best[A] uint64, zeroed here; an anchor that is a positive of some gt gets
 *   max over those gts of (IoU bits << 32 | ~gt index) -- highest IoU wins, lower gt index on ties; candidates = per level the
 *   `topk` valid anchors nearest to the gt centre, ties to the lower anchor index; threshold = mean + unbiased std of the
 *   candidates' IoUs; positive = IoU >= threshold and anchor centre inside the gt by more than 0.01.  valid: A bytes or NULL.
 * sm3_atss_decode: best -> gt_inds (0 = negative, i + 1) and max_overlaps (-1e8 where unassigned; may be NULL).
 * sm3_gfl_loss_fwd: cls (B,A,C) logits, bbox (B,A,4(reg_max+1)) logits, best (B,A) keys, gts / gt_labels = all images'
 *   ground truth concatenated with gt_off[b] (DEVICE int32[B]) = first gt of image b, valid (B,A) bytes or NULL.  Writes
 *   sums = 4 x 8 doubles [term][level]: 0 sum of GIoU loss x weight_target, 1 DFL x weight_target, 2 QFL x label weight,
 *   3 weight_target (= avg_factor), and pos_count[B] -- the per-level sums mmdet's loss_single returns before normalisation.
 * sm3_gfl_loss_bwd: coef = DEVICE 3 x 8 floats (d total / d sums[0..2][level]); writes d cls, d bbox (every element). */
int sm3_atss_assign(const float* anchors, int A, const int* level_off, const float* level_stride, int num_levels,
                    const float* gts, int k, const uint8_t* valid, int topk, void* best, sm3_stream_t stream);
int sm3_atss_decode(const void* best, int A, int64_t* gt_inds, float* max_overlaps, sm3_stream_t stream);
int sm3_gfl_loss_fwd(const float* cls, const float* bbox, const float* anchors, int B, int A, int C, int reg_max,
                     const int* level_off, const float* level_stride, int num_levels, const void* best, const float* gts,
                     const int64_t* gt_labels, const int* gt_off, const uint8_t* valid, float pos_weight, float beta,
                     double* sums, int* pos_count, sm3_stream_t stream);
int sm3_gfl_loss_bwd(const float* cls, const float* bbox, const float* anchors, int B, int A, int C, int reg_max,
                     const int* level_off, const float* level_stride, int num_levels, const void* best, const float* gts,
                     const int64_t* gt_labels, const int* gt_off, const uint8_t* valid, float pos_weight, float beta,
                     const float* coef, float* dcls, float* dbbox, sm3_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SM3DET_HIP_H */
