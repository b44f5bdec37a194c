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
 *   (n_rois,channels,pooled_h,pooled_w) NCHW contiguous.  grad_input must arrive zero-filled
 *   (mmcv/ops/roi_align_rotated.py:92) and is accumulated with fp32 atomics. */
int sm3_roi_align_rotated_forward(const float* input, const float* rois, float* output, int n_rois,
                                  int batch, int channels, int height, int width, int pooled_h,
                                  int pooled_w, float spatial_scale, int sampling_ratio, int aligned,
                                  int clockwise, int layout, sm3_stream_t stream);
int sm3_roi_align_rotated_backward(const float* grad_output, const float* rois, float* grad_input,
                                   int n_rois, int batch, int channels, int height, int width,
                                   int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                   int aligned, int clockwise, int layout, sm3_stream_t stream);

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
#define SM3_EPI_BIAS_GELU 2      /* aux_out = acc + bias ; C = gelu_erf(aux_out)              (FFN first linear) */
#define SM3_EPI_BIAS_SCALE_RES 3 /* aux_out = y = acc + bias ; C = aux_in + gamma[n]*rowscale[m/rows_per_scale]*y */
#define SM3_EPI_GELU_BWD 4       /* C = acc * gelu_erf'(aux_in)                               (dgrad through GELU) */
typedef struct sm3_gemm_desc {
  int32_t mode, epilogue;
  const float* A;
  const float* B;
  float* C;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  const int32_t* group_offsets;
  int32_t num_groups;
  int32_t splits; /* TN only: split-K factor per group */
  int64_t stride_b, stride_bias;
  const float* bias;
  const float* aux_in;
  float* aux_out;
  const float* gamma;
  const float* rowscale;
  int32_t rows_per_scale;
  int32_t ld_aux;
} sm3_gemm_desc;
size_t sm3_gemm_f32_workspace_bytes(const sm3_gemm_desc* desc);
int sm3_gemm_f32(const sm3_gemm_desc* desc, void* workspace, size_t workspace_bytes, sm3_stream_t stream);
/* out[g][n] = sum over rows of group g of x[r][n]  (bias gradients); out is overwritten */
int sm3_colsum_f32(const float* x, int ld, int m, int n, const int32_t* group_offsets, int num_groups, float* out,
                   sm3_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SM3DET_HIP_H */
