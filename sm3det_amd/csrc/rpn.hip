// rpn.hip -- Oriented-RPN proposal glue on the device:
//   * relu_bwd: gradient of the F.relu between rpn_conv and the 1x1 heads (rotated_rpn_head.py:43-50);
//   * sigmoid:  scores = rpn_cls_score.sigmoid()  (oriented_rpn_head.py:236-238);
//   * decode:   gather the top-k anchors / deltas of a level (:248-254) and run MidpointOffsetCoder.decode =
//     delta2bbox (mmrotate/core/bbox/coder/delta_midpointoffset_rbbox_coder.py:150-238) -> poly2obb_le90
//     (mmrotate/core/bbox/transforms.py:301-331) plus obb2xyxy_le90 (:685-702) for the horizontal NMS, one thread per
//     box, reference operation order (compiled with -ffp-contract=off so that every fp32 op rounds like torch's).
#include <math.h>

#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dx, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = v[e] > 0.f ? g[e] : 0.f;
    reinterpret_cast<f32x4*>(dx)[i] = o;
  }
}

__global__ __launch_bounds__(256) void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = 1.0f / (1.0f + expf(-x[i]));
}

struct DecodeCfg {
  float mean[6], stdv[6];
  float max_ratio;  // |log(wh_ratio_clip)|
};

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// python-style float modulo (result has the sign of the divisor), as torch's `%`
__device__ __forceinline__ float pymod(float a, float b) {
  float r = fmodf(a, b);
  if (r != 0.f && ((r < 0.f) != (b < 0.f))) r += b;
  return r;
}

__global__ __launch_bounds__(256) void rpn_decode_le90_kernel(const float* __restrict__ anchors,
                                                             const float* __restrict__ deltas,
                                                             const float* __restrict__ scores,
                                                             const int64_t* __restrict__ order, int n, DecodeCfg c,
                                                             float* __restrict__ proposals,
                                                             float* __restrict__ hboxes,
                                                             float* __restrict__ scores_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long src = order ? order[i] : i;
  const float* a = anchors + src * 4;
  const float* d = deltas + src * 6;
  float dn[6];
#pragma unroll
  for (int k = 0; k < 6; k++) dn[k] = d[k] * c.stdv[k] + c.mean[k];
  const float dx = dn[0], dy = dn[1];
  const float dw = clampf(dn[2], -c.max_ratio, c.max_ratio), dh = clampf(dn[3], -c.max_ratio, c.max_ratio);
  const float px = (a[0] + a[2]) * 0.5f, py = (a[1] + a[3]) * 0.5f;
  const float pw = a[2] - a[0], ph = a[3] - a[1];
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  const float gx = px + pw * dx, gy = py + ph * dy;
  const float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
  const float da = clampf(dn[4], -0.5f, 0.5f), db = clampf(dn[5], -0.5f, 0.5f);
  const float ga = gx + da * gw, ga_ = gx - da * gw, gb = gy + db * gh, gb_ = gy - db * gh;
  // polys = [ga, y1, x2, gb, _ga, y2, x1, _gb], centred, every vertex pushed out to the longest half-diagonal
  float cx[4] = {ga - gx, x2 - gx, ga_ - gx, x1 - gx};
  float cy[4] = {y1 - gy, gb - gy, y2 - gy, gb_ - gy};
  float len[4], mx = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    len[k] = sqrtf(cx[k] * cx[k] + cy[k] * cy[k]);
    mx = k == 0 ? len[0] : fmaxf(mx, len[k]);
  }
  float qx[4], qy[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float s = mx / len[k];
    qx[k] = cx[k] * s + gx;
    qy[k] = cy[k] * s + gy;
  }
  // poly2obb_le90
  const float e1 = sqrtf((qx[0] - qx[1]) * (qx[0] - qx[1]) + (qy[0] - qy[1]) * (qy[0] - qy[1]));
  const float e2 = sqrtf((qx[1] - qx[2]) * (qx[1] - qx[2]) + (qy[1] - qy[2]) * (qy[1] - qy[2]));
  const float ang1 = atan2f(qy[1] - qy[0], qx[1] - qx[0]);
  const float ang2 = atan2f(qy[3] - qy[0], qx[3] - qx[0]);
  float ang = e1 > e2 ? ang1 : ang2;
  const float pi = 3.14159265358979323846f, hpi = 1.57079632679489661923f;
  ang = pymod(ang + hpi, pi) - hpi;  // norm_angle(., 'le90')
  const float ocx = (qx[0] + qx[2]) / 2.0f, ocy = (qy[0] + qy[2]) / 2.0f;
  const float w = fmaxf(e1, e2), h = fminf(e1, e2);
  float* p = proposals + (long)i * 5;
  p[0] = ocx; p[1] = ocy; p[2] = w; p[3] = h; p[4] = ang;
  // obb2xyxy_le90
  const float cs = cosf(ang), sn = sinf(ang);
  const float xb = fabsf(w / 2 * cs) + fabsf(h / 2 * sn);
  const float yb = fabsf(w / 2 * sn) + fabsf(h / 2 * cs);
  float* hb = hboxes + (long)i * 4;
  hb[0] = ocx - xb; hb[1] = ocy - yb; hb[2] = ocx + xb; hb[3] = ocy + yb;
  if (scores_out) scores_out[i] = scores[src];
}


// norm_angle(., 'le90')  (transforms.py:850-867)
__device__ __forceinline__ float norm_le90(float a) {
  const float pi = 3.14159265358979323846f, hpi = 1.57079632679489661923f;
  return pymod(a + hpi, pi) - hpi;
}

// MidpointOffsetCoder.encode = bbox2delta (delta_midpointoffset_rbbox_coder.py:87-148), angle version le90:
// proposals (n,4) x1,y1,x2,y2; gt (n,5) cx,cy,w,h,a -> deltas (n,6) dx,dy,dw,dh,da,db
__device__ __forceinline__ void midpoint_encode_le90(const float* __restrict__ p, const float* __restrict__ g,
                                                     const DecodeCfg& c, float* __restrict__ o) {
  const float px = (p[0] + p[2]) * 0.5f, py = (p[1] + p[3]) * 0.5f, pw = p[2] - p[0], ph = p[3] - p[1];
  const float cx = g[0], cy = g[1], w = g[2], h = g[3], a = g[4];
  const float cs = cosf(a), sn = sinf(a);
  // obb2xyxy_le90
  const float xb = fabsf(w / 2 * cs) + fabsf(h / 2 * sn), yb = fabsf(w / 2 * sn) + fabsf(h / 2 * cs);
  const float hx1 = cx - xb, hy1 = cy - yb, hx2 = cx + xb, hy2 = cy + yb;
  const float gx = (hx1 + hx2) * 0.5f, gy = (hy1 + hy2) * 0.5f, gw = hx2 - hx1, gh = hy2 - hy1;
  // obb2poly_le90 (transforms.py:474-499): corners (tl, tr, br, bl) rotated by a, then shifted
  const float rx[4] = {-w * 0.5f, w * 0.5f, w * 0.5f, -w * 0.5f};
  const float ry[4] = {-h * 0.5f, -h * 0.5f, h * 0.5f, h * 0.5f};
  float qx[4], qy[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    qx[k] = (cs * rx[k] + -sn * ry[k]) + cx;
    qy[k] = (sn * rx[k] + cs * ry[k]) + cy;
  }
  float ymin = qy[0], xmax = qx[0];
#pragma unroll
  for (int k = 1; k < 4; k++) {
    ymin = fminf(ymin, qy[k]);
    xmax = fmaxf(xmax, qx[k]);
  }
  float ga = -1000.f, gb = -1000.f;  // x of the topmost vertex / y of the rightmost vertex (0.1 px tolerance)
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float xa = fabsf(qy[k] - ymin) > 0.1f ? -1000.f : qx[k];
    const float yb2 = fabsf(qx[k] - xmax) > 0.1f ? -1000.f : qy[k];
    ga = k == 0 ? xa : fmaxf(ga, xa);
    gb = k == 0 ? yb2 : fmaxf(gb, yb2);
  }
  float d[6] = {(gx - px) / pw, (gy - py) / ph, logf(gw / pw), logf(gh / ph), (ga - gx) / gw, (gb - gy) / gh};
#pragma unroll
  for (int k = 0; k < 6; k++) o[k] = (d[k] - c.mean[k]) / c.stdv[k];
}

__global__ __launch_bounds__(256) void midpoint_encode_le90_kernel(const float* __restrict__ proposals,
                                                                  const float* __restrict__ gt, int n, DecodeCfg c,
                                                                  float* __restrict__ deltas) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  midpoint_encode_le90(proposals + (long)i * 4, gt + (long)i * 5, c, deltas + (long)i * 6);
}

// obb2xyxy(., 'le90') (transforms.py:685-702): (n,5) cx,cy,w,h,a -> (n,4) x1,y1,x2,y2 of the enclosing horizontal box
__global__ __launch_bounds__(256) void obb2xyxy_le90_kernel(const float* __restrict__ obb, int n, int stride,
                                                           float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* g = obb + (long)i * stride;
  const float cs = cosf(g[4]), sn = sinf(g[4]);
  const float xb = fabsf(g[2] / 2 * cs) + fabsf(g[3] / 2 * sn), yb = fabsf(g[2] / 2 * sn) + fabsf(g[3] / 2 * cs);
  float* o = out + (long)i * 4;
  o[0] = g[0] - xb; o[1] = g[1] - yb; o[2] = g[0] + xb; o[3] = g[1] + yb;
}

struct XywhaCfg {
  float mean[5], stdv[5];
  float max_ratio, norm_factor_pi;  // norm_factor * pi, 0 = None
  int edge_swap, proj_xy, clamp_h, clamp_w;  // clamp_* = max_shape (0 = None)
};

// DeltaXYWHAOBBoxCoder.decode = delta2bbox (delta_xywha_rbbox_coder.py:180-283), le90, add_ctr_clamp False,
// class-agnostic deltas (n,5)
__global__ __launch_bounds__(256) void xywha_decode_le90_kernel(const float* __restrict__ rois,
                                                               const float* __restrict__ deltas, int n, XywhaCfg c,
                                                               float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = rois + (long)i * 5;
  const float* d = deltas + (long)i * 5;
  float dn[5];
#pragma unroll
  for (int k = 0; k < 5; k++) dn[k] = d[k] * c.stdv[k] + c.mean[k];
  float da = dn[4];
  if (c.norm_factor_pi != 0.f) da *= c.norm_factor_pi;
  const float px = r[0], py = r[1], pw = r[2], ph = r[3], pa = r[4];
  const float dxw = pw * dn[0], dyh = ph * dn[1];
  const float dw = clampf(dn[2], -c.max_ratio, c.max_ratio), dh = clampf(dn[3], -c.max_ratio, c.max_ratio);
  const float gw = pw * expf(dw), gh = ph * expf(dh);
  float gx, gy;
  if (c.proj_xy) {
    const float cs = cosf(pa), sn = sinf(pa);
    gx = dn[0] * pw * cs - dn[1] * ph * sn + px;
    gy = dn[0] * pw * sn + dn[1] * ph * cs + py;
  } else {
    gx = px + dxw;
    gy = py + dyh;
  }
  float ga = norm_le90(pa + da);
  if (c.clamp_h > 0) {
    gx = clampf(gx, 0.f, (float)(c.clamp_w - 1));
    gy = clampf(gy, 0.f, (float)(c.clamp_h - 1));
  }
  float* o = out + (long)i * 5;
  if (c.edge_swap) {
    const bool keep = gw > gh;
    o[0] = gx; o[1] = gy; o[2] = keep ? gw : gh; o[3] = keep ? gh : gw;
    o[4] = norm_le90(keep ? ga : ga + 1.57079632679489661923f);
  } else {
    o[0] = gx; o[1] = gy; o[2] = gw; o[3] = gh; o[4] = ga;
  }
}

// DeltaXYWHAOBBoxCoder.encode = bbox2delta (:112-176), le90
__device__ __forceinline__ void xywha_encode_le90(const float* __restrict__ p, const float* __restrict__ g,
                                                  const XywhaCfg& c, float* __restrict__ o) {
  const float px = p[0], py = p[1], pw = p[2], ph = p[3], pa = p[4];
  const float gx = g[0], gy = g[1], gw = g[2], gh = g[3], ga = g[4];
  float dx, dy, dw, dh, da;
  if (c.proj_xy) {
    const float cs = cosf(pa), sn = sinf(pa);
    dx = (cs * (gx - px) + sn * (gy - py)) / pw;
    dy = (-sn * (gx - px) + cs * (gy - py)) / ph;
  } else {
    dx = (gx - px) / pw;
    dy = (gy - py) / ph;
  }
  if (c.edge_swap) {
    const float t1 = norm_le90(ga - pa), t2 = norm_le90(ga - pa + 1.57079632679489661923f);
    const bool first = fabsf(t1) < fabsf(t2);
    da = first ? t1 : t2;
    dw = logf((first ? gw : gh) / pw);
    dh = logf((first ? gh : gw) / ph);
  } else {
    da = norm_le90(ga - pa);
    dw = logf(gw / pw);
    dh = logf(gh / ph);
  }
  if (c.norm_factor_pi != 0.f) da /= c.norm_factor_pi;
  const float d[5] = {dx, dy, dw, dh, da};
#pragma unroll
  for (int k = 0; k < 5; k++) o[k] = (d[k] - c.mean[k]) / c.stdv[k];
}

__global__ __launch_bounds__(256) void xywha_encode_le90_kernel(const float* __restrict__ proposals,
                                                               const float* __restrict__ gt, int n, XywhaCfg c,
                                                               float* __restrict__ deltas) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xywha_encode_le90(proposals + (long)i * 5, gt + (long)i * 5, c, deltas + (long)i * 5);
}


// ======================================================================================== detection losses
// Targets + losses of the two-stage branch, fused and sync-free (reference call sites:
// oriented_rpn_head.py:26-187 `_get_targets_single` / `loss_single`, rotated_rpn_head.py:152-372 `get_targets` / `loss`,
// rotated_bbox_head.py:141-356 `_get_target_single` / `get_targets` / `loss`).  The reference materialises dense
// (anchors x 6) target / weight tensors per image, unmaps them, re-splits them per level and then evaluates the loss
// over all 261 888 anchors although only the <= 256 sampled ones per image carry a weight; here one small kernel visits
// the SAMPLED anchors only: gather prediction -> encode the target from (anchor, matched gt) -> loss term -> per-level
// sums, and a second one writes the gradient at those positions.  The loss functions are mmdet 2.x's
// (CrossEntropyLoss(use_sigmoid) = BCE-with-logits, CrossEntropyLoss = softmax CE, SmoothL1Loss(beta), each
// `sum(loss * weight) / (avg_factor + eps)`, eps = FLT_EPSILON) -- mmdet is not vendored by the reference: restated,
// parity unpinned for those formulas (oracle/loss_oracle.py); everything around them is pinned to the live reference.
constexpr float kAvgEps = 1.1920928955078125e-07f;  // torch.finfo(torch.float32).eps (mmdet weight_reduce_loss)

struct RpnLossArgs {
  sm3_rpn_loss_level lv[SM3_RPN_MAX_LEVELS];
  int num_levels, A;
  const float* anchors;
  const int64_t* idx;
  const uint8_t* is_pos;
  const uint8_t* valid;
  const int64_t* gt_inds;
  const float* gts;
  int B, S, Atot, Kmax;
  const int64_t* n_pos;
  const int64_t* n_neg;
  DecodeCfg coder;
  float beta, w_cls, w_bbox, pos_weight;
};

__device__ __forceinline__ float rpn_avg_factor(const RpnLossArgs& a) {
  // num_total_samples = sum over images of max(#pos, 1) + max(#neg, 1)   (rotated_rpn_head.py:232-233, :332-333)
  long n = 0;
  for (int b = 0; b < a.B; b++) n += max(a.n_pos[b], (int64_t)1) + max(a.n_neg[b], (int64_t)1);
  return (float)n + kAvgEps;
}

// sample s -> (level, image, position, anchor-in-position); false when the slot is unused
__device__ __forceinline__ bool rpn_locate(const RpnLossArgs& a, int s, int& b, int& l, long& pos, int& an, long& j) {
  if (!a.valid[s]) return false;
  b = s / a.S;
  j = a.idx[s];
  long local = j;
  l = 0;
  for (int q = 0; q < a.num_levels - 1; q++) {
    if (l == q && local >= a.lv[q].num_anchors) {
      local -= a.lv[q].num_anchors;
      l = q + 1;
    }
  }
  pos = local / a.A;
  an = (int)(local - pos * a.A);
  return true;
}

__global__ __launch_bounds__(256) void rpn_loss_fwd_kernel(RpnLossArgs a, float* __restrict__ out_cls,
                                                          float* __restrict__ out_bbox) {
  float acc_c[SM3_RPN_MAX_LEVELS], acc_b[SM3_RPN_MAX_LEVELS];
#pragma unroll
  for (int q = 0; q < SM3_RPN_MAX_LEVELS; q++) acc_c[q] = acc_b[q] = 0.f;
  const int total = a.B * a.S;
  for (int s = threadIdx.x; s < total; s += 256) {
    int b, l, an;
    long pos, j;
    if (!rpn_locate(a, s, b, l, pos, an, j)) continue;
    sm3_rpn_loss_level L = a.lv[0];
#pragma unroll
    for (int q = 1; q < SM3_RPN_MAX_LEVELS; q++)
      if (q == l) L = a.lv[q];
    const float x = L.cls[b * L.cls_stride[0] + pos * L.cls_stride[1] + an * L.cls_stride[2]];
    const bool fg = a.is_pos[s] != 0;
    const float lw = fg ? (a.pos_weight > 0.f ? a.pos_weight : 1.f) : 1.f;  // :112-118
    // binary_cross_entropy_with_logits(x, t): max(x, 0) - x t + log(1 + exp(-|x|))
    const float bce = (fmaxf(x, 0.f) - (fg ? x : 0.f) + log1pf(expf(-fabsf(x)))) * lw;
    float sl1 = 0.f;
    if (fg) {
      const long g = a.gt_inds[(long)b * a.Atot + j] - 1;
      float tgt[6];
      midpoint_encode_le90(a.anchors + j * 4, a.gts + ((long)b * a.Kmax + g) * 5, a.coder, tgt);
#pragma unroll
      for (int c = 0; c < 6; c++) {
        const float pr = L.reg[b * L.reg_stride[0] + pos * L.reg_stride[1] + (an * 6 + c) * L.reg_stride[2]];
        const float d = fabsf(pr - tgt[c]);
        sl1 += d < a.beta ? 0.5f * d * d / a.beta : d - 0.5f * a.beta;
      }
    }
#pragma unroll
    for (int q = 0; q < SM3_RPN_MAX_LEVELS; q++) {
      acc_c[q] += q == l ? bce : 0.f;
      acc_b[q] += q == l ? sl1 : 0.f;
    }
  }
  __shared__ float red[4][2 * SM3_RPN_MAX_LEVELS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < SM3_RPN_MAX_LEVELS; q++) {
    float c = acc_c[q], bb = acc_b[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      c += __shfl_xor(c, o, 64);
      bb += __shfl_xor(bb, o, 64);
    }
    if (lane == 0) {
      red[wave][q] = c;
      red[wave][SM3_RPN_MAX_LEVELS + q] = bb;
    }
  }
  __syncthreads();
  if (threadIdx.x < a.num_levels) {
    const int q = threadIdx.x;
    const float avg = rpn_avg_factor(a);
    out_cls[q] = a.w_cls * (((red[0][q] + red[1][q]) + (red[2][q] + red[3][q])) / avg);
    out_bbox[q] = a.w_bbox * (((red[0][SM3_RPN_MAX_LEVELS + q] + red[1][SM3_RPN_MAX_LEVELS + q]) +
                               (red[2][SM3_RPN_MAX_LEVELS + q] + red[3][SM3_RPN_MAX_LEVELS + q])) / avg);
  }
}

// gradient at the sampled anchors (the dense gradient maps were zero-filled by the caller; sampled anchors are unique per
// image, so plain stores)
__global__ __launch_bounds__(256) void rpn_loss_bwd_kernel(RpnLossArgs a, const float* __restrict__ g_cls,
                                                          const float* __restrict__ g_bbox) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= a.B * a.S) return;
  int b, l, an;
  long pos, j;
  if (!rpn_locate(a, s, b, l, pos, an, j)) return;
  sm3_rpn_loss_level L = a.lv[0];
#pragma unroll
  for (int q = 1; q < SM3_RPN_MAX_LEVELS; q++)
    if (q == l) L = a.lv[q];
  const float avg = rpn_avg_factor(a);
  const float x = L.cls[b * L.cls_stride[0] + pos * L.cls_stride[1] + an * L.cls_stride[2]];
  const bool fg = a.is_pos[s] != 0;
  const float lw = fg ? (a.pos_weight > 0.f ? a.pos_weight : 1.f) : 1.f;
  const float sg = 1.f / (1.f + expf(-x));
  L.dcls[b * L.dcls_stride[0] + pos * L.dcls_stride[1] + an * L.dcls_stride[2]] =
      (sg - (fg ? 1.f : 0.f)) * lw * (a.w_cls / avg) * g_cls[l];
  if (fg) {
    const long g = a.gt_inds[(long)b * a.Atot + j] - 1;
    float tgt[6];
    midpoint_encode_le90(a.anchors + j * 4, a.gts + ((long)b * a.Kmax + g) * 5, a.coder, tgt);
    const float k = (a.w_bbox / avg) * g_bbox[l];
#pragma unroll
    for (int c = 0; c < 6; c++) {
      const float d = L.reg[b * L.reg_stride[0] + pos * L.reg_stride[1] + (an * 6 + c) * L.reg_stride[2]] - tgt[c];
      const float gr = fabsf(d) < a.beta ? d / a.beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      L.dreg[b * L.dreg_stride[0] + pos * L.dreg_stride[1] + (an * 6 + c) * L.dreg_stride[2]] = gr * k;
    }
  }
}

// RotatedBBoxHead.loss on fixed-size sample blocks (rotated_bbox_head.py:275-356) with the targets of
// _get_target_single (:141-207) built in place: row i is a sampled RoI (valid[i]), labels[i] in [0, C) for positives
// (matched gt gts[i]) and C for negatives.  out = [loss_cls, loss_bbox, acc].
struct RcnnLossArgs {
  const float* cls;
  const float* reg;
  float* dcls;
  float* dreg;
  int ld_cls, ld_reg, C1, N;
  const int64_t* labels;
  const uint8_t* valid;
  const float* rois;     // (N, 5)
  const float* gts;      // (N, 5)
  const float* label_w;  // optional (N): the reference API's precomputed label_weights (then every row counts)
  const float* targets;  // optional (N, 5): precomputed bbox_targets instead of encode(rois, gts)
  XywhaCfg coder;
  float beta, w_cls, w_bbox, pos_weight;
};

__device__ __forceinline__ bool rcnn_row(const RcnnLossArgs& a, int i, int C, int& lab, bool& fg, float& lw) {
  if (a.valid && !a.valid[i]) return false;
  lab = (int)a.labels[i];
  fg = lab >= 0 && lab < C;
  lw = a.label_w ? a.label_w[i] : (fg ? (a.pos_weight > 0.f ? a.pos_weight : 1.f) : 1.f);  // :190-201
  return true;
}

__device__ __forceinline__ void rcnn_target(const RcnnLossArgs& a, int i, float* tgt) {
  if (a.targets) {
#pragma unroll
    for (int c = 0; c < 5; c++) tgt[c] = a.targets[(long)i * 5 + c];
  } else {
    xywha_encode_le90(a.rois + (long)i * 5, a.gts + (long)i * 5, a.coder, tgt);
  }
}

__device__ __forceinline__ float rcnn_row_lse(const float* __restrict__ x, int C1, float& mx) {
  mx = x[0];
  for (int c = 1; c < C1; c++) mx = fmaxf(mx, x[c]);
  float se = 0.f;
  for (int c = 0; c < C1; c++) se += expf(x[c] - mx);
  return mx + logf(se);
}

__global__ __launch_bounds__(512) void rcnn_loss_fwd_kernel(RcnnLossArgs a, float* __restrict__ out,
                                                           float* __restrict__ counts) {
  float ce = 0.f, sl1 = 0.f, nvalid = 0.f, nw = 0.f, ncorrect = 0.f;
  const int C = a.C1 - 1;
  for (int i = threadIdx.x; i < a.N; i += 512) {
    int lab;
    bool fg;
    float lw;
    if (!rcnn_row(a, i, C, lab, fg, lw)) continue;
    const float* x = a.cls + (long)i * a.ld_cls;
    float mx;
    const float lse = rcnn_row_lse(x, a.C1, mx);
    ce += (lse - x[lab]) * lw;
    nvalid += 1.f;
    nw += lw > 0.f ? 1.f : 0.f;
    int am = 0;
    for (int c = 1; c < a.C1; c++)
      if (x[c] > x[am]) am = c;
    ncorrect += am == lab ? 1.f : 0.f;
    if (fg) {
      float tgt[5];
      rcnn_target(a, i, tgt);
#pragma unroll
      for (int c = 0; c < 5; c++) {
        const float d = fabsf(a.reg[(long)i * a.ld_reg + c] - tgt[c]);
        sl1 += d < a.beta ? 0.5f * d * d / a.beta : d - 0.5f * a.beta;
      }
    }
  }
  __shared__ float red[8][5];
  float v[5] = {ce, sl1, nvalid, nw, ncorrect};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 5; q++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[q] += __shfl_xor(v[q], o, 64);
    if (lane == 0) red[wave][q] = v[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[5];
#pragma unroll
    for (int q = 0; q < 5; q++) {
      t[q] = 0.f;
      for (int w = 0; w < 8; w++) t[q] += red[w][q];
    }
    const float avg_cls = fmaxf(t[3], 1.f) + kAvgEps;  // max(sum(label_weights > 0), 1)  (:305)
    const float avg_box = t[2] + kAvgEps;              // bbox_targets.size(0) = all sampled RoIs (:347)
    out[0] = a.w_cls * (t[0] / avg_cls);
    out[1] = t[2] > 0.f ? a.w_bbox * (t[1] / avg_box) : 0.f;
    out[2] = t[2] > 0.f ? t[4] * (100.f / t[2]) : 0.f;  // accuracy(cls_score, labels), top-1, per cent
    counts[0] = avg_cls;
    counts[1] = avg_box;
  }
}

__global__ __launch_bounds__(256) void rcnn_loss_bwd_kernel(RcnnLossArgs a, const float* __restrict__ counts,
                                                           const float* __restrict__ g_cls,
                                                           const float* __restrict__ g_bbox) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.N) return;
  float* dc = a.dcls + (long)i * a.ld_cls;
  float* dr = a.dreg + (long)i * a.ld_reg;
  const int C = a.C1 - 1;
  int lab;
  bool fg;
  float lw;
  if (!rcnn_row(a, i, C, lab, fg, lw)) {
    for (int c = 0; c < a.C1; c++) dc[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 5; c++) dr[c] = 0.f;
    return;
  }
  const float* x = a.cls + (long)i * a.ld_cls;
  float mx;
  const float lse = rcnn_row_lse(x, a.C1, mx);
  const float kc = lw * (a.w_cls / counts[0]) * g_cls[0];
  for (int c = 0; c < a.C1; c++) dc[c] = (expf(x[c] - lse) - (c == lab ? 1.f : 0.f)) * kc;
  if (fg) {
    float tgt[5];
    rcnn_target(a, i, tgt);
    const float kb = (a.w_bbox / counts[1]) * g_bbox[0];
#pragma unroll
    for (int c = 0; c < 5; c++) {
      const float d = a.reg[(long)i * a.ld_reg + c] - tgt[c];
      dr[c] = (fabsf(d) < a.beta ? d / a.beta : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f))) * kb;
    }
  } else {
#pragma unroll
    for (int c = 0; c < 5; c++) dr[c] = 0.f;
  }
}

int blocks_for(long n) {
  long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  return b < 1 ? 1 : (int)b;
}

}  // namespace

extern "C" {

int sm3_relu_bwd(const float* dy, const float* y, float* dx, long n, sm3_stream_t stream) {
  if (!dy || !y || !dx || n < 0) return SM3_ERR_INVALID_ARG;
  if (n & 3) return SM3_ERR_UNSUPPORTED;
  if (n == 0) return SM3_OK;
  relu_bwd_kernel<<<blocks_for(n / 4), 256, 0, (hipStream_t)stream>>>(dy, y, dx, n / 4);
  return launch_status();
}

int sm3_sigmoid_f32(const float* x, float* y, long n, sm3_stream_t stream) {
  if (!x || !y || n < 0) return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  sigmoid_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(x, y, n);
  return launch_status();
}

int sm3_rpn_decode_le90(const float* anchors, const float* deltas, const float* scores, const int64_t* order, int n,
                        const float* means6, const float* stds6, float wh_ratio_clip, float* proposals,
                        float* hboxes, float* scores_out, sm3_stream_t stream) {
  if (!anchors || !deltas || !proposals || !hboxes || !means6 || !stds6 || n < 0 || !(wh_ratio_clip > 0.f))
    return SM3_ERR_INVALID_ARG;
  if (scores_out && !scores) return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  DecodeCfg c;
  for (int k = 0; k < 6; k++) {
    c.mean[k] = means6[k];
    c.stdv[k] = stds6[k];
  }
  c.max_ratio = (float)fabs(log((double)wh_ratio_clip));  // np.abs(np.log(wh_ratio_clip)) is a double in the reference
  rpn_decode_le90_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(anchors, deltas, scores, order, n, c,
                                                                          proposals, hboxes, scores_out);
  return launch_status();
}


int sm3_midpoint_offset_encode_le90(const float* proposals, const float* gt, int n, const float* means6,
                                    const float* stds6, float* deltas, sm3_stream_t stream) {
  if (!proposals || !gt || !deltas || !means6 || !stds6 || n < 0) return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  DecodeCfg c;
  for (int k = 0; k < 6; k++) {
    c.mean[k] = means6[k];
    c.stdv[k] = stds6[k];
  }
  c.max_ratio = 0.f;
  midpoint_encode_le90_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(proposals, gt, n, c, deltas);
  return launch_status();
}

static int fill_xywha(XywhaCfg& c, const float* means5, const float* stds5, float wh_ratio_clip, float norm_factor,
                      int edge_swap, int proj_xy, int max_h, int max_w) {
  if (!means5 || !stds5) return SM3_ERR_INVALID_ARG;
  for (int k = 0; k < 5; k++) {
    c.mean[k] = means5[k];
    c.stdv[k] = stds5[k];
  }
  c.max_ratio = wh_ratio_clip > 0.f ? (float)fabs(log((double)wh_ratio_clip)) : 0.f;
  c.norm_factor_pi = norm_factor != 0.f ? (float)((double)norm_factor * 3.14159265358979323846) : 0.f;
  c.edge_swap = edge_swap;
  c.proj_xy = proj_xy;
  c.clamp_h = max_h;
  c.clamp_w = max_w;
  return SM3_OK;
}

int sm3_delta_xywha_decode_le90(const float* rois, const float* deltas, int n, const float* means5,
                                const float* stds5, float wh_ratio_clip, float norm_factor, int edge_swap,
                                int proj_xy, int max_h, int max_w, float* out, sm3_stream_t stream) {
  if (!rois || !deltas || !out || n < 0 || !(wh_ratio_clip > 0.f)) return SM3_ERR_INVALID_ARG;
  XywhaCfg c;
  if (int rc = fill_xywha(c, means5, stds5, wh_ratio_clip, norm_factor, edge_swap, proj_xy, max_h, max_w)) return rc;
  if (n == 0) return SM3_OK;
  xywha_decode_le90_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(rois, deltas, n, c, out);
  return launch_status();
}

int sm3_delta_xywha_encode_le90(const float* proposals, const float* gt, int n, const float* means5,
                                const float* stds5, float norm_factor, int edge_swap, int proj_xy, float* deltas,
                                sm3_stream_t stream) {
  if (!proposals || !gt || !deltas || n < 0) return SM3_ERR_INVALID_ARG;
  XywhaCfg c;
  if (int rc = fill_xywha(c, means5, stds5, 0.f, norm_factor, edge_swap, proj_xy, 0, 0)) return rc;
  if (n == 0) return SM3_OK;
  xywha_encode_le90_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(proposals, gt, n, c, deltas);
  return launch_status();
}

int sm3_obb2xyxy_le90(const float* obb, int n, int stride, float* out, sm3_stream_t stream) {
  if (!obb || !out || n < 0 || stride < 5) return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  obb2xyxy_le90_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(obb, n, stride, out);
  return launch_status();
}

static int fill_rpn_loss(RpnLossArgs& a, const sm3_rpn_loss_desc* d) {
  if (!d || d->num_levels < 1 || d->num_levels > SM3_RPN_MAX_LEVELS || d->anchors_per_pos < 1 || d->batch < 1 ||
      d->samples < 1 || d->max_gts < 1 || !d->anchors || !d->idx || !d->is_pos || !d->valid || !d->gt_inds || !d->gts ||
      !d->n_pos || !d->n_neg || !(d->beta > 0.f))
    return SM3_ERR_INVALID_ARG;
  long tot = 0;
  for (int l = 0; l < d->num_levels; l++) {
    if (!d->level[l].cls || !d->level[l].reg || d->level[l].num_anchors < 0) return SM3_ERR_INVALID_ARG;
    a.lv[l] = d->level[l];
    tot += d->level[l].num_anchors;
  }
  for (int l = d->num_levels; l < SM3_RPN_MAX_LEVELS; l++) a.lv[l] = d->level[d->num_levels - 1];
  if (tot != d->total_anchors) return SM3_ERR_INVALID_ARG;
  a.num_levels = d->num_levels; a.A = d->anchors_per_pos;
  a.anchors = d->anchors; a.idx = d->idx; a.is_pos = d->is_pos; a.valid = d->valid; a.gt_inds = d->gt_inds;
  a.gts = d->gts; a.B = d->batch; a.S = d->samples; a.Atot = d->total_anchors; a.Kmax = d->max_gts;
  a.n_pos = d->n_pos; a.n_neg = d->n_neg;
  for (int k = 0; k < 6; k++) {
    a.coder.mean[k] = d->means[k];
    a.coder.stdv[k] = d->stds[k];
  }
  a.coder.max_ratio = 0.f;
  a.beta = d->beta; a.w_cls = d->loss_weight_cls; a.w_bbox = d->loss_weight_bbox; a.pos_weight = d->pos_weight;
  return SM3_OK;
}

int sm3_rpn_loss_forward(const sm3_rpn_loss_desc* d, float* loss_cls, float* loss_bbox, sm3_stream_t stream) {
  RpnLossArgs a;
  if (int rc = fill_rpn_loss(a, d)) return rc;
  if (!loss_cls || !loss_bbox) return SM3_ERR_INVALID_ARG;
  rpn_loss_fwd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(a, loss_cls, loss_bbox);
  return launch_status();
}

int sm3_rpn_loss_backward(const sm3_rpn_loss_desc* d, const float* dloss_cls, const float* dloss_bbox,
                          sm3_stream_t stream) {
  RpnLossArgs a;
  if (int rc = fill_rpn_loss(a, d)) return rc;
  if (!dloss_cls || !dloss_bbox) return SM3_ERR_INVALID_ARG;
  for (int l = 0; l < d->num_levels; l++)
    if (!d->level[l].dcls || !d->level[l].dreg) return SM3_ERR_INVALID_ARG;
  const int total = d->batch * d->samples;
  rpn_loss_bwd_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, dloss_cls, dloss_bbox);
  return launch_status();
}

static int fill_rcnn_loss(RcnnLossArgs& a, const sm3_rcnn_loss_desc* d) {
  if (!d || d->num_rois < 0 || d->num_classes < 1 || !d->cls_score || !d->bbox_pred || !d->labels ||
      (!d->valid && !d->label_weights) || (!d->bbox_targets && (!d->rois || !d->gts)) ||
      d->ld_cls < d->num_classes + 1 || d->ld_reg < 5 || !(d->beta > 0.f))
    return SM3_ERR_INVALID_ARG;
  a.cls = d->cls_score; a.reg = d->bbox_pred; a.dcls = d->dcls_score; a.dreg = d->dbbox_pred;
  a.ld_cls = d->ld_cls; a.ld_reg = d->ld_reg; a.C1 = d->num_classes + 1; a.N = d->num_rois;
  a.labels = d->labels; a.valid = d->valid; a.rois = d->rois; a.gts = d->gts;
  a.label_w = d->label_weights; a.targets = d->bbox_targets;
  if (int rc = fill_xywha(a.coder, d->means, d->stds, 0.f, d->norm_factor, d->edge_swap, d->proj_xy, 0, 0)) return rc;
  a.beta = d->beta; a.w_cls = d->loss_weight_cls; a.w_bbox = d->loss_weight_bbox; a.pos_weight = d->pos_weight;
  return SM3_OK;
}

int sm3_rcnn_loss_forward(const sm3_rcnn_loss_desc* d, float* out3, float* counts2, sm3_stream_t stream) {
  RcnnLossArgs a;
  if (int rc = fill_rcnn_loss(a, d)) return rc;
  if (!out3 || !counts2) return SM3_ERR_INVALID_ARG;
  rcnn_loss_fwd_kernel<<<1, 512, 0, (hipStream_t)stream>>>(a, out3, counts2);
  return launch_status();
}

int sm3_rcnn_loss_backward(const sm3_rcnn_loss_desc* d, const float* counts2, const float* dloss_cls,
                           const float* dloss_bbox, sm3_stream_t stream) {
  RcnnLossArgs a;
  if (int rc = fill_rcnn_loss(a, d)) return rc;
  if (!counts2 || !dloss_cls || !dloss_bbox || !a.dcls || !a.dreg) return SM3_ERR_INVALID_ARG;
  if (a.N == 0) return SM3_OK;
  rcnn_loss_bwd_kernel<<<(a.N + 255) / 256, 256, 0, (hipStream_t)stream>>>(a, counts2, dloss_cls, dloss_bbox);
  return launch_status();
}

}  // extern "C"
