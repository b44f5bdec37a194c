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
__global__ __launch_bounds__(256) void midpoint_encode_le90_kernel(const float* __restrict__ proposals,
                                                                  const float* __restrict__ gt, int n, DecodeCfg c,
                                                                  float* __restrict__ deltas) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = proposals + (long)i * 4;
  const float* g = gt + (long)i * 5;
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
  float* o = deltas + (long)i * 6;
#pragma unroll
  for (int k = 0; k < 6; k++) o[k] = (d[k] - c.mean[k]) / c.stdv[k];
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
__global__ __launch_bounds__(256) void xywha_encode_le90_kernel(const float* __restrict__ proposals,
                                                               const float* __restrict__ gt, int n, XywhaCfg c,
                                                               float* __restrict__ deltas) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* p = proposals + (long)i * 5;
  const float* g = gt + (long)i * 5;
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
  float* o = deltas + (long)i * 5;
#pragma unroll
  for (int k = 0; k < 5; k++) o[k] = (d[k] - c.mean[k]) / c.stdv[k];
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

}  // extern "C"
