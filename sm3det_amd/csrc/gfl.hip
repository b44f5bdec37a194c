// gfl.hip -- the loss side of the SAR branch's GFLHead on the device (libsm3det_hip.so, gfx950): ATSS assignment and the
// Quality-Focal / Distribution-Focal / GIoU losses with the Integral decoder as three kernels + their backward.
//
// What it restates: mmdet 2.x (NOT vendored by the reference; consumed through local_configs/main_SM3Det.py:29-48,145-149 and
// mmrotate/models/detectors/trisource_H1stage_R2stage_detector.py:235-369) --
//   mmdet/core/bbox/assigners/atss_assigner.py:47-201          ATSSAssigner.assign
//   mmdet/models/dense_heads/gfl_head.py:16-50, 210-330         Integral, GFLHead.loss_single / loss
//   mmdet/models/losses/gfocal_loss.py:12-52, 95-118            quality_focal_loss, distribution_focal_loss
//   mmdet/models/losses/iou_loss.py:120-135                     giou_loss (bbox_overlaps mode='giou', eps 1e-7)
//   mmdet/core/bbox/transforms.py                               distance2bbox / bbox2distance
// "parity unpinned" by reference files (none exists under /root/reference); pinned instead on this package's two independent
// restatements (the masked torch form sm3det_amd/gfl_losses.py and mmdet's indexing form oracle/gfl_oracle.py).
//
// Round 5 ran these as ~200 small torch launches per step (plus the tree's only hipBLASLt calls, the Integral's F.linear).
// Here: one workgroup per ground-truth box selects its candidates (per pyramid level the `topk` anchors nearest to the gt
// centre: per-lane sorted lists in registers, merged by wave-wide minima), sets its mean + std IoU threshold and claims
// its positives with one 64-bit atomicMax per anchor (IoU bits | gt index: the highest IoU wins, the lower gt on ties); one
// pass per anchor evaluates all three losses and one more their gradients.  Ties between equal centre distances go to the
// LOWER anchor index (torch.topk leaves that order unspecified; the oracle follows the same rule).
#include "common.h"

namespace {

constexpr int GFL_MAX_LEVELS = 8;
constexpr int GFL_MAX_TOPK = 16;
struct Levels {
  int n;
  int off[GFL_MAX_LEVELS + 1];  // anchors of level l: [off[l], off[l + 1])
  float stride[GFL_MAX_LEVELS];
};

__device__ __forceinline__ int level_of(const Levels& lv, int a) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < GFL_MAX_LEVELS; i++) l += (i < lv.n && a >= lv.off[i]) ? 1 : 0;
  return l;
}

// mmdet bbox_overlaps(mode='iou'), (x1, y1, x2, y2) boxes
__device__ __forceinline__ float hbb_iou(const float* a, const float* b, float eps) {
  const float a1 = (a[2] - a[0]) * (a[3] - a[1]);
  const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
  const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
  const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  const float ov = w * h;
  return ov / fmaxf(a1 + a2 - ov, eps);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w < v ? w : v;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------------------------- ATSS
// One workgroup per gt, one wave per pyramid level.  best[a] (zeroed by the caller) receives, for every anchor that is a
// positive of some gt, max over those gts of (IoU bits << 32 | ~gt index).
__global__ __launch_bounds__(64 * GFL_MAX_LEVELS) void atss_assign_kernel(const float* __restrict__ anchors, Levels lv,
                                                                         const float* __restrict__ gts, int k,
                                                                         const uint8_t* __restrict__ valid, int topk,
                                                                         unsigned long long* __restrict__ best) {
  __shared__ int c_idx[GFL_MAX_LEVELS][GFL_MAX_TOPK];
  __shared__ float c_iou[GFL_MAX_LEVELS][GFL_MAX_TOPK];
  __shared__ float s_thr;
  const int g = blockIdx.x, l = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float gt[4] = {gts[4 * g], gts[4 * g + 1], gts[4 * g + 2], gts[4 * g + 3]};
  const float gx = (gt[0] + gt[2]) / 2.0f, gy = (gt[1] + gt[3]) / 2.0f;
  if (l < lv.n) {
    // per lane: the `topk` nearest of its strided share, ascending by (distance, index), in registers (static indices only)
    unsigned long long lst[GFL_MAX_TOPK];
#pragma unroll
    for (int t = 0; t < GFL_MAX_TOPK; t++) lst[t] = ~0ull;
    const int s = lv.off[l], e = lv.off[l + 1];
    const int sel = min(topk, e - s);
    for (int a = s + lane; a < e; a += 64) {
      const float* b = anchors + 4 * (long)a;
      const float dx = (b[0] + b[2]) / 2.0f - gx, dy = (b[1] + b[3]) / 2.0f - gy;
      float d = sqrtf(dx * dx + dy * dy);
      if (valid && !valid[a]) d = __builtin_inff();
      unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)a;  // d >= 0: bits are ordered
#pragma unroll
      for (int t = 0; t < GFL_MAX_TOPK; t++) {  // insertion into the sorted list: carry the larger key down
        if (t < topk) {
          const unsigned long long cur = lst[t];
          const bool sw = key < cur;
          lst[t] = sw ? key : cur;
          key = sw ? cur : key;
        }
      }
    }
    // merge: `sel` rounds of wave-wide minimum over the lanes' heads; the winner pops its head
    for (int t = 0; t < sel; t++) {
      const unsigned long long m = wave_min_u64(lst[0]);
      if (lst[0] == m) {  // exactly one lane (indices are unique)
#pragma unroll
        for (int u = 0; u + 1 < GFL_MAX_TOPK; u++) lst[u] = lst[u + 1];
        lst[GFL_MAX_TOPK - 1] = ~0ull;
      }
      if (lane == 0) {
        const bool fin = (unsigned)(m >> 32) < 0x7f800000u;  // finite distance: a valid anchor
        const int a = (int)(unsigned)m;
        c_idx[l][t] = fin ? a : -1;
        c_iou[l][t] = fin ? hbb_iou(anchors + 4 * (long)a, gt, 1e-6f) : 0.f;
      }
    }
    if (lane == 0)
      for (int t = sel; t < GFL_MAX_TOPK; t++) c_idx[l][t] = -1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // mean + unbiased std of the valid candidates' IoUs (<= levels x topk values), double accumulation
    double sum = 0;
    int n = 0;
    for (int i = 0; i < lv.n; i++)
      for (int t = 0; t < topk; t++)
        if (c_idx[i][t] >= 0) { sum += (double)c_iou[i][t]; n++; }
    const double mean = n > 0 ? sum / n : 0.0;
    double var = 0;
    for (int i = 0; i < lv.n; i++)
      for (int t = 0; t < topk; t++)
        if (c_idx[i][t] >= 0) { const double d = (double)c_iou[i][t] - mean; var += d * d; }
    // torch: cand_ov.mean(0) + cand_ov.std(0) in fp32; n == 1 gives std = nan there (no positive): reproduced by the NaN compare
    const float stdv = n > 1 ? (float)sqrt(var / (n - 1)) : __builtin_nanf("");
    s_thr = (float)mean + stdv;
  }
  __syncthreads();
  const float thr = s_thr;
  if (l < lv.n && lane < topk) {
    const int a = c_idx[l][lane];
    if (a >= 0) {
      const float iou = c_iou[l][lane];
      const float* b = anchors + 4 * (long)a;
      const float cx = (b[0] + b[2]) / 2.0f, cy = (b[1] + b[3]) / 2.0f;
      const float side = fminf(fminf(cx - gt[0], cy - gt[1]), fminf(gt[2] - cx, gt[3] - cy));
      if (iou >= thr && side > 0.01f)
        atomicMax(best + a, ((unsigned long long)__float_as_uint(iou) << 32) | (unsigned)(0xffffffffu - (unsigned)g));
    }
  }
}

__global__ void atss_decode_kernel(const unsigned long long* __restrict__ best, int A, int64_t* __restrict__ gt_inds,
                                   float* __restrict__ max_ov) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= A) return;
  const unsigned long long b = best[a];
  gt_inds[a] = b ? (int64_t)(0xffffffffu - (unsigned)b) + 1 : 0;
  if (max_ov) max_ov[a] = b ? __uint_as_float((unsigned)(b >> 32)) : -100000000.0f;
}

// ---------------------------------------------------------------------------------------------------------------- losses
struct GflCfg {
  int B, A, C, R;  // images, anchors per image, classes, reg_max
  float beta;      // QFL exponent
};

__device__ __forceinline__ float bce_logits(float x, float t) {  // F.binary_cross_entropy_with_logits
  return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// What one anchor of one image needs from the assignment: positive?, its gt box / label, the label weight
struct AnchorTarget {
  bool pos;
  float t[4];
  int label;
  float lw;
};
__device__ __forceinline__ AnchorTarget anchor_target(const unsigned long long* __restrict__ best, const float* __restrict__ gts,
                                                      const int64_t* __restrict__ gt_labels, const int* __restrict__ gt_off,
                                                      const uint8_t* __restrict__ valid, float pos_weight, int b, int a, int A) {
  AnchorTarget o;
  const bool v = valid ? valid[(long)b * A + a] != 0 : true;
  const unsigned long long key = best[(long)b * A + a];
  o.pos = key != 0 && v;
  o.lw = v ? 1.f : 0.f;
  o.label = -1;
  o.t[0] = o.t[1] = o.t[2] = o.t[3] = 0.f;
  if (key != 0) {
    const int g = gt_off[b] + (int)(0xffffffffu - (unsigned)key);
#pragma unroll
    for (int i = 0; i < 4; i++) o.t[i] = gts[4 * (long)g + i];
    o.label = (int)gt_labels[g];
    if (pos_weight > 0.f) o.lw *= pos_weight;
  }
  return o;
}

// Integral + distance2bbox + the box-side losses of ONE positive anchor.  Forward values; when GRAD, also d(loss)/d(logits)
// into dbp (68 values) for the weights cbox (GIoU term) and cdfl (DFL term).
template <bool GRAD>
__device__ __forceinline__ void box_terms(const float* __restrict__ bp, int R, float cx, float cy, const float (&tt)[4],
                                          float& score, float& giou_l, float& dfl, float cbox, float cdfl,
                                          float* __restrict__ dbp) {
  float corner[4], lse[4], tgt[4];
  // target side distances in stride units, clamped to [0, R - 0.1] (bbox2distance, eps 0.1)
  tgt[0] = cx - tt[0];
  tgt[1] = cy - tt[1];
  tgt[2] = tt[2] - cx;
  tgt[3] = tt[3] - cy;
  dfl = 0.f;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    tgt[s] = fminf(fmaxf(tgt[s], 0.f), (float)R - 0.1f);
    const float* x = bp + s * (R + 1);
    float mx = x[0];
    for (int j = 1; j <= R; j++) mx = fmaxf(mx, x[j]);
    float se = 0.f, sj = 0.f;
    for (int j = 0; j <= R; j++) {
      const float e = expf(x[j] - mx);
      se += e;
      sj += e * (float)j;
    }
    corner[s] = sj / se;
    lse[s] = mx + logf(se);
    const int dl = (int)tgt[s], dr = dl + 1;
    const float wl = (float)dr - tgt[s], wr = tgt[s] - (float)dl;
    dfl += (lse[s] - x[dl]) * wl + (lse[s] - x[dr]) * wr;
  }
  const float p[4] = {cx - corner[0], cy - corner[1], cx + corner[2], cy + corner[3]};
  score = hbb_iou(p, tt, 1e-6f);
  // giou_loss: 1 - giou, bbox_overlaps(mode='giou', is_aligned=True, eps=1e-7)
  const float eps = 1e-7f;
  const float a1 = (p[2] - p[0]) * (p[3] - p[1]), a2 = (tt[2] - tt[0]) * (tt[3] - tt[1]);
  const float ltx = fmaxf(p[0], tt[0]), lty = fmaxf(p[1], tt[1]), rbx = fminf(p[2], tt[2]), rby = fminf(p[3], tt[3]);
  const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
  const float ov = w * h;
  const float un = a1 + a2 - ov, unc = fmaxf(un, eps);
  const float iou = ov / unc;
  const float ex1 = fminf(p[0], tt[0]), ey1 = fminf(p[1], tt[1]), ex2 = fmaxf(p[2], tt[2]), ey2 = fmaxf(p[3], tt[3]);
  const float ew = fmaxf(ex2 - ex1, 0.f), eh = fmaxf(ey2 - ey1, 0.f);
  const float ear = ew * eh, ea = fmaxf(ear, eps);
  giou_l = 1.f - (iou - (ea - unc) / ea);
  if (GRAD) {
    // reverse mode through the expression above; upstream d(loss)/d(giou_l) = cbox
    const float g_giou = -cbox;                    // d/d giou
    const float g_iou = g_giou;
    const float g_ea = g_giou * (-(unc) / (ea * ea));
    float g_unc = g_giou * (1.f / ea) + g_iou * (-ov / (unc * unc));
    float g_ov = g_iou / unc;
    const float g_un = un >= eps ? g_unc : 0.f;    // clamp(min=eps)
    const float g_a1 = g_un;
    g_ov += -g_un;
    const float g_w = g_ov * h * (rbx - ltx >= 0.f ? 1.f : 0.f), g_h = g_ov * w * (rby - lty >= 0.f ? 1.f : 0.f);
    const float g_ear = ear >= eps ? g_ea : 0.f;
    const float g_ew = g_ear * eh * (ex2 - ex1 >= 0.f ? 1.f : 0.f), g_eh = g_ear * ew * (ey2 - ey1 >= 0.f ? 1.f : 0.f);
    // torch.max / torch.min (binary) send the gradient to the selected operand, half to each on a tie
    auto sel_max = [](float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); };  // share of a in max(a, b)
    auto sel_min = [](float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); };
    float gp[4];
    gp[0] = -g_w * sel_max(p[0], tt[0]) - g_ew * sel_min(p[0], tt[0]) - g_a1 * (p[3] - p[1]);
    gp[1] = -g_h * sel_max(p[1], tt[1]) - g_eh * sel_min(p[1], tt[1]) - g_a1 * (p[2] - p[0]);
    gp[2] = g_w * sel_min(p[2], tt[2]) + g_ew * sel_max(p[2], tt[2]) + g_a1 * (p[3] - p[1]);
    gp[3] = g_h * sel_min(p[3], tt[3]) + g_eh * sel_max(p[3], tt[3]) + g_a1 * (p[2] - p[0]);
    const float gc[4] = {-gp[0], -gp[1], gp[2], gp[3]};  // p = centre -/+ corner
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const float* x = bp + s * (R + 1);
      const int dl = (int)tgt[s], dr = dl + 1;
      const float wl = (float)dr - tgt[s], wr = tgt[s] - (float)dl;
      for (int j = 0; j <= R; j++) {
        const float pj = expf(x[j] - lse[s]);
        float d = gc[s] * pj * ((float)j - corner[s]);                      // Integral: d corner / d logit
        d += cdfl * ((wl + wr) * pj - (j == dl ? wl : 0.f) - (j == dr ? wr : 0.f));  // the two cross-entropies
        dbp[s * (R + 1) + j] = d;
      }
    }
  }
}

// Per anchor of every image: the three loss terms.  sums[term][level] (double, zeroed by the caller): 0 GIoU x weight,
// 1 DFL x weight, 2 QFL x label weight, 3 weight (= avg_factor terms); pos_count[b] (zeroed) counts the positives.
__global__ __launch_bounds__(256) void gfl_loss_fwd_kernel(GflCfg c, Levels lv, const float* __restrict__ cls,
                                                           const float* __restrict__ bbox, const float* __restrict__ anchors,
                                                           const unsigned long long* __restrict__ best,
                                                           const float* __restrict__ gts, const int64_t* __restrict__ gt_labels,
                                                           const int* __restrict__ gt_off, const uint8_t* __restrict__ valid,
                                                           float pos_weight, double* __restrict__ sums,
                                                           int* __restrict__ pos_count) {
  __shared__ float acc[4][GFL_MAX_LEVELS];
  __shared__ int npos;
  if (threadIdx.x < 4 * GFL_MAX_LEVELS) (&acc[0][0])[threadIdx.x] = 0.f;
  if (threadIdx.x == 0) npos = 0;
  __syncthreads();
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (a < c.A) {
    const int l = level_of(lv, a);
    const AnchorTarget tg = anchor_target(best, gts, gt_labels, gt_off, valid, pos_weight, b, a, c.A);
    const float* x = cls + ((long)b * c.A + a) * c.C;
    float smax = 0.f;
    for (int j = 0; j < c.C; j++) smax = fmaxf(smax, sigmoidf(x[j]));
    const float wt = tg.pos ? smax : 0.f;
    float score = 0.f, box_l = 0.f, dfl_l = 0.f;
    if (tg.pos) {
      const float* an = anchors + 4 * (long)a;
      const float st = lv.stride[l];
      const float cx = (an[0] + an[2]) / 2.0f / st, cy = (an[1] + an[3]) / 2.0f / st;
      const float tt[4] = {tg.t[0] / st, tg.t[1] / st, tg.t[2] / st, tg.t[3] / st};
      float giou_l;
      box_terms<false>(bbox + ((long)b * c.A + a) * 4 * (c.R + 1), c.R, cx, cy, tt, score, giou_l, dfl_l, 0.f, 0.f, nullptr);
      box_l = giou_l * wt;
      dfl_l *= wt;
    }
    float qfl = 0.f;
    for (int j = 0; j < c.C; j++) {
      const float sg = sigmoidf(x[j]);
      if (tg.pos && j == tg.label) qfl += bce_logits(x[j], score) * powf(fabsf(score - sg), c.beta);
      else qfl += bce_logits(x[j], 0.f) * powf(sg, c.beta);
    }
    atomicAdd(&acc[0][l], box_l);
    atomicAdd(&acc[1][l], dfl_l);
    atomicAdd(&acc[2][l], qfl * tg.lw);
    atomicAdd(&acc[3][l], wt);
    if (tg.pos) atomicAdd(&npos, 1);
  }
  __syncthreads();
  if (threadIdx.x < 4 * GFL_MAX_LEVELS) {
    const int t = threadIdx.x / GFL_MAX_LEVELS, l = threadIdx.x % GFL_MAX_LEVELS;
    if (l < lv.n && acc[t][l] != 0.f) atomicAdd(sums + t * GFL_MAX_LEVELS + l, (double)acc[t][l]);
  }
  if (threadIdx.x == 0 && npos) atomicAdd(pos_count + b, npos);
}

// Gradients of sum_l (g_box[l] * box_sum[l] + g_dfl[l] * dfl_sum[l] + g_cls[l] * cls_sum[l]) w.r.t. the logits; the three
// coefficient vectors (upstream gradient x loss weight / normaliser) arrive per level.  weight_targets and the quality score
// are detached in mmdet (computed under no-grad / .detach()), so they are constants here.
__global__ __launch_bounds__(256) void gfl_loss_bwd_kernel(GflCfg c, Levels lv, const float* __restrict__ cls,
                                                           const float* __restrict__ bbox, const float* __restrict__ anchors,
                                                           const unsigned long long* __restrict__ best,
                                                           const float* __restrict__ gts, const int64_t* __restrict__ gt_labels,
                                                           const int* __restrict__ gt_off, const uint8_t* __restrict__ valid,
                                                           float pos_weight, const float* __restrict__ coef,
                                                           float* __restrict__ dcls, float* __restrict__ dbbox) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (a >= c.A) return;
  const int l = level_of(lv, a);
  const float cbox = coef[l], cdfl = coef[GFL_MAX_LEVELS + l], ccls = coef[2 * GFL_MAX_LEVELS + l];
  const AnchorTarget tg = anchor_target(best, gts, gt_labels, gt_off, valid, pos_weight, b, a, c.A);
  const float* x = cls + ((long)b * c.A + a) * c.C;
  float* dx = dcls + ((long)b * c.A + a) * c.C;
  float* db = dbbox + ((long)b * c.A + a) * 4 * (c.R + 1);
  float score = 0.f;
  if (tg.pos) {
    float smax = 0.f;
    for (int j = 0; j < c.C; j++) smax = fmaxf(smax, sigmoidf(x[j]));
    const float* an = anchors + 4 * (long)a;
    const float st = lv.stride[l];
    const float cx = (an[0] + an[2]) / 2.0f / st, cy = (an[1] + an[3]) / 2.0f / st;
    const float tt[4] = {tg.t[0] / st, tg.t[1] / st, tg.t[2] / st, tg.t[3] / st};
    float giou_l, dfl_l;
    box_terms<true>(bbox + ((long)b * c.A + a) * 4 * (c.R + 1), c.R, cx, cy, tt, score, giou_l, dfl_l, cbox * smax,
                    cdfl * smax, db);
  } else {
    for (int j = 0; j < 4 * (c.R + 1); j++) db[j] = 0.f;
  }
  const float kc = ccls * tg.lw;
  for (int j = 0; j < c.C; j++) {
    const float xv = x[j], sg = sigmoidf(xv);
    float d;
    if (tg.pos && j == tg.label) {
      // BCE(x, t) |t - s|^beta, t constant: (s - t) m^beta + BCE beta m^(beta - 1) sign(s - t) s (1 - s)
      const float m = fabsf(score - sg);
      const float sgn = sg > score ? 1.f : (sg < score ? -1.f : 0.f);
      d = (sg - score) * powf(m, c.beta) + bce_logits(xv, score) * c.beta * powf(m, c.beta - 1.f) * sgn * sg * (1.f - sg);
    } else {
      // softplus(x) s^beta: s^(beta + 1) + softplus(x) beta s^beta (1 - s)
      d = powf(sg, c.beta + 1.f) + bce_logits(xv, 0.f) * c.beta * powf(sg, c.beta) * (1.f - sg);
    }
    dx[j] = kc * d;
  }
}

}  // namespace

extern "C" {

// best: A uint64 (zeroed here).  One launch per image (its own gts / valid flags), anchors shared.
int sm3_atss_assign(const float* anchors, int A, const int* level_off, const float* level_stride, int num_levels,
                    const float* gts, int k, const uint8_t* valid, int topk, void* best, sm3_stream_t stream) {
  if (!anchors || !level_off || !best || A < 0 || num_levels <= 0 || num_levels > GFL_MAX_LEVELS || topk <= 0 ||
      topk > GFL_MAX_TOPK || k < 0 || (k > 0 && !gts))
    return SM3_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  sm3_zero_async(best, (size_t)A * 8, st);
  if (k == 0 || A == 0) return launch_status();
  Levels lv;
  lv.n = num_levels;
  for (int i = 0; i <= GFL_MAX_LEVELS; i++) lv.off[i] = i <= num_levels ? level_off[i] : level_off[num_levels];
  for (int i = 0; i < GFL_MAX_LEVELS; i++) lv.stride[i] = (level_stride && i < num_levels) ? level_stride[i] : 1.f;
  if (lv.off[0] != 0 || lv.off[num_levels] != A) return SM3_ERR_INVALID_ARG;
  atss_assign_kernel<<<k, 64 * GFL_MAX_LEVELS, 0, st>>>(anchors, lv, gts, k, valid, topk, (unsigned long long*)best);
  return launch_status();
}

int sm3_atss_decode(const void* best, int A, int64_t* gt_inds, float* max_overlaps, sm3_stream_t stream) {
  if (!best || !gt_inds || A < 0) return SM3_ERR_INVALID_ARG;
  if (A == 0) return SM3_OK;
  atss_decode_kernel<<<(A + 255) / 256, 256, 0, (hipStream_t)stream>>>((const unsigned long long*)best, A, gt_inds,
                                                                       max_overlaps);
  return launch_status();
}

static int fill_levels_gfl(Levels& lv, const int* level_off, const float* level_stride, int num_levels, int A) {
  if (!level_off || !level_stride || num_levels <= 0 || num_levels > GFL_MAX_LEVELS) return SM3_ERR_INVALID_ARG;
  lv.n = num_levels;
  for (int i = 0; i <= GFL_MAX_LEVELS; i++) lv.off[i] = i <= num_levels ? level_off[i] : level_off[num_levels];
  for (int i = 0; i < GFL_MAX_LEVELS; i++) lv.stride[i] = i < num_levels ? level_stride[i] : 1.f;
  if (lv.off[0] != 0 || lv.off[num_levels] != A) return SM3_ERR_INVALID_ARG;
  return SM3_OK;
}

// sums: 4 x 8 doubles, pos_count: B int32 -- both zeroed here.  best: (B, A) keys of sm3_atss_assign; gts / gt_labels: the
// images' ground truth concatenated, gt_off[b] = first gt of image b (device int32[B]).
int sm3_gfl_loss_fwd(const float* cls, const float* bbox, const float* anchors, int B, int A, int C, int reg_max,
                     const int* level_off, const float* level_stride, int num_levels, const void* best, const float* gts,
                     const int64_t* gt_labels, const int* gt_off, const uint8_t* valid, float pos_weight, float beta,
                     double* sums, int* pos_count, sm3_stream_t stream) {
  if (!cls || !bbox || !anchors || !best || !sums || !pos_count || B <= 0 || A <= 0 || C <= 0 || reg_max <= 0 || reg_max > 31)
    return SM3_ERR_INVALID_ARG;
  Levels lv;
  const int rc = fill_levels_gfl(lv, level_off, level_stride, num_levels, A);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  sm3_zero_async(sums, sizeof(double) * 4 * GFL_MAX_LEVELS, st);
  sm3_zero_async(pos_count, sizeof(int) * (size_t)B, st);
  GflCfg c{B, A, C, reg_max, beta};
  dim3 grid((A + 255) / 256, B);
  gfl_loss_fwd_kernel<<<grid, 256, 0, st>>>(c, lv, cls, bbox, anchors, (const unsigned long long*)best, gts, gt_labels,
                                            gt_off, valid, pos_weight, sums, pos_count);
  return launch_status();
}

// coef: 3 x 8 floats on the device (GIoU, DFL, QFL coefficient per level).  dcls (B, A, C), dbbox (B, A, 4 (reg_max + 1)).
int sm3_gfl_loss_bwd(const float* cls, const float* bbox, const float* anchors, int B, int A, int C, int reg_max,
                     const int* level_off, const float* level_stride, int num_levels, const void* best, const float* gts,
                     const int64_t* gt_labels, const int* gt_off, const uint8_t* valid, float pos_weight, float beta,
                     const float* coef, float* dcls, float* dbbox, sm3_stream_t stream) {
  if (!cls || !bbox || !anchors || !best || !coef || !dcls || !dbbox || B <= 0 || A <= 0 || C <= 0 || reg_max <= 0 ||
      reg_max > 31)
    return SM3_ERR_INVALID_ARG;
  Levels lv;
  const int rc = fill_levels_gfl(lv, level_off, level_stride, num_levels, A);
  if (rc) return rc;
  GflCfg c{B, A, C, reg_max, beta};
  dim3 grid((A + 255) / 256, B);
  gfl_loss_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(c, lv, cls, bbox, anchors, (const unsigned long long*)best, gts,
                                                             gt_labels, gt_off, valid, pos_weight, coef, dcls, dbbox);
  return launch_status();
}

}  // extern "C"
