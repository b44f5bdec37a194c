/*
 * oracle/ops_oracle.c -- CPU restatement (plain C99) of the reference's rotated-detection
 * operators.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg as the checker; never by the product path (sm3det_amd/).
 *
 * Every function cites the reference file:line it follows.  Paths are relative to
 * /root/reference/mmcv/mmcv/ops/csrc/.  Parity pinned: tests/test_oracle_ops.py checks this file
 * against (a) the mmcv golden vectors in mmcv/tests/test_ops/test_{box_iou_rotated,nms_rotated,
 * nms,roi_align_rotated}.py and (b) bit-exactly against oracle/_ref (the reference's own C++
 * compiled here by oracle/build_ref.py) on seeded random + degenerate inputs.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC ops_oracle.c -o libops_oracle.so -lm
 * (-ffp-contract=off: the reference is built for baseline x86-64, i.e. no FMA contraction;
 *  float arithmetic below must round exactly like it.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } pt_t;

static inline float dot2(pt_t a, pt_t b) { return a.x * b.x + a.y * b.y; }     /* utils.hpp:46-49 */
static inline float cross2(pt_t a, pt_t b) { return a.x * b.y - b.x * a.y; }   /* utils.hpp:51-54 */
static inline pt_t psub(pt_t a, pt_t b) { pt_t r = {a.x - b.x, a.y - b.y}; return r; }

/* common/box_iou_rotated_utils.hpp:56-75 get_rotated_vertices<float> */
static void rotated_vertices(float xc, float yc, float w, float h, float a, pt_t p[4]) {
  double theta = a;
  float c2 = (float)cos(theta) * 0.5f;
  float s2 = (float)sin(theta) * 0.5f;
  p[0].x = xc - s2 * h - c2 * w;
  p[0].y = yc + c2 * h - s2 * w;
  p[1].x = xc + s2 * h - c2 * w;
  p[1].y = yc - c2 * h - s2 * w;
  p[2].x = 2 * xc - p[0].x;
  p[2].y = 2 * yc - p[0].y;
  p[3].x = 2 * xc - p[1].x;
  p[3].y = 2 * yc - p[1].y;
}

/* utils.hpp:77-155 get_intersection_points<float> */
static int intersection_points(const pt_t p1[4], const pt_t p2[4], pt_t out[24]) {
  pt_t v1[4], v2[4];
  for (int i = 0; i < 4; i++) {
    v1[i] = psub(p1[(i + 1) % 4], p1[i]);
    v2[i] = psub(p2[(i + 1) % 4], p2[i]);
  }
  int num = 0;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      float det = cross2(v2[j], v1[i]);
      if (fabs(det) <= 1e-14) continue;          /* double compare, utils.hpp:96 */
      pt_t v12 = psub(p2[j], p1[i]);
      float t1 = cross2(v2[j], v12) / det;
      float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        out[num].x = p1[i].x + v1[i].x * t1;
        out[num].y = p1[i].y + v1[i].y * t1;
        num++;
      }
    }
  }
  { /* vertices of rect1 inside rect2, utils.hpp:112-132 */
    pt_t AB = v2[0], DA = v2[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; i++) {
      pt_t AP = psub(p1[i], p2[0]);
      float APdotAB = dot2(AP, AB);
      float APdotAD = -dot2(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        out[num++] = p1[i];
    }
  }
  { /* reverse check, utils.hpp:134-152 */
    pt_t AB = v1[0], DA = v1[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
    for (int i = 0; i < 4; i++) {
      pt_t AP = psub(p2[i], p1[0]);
      float APdotAB = dot2(AP, AB);
      float APdotAD = -dot2(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        out[num++] = p2[i];
    }
  }
  return num;
}

/* comparator of the CPU hull sort, utils.hpp:214-222 (NOT a strict weak order: 1e-6 tie band) */
static inline int hull_less(pt_t A, pt_t B) {
  float t = cross2(A, B);
  if (fabs(t) < 1e-6) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

/* libstdc++ std::sort restated step by step (bits/stl_algo.h: __sort -> __introsort_loop with
 * _S_threshold = 16 -> __final_insertion_sort).  The comparator above is not a strict weak order,
 * so the ALGORITHM decides the result; this follows GCC's exactly (SURVEY.md section 7 "hard parts"). */
static void ins_unguarded(pt_t* first, pt_t* i) {          /* __unguarded_linear_insert */
  (void)first;
  pt_t val = *i;
  pt_t* next = i - 1;
  while (hull_less(val, *next)) { *i = *next; i = next; --next; }
  *i = val;
}
static void ins_sort(pt_t* first, pt_t* last) {             /* __insertion_sort */
  if (first == last) return;
  for (pt_t* i = first + 1; i != last; ++i) {
    if (hull_less(*i, *first)) {
      pt_t val = *i;
      memmove(first + 1, first, (size_t)(i - first) * sizeof(pt_t));
      *first = val;
    } else {
      ins_unguarded(first, i);
    }
  }
}
static inline void pswap(pt_t* a, pt_t* b) { pt_t t = *a; *a = *b; *b = t; }
static void gcc_std_sort(pt_t* first, pt_t* last) {
  if (first == last) return;
  if (last - first > 16) {
    /* __introsort_loop leaves partitions of <= 16 unsorted; replicate without the inner final sorts */
    pt_t* stack_lo[8]; pt_t* stack_hi[8]; int sp = 0;
    pt_t* lo = first; pt_t* hi = last;
    for (;;) {
      while (hi - lo > 16) {
        pt_t* mid = lo + (hi - lo) / 2;
        pt_t *a = lo + 1, *b = mid, *c = hi - 1;
        if (hull_less(*a, *b)) {
          if (hull_less(*b, *c)) pswap(lo, b);
          else if (hull_less(*a, *c)) pswap(lo, c);
          else pswap(lo, a);
        } else if (hull_less(*a, *c)) pswap(lo, a);
        else if (hull_less(*b, *c)) pswap(lo, c);
        else pswap(lo, b);
        pt_t* f = lo + 1; pt_t* l = hi;
        for (;;) {
          while (hull_less(*f, *lo)) ++f;
          --l;
          while (hull_less(*lo, *l)) --l;
          if (!(f < l)) break;
          pswap(f, l);
          ++f;
        }
        /* libstdc++: __introsort_loop(cut, last) first (recursion), then last = cut */
        stack_lo[sp] = lo; stack_hi[sp] = f; sp++;   /* remember the left part for later */
        lo = f;                                      /* descend into the right part now  */
      }
      if (sp == 0) break;
      sp--; lo = stack_lo[sp]; hi = stack_hi[sp];
    }
    /* __final_insertion_sort for n > 16 */
    ins_sort(first, first + 16);
    for (pt_t* i = first + 16; i != last; ++i) ins_unguarded(first, i);
  } else {
    ins_sort(first, last);
  }
}

/* utils.hpp:157-272 convex_hull_graham<float>(p, num_in, q, shift_to_zero=true), CPU branch */
static int convex_hull(const pt_t p[24], int num_in, pt_t q[24]) {
  int t = 0;
  for (int i = 1; i < num_in; i++)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  pt_t start = p[t];
  for (int i = 0; i < num_in; i++) q[i] = psub(p[i], start);
  pt_t tmp = q[0]; q[0] = q[t]; q[t] = tmp;
  float dist[24];
  gcc_std_sort(q + 1, q + num_in);
  for (int i = 0; i < num_in; i++) dist[i] = dot2(q[i], q[i]);
  int k;
  for (k = 1; k < num_in; k++)
    if (dist[k] > 1e-8) break;
  if (k == num_in) { q[0] = p[t]; return 1; }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num_in; i++) {
    while (m > 1 && cross2(psub(q[i], q[m - 2]), psub(q[m - 1], q[m - 2])) >= 0) m--;
    q[m++] = q[i];
  }
  return m;
}

/* utils.hpp:285-297 polygon_area<float> */
static float polygon_area(const pt_t q[24], int m) {
  if (m <= 2) return 0;
  float area = 0;
  for (int i = 1; i < m - 1; i++)
    area += fabs(cross2(psub(q[i], q[0]), psub(q[i + 1], q[0])));   /* fabs(double) -> float += */
  return area / 2.0;
}

/* utils.hpp:344-378 single_box_iou_rotated<float> (+ rotated_boxes_intersection 299-321) */
float oracle_single_box_iou_rotated(const float* b1, const float* b2, int mode_flag) {
  double csx = (b1[0] + b2[0]) / 2.0;
  double csy = (b1[1] + b2[1]) / 2.0;
  float x1 = b1[0] - csx, y1 = b1[1] - csy, w1 = b1[2], h1 = b1[3], a1 = b1[4];
  float x2 = b2[0] - csx, y2 = b2[1] - csy, w2 = b2[2], h2 = b2[3], a2 = b2[4];
  const float area1 = w1 * h1;
  const float area2 = w2 * h2;
  if (area1 < 1e-14 || area2 < 1e-14) return 0.f;
  pt_t inter[24], ordered[24], p1[4], p2[4];
  rotated_vertices(x1, y1, w1, h1, a1, p1);
  rotated_vertices(x2, y2, w2, h2, a2, p2);
  int num = intersection_points(p1, p2, inter);
  float intersection;
  if (num <= 2) intersection = 0.0;
  else {
    int nc = convex_hull(inter, num, ordered);
    intersection = polygon_area(ordered, nc);
  }
  float baseS = 1.0;
  if (mode_flag == 0) baseS = (area1 + area2 - intersection);
  else if (mode_flag == 1) baseS = area1;
  return intersection / baseS;
}

/* pytorch/cpu/box_iou_rotated.cpp:8-30 box_iou_rotated_cpu_kernel<float> */
void oracle_box_iou_rotated(const float* boxes1, int n, const float* boxes2, int m, float* ious,
                            int mode_flag, int aligned) {
  if (aligned) {
    for (int i = 0; i < n; i++)
      ious[i] = oracle_single_box_iou_rotated(boxes1 + 5 * i, boxes2 + 5 * i, mode_flag);
  } else {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < m; j++)
        ious[(size_t)i * m + j] =
            oracle_single_box_iou_rotated(boxes1 + 5 * i, boxes2 + 5 * j, mode_flag);
  }
}

/* stable descending argsort (score desc, index asc) == torch CPU sort(descending=True) on tie-free
 * data; ties are resolved by original index (documented deviation: torch does not promise it). */
typedef struct { float s; int64_t i; } si_t;
static int cmp_si(const void* a, const void* b) {
  const si_t* x = (const si_t*)a; const si_t* y = (const si_t*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->i > y->i) - (x->i < y->i);
}
static int64_t* argsort_desc(const float* scores, int64_t n) {
  si_t* t = (si_t*)malloc(sizeof(si_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) { t[i].s = scores[i]; t[i].i = i; }
  qsort(t, (size_t)n, sizeof(si_t), cmp_si);
  int64_t* o = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) o[i] = t[i].i;
  free(t);
  return o;
}

/* pytorch/cpu/nms_rotated.cpp:7-57 nms_rotated_cpu_kernel<float>: suppress when iou >= thr (line 51),
 * label column (if any) ignored; dets row stride = `stride` floats (5 or 6). Returns #kept. */
int64_t oracle_nms_rotated(const float* dets, int stride, const float* scores, int64_t n,
                           float iou_threshold, int64_t* keep) {
  if (n == 0) return 0;
  int64_t* order = argsort_desc(scores, n);
  uint8_t* sup = (uint8_t*)calloc((size_t)n, 1);
  int64_t nk = 0;
  for (int64_t _i = 0; _i < n; _i++) {
    int64_t i = order[_i];
    if (sup[i] == 1) continue;
    keep[nk++] = i;
    for (int64_t _j = _i + 1; _j < n; _j++) {
      int64_t j = order[_j];
      if (sup[j] == 1) continue;
      float ovr = oracle_single_box_iou_rotated(dets + i * stride, dets + j * stride, 0);
      if (ovr >= iou_threshold) sup[j] = 1;
    }
  }
  free(order); free(sup);
  return nk;
}

/* pytorch/cpu/nms.cpp:5-54 nms_cpu: ovr = inter / (iarea + area_j - inter) > thr, offset in {0,1} */
int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float iou_threshold,
                   int offset, int64_t* keep) {
  if (n == 0) return 0;
  int64_t* order = argsort_desc(scores, n);
  float* areas = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t i = 0; i < n; i++)
    areas[i] = (boxes[4 * i + 2] - boxes[4 * i + 0] + offset) * (boxes[4 * i + 3] - boxes[4 * i + 1] + offset);
  uint8_t* sel = (uint8_t*)malloc((size_t)n);
  memset(sel, 1, (size_t)n);
  for (int64_t _i = 0; _i < n; _i++) {
    if (!sel[_i]) continue;
    int64_t i = order[_i];
    float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
    float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < n; _j++) {
      if (!sel[_j]) continue;
      int64_t j = order[_j];
      float xx1 = fmaxf(ix1, boxes[4 * j]);
      float yy1 = fmaxf(iy1, boxes[4 * j + 1]);
      float xx2 = fminf(ix2, boxes[4 * j + 2]);
      float yy2 = fminf(iy2, boxes[4 * j + 3]);
      float w = fmaxf(0.f, xx2 - xx1 + offset);
      float h = fmaxf(0.f, yy2 - yy1 + offset);
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr > iou_threshold) sel[_j] = 0;
    }
  }
  int64_t nk = 0;
  for (int64_t _i = 0; _i < n; _i++)
    if (sel[_i]) keep[nk++] = order[_i];
  free(order); free(areas); free(sel);
  return nk;
}

/* bilinear sample set-up shared by fwd (pre_calc_for_bilinear_interpolate, cpu/roi_align_rotated.cpp:24-113)
 * and bwd (bilinear_interpolate_gradient, :215-263). Returns 0 when the sample is outside. */
static int bilinear_setup(int height, int width, float y, float x, int* yl, int* xl, int* yh, int* xh,
                          float* w1, float* w2, float* w3, float* w4) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) return 0;
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else y_high = y_low + 1;
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else x_high = x_low + 1;
  float ly = y - y_low, lx = x - x_low;
  float hy = 1. - ly, hx = 1. - lx;
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
  *yl = y_low; *xl = x_low; *yh = y_high; *xh = x_high;
  return 1;
}

typedef struct {
  float cw, ch, rw, rh, cosv, sinv, bin_h, bin_w, start_h, start_w;
  int grid_h, grid_w, batch;
} roi_geom_t;

/* per-RoI geometry, cpu/roi_align_rotated.cpp:129-172 (fwd) == :283-330 (bwd) */
static void roi_geometry(const float* roi, float spatial_scale, int aligned, int clockwise, int ph, int pw,
                         int sampling_ratio, roi_geom_t* g) {
  g->batch = (int)roi[0];
  float offset = aligned ? 0.5f : 0.0f;
  g->cw = roi[1] * spatial_scale - offset;
  g->ch = roi[2] * spatial_scale - offset;
  g->rw = roi[3] * spatial_scale;
  g->rh = roi[4] * spatial_scale;
  float theta = roi[5];
  if (clockwise) theta = -theta;
  g->cosv = cos(theta);   /* `T cos_theta = cos(theta)`: unqualified ::cos(double) (only <cmath> is included), result narrowed to float */
  g->sinv = sin(theta);
  if (!aligned) { g->rw = fmaxf(g->rw, 1.f); g->rh = fmaxf(g->rh, 1.f); }
  g->bin_h = g->rh / (float)ph;
  g->bin_w = g->rw / (float)pw;
  g->grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(g->rh / ph);
  g->grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(g->rw / pw);
  g->start_h = -g->rh / 2.0;
  g->start_w = -g->rw / 2.0;
}

/* cpu/roi_align_rotated.cpp:115-212 ROIAlignRotatedForward<float>; input NCHW, rois (n,6), output (n,C,ph,pw) */
void oracle_roi_align_rotated_forward(const float* input, const float* rois, float* output, int n_rois,
                                      int channels, int height, int width, int ph_n, int pw_n,
                                      float spatial_scale, int sampling_ratio, int aligned, int clockwise) {
  for (int n = 0; n < n_rois; n++) {
    roi_geom_t g;
    roi_geometry(rois + 6 * n, spatial_scale, aligned, clockwise, ph_n, pw_n, sampling_ratio, &g);
    int cnt_i = g.grid_h * g.grid_w; if (cnt_i < 1) cnt_i = 1;
    const float count = (float)cnt_i;
    for (int c = 0; c < channels; c++) {
      const float* in = input + ((size_t)g.batch * channels + c) * height * width;
      for (int ph = 0; ph < ph_n; ph++) {
        for (int pw = 0; pw < pw_n; pw++) {
          float val = 0.;
          for (int iy = 0; iy < g.grid_h; iy++) {
            const float yy = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ix++) {
              const float xx = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
              float y = yy * g.cosv - xx * g.sinv + g.ch;
              float x = yy * g.sinv + xx * g.cosv + g.cw;
              int yl, xl, yh, xh; float w1, w2, w3, w4;
              if (!bilinear_setup(height, width, y, x, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4)) {
                /* reference adds 0*in[0] four times (:52-66, :189-192): a float no-op unless in[0] is inf/nan */
                val += 0.f * in[0] + 0.f * in[0] + 0.f * in[0] + 0.f * in[0];
                continue;
              }
              val += w1 * in[yl * width + xl] + w2 * in[yl * width + xh] + w3 * in[yh * width + xl] +
                     w4 * in[yh * width + xh];
            }
          }
          val /= count;
          output[(((size_t)n * channels + c) * ph_n + ph) * pw_n + pw] = val;
        }
      }
    }
  }
}

/* cpu/roi_align_rotated.cpp:272-372 ROIAlignRotatedBackward<float> (grad_output contiguous) */
void oracle_roi_align_rotated_backward(const float* grad_output, const float* rois, float* grad_input,
                                       int n_rois, int channels, int height, int width, int ph_n, int pw_n,
                                       float spatial_scale, int sampling_ratio, int aligned, int clockwise) {
  for (int n = 0; n < n_rois; n++) {
    roi_geom_t g;
    roi_geometry(rois + 6 * n, spatial_scale, aligned, clockwise, ph_n, pw_n, sampling_ratio, &g);
    const float count = (float)(g.grid_h * g.grid_w);
    for (int c = 0; c < channels; c++) {
      float* gin = grad_input + ((size_t)g.batch * channels + c) * height * width;
      for (int ph = 0; ph < ph_n; ph++) {
        for (int pw = 0; pw < pw_n; pw++) {
          const float go = grad_output[(((size_t)n * channels + c) * ph_n + ph) * pw_n + pw];
          for (int iy = 0; iy < g.grid_h; iy++) {
            const float yy = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
            for (int ix = 0; ix < g.grid_w; ix++) {
              const float xx = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
              float y = yy * g.cosv - xx * g.sinv + g.ch;
              float x = yy * g.sinv + xx * g.cosv + g.cw;
              int yl, xl, yh, xh; float w1, w2, w3, w4;
              if (!bilinear_setup(height, width, y, x, &yl, &xl, &yh, &xh, &w1, &w2, &w3, &w4)) continue;
              gin[yl * width + xl] += go * w1 / count;
              gin[yl * width + xh] += go * w2 / count;
              gin[yh * width + xl] += go * w3 / count;
              gin[yh * width + xh] += go * w4 / count;
            }
          }
        }
      }
    }
  }
}
