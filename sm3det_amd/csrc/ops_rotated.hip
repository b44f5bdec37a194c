// ops_rotated.hip -- rotated-detection operators for gfx950 (MI355X), wave64.
//
//   sm3_box_iou_rotated, sm3_nms, sm3_nms_rotated, sm3_argsort_desc_f32,
//   sm3_roi_align_rotated_{forward,backward}
//
// Built with -ffp-contract=off: the parity target is the reference's CPU path (baseline x86-64, no FMA),
// and NMS keep-lists must be bit-exact, so every float op here rounds exactly like the scalar C++ does.
// Semantics follow (paths relative to /root/reference/mmcv/mmcv/ops/csrc):
//   common/box_iou_rotated_utils.hpp, pytorch/cpu/{box_iou_rotated,nms_rotated,nms,roi_align_rotated}.cpp
// The structure is NOT the reference's CUDA: 64x64 suppression tiles are one wavefront each, bitmasks are
// built with 64-bit ballots-free per-lane words, and the greedy sweep runs on the device (no D2H of the mask).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/sm3det_hip.h"
#include <stdlib.h>

#include "common.h"

namespace {

struct Pt {
  float x, y;
};
__device__ __forceinline__ float dot2(Pt a, Pt b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float cross2(Pt a, Pt b) { return a.x * b.y - b.x * a.y; }
__device__ __forceinline__ Pt psub(Pt a, Pt b) { return Pt{a.x - b.x, a.y - b.y}; }

// box_iou_rotated_utils.hpp:56-75
__device__ __forceinline__ void rotated_vertices(float xc, float yc, float w, float h, float a, Pt* p) {
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

// box_iou_rotated_utils.hpp:214-222 (CPU comparator; 1e-6 tie band, not a strict weak order)
__device__ __forceinline__ bool hull_less(Pt A, Pt B) {
  float t = cross2(A, B);
  if (fabs((double)t) < 1e-6) return dot2(A, A) < dot2(B, B);
  return t > 0;
}

// ---- per-lane scratch of the polygon clipping, in LDS --------------------------------------------------------------------
// The reference algorithm indexes two 24-point arrays and an 8-entry introsort stack with run-time indices.  As plain local
// arrays those live in private (scratch) memory: 528 bytes per lane, HBM-backed, in every kernel that computes a rotated IoU
// (round 5; it also was the one thing the kernels that faulted under two processes sharing a GPU had in common).  Here they are
// a slice of LDS: one 24-point array per lane (the hull candidates are shifted, sorted and scanned IN PLACE: q[i] is only
// written at positions <= the one being read) and the sort's stack as 8 packed (lo, hi) index pairs.  Element e of lane l sits
// at word (e * 64 + l): a wave instruction in which every lane reads a different element still touches 64 distinct 8-byte
// slots, one per lane column -- conflict-free for ds_read_b64 / ds_write_b64.  No private segment is left.
constexpr int IOU_PTS = 24;
constexpr int IOU_STACK = 8;
constexpr int IOU_LDS_WORDS_PER_WAVE = IOU_PTS * 64 * 2 + IOU_STACK * 64;  // 32-bit words: 14 336 B per wave
struct IouScratch {
  Pt* pts;          // this lane's column of the wave's [IOU_PTS][64] point array
  uint32_t* stack;  // this lane's column of the wave's [IOU_STACK][64] array
  __device__ __forceinline__ Pt get(int i) const { return pts[i * 64]; }
  __device__ __forceinline__ void set(int i, Pt v) const { pts[i * 64] = v; }
};
__device__ __forceinline__ IouScratch iou_scratch(uint32_t* wave_words, int lane) {
  return IouScratch{reinterpret_cast<Pt*>(wave_words) + lane, wave_words + IOU_PTS * 64 * 2 + lane};
}

__device__ __forceinline__ void ins_unguarded(const IouScratch& q, int i) {  // libstdc++ __unguarded_linear_insert
  const Pt val = q.get(i);
  int next = i - 1;
  while (hull_less(val, q.get(next))) {
    q.set(i, q.get(next));
    i = next;
    --next;
  }
  q.set(i, val);
}
__device__ __forceinline__ void ins_sort(const IouScratch& q, int first, int last) {  // libstdc++ __insertion_sort
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (hull_less(q.get(i), q.get(first))) {
      const Pt val = q.get(i);
      for (int j = i; j != first; --j) q.set(j, q.get(j - 1));
      q.set(first, val);
    } else {
      ins_unguarded(q, i);
    }
  }
}
__device__ __forceinline__ void pswap(const IouScratch& q, int a, int b) {
  const Pt t = q.get(a);
  q.set(a, q.get(b));
  q.set(b, t);
}
// libstdc++ std::sort (introsort, threshold 16, then final insertion sort) -- the CPU reference sorts the
// hull candidates with it (utils.hpp:213); because the comparator is not a strict weak order the algorithm
// itself must be reproduced for bit-exact areas.  n <= 23 here, so the depth limit is never reached.
__device__ void gcc_std_sort(const IouScratch& q, int first, int last) {
  if (last - first > 16) {
    int sp = 0;
    int lo = first;
    int hi = last;
    for (;;) {
      while (hi - lo > 16) {
        const int mid = lo + (hi - lo) / 2;
        const int a = lo + 1, b = mid, c = hi - 1;
        if (hull_less(q.get(a), q.get(b))) {
          if (hull_less(q.get(b), q.get(c))) pswap(q, lo, b);
          else if (hull_less(q.get(a), q.get(c))) pswap(q, lo, c);
          else pswap(q, lo, a);
        } else if (hull_less(q.get(a), q.get(c))) pswap(q, lo, a);
        else if (hull_less(q.get(b), q.get(c))) pswap(q, lo, c);
        else pswap(q, lo, b);
        int f = lo + 1;
        int l = hi;
        const Pt pivot = q.get(lo);  // (the partition never moves *lo)
        for (;;) {
          while (hull_less(q.get(f), pivot)) ++f;
          --l;
          while (hull_less(pivot, q.get(l))) --l;
          if (!(f < l)) break;
          pswap(q, f, l);
          ++f;
        }
        q.stack[sp * 64] = (uint32_t)lo | ((uint32_t)f << 8);
        sp++;
        lo = f;
      }
      if (sp == 0) break;
      sp--;
      const uint32_t e = q.stack[sp * 64];
      lo = (int)(e & 255u);
      hi = (int)(e >> 8);
    }
    ins_sort(q, first, first + 16);
    for (int i = first + 16; i != last; ++i) ins_unguarded(q, i);
  } else {
    ins_sort(q, first, last);
  }
}

// single_box_iou_rotated<float>, box_iou_rotated_utils.hpp:344-378 with
// get_intersection_points :77-155, convex_hull_graham (CPU branch) :157-272, polygon_area :285-297.
// `q`: this lane's LDS scratch (iou_scratch); every lane of a wave that calls this needs its own column.
struct Box5 {  // (cx, cy, w, h, angle) by value: no pointer to a local array, so nothing is forced into private memory
  float x, y, w, h, a;
};
__device__ __forceinline__ Box5 load_box(const float* __restrict__ b) { return Box5{b[0], b[1], b[2], b[3], b[4]}; }
__device__ float single_box_iou_rotated(const Box5 b1, const Box5 b2, int mode_flag, const IouScratch& q) {
  double csx = (b1.x + b2.x) / 2.0;
  double csy = (b1.y + b2.y) / 2.0;
  float x1 = b1.x - csx, y1 = b1.y - csy, w1 = b1.w, h1 = b1.h, a1 = b1.a;
  float x2 = b2.x - csx, y2 = b2.y - csy, w2 = b2.w, h2 = b2.h, a2 = b2.a;
  const float area1 = w1 * h1;
  const float area2 = w2 * h2;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;

  Pt p1[4], p2[4], v1[4], v2[4];
  rotated_vertices(x1, y1, w1, h1, a1, p1);
  rotated_vertices(x2, y2, w2, h2, a2, p2);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    v1[i] = psub(p1[(i + 1) & 3], p1[i]);
    v2[i] = psub(p2[(i + 1) & 3], p2[i]);
  }
  int num = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float det = cross2(v2[j], v1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      Pt v12 = psub(p2[j], p1[i]);
      float t1 = cross2(v2[j], v12) / det;
      float t2 = cross2(v1[i], v12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        q.set(num, Pt{p1[i].x + v1[i].x * t1, p1[i].y + v1[i].y * t1});
        num++;
      }
    }
  }
  {
    Pt AB = v2[0], DA = v2[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Pt AP = psub(p1[i], p2[0]);
      float APdotAB = dot2(AP, AB);
      float APdotAD = -dot2(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        q.set(num++, p1[i]);
    }
  }
  {
    Pt AB = v1[0], DA = v1[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Pt AP = psub(p2[i], p1[0]);
      float APdotAB = dot2(AP, AB);
      float APdotAD = -dot2(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        q.set(num++, p2[i]);
    }
  }
  float intersection = 0.f;
  if (num > 2) {
    // convex hull (Graham scan), shift_to_zero = true
    int t = 0;
    Pt best = q.get(0);
    for (int i = 1; i < num; i++) {
      const Pt c = q.get(i);
      if (c.y < best.y || (c.y == best.y && c.x < best.x)) {
        t = i;
        best = c;
      }
    }
    const Pt start = best;
    for (int i = 0; i < num; i++) q.set(i, psub(q.get(i), start));
    pswap(q, 0, t);
    gcc_std_sort(q, 1, num);
    int k;
    for (k = 1; k < num; k++) {
      const Pt c = q.get(k);
      if ((double)dot2(c, c) > 1e-8) break;
    }
    int m;
    if (k == num) {
      m = 1;
    } else {
      q.set(1, q.get(k));
      m = 2;
      for (int i = k + 1; i < num; i++) {
        const Pt qi = q.get(i);
        while (m > 1) {
          const Pt a = q.get(m - 2);
          if (!(cross2(psub(qi, a), psub(q.get(m - 1), a)) >= 0)) break;
          m--;
        }
        q.set(m++, qi);
      }
    }
    if (m > 2) {
      float area = 0.f;
      const Pt q0 = q.get(0);
      for (int i = 1; i < m - 1; i++) area += fabsf(cross2(psub(q.get(i), q0), psub(q.get(i + 1), q0)));
      intersection = area / 2.0f;
    }
  }
  float baseS = 1.0f;
  if (mode_flag == 0) baseS = (area1 + area2 - intersection);
  else if (mode_flag == 1) baseS = area1;
  return intersection / baseS;
}

// ---------------------------------------------------------------- box_iou_rotated
// Disjointness pre-test shared by box_iou_rotated and nms_rotated: two rectangles whose circumscribed circles are apart
// cannot intersect, and for such a pair the reference finds no intersection point, hence intersection = 0 and
// IoU = +0.0f EXACTLY (utils.hpp:344-378: `0 / baseS`, baseS > 0 because both areas passed the 1e-14 test; a pair that
// fails that test returns 0 as well).  The test is conservative -- 1 % on the squared radius sum, orders of magnitude
// above the fp32 rounding of either side -- and written so that NaN / inf coordinates fall through to the full
// computation.  It is not an approximation: it only decides WHICH lanes run the ~1100-instruction polygon clipping.
// Thin boxes are excluded from the shortcut: when an extent is below ~1e-7 of the coordinates the reference's edge vector
// collapses to exactly zero in fp32, its point-in-rectangle test (utils.hpp:118-153) degenerates into a strip test, and
// it reports intersections (IoU = 1, inf, negative ...) for boxes that are far apart -- behaviour the kernels reproduce
// bit for bit by running the full computation whenever the smallest extent of either box is below 1e-3 of the pair's
// span (tests/test_oracle_ops.py::test_circumscribed_circle_pretest_is_conservative hunts that boundary on the oracle).
__device__ __forceinline__ float circum_radius(const float* __restrict__ b) {
  return 0.5f * sqrtf(b[2] * b[2] + b[3] * b[3]);
}
__device__ __forceinline__ float min_extent(const float* __restrict__ b) { return fminf(fabsf(b[2]), fabsf(b[3])); }
__device__ __forceinline__ bool may_intersect(float x1, float y1, float r1, float e1, float x2, float y2, float r2,
                                              float e2) {
  const float dx = x1 - x2, dy = y1 - y2, rs = r1 + r2;
  const bool apart = dx * dx + dy * dy > rs * rs * 1.01f + 1e-12f;
  const bool solid = fminf(e1, e2) >= 1e-3f * (fabsf(dx) + fabsf(dy) + rs);
  return !(apart && solid);
}

// One wave owns 64 * ROUNDS consecutive pairs: ROUNDS rounds of 64 cheap pre-tests (unrolled: the box loads of all rounds
// are in flight together) write the zeros directly and collect the surviving pairs in an LDS list; the list is then
// clipped 64 pairs at a time with all lanes busy.  On spread-out boxes (2000 x 512 uniform in 1024^2: 3 % of the pairs
// survive) that is one clipping pass per 64 * ROUNDS pairs instead of ROUNDS.  ROUNDS is chosen on the host so that
// small problems still spread over the chip (1 round = the plain one-pair-per-lane form with the zero shortcut).
template <int ROUNDS>
__global__ __launch_bounds__(256) void box_iou_rotated_kernel(const float* __restrict__ boxes1,
                                                              const float* __restrict__ boxes2,
                                                              float* __restrict__ ious, int n1, int n2,
                                                              int mode_flag, int aligned) {
  constexpr int CHUNK = 64 * ROUNDS;
  __shared__ unsigned short cand[4][CHUNK];
  __shared__ uint32_t iou_lds[4][IOU_LDS_WORDS_PER_WAVE];  // per-lane polygon scratch (see IouScratch)
  const long total = aligned ? (long)n1 : (long)n1 * n2;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long nchunks = (total + CHUNK - 1) / CHUNK;
  for (long ch = (long)blockIdx.x * 4 + wv; ch < nchunks; ch += (long)gridDim.x * 4) {
    const long base = ch * CHUNK;
    bool ok[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const long idx = base + r * 64 + lane;
      ok[r] = idx < total;
      if (ok[r]) {
        const long i = aligned ? idx : idx / n2;
        const long j = aligned ? idx : idx - i * n2;
        const float* a = boxes1 + 5 * i;
        const float* b = boxes2 + 5 * j;
        ok[r] = may_intersect(a[0], a[1], circum_radius(a), min_extent(a), b[0], b[1], circum_radius(b), min_extent(b));
        if (!ok[r]) ious[idx] = 0.f;
      }
    }
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < ROUNDS; r++) {
      const unsigned long long bal = __ballot(ok[r]);
      if (ok[r]) cand[wv][cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(r * 64 + lane);
      cnt += __popcll(bal);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the list is read by other lanes of this wave
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < cnt; e += 64) {
      const long idx = base + cand[wv][e];
      const long i = aligned ? idx : idx / n2;
      const long j = aligned ? idx : idx - i * n2;
      ious[idx] = single_box_iou_rotated(load_box(boxes1 + 5 * i), load_box(boxes2 + 5 * j), mode_flag,
                                         iou_scratch(iou_lds[wv], lane));
    }
    __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next chunk
  }
}

// ---------------------------------------------------------------- argsort (bitonic on 64-bit keys)
// key = (~orderable(score)) << 32 | index  -> ascending key order == descending score, ties by index.
__device__ __forceinline__ uint32_t orderable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__global__ void sort_init_kernel(const float* __restrict__ scores, int n, int npad, uint64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  keys[i] = (i < n) ? (((uint64_t)(~orderable(scores[i]))) << 32) | (uint32_t)i : ~0ull;
}
constexpr int SORT_CHUNK = 4096;  // elements per workgroup in LDS (32 KiB of keys), 1024 threads
// Runs every (k, j) stage with j < SORT_CHUNK for k in [k_lo, k_hi] on one chunk held in LDS.
__global__ __launch_bounds__(1024) void sort_lds_kernel(uint64_t* __restrict__ keys, int npad, int k_lo, int k_hi) {
  __shared__ uint64_t s[SORT_CHUNK];
  const int base = blockIdx.x * SORT_CHUNK;
  const int cnt = min(SORT_CHUNK, npad - base);  // npad is a power of two: cnt == SORT_CHUNK or npad
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) s[i] = keys[base + i];
  __syncthreads();
  for (int k = k_lo; k <= k_hi; k <<= 1) {
    int jstart = min(k >> 1, SORT_CHUNK >> 1);
    for (int j = jstart; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (cnt >> 1); t += blockDim.x) {
        int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // lower index of the pair
        int p = i | j;
        bool up = (((base + i) & k) == 0);
        uint64_t a = s[i], b = s[p];
        if ((a > b) == up) {
          s[i] = b;
          s[p] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < cnt; i += blockDim.x) keys[base + i] = s[i];
}
__global__ void sort_global_step_kernel(uint64_t* __restrict__ keys, int npad, int k, int j) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (npad >> 1)) return;
  int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
  int p = i | j;
  bool up = ((i & k) == 0);
  uint64_t a = keys[i], b = keys[p];
  if ((a > b) == up) {
    keys[i] = b;
    keys[p] = a;
  }
}
__global__ void sort_emit_kernel(const uint64_t* __restrict__ keys, int n, int64_t* __restrict__ order) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) order[i] = (int64_t)(uint32_t)(keys[i] & 0xffffffffu);
}

inline int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

int argsort_desc(const float* scores, int n, int64_t* order, void* ws, size_t ws_bytes, hipStream_t st) {
  if (n <= 0) return SM3_OK;
  const int npad = next_pow2(n);
  if (ws_bytes < (size_t)npad * 8) return SM3_ERR_WORKSPACE;
  uint64_t* keys = (uint64_t*)ws;
  sort_init_kernel<<<(npad + 255) / 256, 256, 0, st>>>(scores, n, npad, keys);
  const int nchunk = (npad + SORT_CHUNK - 1) / SORT_CHUNK;
  // all stages with k <= SORT_CHUNK are chunk-local
  sort_lds_kernel<<<nchunk, 1024, 0, st>>>(keys, npad, 2, min(npad, SORT_CHUNK));
  for (int k = SORT_CHUNK << 1; k <= npad; k <<= 1) {
    for (int j = k >> 1; j >= SORT_CHUNK; j >>= 1)
      sort_global_step_kernel<<<((npad >> 1) + 255) / 256, 256, 0, st>>>(keys, npad, k, j);
    sort_lds_kernel<<<nchunk, 1024, 0, st>>>(keys, npad, k, k);
  }
  sort_emit_kernel<<<(n + 255) / 256, 256, 0, st>>>(keys, n, order);
  return SM3_OK;
}

// ---------------------------------------------------------------- top-k (k <= 2048) without the full sort
// The proposal stage keeps the nms_pre = 2000 best of up to 196 608 anchor scores per level (oriented_rpn_head.py:239-244
// `scores.topk(nms_pre)`).  A full bitonic sort of 2^18 keys is 29 launches; here every 4096-key chunk is sorted once in
// LDS (the existing kernel: even chunks come out ascending, odd ones descending -- the direction bit of the k = 4096
// stage), and a tree of bitonic MERGES of two 2048-runs keeps the better 2048 keys of each pair: log2(chunks) launches
// of 12 LDS stages each.  Keys are (~orderable(score)) << 32 | index, unique, so the first k keys are exactly the
// first k of the full sort (ties by lower index).
constexpr int TOPK_RUN = SORT_CHUNK / 2;
__global__ void topk_init_kernel(const float* __restrict__ scores, int n, int npad, uint64_t* __restrict__ keys) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npad) return;
  keys[i] = (i < n) ? (((uint64_t)(~orderable(scores[i]))) << 32) | (uint32_t)i : ~0ull;
}
// output run p = the 2048 smallest keys of input runs 2p and 2p + 1, ascending.  first != 0: the inputs are whole sorted
// chunks of 4096 (run 2p = first half of an ascending chunk, run 2p + 1 = last half of a descending chunk, i.e. already
// reversed); otherwise ascending runs of 2048 back to back.
__global__ __launch_bounds__(1024) void topk_merge_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                         int nruns, int first) {
  __shared__ uint64_t s[SORT_CHUNK];
  const int p = blockIdx.x;
  const bool has_b = 2 * p + 1 < nruns;
  const uint64_t* a = in + (size_t)(2 * p) * (first ? SORT_CHUNK : TOPK_RUN);
  const uint64_t* b = in + (size_t)(2 * p + 1) * (first ? SORT_CHUNK : TOPK_RUN) + (first ? TOPK_RUN : 0);
  for (int i = threadIdx.x; i < TOPK_RUN; i += blockDim.x) {
    s[i] = a[i];
    const uint64_t v = has_b ? b[i] : ~0ull;
    if (first) s[TOPK_RUN + i] = v;  // descending already
    else s[SORT_CHUNK - 1 - i] = v;  // ascending run, reversed: ascending + descending = bitonic
  }
  __syncthreads();
  for (int j = TOPK_RUN; j > 0; j >>= 1) {
    for (int t = threadIdx.x; t < TOPK_RUN; t += blockDim.x) {
      const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
      const int q = i | j;
      const uint64_t x = s[i], y = s[q];
      if (x > y) {
        s[i] = y;
        s[q] = x;
      }
    }
    __syncthreads();
  }
  for (int i = threadIdx.x; i < TOPK_RUN; i += blockDim.x) out[(size_t)p * TOPK_RUN + i] = s[i];
}

size_t topk_ws_bytes(int n) {
  const size_t npad = ((size_t)(n > 0 ? n : 1) + SORT_CHUNK - 1) / SORT_CHUNK * SORT_CHUNK;
  return (npad + npad / 2) * 8;
}

int topk_desc(const float* scores, int n, int k, int64_t* order, void* ws, size_t ws_bytes, hipStream_t st) {
  if (k > n) k = n;
  if (k <= 0) return SM3_OK;
  if (k > TOPK_RUN) return SM3_ERR_UNSUPPORTED;
  if (ws_bytes < topk_ws_bytes(n)) return SM3_ERR_WORKSPACE;
  const int npad = (n + SORT_CHUNK - 1) / SORT_CHUNK * SORT_CHUNK;
  uint64_t* bufa = (uint64_t*)ws;
  uint64_t* bufb = bufa + npad;
  topk_init_kernel<<<(npad + 255) / 256, 256, 0, st>>>(scores, n, npad, bufa);
  int nruns = npad / SORT_CHUNK;
  sort_lds_kernel<<<nruns, 1024, 0, st>>>(bufa, npad, 2, SORT_CHUNK);
  const uint64_t* cur = bufa;  // one chunk: ascending, the first k keys are the answer
  uint64_t* nxt = bufb;
  int first = 1;
  while (nruns > 1) {
    const int nout = (nruns + 1) / 2;
    topk_merge_kernel<<<nout, 1024, 0, st>>>(cur, nxt, nruns, first);
    cur = nxt;
    nxt = (nxt == bufb) ? bufa : bufb;
    nruns = nout;
    first = 0;
  }
  sort_emit_kernel<<<(k + 255) / 256, 256, 0, st>>>(cur, k, order);
  return SM3_OK;
}

// ---------------------------------------------------------------- NMS (shared structure)
// Stage 1: suppression bit-matrix over the score-sorted boxes. Tile = 64 rows x 64 cols = ONE wavefront:
//   lane r owns sorted box rb*64+r and builds its 64-bit word against the 64 column boxes staged in LDS.
//   Only tiles with cb >= rb are launched (triangular grid); within the diagonal tile only bits c > r are set.
// Stage 2: greedy sweep on the device, one workgroup; per 64-box block wave 0 resolves the intra-block
//   dependencies from the diagonal words, then all threads OR the kept rows into the removal vector.
__device__ __forceinline__ void tri_decode(int t, int nblk, int& rb, int& cb) {
  // t enumerates pairs (rb <= cb) row-major: rows have nblk, nblk-1, ... entries
  int r = 0;
  int rem = t;
  // closed form via float sqrt then fix-up
  float fn = (float)nblk;
  r = (int)floorf(((2.f * fn + 1.f) - sqrtf((2.f * fn + 1.f) * (2.f * fn + 1.f) - 8.f * (float)t)) * 0.5f);
  if (r < 0) r = 0;
  if (r > nblk - 1) r = nblk - 1;
  while (r > 0 && (long)r * nblk - (long)r * (r - 1) / 2 > t) r--;
  while ((long)(r + 1) * nblk - (long)(r + 1) * r / 2 <= t) r++;
  rem = t - (int)((long)r * nblk - (long)r * (r - 1) / 2);
  rb = r;
  cb = r + rem;
}

// 256 threads per 64x64 tile: wave w owns columns [16w, 16w+16) of the tile's 64 row boxes.  Phase 1 runs the
// circumscribed-circle pre-test on its 64 x 16 pairs (lane = row) and compacts the survivors into an LDS list; phase 2
// clips the listed pairs 64 at a time and ORs the suppression bits into the row words (ds_or_b64).  Round 2 ran the full
// rotated IoU on all 4096 pairs of a tile: 994 VALU lane-instructions per pair on the 10 000-box bench shape, where 3 %
// of the pairs can intersect at all.
__global__ __launch_bounds__(256) void nms_rotated_mask_kernel(const float* __restrict__ dets, int stride,
                                                              const int64_t* __restrict__ order, int n,
                                                              int nblk, float thr, int multi_label,
                                                              uint64_t* __restrict__ mask,
                                                              uint64_t* __restrict__ diagt) {
  int rb, cb;
  tri_decode(blockIdx.x, nblk, rb, cb);
  // per box: x, y, w, h, angle, label, circumscribed radius, smallest extent
  __shared__ float cbox[64 * 8], rbox[64 * 8];
  __shared__ unsigned long long rowword[64];
  __shared__ unsigned short cand[4][64 * 16];
  __shared__ int cand_n[4];
  // per-lane polygon scratch (see IouScratch) for TWO clipping waves: the survivors of all four waves' pre-tests are clipped
  // as ONE list, 128 pairs per round (3 % of a tile's 4096 pairs survive on the bench boxes: ~120 -- one full round instead
  // of four rounds with 30 of 64 lanes busy), and 35 KB of LDS per workgroup keep four workgroups on a CU
  __shared__ uint32_t iou_lds[2][IOU_LDS_WORDS_PER_WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (wv < 2) {
    float* dst = wv == 0 ? cbox : rbox;
    const int bi = (wv == 0 ? cb : rb) * 64 + lane;
    if (bi < n) {
      const float* d = dets + order[bi] * (int64_t)stride;
#pragma unroll
      for (int k = 0; k < 5; k++) dst[lane * 8 + k] = d[k];
      dst[lane * 8 + 5] = multi_label ? d[5] : 0.f;
      dst[lane * 8 + 6] = circum_radius(d);
      dst[lane * 8 + 7] = min_extent(d);
    }
  } else if (wv == 2) {
    rowword[lane] = 0ull;
  }
  __syncthreads();
  const int ri = rb * 64 + lane;
  const int ncol = min(64, n - cb * 64);
  // a pair whose boxes cannot intersect has IoU = +0 exactly (see may_intersect): it can only be suppressed when the
  // threshold is not positive, in which case every pair goes through the full computation as before
  const bool prefilter = thr > 0.f;
  // phase 1: lane = row box, 16 cheap tests against this wave's columns; survivors into the wave's LDS list
  int cnt = 0;
  for (int cc = 0; cc < 16; cc++) {
    const int c = 16 * wv + cc;
    bool ok = ri < n && c < ncol && (rb != cb || c > lane) &&
              !(multi_label && cbox[c * 8 + 5] != rbox[lane * 8 + 5]);
    if (ok && prefilter)
      ok = may_intersect(rbox[lane * 8], rbox[lane * 8 + 1], rbox[lane * 8 + 6], rbox[lane * 8 + 7], cbox[c * 8],
                         cbox[c * 8 + 1], cbox[c * 8 + 6], cbox[c * 8 + 7]);
    const unsigned long long bal = __ballot(ok);
    if (ok) cand[wv][cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(lane * 64 + c);
    cnt += __popcll(bal);
  }
  if (lane == 0) cand_n[wv] = cnt;
  __syncthreads();
  // phase 2: the surviving pairs of the whole tile, 128 at a time on waves 0 and 1 with every lane busy
  if (wv < 2) {
    const int n0 = cand_n[0], n1 = n0 + cand_n[1], n2 = n1 + cand_n[2], total = n2 + cand_n[3];
    for (int e = wv * 64 + lane; e < total; e += 128) {
      const int w = (e >= n0) + (e >= n1) + (e >= n2);
      const int rc = cand[w][e - (w == 0 ? 0 : w == 1 ? n0 : w == 2 ? n1 : n2)], r = rc >> 6, c = rc & 63;
      // row box is the higher-scoring one: reference calls iou(dets[i], dets[j]) with i kept, j candidate
      const float iou =
          single_box_iou_rotated(load_box(&rbox[r * 8]), load_box(&cbox[c * 8]), 0, iou_scratch(iou_lds[wv], lane));
      if (iou >= thr) atomicOr(&rowword[r], 1ull << c);  // cpu/nms_rotated.cpp:51 uses >=
    }
  }
  __syncthreads();
  if (wv == 0 && ri < n) mask[(size_t)ri * nblk + cb] = rowword[lane];
  if (wv == 1 && rb == cb) {  // the diagonal tile transposed, for the sweep's parallel intra-block resolution
    uint64_t t = 0;
#pragma unroll 8
    for (int r = 0; r < 64; r++) t |= ((rowword[r] >> lane) & 1ull) << r;
    diagt[(size_t)rb * 64 + lane] = t;
  }
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes,
                                                     const int64_t* __restrict__ order, int n, int nblk,
                                                     float thr, float offset, uint64_t* __restrict__ mask,
                                                     uint64_t* __restrict__ diagt) {
  int rb, cb;
  tri_decode(blockIdx.x, nblk, rb, cb);
  __shared__ float cbox[64 * 5];
  const int lane = threadIdx.x;
  const int ci = cb * 64 + lane;
  if (ci < n) {
    const float* d = boxes + order[ci] * 4;
    float x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
    cbox[lane * 5 + 0] = x1;
    cbox[lane * 5 + 1] = y1;
    cbox[lane * 5 + 2] = x2;
    cbox[lane * 5 + 3] = y2;
    cbox[lane * 5 + 4] = (x2 - x1 + offset) * (y2 - y1 + offset);  // cpu/nms.cpp:14
  }
  __syncthreads();
  const int ri = rb * 64 + lane;
  uint64_t word = 0;
  if (ri < n) {
    const float* d = boxes + order[ri] * 4;
    const float ix1 = d[0], iy1 = d[1], ix2 = d[2], iy2 = d[3];
    const float iarea = (ix2 - ix1 + offset) * (iy2 - iy1 + offset);
    const int ncol = min(64, n - cb * 64);
    const int cstart = (rb == cb) ? lane + 1 : 0;
    for (int c = cstart; c < ncol; c++) {
      float xx1 = fmaxf(ix1, cbox[c * 5 + 0]);
      float yy1 = fmaxf(iy1, cbox[c * 5 + 1]);
      float xx2 = fminf(ix2, cbox[c * 5 + 2]);
      float yy2 = fminf(iy2, cbox[c * 5 + 3]);
      float w = fmaxf(0.f, xx2 - xx1 + offset);
      float h = fmaxf(0.f, yy2 - yy1 + offset);
      float inter = w * h;
      float ovr = inter / (iarea + cbox[c * 5 + 4] - inter);  // division form, cpu/nms.cpp:46
      if (ovr > thr) word |= (1ull << c);
    }
    mask[(size_t)ri * nblk + cb] = word;
  }
  if (rb == cb) {  // the diagonal tile transposed (all 64 lanes vote), for the sweep's parallel intra-block resolution
    uint64_t t = 0;
#pragma unroll 8
    for (int c = 0; c < 64; c++) {
      const uint64_t col = __ballot((word >> c) & 1ull);
      if (lane == c) t = col;
    }
    diagt[(size_t)rb * 64 + lane] = t;
  }
}


constexpr int SWEEP_THREADS = 1024;
__global__ __launch_bounds__(SWEEP_THREADS) void nms_sweep_kernel(const uint64_t* __restrict__ mask,
                                                                 const int64_t* __restrict__ order, int n,
                                                                 int nblk, uint64_t* __restrict__ remv_g,
                                                                 int64_t* __restrict__ keep,
                                                                 int32_t* __restrict__ num_keep) {
  // removal vector lives in global scratch (nblk words) so that N is unbounded; it is L2-resident.
  __shared__ uint64_t s_kept;
  __shared__ int s_count;
  const int tid = threadIdx.x;
  for (int c = tid; c < nblk; c += SWEEP_THREADS) remv_g[c] = 0;
  if (tid == 0) s_count = 0;
  __syncthreads();
  for (int blk = 0; blk < nblk; blk++) {
    if (tid < 64) {
      const int lane = tid;
      const int i = blk * 64 + lane;
      uint64_t word = (i < n) ? mask[(size_t)i * nblk + blk] : 0ull;
      uint64_t removed = remv_g[blk];
      const int nvalid = min(64, n - blk * 64);
      uint64_t kept = 0;
      for (int b = 0; b < nvalid; b++) {
        uint64_t wb = __shfl(word, b, 64);
        if (!((removed >> b) & 1ull)) {
          kept |= (1ull << b);
          removed |= wb;
        }
      }
      const int base = s_count;
      if ((kept >> lane) & 1ull) {
        int pos = __popcll(kept & ((1ull << lane) - 1ull));
        keep[base + pos] = order[i];
      }
      if (lane == 0) {
        s_kept = kept;
        s_count = base + __popcll(kept);
      }
    }
    __syncthreads();
    const uint64_t kept = s_kept;
    // OR the kept rows into the removal vector.  All 1024 threads stream the 64 x (nblk-blk-1) sub-matrix of this
    // block row with independent loads (a per-column loop over the kept rows is a chain of ~60 dependent L2 round
    // trips per block: 2.5 ms at N = 8768); words of suppressed rows are skipped, hits are merged with 64-bit atomics.
    const int ncols = nblk - (blk + 1);
    const uint64_t* mrow = mask + (size_t)blk * 64 * nblk + (blk + 1);
    for (int idx = tid; idx < 64 * ncols; idx += SWEEP_THREADS) {
      const int b = idx / ncols, c = idx - b * ncols;
      if ((kept >> b) & 1ull) {
        const uint64_t w = mrow[(size_t)b * nblk + c];
        if (w) atomicOr((unsigned long long*)(remv_g + blk + 1 + c), (unsigned long long)w);
      }
    }
    __syncthreads();
  }
  if (tid == 0) *num_keep = s_count;
}


// The same sweep with the removal vector in LDS and the next block row PREFETCHED.  The sweep is a chain of nblk serial
// steps; in the kernel above every step pays two global round trips (the row words after the kept set is known, and the
// L2 atomics that the next step's `removed` word has to wait for): 6.4 us per step, 0.88 ms at N = 8768 although only
// 4.8 MB move (rocprofv3, round 3).  Here (i) remv[] lives in LDS (ds_or_b64, no L2 round trip), and (ii) while block
// `blk` is being resolved every thread already holds the words of block row `blk + 1` in registers -- they do not depend
// on which boxes survive, only their USE does -- so a step costs the 64-box serial scan plus one LDS phase.
// Columns past 16 * PREF = 256 of a block row (N > 16 448) are read after the barrier as before.  Needs nblk * 8 bytes
// of LDS.
constexpr int SWEEP_CH = 3;  // 64-column chunks per row held in registers: rows of up to 192 column blocks (N <= 12 352)
__global__ __launch_bounds__(SWEEP_THREADS) void nms_sweep_lds_kernel(const uint64_t* __restrict__ mask,
                                                                     const uint64_t* __restrict__ diagt,
                                                                     const int64_t* __restrict__ order, int n,
                                                                     int nblk, int64_t* __restrict__ keep,
                                                                     int32_t* __restrict__ num_keep) {
  extern __shared__ __attribute__((aligned(16))) uint64_t remv[];  // [nblk + 4]: the loop runs whole triples of steps
  __shared__ uint64_t s_kept[2];
  __shared__ int s_count;
  const int tid = threadIdx.x;
  for (int c = tid; c < nblk + 4; c += SWEEP_THREADS) remv[c] = 0;
  if (tid < 2) s_kept[tid] = 0;
  if (tid == 0) s_count = 0;
  // wave w owns rows 4w .. 4w + 3 of a block, lane = column (64 contiguous words = 512 bytes per row and load).  A thread
  // ORs its four rows in registers and issues ONE conflict-free LDS update per 64-column chunk: 48 wave-level ds_or_b64 per
  // block.  (Before: thread = (row, column lane), 160 wave-level updates with 4 lanes per address -- the LDS pipe, not the
  // scan, set the 2.2 us per block; a flat index / ncols mapping before that cost ~1300 VALU instructions per thread.)
  const int wv = tid >> 6, lane = tid & 63;
  // block row `blk` of the mask (right of the diagonal); for wave 0 also the TRANSPOSED diagonal word of box (blk, lane)
  // (bit b' set <=> box b' of this block suppresses it), the first word right of the diagonal and the box's original index.  Everything is predicated per lane, nothing branches: blocks past the end load
  // nothing and decide nothing.
  auto fetch = [&](int blk, uint64_t (&w)[4][SWEEP_CH], uint64_t& diag, uint64_t& diag1, int64_t& ord) {
    const int ncols = nblk - (blk + 1);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = blk * 64 + 4 * wv + r;
      const uint64_t* mrow = mask + (size_t)row * nblk + (blk + 1);
#pragma unroll
      for (int j = 0; j < SWEEP_CH; j++) {
        const int c = lane + 64 * j;
        w[r][j] = (row < n && c < ncols) ? mrow[c] : 0ull;
      }
    }
    diag = 0;
    diag1 = 0;
    ord = 0;
    if (tid < 64 && blk * 64 + tid < n) {
      diag = diagt[(size_t)blk * 64 + tid];
      if (ncols > 0) diag1 = mask[(size_t)(blk * 64 + tid) * nblk + blk + 1];
      ord = order[blk * 64 + tid];  // the kept boxes' original indices: loaded ahead, not inside the serial step
    }
  };
  // Step `blk`, ONE barrier:
  //   * wave 0 decides the 64 boxes of block blk (the serial scan) and ORs the first word right of the diagonal of every
  //     survivor into remv[blk + 1] itself, so that the next scan does not depend on anybody else;
  //   * meanwhile all 16 waves OR the rows of the survivors of block blk - 1 (decided in the previous step, `s_kept`
  //     double-buffered) into remv[blk ..]: OR is idempotent, so re-applying the word wave 0 already added is harmless, and
  //     remv[c] is complete for blocks <= c - 2 at the barrier before step c, which is all the scan of block c needs on
  //     top of wave 0's own contribution for block c - 1.
  auto step = [&](int blk, uint64_t dcur, uint64_t d1cur, int64_t ocur, uint64_t (&pw)[4][SWEEP_CH], uint64_t& pd,
                  uint64_t& pd1, int64_t& po) {
    // `pw` holds the rows of block blk - 1 (blk = 0: nothing, s_kept is 0)
    const uint64_t kprev = s_kept[(blk + 1) & 1];
    const unsigned k4 = (unsigned)(kprev >> (4 * wv)) & 15u;  // this wave's four rows
    const int pblk = blk - 1;
    const int pcols = nblk - blk;  // columns right of the diagonal of block blk - 1
    if (k4) {
#pragma unroll
      for (int j = 0; j < SWEEP_CH; j++) {
        uint64_t v = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) v |= ((k4 >> r) & 1u) ? pw[r][j] : 0ull;
        if (v != 0) atomicOr((unsigned long long*)(remv + blk + lane + 64 * j), (unsigned long long)v);
      }
      if (pcols > 64 * SWEEP_CH) {  // longer rows: the part that was not prefetched
        for (int c = lane + 64 * SWEEP_CH; c < pcols; c += 64) {
          uint64_t v = 0;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int row = pblk * 64 + 4 * wv + r;
            if (((k4 >> r) & 1u) && row < n) v |= mask[(size_t)row * nblk + blk + c];
          }
          if (v != 0) atomicOr((unsigned long long*)(remv + blk + c), (unsigned long long)v);
        }
      }
    }
    // the set is consumed: refill it with the block two steps on (in flight during the scans of this and the next step)
    fetch(blk + 2, pw, pd, pd1, po);
    if (tid < 64) {
      const uint64_t r0 = remv[blk];  // after this wave's own ORs above and of the previous step: LDS is in order per wave
      // (the builtins return int: without the uint32_t casts a set bit 31 of the low word sign-extends into the high one)
      const uint64_t removed = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(r0 >> 32)) << 32) |
                               (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)r0);
      const int nvalid = min(64, n - blk * 64);  // <= 0 for the padding steps past the last block
      const uint64_t valid = nvalid >= 64 ? ~0ull : (nvalid > 0 ? (1ull << nvalid) - 1ull : 0ull);
      // Greedy order inside the block (box b survives unless an earlier SURVIVOR of this block suppresses it) as a parallel
      // fixed point: K <- { b in S : no b' in K suppresses b }, S = the boxes earlier blocks left alive, starting from
      // K = S.  Suppression only runs from lower to higher indices, so after t rounds the first t boxes are final
      // (induction on the index): at most 64 rounds, and a round that changes nothing has reached the unique solution of
      // the recursion = the sequential result.  A round is one AND, one compare and one ballot over the transposed
      // diagonal words; the number of rounds is the longest suppression chain + 1 (2-4 on detection boxes).  The
      // sequential scan this replaces cost 64 x 8 scalar-unit instructions at one issue per ~10 cycles = 2.2 us per block
      // whatever the data -- it, not the memory traffic, was the sweep.
      const uint64_t S = valid & ~removed;
      const bool in_s = (S >> lane) & 1ull;
      uint64_t kept = S;
      for (int round = 0; round < 66; round++) {
        const uint64_t nk = __ballot(in_s && (dcur & kept) == 0ull);
        if (nk == kept) break;
        kept = nk;
      }
      const int base = s_count;
      if ((kept >> lane) & 1ull) {
        const int pos = __popcll(kept & ((1ull << lane) - 1ull));
        keep[base + pos] = ocur;
        if (d1cur) atomicOr((unsigned long long*)(remv + blk + 1), (unsigned long long)d1cur);
      }
      if (lane == 0) {
        s_kept[blk & 1] = kept;
        s_count = base + __popcll(kept);
      }
    }
    __syncthreads();
  };
  // three register sets in rotation: {rows of the block being applied, the block being scanned, the block after it}
  uint64_t w0[4][SWEEP_CH], w1[4][SWEEP_CH], w2[4][SWEEP_CH], d0, d1, d2 = 0, e0, e1, e2 = 0;
  int64_t o0, o1, o2 = 0;
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int j = 0; j < SWEEP_CH; j++) w2[r][j] = 0;
  fetch(0, w0, d0, e0, o0);
  fetch(1, w1, d1, e1, o1);
  __syncthreads();
  // whole triples of steps, no branch between them (see fetch): the steps past the last block apply and decide nothing
  for (int blk = 0; blk < nblk; blk += 3) {
    step(blk, d0, e0, o0, w2, d2, e2, o2);
    step(blk + 1, d1, e1, o1, w0, d0, e0, o0);
    step(blk + 2, d2, e2, o2, w1, d1, e1, o1);
  }
  if (tid == 0) *num_keep = s_count;
}

size_t nms_ws_layout(int n, size_t* off_order, size_t* off_mask, size_t* off_remv, size_t* off_sort,
                     size_t* off_diagt) {
  const int nblk = (n + 63) / 64;
  size_t o = 0;
  *off_order = o;
  o += align_up((size_t)n * 8, 256);
  *off_mask = o;
  o += align_up((size_t)n * nblk * 8, 256);
  *off_remv = o;
  o += align_up((size_t)nblk * 8, 256);
  *off_sort = o;
  o += align_up((size_t)next_pow2(n > 0 ? n : 1) * 8, 256);
  *off_diagt = o;  // the diagonal 64x64 tiles transposed: word (blk, c) = which boxes of block blk suppress its box c
  o += align_up((size_t)nblk * 64 * 8, 256);
  return o;
}

// ---------------------------------------------------------------- RoIAlignRotated
struct RoiGeom {
  float cw, ch, rw, rh, cosv, sinv, bin_h, bin_w, start_h, start_w;
  int grid_h, grid_w, batch;
};
// cpu/roi_align_rotated.cpp:129-172
__device__ __forceinline__ RoiGeom roi_geometry(const float* __restrict__ roi, float spatial_scale,
                                                int aligned, int clockwise, int ph, int pw,
                                                int sampling_ratio) {
  RoiGeom g;
  g.batch = (int)roi[0];
  float offset = aligned ? 0.5f : 0.0f;
  g.cw = roi[1] * spatial_scale - offset;
  g.ch = roi[2] * spatial_scale - offset;
  g.rw = roi[3] * spatial_scale;
  g.rh = roi[4] * spatial_scale;
  float theta = roi[5];
  if (clockwise) theta = -theta;
  g.cosv = (float)cos((double)theta);
  g.sinv = (float)sin((double)theta);
  if (!aligned) {
    g.rw = fmaxf(g.rw, 1.f);
    g.rh = fmaxf(g.rh, 1.f);
  }
  g.bin_h = g.rh / (float)ph;
  g.bin_w = g.rw / (float)pw;
  g.grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(g.rh / ph);
  g.grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(g.rw / pw);
  g.start_h = (float)(-(double)g.rh / 2.0);
  g.start_w = (float)(-(double)g.rw / 2.0);
  return g;
}

struct Sample {
  int p1, p2, p3, p4;  // offsets y*W+x of the four corners (-1 => sample contributes nothing)
  float w1, w2, w3, w4;
};

// pre_calc_for_bilinear_interpolate (cpu/roi_align_rotated.cpp:24-113) for one sample point
__device__ __forceinline__ Sample make_sample(const RoiGeom& g, int height, int width, int ph, int pw, int iy,
                                              int ix) {
  const float yy = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
  const float xx = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
  float y = yy * g.cosv - xx * g.sinv + g.ch;
  float x = yy * g.sinv + xx * g.cosv + g.cw;
  Sample s;
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
    s.p1 = s.p2 = s.p3 = s.p4 = -1;
    s.w1 = s.w2 = s.w3 = s.w4 = 0.f;
    return s;
  }
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) {
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) {
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low, lx = x - x_low;
  float hy = (float)(1. - (double)ly), hx = (float)(1. - (double)lx);
  s.w1 = hy * hx;
  s.w2 = hy * lx;
  s.w3 = ly * hx;
  s.w4 = ly * lx;
  s.p1 = y_low * width + x_low;
  s.p2 = y_low * width + x_high;
  s.p3 = y_high * width + x_low;
  s.p4 = y_high * width + x_high;
  return s;
}

// Multi-level variant (RotatedSingleRoIExtractor.forward, mmrotate/models/roi_heads/roi_extractors/
// rotate_single_level_roi_extractor.py:103-140): every RoI reads the pyramid level map_roi_levels (:66-84) assigns
// to it.  The reference does nonzero() + gather + one RoIAlign launch + scatter per level (4 host syncs); here the
// level is evaluated by the RoI's own workgroup and ONE launch serves all levels.
constexpr int ROI_MAX_LEVELS = 8;
struct RoiLevels {
  const float* in[ROI_MAX_LEVELS];
  float* gin[ROI_MAX_LEVELS];
  int h[ROI_MAX_LEVELS], w[ROI_MAX_LEVELS];
  float scale[ROI_MAX_LEVELS];
  int n;
  float finest;
  int32_t* levels_out;
};

// scale = sqrt(w*h); floor(log2(scale / finest_scale + 1e-6)) clamped to [0, n-1]   (:81-84, fp32 like torch)
__device__ __forceinline__ int roi_target_level(const float* __restrict__ roi, float finest, int nlev) {
  const float sc = sqrtf(roi[3] * roi[4]);
  float l = floorf(log2f(sc / finest + 1e-6f));
  l = fminf(fmaxf(l, 0.f), (float)(nlev - 1));
  return (int)l;  // NaN (negative w*h) -> 0 after the clamps above would be UB-free: fmaxf(NaN,0) = 0
}

constexpr int ROI_THREADS = 256;
constexpr int ROI_MAX_LDS_SAMPLES = 1024;  // 32 KiB of Sample; larger adaptive grids recompute on the fly

// One workgroup per RoI.  Phase 1: the (bin, iy, ix) sample table is computed once into LDS and shared by all
// channels (the reference's pre_calc idea).  Phase 2: threads sweep (c, bin) in OUTPUT order -> coalesced
// stores; layout 1 (NHWC input) sweeps c fastest -> coalesced 4-byte gathers across channels, and the tile
// is transposed through LDS when it fits.
template <int LAYOUT, int MULTI = 0>
__global__ __launch_bounds__(ROI_THREADS) void roi_align_rotated_fwd_kernel(
    const float* __restrict__ input, const float* __restrict__ rois, float* __restrict__ output, int channels,
    int height, int width, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned, int clockwise,
    RoiLevels lv = RoiLevels()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Sample* tab = (Sample*)smem;
  const int n = blockIdx.x;
  if (MULTI) {  // block-uniform: this RoI's pyramid level
    const int l = roi_target_level(rois + 6 * (size_t)n, lv.finest, lv.n);
    input = lv.in[l];
    height = lv.h[l];
    width = lv.w[l];
    spatial_scale = lv.scale[l];
    if (lv.levels_out && threadIdx.x == 0) lv.levels_out[n] = l;
  }
  __shared__ RoiGeom g_sh;  // the RoI's geometry (two double-precision sincos) once per workgroup, not once per thread
  if (threadIdx.x == 0) g_sh = roi_geometry(rois + 6 * (size_t)n, spatial_scale, aligned, clockwise, PH, PW, sampling_ratio);
  __syncthreads();
  const RoiGeom g = g_sh;
  const int bins = PH * PW;
  const int spb = g.grid_h * g.grid_w;  // samples per bin
  const int nsamp = bins * spb;
  const bool use_tab = nsamp <= ROI_MAX_LDS_SAMPLES;
  if (use_tab) {
    for (int s = threadIdx.x; s < nsamp; s += ROI_THREADS) {
      int bin = s / spb, r = s - bin * spb;
      int iy = r / g.grid_w, ix = r - iy * g.grid_w;
      tab[s] = make_sample(g, height, width, bin / PW, bin % PW, iy, ix);
    }
  }
  __syncthreads();
  int cnt_i = spb < 1 ? 1 : spb;
  const float count = (float)cnt_i;
  const size_t plane = (size_t)height * width;
  const int total = channels * bins;
  float* out = output + (size_t)n * total;
  for (int idx = threadIdx.x; idx < total; idx += ROI_THREADS) {
    int c, bin;
    if (LAYOUT == 0) {
      c = idx / bins;
      bin = idx - c * bins;
    } else {
      bin = idx / channels;
      c = idx - bin * channels;
    }
    const float* in;
    size_t cstride;
    if (LAYOUT == 0) {
      in = input + ((size_t)g.batch * channels + c) * plane;
      cstride = 1;
    } else {
      in = input + (size_t)g.batch * plane * channels + c;
      cstride = channels;
    }
    float val = 0.f;
    for (int r = 0; r < spb; r++) {
      Sample s;
      if (use_tab) s = tab[bin * spb + r];
      else s = make_sample(g, height, width, bin / PW, bin % PW, r / g.grid_w, r % g.grid_w);
      if (s.p1 < 0) {
        // reference multiplies the zero weights with input[pos 0] (cpu/roi_align_rotated.cpp:52-66,189-192)
        float v0 = in[0];
        val += 0.f * v0 + 0.f * v0 + 0.f * v0 + 0.f * v0;
      } else {
        val += s.w1 * in[s.p1 * cstride] + s.w2 * in[s.p2 * cstride] + s.w3 * in[s.p3 * cstride] +
               s.w4 * in[s.p4 * cstride];
      }
    }
    val /= count;
    out[(size_t)c * bins + bin] = val;
  }
}

// NHWC forward with CHANNEL VECTORS (round 4).  The kernel above gathers one float per lane and corner: 64 four-byte
// loads per output element, 12.5 k wave-load instructions of 256 B per RoI -- bound by the texture addresser's
// instruction rate, not by bytes (0.73 TB/s on the 256 x 256 x 256 level).  Here a lane owns FOUR consecutive channels of
// one bin: a pixel's channels are contiguous in NHWC, so one wave instruction moves a whole 1 KB pixel row (C = 256) and a
// bin costs 16 of them (4 samples x 4 corners) instead of 256.  Same taps, same weights, same order of the fp32
// operations per channel as the scalar kernel (and as cpu/roi_align_rotated.cpp:175-200), so the results are bit-identical
// to it.  The (C x bins) result tile is assembled in LDS in OUTPUT order and leaves as contiguous 16-byte stores.
// Requires C % 4 == 0 and C * bins * 4 B + the sample table within the LDS budget (host checks; else the scalar kernel).
typedef float rf4 __attribute__((ext_vector_type(4)));
template <int MULTI>
__global__ __launch_bounds__(ROI_THREADS) void roi_align_rotated_fwd_vec_kernel(
    const float* __restrict__ input, const float* __restrict__ rois, float* __restrict__ output, int channels,
    int height, int width, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned, int clockwise,
    int tab_bytes, RoiLevels lv = RoiLevels()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Sample* tab = (Sample*)smem;
  float* tile = (float*)(smem + tab_bytes);  // [channels][bins], the RoI's block of the output
  const int n = blockIdx.x;
  if (MULTI) {  // block-uniform: this RoI's pyramid level
    const int l = roi_target_level(rois + 6 * (size_t)n, lv.finest, lv.n);
    input = lv.in[l];
    height = lv.h[l];
    width = lv.w[l];
    spatial_scale = lv.scale[l];
    if (lv.levels_out && threadIdx.x == 0) lv.levels_out[n] = l;
  }
  // the RoI's geometry (two double-precision sincos) once per workgroup, not once per thread
  __shared__ RoiGeom g_sh;
  if (threadIdx.x == 0) g_sh = roi_geometry(rois + 6 * (size_t)n, spatial_scale, aligned, clockwise, PH, PW, sampling_ratio);
  __syncthreads();
  const RoiGeom g = g_sh;
  const int bins = PH * PW;
  const int spb = g.grid_h * g.grid_w;  // samples per bin
  const int nsamp = bins * spb;
  const bool use_tab = nsamp * (int)sizeof(Sample) <= tab_bytes;
  if (use_tab) {
    for (int s = threadIdx.x; s < nsamp; s += ROI_THREADS) {
      int bin = s / spb, r = s - bin * spb;
      int iy = r / g.grid_w, ix = r - iy * g.grid_w;
      tab[s] = make_sample(g, height, width, bin / PW, bin % PW, iy, ix);
    }
  }
  __syncthreads();
  const int cnt_i = spb < 1 ? 1 : spb;
  const float count = (float)cnt_i;
  const int CQ = channels >> 2;
  const float* base = input + (size_t)g.batch * height * width * channels;
  for (int u = threadIdx.x; u < bins * CQ; u += ROI_THREADS) {
    const int bin = u / CQ, cq = u - bin * CQ;
    const float* in = base + 4 * cq;
    rf4 val = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < spb; r++) {
      Sample s;
      if (use_tab) s = tab[bin * spb + r];
      else s = make_sample(g, height, width, bin / PW, bin % PW, r / g.grid_w, r % g.grid_w);
      if (s.p1 < 0) {
        // reference multiplies the zero weights with input[pos 0] (cpu/roi_align_rotated.cpp:52-66,189-192)
        const rf4 v0 = *reinterpret_cast<const rf4*>(in);
        val += 0.f * v0 + 0.f * v0 + 0.f * v0 + 0.f * v0;
      } else {
        const rf4 v1 = *reinterpret_cast<const rf4*>(in + (size_t)s.p1 * channels);
        const rf4 v2 = *reinterpret_cast<const rf4*>(in + (size_t)s.p2 * channels);
        const rf4 v3 = *reinterpret_cast<const rf4*>(in + (size_t)s.p3 * channels);
        const rf4 v4 = *reinterpret_cast<const rf4*>(in + (size_t)s.p4 * channels);
        val += s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
      }
    }
    val /= count;
#pragma unroll
    for (int e = 0; e < 4; e++) tile[(4 * cq + e) * bins + bin] = val[e];
  }
  __syncthreads();
  const int total = channels * bins;  // a multiple of 4: the RoI's output block starts 16-byte aligned
  rf4* out4 = reinterpret_cast<rf4*>(output + (size_t)n * total);
  const rf4* t4 = reinterpret_cast<const rf4*>(tile);
  for (int i = threadIdx.x; i < (total >> 2); i += ROI_THREADS) out4[i] = t4[i];
}

template <int LAYOUT, int MULTI = 0>
__global__ __launch_bounds__(ROI_THREADS) void roi_align_rotated_bwd_kernel(
    const float* __restrict__ grad_output, const float* __restrict__ rois, float* __restrict__ grad_input,
    int channels, int height, int width, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
    int clockwise, RoiLevels lv = RoiLevels()) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Sample* tab = (Sample*)smem;
  const int n = blockIdx.x;
  if (MULTI) {
    const int l = roi_target_level(rois + 6 * (size_t)n, lv.finest, lv.n);
    grad_input = lv.gin[l];
    height = lv.h[l];
    width = lv.w[l];
    spatial_scale = lv.scale[l];
  }
  __shared__ RoiGeom g_sh;  // the RoI's geometry (two double-precision sincos) once per workgroup, not once per thread
  if (threadIdx.x == 0) g_sh = roi_geometry(rois + 6 * (size_t)n, spatial_scale, aligned, clockwise, PH, PW, sampling_ratio);
  __syncthreads();
  const RoiGeom g = g_sh;
  const int bins = PH * PW;
  const int spb = g.grid_h * g.grid_w;
  const int nsamp = bins * spb;
  const bool use_tab = nsamp <= ROI_MAX_LDS_SAMPLES;
  if (use_tab) {
    for (int s = threadIdx.x; s < nsamp; s += ROI_THREADS) {
      int bin = s / spb, r = s - bin * spb;
      int iy = r / g.grid_w, ix = r - iy * g.grid_w;
      tab[s] = make_sample(g, height, width, bin / PW, bin % PW, iy, ix);
    }
  }
  __syncthreads();
  const float count = (float)spb;  // cpu/roi_align_rotated.cpp:331
  const size_t plane = (size_t)height * width;
  const int total = channels * bins;
  const float* go = grad_output + (size_t)n * total;
  for (int idx = threadIdx.x; idx < total; idx += ROI_THREADS) {
    int c, bin;
    if (LAYOUT == 0) {
      c = idx / bins;
      bin = idx - c * bins;
    } else {
      bin = idx / channels;
      c = idx - bin * channels;
    }
    float* gin;
    size_t cstride;
    if (LAYOUT == 0) {
      gin = grad_input + ((size_t)g.batch * channels + c) * plane;
      cstride = 1;
    } else {
      gin = grad_input + (size_t)g.batch * plane * channels + c;
      cstride = channels;
    }
    const float gval = go[(size_t)c * bins + bin];
    for (int r = 0; r < spb; r++) {
      Sample s;
      if (use_tab) s = tab[bin * spb + r];
      else s = make_sample(g, height, width, bin / PW, bin % PW, r / g.grid_w, r % g.grid_w);
      if (s.p1 < 0) continue;
      atomicAdd(gin + s.p1 * cstride, gval * s.w1 / count);
      atomicAdd(gin + s.p2 * cstride, gval * s.w2 / count);
      atomicAdd(gin + s.p3 * cstride, gval * s.w3 / count);
      atomicAdd(gin + s.p4 * cstride, gval * s.w4 / count);
    }
  }
}


// ---------------------------------------------------------------- RoIAlignRotated backward, TILED (round 4)
// The kernel above scatters every (RoI, bin, sample, corner) contribution with a global fp32 atomic: 410 MB of atomic
// traffic for 512 RoIs on the 256 x 256 x 256 level against 160 MB of algorithmic bytes, 0.05 of the HBM roofline.  The
// tiled form inverts the loop into a GATHER: a counting sort files every corner contribution under the PIXEL it lands on
// (key = 8 x 8-pixel tile * 64 + pixel inside the tile; entry = RoI-bin index, bilinear weight / count), and one wave per
// 16 pixels of a tile walks their entries in pixel order with the lanes along the channels: an entry is one coalesced
// 1 KiB read of the bin's channel vector from the TRANSPOSED gradient (n, bins, C), a pixel's sum lives in registers and
// is added to grad_input once -- no atomic of any kind (a first version accumulated unsorted entries of a tile with
// ds_add_f32: 0.77 ms, 0.74 of them the LDS float atomics; its counting sort on 2048 tile counters spent 0.11 ms per
// pass on same-line global atomics, 0.02 ms on the 131 k pixel keys), and `grad_input +=` semantics (the reference's
// atomicAdd form) are kept because
// every pixel belongs to exactly one workgroup.  NHWC maps, sampling_ratio > 0.  Same samples and weights as the scatter
// kernels (make_sample); the sum order inside a pixel differs (as it does between two runs of the atomic form): covered
// by the same 1e-4 tolerance.
constexpr int RT = 8;                 // tile edge in pixels
constexpr int RT_PX = RT * RT;
constexpr int RT_CH = 256;            // channels per workgroup (a float4 per lane)
struct RoiTileLevels {
  float* gin[ROI_MAX_LEVELS];
  int h[ROI_MAX_LEVELS], w[ROI_MAX_LEVELS], tiles_x[ROI_MAX_LEVELS], tiles_y[ROI_MAX_LEVELS];
  int tile_base[ROI_MAX_LEVELS + 1];  // first global tile index of each level (batch-major inside a level)
  float scale[ROI_MAX_LEVELS];
  int n;
  float finest;
};
struct RoiEntry {
  int rb;    // roi * bins + bin
  float w;   // bilinear weight / samples per bin
};

// PASS 0: count the entries of every (tile, pixel) key; PASS 1: file them (offsets from the scan, cursors zeroed).  One
// workgroup per RoI: its geometry (two double-precision sincos) is computed once and shared.
template <int PASS>
__global__ __launch_bounds__(256) void roi_bwd_bin_kernel(const float* __restrict__ rois, int n_rois, int PH, int PW,
                                                         int sampling_ratio, int aligned, int clockwise, RoiTileLevels lv,
                                                         int* __restrict__ counts, const int* __restrict__ offsets,
                                                         RoiEntry* __restrict__ entries) {
  const int bins = PH * PW, spb = sampling_ratio * sampling_ratio;
  const int n = blockIdx.x;
  const float* roi = rois + 6 * (size_t)n;
  __shared__ RoiGeom g_sh;
  __shared__ int l_sh;
  if (threadIdx.x == 0) {
    const int l0 = lv.n > 1 ? roi_target_level(roi, lv.finest, lv.n) : 0;
    l_sh = l0;
    g_sh = roi_geometry(roi, lv.scale[l0], aligned, clockwise, PH, PW, sampling_ratio);
  }
  __syncthreads();
  const RoiGeom g = g_sh;
  const int l = l_sh;
  const int H = lv.h[l], W = lv.w[l];
  const float count = (float)spb;
  for (int r0 = threadIdx.x; r0 < bins * spb; r0 += 256) {
    const int bin = r0 / spb, r = r0 - bin * spb;
    const Sample s = make_sample(g, H, W, bin / PW, bin % PW, r / g.grid_w, r % g.grid_w);
    if (s.p1 < 0) continue;
    const int ps[4] = {s.p1, s.p2, s.p3, s.p4};
    const float ws[4] = {s.w1, s.w2, s.w3, s.w4};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int y = ps[c] / W, x = ps[c] - y * W;
      const int tile = lv.tile_base[l] + (g.batch * lv.tiles_y[l] + y / RT) * lv.tiles_x[l] + x / RT;
      const int key = tile * RT_PX + (y % RT) * RT + (x % RT);
      if (PASS == 0) {
        atomicAdd(counts + key, 1);
      } else {
        const int slot = offsets[key] + atomicAdd(counts + key, 1);
        entries[slot] = RoiEntry{n * bins + bin, ws[c] / count};
      }
    }
  }
}

// exclusive scan of the key counts in two launches of nkeys / 1024 workgroups: (1) the sum of every 1024-key block,
// (2) each block adds up the sums of the blocks before it (a few hundred words), scans its own keys and zeroes them
// (they become the cursors of the filing pass).  A single-workgroup scan took 138 us over the 131 k keys of a 256 x 256 map.
constexpr int RS_KEYS = 1024;
__device__ __forceinline__ int block_sum_256(int v, int* red) {
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const int s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}
__global__ __launch_bounds__(256) void roi_bwd_blocksum_kernel(const int* __restrict__ counts, int* __restrict__ bsum, int nkeys) {
  __shared__ int red[4];
  const int i = blockIdx.x * RS_KEYS + threadIdx.x * 4;
  int v = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) v += i + u < nkeys ? counts[i + u] : 0;
  const int s = block_sum_256(v, red);
  if (threadIdx.x == 0) bsum[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void roi_bwd_scan_kernel(int* __restrict__ counts, int* __restrict__ offsets,
                                                          const int* __restrict__ bsum, int nkeys) {
  __shared__ int red[4];
  __shared__ int wsum[4];
  int before = 0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += 256) before += bsum[j];
  const int base = block_sum_256(before, red);
  const int i = blockIdx.x * RS_KEYS + threadIdx.x * 4;
  int c[4], v = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) {
    c[u] = i + u < nkeys ? counts[i + u] : 0;
    v += c[u];
  }
  // inclusive scan of the per-thread sums: inside the wave by shuffles, across the four waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int n = __shfl_up(inc, o);
    if (lane >= o) inc += n;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int run = base + inc - v;
  for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
  for (int u = 0; u < 4; u++)
    if (i + u < nkeys) {
      offsets[i + u] = run;
      counts[i + u] = 0;
      run += c[u];
    }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) offsets[nkeys] = run;
}

// one workgroup per (tile, 256-channel chunk), one WAVE per 16 pixels of the tile, a lane per four channels: an entry is
// one 1 KiB wave read of the bin's channel vector.  A wave pulls 64 entries at a time into registers (lane i holds entry
// i; v_readlane hands them round as scalars), keeps 16 vector reads in flight, carries the running sum of the current
// pixel in registers and adds it to grad_input once per pixel -- a plain read-add-write, the pixel is this wave's alone.
__global__ __launch_bounds__(256) void roi_bwd_tile_kernel(const float* __restrict__ goT, const int* __restrict__ offsets,
                                                          const RoiEntry* __restrict__ entries, RoiTileLevels lv,
                                                          int channels, int overwrite) {
  __shared__ int po[RT_PX + 1];
  const int t = blockIdx.x;
  if (threadIdx.x <= RT_PX) po[threadIdx.x] = offsets[(size_t)t * RT_PX + threadIdx.x];
  __syncthreads();
  if (!overwrite && po[RT_PX] == po[0]) return;  // nothing lands here: grad_input keeps its values
  int l = 0;
  while (l + 1 < lv.n && t >= lv.tile_base[l + 1]) l++;
  const int tl = t - lv.tile_base[l];
  const int tx = tl % lv.tiles_x[l], t2 = tl / lv.tiles_x[l];
  const int ty = t2 % lv.tiles_y[l], b = t2 / lv.tiles_y[l];
  const int H = lv.h[l], W = lv.w[l];
  float* gin = lv.gin[l] + (size_t)b * H * W * channels;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p0 = wave * (RT_PX / 4);
  const int C4 = channels >> 2;
  const int c4 = blockIdx.y * 64 + lane;
  const bool lane_on = c4 < C4;
  const float4* gcol = (const float4*)goT + (lane_on ? c4 : 0);
  const int e0 = __builtin_amdgcn_readfirstlane(po[p0]), e1 = __builtin_amdgcn_readfirstlane(po[p0 + RT_PX / 4]);
  int cur = -1;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  auto pixel = [&](int px) {
    const int y = ty * RT + px / RT, x = tx * RT + px % RT;
    return y < H && x < W ? (float4*)(gin + ((size_t)y * W + x) * channels) + c4 : nullptr;
  };
  // overwrite mode: the caller's maps are NOT zero-filled; every in-bounds pixel is written, zeros where nothing lands
  auto zeros = [&](int from, int to) {
    if (overwrite && lane_on)
      for (int px = from; px < to; px++)
        if (float4* dst = pixel(px)) *dst = make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto flush = [&]() {
    if (cur >= 0 && lane_on) {
      float4* dst = pixel(cur);  // (entries only exist for in-bounds pixels)
      float4 o = overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *dst;
      o.x += acc.x, o.y += acc.y, o.z += acc.z, o.w += acc.w;
      *dst = o;
    }
  };
  for (int base = e0; base < e1; base += 64) {
    const int cnt = min(64, e1 - base);
    RoiEntry a = RoiEntry{0, 0.f};
    if (lane < cnt) a = entries[base + lane];
    int px_v = p0;  // the pixel of my entry: p0 + the number of later pixels of this wave that start at or before it
#pragma unroll
    for (int p = 1; p < RT_PX / 4; p++) px_v += po[p0 + p] <= base + lane;
    const int w_bits = __float_as_int(a.w);
    for (int u0 = 0; u0 < cnt; u0 += 16) {
      float4 g[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int rb = __builtin_amdgcn_readlane(a.rb, u0 + u);  // (lanes >= cnt hold entry 0 of the gradient: in bounds)
        g[u] = gcol[(size_t)rb * C4];
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (u0 + u < cnt) {
          const int px = __builtin_amdgcn_readlane(px_v, u0 + u);
          const float w = __int_as_float(__builtin_amdgcn_readlane(w_bits, u0 + u));
          if (px != cur) {
            flush();
            zeros(cur < 0 ? p0 : cur + 1, px);
            cur = px;
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          acc.x += g[u].x * w, acc.y += g[u].y * w, acc.z += g[u].z * w, acc.w += g[u].w * w;
        }
      }
    }
  }
  flush();
  zeros(cur < 0 ? p0 : cur + 1, p0 + RT_PX / 4);
}

// ---------------------------------------------------------------- MaxIoU assignment
// mmdet MaxIoUAssigner.assign_wrt_overlaps (the assigner of `rpn` and `rcnn` in local_configs/main_SM3Det.py:165-196,
// called from oriented_rpn_head.py:76-78 and oriented_standard_roi_head.py:68-70) without the (k x n) overlap matrix:
// one thread per box walks the k ground-truth boxes.  rotated = 1: boxes (cx,cy,w,h,a), IoU of RBboxOverlaps2D =
// rbbox_overlaps (rotate_iou2d_calculator.py:52-87: w,h clamped to >= 1e-3, then box_iou_rotated); rotated = 0: boxes
// (x1,y1,x2,y2), mmdet bbox_overlaps 'iou' (eps 1e-6).
// The rotated IoU is ONE out-of-line function: pass 2 finds a gt's low-quality matches by recomputing the IoU and testing
// it for equality with the maximum pass 1 published, so both kernels must execute the very same instruction sequence (two
// inlined copies may be scheduled / simplified differently by the compiler).
__device__ __noinline__ float assign_iou_rotated(const float* __restrict__ g, const float* __restrict__ b, Pt* lds_pts,
                                                 uint32_t* lds_stack) {
  const Box5 gg{g[0], g[1], fmaxf(g[2], 1e-3f), fmaxf(g[3], 1e-3f), g[4]};
  const Box5 bb{b[0], b[1], fmaxf(b[2], 1e-3f), fmaxf(b[3], 1e-3f), b[4]};
  return single_box_iou_rotated(gg, bb, 0, IouScratch{lds_pts, lds_stack});
}

// ROT: the rotated instantiation carries the polygon scratch in LDS (57 KB per workgroup); the horizontal one none
template <int ROT>
__device__ __forceinline__ float assign_iou(const float* __restrict__ g, const float* __restrict__ b, const IouScratch& q) {
  if (ROT) return assign_iou_rotated(g, b, q.pts, q.stack);
  const float a1 = (g[2] - g[0]) * (g[3] - g[1]);
  const float a2 = (b[2] - b[0]) * (b[3] - b[1]);
  const float w = fmaxf(fminf(g[2], b[2]) - fmaxf(g[0], b[0]), 0.f);
  const float h = fmaxf(fminf(g[3], b[3]) - fmaxf(g[1], b[1]), 0.f);
  const float ov = w * h;
  const float uni = fmaxf(a1 + a2 - ov, 1e-6f);
  return ov / uni;
}

// pass 1: max_ov[j] = max_i iou(gt_i, box_j), argmax[j] = first i reaching it; gt_max_bits[i] = max_j (as ordered
// bits: IoUs are >= 0, so the unsigned order of the bit patterns is the float order).  gt_max_bits zeroed by the caller.
template <int ROT>
__global__ __launch_bounds__(256) void max_iou_pass1_kernel(const float* __restrict__ boxes, int box_stride, int n,
                                                           const float* __restrict__ gts, int gt_stride, int k,
                                                           float* __restrict__ max_ov,
                                                           int32_t* __restrict__ argmax,
                                                           unsigned* __restrict__ gt_max_bits,
                                                           const uint8_t* __restrict__ flags) {
  __shared__ uint32_t iou_lds[ROT ? 4 : 1][ROT ? IOU_LDS_WORDS_PER_WAVE : 1];
  const IouScratch q = ROT ? iou_scratch(iou_lds[threadIdx.x >> 6], threadIdx.x & 63) : IouScratch{nullptr, nullptr};
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (flags && !flags[j]) {  // not a candidate (outside the image): no IoU, no vote for a gt's maximum
    max_ov[j] = 0.f;
    argmax[j] = 0;
    return;
  }
  const float* b = boxes + (long)j * box_stride;
  float best = -1.f;
  int bi = 0;
  for (int i = 0; i < k; i++) {
    const float ov = assign_iou<ROT>(gts + (long)i * gt_stride, b, q);
    if (ov > best) {
      best = ov;
      bi = i;
    }
    // a non-finite IoU (degenerate box with inf / NaN coordinates) has a bit pattern above every finite float: it must not
    // become the gt's maximum and switch off its low-quality matching.  An IoU of 0 cannot raise the zero-initialised
    // maximum: skipping it removes the same-address atomics of the ~98 % of the anchors that miss a given gt (262 k
    // anchors x 8 gts on 8 addresses were 0.3 of the 0.39 ms this operator took)
    if (ov > 0.f && ov <= 2.f) atomicMax(gt_max_bits + i, __float_as_uint(ov));
  }
  max_ov[j] = k > 0 ? best : 0.f;
  argmax[j] = bi;
}

// pass 2: the assignment rule (negatives, positives, low-quality matches in gt order -- a later gt overrides an
// earlier one, as the reference's python loop does), labels of the positives
template <int ROT>
__global__ __launch_bounds__(256) void max_iou_pass2_kernel(const float* __restrict__ boxes, int box_stride, int n,
                                                           const float* __restrict__ gts, int gt_stride, int k,
                                                           const float* __restrict__ max_ov,
                                                           const int32_t* __restrict__ argmax,
                                                           const unsigned* __restrict__ gt_max_bits, float pos_thr,
                                                           float neg_thr, float min_pos, int match_low_quality,
                                                           const int64_t* __restrict__ gt_labels,
                                                           int64_t* __restrict__ gt_inds,
                                                           int64_t* __restrict__ labels,
                                                           const uint8_t* __restrict__ flags) {
  __shared__ uint32_t iou_lds[ROT ? 4 : 1][ROT ? IOU_LDS_WORDS_PER_WAVE : 1];
  const IouScratch q = ROT ? iou_scratch(iou_lds[threadIdx.x >> 6], threadIdx.x & 63) : IouScratch{nullptr, nullptr};
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (flags && !flags[j]) {
    gt_inds[j] = -1;
    if (labels) labels[j] = -1;
    return;
  }
  int a = -1;
  if (k == 0) {
    a = 0;  // no ground truth: everything is background (max_iou_assigner.py: assigned_gt_inds[:] = 0)
  } else {
    const float mo = max_ov[j];
    if (mo >= 0.f && mo < neg_thr) a = 0;
    if (mo >= pos_thr) a = argmax[j] + 1;
    if (match_low_quality) {
      const float* b = boxes + (long)j * box_stride;
      for (int i = 0; i < k; i++) {
        const float gmax = __uint_as_float(gt_max_bits[i]);
        if (gmax >= min_pos && assign_iou<ROT>(gts + (long)i * gt_stride, b, q) == gmax) a = i + 1;
      }
    }
  }
  gt_inds[j] = a;
  if (labels) labels[j] = (a > 0 && gt_labels) ? gt_labels[a - 1] : -1;
}

}  // namespace

// =================================================================================================== C ABI
extern "C" {

size_t sm3_max_iou_assign_workspace_bytes(int n, int k) {
  return align_up((size_t)(n > 0 ? n : 1) * sizeof(int32_t), 256) + (size_t)(k > 0 ? k : 1) * sizeof(unsigned);
}

int sm3_max_iou_assign_masked(const float* boxes, int box_stride, int n, const uint8_t* box_flags, const float* gts,
                              int gt_stride, int k, int rotated, float pos_iou_thr, float neg_iou_thr,
                              float min_pos_iou, int match_low_quality, const int64_t* gt_labels, int64_t* gt_inds,
                              float* max_overlaps, int64_t* labels, void* workspace, size_t workspace_bytes,
                              sm3_stream_t stream) {
  if (n < 0 || k < 0 || (rotated ? (box_stride < 5 || gt_stride < 5) : (box_stride < 4 || gt_stride < 4)))
    return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  if (!boxes || !gt_inds || !max_overlaps || (k > 0 && !gts)) return SM3_ERR_INVALID_ARG;
  const size_t need = sm3_max_iou_assign_workspace_bytes(n, k);
  if (!workspace || workspace_bytes < need) return SM3_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int32_t* argmax = (int32_t*)workspace;
  unsigned* gmax = (unsigned*)((char*)workspace + align_up((size_t)n * sizeof(int32_t), 256));
  sm3_zero_async(gmax, sizeof(unsigned) * (size_t)(k > 0 ? k : 1), st);
  const int blocks = (n + 255) / 256;
  if (rotated) {
    max_iou_pass1_kernel<1><<<blocks, 256, 0, st>>>(boxes, box_stride, n, gts, gt_stride, k, max_overlaps, argmax, gmax,
                                                    box_flags);
    max_iou_pass2_kernel<1><<<blocks, 256, 0, st>>>(boxes, box_stride, n, gts, gt_stride, k, max_overlaps, argmax, gmax,
                                                    pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality, gt_labels,
                                                    gt_inds, labels, box_flags);
  } else {
    max_iou_pass1_kernel<0><<<blocks, 256, 0, st>>>(boxes, box_stride, n, gts, gt_stride, k, max_overlaps, argmax, gmax,
                                                    box_flags);
    max_iou_pass2_kernel<0><<<blocks, 256, 0, st>>>(boxes, box_stride, n, gts, gt_stride, k, max_overlaps, argmax, gmax,
                                                    pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality, gt_labels,
                                                    gt_inds, labels, box_flags);
  }
  return launch_status();
}

int sm3_max_iou_assign(const float* boxes, int box_stride, int n, const float* gts, int gt_stride, int k, int rotated,
                       float pos_iou_thr, float neg_iou_thr, float min_pos_iou, int match_low_quality,
                       const int64_t* gt_labels, int64_t* gt_inds, float* max_overlaps, int64_t* labels,
                       void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  return sm3_max_iou_assign_masked(boxes, box_stride, n, nullptr, gts, gt_stride, k, rotated, pos_iou_thr, neg_iou_thr,
                                   min_pos_iou, match_low_quality, gt_labels, gt_inds, max_overlaps, labels, workspace,
                                   workspace_bytes, stream);
}

int sm3_box_iou_rotated(const float* boxes1, const float* boxes2, float* ious, int n1, int n2, int mode_flag,
                        int aligned, sm3_stream_t stream) {
  if (n1 < 0 || n2 < 0 || (mode_flag != 0 && mode_flag != 1)) return SM3_ERR_INVALID_ARG;
  if (aligned && n1 != n2) return SM3_ERR_INVALID_ARG;
  const long total = aligned ? (long)n1 : (long)n1 * n2;
  if (total == 0) return SM3_OK;
  if (!boxes1 || !boxes2 || !ious) return SM3_ERR_INVALID_ARG;
  // pairs per wave: enough waves to cover the 1024 SIMDs a few times over before the chunks grow
  const int rounds = total >= 64l * 16 * 4096 ? 16 : (total >= 64l * 4 * 2048 ? 4 : 1);
  long blocks = ((total + 64 * rounds - 1) / (64 * rounds) + 3) / 4;  // four waves per workgroup, one chunk per wave
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (rounds == 16) box_iou_rotated_kernel<16><<<(int)blocks, 256, 0, st>>>(boxes1, boxes2, ious, n1, n2, mode_flag, aligned);
  else if (rounds == 4) box_iou_rotated_kernel<4><<<(int)blocks, 256, 0, st>>>(boxes1, boxes2, ious, n1, n2, mode_flag, aligned);
  else box_iou_rotated_kernel<1><<<(int)blocks, 256, 0, st>>>(boxes1, boxes2, ious, n1, n2, mode_flag, aligned);
  return launch_status();
}

size_t sm3_topk_desc_workspace_bytes(int n) { return topk_ws_bytes(n); }

int sm3_topk_desc_f32(const float* scores, int n, int k, int64_t* order, void* workspace, size_t workspace_bytes,
                      sm3_stream_t stream) {
  if (n < 0 || k < 0) return SM3_ERR_INVALID_ARG;
  if (n == 0 || k == 0) return SM3_OK;
  if (!scores || !order || !workspace) return SM3_ERR_INVALID_ARG;
  int rc = topk_desc(scores, n, k, order, workspace, workspace_bytes, (hipStream_t)stream);
  return rc ? rc : launch_status();
}

size_t sm3_argsort_desc_workspace_bytes(int n) { return (size_t)next_pow2(n > 0 ? n : 1) * 8; }

int sm3_argsort_desc_f32(const float* scores, int n, int64_t* order, void* workspace, size_t workspace_bytes,
                         sm3_stream_t stream) {
  if (n < 0) return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  if (!scores || !order || !workspace) return SM3_ERR_INVALID_ARG;
  int rc = argsort_desc(scores, n, order, workspace, workspace_bytes, (hipStream_t)stream);
  return rc ? rc : launch_status();
}

size_t sm3_nms_workspace_bytes(int n) {
  size_t a, b, c, d, e;
  return nms_ws_layout(n > 0 ? n : 1, &a, &b, &c, &d, &e);
}
size_t sm3_nms_rotated_workspace_bytes(int n) { return sm3_nms_workspace_bytes(n); }

static int nms_common(bool rotated, const float* boxes, int stride, const float* scores, const int64_t* order,
                      int n, float thr, int offset, int multi_label, int64_t* keep, int32_t* num_keep,
                      void* ws, size_t ws_bytes, hipStream_t st) {
  if (n < 0 || !num_keep) return SM3_ERR_INVALID_ARG;
  if (n == 0) {
    sm3_zero_async(num_keep, sizeof(int32_t), st);
    return launch_status();
  }
  if (!boxes || !keep || !ws) return SM3_ERR_INVALID_ARG;
  if (rotated && (stride < 5 || (multi_label && stride < 6))) return SM3_ERR_INVALID_ARG;
  if (!rotated && offset != 0 && offset != 1) return SM3_ERR_INVALID_ARG;
  size_t o_order, o_mask, o_remv, o_sort, o_diagt;
  size_t need = nms_ws_layout(n, &o_order, &o_mask, &o_remv, &o_sort, &o_diagt);
  if (ws_bytes < need) return SM3_ERR_WORKSPACE;
  char* w = (char*)ws;
  const int nblk = (n + 63) / 64;
  if (!order) {
    if (!scores) return SM3_ERR_INVALID_ARG;
    int rc = argsort_desc(scores, n, (int64_t*)(w + o_order), w + o_sort, ws_bytes - o_sort, st);
    if (rc) return rc;
    order = (const int64_t*)(w + o_order);
  }
  uint64_t* mask = (uint64_t*)(w + o_mask);
  uint64_t* diagt = (uint64_t*)(w + o_diagt);
  const long ntiles = (long)nblk * (nblk + 1) / 2;
  if (rotated)
    nms_rotated_mask_kernel<<<(int)ntiles, 256, 0, st>>>(boxes, stride, order, n, nblk, thr, multi_label, mask, diagt);
  else
    nms_mask_kernel<<<(int)ntiles, 64, 0, st>>>(boxes, order, n, nblk, thr, (float)offset, mask, diagt);
  // removal vector in LDS, next block row prefetched; the request must stay within the 64 KiB a launch may ask for without
  // hipFuncSetAttribute (N <= 524 032), beyond that the global-memory sweep runs
  if ((size_t)(nblk + 4) * 8 <= 64 * 1024)
    nms_sweep_lds_kernel<<<1, SWEEP_THREADS, (size_t)(nblk + 4) * 8, st>>>(mask, diagt, order, n, nblk, keep, num_keep);
  else
    nms_sweep_kernel<<<1, SWEEP_THREADS, 0, st>>>(mask, order, n, nblk, (uint64_t*)(w + o_remv), keep, num_keep);
  return launch_status();
}

int sm3_nms(const float* boxes, const float* scores, const int64_t* order, int n, float iou_threshold, int offset,
            int64_t* keep, int32_t* num_keep, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  return nms_common(false, boxes, 4, scores, order, n, iou_threshold, offset, 0, keep, num_keep, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

int sm3_nms_rotated(const float* dets, int dets_stride, const float* scores, const int64_t* order, int n,
                    float iou_threshold, int multi_label, int64_t* keep, int32_t* num_keep, void* workspace,
                    size_t workspace_bytes, sm3_stream_t stream) {
  return nms_common(true, dets, dets_stride, scores, order, n, iou_threshold, 0, multi_label, keep, num_keep,
                    workspace, workspace_bytes, (hipStream_t)stream);
}

static size_t roi_lds_bytes(int pooled_h, int pooled_w, int sampling_ratio) {
  // adaptive grids (sampling_ratio <= 0) are data dependent: reserve the cap
  long ns = sampling_ratio > 0 ? (long)pooled_h * pooled_w * sampling_ratio * sampling_ratio
                               : ROI_MAX_LDS_SAMPLES;
  if (ns > ROI_MAX_LDS_SAMPLES) ns = 0;
  return (size_t)(ns > 0 ? ns : 1) * sizeof(Sample);
}

// the channel-vector forward (NHWC): LDS = sample table + the (C x bins) output tile; 0 -> does not fit / not applicable
static size_t roi_vec_lds_bytes(int channels, int ph, int pw, int sampling_ratio, size_t* tab_bytes) {
  if (channels & 3) return 0;
  size_t tb = roi_lds_bytes(ph, pw, sampling_ratio);
  tb = align_up(tb, 16);
  const size_t tile = (size_t)channels * ph * pw * sizeof(float);
  if (tb + tile > 64 * 1024) return 0;  // the default dynamic-LDS limit of a launch; two RoIs per CU
  *tab_bytes = tb;
  return tb + tile;
}

int sm3_roi_align_rotated_forward(const float* input, const float* rois, float* output, int n_rois, int batch,
                                  int channels, int height, int width, int pooled_h, int pooled_w,
                                  float spatial_scale, int sampling_ratio, int aligned, int clockwise, int layout,
                                  sm3_stream_t stream) {
  if (n_rois < 0 || batch < 0 || channels <= 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0 ||
      (layout != 0 && layout != 1))
    return SM3_ERR_INVALID_ARG;
  if (n_rois == 0) return SM3_OK;
  if (!input || !rois || !output) return SM3_ERR_INVALID_ARG;
  size_t lds = roi_lds_bytes(pooled_h, pooled_w, sampling_ratio);
  hipStream_t st = (hipStream_t)stream;
  size_t tb = 0;
  const size_t vlds = layout == 1 ? roi_vec_lds_bytes(channels, pooled_h, pooled_w, sampling_ratio, &tb) : 0;
  if (layout == 0)
    roi_align_rotated_fwd_kernel<0><<<n_rois, ROI_THREADS, lds, st>>>(input, rois, output, channels, height, width,
                                                                      pooled_h, pooled_w, spatial_scale,
                                                                      sampling_ratio, aligned, clockwise);
  else if (vlds)
    roi_align_rotated_fwd_vec_kernel<0><<<n_rois, ROI_THREADS, vlds, st>>>(input, rois, output, channels, height, width,
                                                                           pooled_h, pooled_w, spatial_scale,
                                                                           sampling_ratio, aligned, clockwise, (int)tb);
  else
    roi_align_rotated_fwd_kernel<1><<<n_rois, ROI_THREADS, lds, st>>>(input, rois, output, channels, height, width,
                                                                      pooled_h, pooled_w, spatial_scale,
                                                                      sampling_ratio, aligned, clockwise);
  return launch_status();
}

int sm3_roi_align_rotated_backward(const float* grad_output, const float* rois, float* grad_input, int n_rois,
                                   int batch, int channels, int height, int width, int pooled_h, int pooled_w,
                                   float spatial_scale, int sampling_ratio, int aligned, int clockwise, int layout,
                                   sm3_stream_t stream) {
  if (n_rois < 0 || batch < 0 || channels <= 0 || height <= 0 || width <= 0 || pooled_h <= 0 || pooled_w <= 0 ||
      (layout != 0 && layout != 1))
    return SM3_ERR_INVALID_ARG;
  if (n_rois == 0) return SM3_OK;
  if (!grad_output || !rois || !grad_input) return SM3_ERR_INVALID_ARG;
  size_t lds = roi_lds_bytes(pooled_h, pooled_w, sampling_ratio);
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0)
    roi_align_rotated_bwd_kernel<0><<<n_rois, ROI_THREADS, lds, st>>>(grad_output, rois, grad_input, channels,
                                                                      height, width, pooled_h, pooled_w,
                                                                      spatial_scale, sampling_ratio, aligned,
                                                                      clockwise);
  else
    roi_align_rotated_bwd_kernel<1><<<n_rois, ROI_THREADS, lds, st>>>(grad_output, rois, grad_input, channels,
                                                                      height, width, pooled_h, pooled_w,
                                                                      spatial_scale, sampling_ratio, aligned,
                                                                      clockwise);
  return launch_status();
}


static int fill_levels(RoiLevels& lv, const float* const* inputs, float* const* grad_inputs, const int* heights,
                       const int* widths, const float* scales, int nlev, float finest, int32_t* levels_out) {
  if (nlev < 1 || nlev > ROI_MAX_LEVELS || !heights || !widths || !scales || !(finest > 0.f)) return SM3_ERR_INVALID_ARG;
  for (int i = 0; i < nlev; i++) {
    lv.in[i] = inputs ? inputs[i] : nullptr;
    lv.gin[i] = grad_inputs ? grad_inputs[i] : nullptr;
    if ((inputs && !inputs[i]) || (grad_inputs && !grad_inputs[i]) || heights[i] <= 0 || widths[i] <= 0)
      return SM3_ERR_INVALID_ARG;
    lv.h[i] = heights[i];
    lv.w[i] = widths[i];
    lv.scale[i] = scales[i];
  }
  lv.n = nlev;
  lv.finest = finest;
  lv.levels_out = levels_out;
  return SM3_OK;
}

int sm3_roi_align_rotated_multilevel_forward(const float* const* inputs, const int* heights, const int* widths,
                                             const float* scales, int num_levels, float finest_scale,
                                             const float* rois, float* output, int32_t* levels_out, int n_rois,
                                             int channels, int pooled_h, int pooled_w, int sampling_ratio,
                                             int aligned, int clockwise, int layout, sm3_stream_t stream) {
  if (n_rois < 0 || channels <= 0 || pooled_h <= 0 || pooled_w <= 0 || (layout != 0 && layout != 1) || !inputs)
    return SM3_ERR_INVALID_ARG;
  if (n_rois == 0) return SM3_OK;
  if (!rois || !output) return SM3_ERR_INVALID_ARG;
  RoiLevels lv;
  int rc = fill_levels(lv, inputs, nullptr, heights, widths, scales, num_levels, finest_scale, levels_out);
  if (rc) return rc;
  size_t lds = roi_lds_bytes(pooled_h, pooled_w, sampling_ratio);
  hipStream_t st = (hipStream_t)stream;
  size_t tb = 0;
  const size_t vlds = layout == 1 ? roi_vec_lds_bytes(channels, pooled_h, pooled_w, sampling_ratio, &tb) : 0;
  if (layout == 0)
    roi_align_rotated_fwd_kernel<0, 1><<<n_rois, ROI_THREADS, lds, st>>>(nullptr, rois, output, channels, 0, 0,
                                                                         pooled_h, pooled_w, 0.f, sampling_ratio,
                                                                         aligned, clockwise, lv);
  else if (vlds)
    roi_align_rotated_fwd_vec_kernel<1><<<n_rois, ROI_THREADS, vlds, st>>>(nullptr, rois, output, channels, 0, 0, pooled_h,
                                                                           pooled_w, 0.f, sampling_ratio, aligned,
                                                                           clockwise, (int)tb, lv);
  else
    roi_align_rotated_fwd_kernel<1, 1><<<n_rois, ROI_THREADS, lds, st>>>(nullptr, rois, output, channels, 0, 0,
                                                                         pooled_h, pooled_w, 0.f, sampling_ratio,
                                                                         aligned, clockwise, lv);
  return launch_status();
}

int sm3_roi_align_rotated_multilevel_backward(const float* grad_output, const float* rois,
                                              float* const* grad_inputs, const int* heights, const int* widths,
                                              const float* scales, int num_levels, float finest_scale, int n_rois,
                                              int channels, int pooled_h, int pooled_w, int sampling_ratio,
                                              int aligned, int clockwise, int layout, sm3_stream_t stream) {
  if (n_rois < 0 || channels <= 0 || pooled_h <= 0 || pooled_w <= 0 || (layout != 0 && layout != 1) || !grad_inputs)
    return SM3_ERR_INVALID_ARG;
  if (n_rois == 0) return SM3_OK;
  if (!rois || !grad_output) return SM3_ERR_INVALID_ARG;
  RoiLevels lv;
  int rc = fill_levels(lv, nullptr, grad_inputs, heights, widths, scales, num_levels, finest_scale, nullptr);
  if (rc) return rc;
  size_t lds = roi_lds_bytes(pooled_h, pooled_w, sampling_ratio);
  hipStream_t st = (hipStream_t)stream;
  if (layout == 0)
    roi_align_rotated_bwd_kernel<0, 1><<<n_rois, ROI_THREADS, lds, st>>>(grad_output, rois, nullptr, channels, 0, 0,
                                                                         pooled_h, pooled_w, 0.f, sampling_ratio,
                                                                         aligned, clockwise, lv);
  else
    roi_align_rotated_bwd_kernel<1, 1><<<n_rois, ROI_THREADS, lds, st>>>(grad_output, rois, nullptr, channels, 0, 0,
                                                                         pooled_h, pooled_w, 0.f, sampling_ratio,
                                                                         aligned, clockwise, lv);
  return launch_status();
}


static int fill_tile_levels(RoiTileLevels& tl, float* const* grad_inputs, const int* heights, const int* widths,
                            const float* scales, int nlev, float finest, int batch) {
  if (nlev < 1 || nlev > ROI_MAX_LEVELS || !heights || !widths || !scales || batch <= 0) return -1;
  int base = 0;
  for (int i = 0; i < nlev; i++) {
    if (heights[i] <= 0 || widths[i] <= 0 || (grad_inputs && !grad_inputs[i])) return -1;
    tl.gin[i] = grad_inputs ? grad_inputs[i] : nullptr;
    tl.h[i] = heights[i];
    tl.w[i] = widths[i];
    tl.tiles_x[i] = (widths[i] + RT - 1) / RT;
    tl.tiles_y[i] = (heights[i] + RT - 1) / RT;
    tl.scale[i] = scales[i];
    tl.tile_base[i] = base;
    base += batch * tl.tiles_x[i] * tl.tiles_y[i];
  }
  tl.tile_base[nlev] = base;
  tl.n = nlev;
  tl.finest = finest;
  return base;
}

size_t sm3_roi_align_rotated_backward_tiled_workspace_bytes(int n_rois, int batch, int channels, int pooled_h, int pooled_w,
                                                            int sampling_ratio, const int* heights, const int* widths,
                                                            int num_levels) {
  if (n_rois <= 0 || batch <= 0 || channels <= 0 || pooled_h <= 0 || pooled_w <= 0 || sampling_ratio <= 0 || !heights ||
      !widths || num_levels < 1 || num_levels > ROI_MAX_LEVELS)
    return 0;
  long tiles = 0;
  for (int i = 0; i < num_levels; i++)
    tiles += (long)batch * ((heights[i] + RT - 1) / RT) * ((widths[i] + RT - 1) / RT);
  const long bins = (long)pooled_h * pooled_w;
  const long ents = (long)n_rois * bins * sampling_ratio * sampling_ratio * 4;
  return align_up((size_t)(2 * tiles * RT_PX + 2 + tiles * RT_PX / RS_KEYS + 1) * sizeof(int), 256) +
         align_up((size_t)ents * sizeof(RoiEntry), 256) +
         align_up((size_t)n_rois * bins * channels * sizeof(float), 256);
}

int sm3_roi_align_rotated_backward_tiled(const float* grad_output, const float* rois, float* const* grad_inputs,
                                         const int* heights, const int* widths, const float* scales, int num_levels,
                                         float finest_scale, int n_rois, int batch, int channels, int pooled_h,
                                         int pooled_w, int sampling_ratio, int aligned, int clockwise, int overwrite,
                                         void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (n_rois < 0 || batch <= 0 || channels <= 0 || pooled_h <= 0 || pooled_w <= 0 || !grad_inputs) return SM3_ERR_INVALID_ARG;
  if (sampling_ratio <= 0 || channels % 4) return SM3_ERR_UNSUPPORTED;  // adaptive grids, odd widths: the scatter form
  if (n_rois == 0) {
    if (overwrite)
      for (int i = 0; i < num_levels; i++)
        sm3_zero_async(grad_inputs[i], (size_t)batch * heights[i] * widths[i] * channels * sizeof(float), (hipStream_t)stream);
    return overwrite ? launch_status() : SM3_OK;
  }
  if (!rois || !grad_output || (num_levels > 1 && !(finest_scale > 0.f))) return SM3_ERR_INVALID_ARG;
  RoiTileLevels tl;
  const int ntiles = fill_tile_levels(tl, grad_inputs, heights, widths, scales, num_levels, finest_scale, batch);
  if (ntiles <= 0) return SM3_ERR_INVALID_ARG;
  const size_t need = sm3_roi_align_rotated_backward_tiled_workspace_bytes(n_rois, batch, channels, pooled_h, pooled_w,
                                                                          sampling_ratio, heights, widths, num_levels);
  if (!workspace || workspace_bytes < need) return SM3_ERR_WORKSPACE;
  if (n_rois > 65535) return SM3_ERR_UNSUPPORTED;  // batch dimension of the transpose launch
  hipStream_t st = (hipStream_t)stream;
  const long bins = (long)pooled_h * pooled_w;
  const long ents = (long)n_rois * bins * sampling_ratio * sampling_ratio * 4;
  char* w = (char*)workspace;
  const long nkeys = (long)ntiles * RT_PX;
  if (nkeys > 0x7ffffff0l) return SM3_ERR_UNSUPPORTED;
  int* counts = (int*)w;
  int* offsets = counts + nkeys;
  int* bsum = offsets + nkeys + 1;
  const int nblocks = (int)((nkeys + RS_KEYS - 1) / RS_KEYS);
  w += align_up((size_t)(2 * nkeys + 2 + nkeys / RS_KEYS + 1) * sizeof(int), 256);
  RoiEntry* entries = (RoiEntry*)w;
  w += align_up((size_t)ents * sizeof(RoiEntry), 256);
  float* goT = (float*)w;
  sm3_zero_async(counts, (size_t)nkeys * sizeof(int), st);
  // (n, C, bins) -> (n, bins, C): an entry then reads one contiguous channel vector
  int rc = sm3_transpose_f32(grad_output, goT, n_rois, channels, (int)bins, stream);
  if (rc) return rc;
  const int nb = n_rois;  // one workgroup per RoI
  roi_bwd_bin_kernel<0><<<nb, 256, 0, st>>>(rois, n_rois, pooled_h, pooled_w, sampling_ratio, aligned, clockwise, tl, counts,
                                            nullptr, nullptr);
  roi_bwd_blocksum_kernel<<<nblocks, 256, 0, st>>>(counts, bsum, (int)nkeys);
  roi_bwd_scan_kernel<<<nblocks, 256, 0, st>>>(counts, offsets, bsum, (int)nkeys);
  roi_bwd_bin_kernel<1><<<nb, 256, 0, st>>>(rois, n_rois, pooled_h, pooled_w, sampling_ratio, aligned, clockwise, tl, counts,
                                            offsets, entries);
  dim3 tgrid((unsigned)ntiles, (unsigned)((channels + RT_CH - 1) / RT_CH));
  roi_bwd_tile_kernel<<<tgrid, 256, 0, st>>>(goT, offsets, entries, tl, channels, overwrite != 0);
  return launch_status();
}

}  // extern "C"
