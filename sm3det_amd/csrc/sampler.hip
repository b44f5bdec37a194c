// sampler.hip -- RandomSampler / RRandomSampler (mmdet core/bbox/samplers/random_sampler.py as the reference configs use it,
// mmrotate/core/bbox/samplers/rotate_random_sampler.py:10) as a fixed-size, sync-free selection on the device.
//
// The rule (sm3det_amd/assign.py `RandomSampler.sample_fixed`, which restates `_sample_pos` / `_sample_neg`): with one
// uniform random key per candidate, take the min(int(num * pos_fraction), #positives) positives with the SMALLEST keys
// (a uniformly random subset in a uniformly random order -- what `randperm(len(candidates))[:k]` draws), then the
// negatives with the smallest keys until `num` slots are filled (optionally capped at neg_pos_ub * max(n_pos, 1)).
//
// The host form sorts all n keys twice (n = 261 888 anchors in the RPN stage: two radix sorts of ~10 launches each) and
// then spends ~25 elementwise launches on the slot arithmetic.  Here: (1) one pass counts positives and negatives;
// (2) one pass appends the candidates whose key lies below a threshold tau = (m + 6 sqrt(m) + 24) / count to a short list
// (m = the number wanted: the list then holds m + O(sqrt m) entries, and fewer than m with probability < 1e-9);
// (3) one workgroup sorts the two lists in LDS by (key, index) and writes the slots.  If a list comes up short or
// overflows, the same workgroup rebuilds it with another threshold (doubling, then bisection; a scan of all n by one
// workgroup: slow, exact, practically never taken), so the result is the exact m smallest keys -- identical to the host
// form on the same keys.
#include "common.h"

namespace {

constexpr int SP_CAP = 4096;      // list capacity per class (power of two: the LDS bitonic sort's size)
constexpr int SP_THREADS = 1024;  // emit kernel
struct SpEntry {
  float key;
  int idx;
};
struct SpWork {
  int counts[2];   // positives, negatives among the n candidates
  int listed[2];   // entries appended to the lists
  SpEntry list[2][SP_CAP];
};

__device__ __forceinline__ int wanted_pos(int P, int exp_pos) { return min(P, exp_pos); }
__device__ __forceinline__ int wanted_neg(int N, int m_pos, int num, float neg_pos_ub) {
  int m = min(N, num - m_pos);
  if (neg_pos_ub >= 0.f) m = min(m, (int)(long)(neg_pos_ub * (float)max(m_pos, 1)));
  return max(m, 0);
}
__device__ __forceinline__ float threshold(int m, int count) {
  if (m <= 0) return -1.f;            // nothing wanted: keys are >= 0
  const float want = (float)m + 6.f * sqrtf((float)m) + 24.f;
  return want >= (float)count ? 2.f : want / (float)count;  // keys are < 1: 2 takes every candidate
}

__global__ __launch_bounds__(256) void sampler_count_kernel(const int64_t* __restrict__ gt_inds, int n, SpWork* w) {
  int p = 0, q = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t g = gt_inds[i];
    p += g > 0;
    q += g == 0;
  }
  for (int o = 32; o; o >>= 1) {
    p += __shfl_xor(p, o);
    q += __shfl_xor(q, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (p) atomicAdd(&w->counts[0], p);
    if (q) atomicAdd(&w->counts[1], q);
  }
}

__global__ __launch_bounds__(256) void sampler_list_kernel(const int64_t* __restrict__ gt_inds, const float* __restrict__ key,
                                                          int n, int num, int exp_pos, float neg_pos_ub, SpWork* w) {
  const int P = w->counts[0], N = w->counts[1];
  const int m_pos = wanted_pos(P, exp_pos), m_neg = wanted_neg(N, m_pos, num, neg_pos_ub);
  const float tp = threshold(m_pos, P), tn = threshold(m_neg, N);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int64_t g = gt_inds[i];
    if (g < 0) continue;
    const int c = g > 0 ? 0 : 1;
    const float k = key[i];
    if (k < (c ? tn : tp)) {
      const int slot = atomicAdd(&w->listed[c], 1);
      if (slot < SP_CAP) w->list[c][slot] = SpEntry{k, i};
    }
  }
}

// (key, index) ascending
__device__ __forceinline__ bool before(const SpEntry& a, const SpEntry& b) { return a.key < b.key || (a.key == b.key && a.idx < b.idx); }

__global__ __launch_bounds__(SP_THREADS) void sampler_emit_kernel(const int64_t* __restrict__ gt_inds,
                                                                 const float* __restrict__ key, int n, int num, int exp_pos,
                                                                 float neg_pos_ub, SpWork* w, int64_t* __restrict__ idx_out,
                                                                 uint8_t* __restrict__ is_pos_out, uint8_t* __restrict__ valid_out,
                                                                 int64_t* __restrict__ n_pos_out, int64_t* __restrict__ n_neg_out) {
  __shared__ SpEntry s[SP_CAP];
  __shared__ int relisted, got_pos, got_neg;
  const int P = w->counts[0], N = w->counts[1];
  const int m_pos = wanted_pos(P, exp_pos), m_neg = wanted_neg(N, m_pos, num, neg_pos_ub);
  for (int c = 0; c < 2; c++) {
    const int m = c ? m_neg : m_pos;
    const int count = c ? N : P;
    // Slow path, same workgroup: the list holds fewer than m entries (probability < 1e-9 per call for uniform keys) or
    // overflowed (keys that are not uniform).  Re-list with another threshold -- doubled while no upper bound is known,
    // then bisected between "too few" and "too many" -- until m <= entries <= SP_CAP.  A scan of all n by one workgroup:
    // slow, exact, practically never taken.  (Only SP_CAP - m candidates sharing one key bit for bit can defeat the
    // bisection; the last round then lists the "too many" side and the truncated list is used: still m valid samples.)
    int listed = w->listed[c];
    float tau = threshold(m, count), lo = -1.f, hi = 3.f;
    bool ok = listed >= m && listed <= SP_CAP;
    if (!ok) (listed < m ? lo : hi) = tau;
    for (int it = 0; !ok && it < 64; it++) {
      if (hi > 2.5f) tau = tau >= 0.5f ? 2.f : fmaxf(tau * 2.f, 1e-6f);
      else tau = 0.5f * (fmaxf(lo, 0.f) + hi);
      if (it == 63) tau = hi > 2.5f ? 2.f : hi;  // give up separating: the "too many" side, truncated (>= m valid entries)
      if (threadIdx.x == 0) relisted = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < n; i += SP_THREADS) {
        const int64_t g = gt_inds[i];
        if (g >= 0 && (g > 0 ? 0 : 1) == c && key[i] < tau) {
          const int slot = atomicAdd(&relisted, 1);
          if (slot < SP_CAP) w->list[c][slot] = SpEntry{key[i], i};
        }
      }
      __threadfence_block();
      __syncthreads();
      listed = relisted;
      __syncthreads();
      if (listed < m) lo = tau;
      else if (listed > SP_CAP) hi = tau;
      else ok = true;
    }
    const int have = min(listed, SP_CAP);
    // sort the smallest power of two >= have entries in LDS (padding sorts last)
    int size = 64;
    while (size < have) size <<= 1;
    for (int i = threadIdx.x; i < size; i += SP_THREADS) s[i] = i < have ? w->list[c][i] : SpEntry{3.f, 0x7fffffff};
    __syncthreads();
    for (int k = 2; k <= size; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < size; i += SP_THREADS) {
          const int l = i ^ j;
          if (l > i) {
            const SpEntry a = s[i], b = s[l];
            const bool up = (i & k) == 0;
            if (up ? before(b, a) : before(a, b)) {
              s[i] = b;
              s[l] = a;
            }
          }
        }
        __syncthreads();
      }
    // positives fill the slots [0, m_pos), negatives [m_pos, m_pos + m_neg).  Only entries that exist are handed out: with
    // consistent counters have >= m always; should the scratch ever hold stale counters (a capture that recorded its
    // zero-fill but never replayed it), the surplus slots are marked invalid instead of carrying a padding index
    const int base = c ? m_pos : 0;
    const int got_c = min(m, have);
    for (int i = threadIdx.x; i < m; i += SP_THREADS) {
      const bool real = i < got_c;
      idx_out[base + i] = real ? s[i].idx : 0;
      is_pos_out[base + i] = real && c == 0;
      valid_out[base + i] = real;
    }
    if (threadIdx.x == 0) (c ? got_neg : got_pos) = got_c;
    __syncthreads();
  }
  // unused slots: what the host form leaves there (the clamped lookup of the negative order): any in-range index, flags 0
  for (int i = m_pos + m_neg + threadIdx.x; i < num; i += SP_THREADS) {
    idx_out[i] = 0;
    is_pos_out[i] = 0;
    valid_out[i] = 0;
  }
  if (threadIdx.x == 0) {
    *n_pos_out = got_pos;  // what was actually written (= m_pos / m_neg)
    *n_neg_out = got_neg;
    w->counts[0] = w->counts[1] = w->listed[0] = w->listed[1] = 0;  // ready for the next call
  }
}

// one thread per sampled slot: the RoI row, its class label, its matched gt (OrientedStandardRoIHead.forward_train's
// `bbox_results` inputs + RotatedBBoxHead.get_targets' per-sample label / gt, oriented_standard_roi_head.py:66-92,
// rotated_bbox_head.py:131-204) for one image, written at slot offset `out0` of the batch-wide blocks
__global__ __launch_bounds__(256) void rcnn_gather_samples_kernel(
    const float* __restrict__ gts, const int64_t* __restrict__ gt_labels, int k, int prepended,
    const float* __restrict__ props, int ld_props, const int64_t* __restrict__ gt_inds_all,
    const int64_t* __restrict__ labels, const int64_t* __restrict__ idx, const uint8_t* __restrict__ is_pos,
    const uint8_t* __restrict__ valid, int S, int num_classes, float batch_index, long out0, float* __restrict__ rois_out,
    int64_t* __restrict__ labels_out, float* __restrict__ gts_out, uint8_t* __restrict__ valid_out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= S) return;
  const long j = idx[s];
  const bool v = valid[s] != 0, pos = is_pos[s] != 0;
  const int kp = prepended ? k : 0;
  float box[5] = {0.f, 0.f, 1.f, 1.f, 0.f};  // unused slots: a harmless unit box (their rows carry no weight)
  if (v) {
    const float* src = j < kp ? gts + 5 * j : props + (long)ld_props * (j - kp);
#pragma unroll
    for (int c = 0; c < 5; c++) box[c] = src[c];
  }
  float* r = rois_out + 6 * (out0 + s);
  r[0] = batch_index;
#pragma unroll
  for (int c = 0; c < 5; c++) r[1 + c] = box[c];
  float* go = gts_out + 5 * (out0 + s);
  if (k > 0) {
    const long gi = max(gt_inds_all[j] - 1, (long)0);
#pragma unroll
    for (int c = 0; c < 5; c++) go[c] = gts[5 * gi + c];
  } else {
#pragma unroll
    for (int c = 0; c < 5; c++) go[c] = 0.f;
  }
  int64_t lab = num_classes;
  if (pos) lab = j < kp ? gt_labels[j] : labels[j - kp];
  labels_out[out0 + s] = lab;
  valid_out[out0 + s] = v;
}

}  // namespace

extern "C" {

// gts (k,5), gt_labels (k), props (n, ld_props >= 5), gt_inds_all (prepended ? k + n : n) as the sampler saw them, labels (n)
// from the assigner, idx / is_pos / valid (S) from sm3_random_sample_fixed; outputs are batch-wide blocks, this image's
// rows start at slot out0.
int sm3_rcnn_gather_samples(const float* gts, const int64_t* gt_labels, int k, int prepended, const float* props,
                            int ld_props, const int64_t* gt_inds_all, const int64_t* labels, const int64_t* idx,
                            const uint8_t* is_pos, const uint8_t* valid, int S, int num_classes, float batch_index, long out0,
                            float* rois_out, int64_t* labels_out, float* gts_out, uint8_t* valid_out, sm3_stream_t stream) {
  if (S <= 0 || k < 0 || ld_props < 5 || !idx || !is_pos || !valid || !rois_out || !labels_out || !gts_out || !valid_out ||
      !gt_inds_all)
    return SM3_ERR_INVALID_ARG;
  if (k > 0 && (!gts || !gt_labels)) return SM3_ERR_INVALID_ARG;
  rcnn_gather_samples_kernel<<<(S + 255) / 256, 256, 0, (hipStream_t)stream>>>(
      gts, gt_labels, k, prepended, props, ld_props, gt_inds_all, labels, idx, is_pos, valid, S, num_classes, batch_index, out0,
      rois_out, labels_out, gts_out, valid_out);
  return launch_status();
}

size_t sm3_random_sample_workspace_bytes(void) { return sizeof(SpWork); }

// workspace: sm3_random_sample_workspace_bytes() bytes whose first 16 are ZERO on entry (the call leaves them zero: a
// buffer zeroed once can be reused by every later call on the same stream).
int sm3_random_sample_fixed(const int64_t* gt_inds, const float* key, int n, int num, int exp_pos, float neg_pos_ub,
                            int64_t* idx_out, uint8_t* is_pos_out, uint8_t* valid_out, int64_t* n_pos_out,
                            int64_t* n_neg_out, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (n < 0 || num <= 0 || exp_pos < 0 || exp_pos > num || !idx_out || !is_pos_out || !valid_out || !n_pos_out || !n_neg_out)
    return SM3_ERR_INVALID_ARG;
  if (num > SP_CAP / 2) return SM3_ERR_UNSUPPORTED;  // the lists hold the wanted count + its 6 sigma margin
  if (!workspace || workspace_bytes < sizeof(SpWork)) return SM3_ERR_WORKSPACE;
  if (n > 0 && (!gt_inds || !key)) return SM3_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  SpWork* w = (SpWork*)workspace;
  // the four counters are zeroed by every call (a kernel node: ordered under capture and replay), not only left zero by the
  // previous one: the scratch may have been allocated inside a capture whose zero-fill never ran
  sm3_zero_async(w, 16, st);
  if (n > 0) {
    const int blocks = min((n + 255) / 256, 1024);
    sampler_count_kernel<<<blocks, 256, 0, st>>>(gt_inds, n, w);
    sampler_list_kernel<<<blocks, 256, 0, st>>>(gt_inds, key, n, num, exp_pos, neg_pos_ub, w);
  }
  sampler_emit_kernel<<<1, SP_THREADS, 0, st>>>(gt_inds, key, n, num, exp_pos, neg_pos_ub, w, idx_out, is_pos_out, valid_out,
                                               n_pos_out, n_neg_out);
  return launch_status();
}

}  // extern "C"
