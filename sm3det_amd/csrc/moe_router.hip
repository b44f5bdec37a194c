// moe_router.hip -- cosine top-k router of the grid-level sparse MoE, forward and backward, for gfx950.
//
// Reference semantics (mmrotate/models/backbones/convnext_moe.py): CosineTopKGate.forward :99-106,
// MoE_layer.noisy_top_k_gating :194-223, _prob_in_top_k :152-174, _gates_to_load :149-150.
//
// Work split: one workgroup = one wavefront = 16 tokens, FOUR lanes per token.  The 16 x P tile of projected features
// is brought into LDS with coalesced 16-byte loads (row stride P+1 floats), the similarity matrix sits next to it; the
// four lanes of a token split the O(P*E) loops over P (p = sub, sub+4, ...) and meet through two xor-shuffles, the O(E)
// routing arithmetic (noise, top-(k+1), softmax, Normal-CDF load) runs redundantly in all four.  The backward builds
// its dh rows in the same LDS tile and streams them out coalesced.  History: a thread-per-token version reading strided
// rows from global memory took 158 us at T = 8192; one token per lane (64 per wave) left the stage-2 / stage-3 launches
// with 128 / 32 wavefronts on 256 CUs (47 / 72 us per launch, pure latency); four lanes per token quadruple the waves.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float normal_cdf(float z) { return 0.5f * (1.0f + erff(z * 0.70710678118654752440f)); }
__device__ __forceinline__ float normal_pdf(float z) { return 0.39894228040143267794f * __expf(-0.5f * z * z); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }  // torch threshold 20

constexpr int RT_THREADS = 64;  // one wavefront per workgroup
#ifndef SM3_ROUTER_LPT
#define SM3_ROUTER_LPT 4
#endif
constexpr int LPT = SM3_ROUTER_LPT;  // lanes per token (A/B builds: --variant lpt8 / lpt2; round 4, same box: 8 lanes change
                                     // nothing -- fwd 0.205 -> 0.21, bwd 0.23 -> 0.20, the partials reduce +0.02 ms per step)
constexpr int RT_TOKENS = RT_THREADS / LPT;  // tokens per workgroup

// cooperative, coalesced copy of the [16 tokens][P] feature tile into LDS (row stride P+1).  Eight 16-byte loads per lane
// are issued before the first LDS store (a load -> store loop body made every iteration a full L2/HBM round trip: the
// 12 + 6 dependent trips of the two prologue loops were most of the kernel's 26 us).
__device__ __forceinline__ void load_h_tile(const float* __restrict__ hcat, int ldh, int P, int t0, int T, float* hs) {
  const int nq = P >> 2, ldt = P + 1;
  const int total = RT_TOKENS * nq;
  for (int base = 0; base < total; base += 8 * RT_THREADS) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = min(base + u * RT_THREADS + (int)threadIdx.x, total - 1);
      const int r = i / nq, q = i - r * nq;
      v[u] = ld4(hcat + (long)min(t0 + r, T - 1) * ldh + 4 * q);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = base + u * RT_THREADS + (int)threadIdx.x;
      if (i < total) {
        const int r = i / nq, q = i - r * nq;
        const f32x4 w = (t0 + r < T) ? v[u] : f32x4{0.f, 0.f, 0.f, 0.f};
        float* d = hs + r * ldt + 4 * q;
        d[0] = w[0]; d[1] = w[1]; d[2] = w[2]; d[3] = w[3];
      }
    }
  }
}

// column-normalised similarity matrix (P, E) -> LDS rows of ET floats (zero padded); same batching
template <int ET>
__device__ __forceinline__ void load_snorm(const float* __restrict__ snorm, int P, int E, float* s_s) {
  if (ET == E) {  // straight copy, P * E is a multiple of 4 (P is)
    const int total = (P * ET) >> 2;
    for (int base = 0; base < total; base += 8 * RT_THREADS) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = ld4(snorm + 4 * (long)min(base + u * RT_THREADS + (int)threadIdx.x, total - 1));
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = base + u * RT_THREADS + (int)threadIdx.x;
        if (i < total) st4(s_s + 4 * i, v[u]);
      }
    }
  } else {
    const int total = P * ET;
    for (int base = 0; base < total; base += 8 * RT_THREADS) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = min(base + u * RT_THREADS + (int)threadIdx.x, total - 1);
        const int p = i / ET, e = i - p * ET;
        v[u] = snorm[p * E + min(e, E - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int i = base + u * RT_THREADS + (int)threadIdx.x;
        if (i < total) s_s[i] = (i % ET) < E ? v[u] : 0.f;
      }
    }
  }
}

// hcat row layout: [h (P) | raw (E) | pad]; snorm = column-normalised sim_matrix (P,E); scale = exp(min(tau, ln 100)).
// snorm == NULL selects the LINEAR gate (gating='linear', convnext_moe.py:195-196: clean_logits = x @ w_gate): the first E
// of the P columns of hcat already ARE the clean logits.
// Outputs per token: top_idx/top_val (m = min(k+1,E), descending), gates (k, softmax of the top k), clean (E),
// sigma (E, train only), hnorm; per-workgroup partial sums [importance (E) | load (E)] to `partials`.
template <int ET>
__global__ __launch_bounds__(RT_THREADS) void moe_router_fwd_kernel(
    const float* __restrict__ hcat, int ldh, int P, const float* __restrict__ snorm, const float* __restrict__ scale_p,
    const float* __restrict__ noise, int T, int E, int k, int train, int32_t* __restrict__ top_idx,
    float* __restrict__ top_val, float* __restrict__ gates, float* __restrict__ clean_o, float* __restrict__ sigma_o,
    float* __restrict__ hnorm_o, float* __restrict__ partials, const int32_t* __restrict__ forced) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_s = sm;                         // snorm, row stride ET (zero padded), 16-byte aligned rows
  float* hs = sm + (long)P * ET;           // [16][P+1]
  const bool linear = snorm == nullptr;
  if (!linear) load_snorm<ET>(snorm, P, E, s_s);
  const int t0 = blockIdx.x * RT_TOKENS;
  load_h_tile(hcat, ldh, P, t0, T, hs);
  __syncthreads();
  const int tl = threadIdx.x / LPT, sub = threadIdx.x % LPT;  // token of the tile, lane of the token
  const int t = t0 + tl;
  const bool tv = t < T;
  const int m = min(k + 1, E);
  const float scale = linear ? 1.f : *scale_p;
  const bool smooth = train && (k < E);
  float impv[ET], ldv[ET];
#pragma unroll
  for (int e = 0; e < ET; e++) impv[e] = ldv[e] = 0.f;
  if (tv) {
    const float* hrow = hs + tl * (P + 1);
    float dot[ET];
#pragma unroll
    for (int e = 0; e < ET; e++) dot[e] = 0.f;
    float nn = 0.f;
    // per-token scattered operands requested before the dot loop: their round trip overlaps it
    const float* h = hcat + (long)t * ldh;
    float raw_n[ET], nz[ET];
#pragma unroll
    for (int e = 0; e < ET; e++) {
      raw_n[e] = (e < E && train) ? h[P + e] : 0.f;
      nz[e] = (e < E && train) ? noise[(long)t * E + e] : 0.f;
    }
    if (!linear) {
#pragma unroll 4
      for (int p = sub; p < P; p += LPT) {
        const float hv = hrow[p];
        nn += hv * hv;
        const float* srow = s_s + p * ET;
#pragma unroll
        for (int e = 0; e < ET; e++) dot[e] += hv * srow[e];
      }
      nn = group_sum<LPT>(nn);  // the 4 lanes of a token are adjacent: xor 1, 2 (all lanes of the group are in `tv`)
#pragma unroll
      for (int e = 0; e < ET; e++) dot[e] = group_sum<LPT>(dot[e]);
    } else {
#pragma unroll
      for (int e = 0; e < ET; e++) dot[e] = e < E ? hrow[e] : 0.f;
    }
    const float hn = linear ? 1.f : sqrtf(nn);
    const float inv = 1.0f / fmaxf(hn, 1e-12f);  // F.normalize eps
    float logit[ET], cl[ET], sg[ET];
#pragma unroll
    for (int e = 0; e < ET; e++) {
      cl[e] = (e < E) ? dot[e] * inv * scale : -INFINITY;
      sg[e] = 1.f;
      logit[e] = cl[e];
      if (e < E && train) {
        sg[e] = softplus_f(raw_n[e]) + 1e-2f;
        logit[e] = cl[e] + nz[e] * sg[e];
      }
    }
    // top-m selection (descending); ties -> lower index, like a stable descending sort.
    // `forced` (NULL in production; precision tests only): the k experts of every token are GIVEN (teacher-forced
    // routing: e.g. the reference's own choice, so that a mixed-precision run can be compared with an fp32 reference
    // without the butterfly effect of a near-tie routed the other way).  The forced experts win the selection -- ordered
    // among themselves by this run's logits -- and the runner-up is the best of the others; every value below (gates,
    // thresholds, load) is computed from this run's logits as usual.
    unsigned fset = 0;
    if (forced) {
      for (int j = 0; j < k; j++) fset |= 1u << forced[(long)t * k + j];
    }
    float tvv[ET];
    int tii[ET];
    unsigned used = 0;
#pragma unroll
    for (int j = 0; j < ET; j++) {
      tvv[j] = 0.f;
      tii[j] = -1;
      if (j < m) {
        float best = -INFINITY;
        int bi = -1;
        bool bf = false;
#pragma unroll
        for (int e = 0; e < ET; e++) {
          const bool fe = (fset >> e) & 1u;
          if (e < E && !((used >> e) & 1u) && (bi < 0 || (fe && !bf) || (fe == bf && logit[e] > best))) {
            best = logit[e];
            bi = e;
            bf = fe;
          }
        }
        used |= 1u << bi;
        tvv[j] = best;
        tii[j] = bi;
      }
    }
    float gsum = 0.f, gk[ET];
#pragma unroll
    for (int j = 0; j < ET; j++) {
      gk[j] = 0.f;
      if (j < k) {
        gk[j] = __expf(tvv[j] - tvv[0]);
        gsum += gk[j];
      }
    }
    float vin = 0.f, vout = 0.f;
#pragma unroll
    for (int j = 0; j < ET; j++) {
      if (j < k) gk[j] /= gsum;
      if (j == k) vin = tvv[j];
      if (j == k - 1) vout = tvv[j];
    }
#pragma unroll
    for (int e = 0; e < ET; e++) {
      float gg = 0.f;
#pragma unroll
      for (int j = 0; j < ET; j++)
        if (j < k && tii[j] == e) gg = gk[j];
      impv[e] = gg;
      if (smooth) {
        const float thr = (logit[e] > vin) ? vin : vout;  // _prob_in_top_k :159-173
        ldv[e] = (e < E) ? normal_cdf((cl[e] - thr) / sg[e]) : 0.f;
      } else {
        ldv[e] = gg > 0.f ? 1.f : 0.f;  // _gates_to_load :149-150
      }
    }
    if (sub == 0) {
#pragma unroll
      for (int j = 0; j < ET; j++) {
        if (j < m) {
          top_idx[(long)t * m + j] = tii[j];
          top_val[(long)t * m + j] = tvv[j];
        }
        if (j < k) gates[(long)t * k + j] = gk[j];
      }
#pragma unroll
      for (int e = 0; e < ET; e++)
        if (e < E) {
          clean_o[(long)t * E + e] = cl[e];
          if (train) sigma_o[(long)t * E + e] = sg[e];
        }
      hnorm_o[t] = hn;
    } else {  // the token is counted once in the importance / load partials
#pragma unroll
      for (int e = 0; e < ET; e++) impv[e] = ldv[e] = 0.f;
    }
  }
  // per-workgroup partials: deterministic wave shuffle tree
#pragma unroll
  for (int e = 0; e < ET; e++) {
    const float a = group_sum<64>(impv[e]);
    const float c = group_sum<64>(ldv[e]);
    if (threadIdx.x == 0 && e < E) {
      partials[(long)blockIdx.x * 2 * E + e] = a;
      partials[(long)blockIdx.x * 2 * E + E + e] = c;
    }
  }
}

// backward of the router: per token dlogits from (i) the combine (dgate), (ii) the importance term, (iii) the load
// term (train), then through noise / softplus / cosine normalisation.  Writes
//   dhcat[t] = [dh (P) | draw (E) | 0...]   (row-major, ld = ldh; the gate GEMMs turn it into dWp, dWn, dx)
//   dcn[t,e] = dclean[t,e] / max(|h_t|, eps)      (so that dSnorm = scale * h^T . dcn is one TN GEMM)
//   ds_part[workgroup] = sum_t sum_e dclean[t,e] * clean[t,e] / scale   (d scale)
template <int ET>
__global__ __launch_bounds__(RT_THREADS) void moe_router_bwd_kernel(
    const float* __restrict__ hcat, int ldh, int P, const float* __restrict__ snorm, const float* __restrict__ scale_p,
    const float* __restrict__ noise, int T, int E, int k, int train, const int32_t* __restrict__ top_idx,
    const float* __restrict__ top_val, const float* __restrict__ gates, const float* __restrict__ clean_i,
    const float* __restrict__ sigma_i, const float* __restrict__ hnorm_i, const float* __restrict__ dgate,
    const float* __restrict__ dimp, const float* __restrict__ dload, float* __restrict__ dhcat,
    float* __restrict__ dcn, double* __restrict__ ds_part, double* __restrict__ ds_sq) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* s_s = sm;
  float* hs = sm + (long)P * ET;  // [16][P+1]: h on the way in, dh on the way out
  const bool linear = snorm == nullptr;
  if (!linear) load_snorm<ET>(snorm, P, E, s_s);
  const int t0 = blockIdx.x * RT_TOKENS;
  load_h_tile(hcat, ldh, P, t0, T, hs);
  __syncthreads();
  const int tl = threadIdx.x / LPT, sub = threadIdx.x % LPT;
  const int t = t0 + tl;
  const bool tv = t < T;
  const int m = min(k + 1, E);
  const float scale = linear ? 1.f : *scale_p;
  const bool smooth = train && (k < E);
  double ds_local = 0.0;  // d(scale) is one number summed over every token and expert with mixed signs: the terms are
  float draw[ET];           // added in double from the first product on (gate_prep_bwd finishes the sum in double too)
#pragma unroll
  for (int e = 0; e < ET; e++) draw[e] = 0.f;
  if (tv) {
    const long tt = t;
    const float* h = hcat + tt * ldh;
    float dlogit[ET], dclean[ET], dsig[ET];
#pragma unroll
    for (int e = 0; e < ET; e++) dlogit[e] = dclean[e] = dsig[e] = 0.f;
    int tii[ET];
    float tvv[ET];
#pragma unroll
    for (int j = 0; j < ET; j++) {
      tii[j] = -1;
      tvv[j] = 0.f;
      if (j < m) {
        tii[j] = top_idx[tt * m + j];
        tvv[j] = top_val[tt * m + j];
      }
    }
    float gk[ET], dg[ET], dotg = 0.f;  // softmax backward
#pragma unroll
    for (int j = 0; j < ET; j++) {
      gk[j] = dg[j] = 0.f;
      if (j < k) {
        gk[j] = gates[tt * k + j];
        dg[j] = dgate[tt * k + j] + dimp[tii[j]];
        dotg += gk[j] * dg[j];
      }
    }
#pragma unroll
    for (int j = 0; j < ET; j++)
      if (j < k) {
        const float dv = gk[j] * (dg[j] - dotg);
#pragma unroll
        for (int e = 0; e < ET; e++)
          if (e == tii[j]) dlogit[e] += dv;
      }
    float cl[ET];
#pragma unroll
    for (int e = 0; e < ET; e++) cl[e] = (e < E) ? clean_i[tt * E + e] : 0.f;
    if (smooth) {
      float vin = 0.f, vout = 0.f;
      int iin = -1, iout = -1;
#pragma unroll
      for (int j = 0; j < ET; j++) {
        if (j == k) { vin = tvv[j]; iin = tii[j]; }
        if (j == k - 1) { vout = tvv[j]; iout = tii[j]; }
      }
      float dthr_in = 0.f, dthr_out = 0.f;
#pragma unroll
      for (int e = 0; e < ET; e++)
        if (e < E) {
          const float sg = sigma_i[tt * E + e];
          const float lgt = cl[e] + noise[tt * E + e] * sg;
          const bool is_in = lgt > vin;
          const float thr = is_in ? vin : vout;
          const float z = (cl[e] - thr) / sg;
          const float q = dload[e] * normal_pdf(z) / sg;
          dclean[e] += q;
          dsig[e] -= q * z;
          if (is_in) dthr_in -= q; else dthr_out -= q;
        }
#pragma unroll
      for (int e = 0; e < ET; e++) {
        if (e == iin) dlogit[e] += dthr_in;
        if (e == iout) dlogit[e] += dthr_out;
      }
    }
#pragma unroll
    for (int e = 0; e < ET; e++)
      if (e < E) {
        dclean[e] += dlogit[e];
        if (train) {
          dsig[e] += noise[tt * E + e] * dlogit[e];
          draw[e] = dsig[e] / (1.0f + __expf(-h[P + e]));  // d softplus = sigmoid
        }
      }
    const float hn = linear ? 1.f : hnorm_i[tt];
    const float inv = 1.0f / fmaxf(hn, 1e-12f);
    double dsl = 0.0;
#pragma unroll
    for (int e = 0; e < ET; e++)
      if (e < E) {
        if (sub == 0) dcn[tt * E + e] = dclean[e] * inv;
        dsl += (double)dclean[e] * (double)cl[e];
      }
    ds_local = sub == 0 ? dsl / (double)scale : 0.0;  // the token is counted once
    // dh = (dhh - hh <hh, dhh>) * inv,  dhh = scale * snorm . dclean,  hh = h * inv  (row lives in LDS); the four
    // lanes of the token split p
    float* hrow = hs + tl * (P + 1);
    float proj = 0.f;
    if (linear) {  // d clean IS d h on the first E columns
      for (int p = sub; p < P; p += LPT) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < ET; e++)
          if (e == p) a = dclean[e];
        hrow[p] = a;
      }
    } else {
#pragma unroll 4
    for (int p = sub; p < P; p += LPT) {
      const float* srow = s_s + p * ET;
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < ET; e++) a += srow[e] * dclean[e];
      proj += a * scale * hrow[p] * inv;
    }
    proj = group_sum<LPT>(proj);
    if (hn < 1e-12f) proj = 0.f;  // clamp region of F.normalize: d/dh (h/eps) = dhh/eps
#pragma unroll 4
    for (int p = sub; p < P; p += LPT) {
      const float* srow = s_s + p * ET;
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < ET; e++) a += srow[e] * dclean[e];
      hrow[p] = (a * scale - hrow[p] * inv * proj) * inv;
    }
    }
  }
  __syncthreads();
  // coalesced write-out of the dh tile, then the [draw | 0] tail of each row
  {
    const int nq = P >> 2, ldt = P + 1;
    for (int i = threadIdx.x; i < RT_TOKENS * nq; i += RT_THREADS) {
      const int r = i / nq, q = i - r * nq;
      if (t0 + r < T) {
        const float* d = hs + r * ldt + 4 * q;
        st4(dhcat + (long)(t0 + r) * ldh + 4 * q, f32x4{d[0], d[1], d[2], d[3]});
      }
    }
    if (tv && sub == 0) {
      float* dh = dhcat + (long)t * ldh;
#pragma unroll
      for (int e = 0; e < ET; e++)
        if (e < E) dh[P + e] = draw[e];
      for (int c = P + E; c < ldh; c++) dh[c] = 0.f;
    }
  }
  double a = ds_local, q = ds_local * ds_local;  // q: the PER-TOKEN terms' squares (conditioning of the sum, tests)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (threadIdx.x == 0) {
    ds_part[blockIdx.x] = a;
    if (ds_sq) ds_sq[blockIdx.x] = q;
  }
}

}  // namespace

extern "C" {

int sm3_moe_router_partial_rows(int T) { return (T + RT_TOKENS - 1) / RT_TOKENS; }

int sm3_moe_router_fwd(const float* hcat, int ldh, int P, const float* snorm, const float* scale, const float* noise,
                       int T, int E, int k, int train, int32_t* top_idx, float* top_val, float* gates, float* clean,
                       float* sigma, float* hnorm, float* partials, const int32_t* forced_topk, sm3_stream_t stream) {
  if (!hcat || (snorm && !scale) || !top_idx || !top_val || !gates || !clean || !hnorm || !partials)
    return SM3_ERR_INVALID_ARG;
  if (!snorm && P < E) return SM3_ERR_INVALID_ARG;  // linear gate: the logits are the first E of the P columns
  if (T <= 0 || E < 1 || E > 32 || k < 1 || k > E || (P & 3) || P + E > ldh || (ldh & 3)) return SM3_ERR_INVALID_ARG;
  if (train && (!noise || !sigma)) return SM3_ERR_INVALID_ARG;
  const int nblk = sm3_moe_router_partial_rows(T);
  hipStream_t st = (hipStream_t)stream;
#define CALL(ET)                                                                                                   \
  moe_router_fwd_kernel<ET><<<nblk, RT_THREADS, ((size_t)P * ET + (size_t)RT_TOKENS * (P + 1)) * sizeof(float), st>>>( \
      hcat, ldh, P, snorm, scale, noise, T, E, k, train, top_idx, top_val, gates, clean, sigma, hnorm, partials,     \
      forced_topk)
  if (E <= 4) CALL(4);
  else if (E <= 8) CALL(8);
  else if (E <= 16) CALL(16);
  else CALL(32);
#undef CALL
  return launch_status();
}

int sm3_moe_router_bwd(const float* hcat, int ldh, int P, const float* snorm, const float* scale, const float* noise,
                       int T, int E, int k, int train, const int32_t* top_idx, const float* top_val,
                       const float* gates, const float* clean, const float* sigma, const float* hnorm,
                       const float* dgate, const float* dimp, const float* dload, float* dhcat, float* dcn,
                       double* ds_part, double* ds_sq, sm3_stream_t stream) {
  if (!hcat || (snorm && !scale) || !top_idx || !top_val || !gates || !clean || !hnorm || !dgate || !dimp || !dload ||
      !dhcat || !dcn || !ds_part)
    return SM3_ERR_INVALID_ARG;
  if (!snorm && P < E) return SM3_ERR_INVALID_ARG;
  if (T <= 0 || E < 1 || E > 32 || k < 1 || k > E || (P & 3) || P + E > ldh || (ldh & 3)) return SM3_ERR_INVALID_ARG;
  if (train && (!noise || !sigma)) return SM3_ERR_INVALID_ARG;
  const int nblk = sm3_moe_router_partial_rows(T);
  hipStream_t st = (hipStream_t)stream;
#define CALL(ET)                                                                                                   \
  moe_router_bwd_kernel<ET><<<nblk, RT_THREADS, ((size_t)P * ET + (size_t)RT_TOKENS * (P + 1)) * sizeof(float), st>>>( \
      hcat, ldh, P, snorm, scale, noise, T, E, k, train, top_idx, top_val, gates, clean, sigma, hnorm, dgate, dimp,    \
      dload, dhcat, dcn, ds_part, ds_sq)
  if (E <= 4) CALL(4);
  else if (E <= 8) CALL(8);
  else if (E <= 16) CALL(16);
  else CALL(32);
#undef CALL
  return launch_status();
}

}  // extern "C"
