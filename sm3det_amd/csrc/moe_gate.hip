// moe_gate.hip -- the small-tensor side of the cosine top-k gate, one launch each instead of ~60 tiny elementwise
// launches per MoE block and step:
//   * gate_prep fwd/bwd: build the fused gate matrix [Wp; Wn^T; 0] / bias [bp; 0], F.normalize(sim_matrix, dim=0) and
//     exp(clamp(temperature, max)) -- CosineTopKGate.forward, mmrotate/models/backbones/convnext_moe.py:96-105, and
//     the `x @ w_noise` operand of noisy_top_k_gating (:199-201) -- and route their gradients back to the reference
//     parameters (cosine_projector.{weight,bias}, w_noise, sim_matrix, temperature).
//   * aux_loss fwd/bwd: importance = gates.sum(0), load = prob_in_top_k.sum(0) (both arrive as per-workgroup partial
//     sums from the router kernel), loss = (cv_squared(importance) + cv_squared(load)) * loss_coef (:140-147,
//     :234-238) and its gradient w.r.t. the two E-vectors.
// All of it is latency-bound small work: single-workgroup reductions, no atomics, deterministic.
#include "common.h"

namespace {

constexpr int GP_THREADS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// grid = PC + 1.  Workgroup r < PC writes row r of wcat and bcat[r]; workgroup PC normalises the sim matrix.
__global__ __launch_bounds__(GP_THREADS) void gate_prep_fwd_kernel(const float* __restrict__ wp,
                                                                  const float* __restrict__ bp,
                                                                  const float* __restrict__ wn,
                                                                  const float* __restrict__ sim,
                                                                  const float* __restrict__ temperature,
                                                                  float clamp_max, int P, int C, int E, int PC,
                                                                  float* __restrict__ wcat, float* __restrict__ bcat,
                                                                  float* __restrict__ snorm,
                                                                  float* __restrict__ scale) {
  const int r = blockIdx.x;
  if (r < PC) {
    float* dst = wcat + (long)r * C;
    if (r < P) {
      const float* src = wp + (long)r * C;
      for (int c = threadIdx.x; c < C; c += GP_THREADS) dst[c] = src[c];
    } else if (r < P + E) {
      const int e = r - P;
      for (int c = threadIdx.x; c < C; c += GP_THREADS) dst[c] = wn[(long)c * E + e];
    } else {
      for (int c = threadIdx.x; c < C; c += GP_THREADS) dst[c] = 0.f;
    }
    if (threadIdx.x == 0) bcat[r] = r < P ? bp[r] : 0.f;
    return;
  }
  // F.normalize(sim, dim=0): column e divided by max(||col||_2, 1e-12); one wave per column
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int e = wave; e < E; e += GP_THREADS / 64) {
    float s = 0.f;
    for (int p = lane; p < P; p += 64) {
      const float v = sim[(long)p * E + e];
      s += v * v;
    }
    s = wave_sum(s);
    const float inv = 1.f / fmaxf(sqrtf(s), 1e-12f);
    for (int p = lane; p < P; p += 64) snorm[(long)p * E + e] = sim[(long)p * E + e] * inv;
  }
  if (threadIdx.x == 0) scale[0] = expf(fminf(temperature[0], clamp_max));
}

// grid = P + E + 1.  r < P: dwp row / dbp; P <= r < P+E: column e of dwn; last: dsim, dtemperature.
// dsn is the gradient w.r.t. (snorm * scale) (i.e. the router's h^T @ dlogits); ds_part are the router's per-workgroup
// partial sums of d(scale).
__global__ __launch_bounds__(GP_THREADS) void gate_prep_bwd_kernel(
    const float* __restrict__ dwcat, const float* __restrict__ dbcat, const float* __restrict__ dsn,
    const double* __restrict__ ds_part, int n_part, const float* __restrict__ sim,
    const float* __restrict__ temperature, float clamp_max, int P, int C, int E, float* __restrict__ dwp,
    float* __restrict__ dbp, float* __restrict__ dwn, float* __restrict__ dsim, float* __restrict__ dtemp) {
  const int r = blockIdx.x;
  if (r < P) {
    const float* src = dwcat + (long)r * C;
    float* dst = dwp + (long)r * C;
    for (int c = threadIdx.x; c < C; c += GP_THREADS) dst[c] = src[c];
    if (threadIdx.x == 0) dbp[r] = dbcat[r];
    return;
  }
  if (r < P + E) {
    const int e = r - P;
    const float* src = dwcat + (long)r * C;
    for (int c = threadIdx.x; c < C; c += GP_THREADS) dwn[(long)c * E + e] = src[c];
    return;
  }
  const float t = temperature[0];
  const float scale = expf(fminf(t, clamp_max));
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int e = wave; e < E; e += GP_THREADS / 64) {
    float ss = 0.f, gv = 0.f;  // ||col||^2 and <g, col>
    for (int p = lane; p < P; p += 64) {
      const float v = sim[(long)p * E + e];
      const float g = dsn[(long)p * E + e] * scale;
      ss += v * v;
      gv += g * v;
    }
    ss = wave_sum(ss);
    gv = wave_sum(gv);
    const float n = sqrtf(ss);
    for (int p = lane; p < P; p += 64) {
      const float v = sim[(long)p * E + e];
      const float g = dsn[(long)p * E + e] * scale;
      // v/max(n,eps): above the clamp d = (g - v <g,v>/n^2)/n ; below it the norm is treated as the constant eps
      dsim[(long)p * E + e] = n >= 1e-12f ? (g - v * (gv / ss)) / n : g / 1e-12f;
    }
  }
  // d(scale): a fully cancelling sum over all tokens and experts -- double from the first product (router backward) to here
  __shared__ double red[GP_THREADS / 64];
  double s = 0.0;
  for (int i = threadIdx.x; i < n_part; i += GP_THREADS) s += ds_part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double dscale = (red[0] + red[1]) + (red[2] + red[3]);
    dtemp[0] = t <= clamp_max ? (float)(dscale * (double)scale) : 0.f;  // clamp(max=) passes the gradient where t <= max
  }
}

// cv_squared of an E-vector held by one thread (reference :140-147): unbiased variance / (mean^2 + 1e-10), 0 if E == 1.
// Evaluated in double: E values, and (x - mean) cancels badly in fp32 when the experts are well balanced.
__device__ float cv_squared(const float* x, int E) {
  if (E == 1) return 0.f;
  double m = 0.0;
  for (int i = 0; i < E; i++) m += (double)x[i];
  m /= (double)E;
  double v = 0.0;
  for (int i = 0; i < E; i++) v += ((double)x[i] - m) * ((double)x[i] - m);
  v /= (double)(E - 1);
  return (float)(v / (m * m + 1e-10));
}

__device__ void cv_squared_grad(const float* x, int E, float g, float* dx) {
  if (E == 1) {
    dx[0] = 0.f;
    return;
  }
  double m = 0.0;
  for (int i = 0; i < E; i++) m += (double)x[i];
  m /= (double)E;
  double v = 0.0;
  for (int i = 0; i < E; i++) v += ((double)x[i] - m) * ((double)x[i] - m);
  v /= (double)(E - 1);
  const double den = m * m + 1e-10;
  const double dm = -v * 2.0 * m / (den * den) / (double)E;
  for (int i = 0; i < E; i++) dx[i] = (float)((double)g * (2.0 * ((double)x[i] - m) / (double)(E - 1) / den + dm));
}

// partials (nblk, 2E) -> tot (2E) = [importance | load]; loss = coef * (cv2(importance) + cv2(load)).  One workgroup.
__global__ __launch_bounds__(GP_THREADS) void aux_loss_fwd_kernel(const float* __restrict__ partials, int nblk, int E,
                                                                 float coef, float* __restrict__ tot,
                                                                 float* __restrict__ loss) {
  extern __shared__ float sm[];  // [GP_THREADS / ncol][ncol]
  const int ncol = 2 * E;
  const int col = threadIdx.x % ncol, grp = threadIdx.x / ncol, ngrp = GP_THREADS / ncol;
  float s = 0.f;
  if (grp < ngrp) {
    float u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // 8 independent loads in flight (one workgroup walks
    int r = grp;                                             // up to 2048 partial rows: latency-bound otherwise)
    for (; r + 7 * ngrp < nblk; r += 8 * ngrp) {
#pragma unroll
      for (int q = 0; q < 8; q++) u[q] += partials[(long)(r + q * ngrp) * ncol + col];
    }
    for (; r < nblk; r += ngrp) u[0] += partials[(long)r * ncol + col];
    s = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    sm[grp * ncol + col] = s;
  }
  __syncthreads();
  if (threadIdx.x < ncol) {
    float t = 0.f;
    for (int g = 0; g < ngrp; g++) t += sm[g * ncol + threadIdx.x];
    tot[threadIdx.x] = t;
    sm[threadIdx.x] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = coef * (cv_squared(sm, E) + cv_squared(sm + E, E));
}

__global__ void aux_loss_bwd_kernel(const float* __restrict__ tot, const float* __restrict__ dloss, int E, float coef,
                                    float* __restrict__ dimp, float* __restrict__ dload) {
  if (threadIdx.x == 0) cv_squared_grad(tot, E, dloss[0] * coef, dimp);
  if (threadIdx.x == 1) cv_squared_grad(tot + E, E, dloss[0] * coef, dload);
}

}  // namespace

extern "C" {

int sm3_moe_gate_prep_fwd(const float* wp, const float* bp, const float* wn, const float* sim,
                          const float* temperature, float clamp_max, int P, int C, int E, int PC, float* wcat,
                          float* bcat, float* snorm, float* scale, sm3_stream_t stream) {
  if (!wp || !bp || !wn || !sim || !temperature || !wcat || !bcat || !snorm || !scale) return SM3_ERR_INVALID_ARG;
  if (P <= 0 || C <= 0 || E <= 0 || PC < P + E) return SM3_ERR_INVALID_ARG;
  gate_prep_fwd_kernel<<<PC + 1, GP_THREADS, 0, (hipStream_t)stream>>>(wp, bp, wn, sim, temperature, clamp_max, P, C,
                                                                      E, PC, wcat, bcat, snorm, scale);
  return launch_status();
}

int sm3_moe_gate_prep_bwd(const float* dwcat, const float* dbcat, const float* dsn, const double* ds_part, int n_part,
                          const float* sim, const float* temperature, float clamp_max, int P, int C, int E,
                          float* dwp, float* dbp, float* dwn, float* dsim, float* dtemp, sm3_stream_t stream) {
  if (!dwcat || !dbcat || !dsn || !ds_part || !sim || !temperature || !dwp || !dbp || !dwn || !dsim || !dtemp)
    return SM3_ERR_INVALID_ARG;
  if (P <= 0 || C <= 0 || E <= 0 || n_part < 0) return SM3_ERR_INVALID_ARG;
  gate_prep_bwd_kernel<<<P + E + 1, GP_THREADS, 0, (hipStream_t)stream>>>(
      dwcat, dbcat, dsn, ds_part, n_part, sim, temperature, clamp_max, P, C, E, dwp, dbp, dwn, dsim, dtemp);
  return launch_status();
}

int sm3_moe_aux_loss_fwd(const float* partials, int nblk, int E, float coef, float* tot, float* loss,
                         sm3_stream_t stream) {
  if (!partials || !tot || !loss || nblk <= 0 || E <= 0 || 2 * E > GP_THREADS) return SM3_ERR_INVALID_ARG;
  aux_loss_fwd_kernel<<<1, GP_THREADS, sizeof(float) * GP_THREADS, (hipStream_t)stream>>>(partials, nblk, E, coef, tot,
                                                                                          loss);
  return launch_status();
}

int sm3_moe_aux_loss_bwd(const float* tot, const float* dloss, int E, float coef, float* dimp, float* dload,
                         sm3_stream_t stream) {
  if (!tot || !dloss || !dimp || !dload || E <= 0) return SM3_ERR_INVALID_ARG;
  aux_loss_bwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(tot, dloss, E, coef, dimp, dload);
  return launch_status();
}

}  // extern "C"
