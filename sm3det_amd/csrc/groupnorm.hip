// groupnorm.hip -- GroupNorm (+ ReLU) on NHWC tokens for gfx950: the normalisation of the GFL head's conv towers
// (`ConvModule(conv 3x3, norm_cfg=dict(type='GN', num_groups=32), ReLU)` x 4 per tower; reference config
// local_configs/main_SM3Det.py:29-48 `type='GFLHead', stacked_convs=4, feat_channels=256`; the class itself lives in
// mmdet 2.x, which the reference does not vendor -- semantics restated from torch.nn.GroupNorm).
//
//   y[b,p,c] = max(0, (x[b,p,c] - mean[b,g]) * rstd[b,g] * gamma[c] + beta[c]),  g = c / (C/G), stats over (p, c in g)
//
// x is (B, P, C) with P = H*W pixels: lanes run over channel quads (16 B, coalesced), a workgroup strides over pixels.
// Two passes per direction (the statistics need every pixel of an image): `stats` leaves per-workgroup partial sums,
// `apply` folds them in its prologue (<= 64 partial rows per image) and streams the image once more.  HBM-bound:
// forward 3 passes of B*P*C*4 (read, read, write), backward 5 (dy, x twice each + dx).  Deterministic: no atomics.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_BLOCKS = 64;  // workgroups per image
constexpr int GN_MAX_C = 1024;

inline int gn_blocks(long P, int C) {
  const int ppi = GN_THREADS / (C / 4);            // pixels per iteration of a workgroup
  long nb = (P + (long)ppi * 8 - 1) / ((long)ppi * 8);  // >= 8 iterations each
  if (nb > GN_MAX_BLOCKS) nb = GN_MAX_BLOCKS;
  return (int)(nb < 1 ? 1 : nb);
}

// Folds per-thread (NA accumulators per channel quad) over the pixel lanes of the workgroup and, optionally, over the
// quads of each group.  red: LDS [pixel lanes][nq][NA] floats.
template <int NA>
__device__ __forceinline__ void fold_pixel_lanes(float (&acc)[NA][4], float* red, int nq, int q, int pl, int npl,
                                                 bool active) {
  if (active) {
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[((pl * nq + q) * NA + a) * 4 + e] = acc[a][e];
  }
  __syncthreads();
  if (active && pl == 0) {
    for (int l = 1; l < npl; l++)
#pragma unroll
      for (int a = 0; a < NA; a++)
#pragma unroll
        for (int e = 0; e < 4; e++) acc[a][e] += red[((l * nq + q) * NA + a) * 4 + e];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[(q * NA + a) * 4 + e] = acc[a][e];  // row 0 now holds the workgroup totals
  }
  __syncthreads();
}

// gpart[(b*nblk + blk)*G + g] = (sum x, sum x^2) over this workgroup's pixels and the channels of group g
__global__ __launch_bounds__(GN_THREADS) void gn_stats_kernel(const float* __restrict__ x, long P, int C, int G,
                                                             float* __restrict__ gpart) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int nq = C >> 2, npl = GN_THREADS / nq;
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  const bool active = pl < npl;
  const int b = blockIdx.y, nblk = gridDim.x;
  const float* xb = x + (long)b * P * C;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (active)
    for (long p = (long)blockIdx.x * npl + pl; p < P; p += (long)nblk * npl) {
      const f32x4 v = ld4(xb + p * C + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acc[0][e] += v[e];
        acc[1][e] += v[e] * v[e];
      }
    }
  fold_pixel_lanes<2>(acc, red, nq, q, pl, npl, active);
  const int qpg = (C / G) >> 2;  // quads per group
  if (threadIdx.x < G) {
    float s = 0.f, ss = 0.f;
    for (int i = 0; i < qpg; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        s += red[((threadIdx.x * qpg + i) * 2 + 0) * 4 + e];
        ss += red[((threadIdx.x * qpg + i) * 2 + 1) * 4 + e];
      }
    float* o = gpart + (((long)b * nblk + blockIdx.x) * G + threadIdx.x) * 2;
    o[0] = s;
    o[1] = ss;
  }
}

// y = relu?(gn(x)); stats[b][g] = (mean, rstd) written by workgroup 0 of each image
__global__ __launch_bounds__(GN_THREADS) void gn_apply_fwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ gpart, int npart, float eps,
                                                                 int relu, float* __restrict__ y,
                                                                 float* __restrict__ stats, long P, int C, int G) {
  __shared__ float s_mean[GN_MAX_C / 4], s_rstd[GN_MAX_C / 4];
  const int b = blockIdx.y;
  if (threadIdx.x < G) {
    double s = 0.0, ss = 0.0;
    for (int i = 0; i < npart; i++) {
      const float* o = gpart + (((long)b * npart + i) * G + threadIdx.x) * 2;
      s += (double)o[0];
      ss += (double)o[1];
    }
    const double n = (double)P * (C / G);
    const double mean = s / n;
    double var = ss / n - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    s_mean[threadIdx.x] = (float)mean;
    s_rstd[threadIdx.x] = rstd;
    if (blockIdx.x == 0 && stats) {
      stats[((long)b * G + threadIdx.x) * 2] = (float)mean;
      stats[((long)b * G + threadIdx.x) * 2 + 1] = rstd;
    }
  }
  __syncthreads();
  const int nq = C >> 2, npl = GN_THREADS / nq;
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  if (pl >= npl) return;
  const int g = (4 * q) / (C / G);
  const float mean = s_mean[g], rstd = s_rstd[g];
  const f32x4 gw = ld4(gamma + 4 * q) * rstd, bw = ld4(beta + 4 * q);
  const float* xb = x + (long)b * P * C;
  float* yb = y + (long)b * P * C;
  for (long p = (long)blockIdx.x * npl + pl; p < P; p += (long)gridDim.x * npl) {
    f32x4 v = (ld4(xb + p * C + 4 * q) - mean) * gw + bw;
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = fmaxf(v[e], 0.f);
    }
    st4(yb + p * C + 4 * q, v);
  }
}

// backward pass 1: dy' = dy * (y > 0); per workgroup: gpart[..][g] = (sum dy' gamma, sum dy' gamma xhat),
// cpart[(b*nblk + blk)][0][c] = sum dy' xhat (d gamma), [1][c] = sum dy' (d beta)
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_stats_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ y,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ stats, int relu, long P,
                                                                 int C, int G, float* __restrict__ gpart,
                                                                 float* __restrict__ cpart) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  const int nq = C >> 2, npl = GN_THREADS / nq;
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  const bool active = pl < npl;
  const int b = blockIdx.y, nblk = gridDim.x;
  const int g = (4 * q) / (C / G);
  const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
  const long base = (long)b * P * C;
  float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // [0] = dy' xhat, [1] = dy'
  if (active)
    for (long p = (long)blockIdx.x * npl + pl; p < P; p += (long)nblk * npl) {
      const long o = base + p * C + 4 * q;
      f32x4 d = ld4(dy + o);
      if (relu) {
        const f32x4 yy = ld4(y + o);
#pragma unroll
        for (int e = 0; e < 4; e++) d[e] = yy[e] > 0.f ? d[e] : 0.f;
      }
      const f32x4 xh = (ld4(x + o) - mean) * rstd;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        acc[0][e] += d[e] * xh[e];
        acc[1][e] += d[e];
      }
    }
  fold_pixel_lanes<2>(acc, red, nq, q, pl, npl, active);
  // red row 0: [q][a][e] = workgroup totals per channel
  float* cp = cpart + ((long)b * nblk + blockIdx.x) * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += GN_THREADS) {
    const int a = i / C, c = i - a * C;
    cp[i] = red[((c >> 2) * 2 + a) * 4 + (c & 3)];
  }
  const int cpg = C / G;
  if (threadIdx.x < G) {
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < cpg; i++) {
      const int c = threadIdx.x * cpg + i;
      const float gm = gamma[c];
      s2 += gm * red[((c >> 2) * 2 + 0) * 4 + (c & 3)];
      s1 += gm * red[((c >> 2) * 2 + 1) * 4 + (c & 3)];
    }
    float* o = gpart + (((long)b * nblk + blockIdx.x) * G + threadIdx.x) * 2;
    o[0] = s1;
    o[1] = s2;
  }
}

// backward pass 2: dx = rstd * (gamma dy' - (S1 + xhat S2) / n)
__global__ __launch_bounds__(GN_THREADS) void gn_bwd_apply_kernel(const float* __restrict__ dy,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ y,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ stats,
                                                                 const float* __restrict__ gpart, int npart, int relu,
                                                                 float* __restrict__ dx, long P, int C, int G) {
  __shared__ float s_1[GN_MAX_C / 4], s_2[GN_MAX_C / 4];
  const int b = blockIdx.y;
  if (threadIdx.x < G) {
    double a = 0.0, c = 0.0;
    for (int i = 0; i < npart; i++) {
      const float* o = gpart + (((long)b * npart + i) * G + threadIdx.x) * 2;
      a += (double)o[0];
      c += (double)o[1];
    }
    const double n = (double)P * (C / G);
    s_1[threadIdx.x] = (float)(a / n);
    s_2[threadIdx.x] = (float)(c / n);
  }
  __syncthreads();
  const int nq = C >> 2, npl = GN_THREADS / nq;
  const int q = threadIdx.x % nq, pl = threadIdx.x / nq;
  if (pl >= npl) return;
  const int g = (4 * q) / (C / G);
  const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
  const float m1 = s_1[g], m2 = s_2[g];
  const f32x4 gw = ld4(gamma + 4 * q);
  const long base = (long)b * P * C;
  for (long p = (long)blockIdx.x * npl + pl; p < P; p += (long)gridDim.x * npl) {
    const long o = base + p * C + 4 * q;
    f32x4 d = ld4(dy + o);
    if (relu) {
      const f32x4 yy = ld4(y + o);
#pragma unroll
      for (int e = 0; e < 4; e++) d[e] = yy[e] > 0.f ? d[e] : 0.f;
    }
    const f32x4 xh = (ld4(x + o) - mean) * rstd;
    st4(dx + o, (d * gw - m1 - xh * m2) * rstd);
  }
}

bool gn_dims_ok(int B, long P, int C, int G) {
  return B > 0 && P > 0 && C > 0 && G > 0 && C <= GN_MAX_C && (C % G) == 0 && ((C / G) % 4) == 0 && (C / 4) <= GN_THREADS &&
         G <= GN_MAX_C / 4;
}

}  // namespace

extern "C" {

int sm3_groupnorm_blocks(long P, int C) { return (P > 0 && C >= 4) ? gn_blocks(P, C) : 0; }

size_t sm3_groupnorm_workspace_bytes(int B, long P, int C, int G) {
  if (!gn_dims_ok(B, P, C, G)) return 0;
  const size_t nblk = gn_blocks(P, C);
  return (size_t)B * nblk * ((size_t)G * 2 + (size_t)2 * C) * sizeof(float);
}

int sm3_groupnorm_fwd(const float* x, const float* gamma, const float* beta, float eps, int relu, float* y,
                      float* stats, int B, long P, int C, int G, void* workspace, size_t workspace_bytes,
                      sm3_stream_t stream) {
  if (!x || !gamma || !beta || !y || !stats || !gn_dims_ok(B, P, C, G)) return SM3_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < sm3_groupnorm_workspace_bytes(B, P, C, G)) return SM3_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = gn_blocks(P, C);
  float* gpart = (float*)workspace;
  const size_t lds = (size_t)GN_THREADS * 2 * 4 * sizeof(float);
  gn_stats_kernel<<<dim3(nblk, B), GN_THREADS, lds, st>>>(x, P, C, G, gpart);
  long nb2 = (P + 63) / 64;
  if (nb2 > 256) nb2 = 256;
  gn_apply_fwd_kernel<<<dim3((int)nb2, B), GN_THREADS, 0, st>>>(x, gamma, beta, gpart, nblk, eps, relu, y, stats, P, C, G);
  return launch_status();
}

/* dgamma_dbeta_part: (B * sm3_groupnorm_blocks(P, C)) rows x 2C partial sums [d gamma | d beta]; the caller reduces the
 * rows (sm3_row_partials_reduce) -- on another stream if it likes. */
int sm3_groupnorm_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* stats,
                      int relu, float* dx, float* dgamma_dbeta_part, int B, long P, int C, int G, void* workspace,
                      size_t workspace_bytes, sm3_stream_t stream) {
  if (!dy || !x || !gamma || !stats || !dx || !dgamma_dbeta_part || (relu && !y) || !gn_dims_ok(B, P, C, G))
    return SM3_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < sm3_groupnorm_workspace_bytes(B, P, C, G)) return SM3_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = gn_blocks(P, C);
  float* gpart = (float*)workspace;
  const size_t lds = (size_t)GN_THREADS * 2 * 4 * sizeof(float);
  gn_bwd_stats_kernel<<<dim3(nblk, B), GN_THREADS, lds, st>>>(dy, x, y, gamma, stats, relu, P, C, G, gpart,
                                                            dgamma_dbeta_part);
  long nb2 = (P + 63) / 64;
  if (nb2 > 256) nb2 = 256;
  gn_bwd_apply_kernel<<<dim3((int)nb2, B), GN_THREADS, 0, st>>>(dy, x, y, gamma, stats, gpart, nblk, relu, dx, P, C, G);
  return launch_status();
}

}  // extern "C"
