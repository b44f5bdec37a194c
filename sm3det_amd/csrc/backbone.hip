// backbone.hip -- the non-GEMM kernels of the SM3Det grid-level sparse-MoE ConvNeXt backbone on gfx950.
//
// Reference semantics (mmrotate/models/backbones/convnext_moe.py): LayerNorm2d :30-47, ConvNeXtBlock :343-372
// (depthwise 7x7 :347, LN :351, layer scale :368, residual :370), CosineTopKGate :99-106, noisy_top_k_gating
// :194-223, _prob_in_top_k :152-174, SparseDispatcher :250-293.  All activations are token-major (T, C) float32
// (= NHWC); HBM-bound kernels read/write 16 B per lane with lanes running over channels (coalesced), reductions
// stay inside a (sub-)wavefront, and nothing on this path synchronises with the host (the reference's
// `SparseDispatcher` does `.cpu()` per MoE block, :259).
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ float hsum4(f32x4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ============================================================================================== stem patchify
// x (B,3,H,W) NCHW -> a (B*H/4*W/4, 64): column c*16 + kh*4 + kw for c < 3 (= Conv2d(3,C0,4,4).weight.view(C0,48)
// order), columns 48..63 zero (K padded to the GEMM's BK multiple).
__global__ void stem_patchify_kernel(const float* __restrict__ x, float* __restrict__ a, int B, int H, int W) {
  const int Ho = H / 4, Wo = W / 4;
  const long total = (long)B * Ho * Wo * 16;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = idx & 15;  // c*4 + kh, 12..15 = zero pad
    const long tok = idx >> 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (q < 12) {
      const int c = q >> 2, kh = q & 3;
      const int wo = tok % Wo;
      const long t2 = tok / Wo;
      const int ho = t2 % Ho;
      const int b = t2 / Ho;
      v = ld4(x + (((long)b * 3 + c) * H + (4 * ho + kh)) * W + 4 * wo);
    }
    st4(a + tok * 64 + q * 4, v);
  }
}


// ---- column reductions of the row kernels ------------------------------------------------------------------------
// Each lane accumulates NSETS x NV float4 column partials over the tokens it visited.  They are folded across the
// wave's token groups (shuffles), across the block's 4 waves (LDS) and written to partials[block][set][C]; a second
// tiny kernel sums over blocks.  No atomics: contended same-address L2 atomics cost ~200 us per launch here.
constexpr int ROW_MAX_BLOCKS = 512;
template <int G, int NV, int NSETS>
__device__ __forceinline__ void block_colsum_store(f32x4 (&acc)[NSETS][NV], float* __restrict__ partials, int C) {
  extern __shared__ __attribute__((aligned(16))) float s_red[];  // [4][NSETS][C]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lg = lane % G, tg = lane / G;
  const int nq = C >> 2;
#pragma unroll
  for (int s = 0; s < NSETS; s++)
#pragma unroll
    for (int i = 0; i < NV; i++) {
      f32x4 v = acc[s][i];
#pragma unroll
      for (int j = 0; j < 4; j++)
        for (int o = G; o < 64; o <<= 1) v[j] += __shfl_xor(v[j], o, 64);
      const int q = lg + i * G;
      if (tg == 0 && q < nq) st4(s_red + ((long)wv * NSETS + s) * C + 4 * q, v);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < NSETS * C; i += blockDim.x) {
    const float t = (s_red[i] + s_red[(long)NSETS * C + i]) + (s_red[2L * NSETS * C + i] + s_red[3L * NSETS * C + i]);
    partials[(long)blockIdx.x * NSETS * C + i] = t;
  }
}
// out[c] = sum_b partials[b][c]; block = 16 column quads (64 columns, 16 B per lane) x 64 row lanes, 8 independent
// loads in flight per thread: <= 512 partial rows are ONE round trip.  The partials were just written by another kernel
// (cold in this XCD's L2, ~2 us per dependent trip inside the step): one lane per column walking them four at a time
// cost 11-15 us per call, 44 calls per training step.
constexpr int PR_THREADS = 1024;
__global__ __launch_bounds__(PR_THREADS) void partials_reduce_kernel(const float* __restrict__ partials, int nblocks,
                                                                    int ncols, float* __restrict__ out) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  __shared__ v4 red[64][17];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + 4 * cq;  // ncols is a multiple of 4 (C is)
  v4 s[8];
#pragma unroll
  for (int u = 0; u < 8; u++) s[u] = v4{0.f, 0.f, 0.f, 0.f};
  if (c < ncols) {
    for (int b = rl; b < nblocks; b += 512) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int r = b + 64 * u;
        const v4 v = *reinterpret_cast<const v4*>(partials + (long)min(r, nblocks - 1) * ncols + c);
        if (r < nblocks) s[u] += v;  // clamped address + select: the batch of loads stays in flight
      }
    }
  }
  red[rl][cq] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
  __syncthreads();
  if (threadIdx.x < 256) {  // 4 partial sums of 16 row lanes per column, then 4 -> 1 through a shuffle
    const int col = threadIdx.x >> 2, part = threadIdx.x & 3;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) t += red[16 * part + i][col >> 2][col & 3];
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    const int oc = blockIdx.x * 64 + col;
    if (part == 0 && oc < ncols) out[oc] = t;
  }
}

// ============================================================================================== LayerNorm rows
// out_mode 0: y row = token; 1: patch-major rows for the 2x2/s2 downsample conv: token (b,h,w) -> row
// (b,h/2,w/2), column block ((h&1)*2 + (w&1))*C  (so the conv becomes one NT GEMM with K = 4C).
// out_mode 2: token rows like mode 0, stored as fp16 (the AMP data path: the block norm's output only feeds GEMMs whose
// operands are rounded to half anyway -- the reference's autocast casts it on entry to nn.Linear).
__device__ __forceinline__ long ln_out_offset(long tok, int C, int mode, int H, int W) {
  if (mode == 0 || mode == 2) return tok * C;
  const int w = tok % W;
  const long t2 = tok / W;
  const int h = t2 % H;
  const long b = t2 / H;
  const long row = (b * (H / 2) + (h >> 1)) * (W / 2) + (w >> 1);
  return row * (4L * C) + (long)(((h & 1) << 1) | (w & 1)) * C;
}

// X16: the rows of x are stored as fp16 (the depthwise output of the AMP data path); statistics and arithmetic in fp32 as
// F.layer_norm under autocast (convnext_moe.py:30-47)
__device__ __forceinline__ f32x4 ld4h(const float* base, long o) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  const f16x4 h = *reinterpret_cast<const f16x4*>(reinterpret_cast<const _Float16*>(base) + o);
  return f32x4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}

template <int G, int NV, int X16>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, float eps,
                                                           float* __restrict__ y, float* __restrict__ mean_o,
                                                           float* __restrict__ rstd_o, long T, int C, int mode, int H,
                                                           int W) {
  constexpr int TPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int lg = lane % G, tg = lane / G;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const int nq = C >> 2;
  for (long t0 = wave * TPW; t0 < T; t0 += nwaves * TPW) {
    const long tok = t0 + tg;
    const bool tv = tok < T;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int q = lg + i * G;
      v[i] = (tv && q < nq) ? (X16 ? ld4h(x, tok * C + 4 * q) : ld4(x + tok * C + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
      s += hsum4(v[i]);
    }
    const float mean = group_sum<G>(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int q = lg + i * G;
      if (q < nq) {
        f32x4 d = v[i] - mean;
        ss += hsum4(d * d);
      }
    }
    const float var = group_sum<G>(ss) / (float)C;
    const float rstd = rsqrtf(var + eps);
    if (!tv) continue;
    const long ob = ln_out_offset(tok, C, mode, H, W);
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int q = lg + i * G;
      if (q < nq) {
        const f32x4 o = (v[i] - mean) * rstd * ld4(w + 4 * q) + ld4(b + 4 * q);
        if (mode == 2) {
          typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
          *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(y) + ob + 4 * q) =
              f16x4{(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
        } else {
          st4(y + ob + 4 * q, o);
        }
      }
    }
    if (lg == 0) {
      if (mean_o) mean_o[tok] = mean;
      if (rstd_o) rstd_o[tok] = rstd;
    }
  }
}

// backward: dx = rstd * (dyh - mean(dyh) - xh * mean(dyh * xh)), dyh = dy * w ; dw += dy * xh ; db += dy
// `dy` is read through the same out_mode mapping the forward wrote y with.  dwdb (2C) must be zeroed by the caller.
template <int G, int NV, int X16>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ mean_i,
                                                           const float* __restrict__ rstd_i, float* __restrict__ dx,
                                                           float* __restrict__ partials, long T, int C, int mode,
                                                           int H, int W, int accumulate_dx) {
  constexpr int TPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int lg = lane % G, tg = lane / G;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const int nq = C >> 2;
  f32x4 acc[2][NV], wv[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    acc[0][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = lg + i * G;
    wv[i] = (q < nq) ? ld4(w + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // U token groups per iteration: all 2*U*NV 16-byte loads of an iteration are issued before the first reduction, so
  // a wave keeps U times the bytes in flight (the kernel runs at <= 512 workgroups = 2 waves per SIMD).
  constexpr int U = NV <= 4 ? 2 : 1;
  for (long t0 = wave * TPW; t0 < T; t0 += nwaves * TPW * U) {
    long tok[U];
    bool tv[U];
    long ob[U];
    float mean[U], rstd[U];
    f32x4 xh[U][NV], g[U][NV];
#pragma unroll
    for (int u = 0; u < U; u++) {
      tok[u] = t0 + (long)u * nwaves * TPW + tg;
      tv[u] = tok[u] < T;
      ob[u] = tv[u] ? ln_out_offset(tok[u], C, mode, H, W) : 0;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int q = lg + i * G;
        const bool ok = tv[u] && q < nq;
        xh[u][i] = ok ? (X16 ? ld4h(x, tok[u] * C + 4 * q) : ld4(x + tok[u] * C + 4 * q)) : f32x4{0.f, 0.f, 0.f, 0.f};
        g[u][i] = ok ? ld4(dy + ob[u] + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      mean[u] = tv[u] ? mean_i[tok[u]] : 0.f;
      rstd[u] = tv[u] ? rstd_i[tok[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int q = lg + i * G;
        if (tv[u] && q < nq) {
          xh[u][i] = (xh[u][i] - mean[u]) * rstd[u];
          const f32x4 d = g[u][i];
          acc[0][i] += d * xh[u][i];
          acc[1][i] += d;
          g[u][i] = d * wv[i];
          s1 += hsum4(g[u][i]);
          s2 += hsum4(g[u][i] * xh[u][i]);
        }
      }
      const float c1 = group_sum<G>(s1) / (float)C;
      const float c2 = group_sum<G>(s2) / (float)C;
      if (!tv[u]) continue;
#pragma unroll
      for (int i = 0; i < NV; i++) {
        const int q = lg + i * G;
        if (q < nq) {
          f32x4 r = (g[u][i] - c1 - xh[u][i] * c2) * rstd[u];
          float* o = dx + tok[u] * C + 4 * q;
          if (accumulate_dx) r += ld4(o);
          st4(o, r);
        }
      }
    }
  }
  block_colsum_store<G, NV, 2>(acc, partials, C);
}

// ============================================================================================== depthwise 7x7
// y[b,h,w,c] = bias[c] + sum_{ky,kx} x[b,h+ky-3,w+kx-3,c] * w49[ky*7+kx][c]  (+ addend[b,h,w,c]), zero padding.
// Thread = one channel quad x a 2x8 output strip; lanes run over channel quads (16 B each, coalesced); the 8x14
// input patch is streamed row by row and reused for both output rows from registers.
constexpr int DW_RY = 2, DW_RX = 8;
__global__ __launch_bounds__(256) void dwconv7_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w49,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ addend, float* __restrict__ y,
                                                         int B, int H, int W, int C, int spb, int flip) {
  const int nq = C >> 2;
  const int cq = threadIdx.x % nq;
  const int sl = threadIdx.x / nq;
  if (sl >= spb) return;
  const int sw = (W + DW_RX - 1) / DW_RX, sh = (H + DW_RY - 1) / DW_RY;
  const long strip = (long)blockIdx.x * spb + sl;
  const long nstrips = (long)B * sh * sw;
  if (strip >= nstrips) return;
  const int sx = strip % sw;
  const long t2 = strip / sw;
  const int sy = t2 % sh;
  const int b = t2 / sh;
  const int ox0 = sx * DW_RX, oy0 = sy * DW_RY;
  f32x4 acc[DW_RY][DW_RX];
  const f32x4 bv = bias ? ld4(bias + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < DW_RY; r++)
#pragma unroll
    for (int c = 0; c < DW_RX; c++) acc[r][c] = bv;
  const float* xb = x + (long)b * H * W * C + 4 * cq;
#pragma unroll 1
  for (int ir = 0; ir < DW_RY + 6; ir++) {
    const int iy = oy0 - 3 + ir;
    if (iy < 0 || iy >= H) continue;
    f32x4 in[DW_RX + 6];
#pragma unroll
    for (int c = 0; c < DW_RX + 6; c++) {
      const int ix = ox0 - 3 + c;
      in[c] = (ix >= 0 && ix < W) ? ld4(xb + ((long)iy * W + ix) * C) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int r = 0; r < DW_RY; r++) {
      const int ky = ir - r;
      if (ky < 0 || ky > 6) continue;
#pragma unroll
      for (int kx = 0; kx < 7; kx++) {
        const int t = flip ? 48 - (ky * 7 + kx) : ky * 7 + kx;  // flip: correlation with the reversed taps (dgrad)
        const f32x4 wv = ld4(w49 + (long)t * C + 4 * cq);
#pragma unroll
        for (int c = 0; c < DW_RX; c++) acc[r][c] += in[c + kx] * wv;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < DW_RY; r++)
#pragma unroll
    for (int c = 0; c < DW_RX; c++) {
      if (oy0 + r >= H || ox0 + c >= W) continue;
      const long o = (((long)b * H + oy0 + r) * W + ox0 + c) * C + 4 * cq;
      f32x4 v = acc[r][c];
      if (addend) v += ld4(addend + o);
      st4(y + o, v);
    }
}

// weight / bias gradient: dw49[ky*7+kx][c] += sum_p du[p] * x[p + (ky-3, kx-3)], dbias[c] += sum_p du[p].
// blockIdx.y = ky; each thread owns one channel quad, loops over output strips (1 x 8) accumulating its 7 taps in
// registers, then the block folds its strips through LDS and issues one atomic per (tap, channel).
__global__ __launch_bounds__(256) void dwconv7_bwd_weight_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ du,
                                                                float* __restrict__ dw49, float* __restrict__ dbias,
                                                                int B, int H, int W, int C, int spb) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [spb][8][C]
  const int nq = C >> 2;
  const int cq = threadIdx.x % nq;
  const int sl = threadIdx.x / nq;
  const int ky = blockIdx.y;
  const int sw = (W + DW_RX - 1) / DW_RX;
  const long nstrips = (long)B * H * sw;
  f32x4 aw[7], ab = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 7; k++) aw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (sl < spb) {
    for (long strip = (long)blockIdx.x * spb + sl; strip < nstrips; strip += (long)gridDim.x * spb) {
      const int sx = strip % sw;
      const long t2 = strip / sw;
      const int oy = t2 % H;
      const int b = t2 / H;
      const int ox0 = sx * DW_RX;
      const int iy = oy + ky - 3;
      f32x4 g[DW_RX];
#pragma unroll
      for (int c = 0; c < DW_RX; c++) {
        g[c] = (ox0 + c < W) ? ld4(du + (((long)b * H + oy) * W + ox0 + c) * C + 4 * cq)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
        if (ky == 0) ab += g[c];
      }
      if (iy < 0 || iy >= H) continue;
      f32x4 in[DW_RX + 6];
#pragma unroll
      for (int c = 0; c < DW_RX + 6; c++) {
        const int ix = ox0 - 3 + c;
        in[c] = (ix >= 0 && ix < W) ? ld4(x + (((long)b * H + iy) * W + ix) * C + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int kx = 0; kx < 7; kx++)
#pragma unroll
        for (int c = 0; c < DW_RX; c++) aw[kx] += g[c] * in[c + kx];
    }
  }
  if (sl < spb) {
#pragma unroll
    for (int k = 0; k < 7; k++) st4(red + ((long)sl * 8 + k) * C + 4 * cq, aw[k]);
    st4(red + ((long)sl * 8 + 7) * C + 4 * cq, ab);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * C; i += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < spb; j++) s += red[(long)j * 8 * C + i];
    const int k = i / C, c = i - k * C;
    if (k < 7) atomicAdd(dw49 + (long)(ky * 7 + k) * C + c, s);
    else if (ky == 0) atomicAdd(dbias + c, s);
  }
}

// ============================================================================================== layer-scale backward prep
// dense block: dY = gamma * rs[b] * dOut ; dgamma[c] = sum_t rs[b] * dOut[t,c] * Y[t,c] ; db2[c] = sum_t dY[t,c]
template <int G, int NV>
__global__ __launch_bounds__(256) void scale_bwd_prep_kernel(const float* __restrict__ dout,
                                                            const float* __restrict__ yv,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ rowscale, int rows_per_scale,
                                                            float* __restrict__ dy, float* __restrict__ partials, long T,
                                                            int C) {
  constexpr int TPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int lg = lane % G, tg = lane / G;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const int nq = C >> 2;
  f32x4 acc[2][NV], gv[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    acc[0][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = lg + i * G;
    gv[i] = (q < nq) ? ld4(gamma + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long t0 = wave * TPW; t0 < T; t0 += nwaves * TPW) {
    const long tok = t0 + tg;
    if (tok >= T) continue;
    const float rs = rowscale ? rowscale[tok / rows_per_scale] : 1.f;
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int q = lg + i * G;
      if (q < nq) {
        const f32x4 d = ld4(dout + tok * C + 4 * q) * rs;
        acc[0][i] += d * ld4(yv + tok * C + 4 * q);
        const f32x4 o = d * gv[i];
        acc[1][i] += o;
        st4(dy + tok * C + 4 * q, o);
      }
    }
  }
  block_colsum_store<G, NV, 2>(acc, partials, C);
}

// ============================================================================================== MoE plan (dispatch tables)
// Deterministic expert-major slot assignment without a host sync: slot(t,j) = offsets[e] + rank of (t,j) among the
// pairs routed to e in (t,j) order.  Three launches: per-block histogram, one-block scan, per-block ranks.
constexpr int PLAN_TB = 256;
__global__ __launch_bounds__(PLAN_TB) void moe_hist_kernel(const int32_t* __restrict__ top_idx, int m, int T, int E,
                                                          int k, int32_t* __restrict__ counts) {
  extern __shared__ int s_cnt[];  // E
  for (int i = threadIdx.x; i < E; i += blockDim.x) s_cnt[i] = 0;
  __syncthreads();
  const int t = blockIdx.x * PLAN_TB + threadIdx.x;
  if (t < T)
    for (int j = 0; j < k; j++) atomicAdd(&s_cnt[top_idx[(long)t * m + j]], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < E; i += blockDim.x) counts[(long)blockIdx.x * E + i] = s_cnt[i];
}
// counts (nblk,E) -> base (nblk,E) exclusive over blocks + expert offsets; offsets (E+1).  One wave per expert (waves
// stride over the experts): every lane owns a contiguous chunk of the histogram rows, the chunks meet in a wave scan
// (one lane per expert walking all rows twice was ~10 us of dependent loads per MoE block).
__device__ __forceinline__ int scan_chunk_sum(const int32_t* __restrict__ counts, int E, int e, int b0, int b1) {
  int s = 0;
  for (int b = b0; b < b1; b++) s += counts[(long)b * E + e];
  return s;
}
__global__ __launch_bounds__(1024) void moe_scan_kernel(const int32_t* __restrict__ counts, int nblk, int E,
                                                       int32_t* __restrict__ base, int32_t* __restrict__ offsets) {
  extern __shared__ int s_tot[];  // E+1
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int per = (nblk + 63) / 64;
  const int b0 = min(nblk, lane * per), b1 = min(nblk, b0 + per);
  for (int e = wave; e < E; e += nw) {
    int t = scan_chunk_sum(counts, E, e, b0, b1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) s_tot[e] = t;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int i = 0; i < E; i++) {
      const int c = s_tot[i];
      s_tot[i] = run;
      offsets[i] = run;
      run += c;
    }
    offsets[E] = run;
  }
  __syncthreads();
  for (int e = wave; e < E; e += nw) {
    const int mine = scan_chunk_sum(counts, E, e, b0, b1);
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    int run = s_tot[e] + incl - mine;
    for (int b = b0; b < b1; b++) {
      base[(long)b * E + e] = run;
      run += counts[(long)b * E + e];
    }
  }
}
__global__ __launch_bounds__(PLAN_TB) void moe_rank_kernel(const int32_t* __restrict__ top_idx, int m, int T, int E,
                                                          int k, const int32_t* __restrict__ base,
                                                          int32_t* __restrict__ slot_token,
                                                          int32_t* __restrict__ token_slot) {
  __shared__ int s_wave[PLAN_TB / 64];
  const int t = blockIdx.x * PLAN_TB + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int e = 0; e < E; e++) {
    // number of this thread's (t,j) pairs routed to e (0 or 1: top-k indices are distinct), and which j
    int jj = -1;
    if (t < T)
      for (int j = 0; j < k; j++)
        if (top_idx[(long)t * m + j] == e) jj = j;
    const unsigned long long bal = __ballot(jj >= 0);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wv; w++) woff += s_wave[w];
    if (jj >= 0) {
      const int slot = base[(long)blockIdx.x * E + e] + woff + before;
      slot_token[slot] = t;
      token_slot[(long)t * k + jj] = slot;
    }
    __syncthreads();
  }
}

// Xslot[s,:] = X[slot_token[s],:]
__global__ void moe_dispatch_kernel(const float* __restrict__ x, const int32_t* __restrict__ slot_token,
                                    float* __restrict__ xs, long S, int C) {
  const int nq = C >> 2;
  const long total = S * nq;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long s = idx / nq;
    const int q = idx - s * nq;
    st4(xs + s * C + 4 * q, ld4(x + (long)slot_token[s] * C + 4 * q));
  }
}

// out[t,:] = shortcut[t,:] + gamma * rs[b] * sum_j gates[t,j] * Yslot[token_slot[t,j],:]     (combine :269-284, :368-370)
__global__ void moe_combine_fwd_kernel(const float* __restrict__ yslot, const int32_t* __restrict__ token_slot,
                                       const float* __restrict__ gates, const float* __restrict__ shortcut,
                                       const float* __restrict__ gamma, const float* __restrict__ rowscale,
                                       int rows_per_scale, float* __restrict__ out, long T, int C, int k) {
  const int nq = C >> 2;
  const long total = T * nq;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long t = idx / nq;
    const int q = idx - t * nq;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < k; j0 += 4) {  // indices of up to four experts, then their vectors: two round trips, not 2 k
      long sl[4];
      float g[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const bool on = j0 + u < k;
        sl[u] = on ? token_slot[t * k + j0 + u] : 0;
        g[u] = on ? gates[t * k + j0 + u] : 0.f;
      }
      f32x4 yv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) yv[u] = j0 + u < k ? ld4(yslot + sl[u] * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (j0 + u < k) acc += yv[u] * g[u];
    }
    float sc = rowscale ? rowscale[t / rows_per_scale] : 1.f;
    st4(out + t * C + 4 * q, ld4(shortcut + t * C + 4 * q) + ld4(gamma + 4 * q) * sc * acc);
  }
}

// backward of the combine: d = gamma * rs * dOut[t] ; dYslot[slot_j] = g_j * d ; dgate[t,j] = <d, Yslot[slot_j]> ;
// dgamma[c] += rs * dOut[t,c] * sum_j g_j Yslot[slot_j][c]
template <int G, int NV>
__global__ __launch_bounds__(256) void moe_combine_bwd_kernel(
    const float* __restrict__ dout, const float* __restrict__ yslot, const int32_t* __restrict__ token_slot,
    const float* __restrict__ gates, const float* __restrict__ gamma, const float* __restrict__ rowscale,
    int rows_per_scale, float* __restrict__ dyslot, float* __restrict__ dgate, float* __restrict__ partials, long T,
    int C, int k) {
  constexpr int TPW = 64 / G;
  const int lane = threadIdx.x & 63;
  const int lg = lane % G, tg = lane / G;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
  const int nq = C >> 2;
  f32x4 acc[1][NV], gv[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    acc[0][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q = lg + i * G;
    gv[i] = (q < nq) ? ld4(gamma + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  for (long t0 = wave * TPW; t0 < T; t0 += nwaves * TPW) {
    const long tok = t0 + tg;
    const bool tv = tok < T;
    const float rs = (tv && rowscale) ? rowscale[tok / rows_per_scale] : 1.f;
    f32x4 d[NV], ym[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int q = lg + i * G;
      d[i] = (tv && q < nq) ? ld4(dout + tok * C + 4 * q) * rs : f32x4{0.f, 0.f, 0.f, 0.f};
      ym[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // all k slot indices and gates first, then all k x NV expert-output vectors (independent requests: one round trip for
    // the indices, one for the vectors -- the j-loop used to chain index -> vector -> store per expert), then the math
    // experts per batch (top-k of the SM3Det configs: 2).  The widest rows (NV > 4: C > 1024, no SM3Det stage) take one expert
    // at a time: four experts x eight vectors were 128 registers of look-ahead, 141 of them spilled
    constexpr int KB = NV > 4 ? 1 : 4;
    for (int j0 = 0; j0 < k; j0 += KB) {
      float g[KB];
      long sl[KB];
#pragma unroll
      for (int u = 0; u < KB; u++) {
        const bool on = tv && j0 + u < k;
        g[u] = on ? gates[tok * k + j0 + u] : 0.f;
        sl[u] = on ? token_slot[tok * k + j0 + u] : 0;
      }
      f32x4 yv[KB][NV];
#pragma unroll
      for (int u = 0; u < KB; u++)
#pragma unroll
        for (int i = 0; i < NV; i++) {
          const int q = lg + i * G;
          yv[u][i] = (tv && j0 + u < k && q < nq) ? ld4(yslot + sl[u] * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int u = 0; u < KB; u++) {
        if (j0 + u >= k) break;
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < NV; i++) {
          const int q = lg + i * G;
          if (tv && q < nq) {
            const f32x4 dd = d[i] * gv[i];
            dot += hsum4(dd * yv[u][i]);
            ym[i] += yv[u][i] * g[u];
            st4(dyslot + sl[u] * C + 4 * q, dd * g[u]);
          }
        }
        dot = group_sum<G>(dot);
        if (tv && lg == 0) dgate[tok * k + j0 + u] = dot;
      }
    }
#pragma unroll
    for (int i = 0; i < NV; i++) acc[0][i] += d[i] * ym[i];
  }
  block_colsum_store<G, NV, 1>(acc, partials, C);
}

// dx[t,:] (+)= sum_j dXslot[token_slot[t,j],:]
__global__ void moe_gather_add_kernel(const float* __restrict__ dxslot, const int32_t* __restrict__ token_slot,
                                      float* __restrict__ dx, long T, int C, int k, int accumulate) {
  const int nq = C >> 2;
  const long total = T * nq;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long t = idx / nq;
    const int q = idx - t * nq;
    f32x4 acc = accumulate ? ld4(dx + t * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < k; j0 += 4) {
      long sl[4];
#pragma unroll
      for (int u = 0; u < 4; u++) sl[u] = j0 + u < k ? token_slot[t * k + j0 + u] : 0;
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = j0 + u < k ? ld4(dxslot + sl[u] * C + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (j0 + u < k) acc += v[u];
    }
    st4(dx + t * C + 4 * q, acc);
  }
}

// blocks of 4 waves for a row kernel that handles 64/G tokens per wave-iteration (grid-stride beyond 4096 blocks)
inline int row_blocks(long T, int G) {
  const long waves = (T * G + 63) / 64;
  long b = (waves + 3) / 4;
  if (b > 4096) b = 4096;
  return (int)(b < 1 ? 1 : b);
}
inline int row_blocks_capped(long T, int G) {
  const int b = row_blocks(T, G);
  return b > ROW_MAX_BLOCKS ? ROW_MAX_BLOCKS : b;
}
inline int ew_blocks(long work, int threads = 256) {
  long b = (work + threads - 1) / threads;
  if (b > 256L * 16) b = 256L * 16;
  return (int)(b < 1 ? 1 : b);
}

// (G, NV) dispatch for the row kernels: lanes per token and float4s per lane
#define SM3_ROW_DISPATCH(C, CALL)                                   \
  do {                                                              \
    const int nq_ = (C) >> 2;                                       \
    if (nq_ <= 16) { CALL(16, 1); }                                 \
    else if (nq_ <= 32) { CALL(32, 1); }                            \
    else if (nq_ <= 64) { CALL(64, 1); }                            \
    else if (nq_ <= 128) { CALL(64, 2); }                           \
    else if (nq_ <= 192) { CALL(64, 3); }                           \
    else if (nq_ <= 256) { CALL(64, 4); }                           \
    else if (nq_ <= 512) { CALL(64, 8); }                           \
    else return SM3_ERR_UNSUPPORTED;                                \
  } while (0)

}  // namespace

// LDS-tiled depthwise kernels (dwconv_lds.hip)
bool sm3_dwconv7_lds_supported(int H, int W, int C);
void sm3_dwconv7_lds_fwd(const float* x, const float* w49, const float* bias, const float* addend, float* y, int B,
                         int H, int W, int C, int flip, hipStream_t st);
void sm3_dwconv7_lds_bwd_weight(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                                hipStream_t st);

extern "C" {

int sm3_stem_patchify(const float* x, float* a, int B, int H, int W, sm3_stream_t stream) {
  if (!x || !a || B <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3)) return SM3_ERR_INVALID_ARG;
  const long total = (long)B * (H / 4) * (W / 4) * 16;
  stem_patchify_kernel<<<ew_blocks(total), 256, 0, (hipStream_t)stream>>>(x, a, B, H, W);
  return launch_status();
}

namespace {
// fp16 shadow of an fp32 tensor (round-to-nearest-even, like a torch .half() cast): the half weights of the AMP data path
__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, long n4) {
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<f16x4*>(y)[i] = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  }
}
}  // namespace

int sm3_cast_f32_f16(const float* src, void* dst, long n, sm3_stream_t stream) {
  if (n < 0 || (n & 3)) return SM3_ERR_INVALID_ARG;
  if (n == 0) return SM3_OK;
  if (!src || !dst) return SM3_ERR_INVALID_ARG;
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  cast_f32_f16_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(src, reinterpret_cast<_Float16*>(dst), n / 4);
  return launch_status();
}

int sm3_layernorm_fwd(const float* x, const float* w, const float* b, float eps, float* y, float* mean, float* rstd,
                      long T, int C, int out_mode, int H, int W, sm3_stream_t stream) {
  const int x16 = out_mode & 16;  // SM3_LN_X_F16
  out_mode &= ~16;
  if (!x || !w || !b || !y || T < 0 || C <= 0 || (C & 3) || out_mode < 0 || out_mode > 2) return SM3_ERR_INVALID_ARG;
  if (out_mode == 1 && ((H & 1) || (W & 1))) return SM3_ERR_INVALID_ARG;
  if (T == 0) return SM3_OK;
  hipStream_t st = (hipStream_t)stream;
#define CALL(G, NV)                                                                                                      \
  if (x16) layernorm_fwd_kernel<G, NV, 1><<<row_blocks(T, G), 256, 0, st>>>(x, w, b, eps, y, mean, rstd, T, C, out_mode, H, W); \
  else layernorm_fwd_kernel<G, NV, 0><<<row_blocks(T, G), 256, 0, st>>>(x, w, b, eps, y, mean, rstd, T, C, out_mode, H, W)
  SM3_ROW_DISPATCH(C, CALL);
#undef CALL
  return launch_status();
}

// The same reduction for up to PRM_MAX (partials, out) pairs in ONE launch: the parameter gradients the row kernels of a
// whole backward pass leave as per-workgroup partial rows (d LayerNorm weight / bias, d layer scale, d bias of every block:
// 36 pairs per ConvNeXt-T step) are only needed by the optimizer, so their reductions are collected and run together when
// the backward pass ends (36 launches of ~5 us at the launch floor -> 1).  The table travels by value in the kernel
// arguments: nothing to upload, capturable.
constexpr int PRM_MAX = 64;
struct PartialsTable {
  const float* partials[PRM_MAX];
  float* out[PRM_MAX];
  int nblocks[PRM_MAX], ncols[PRM_MAX], block0[PRM_MAX + 1];
  int n;
};
__global__ __launch_bounds__(PR_THREADS) void partials_reduce_multi_kernel(const PartialsTable t) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  __shared__ v4 red[64][17];
  int e = 0;
  while (e + 1 < t.n && (int)blockIdx.x >= t.block0[e + 1]) e++;
  const float* __restrict__ partials = t.partials[e];
  const int nblocks = t.nblocks[e], ncols = t.ncols[e], bx = blockIdx.x - t.block0[e];
  const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = bx * 64 + 4 * cq;
  v4 s[8];
#pragma unroll
  for (int u = 0; u < 8; u++) s[u] = v4{0.f, 0.f, 0.f, 0.f};
  if (c < ncols) {
    for (int b = rl; b < nblocks; b += 512) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int r = b + 64 * u;
        const v4 v = *reinterpret_cast<const v4*>(partials + (long)min(r, nblocks - 1) * ncols + c);
        if (r < nblocks) s[u] += v;
      }
    }
  }
  red[rl][cq] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));  // same order as the single form
  __syncthreads();
  if (threadIdx.x < 256) {
    const int col = threadIdx.x >> 2, part = threadIdx.x & 3;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) v += red[16 * part + i][col >> 2][col & 3];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    const int oc = bx * 64 + col;
    if (part == 0 && oc < ncols) t.out[e][oc] = v;
  }
}

// number of per-workgroup partial rows the row kernels (layernorm_bwd, scale_bwd_prep, moe_combine_bwd) leave in the
// workspace for (T, C); with the reduction output pointer NULL those kernels skip their own reduce and the caller
// runs sm3_row_partials_reduce -- on another stream if it likes (the results are parameter gradients).
int sm3_row_partial_blocks(long T, int C) {
  if (T <= 0 || C <= 0 || (C & 3)) return 0;
  int nb = 0;
#define CALL(G, NV) nb = row_blocks_capped(T, G)
  SM3_ROW_DISPATCH(C, CALL);
#undef CALL
  return nb;
}

int sm3_row_partials_reduce(const float* partials, int nblocks, int ncols, float* out, sm3_stream_t stream) {
  if (!partials || !out || nblocks <= 0 || ncols <= 0 || (ncols & 3)) return SM3_ERR_INVALID_ARG;  // 16-byte column quads
  partials_reduce_kernel<<<(ncols + 63) / 64, PR_THREADS, 0, (hipStream_t)stream>>>(partials, nblocks, ncols, out);
  return launch_status();
}

int sm3_row_partials_reduce_multi(const float* const* partials, float* const* outs, const int* nblocks, const int* ncols,
                                  int n, sm3_stream_t stream) {
  if (n < 0 || (n > 0 && (!partials || !outs || !nblocks || !ncols))) return SM3_ERR_INVALID_ARG;
  for (int i0 = 0; i0 < n; i0 += PRM_MAX) {
    PartialsTable t;
    t.n = n - i0 < PRM_MAX ? n - i0 : PRM_MAX;
    int blocks = 0;
    for (int i = 0; i < t.n; i++) {
      const int j = i0 + i;
      if (!partials[j] || !outs[j] || nblocks[j] <= 0 || ncols[j] <= 0 || (ncols[j] & 3)) return SM3_ERR_INVALID_ARG;
      t.partials[i] = partials[j]; t.out[i] = outs[j]; t.nblocks[i] = nblocks[j]; t.ncols[i] = ncols[j];
      t.block0[i] = blocks;
      blocks += (ncols[j] + 63) / 64;
    }
    t.block0[t.n] = blocks;
    partials_reduce_multi_kernel<<<blocks, PR_THREADS, 0, (hipStream_t)stream>>>(t);
  }
  return launch_status();
}

size_t sm3_row_reduce_workspace_bytes(int C) { return (size_t)ROW_MAX_BLOCKS * 2 * (C > 0 ? C : 1) * sizeof(float); }

int sm3_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                      float* dx, float* dwdb, long T, int C, int out_mode, int H, int W, int accumulate_dx,
                      void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  const int x16 = out_mode & 16;  // SM3_LN_X_F16
  out_mode &= ~16;
  if (!dy || !x || !w || !mean || !rstd || !dx || !workspace || T <= 0 || C <= 0 || (C & 3) || out_mode < 0 || out_mode > 2)
    return SM3_ERR_INVALID_ARG;
  if (workspace_bytes < sm3_row_reduce_workspace_bytes(C)) return SM3_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  int nb = 1;
  const size_t lds = (size_t)4 * 2 * C * sizeof(float);
#define CALL(G, NV)                                                                                                 \
  nb = row_blocks_capped(T, G);                                                                                     \
  if (x16)                                                                                                          \
    layernorm_bwd_kernel<G, NV, 1><<<nb, 256, lds, st>>>(dy, x, w, mean, rstd, dx, part, T, C, out_mode, H, W, accumulate_dx); \
  else                                                                                                              \
    layernorm_bwd_kernel<G, NV, 0><<<nb, 256, lds, st>>>(dy, x, w, mean, rstd, dx, part, T, C, out_mode, H, W, accumulate_dx)
  SM3_ROW_DISPATCH(C, CALL);
#undef CALL
  if (dwdb) partials_reduce_kernel<<<(2 * C + 63) / 64, PR_THREADS, 0, st>>>(part, nb, 2 * C, dwdb);
  return launch_status();
}

int sm3_dwconv7_fwd(const float* x, const float* w49, const float* bias, const float* addend, float* y, int B, int H,
                    int W, int C, int flip, sm3_stream_t stream) {
  if (!x || !w49 || !y || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || C > 1024 || (flip & ~33)) return SM3_ERR_INVALID_ARG;
  if ((flip & 32) && !sm3_dwconv7_lds_supported(H, W, C)) return SM3_ERR_UNSUPPORTED;  // fp16 output: the LDS-tiled kernels only
  if (sm3_dwconv7_lds_supported(H, W, C)) {
    sm3_dwconv7_lds_fwd(x, w49, bias, addend, y, B, H, W, C, flip, (hipStream_t)stream);
    return launch_status();
  }
  const int nq = C / 4;
  const int spb = 256 / nq > 0 ? 256 / nq : 1;
  const long nstrips = (long)B * ((H + DW_RY - 1) / DW_RY) * ((W + DW_RX - 1) / DW_RX);
  const int blocks = (int)((nstrips + spb - 1) / spb);
  dwconv7_fwd_kernel<<<blocks, nq * spb, 0, (hipStream_t)stream>>>(x, w49, bias, addend, y, B, H, W, C, spb, flip);
  return launch_status();
}

static int dwconv7_bwd_weight_impl(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                                   int zero_fill, sm3_stream_t stream) {
  if (!x || !du || !dw49 || !dbias || B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || C > 1024)
    return SM3_ERR_INVALID_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (!zero_fill) {
    // sm3_dwconv7_bwd_weight_acc: the kernels ADD their partial sums (atomics) into what the caller provides
  } else if (dbias == dw49 + (size_t)49 * C) {  // one (50, C) buffer [dw49; dbias]: one fill
    sm3_zero_async(dw49, sizeof(float) * 50 * C, st);
  } else {
    sm3_zero_async(dw49, sizeof(float) * 49 * C, st);
    sm3_zero_async(dbias, sizeof(float) * C, st);
  }
  if (sm3_dwconv7_lds_supported(H, W, C)) {
    sm3_dwconv7_lds_bwd_weight(x, du, dw49, dbias, B, H, W, C, st);
    return launch_status();
  }
  const int nq = C / 4;
  const int spb = 256 / nq > 0 ? 256 / nq : 1;
  const long nstrips = (long)B * H * ((W + DW_RX - 1) / DW_RX);
  long blocks = (nstrips + spb - 1) / spb;
  if (blocks > 160) blocks = 160;  // x 7 tap rows ~ 4 blocks per CU; each block loops over strips
  dim3 grid((int)blocks, 7);
  const size_t lds = (size_t)spb * 8 * C * sizeof(float);
  dwconv7_bwd_weight_kernel<<<grid, nq * spb, lds, st>>>(x, du, dw49, dbias, B, H, W, C, spb);
  return launch_status();
}

int sm3_dwconv7_bwd_weight(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                           sm3_stream_t stream) {
  return dwconv7_bwd_weight_impl(x, du, dw49, dbias, B, H, W, C, 1, stream);
}

int sm3_dwconv7_bwd_weight_acc(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                               sm3_stream_t stream) {
  return dwconv7_bwd_weight_impl(x, du, dw49, dbias, B, H, W, C, 0, stream);
}

int sm3_scale_bwd_prep(const float* dout, const float* y, const float* gamma, const float* rowscale,
                       int rows_per_scale, float* dy, float* dgamma_db, long T, int C, void* workspace,
                       size_t workspace_bytes, sm3_stream_t stream) {
  if (!dout || !y || !gamma || !dy || !workspace || T <= 0 || C <= 0 || (C & 3)) return SM3_ERR_INVALID_ARG;
  if (workspace_bytes < sm3_row_reduce_workspace_bytes(C)) return SM3_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (rows_per_scale <= 0) rows_per_scale = 1;
  float* part = (float*)workspace;
  int nb = 1;
  const size_t lds = (size_t)4 * 2 * C * sizeof(float);
#define CALL(G, NV)                 \
  nb = row_blocks_capped(T, G);     \
  scale_bwd_prep_kernel<G, NV><<<nb, 256, lds, st>>>(dout, y, gamma, rowscale, rows_per_scale, dy, part, T, C)
  SM3_ROW_DISPATCH(C, CALL);
#undef CALL
  if (dgamma_db) partials_reduce_kernel<<<(2 * C + 63) / 64, PR_THREADS, 0, st>>>(part, nb, 2 * C, dgamma_db);
  return launch_status();
}

size_t sm3_moe_plan_workspace_bytes(int T, int E) {
  const int nblk = (T + PLAN_TB - 1) / PLAN_TB;
  return (size_t)2 * nblk * E * sizeof(int32_t);
}

int sm3_moe_plan(const int32_t* top_idx, int m, int T, int E, int k, int32_t* offsets, int32_t* slot_token,
                 int32_t* token_slot, void* workspace, size_t workspace_bytes, sm3_stream_t stream) {
  if (!top_idx || !offsets || !slot_token || !token_slot || !workspace) return SM3_ERR_INVALID_ARG;
  if (T <= 0 || E < 1 || E > 1024 || k < 1 || k > m) return SM3_ERR_INVALID_ARG;
  if (workspace_bytes < sm3_moe_plan_workspace_bytes(T, E)) return SM3_ERR_WORKSPACE;
  const int nblk = (T + PLAN_TB - 1) / PLAN_TB;
  int32_t* counts = (int32_t*)workspace;
  int32_t* base = counts + (size_t)nblk * E;
  hipStream_t st = (hipStream_t)stream;
  moe_hist_kernel<<<nblk, PLAN_TB, E * sizeof(int), st>>>(top_idx, m, T, E, k, counts);
  const int sthreads = 64 * (E < 16 ? E : 16);
  moe_scan_kernel<<<1, sthreads, (E + 1) * sizeof(int), st>>>(counts, nblk, E, base, offsets);
  moe_rank_kernel<<<nblk, PLAN_TB, 0, st>>>(top_idx, m, T, E, k, base, slot_token, token_slot);
  return launch_status();
}

int sm3_moe_dispatch(const float* x, const int32_t* slot_token, float* xslot, long S, int C, sm3_stream_t stream) {
  if (!x || !slot_token || !xslot || S < 0 || C <= 0 || (C & 3)) return SM3_ERR_INVALID_ARG;
  if (S == 0) return SM3_OK;
  moe_dispatch_kernel<<<ew_blocks(S * (C / 4)), 256, 0, (hipStream_t)stream>>>(x, slot_token, xslot, S, C);
  return launch_status();
}

int sm3_moe_combine_fwd(const float* yslot, const int32_t* token_slot, const float* gates, const float* shortcut,
                        const float* gamma, const float* rowscale, int rows_per_scale, float* out, long T, int C,
                        int k, sm3_stream_t stream) {
  if (!yslot || !token_slot || !gates || !shortcut || !gamma || !out || T <= 0 || C <= 0 || (C & 3) || k < 1)
    return SM3_ERR_INVALID_ARG;
  if (rows_per_scale <= 0) rows_per_scale = 1;
  moe_combine_fwd_kernel<<<ew_blocks(T * (C / 4)), 256, 0, (hipStream_t)stream>>>(
      yslot, token_slot, gates, shortcut, gamma, rowscale, rows_per_scale, out, T, C, k);
  return launch_status();
}

int sm3_moe_combine_bwd(const float* dout, const float* yslot, const int32_t* token_slot, const float* gates,
                        const float* gamma, const float* rowscale, int rows_per_scale, float* dyslot, float* dgate,
                        float* dgamma, long T, int C, int k, void* workspace, size_t workspace_bytes,
                        sm3_stream_t stream) {
  if (!dout || !yslot || !token_slot || !gates || !gamma || !dyslot || !dgate || !workspace || T <= 0 || C <= 0 ||
      (C & 3) || k < 1)
    return SM3_ERR_INVALID_ARG;
  if (workspace_bytes < sm3_row_reduce_workspace_bytes(C)) return SM3_ERR_WORKSPACE;
  if (rows_per_scale <= 0) rows_per_scale = 1;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  int nb = 1;
  const size_t lds = (size_t)4 * C * sizeof(float);
#define CALL(G, NV)                                                                                                 \
  nb = row_blocks_capped(T, G);                                                                                     \
  moe_combine_bwd_kernel<G, NV><<<nb, 256, lds, st>>>(dout, yslot, token_slot, gates, gamma, rowscale, rows_per_scale, \
                                                      dyslot, dgate, part, T, C, k)
  SM3_ROW_DISPATCH(C, CALL);
#undef CALL
  if (dgamma) partials_reduce_kernel<<<(C + 63) / 64, PR_THREADS, 0, st>>>(part, nb, C, dgamma);
  return launch_status();
}

int sm3_moe_gather_add(const float* dxslot, const int32_t* token_slot, float* dx, long T, int C, int k,
                       int accumulate, sm3_stream_t stream) {
  if (!dxslot || !token_slot || !dx || T <= 0 || C <= 0 || (C & 3) || k < 1) return SM3_ERR_INVALID_ARG;
  moe_gather_add_kernel<<<ew_blocks(T * (C / 4)), 256, 0, (hipStream_t)stream>>>(dxslot, token_slot, dx, T, C, k,
                                                                                accumulate);
  return launch_status();
}

}  // extern "C"
