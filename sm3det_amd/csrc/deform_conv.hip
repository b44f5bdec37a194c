// deform_conv.hip -- DeformConv2d sampling kernels for gfx950 (the three device pieces of the reference op;
// the per-group GEMMs around them run on gemm_f32.hip, orchestrated by sm3det_amd/deform_conv_host.py).
//
// Semantics follow the reference's CPU path (paths relative to /root/reference/mmcv/mmcv/ops/csrc):
//   pytorch/cpu/deform_conv.cpp:5-38   deformable_im2col_bilinear_cpu
//   :40-63                              get_gradient_weight_cpu
//   :65-112                             get_coordinate_weight_cpu
//   :114-164 / :166-219 / :221-290      deformable_im2col / col2im / col2im_coord kernels
// Layouts are the reference's: data_im (parallel_imgs, C, H, W) NCHW; data_offset (parallel_imgs,
// dg*2*kh*kw, Ho, Wo); data_col (C*kh*kw, parallel_imgs, Ho, Wo) with leading dimension `ld_col` (>= parallel_imgs
// *Ho*Wo, padded by the host so the column matrix can feed the float4-vectorised GEMM directly).
// Built with -ffp-contract=off like the other detection ops (CPU-exact rounding of the bilinear weights).
#include "common.h"

namespace {

__device__ __forceinline__ float im2col_bilinear(const float* __restrict__ in, int data_width, int height, int width,
                                                 float h, float w) {
  if (h <= -1 || height <= h || w <= -1 || width <= w) return 0;
  int h_low = (int)floorf(h);
  int w_low = (int)floorf(w);
  int h_high = h_low + 1;
  int w_high = w_low + 1;
  float lh = h - h_low;
  float lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = in[h_low * data_width + w_low];
  float v2 = 0;
  if (h_low >= 0 && w_high <= width - 1) v2 = in[h_low * data_width + w_high];
  float v3 = 0;
  if (h_high <= height - 1 && w_low >= 0) v3 = in[h_high * data_width + w_low];
  float v4 = 0;
  if (h_high <= height - 1 && w_high <= width - 1) v4 = in[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

__device__ __forceinline__ float gradient_weight(float argmax_h, float argmax_w, int h, int w, int height, int width) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int hl = (int)floorf(argmax_h), wl = (int)floorf(argmax_w);
  int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (h == hl && w == wl) weight = (h + 1 - argmax_h) * (w + 1 - argmax_w);
  if (h == hl && w == wh) weight = (h + 1 - argmax_h) * (argmax_w + 1 - w);
  if (h == hh && w == wl) weight = (argmax_h + 1 - h) * (w + 1 - argmax_w);
  if (h == hh && w == wh) weight = (argmax_h + 1 - h) * (argmax_w + 1 - w);
  return weight;
}

__device__ __forceinline__ float coordinate_weight(float argmax_h, float argmax_w, int height, int width,
                                                   const float* __restrict__ im, int data_width, int bp_dir) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int hl = (int)floorf(argmax_h), wl = (int)floorf(argmax_w);
  int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (bp_dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - argmax_w) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += -1 * (argmax_w - wl) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += (wl + 1 - argmax_w) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_w - wl) * im[hh * data_width + wh];
  } else if (bp_dir == 1) {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - argmax_h) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += (hl + 1 - argmax_h) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += -1 * (argmax_h - hl) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_h - hl) * im[hh * data_width + wh];
  }
  return weight;
}

struct DcnGeom {
  int channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, imgs, dg, ho, wo;
  long ld_col;
};

// one thread per (c_im, b, h_col, w_col): writes the kh*kw column entries of that input channel
__global__ __launch_bounds__(256) void deform_im2col_kernel(long n, const float* __restrict__ im,
                                                           const float* __restrict__ off, DcnGeom g,
                                                           float* __restrict__ col) {
  const int cpdg = g.channels / g.dg;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)gridDim.x * blockDim.x) {
    const int w_col = index % g.wo;
    const int h_col = (index / g.wo) % g.ho;
    const int b_col = (index / g.wo / g.ho) % g.imgs;
    const int c_im = (index / g.wo / g.ho) / g.imgs;
    const int c_col = c_im * g.kh * g.kw;
    const int dgi = c_im / cpdg;
    const int h_in = h_col * g.stride_h - g.pad_h;
    const int w_in = w_col * g.stride_w - g.pad_w;
    float* cp = col + (long)c_col * g.ld_col + ((long)b_col * g.ho + h_col) * g.wo + w_col;
    const float* ip = im + ((long)b_col * g.channels + c_im) * g.height * g.width;
    const float* op = off + ((long)b_col * g.dg + dgi) * 2 * g.kh * g.kw * g.ho * g.wo;
    for (int i = 0; i < g.kh; ++i)
      for (int j = 0; j < g.kw; ++j) {
        const long oh = ((long)(2 * (i * g.kw + j)) * g.ho + h_col) * g.wo + w_col;
        const long ow = ((long)(2 * (i * g.kw + j) + 1) * g.ho + h_col) * g.wo + w_col;
        const float offset_h = op[oh], offset_w = op[ow];
        float val = 0.f;
        const float h_im = h_in + i * g.dil_h + offset_h;
        const float w_im = w_in + j * g.dil_w + offset_w;
        if (h_im > -1 && w_im > -1 && h_im < g.height && w_im < g.width)
          val = im2col_bilinear(ip, g.width, g.height, g.width, h_im, w_im);
        *cp = val;
        cp += g.ld_col;
      }
  }
}

// grad_im += scatter of the columns (modulated_deformable_col2im_gpu_kernel semantics, deform_conv_cuda_kernel.cuh: every
// column entry goes to the integer pixels within distance < 1 of its sampling point with get_gradient_weight).
// The sampling point and the (at most 2 x 2) target pixels with their weights depend on (b, deformable group, tap,
// h_out, w_out) only, not on the channel: one thread owns such a position for a chunk of channels, computes the geometry
// once and walks the channels (coalesced column reads along w_out, 4 hardware fp32 atomics per entry).  One thread per
// column ENTRY with six 64-bit divisions and a 5 x 5 candidate loop took 7.9 ms at (2,256,128,128).
constexpr int C2I_CHUNK = 32;  // channels per thread
__global__ __launch_bounds__(256) void deform_col2im_kernel(long npos, const float* __restrict__ col,
                                                           const float* __restrict__ off, DcnGeom g,
                                                           float* __restrict__ grad_im) {
  const int cpdg = g.channels / g.dg;
  const int taps = g.kh * g.kw;
  const long per_row = (long)g.imgs * g.ho * g.wo;
  const long index = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (dg, tap, b, h_out, w_out), w_out fastest
  if (index >= npos) return;
  const long rem = index % per_row;
  const int tap = (int)((index / per_row) % taps);
  const int dgi = (int)(index / per_row / taps);
  const int j = tap % g.kw, i = tap / g.kw;
  const int w_out = (int)(rem % g.wo);
  const int h_out = (int)((rem / g.wo) % g.ho);
  const int b = (int)(rem / g.wo / g.ho);
  const float* op = off + ((long)b * g.dg + dgi) * 2 * taps * g.ho * g.wo;
  const float offset_h = op[((long)(2 * tap) * g.ho + h_out) * g.wo + w_out];
  const float offset_w = op[((long)(2 * tap + 1) * g.ho + h_out) * g.wo + w_out];
  const float cur_inv_h = (h_out * g.stride_h - g.pad_h) + i * g.dil_h + offset_h;
  const float cur_inv_w = (w_out * g.stride_w - g.pad_w) + j * g.dil_w + offset_w;
  // candidates: the integer rows y with |cur_inv_h - y| < 1 (floor, and floor + 1 unless the coordinate is integral),
  // same for columns -- exactly the pixels the reference's (int)-centred 5 x 5 search accepts
  const float fh = floorf(cur_inv_h), fw = floorf(cur_inv_w);
  int ys[2], xs[2];
  int ny = 0, nx = 0;
  if (fabsf(cur_inv_h) < 1e9f && fabsf(cur_inv_w) < 1e9f) {
    const int y0 = (int)fh, x0 = (int)fw;
    if (y0 >= 0 && y0 < g.height) ys[ny++] = y0;
    if (cur_inv_h != fh && y0 + 1 >= 0 && y0 + 1 < g.height) ys[ny++] = y0 + 1;
    if (x0 >= 0 && x0 < g.width) xs[nx++] = x0;
    if (cur_inv_w != fw && x0 + 1 >= 0 && x0 + 1 < g.width) xs[nx++] = x0 + 1;
  }
  if (ny == 0 || nx == 0) return;
  float wgt[4];
  int pix[4];
  int np = 0;
  for (int a = 0; a < ny; a++)
    for (int c = 0; c < nx; c++) {
      pix[np] = ys[a] * g.width + xs[c];
      wgt[np] = gradient_weight(cur_inv_h, cur_inv_w, ys[a], xs[c], g.height, g.width);
      np++;
    }
  const int c_begin = dgi * cpdg + blockIdx.y * C2I_CHUNK;
  const int c_end = min(c_begin + C2I_CHUNK, (dgi + 1) * cpdg);
  const long plane = (long)g.height * g.width;
  for (int c = c_begin; c < c_end; c++) {
    const float top = col[((long)c * taps + tap) * g.ld_col + rem];
    float* gp = grad_im + ((long)b * g.channels + c) * plane;
    for (int q = 0; q < np; q++) atomicAdd(gp + pix[q], wgt[q] * top);
  }
}

// The same scatter onto an NHWC gradient map (imgs, H, W, C), zero-filled by the caller: a wave owns one sampling position
// (b, deformable group, tap, h_out, w_out) and its 64 lanes are 64 CHANNELS, so each of the <= 4 target pixels receives
// ONE coalesced 256-byte atomic instead of 64 scattered 4-byte ones on 64 different planes (the NCHW form above spends
// 8 ms of the 10.6 ms backward of (2,256,128,128) in 302 M scattered atomics).  The column matrix is (C*taps, positions)
// with positions contiguous, so a workgroup first turns a [64 channels] x [64 positions] tile of one tap through LDS.
// grid = (position tiles, channel tiles of the group, deformable groups x taps).
__global__ __launch_bounds__(256) void deform_col2im_nhwc_kernel(const float* __restrict__ col,
                                                                const float* __restrict__ off, DcnGeom g,
                                                                float* __restrict__ grad_nhwc) {
  __shared__ float tile[64][65];
  const int taps = g.kh * g.kw;
  const int tap = blockIdx.z % taps, dgi = blockIdx.z / taps;
  const int cpdg = g.channels / g.dg;
  const long per_row = (long)g.imgs * g.ho * g.wo;
  const long p0 = (long)blockIdx.x * 64;
  const int c0 = dgi * cpdg + blockIdx.y * 64, c_lim = (dgi + 1) * cpdg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int c = wave + 4 * i;
    const long p = p0 + lane;
    tile[c][lane] = (c0 + c < c_lim && p < per_row) ? col[((long)(c0 + c) * taps + tap) * g.ld_col + p] : 0.f;
  }
  __syncthreads();
  const int i = tap / g.kw, j = tap % g.kw;
  const int ch = c0 + lane;
  for (int q = 0; q < 16; q++) {
    const int pl = __builtin_amdgcn_readfirstlane(wave * 16 + q);
    const long p = p0 + pl;
    if (p >= per_row) break;
    const int w_out = (int)(p % g.wo);
    const int h_out = (int)((p / g.wo) % g.ho);
    const int b = (int)(p / g.wo / g.ho);
    const float* op = off + ((long)b * g.dg + dgi) * 2 * taps * g.ho * g.wo;
    const float offset_h = op[((long)(2 * tap) * g.ho + h_out) * g.wo + w_out];
    const float offset_w = op[((long)(2 * tap + 1) * g.ho + h_out) * g.wo + w_out];
    const float cur_inv_h = (h_out * g.stride_h - g.pad_h) + i * g.dil_h + offset_h;
    const float cur_inv_w = (w_out * g.stride_w - g.pad_w) + j * g.dil_w + offset_w;
    if (!(fabsf(cur_inv_h) < 1e9f && fabsf(cur_inv_w) < 1e9f)) continue;
    const float fh = floorf(cur_inv_h), fw = floorf(cur_inv_w);
    const int y0 = (int)fh, x0 = (int)fw;
    const float top = tile[lane][pl];
    if (ch >= c_lim) continue;
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int y = y0 + a;
      if (y < 0 || y >= g.height || (a == 1 && cur_inv_h == fh)) continue;
#pragma unroll
      for (int c = 0; c < 2; c++) {
        const int x = x0 + c;
        if (x < 0 || x >= g.width || (c == 1 && cur_inv_w == fw)) continue;
        const float wgt = gradient_weight(cur_inv_h, cur_inv_w, y, x, g.height, g.width);
        atomicAdd(grad_nhwc + (((long)b * g.height + y) * g.width + x) * g.channels + ch, wgt * top);
      }
    }
  }
}

// Input gradient AND offset gradient in one pass over the column matrix (the two reference kernels
// deformable_col2im / deformable_col2im_coord read the same columns and evaluate the same sampling geometry).  NHWC on
// both sides: a wave owns a sampling position (b, deformable group, tap, h_out, w_out), its lanes are 64 channels.
//   * d im: <= 4 coalesced 256-byte atomics per position and channel chunk (as deform_col2im_nhwc_kernel);
//   * d offset: val_h / val_w = sum_c col_c * coordinate_weight_c -- the four corner pixels are the same for every
//     channel, so a lane reads its channel of the four NHWC corner vectors (coalesced) and accumulates both directions;
//     the 64 lanes are folded with shuffles once all channel chunks are done.  (The reference-shaped kernel above runs
//     one thread per (offset channel, position) over all channels with four scattered plane reads per channel: 1.19 ms
//     at (2,256,128,128), the largest part of the backward once the scatter was coalesced.)
// grid = (position tiles of 64, 1, deformable groups x taps); grad_nhwc zero-filled by the caller; grad_off fully written.
__global__ __launch_bounds__(256) void deform_bwd_input_fused_kernel(const float* __restrict__ col,
                                                                    const float* __restrict__ im_nhwc,
                                                                    const float* __restrict__ off, DcnGeom g,
                                                                    float* __restrict__ grad_nhwc,
                                                                    float* __restrict__ grad_off) {
  __shared__ float tile[64][65];
  const int taps = g.kh * g.kw;
  const int tap = blockIdx.z % taps, dgi = blockIdx.z / taps;
  const int cpdg = g.channels / g.dg;
  const long per_row = (long)g.imgs * g.ho * g.wo;
  const long p0 = (long)blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = tap / g.kw, j = tap % g.kw;
  float acc_h[16], acc_w[16];
#pragma unroll
  for (int q = 0; q < 16; q++) acc_h[q] = acc_w[q] = 0.f;
  for (int cb = 0; cb < cpdg; cb += 64) {
    const int c0 = dgi * cpdg + cb, c_lim = (dgi + 1) * cpdg;
    __syncthreads();  // previous chunk's tile fully consumed
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int c = wave + 4 * r;
      const long p = p0 + lane;
      tile[c][lane] = (c0 + c < c_lim && p < per_row) ? col[((long)(c0 + c) * taps + tap) * g.ld_col + p] : 0.f;
    }
    __syncthreads();
    const int ch = c0 + lane;
    const bool ch_ok = ch < c_lim;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int pl = __builtin_amdgcn_readfirstlane(wave * 16 + q);
      const long p = p0 + pl;
      if (p >= per_row) continue;
      const int w_out = (int)(p % g.wo);
      const int h_out = (int)((p / g.wo) % g.ho);
      const int b = (int)(p / g.wo / g.ho);
      const float* op = off + ((long)b * g.dg + dgi) * 2 * taps * g.ho * g.wo;
      const float offset_h = op[((long)(2 * tap) * g.ho + h_out) * g.wo + w_out];
      const float offset_w = op[((long)(2 * tap + 1) * g.ho + h_out) * g.wo + w_out];
      const float ah = (h_out * g.stride_h - g.pad_h) + i * g.dil_h + offset_h;
      const float aw = (w_out * g.stride_w - g.pad_w) + j * g.dil_w + offset_w;
      if (!(fabsf(ah) < 1e9f && fabsf(aw) < 1e9f)) continue;
      const float fh = floorf(ah), fw = floorf(aw);
      const int hl = (int)fh, wl = (int)fw;
      const float top = tile[lane][pl];
      if (!ch_ok) continue;
      // the coordinate weight's validity test (get_coordinate_weight: a point outside (-1, H) x (-1, W) contributes 0)
      const bool inside = !(ah <= -1.f || aw <= -1.f || ah >= g.height || aw >= g.width);
      float im4[4];
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const int y = hl + a;
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const int xx = wl + c;
          const bool pix_ok = y >= 0 && y < g.height && xx >= 0 && xx < g.width;
          const long pix = (((long)b * g.height + (pix_ok ? y : 0)) * g.width + (pix_ok ? xx : 0)) * g.channels + ch;
          im4[2 * a + c] = (inside && pix_ok) ? im_nhwc[pix] : 0.f;
          // d im: every integer pixel within distance < 1 of the sampling point (floor + 1 only if non-integral)
          const bool hit = pix_ok && !(a == 1 && ah == fh) && !(c == 1 && aw == fw);
          if (hit) {
            const float wgt = gradient_weight(ah, aw, y, xx, g.height, g.width);
            atomicAdd(grad_nhwc + pix, wgt * top);
          }
        }
      }
      // get_coordinate_weight (deform_conv.cpp:65-112), corners (hl,wl) (hl,wh) (hh,wl) (hh,wh)
      const float cwh = -1.f * (wl + 1 - aw) * im4[0] + -1.f * (aw - wl) * im4[1] + (wl + 1 - aw) * im4[2] + (aw - wl) * im4[3];
      const float cww = -1.f * (hl + 1 - ah) * im4[0] + (hl + 1 - ah) * im4[1] + -1.f * (ah - hl) * im4[2] + (ah - hl) * im4[3];
      acc_h[q] += cwh * top;
      acc_w[q] += cww * top;
    }
  }
#pragma unroll
  for (int q = 0; q < 16; q++) {
    float vh = acc_h[q], vw = acc_w[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vh += __shfl_xor(vh, o, 64);
      vw += __shfl_xor(vw, o, 64);
    }
    const long p = p0 + wave * 16 + q;
    if (lane == 0 && p < per_row) {
      const int w_out = (int)(p % g.wo);
      const int h_out = (int)((p / g.wo) % g.ho);
      const int b = (int)(p / g.wo / g.ho);
      float* gp = grad_off + ((long)b * g.dg + dgi) * 2 * taps * g.ho * g.wo;
      gp[((long)(2 * tap) * g.ho + h_out) * g.wo + w_out] = vh;
      gp[((long)(2 * tap + 1) * g.ho + h_out) * g.wo + w_out] = vw;
    }
  }
}

// one thread per offset element (b, c_off, h, w): gathers over the channels of its deformable group
__global__ __launch_bounds__(256) void deform_col2im_coord_kernel(long n, const float* __restrict__ col,
                                                                 const float* __restrict__ im,
                                                                 const float* __restrict__ off, DcnGeom g,
                                                                 float* __restrict__ grad_off) {
  const int offset_channels = 2 * g.kh * g.kw * g.dg;
  const int cpdg_col = g.channels * g.kh * g.kw / g.dg;  // column rows per deformable group
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < n; index += (long)gridDim.x * blockDim.x) {
    float val = 0;
    const int w = index % g.wo;
    const int h = (index / g.wo) % g.ho;
    const int c = (index / g.wo / g.ho) % offset_channels;
    const int b = (index / g.wo / g.ho) / offset_channels;
    const int dgi = c / (2 * g.kh * g.kw);
    const int col_step = g.kh * g.kw;
    int cnt = 0;
    const float* colp = col + (long)dgi * cpdg_col * g.ld_col;
    const float* imp = im + ((long)b * g.dg + dgi) * (cpdg_col / g.kh / g.kw) * g.height * g.width;
    const float* op = off + ((long)b * g.dg + dgi) * 2 * g.kh * g.kw * g.ho * g.wo;
    const int offset_c = c - dgi * 2 * g.kh * g.kw;
    for (int col_c = (offset_c / 2); col_c < cpdg_col; col_c += col_step) {
      const long col_pos = (long)col_c * g.ld_col + ((long)b * g.ho + h) * g.wo + w;
      const int bp_dir = offset_c % 2;
      const int j = col_c % g.kw;
      const int i = (col_c / g.kw) % g.kh;
      const int w_in = w * g.stride_w - g.pad_w;
      const int h_in = h * g.stride_h - g.pad_h;
      const float offset_h = op[((long)(2 * (i * g.kw + j)) * g.ho + h) * g.wo + w];
      const float offset_w = op[((long)(2 * (i * g.kw + j) + 1) * g.ho + h) * g.wo + w];
      float inv_h = h_in + i * g.dil_h + offset_h;
      float inv_w = w_in + j * g.dil_w + offset_w;
      if (inv_h <= -1 || inv_w <= -1 || inv_h >= g.height || inv_w >= g.width) inv_h = inv_w = -2;
      const float weight = coordinate_weight(inv_h, inv_w, g.height, g.width, imp + (long)cnt * g.height * g.width,
                                             g.width, bp_dir);
      val += weight * colp[col_pos];
      cnt += 1;
    }
    grad_off[index] = val;
  }
}

inline int blocks_for(long n) {
  long b = (n + 255) / 256;
  if (b > 256L * 32) b = 256L * 32;
  return (int)(b < 1 ? 1 : b);
}

inline bool make_geom(DcnGeom& g, int channels, int height, int width, int kh, int kw, int pad_h, int pad_w,
                      int stride_h, int stride_w, int dil_h, int dil_w, int imgs, int dg, long ld_col) {
  if (channels <= 0 || height <= 0 || width <= 0 || kh <= 0 || kw <= 0 || stride_h <= 0 || stride_w <= 0 ||
      dil_h <= 0 || dil_w <= 0 || imgs <= 0 || dg <= 0 || channels % dg)
    return false;
  g.channels = channels; g.height = height; g.width = width; g.kh = kh; g.kw = kw;
  g.pad_h = pad_h; g.pad_w = pad_w; g.stride_h = stride_h; g.stride_w = stride_w; g.dil_h = dil_h; g.dil_w = dil_w;
  g.imgs = imgs; g.dg = dg;
  g.ho = (height + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  g.wo = (width + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  g.ld_col = ld_col;
  return g.ho > 0 && g.wo > 0 && ld_col >= (long)imgs * g.ho * g.wo;
}

}  // namespace

extern "C" {

int sm3_deform_im2col(const float* im, const float* offset, float* col, int channels, int height, int width, int kh,
                      int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int imgs,
                      int deformable_group, long ld_col, sm3_stream_t stream) {
  DcnGeom g;
  if (!im || !offset || !col ||
      !make_geom(g, channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, imgs,
                 deformable_group, ld_col))
    return SM3_ERR_INVALID_ARG;
  const long n = (long)channels * g.ho * g.wo * imgs;
  deform_im2col_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(n, im, offset, g, col);
  return launch_status();
}

int sm3_deform_col2im(const float* col, const float* offset, float* grad_im, int channels, int height, int width,
                      int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int imgs,
                      int deformable_group, long ld_col, sm3_stream_t stream) {
  DcnGeom g;
  if (!col || !offset || !grad_im ||
      !make_geom(g, channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, imgs,
                 deformable_group, ld_col))
    return SM3_ERR_INVALID_ARG;
  const long npos = (long)deformable_group * kh * kw * g.ho * g.wo * imgs;
  const int cpdg = channels / deformable_group;
  dim3 grid((unsigned)((npos + 255) / 256), (unsigned)((cpdg + C2I_CHUNK - 1) / C2I_CHUNK));
  deform_col2im_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(npos, col, offset, g, grad_im);
  return launch_status();
}

int sm3_deform_col2im_nhwc(const float* col, const float* offset, float* grad_im_nhwc, int channels, int height,
                           int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                           int dil_w, int imgs, int deformable_group, long ld_col, sm3_stream_t stream) {
  DcnGeom g;
  if (!col || !offset || !grad_im_nhwc ||
      !make_geom(g, channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, imgs,
                 deformable_group, ld_col))
    return SM3_ERR_INVALID_ARG;
  const long per_row = (long)g.ho * g.wo * imgs;
  const int cpdg = channels / deformable_group;
  const long zt = (long)deformable_group * kh * kw;
  if (zt > 65535 || (cpdg + 63) / 64 > 65535) return SM3_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((per_row + 63) / 64), (unsigned)((cpdg + 63) / 64), (unsigned)zt);
  deform_col2im_nhwc_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(col, offset, g, grad_im_nhwc);
  return launch_status();
}

int sm3_deform_bwd_input_fused(const float* col, const float* im_nhwc, const float* offset, float* grad_im_nhwc,
                               float* grad_offset, int channels, int height, int width, int kh, int kw, int pad_h,
                               int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int imgs,
                               int deformable_group, long ld_col, sm3_stream_t stream) {
  DcnGeom g;
  if (!col || !im_nhwc || !offset || !grad_im_nhwc || !grad_offset ||
      !make_geom(g, channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, imgs,
                 deformable_group, ld_col))
    return SM3_ERR_INVALID_ARG;
  const long per_row = (long)g.ho * g.wo * imgs;
  const long zt = (long)deformable_group * kh * kw;
  if (zt > 65535) return SM3_ERR_UNSUPPORTED;
  dim3 grid((unsigned)((per_row + 63) / 64), 1, (unsigned)zt);
  deform_bwd_input_fused_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(col, im_nhwc, offset, g, grad_im_nhwc,
                                                                       grad_offset);
  return launch_status();
}

int sm3_deform_col2im_coord(const float* col, const float* im, const float* offset, float* grad_offset, int channels,
                            int height, int width, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                            int dil_h, int dil_w, int imgs, int deformable_group, long ld_col, sm3_stream_t stream) {
  DcnGeom g;
  if (!col || !im || !offset || !grad_offset ||
      !make_geom(g, channels, height, width, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, imgs,
                 deformable_group, ld_col))
    return SM3_ERR_INVALID_ARG;
  const long n = (long)g.ho * g.wo * 2 * kh * kw * deformable_group * imgs;
  deform_col2im_coord_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(n, col, im, offset, g, grad_offset);
  return launch_status();
}

}  // extern "C"
