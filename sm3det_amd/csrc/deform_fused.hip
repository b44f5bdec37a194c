// deform_fused.hip -- DeformConv2d forward WITHOUT the column matrix (round 4).
//
// The reference materialises `columns` (Cin*kh*kw, step*Ho*Wo) with deformable_im2col and multiplies it with the weights
// (mmcv/mmcv/ops/csrc/pytorch/deform_conv.cpp:140-258, common/cuda/deform_conv_cuda_kernel.cuh:190-241); the first
// MI355X version did the same on the MFMA GEMM family: 1.22 GB of HBM traffic for 69.5 MB of algorithmic bytes at
// (2,256,128,128), im2col 302 us + GEMM 331 us.  Here the bilinear sampling happens INSIDE the GEMM's A-operand producer:
//
//   out[pos, cout] = sum_{tap, cin} sample(x, pos, tap)[cin] * W[cout, cin, tap]        M = B*Ho*Wo, N = Cout, K = 9*Cin
//
// * x is read as NHWC (one transpose pass by the host wrapper): the four bilinear corners of (position, tap) are the same
//   for every input channel, so a thread fetches them as 16-byte channel vectors (4 channels of a 16-channel k-tile),
//   blends them with the reference's own expression (w1*v1 + w2*v2 + w3*v3 + w4*v4, -ffp-contract=off: the column values
//   are bit-identical to deformable_im2col's) and writes them k-major into LDS -- the column tile never leaves the CU;
// * the 2 x 9 offsets of a position are loaded once into registers (deformable_group = 1); the corner addresses and
//   weights are recomputed only when the tap changes (every Cin/16 k-steps);
// * weights come as W^T (taps*Cin, Cout) (a 2.4 MB re-layout by the wrapper), staged k-major as well; 64 x 256 tile
//   (positions x output channels: with 128 x 128 tiles two workgroups sampled every position, 516 us; the producer is the
//   bottleneck, not the matrix pipe), 4 waves side by side along the channels, each 2 x 2 v_mfma_f32_32x32x2_f32,
//   double-buffered LDS, one barrier per 16-deep k-step;
// * the output goes straight to the reference layout (B, Cout, Ho, Wo): a lane owns one position, so the 32 lanes of a
//   half-wave write 128 contiguous bytes per output channel.
// Restrictions (else the wrapper keeps the im2col + GEMM path): groups = 1, deformable_groups = 1, Cin % 16 == 0.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 64, BN = 256, BK = 16;  // 64 positions x all 256 output channels: a position is sampled ONCE per tap
constexpr int LDA = BM + 2, LDB = BN + 4;
constexpr int MAX_TAPS = 9;

struct FusedGeom {
  int B, Cin, H, W, Cout, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, Ho, Wo;
};

__global__ __launch_bounds__(256, 3) void deform_conv_fwd_fused_kernel(const float* __restrict__ x,    // (B,H,W,Cin)
                                                                      const float* __restrict__ off,  // (B,2*taps,Ho,Wo)
                                                                      const float* __restrict__ wT,   // (taps*Cin,Cout)
                                                                      float* __restrict__ out,        // (B,Cout,Ho,Wo)
                                                                      FusedGeom g) {
  __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm0 = 0, wn0 = wave * 64;
  const int taps = g.kh * g.kw;
  const long HoWo = (long)g.Ho * g.Wo;
  const long M = (long)g.B * HoWo;
  const int ntn = (g.Cout + BN - 1) / BN;
  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so workgroup id i
  // takes tile (i % 8) * (tiles / 8) + i / 8: an XCD samples ONE contiguous eighth of the positions (its input window +
  // the 2.4 MB of weights instead of the whole map through every L2).  Counted HBM traffic at (2,256,128,128): 919 ->
  // 281 MB (69.5 MB algorithmic); the time did not move (446 -> 446 us: the kernel is bound by its sampling producer and
  // the MFMA issue, not by bytes).  Also measured: two k-tiles of lookahead in registers -- 256 VGPRs, spills, 4 % slower.
  const int nwg = gridDim.x, per_xcd = (nwg + 7) / 8;
  int bid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (per_xcd * 8 != nwg) bid = blockIdx.x;  // (tile counts that do not split evenly keep the plain order)
  const int tile_n = bid % ntn, tile_m = bid / ntn;
  const long m0 = (long)tile_m * BM;
  if (m0 >= M) return;
  const int n0 = tile_n * BN;

  // ---- A producer: thread -> (position p, 4-channel quarter of the k-tile)
  const int p = tid >> 2, quarter = tid & 3;
  const long m = m0 + p;
  const bool pvalid = m < M;
  const long mc = pvalid ? m : M - 1;
  const int b = (int)(mc / HoWo);
  const int rem = (int)(mc - (long)b * HoWo);
  const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
  float oh[MAX_TAPS], ow[MAX_TAPS];
#pragma unroll
  for (int t = 0; t < MAX_TAPS; t++) {
    oh[t] = ow[t] = 0.f;
    if (t < taps) {
      const float* op = off + ((long)b * 2 * taps + 2 * t) * HoWo + rem;
      oh[t] = op[0];
      ow[t] = op[HoWo];
    }
  }
  const float* xb = x + (long)b * g.H * g.W * g.Cin + 4 * quarter;
  // geometry of the current tap (deformable_im2col + im2col_bilinear, deform_conv_cuda_kernel.cuh:190-241 / :60-101)
  long c1 = 0, c2 = 0, c3 = 0, c4 = 0;  // element offsets of the four corners' pixel rows
  float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
  bool k1 = false, k2 = false, k3 = false, k4 = false;
  auto tap_geometry = [&](int t) {
    float oht = 0.f, owt = 0.f;
#pragma unroll
    for (int q = 0; q < MAX_TAPS; q++)
      if (q == t) {
        oht = oh[q];
        owt = ow[q];
      }
    const int i = t / g.kw, j = t - i * g.kw;
    const float h = (float)(ho * g.stride_h - g.pad_h + i * g.dil_h) + oht;
    const float w = (float)(wo * g.stride_w - g.pad_w + j * g.dil_w) + owt;
    k1 = k2 = k3 = k4 = false;
    w1 = w2 = w3 = w4 = 0.f;
    if (pvalid && h > -1 && w > -1 && h < g.H && w < g.W) {
      const int h_low = (int)floorf(h), w_low = (int)floorf(w);
      const int h_high = h_low + 1, w_high = w_low + 1;
      const float lh_ = h - h_low, lw_ = w - w_low;
      const float hh = 1 - lh_, hw = 1 - lw_;
      k1 = h_low >= 0 && w_low >= 0;
      k2 = h_low >= 0 && w_high <= g.W - 1;
      k3 = h_high <= g.H - 1 && w_low >= 0;
      k4 = h_high <= g.H - 1 && w_high <= g.W - 1;
      w1 = hh * hw;
      w2 = hh * lw_;
      w3 = lh_ * hw;
      w4 = lh_ * lw_;
      const int hl = max(h_low, 0), wl = max(w_low, 0), hhi = min(h_high, g.H - 1), whi = min(w_high, g.W - 1);
      c1 = ((long)hl * g.W + wl) * g.Cin;
      c2 = ((long)hl * g.W + whi) * g.Cin;
      c3 = ((long)hhi * g.W + wl) * g.Cin;
      c4 = ((long)hhi * g.W + whi) * g.Cin;
    }
  };
  const int kt_per_tap = g.Cin / BK;
  const int nk = taps * kt_per_tap;
  f32x4 cv[4];  // corner channel vectors of the k-tile in flight
  f32x4 bv[4];  // weight rows of the k-tile in flight
  const int b_row = tid >> 6, b_cq = tid & 63;
  const int b_col = min(n0 + 4 * b_cq, g.Cout - 4);
  auto issue = [&](int kt) {
    const int t = kt / kt_per_tap, c0 = (kt - t * kt_per_tap) * BK;
    if (kt % kt_per_tap == 0) tap_geometry(t);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const float* q = xb + c0;
    {
      // clamped address + select (no branch around the loads: the whole batch stays in flight)
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(q + c1), a2 = *reinterpret_cast<const f32x4*>(q + c2);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(q + c3), a4 = *reinterpret_cast<const f32x4*>(q + c4);
      cv[0] = k1 ? a1 : z;
      cv[1] = k2 ? a2 : z;
      cv[2] = k3 ? a3 : z;
      cv[3] = k4 ? a4 : z;
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      bv[u] = *reinterpret_cast<const f32x4*>(wT + ((long)kt * BK + b_row + 4 * u) * g.Cout + b_col);
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; e++) {
      // the reference's expression, term order kept (contraction off for this file): identical column values
      const float v = w1 * cv[0][e] + w2 * cv[1][e] + w3 * cv[2][e] + w4 * cv[3][e];
      As[buf][4 * quarter + e][p] = v;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) *reinterpret_cast<f32x4*>(&Bs[buf][b_row + 4 * u][4 * b_cq]) = bv[u];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  issue(0);
  stage(0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt++) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue(kt + 1);
#pragma unroll
    for (int kk = 0; kk < BK / 2; kk++) {
      const int krow = 2 * kk + lh;
      float a[2], bb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) a[i] = As[buf][krow][wm0 + 32 * i + l31];
#pragma unroll
      for (int j = 0; j < 2; j++) bb[j] = Bs[buf][krow][wn0 + 32 * j + l31];
      // operands swapped on purpose (as in gemm_f32_kernel.h): D = (B fragment) x (A fragment) = the transposed 32x32
      // tile, so a lane holds 4 consecutive output CHANNELS of ONE position
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bb[j], a[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) stage(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][4q + e]: position row = wm0 + 32 i + l31 ; channel = wn0 + 32 j + 8 q + 4 lh + e
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const long mr = m0 + wm0 + 32 * i + l31;
    if (mr >= M) continue;
    const int ob = (int)(mr / HoWo);
    const long opos = mr - (long)ob * HoWo;
    float* orow = out + (long)ob * g.Cout * HoWo + opos;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int col = n0 + wn0 + 32 * j + 8 * q + 4 * lh + e;
          if (col < g.Cout) orow[(long)col * HoWo] = acc[i][j][4 * q + e];
        }
  }
}

}  // namespace

extern "C" {

int sm3_deform_conv_fwd_fused_supported(int channels, int out_channels, int kh, int kw, int group, int deformable_group) {
  return group == 1 && deformable_group == 1 && channels > 0 && (channels % BK) == 0 && out_channels >= 4 &&
         (out_channels % 4) == 0 && kh * kw <= MAX_TAPS && kh > 0 && kw > 0;
}

int sm3_deform_conv_fwd_fused(const float* x_nhwc, const float* offset, const float* w_t, float* out, int batch, int channels,
                              int height, int width, int out_channels, int kh, int kw, int pad_h, int pad_w, int stride_h,
                              int stride_w, int dil_h, int dil_w, sm3_stream_t stream) {
  if (!x_nhwc || !offset || !w_t || !out || batch <= 0 || height <= 0 || width <= 0) return SM3_ERR_INVALID_ARG;
  if (!sm3_deform_conv_fwd_fused_supported(channels, out_channels, kh, kw, 1, 1)) return SM3_ERR_UNSUPPORTED;
  if (stride_h <= 0 || stride_w <= 0 || dil_h <= 0 || dil_w <= 0 || pad_h < 0 || pad_w < 0) return SM3_ERR_INVALID_ARG;
  FusedGeom g;
  g.B = batch; g.Cin = channels; g.H = height; g.W = width; g.Cout = out_channels; g.kh = kh; g.kw = kw;
  g.pad_h = pad_h; g.pad_w = pad_w; g.stride_h = stride_h; g.stride_w = stride_w; g.dil_h = dil_h; g.dil_w = dil_w;
  g.Ho = (height + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  g.Wo = (width + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  if (g.Ho < 1 || g.Wo < 1) return SM3_ERR_INVALID_ARG;
  const long M = (long)batch * g.Ho * g.Wo;
  const long ntm = (M + BM - 1) / BM, ntn = (out_channels + BN - 1) / BN;
  if (ntm * ntn > 0x7fffffffl) return SM3_ERR_UNSUPPORTED;
  deform_conv_fwd_fused_kernel<<<(unsigned)(ntm * ntn), 256, 0, (hipStream_t)stream>>>(x_nhwc, offset, w_t, out, g);
  return launch_status();
}

}  // extern "C"
