// dwconv_lds.hip -- depthwise 7x7 (padding 3) on NHWC tokens, LDS-tiled variant for gfx950.
//
// ConvNeXtBlock.depthwise_conv (mmrotate/models/backbones/convnext_moe.py:311-312, :347) forward, its input gradient
// (same kernel, reversed taps, residual gradient fused as `addend`) and its weight/bias gradient.
//
// Tile = 16 x 16 output pixels x 32 channels per 256-thread workgroup.  The 22 x 22 x 32 input patch (halo 3) is brought
// into LDS once with coalesced 16-byte loads (pixel stride padded to 40 floats so that the 16-lane groups of a
// ds_read_b128 hit 16 distinct 16-byte slots), the 49 x 32 taps sit next to it; every lane then owns one channel quad
// and a 2 x 4 pixel strip: 80 patch reads + 49 tap reads feed 392 float4 FMAs -> VALU-bound instead of the
// address-arithmetic/latency-bound direct version in backbone.hip (69 us at 2x256x256x96; HBM time 16 us).
// Used when C % 32 == 0 and H, W are multiples of 16 (every stage of ConvNeXt-T/B at 1024^2); otherwise the generic
// kernels in backbone.hip run.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// output element `o` of a tensor stored as fp32, or (OUT16) as fp16 (round to nearest even): the AMP data path keeps the
// convolution output in half, as autocast makes of ConvNeXtBlock.depthwise_conv (convnext_moe.py:347)
template <int OUT16>
__device__ __forceinline__ void st4o(float* y, long o, f32x4 v) {
  if (OUT16) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    *reinterpret_cast<f16x4*>(reinterpret_cast<_Float16*>(y) + o) = f16x4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
  } else {
    st4(y + o, v);
  }
}

#ifndef SM3_DW_MULTITILE
#define SM3_DW_MULTITILE 1  // large maps: workgroups walk several tiles with the next patch in flight (0: A/B build --variant dw_onetile)
#endif
#ifndef SM3_DW_WGRAD_ONE_ROUND
#define SM3_DW_WGRAD_ONE_ROUND 1
#endif
constexpr int TH = 16, TW = 16, CB = 32;          // tile
constexpr int PH = TH + 6, PW = TW + 6;           // patch with halo
constexpr int PS = CB + 8;                        // padded pixel stride (floats) in LDS
constexpr int LDS_PATCH = PH * PW * PS;           // floats (77.4 KB -> two workgroups per CU)
constexpr int TH_B = 8;                           // weight-gradient kernel: 8-row tiles (65.7 KB of LDS)
constexpr int LDS_PATCH_B = (TH_B + 6) * PW * PS;

// ROWS x 22 pixels x 8 quads; out-of-image pixels are zero (padding=3).  All of a thread's 16-byte loads are ISSUED
// before the first LDS store (one wait for the batch instead of one L2/HBM round trip per loop iteration: the
// workgroup's prologue was the longest phase of the kernel, ~15 dependent round trips with 2 workgroups per CU).
template <int ROWS>
__device__ __forceinline__ void load_patch(const float* __restrict__ x, int b, int H, int W, int C, int y0, int x0,
                                           int c0, float* patch) {
  constexpr int TOTAL = ROWS * PW * (CB / 4);
  constexpr int NIT = (TOTAL + 255) / 256;
  f32x4 v[NIT];
#pragma unroll
  for (int i = 0; i < NIT; i++) {
    const int idx = threadIdx.x + 256 * i;
    const int q = idx & 7, px = idx >> 3;
    const int py = px / PW, pxx = px - py * PW;
    const int iy = y0 - 3 + py, ix = x0 - 3 + pxx;
    const bool ok = idx < TOTAL && iy >= 0 && iy < H && ix >= 0 && ix < W;
    // clamped address + select: no branch around the load, so the compiler keeps the whole batch in flight
    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
    const f32x4 t = ld4(x + (((long)b * H + cy) * W + cx) * C + c0 + 4 * q);
    v[i] = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < NIT; i++) {
    const int idx = threadIdx.x + 256 * i;
    if (idx < TOTAL) st4(patch + (idx >> 3) * PS + 4 * (idx & 7), v[i]);
  }
}

// y = conv(x) + bias (+ addend); grid = (W/16 * H/16, C/32, B)
template <int OUT16>
__global__ __launch_bounds__(256) void dwconv7_lds_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w49,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ addend, float* __restrict__ y,
                                                             int H, int W, int C, int flip) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* patch = sm;
  const int tiles_x = W / TW;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int c0 = blockIdx.y * CB, b = blockIdx.z;
  const int y0 = ty * TH, x0 = tx * TW;
  load_patch<PH>(x, b, H, W, C, y0, x0, c0, patch);
  __syncthreads();
  // flip: tap (ky,kx) reads w49[48 - (ky*7+kx)] (correlation with the reversed kernel = input gradient)
  const int tb = flip ? 48 : 0, ts = flip ? -1 : 1;
  const float* taps = w49 + c0;  // 49 x 16 B per lane straight from L1/L2 (every workgroup of a chunk reads the same 6 KB)
  const int cq = threadIdx.x & 7, xg = (threadIdx.x >> 3) & 3, rp = threadIdx.x >> 5;  // 8 quads x 4 x-groups x 8 row pairs
  const int ry = 2 * rp, rx = 4 * xg;
  f32x4 acc[2][4];
  const f32x4 bv = bias ? ld4(bias + c0 + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) acc[r][c] = bv;
  // Patch row ry+ir feeds output row 0 through tap row ky=ir (ir < 7) and output row 1 through ky=ir-1 (ir > 0).  The
  // first and last patch rows are peeled (they feed one output row only), the six middle rows run as a rolled loop (a
  // full unroll hoists all 98 tap loads and spills) with the 7 taps of the NEXT tap row fetched from L1/L2 while the
  // FMAs of this one run.
  f32x4 wprev[7], wcur[7], wnext[7];
#pragma unroll
  for (int kx = 0; kx < 7; kx++) wcur[kx] = ld4(taps + (long)(tb + ts * kx) * C + 4 * cq);
  auto patch_row = [&](int ir, f32x4 (&in)[10]) {
    const float* prow = patch + ((ry + ir) * PW + rx) * PS + 4 * cq;
#pragma unroll
    for (int c = 0; c < 10; c++) in[c] = ld4(prow + c * PS);
  };
  {  // ir = 0: output row 0, tap row 0
#pragma unroll
    for (int kx = 0; kx < 7; kx++) wnext[kx] = ld4(taps + (long)(tb + ts * (7 + kx)) * C + 4 * cq);
    f32x4 in[10];
    patch_row(0, in);
#pragma unroll
    for (int kx = 0; kx < 7; kx++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[0][c] += in[c + kx] * wcur[kx];
#pragma unroll
    for (int kx = 0; kx < 7; kx++) {
      wprev[kx] = wcur[kx];
      wcur[kx] = wnext[kx];
    }
  }
#pragma unroll 1
  for (int ir = 1; ir < 7; ir++) {
    const int kn = ir + 1 < 7 ? ir + 1 : 6;  // clamped: branch-free prefetch
#pragma unroll
    for (int kx = 0; kx < 7; kx++) wnext[kx] = ld4(taps + (long)(tb + ts * (kn * 7 + kx)) * C + 4 * cq);
    f32x4 in[10];
    patch_row(ir, in);
#pragma unroll
    for (int kx = 0; kx < 7; kx++) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        acc[0][c] += in[c + kx] * wcur[kx];
        acc[1][c] += in[c + kx] * wprev[kx];
      }
    }
#pragma unroll
    for (int kx = 0; kx < 7; kx++) {
      wprev[kx] = wcur[kx];
      wcur[kx] = wnext[kx];
    }
  }
  {  // ir = 7: output row 1, tap row 6 (= wprev after the last rotation)
    f32x4 in[10];
    patch_row(7, in);
#pragma unroll
    for (int kx = 0; kx < 7; kx++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[1][c] += in[c + kx] * wprev[kx];
  }
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const long o = (((long)b * H + y0 + ry + r) * W + x0 + rx + c) * C + c0 + 4 * cq;
      f32x4 v = acc[r][c];
      if (addend) v += ld4(addend + o);
      st4o<OUT16>(y, o, v);
    }
}

// ---- several tiles per workgroup (the large maps: more tiles than the chip has workgroup slots) -------------------------
// The one-tile kernel above spends its life in three phases that do not overlap inside a workgroup (patch load ~3 us,
// arithmetic ~1.5 us, stores), and two workgroups per CU are all the LDS allows.  Here a workgroup walks `nt` consecutive
// tiles and the 16-byte loads of tile t + 1's patch are IN FLIGHT (registers) while tile t is computed out of LDS.  The
// taps cannot come from global memory then (vmcnt retires in order: waiting for a tap would wait for the prefetch):
// they live in the 8 pad floats of the first 196 patch pixels (tap ti, quad q -> pixel 4 ti + q / 2, floats 32 + 4 (q & 1):
// eight distinct 16-byte slots per wave), so the LDS footprint -- and two workgroups per CU -- stay as they are.
template <int ROWS>
constexpr int nit_patch() { return (ROWS * PW * (CB / 4) + 255) / 256; }
constexpr int NIT_P = nit_patch<PH>();

// One buffer resource per image (b): byte offsets fit 32 bits and an out-of-range offset reads 0 -- the zero padding
// costs no clamp, no select and no mask kept while the loads are in flight.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const float* img, long floats) {
  const unsigned long long u = reinterpret_cast<unsigned long long>(img);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  const int nrec = __builtin_amdgcn_readfirstlane((int)(floats * 4));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0, nrec, 0x00020000);
}

template <int ROWS>
__device__ __forceinline__ void patch_issue(const float* __restrict__ x, int b, int H, int W, int C, int y0, int x0,
                                            int c0, f32x4 (&v)[nit_patch<ROWS>()]) {
  constexpr int TOTAL = ROWS * PW * (CB / 4);
  const __amdgpu_buffer_rsrc_t rs = image_rsrc(x + (long)b * H * W * C, (long)H * W * C);
  // an opaque zero: the per-load pixel decomposition is recomputed for every tile (a dozen integer instructions per load)
  // instead of being hoisted out of the tile loop into ~50 registers held across the arithmetic
  int z;
  asm volatile("v_mov_b32 %0, 0" : "=v"(z));
  const int tid = threadIdx.x + z;
  const int q4 = (tid & 7) * 4 + c0;
#pragma unroll
  for (int i = 0; i < nit_patch<ROWS>(); i++) {
    const int idx = tid + 256 * i;
    const int px = idx >> 3;
    const int py = (px * 745) >> 14;  // px / 22 for px < 512
    const int pxx = px - py * PW;
    const int iy = y0 - 3 + py, ix = x0 - 3 + pxx;
    const bool ok = (idx < TOTAL) & ((unsigned)iy < (unsigned)H) & ((unsigned)ix < (unsigned)W);
    const unsigned off = (unsigned)(((iy * W + ix) * C + q4) * 4);
    v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off : 0x7fff0000u), 0, 0));
  }
}

template <int ROWS>
__device__ __forceinline__ void patch_commit(const f32x4 (&v)[nit_patch<ROWS>()], float* patch) {
  constexpr int TOTAL = ROWS * PW * (CB / 4);
#pragma unroll
  for (int i = 0; i < nit_patch<ROWS>(); i++) {
    const int idx = threadIdx.x + 256 * i;
    if (idx < TOTAL) st4(patch + (idx >> 3) * PS + 4 * (idx & 7), v[i]);
  }
}

// tiles are numbered ((b * C/32 + chunk) * tiles_y + ty) * tiles_x + tx; grid = ceil(total / nt)
template <int OUT16>
__global__ __launch_bounds__(256, 2) void dwconv7_lds_fwd_mt_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ w49,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ addend,
                                                                float* __restrict__ y, int H, int W, int C, int flip,
                                                                int nt, int total) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* patch = sm;
  const int tiles_x = W / TW, tiles_img = tiles_x * (H / TH), nchunks = C / CB;
  const int t_begin = blockIdx.x * nt, t_end = min(t_begin + nt, total);
  if (t_begin >= t_end) return;
  const int cq = threadIdx.x & 7, xg = (threadIdx.x >> 3) & 3, rp = threadIdx.x >> 5;
  const int ry = 2 * rp, rx = 4 * xg;
  const int tb = flip ? 48 : 0, ts = flip ? -1 : 1;
  const float* tapbase = patch + (cq >> 1) * PS + CB + 4 * (cq & 1);
  auto decode = [&](int t, int& b, int& c0, int& y0, int& x0) {
    const int bc = t / tiles_img, tile = t - bc * tiles_img;
    b = bc / nchunks;
    c0 = (bc - b * nchunks) * CB;
    const int ty = tile / tiles_x;
    y0 = ty * TH;
    x0 = (tile - ty * tiles_x) * TW;
  };
  int b, c0, y0, x0;
  decode(t_begin, b, c0, y0, x0);
  f32x4 v[NIT_P];
  patch_issue<PH>(x, b, H, W, C, y0, x0, c0, v);
  int cur_c0 = -1;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  for (int t = t_begin; t < t_end; t++) {
    if (c0 != cur_c0) {  // (uniform) the taps of this channel chunk -> the pad floats; nobody reads LDS here (barrier below / at the loop end)
      f32x4 tv[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int idx = min((int)threadIdx.x + 256 * i, 49 * 8 - 1);
        tv[i] = ld4(w49 + (long)(idx >> 3) * C + c0 + 4 * (idx & 7));
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < 49 * 8) st4(patch + ((idx >> 3) * 4 + ((idx & 7) >> 1)) * PS + CB + 4 * (idx & 1), tv[i]);
      }
      bv = bias ? ld4(bias + c0 + 4 * cq) : f32x4{0.f, 0.f, 0.f, 0.f};
      cur_c0 = c0;
    }
    patch_commit<PH>(v, patch);
    __syncthreads();
    int nb = b, nc0 = c0, ny0 = y0, nx0 = x0;
    if (t + 1 < t_end) {
      decode(t + 1, nb, nc0, ny0, nx0);
      patch_issue<PH>(x, nb, H, W, C, ny0, nx0, nc0, v);
    }
    f32x4 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) acc[r][c] = bv;
    auto patch_row = [&](int ir, f32x4 (&in)[10]) {
      const float* prow = patch + ((ry + ir) * PW + rx) * PS + 4 * cq;
#pragma unroll
      for (int c = 0; c < 10; c++) in[c] = ld4(prow + c * PS);
    };
    auto tap_row = [&](int ky, f32x4 (&w)[7]) {
#pragma unroll
      for (int kx = 0; kx < 7; kx++) w[kx] = ld4(tapbase + (tb + ts * (ky * 7 + kx)) * (4 * PS));
    };
    f32x4 wprev[7], wcur[7];
    tap_row(0, wcur);
    {  // patch row 0: output row 0, tap row 0
      f32x4 in[10];
      patch_row(0, in);
#pragma unroll
      for (int kx = 0; kx < 7; kx++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[0][c] += in[c + kx] * wcur[kx];
    }
#pragma unroll 1
    for (int ir = 1; ir < 7; ir++) {  // patch row ir: output row 0 through tap row ir, output row 1 through tap row ir - 1
#pragma unroll
      for (int kx = 0; kx < 7; kx++) wprev[kx] = wcur[kx];
      tap_row(ir, wcur);
      f32x4 in[10];
      patch_row(ir, in);
#pragma unroll
      for (int kx = 0; kx < 7; kx++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
          acc[0][c] += in[c + kx] * wcur[kx];
          acc[1][c] += in[c + kx] * wprev[kx];
        }
      }
    }
    {  // patch row 7: output row 1, tap row 6
      f32x4 in[10];
      patch_row(7, in);
#pragma unroll
      for (int kx = 0; kx < 7; kx++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[1][c] += in[c + kx] * wcur[kx];
    }
    const long o0 = (((long)b * H + y0 + ry) * W + x0 + rx) * C + c0 + 4 * cq;
    if (addend) {  // (after the prefetch in vmcnt order: it has landed by now); one batch of eight loads
      f32x4 ad[2][4];
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) ad[r][c] = ld4(addend + o0 + ((long)r * W + c) * C);
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[r][c] += ad[r][c];
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) st4o<OUT16>(y, o0 + ((long)r * W + c) * C, acc[r][c]);
    __syncthreads();  // the patch (and the taps) are free
    b = nb; c0 = nc0; y0 = ny0; x0 = nx0;
  }
}

// dw49[ky*7+kx][c] += sum_p du[p] * x[p + (ky-3, kx-3)] ; dbias[c] += sum_p du[p]   (outputs pre-zeroed)
// grid = (spatial workers, C/32): each workgroup walks tiles with stride gridDim.x keeping its partial sums in
// registers.  Thread = (channel quad, tap row ky, quarter of the tile rows): 8 x 7 x 4 = 224 active threads.
__global__ __launch_bounds__(256) void dwconv7_lds_bwd_weight_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ du,
                                                                    float* __restrict__ dw49,
                                                                    float* __restrict__ dbias, int B, int H, int W,
                                                                    int C) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* patch = sm;                 // x with halo, [TH_B+6][22][PS]
  float* dut = sm + LDS_PATCH_B;     // du tile [TH_B][16][CB]
  const int c0 = blockIdx.y * CB;
  const int tiles_x = W / TW, tiles_y = H / TH_B;
  const int ntiles = B * tiles_y * tiles_x;
  const int cq = threadIdx.x & 7;
  const int ky = (threadIdx.x >> 3) % 7;
  const int qr = (threadIdx.x >> 3) / 7;  // 0..3 active, 4 = idle lanes (threads 224..255)
  const bool active = qr < 4;
  f32x4 aw[7], ab = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 7; k++) aw[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The x patch and the du tile of the NEXT tile of this workgroup are in flight (registers) while this one is multiplied
  // out of LDS (the phases of a tile -- loads ~3 us, arithmetic, barrier -- did not overlap inside a workgroup).
  constexpr int NG = TH_B * TW * (CB / 4) / 256;  // 4 du loads per thread
  f32x4 pv[nit_patch<TH_B + 6>()], gv[NG];
  auto issue = [&](int t) {
    const int tx = t % tiles_x;
    const int t2 = t / tiles_x;
    const int ty = t2 % tiles_y, b = t2 / tiles_y;
    const int y0 = ty * TH_B, x0 = tx * TW;
    patch_issue<TH_B + 6>(x, b, H, W, C, y0, x0, c0, pv);
#pragma unroll
    for (int i = 0; i < NG; i++) {
      const int idx = threadIdx.x + 256 * i;
      const int q = idx & 7, px = idx >> 3;
      gv[i] = ld4(du + (((long)b * H + y0 + (px >> 4)) * W + x0 + (px & 15)) * C + c0 + 4 * q);
    }
  };
  if ((int)blockIdx.x < ntiles) issue(blockIdx.x);
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    patch_commit<TH_B + 6>(pv, patch);  // (the barrier at the end of the previous iteration freed the LDS)
#pragma unroll
    for (int i = 0; i < NG; i++) {
      const int idx = threadIdx.x + 256 * i;
      st4(dut + (idx >> 3) * CB + 4 * (idx & 7), gv[i]);
    }
    __syncthreads();
    if (t + (int)gridDim.x < ntiles) issue(t + gridDim.x);
    if (active) {
#pragma unroll 1
      for (int r = 0; r < TH_B / 4; r++) {
        const int oy = (TH_B / 4) * qr + r;  // output row of the tile; input row oy + ky
        const float* xr = patch + ((oy + ky) * PW) * PS + 4 * cq;
        const float* gr = dut + (oy * TW) * CB + 4 * cq;
        f32x4 in[PW];
#pragma unroll
        for (int c = 0; c < PW; c++) in[c] = ld4(xr + c * PS);
#pragma unroll
        for (int c = 0; c < TW; c++) {
          const f32x4 g = ld4(gr + c * CB);
          if (ky == 0) ab += g;
#pragma unroll
          for (int kx = 0; kx < 7; kx++) aw[kx] += g * in[c + kx];
        }
      }
    }
    __syncthreads();  // tile fully consumed
  }
  // fold the 4 row-quarters through LDS, then one atomic per (tap, channel) per workgroup
  float* red = sm;  // [4][7 ky][8 slots (7 taps + bias)][CB]
  if (active) {
#pragma unroll
    for (int k = 0; k < 7; k++) st4(red + (((qr * 7 + ky) * 8 + k) * CB) + 4 * cq, aw[k]);
    st4(red + (((qr * 7 + ky) * 8 + 7) * CB) + 4 * cq, ab);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 7 * 8 * CB; i += 256) {
    const int c = i % CB, k = (i / CB) & 7, kyy = i / (8 * CB);
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; q++) s += red[((q * 7 + kyy) * 8 + k) * CB + c];
    if (k < 7) atomicAdd(dw49 + (long)(kyy * 7 + k) * C + c0 + c, s);
    else if (kyy == 0) atomicAdd(dbias + c0 + c, s);
  }
}

}  // namespace

// C++ entry points used by backbone.hip's C ABI functions (same library)
// (one image must stay below 2 GiB: the prefetching kernels address it through a buffer resource with 32-bit byte offsets)
bool sm3_dwconv7_lds_supported(int H, int W, int C) {
  return (C % CB) == 0 && (H % TH) == 0 && (W % TW) == 0 && (long)H * W * C * 4 < 0x7fff0000L;
}

// `flags`: bit 0 = reversed taps (input gradient), bit 5 (32) = y is stored as fp16
void sm3_dwconv7_lds_fwd(const float* x, const float* w49, const float* bias, const float* addend, float* y, int B,
                         int H, int W, int C, int flags, hipStream_t st) {
  const int flip = flags & 1;
  const bool out16 = (flags & 32) != 0;
  const size_t lds = (size_t)LDS_PATCH * sizeof(float);
#if SM3_DW_MULTITILE
  // more tiles than workgroup slots (2 per CU): one round of workgroups walking nt tiles each, loads ahead of the arithmetic
  const int total = (W / TW) * (H / TH) * (C / CB) * B;
  const int nt = (total + 511) / 512;
  if (nt >= 3 || (nt == 2 && addend)) {  // (two tiles without an addend: 22.0 vs 20.9 us at 2x128x128x192, same box)
    const int grid = (total + nt - 1) / nt;
    if (out16) dwconv7_lds_fwd_mt_kernel<1><<<grid, 256, lds, st>>>(x, w49, bias, addend, y, H, W, C, flip, nt, total);
    else dwconv7_lds_fwd_mt_kernel<0><<<grid, 256, lds, st>>>(x, w49, bias, addend, y, H, W, C, flip, nt, total);
    return;
  }
#endif
  dim3 grid((W / TW) * (H / TH), C / CB, B);
  if (out16) dwconv7_lds_fwd_kernel<1><<<grid, 256, lds, st>>>(x, w49, bias, addend, y, H, W, C, flip);
  else dwconv7_lds_fwd_kernel<0><<<grid, 256, lds, st>>>(x, w49, bias, addend, y, H, W, C, flip);
}

void sm3_dwconv7_lds_bwd_weight(const float* x, const float* du, float* dw49, float* dbias, int B, int H, int W, int C,
                                hipStream_t st) {
  const int ntiles = B * (H / TH_B) * (W / TW);
#if SM3_DW_WGRAD_ONE_ROUND
  // one round of workgroups (two per CU fit), each walking an equal share of the tiles with the next tile's loads in flight
  int workers = 512 / (C / CB);
  if (workers < 1) workers = 1;
  if (workers > ntiles) workers = ntiles;
  const int per = (ntiles + workers - 1) / workers;
  workers = (ntiles + per - 1) / per;
#else
  int workers = 1024 / (C / CB);  // ~4 workgroups per CU in total
  if (workers < 1) workers = 1;
  if (workers > ntiles) workers = ntiles;
#endif
  dim3 grid(workers, C / CB);
  size_t lds = (size_t)(LDS_PATCH_B + TH_B * TW * CB) * sizeof(float);
  if (lds < (size_t)4 * 7 * 8 * CB * sizeof(float)) lds = (size_t)4 * 7 * 8 * CB * sizeof(float);
  dwconv7_lds_bwd_weight_kernel<<<grid, 256, lds, st>>>(x, du, dw49, dbias, B, H, W, C);
}
