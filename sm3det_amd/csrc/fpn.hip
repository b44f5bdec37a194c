// fpn.hip -- the element-wise half of the FPN top-down path on NHWC tokens (MultitaskFPN.forward,
// mmrotate/models/necks/Multitask_FPN.py:123-135: laterals[i-1] = laterals[i-1] + F.interpolate(laterals[i],
// size=prev_shape, mode='nearest')) and its gradient.  The contractions (1x1 laterals, 3x3 output convs) run on the GEMM
// family (gemm_f32.hip).  HBM-bound: 16-byte lanes over channel quads, one pass.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[b,y,x,:] = fine[b,y,x,:] + coarse[b,y/2,x/2,:]   (H = 2*Hc, W = 2*Wc: nearest == index >> 1)
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ fine,
                                                            const float* __restrict__ coarse,
                                                            float* __restrict__ out, long npix, int H, int W,
                                                            int nq) {
  const long total = npix * nq;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / nq;
    const int q = (int)(i - pix * nq);
    const int x = (int)(pix % W);
    const long t = pix / W;
    const int y = (int)(t % H);
    const long b = t / H;
    const long cp = (b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
    const f32x4 a = *reinterpret_cast<const f32x4*>(fine + (pix * nq + q) * 4);
    const f32x4 c = *reinterpret_cast<const f32x4*>(coarse + (cp * nq + q) * 4);
    *reinterpret_cast<f32x4*>(out + (pix * nq + q) * 4) = a + c;
  }
}

// dcoarse[b,y,x,:] = base[b,y,x,:] + sum over the 2x2 fine pixels that read it
__global__ __launch_bounds__(256) void sumpool2x_add_kernel(const float* __restrict__ dfine,
                                                           const float* __restrict__ base,
                                                           float* __restrict__ dcoarse, long npix, int Hc, int Wc,
                                                           int nq) {
  const long total = npix * nq;
  const int W = 2 * Wc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / nq;
    const int q = (int)(i - pix * nq);
    const int x = (int)(pix % Wc);
    const long t = pix / Wc;
    const int y = (int)(t % Hc);
    const long b = t / Hc;
    const long f00 = ((b * 2 * Hc + 2 * y) * W + 2 * x);
    const float* p0 = dfine + (f00 * nq + q) * 4;
    const float* p1 = dfine + ((f00 + W) * nq + q) * 4;
    f32x4 s = *reinterpret_cast<const f32x4*>(p0) + *reinterpret_cast<const f32x4*>(p0 + 4 * nq);
    s += *reinterpret_cast<const f32x4*>(p1) + *reinterpret_cast<const f32x4*>(p1 + 4 * nq);
    if (base) s += *reinterpret_cast<const f32x4*>(base + (pix * nq + q) * 4);
    *reinterpret_cast<f32x4*>(dcoarse + (pix * nq + q) * 4) = s;
  }
}

int grid_for(long total) {
  long nb = (total + 255) / 256;
  if (nb > 256 * 16) nb = 256 * 16;
  return nb < 1 ? 1 : (int)nb;
}


// Batched 2-D transpose dst[b][c][r] = src[b][r][c] through a 32x33 LDS tile (NHWC <-> NCHW of a feature map:
// rows = H*W, cols = C or the other way round).  Both sides move full 128-byte row segments.
// ADD: dst += src^T (the mmcv RoIAlignRotated backward ACCUMULATES into the caller's grad_input).  Row tiles ride on
// grid.x (no 65535 limit on H*W), column tiles on grid.y.
template <bool ADD>
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           int rows, int cols) {
  __shared__ float tile[32][33];
  const long base = (long)blockIdx.z * rows * cols;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * i][tx] = src[base + (long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < rows && c < cols) {
      float* d = dst + base + (long)c * rows + r;
      *d = ADD ? *d + tile[tx][ty + 8 * i] : tile[tx][ty + 8 * i];
    }
  }
}

}  // namespace

extern "C" {

int sm3_upsample2x_add(const float* fine, const float* coarse, float* out, int B, int H, int W, int C,
                       sm3_stream_t stream) {
  if (!fine || !coarse || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0) return SM3_ERR_INVALID_ARG;
  if ((C & 3) || (H & 1) || (W & 1)) return SM3_ERR_UNSUPPORTED;
  const long npix = (long)B * H * W;
  upsample2x_add_kernel<<<grid_for(npix * (C / 4)), 256, 0, (hipStream_t)stream>>>(fine, coarse, out, npix, H, W,
                                                                                    C / 4);
  return launch_status();
}

int sm3_sumpool2x_add(const float* dfine, const float* base, float* dcoarse, int B, int Hc, int Wc, int C,
                      sm3_stream_t stream) {
  if (!dfine || !dcoarse || B <= 0 || Hc <= 0 || Wc <= 0 || C <= 0) return SM3_ERR_INVALID_ARG;
  if (C & 3) return SM3_ERR_UNSUPPORTED;
  const long npix = (long)B * Hc * Wc;
  sumpool2x_add_kernel<<<grid_for(npix * (C / 4)), 256, 0, (hipStream_t)stream>>>(dfine, base, dcoarse, npix, Hc, Wc,
                                                                                   C / 4);
  return launch_status();
}

int sm3_transpose_f32(const float* src, float* dst, int batch, int rows, int cols, sm3_stream_t stream) {
  if (!src || !dst || batch <= 0 || rows <= 0 || cols <= 0 || batch > 65535) return SM3_ERR_INVALID_ARG;
  dim3 grid((rows + 31) / 32, (cols + 31) / 32, batch);
  if (grid.y > 65535) return SM3_ERR_UNSUPPORTED;
  transpose_f32_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(src, dst, rows, cols);
  return launch_status();
}

int sm3_transpose_add_f32(const float* src, float* dst, int batch, int rows, int cols, sm3_stream_t stream) {
  if (!src || !dst || batch <= 0 || rows <= 0 || cols <= 0 || batch > 65535) return SM3_ERR_INVALID_ARG;
  dim3 grid((rows + 31) / 32, (cols + 31) / 32, batch);
  if (grid.y > 65535) return SM3_ERR_UNSUPPORTED;
  transpose_f32_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(src, dst, rows, cols);
  return launch_status();
}

}  // extern "C"
