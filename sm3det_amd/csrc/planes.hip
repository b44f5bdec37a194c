// planes.hip -- bf16x3 operand planes of an fp32 matrix (libsm3det_hip.so, gfx950).
//
// The bf16x3 GEMMs (gemm_f32_kernel.h, F16 = 2) evaluate an fp32 contraction on the bf16 matrix pipe by splitting every
// operand element EXACTLY into three bf16 pieces x = x0 + x1 + x2 (round-to-nearest-even each).  Done in the GEMM loader,
// that split is 22 vector-ALU instructions per four elements, repeated by every workgroup that touches the element (a
// weight: by every row tile of the launch) -- and with three workgroups per CU the loop is bound by exactly that vector
// work (profiles/r06/gemm_b3_ablations.txt: 2.2 of 12.2 ms).  An operand that is read by many launches / tiles is
// therefore split ONCE into "planes" and the loader moves the pieces as they are.
//
// Plane layout of X[R][K] (K % 8 == 0):  planes[p][o][r][j] = piece p of X[r][8 o + j]   (p < 3, o < K / 8, r < Rp, j < 8)
// i.e. per plane and k-octet one 16-byte granule per row, rows contiguous: 64 lanes fetch 64 rows of one octet as 1 KiB of
// contiguous memory, and a granule IS one lane's fragment of v_mfma_f32_32x32x16_bf16 (8 consecutive k of one row), so it
// goes from HBM to the LDS image with one 16-byte load and one 16-byte store, no arithmetic.
// transpose != 0 writes the planes of X^T (rows of the planes = columns of X): the k-major operand of the NN launches
// (dgrad: dY . W with W stored (out, in)) becomes k-contiguous that way.
//
// Reference arithmetic being reproduced: the fp32 nn.Linear layers of FFN.forward / the expert loop
// (mmrotate/models/backbones/convnext_moe.py:397-405, :244); the split is bit-identical to the loader's (`split3`).
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t cvt2(float lo, float hi) {  // v_cvt_pk_bf16_f32: two RNE bf16 in one dword
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
__device__ __forceinline__ float lo_f(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float hi_f(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// eight fp32 values -> the three 16-byte granules
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& p0, u32x4& p1, u32x4& p2) {
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    h[q] = cvt2(v[2 * q], v[2 * q + 1]);
    const float r0 = v[2 * q] - lo_f(h[q]), r1 = v[2 * q + 1] - hi_f(h[q]);  // exact
    m[q] = cvt2(r0, r1);
    const float s0 = r0 - lo_f(m[q]), s1 = r1 - hi_f(m[q]);                  // exact
    l[q] = cvt2(s0, s1);
  }
  p0 = u32x4{h[0], h[1], h[2], h[3]};
  p1 = u32x4{m[0], m[1], m[2], m[3]};
  p2 = u32x4{l[0], l[1], l[2], l[3]};
}

// One workgroup = one 64 x 64 tile of X through LDS (coalesced fp32 rows in, 1 KiB granule runs out).
// TR = 0: planes of X (octets along X's columns);  TR = 1: planes of X^T (octets along X's rows).
template <int TR>
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, int rows, int cols, long ld,
                                                           uint32_t* __restrict__ planes, long Rp, long row_off,
                                                           long plane_stride_granules) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
  {
    const int c = tid & 63, rr = tid >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int r = rr + 4 * i;
      tile[r][c] = (r0 + r < rows && c0 + c < cols) ? x[(long)(r0 + r) * ld + c0 + c] : 0.f;
    }
  }
  __syncthreads();
  const int lane_r = tid & 63;
#pragma unroll
  for (int it = 0; it < 2; it++) {
    const int o = (tid >> 6) + 4 * it;  // octet inside the tile (0..7)
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = TR ? tile[8 * o + j][lane_r] : tile[lane_r][8 * o + j];
    u32x4 p0, p1, p2;
    split8(v, p0, p1, p2);
    // TR = 0: granule (octet (c0 + 8 o) / 8, row r0 + lane);  TR = 1: granule (octet (r0 + 8 o) / 8, row c0 + lane)
    const long oct = TR ? (r0 >> 3) + o : (c0 >> 3) + o;
    const long row = TR ? c0 + lane_r : r0 + lane_r;
    const bool ok = TR ? (row < cols && 8 * oct < rows) : (row < rows && 8 * oct < cols);
    if (ok) {
      const long g = oct * Rp + row_off + row;  // granule index inside a plane
      u32x4* dst = reinterpret_cast<u32x4*>(planes) + g;
      dst[0] = p0;
      dst[plane_stride_granules] = p1;
      dst[2 * plane_stride_granules] = p2;
    }
  }
}

}  // namespace

extern "C" {

// planes: 3 planes of (K8 octets) x Rp rows x 8 bf16, K8 = (transpose ? rows : cols) / 8.  The matrix lands at rows
// [row_off, row_off + (transpose ? cols : rows)) of every octet block, so several matrices (the experts of a layer) can share
// one planes tensor.  rows / cols of the octet dimension must be a multiple of 8.
int sm3_split_planes_f32(const float* x, int rows, int cols, long ld, void* planes, long Rp, long row_off, int transpose,
                         sm3_stream_t stream) {
  if (!x || !planes || rows <= 0 || cols <= 0 || ld < cols || Rp <= 0 || row_off < 0) return SM3_ERR_INVALID_ARG;
  const long k = transpose ? rows : cols, r = transpose ? cols : rows;
  if ((k & 7) || row_off + r > Rp) return SM3_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(planes) & 15) != 0) return SM3_ERR_INVALID_ARG;
  const long psg = (k / 8) * Rp;  // granules per plane
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (transpose)
    split_planes_kernel<1><<<grid, 256, 0, (hipStream_t)stream>>>(x, rows, cols, ld, (uint32_t*)planes, Rp, row_off, psg);
  else
    split_planes_kernel<0><<<grid, 256, 0, (hipStream_t)stream>>>(x, rows, cols, ld, (uint32_t*)planes, Rp, row_off, psg);
  return launch_status();
}

}  // extern "C"
