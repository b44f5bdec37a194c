// gemm_f32_tn.hip -- instantiations of the fp32 GEMM template for MODE_TN (wgrad: dY^T @ X, split over the token rows)
// and the implicit-GEMM weight gradient of the 3x3 convolutions.  See gemm_f32_kernel.h.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int BK, class TL>
static void go(const GemmParams& p, dim3 grid, hipStream_t st) {
  if (p.csum) gemm_f32_kernel<MODE_TN, EPI_NONE, BK, TL, 0, 0, 1><<<grid, NTHREADS, 0, st>>>(p);  // + column sums of A
  else gemm_f32_kernel<MODE_TN, EPI_NONE, BK, TL, 0><<<grid, NTHREADS, 0, st>>>(p);
}

int launch_tn(const GemmParams& p, int tile, int bk, int gather, dim3 grid, hipStream_t st) {
  if (gather) {  // 2: every k-tile lies inside one image row (rW % 16 == 0): scalar source-pixel arithmetic
    if (tile != 0 || bk != 16) return SM3_ERR_INVALID_ARG;
    if (gather == 2) gemm_f32_kernel<MODE_TN, EPI_NONE, 16, T128x128, 2><<<grid, NTHREADS, 0, st>>>(p);
    else gemm_f32_kernel<MODE_TN, EPI_NONE, 16, T128x128, 1><<<grid, NTHREADS, 0, st>>>(p);
    return SM3_OK;
  }
  switch (tile * 100 + bk) {
    case 16: go<16, T128x128>(p, grid, st); return SM3_OK;
    case 32: go<32, T128x128>(p, grid, st); return SM3_OK;
    case 116: go<16, T128x96>(p, grid, st); return SM3_OK;
    case 132: go<32, T128x96>(p, grid, st); return SM3_OK;
    case 216: go<16, T96x128>(p, grid, st); return SM3_OK;
    case 232: go<32, T96x128>(p, grid, st); return SM3_OK;
    case 316: go<16, T128x192>(p, grid, st); return SM3_OK;
    case 416: go<16, T192x128>(p, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

}  // namespace sm3gemm
