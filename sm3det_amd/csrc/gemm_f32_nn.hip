// gemm_f32_nn.hip -- instantiations of the fp32 GEMM template for MODE_NN (dgrad: dY @ W; EPI_NONE and EPI_GELU_BWD) and
// the implicit-GEMM input gradient of the 3x3 convolutions.  See gemm_f32_kernel.h.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int EPI, int BK, class TL>
static void go(const GemmParams& p, dim3 grid, hipStream_t st) {
  gemm_f32_kernel<MODE_NN, EPI, BK, TL, 0><<<grid, NTHREADS, 0, st>>>(p);
}

template <int EPI>
static int by_tile(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile * 100 + bk) {
    case 16: go<EPI, 16, T128x128>(p, grid, st); return SM3_OK;
    case 32: go<EPI, 32, T128x128>(p, grid, st); return SM3_OK;
    case 116: go<EPI, 16, T128x96>(p, grid, st); return SM3_OK;
    case 132: go<EPI, 32, T128x96>(p, grid, st); return SM3_OK;
    case 316: go<EPI, 16, T128x192>(p, grid, st); return SM3_OK;
    case 516: go<EPI, 16, T64x128>(p, grid, st); return SM3_OK;
    case 532: go<EPI, 32, T64x128>(p, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nn(const GemmParams& p, int epi, int tile, int bk, int gather, dim3 grid, hipStream_t st) {
  if (gather) {
    if (epi != EPI_NONE || tile != 0 || bk != 32) return SM3_ERR_INVALID_ARG;
    gemm_f32_kernel<MODE_NN, EPI_NONE, 32, T128x128, 1><<<grid, NTHREADS, 0, st>>>(p);
    return SM3_OK;
  }
  if (epi == EPI_NONE) return by_tile<EPI_NONE>(p, tile, bk, grid, st);
  if (epi == EPI_GELU_BWD) return by_tile<EPI_GELU_BWD>(p, tile, bk, grid, st);
  return SM3_ERR_INVALID_ARG;
}

}  // namespace sm3gemm
