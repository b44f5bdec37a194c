// gemm_b3_nn.hip -- MODE_NN (input gradient dY . W) instantiations of the bf16x3 form; see gemm_b3_nt.hip.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int EPI>
static int nn_b3_by_tile(const GemmParams& p, int tile, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: gemm_f32_kernel<MODE_NN, EPI, 16, T128x128, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 1: gemm_f32_kernel<MODE_NN, EPI, 16, T128x96, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 5: gemm_f32_kernel<MODE_NN, EPI, 16, T64x128, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 3: gemm_f32_kernel<MODE_NN, EPI, 16, T128x192, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;  // two workgroups per CU
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nn_b3(const GemmParams& p, int epi, int tile, dim3 grid, hipStream_t st) {
  if (epi == EPI_NONE) return nn_b3_by_tile<EPI_NONE>(p, tile, grid, st);
  if (epi == EPI_GELU_BWD) return nn_b3_by_tile<EPI_GELU_BWD>(p, tile, grid, st);
  return SM3_ERR_INVALID_ARG;
}


// 3x3 convolution input gradient (GATHER = 1: transposed gather of dY, weights read as [tap][Cout] x Cin)
int launch_nn_b3_conv(const GemmParams& p, dim3 grid, hipStream_t st) {
  gemm_f32_kernel<MODE_NN, EPI_NONE, 16, T128x128, 1, 2><<<grid, NTHREADS, 0, st>>>(p);
  return SM3_OK;
}

}  // namespace sm3gemm
