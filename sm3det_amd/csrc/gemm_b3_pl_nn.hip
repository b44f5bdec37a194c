// gemm_b3_pl_nn.hip -- MODE_NN (input gradient dY . W) instantiations of the bf16x3 form with operands as bf16x3 planes; see
// gemm_b3_pl_nt.hip.  B = the planes of W^T (k-contiguous form of the k-major operand), A = the planes of dY.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int EPI, int IO>
static int nn_pl_by_tile(const GemmParams& p, int tile, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: gemm_f32_kernel<MODE_NN, EPI, 16, T128x128, 0, 2, 0, IO><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 1: gemm_f32_kernel<MODE_NN, EPI, 16, T128x96, 0, 2, 0, IO><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 5: gemm_f32_kernel<MODE_NN, EPI, 16, T64x128, 0, 2, 0, IO><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nn_b3_pl(const GemmParams& p, int epi, int tile, int io, dim3 grid, hipStream_t st) {
  constexpr int A = IO_APL, B = IO_BPL, AB = IO_APL | IO_BPL;
  if (epi == EPI_GELU_BWD && io == AB) return nn_pl_by_tile<EPI_GELU_BWD, AB>(p, tile, grid, st);  // FC2 dgrad: planes x planes
  if (epi == EPI_GELU_BWD && io == B) return nn_pl_by_tile<EPI_GELU_BWD, B>(p, tile, grid, st);
  if (epi == EPI_NONE && io == B) return nn_pl_by_tile<EPI_NONE, B>(p, tile, grid, st);            // FC1 dgrad
  if (epi == EPI_NONE && io == AB) return nn_pl_by_tile<EPI_NONE, AB>(p, tile, grid, st);
  if (epi == EPI_NONE && io == A) return nn_pl_by_tile<EPI_NONE, A>(p, tile, grid, st);
  return SM3_ERR_UNSUPPORTED;
}

}  // namespace sm3gemm
