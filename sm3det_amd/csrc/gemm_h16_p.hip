// gemm_h16_p.hip -- PERSISTENT instantiations of the GEMM template for the AMP data path (fp16 operands / fp32 accumulation,
// activation-sized tensors stored as fp16: gemm_h16.hip lists the combinations), NT and NN; see gemm_f32_p.hip.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int MODE, int EPI, class TL, int IO>
static void goh_p(const GemmParams& p, int bk, dim3 grid, hipStream_t st) {
  if (bk == 16) gemm_f32_kernel<MODE, EPI, 16, TL, 0, 1, 0, IO, 1><<<grid, NTHREADS, 0, st>>>(p);
  else if (bk == 32) gemm_f32_kernel<MODE, EPI, 32, TL, 0, 1, 0, IO, 1><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE, EPI, 64, TL, 0, 1, 0, IO, 1><<<grid, NTHREADS, 0, st>>>(p);
}

template <int MODE, int EPI, int IO>
static int by_tile_hp(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: goh_p<MODE, EPI, T128x128, IO>(p, bk, grid, st); return SM3_OK;
    case 1: goh_p<MODE, EPI, T128x96, IO>(p, bk, grid, st); return SM3_OK;
    case 5: goh_p<MODE, EPI, T64x128, IO>(p, bk, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

bool has_persistent_h16(int mode, int epi, int tile, int bk, int io) {
  if (!(tile == 0 || tile == 1 || tile == 5) || !(bk == 16 || bk == 32 || bk == 64)) return false;
  if (mode == MODE_NT) {
    if (io == IO_A16) return epi == EPI_NONE || epi == EPI_BIAS || epi == EPI_BIAS_SCALE_RES;
    return io == (IO_A16 | IO_C16 | IO_X16) && epi == EPI_BIAS_GELU;
  }
  if (mode == MODE_NN) return (io == IO_A16 && epi == EPI_NONE) || (io == (IO_C16 | IO_X16) && epi == EPI_GELU_BWD);
  return false;
}

int launch_nt_h16_p(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io == IO_A16) {
    if (epi == EPI_NONE) return by_tile_hp<MODE_NT, EPI_NONE, IO_A16>(p, tile, bk, grid, st);
    if (epi == EPI_BIAS) return by_tile_hp<MODE_NT, EPI_BIAS, IO_A16>(p, tile, bk, grid, st);
    if (epi == EPI_BIAS_SCALE_RES) return by_tile_hp<MODE_NT, EPI_BIAS_SCALE_RES, IO_A16>(p, tile, bk, grid, st);
  } else if (io == (IO_A16 | IO_C16 | IO_X16)) {
    if (epi == EPI_BIAS_GELU) return by_tile_hp<MODE_NT, EPI_BIAS_GELU, IO_A16 | IO_C16 | IO_X16>(p, tile, bk, grid, st);
  }
  return SM3_ERR_UNSUPPORTED;
}

int launch_nn_h16_p(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io == IO_A16 && epi == EPI_NONE) return by_tile_hp<MODE_NN, EPI_NONE, IO_A16>(p, tile, bk, grid, st);
  if (io == (IO_C16 | IO_X16) && epi == EPI_GELU_BWD)
    return by_tile_hp<MODE_NN, EPI_GELU_BWD, IO_C16 | IO_X16>(p, tile, bk, grid, st);
  return SM3_ERR_UNSUPPORTED;
}

}  // namespace sm3gemm
