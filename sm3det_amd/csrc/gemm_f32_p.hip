// gemm_f32_p.hip -- PERSISTENT instantiations of the fp32 GEMM template (gemm_f32_kernel.h, "work items"): NT and NN without
// GATHER, tiles 128x128 / 128x96 / 64x128, k-step 16 / 32.  Launched by sm3_gemm_f32 when a GEMM has more output tiles than
// the chip holds workgroups at once: grid.x = the resident set, p.total_tiles = the tiles it walks.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int MODE, int EPI, int BK, class TL>
static void go_p(const GemmParams& p, dim3 grid, hipStream_t st) {
  gemm_f32_kernel<MODE, EPI, BK, TL, 0, 0, 0, 0, 1><<<grid, NTHREADS, 0, st>>>(p);
}

template <int MODE, int EPI>
static int by_tile_p(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile * 100 + bk) {
    case 16: go_p<MODE, EPI, 16, T128x128>(p, grid, st); return SM3_OK;
    case 32: go_p<MODE, EPI, 32, T128x128>(p, grid, st); return SM3_OK;
    case 116: go_p<MODE, EPI, 16, T128x96>(p, grid, st); return SM3_OK;
    case 132: go_p<MODE, EPI, 32, T128x96>(p, grid, st); return SM3_OK;
    case 516: go_p<MODE, EPI, 16, T64x128>(p, grid, st); return SM3_OK;
    case 532: go_p<MODE, EPI, 32, T64x128>(p, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

bool has_persistent_f32(int tile, int bk) { return (tile == 0 || tile == 1 || tile == 5) && (bk == 16 || bk == 32); }

int launch_nt_p(const GemmParams& p, int epi, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_NONE: return by_tile_p<MODE_NT, EPI_NONE>(p, tile, bk, grid, st);
    case EPI_BIAS: return by_tile_p<MODE_NT, EPI_BIAS>(p, tile, bk, grid, st);
    case EPI_BIAS_GELU: return by_tile_p<MODE_NT, EPI_BIAS_GELU>(p, tile, bk, grid, st);
    case EPI_BIAS_SCALE_RES: return by_tile_p<MODE_NT, EPI_BIAS_SCALE_RES>(p, tile, bk, grid, st);
    case EPI_BIAS_RELU: return by_tile_p<MODE_NT, EPI_BIAS_RELU>(p, tile, bk, grid, st);
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nn_p(const GemmParams& p, int epi, int tile, int bk, dim3 grid, hipStream_t st) {
  if (epi == EPI_NONE) return by_tile_p<MODE_NN, EPI_NONE>(p, tile, bk, grid, st);
  if (epi == EPI_GELU_BWD) return by_tile_p<MODE_NN, EPI_GELU_BWD>(p, tile, bk, grid, st);
  return SM3_ERR_INVALID_ARG;
}

}  // namespace sm3gemm
