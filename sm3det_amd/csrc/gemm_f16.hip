// gemm_f16.hip -- instantiations of the GEMM template with fp16 operands / fp32 accumulation
// (v_mfma_f32_32x32x16_f16), the arithmetic of the reference's AMP configs (local_configs/SM3Det_convnext_t.py:8
// `fp16 = dict(loss_scale='dynamic')`: autocast runs every nn.Linear in half precision with fp32 accumulation,
// mmcv/mmcv/runner/fp16_utils.py:71-149).  Operands are converted on the fly from the fp32 tensors; see gemm_f32_kernel.h.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int MODE, int EPI, class TL>
static void go16(const GemmParams& p, int bk, dim3 grid, hipStream_t st) {
  if (bk == 16) gemm_f32_kernel<MODE, EPI, 16, TL, 0, 1><<<grid, NTHREADS, 0, st>>>(p);
  else if (bk == 32) gemm_f32_kernel<MODE, EPI, 32, TL, 0, 1><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE, EPI, 64, TL, 0, 1><<<grid, NTHREADS, 0, st>>>(p);
}

template <int MODE, int EPI>
static int by_tile16(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: go16<MODE, EPI, T128x128>(p, bk, grid, st); return SM3_OK;
    case 1: go16<MODE, EPI, T128x96>(p, bk, grid, st); return SM3_OK;
    case 5: go16<MODE, EPI, T64x128>(p, bk, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nt16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io) return launch_nt_h16(p, epi, tile, bk, io, grid, st);
  switch (epi) {
    case EPI_NONE: return by_tile16<MODE_NT, EPI_NONE>(p, tile, bk, grid, st);
    case EPI_BIAS: return by_tile16<MODE_NT, EPI_BIAS>(p, tile, bk, grid, st);
    case EPI_BIAS_GELU: return by_tile16<MODE_NT, EPI_BIAS_GELU>(p, tile, bk, grid, st);
    case EPI_BIAS_SCALE_RES: return by_tile16<MODE_NT, EPI_BIAS_SCALE_RES>(p, tile, bk, grid, st);
    case EPI_BIAS_RELU: return by_tile16<MODE_NT, EPI_BIAS_RELU>(p, tile, bk, grid, st);
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nn16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io) return launch_nn_h16(p, epi, tile, bk, io, grid, st);
  if (epi == EPI_NONE) return by_tile16<MODE_NN, EPI_NONE>(p, tile, bk, grid, st);
  if (epi == EPI_GELU_BWD) return by_tile16<MODE_NN, EPI_GELU_BWD>(p, tile, bk, grid, st);
  return SM3_ERR_INVALID_ARG;
}

template <class TL>
static void go16_tn(const GemmParams& p, int bk, dim3 grid, hipStream_t st) {
  if (!p.csum) return go16<MODE_TN, EPI_NONE, TL>(p, bk, grid, st);
  // + column sums of the (fp32) A operand: the bias gradient next to the weight gradient, as in the fp32 form
  if (bk == 16) gemm_f32_kernel<MODE_TN, EPI_NONE, 16, TL, 0, 1, 1><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE_TN, EPI_NONE, 32, TL, 0, 1, 1><<<grid, NTHREADS, 0, st>>>(p);
  // (k-step 64 -- reachable through the tuning override only -- falls back to 32 here: its 128x128 instantiation with the
  // column sums needs more than 256 registers)
}

int launch_tn16(const GemmParams& p, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io) return launch_tn_h16(p, tile, bk, io, grid, st);
  switch (tile) {
    case 0: go16_tn<T128x128>(p, bk, grid, st); return SM3_OK;
    case 1: go16_tn<T128x96>(p, bk, grid, st); return SM3_OK;
    case 2: go16_tn<T96x128>(p, bk, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

}  // namespace sm3gemm
