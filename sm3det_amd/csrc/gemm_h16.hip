// gemm_h16.hip -- instantiations of the GEMM template for the AMP DATA PATH: fp16 operands / fp32 accumulation
// (v_mfma_f32_32x32x16_f16) with the activation-sized tensors STORED as fp16 in HBM (gemm_f32_kernel.h, IO flags) -- what
// `wrap_fp16_model` / autocast make of the reference's FFN (mmcv/mmcv/runner/fp16_utils.py:71-149, convnext_moe.py:397-405):
//   NT  A16            x(f16) . W^T            -> fp32   (FC2 + layer scale + residual; expert FC2; gate projection)
//   NT  A16|C16|X16    x(f16) . W^T            -> GELU (f16), GELU' (f16)                    (FC1)
//   NN  C16|X16        dy(fp32) . W            -> x GELU' (f16) = dh (f16)                   (FC2 input gradient)
//   NN  A16            dh(f16) . W             -> fp32                                      (FC1 input gradient)
//   TN  B16            dy(fp32)^T . act(f16)   -> fp32                                      (FC2 / gate weight gradient)
//   TN  A16|B16        dh(f16)^T . x(f16)      -> fp32                                      (FC1 weight gradient)
// Weights, biases, the residual stream and all C-wide gradients stay fp32.  With `| B16` on an NT / NN form the B operand is
// read from an fp16 SHADOW of the (fp32 master) weights: what the half model of wrap_fp16_model holds -- a third fewer
// operand bytes through the L1 (default; SM3_AMP_W16=0 reads the fp32 masters; backbone_ops._shadow).
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int MODE, int EPI, class TL, int IO>
static void goh(const GemmParams& p, int bk, dim3 grid, hipStream_t st) {
  if (bk == 16) gemm_f32_kernel<MODE, EPI, 16, TL, 0, 1, 0, IO><<<grid, NTHREADS, 0, st>>>(p);
  else if (bk == 32) gemm_f32_kernel<MODE, EPI, 32, TL, 0, 1, 0, IO><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE, EPI, 64, TL, 0, 1, 0, IO><<<grid, NTHREADS, 0, st>>>(p);
}

template <int MODE, int EPI, int IO>
static int by_tile_h(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: goh<MODE, EPI, T128x128, IO>(p, bk, grid, st); return SM3_OK;
    case 1: goh<MODE, EPI, T128x96, IO>(p, bk, grid, st); return SM3_OK;
    case 5: goh<MODE, EPI, T64x128, IO>(p, bk, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nt_h16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io == IO_A16) {
    if (epi == EPI_NONE) return by_tile_h<MODE_NT, EPI_NONE, IO_A16>(p, tile, bk, grid, st);
    if (epi == EPI_BIAS) return by_tile_h<MODE_NT, EPI_BIAS, IO_A16>(p, tile, bk, grid, st);
    if (epi == EPI_BIAS_SCALE_RES) return by_tile_h<MODE_NT, EPI_BIAS_SCALE_RES, IO_A16>(p, tile, bk, grid, st);
  } else if (io == (IO_A16 | IO_C16 | IO_X16)) {
    if (epi == EPI_BIAS_GELU) return by_tile_h<MODE_NT, EPI_BIAS_GELU, IO_A16 | IO_C16 | IO_X16>(p, tile, bk, grid, st);
  } else if (io == (IO_A16 | IO_B16)) {  // weights read from their fp16 shadow
    if (epi == EPI_NONE) return by_tile_h<MODE_NT, EPI_NONE, IO_A16 | IO_B16>(p, tile, bk, grid, st);
    if (epi == EPI_BIAS) return by_tile_h<MODE_NT, EPI_BIAS, IO_A16 | IO_B16>(p, tile, bk, grid, st);
    if (epi == EPI_BIAS_SCALE_RES) return by_tile_h<MODE_NT, EPI_BIAS_SCALE_RES, IO_A16 | IO_B16>(p, tile, bk, grid, st);
  } else if (io == (IO_A16 | IO_B16 | IO_C16 | IO_X16)) {
    if (epi == EPI_BIAS_GELU)
      return by_tile_h<MODE_NT, EPI_BIAS_GELU, IO_A16 | IO_B16 | IO_C16 | IO_X16>(p, tile, bk, grid, st);
  }
  return SM3_ERR_UNSUPPORTED;
}

int launch_nn_h16(const GemmParams& p, int epi, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io == IO_A16 && epi == EPI_NONE) return by_tile_h<MODE_NN, EPI_NONE, IO_A16>(p, tile, bk, grid, st);
  if (io == (IO_C16 | IO_X16) && epi == EPI_GELU_BWD)
    return by_tile_h<MODE_NN, EPI_GELU_BWD, IO_C16 | IO_X16>(p, tile, bk, grid, st);
  if (io == (IO_A16 | IO_B16) && epi == EPI_NONE) return by_tile_h<MODE_NN, EPI_NONE, IO_A16 | IO_B16>(p, tile, bk, grid, st);
  if (io == (IO_B16 | IO_C16 | IO_X16) && epi == EPI_GELU_BWD)
    return by_tile_h<MODE_NN, EPI_GELU_BWD, IO_B16 | IO_C16 | IO_X16>(p, tile, bk, grid, st);
  return SM3_ERR_UNSUPPORTED;
}

template <class TL, int IO>
static void goh_tn(const GemmParams& p, int bk, dim3 grid, hipStream_t st) {
  if (IO & IO_A16 || !p.csum) return goh<MODE_TN, EPI_NONE, TL, IO>(p, bk, grid, st);
  // + column sums of the fp32 A operand (dy): the FC2 / gate bias gradient as a by-product
  constexpr int IOC = IO & ~IO_A16;
  if (bk == 16) gemm_f32_kernel<MODE_TN, EPI_NONE, 16, TL, 0, 1, 1, IOC><<<grid, NTHREADS, 0, st>>>(p);
  else if (bk == 32) gemm_f32_kernel<MODE_TN, EPI_NONE, 32, TL, 0, 1, 1, IOC><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE_TN, EPI_NONE, 64, TL, 0, 1, 1, IOC><<<grid, NTHREADS, 0, st>>>(p);
}

template <int IO>
static int tn_by_tile_h(const GemmParams& p, int tile, int bk, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: goh_tn<T128x128, IO>(p, bk, grid, st); return SM3_OK;
    case 1: goh_tn<T128x96, IO>(p, bk, grid, st); return SM3_OK;
    case 2: goh_tn<T96x128, IO>(p, bk, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_tn_h16(const GemmParams& p, int tile, int bk, int io, dim3 grid, hipStream_t st) {
  if (io == IO_B16) return tn_by_tile_h<IO_B16>(p, tile, bk, grid, st);
  if (io == (IO_A16 | IO_B16)) return tn_by_tile_h<IO_A16 | IO_B16>(p, tile, bk, grid, st);
  return SM3_ERR_UNSUPPORTED;
}

}  // namespace sm3gemm
