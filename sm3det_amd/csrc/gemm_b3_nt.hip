// gemm_b3_nt.hip -- MODE_NT instantiations of the GEMM template in its bf16x3 form (F16 = 2): fp32 tensors, every operand
// element split exactly into three bf16 pieces in the loader, six v_mfma_f32_32x32x16_bf16 products per 16 k with fp32
// accumulation -- fp32-equivalent results at 6/16 of the issue time of v_mfma_f32_32x32x2_f32 (gemm_f32_kernel.h, "B3").
// Reference arithmetic being reproduced: the fp32 nn.Linear / 1x1 convolutions of FFN.forward and the expert loop
// (mmrotate/models/backbones/convnext_moe.py:397-405, :244).
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int EPI>
static int nt_b3_by_tile(const GemmParams& p, int tile, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: gemm_f32_kernel<MODE_NT, EPI, 16, T128x128, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 1: gemm_f32_kernel<MODE_NT, EPI, 16, T128x96, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 5: gemm_f32_kernel<MODE_NT, EPI, 16, T64x128, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 3: gemm_f32_kernel<MODE_NT, EPI, 16, T128x192, 0, 2><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;  // two workgroups per CU
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nt_b3(const GemmParams& p, int epi, int tile, dim3 grid, hipStream_t st) {
  switch (epi) {
    case EPI_NONE: return nt_b3_by_tile<EPI_NONE>(p, tile, grid, st);
    case EPI_BIAS: return nt_b3_by_tile<EPI_BIAS>(p, tile, grid, st);
    case EPI_BIAS_GELU: return nt_b3_by_tile<EPI_BIAS_GELU>(p, tile, grid, st);
    case EPI_BIAS_SCALE_RES: return nt_b3_by_tile<EPI_BIAS_SCALE_RES>(p, tile, grid, st);
    case EPI_BIAS_RELU: return nt_b3_by_tile<EPI_BIAS_RELU>(p, tile, grid, st);
  }
  return SM3_ERR_INVALID_ARG;
}


// 3x3 convolution forward (GATHER = 1): the A operand is gathered tap by tap from the NHWC input
int launch_nt_b3_conv(const GemmParams& p, int epi, dim3 grid, hipStream_t st) {
  if (epi == EPI_NONE) gemm_f32_kernel<MODE_NT, EPI_NONE, 16, T128x128, 1, 2><<<grid, NTHREADS, 0, st>>>(p);
  else if (epi == EPI_BIAS) gemm_f32_kernel<MODE_NT, EPI_BIAS, 16, T128x128, 1, 2><<<grid, NTHREADS, 0, st>>>(p);
  else if (epi == EPI_BIAS_RELU) gemm_f32_kernel<MODE_NT, EPI_BIAS_RELU, 16, T128x128, 1, 2><<<grid, NTHREADS, 0, st>>>(p);
  else return SM3_ERR_INVALID_ARG;
  return SM3_OK;
}

}  // namespace sm3gemm
