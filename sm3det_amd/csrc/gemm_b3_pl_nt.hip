// gemm_b3_pl_nt.hip -- MODE_NT instantiations of the bf16x3 GEMM form whose operands arrive as bf16x3 PLANES (planes.hip):
// the loader moves 16-byte granules HBM -> VGPR -> LDS with no arithmetic (the fp32-operand form splits every element in
// every workgroup: 22 vector-ALU instructions per four elements, the bound of the loop at three workgroups per CU).
// io: IO_APL (A = the LayerNorm output / dispatched expert inputs, split by their producer), IO_BPL (B = weight planes kept
// per optimizer step), or both.  Reference arithmetic: FFN.forward / the expert loop, convnext_moe.py:397-405, :244.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <int EPI, int IO>
static int nt_pl_by_tile(const GemmParams& p, int tile, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: gemm_f32_kernel<MODE_NT, EPI, 16, T128x128, 0, 2, 0, IO><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 1: gemm_f32_kernel<MODE_NT, EPI, 16, T128x96, 0, 2, 0, IO><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
    case 5: gemm_f32_kernel<MODE_NT, EPI, 16, T64x128, 0, 2, 0, IO><<<grid, NTHREADS, 0, st>>>(p); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}

int launch_nt_b3_pl(const GemmParams& p, int epi, int tile, int io, dim3 grid, hipStream_t st) {
  constexpr int A = IO_APL, B = IO_BPL, AB = IO_APL | IO_BPL;
  if (epi == EPI_BIAS_GELU && io == AB) return nt_pl_by_tile<EPI_BIAS_GELU, AB>(p, tile, grid, st);   // FC1: planes x planes
  if (epi == EPI_BIAS_GELU && io == B) return nt_pl_by_tile<EPI_BIAS_GELU, B>(p, tile, grid, st);
  if (epi == EPI_BIAS_SCALE_RES && io == B) return nt_pl_by_tile<EPI_BIAS_SCALE_RES, B>(p, tile, grid, st);  // FC2 (dense block)
  if (epi == EPI_BIAS && io == B) return nt_pl_by_tile<EPI_BIAS, B>(p, tile, grid, st);                // FC2 (experts)
  if (epi == EPI_BIAS && io == A) return nt_pl_by_tile<EPI_BIAS, A>(p, tile, grid, st);                // gate projection
  if (epi == EPI_BIAS && io == AB) return nt_pl_by_tile<EPI_BIAS, AB>(p, tile, grid, st);
  if (epi == EPI_NONE && io == AB) return nt_pl_by_tile<EPI_NONE, AB>(p, tile, grid, st);
  if (epi == EPI_NONE && io == B) return nt_pl_by_tile<EPI_NONE, B>(p, tile, grid, st);
  if (epi == EPI_NONE && io == A) return nt_pl_by_tile<EPI_NONE, A>(p, tile, grid, st);
  return SM3_ERR_UNSUPPORTED;
}

}  // namespace sm3gemm
