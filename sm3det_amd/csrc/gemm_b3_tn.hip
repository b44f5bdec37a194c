// gemm_b3_tn.hip -- MODE_TN (weight gradient dY^T . X, split over the token rows) instantiations of the bf16x3 form, with
// and without the column sums of A (the bias gradient) as a by-product; see gemm_b3_nt.hip.
#include "gemm_f32_kernel.h"

namespace sm3gemm {

template <class TL>
static void go_tn_b3(const GemmParams& p, dim3 grid, hipStream_t st) {
  if (p.csum) gemm_f32_kernel<MODE_TN, EPI_NONE, 16, TL, 0, 2, 1><<<grid, NTHREADS, 0, st>>>(p);
  else gemm_f32_kernel<MODE_TN, EPI_NONE, 16, TL, 0, 2><<<grid, NTHREADS, 0, st>>>(p);
}

int launch_tn_b3(const GemmParams& p, int tile, dim3 grid, hipStream_t st) {
  switch (tile) {
    case 0: go_tn_b3<T128x128>(p, grid, st); return SM3_OK;
    case 1: go_tn_b3<T128x96>(p, grid, st); return SM3_OK;
    case 2: go_tn_b3<T96x128>(p, grid, st); return SM3_OK;
    case 3: go_tn_b3<T128x192>(p, grid, st); return SM3_OK;  // six 32x32 blocks per wave: two workgroups per CU
    case 4: go_tn_b3<T192x128>(p, grid, st); return SM3_OK;
  }
  return SM3_ERR_INVALID_ARG;
}


// 3x3 convolution weight gradient (GATHER = 2: a k-tile of 16 output positions lies inside one image row)
int launch_tn_b3_conv(const GemmParams& p, dim3 grid, hipStream_t st) {
  gemm_f32_kernel<MODE_TN, EPI_NONE, 16, T128x128, 2, 2><<<grid, NTHREADS, 0, st>>>(p);
  return SM3_OK;
}

}  // namespace sm3gemm
