// mfma_peak.hip -- measurement aid (not part of libsm3det_hip.so): the fp32 matrix rate this chip SUSTAINS with live data.
// Every wave runs `iters` x 8 independent v_mfma_f32_32x32x2_f32 (8 accumulator tiles: no dependent-issue stall) on
// operand fragments read from memory (random or zero), and stamps s_memtime / s_memrealtime around the loop:
//   * wall time of the launch  -> TFLOP/s with all SIMDs busy (the roofline the GEMM family can actually be priced against)
//   * s_memtime ticks / MFMA   -> 64 if s_memtime counts shader cycles (issue-bound), tells the shader clock via realtime
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/probes/mfma_peak.so scripts/probes/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

extern "C" __global__ __launch_bounds__(256, 1) void mfma_peak_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                     unsigned long long* __restrict__ stamps, int iters) {
  const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  float a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    a[i] = src[(gid * 16 + i) & 0xfffff];
    b[i] = src[(gid * 16 + 8 + i) & 0xfffff];
  }
  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  dst[gid] = s;
  if ((tid & 63) == 0) {
    unsigned long long* st = stamps + ((size_t)blockIdx.x * 4 + (tid >> 6)) * 4;
    st[0] = t0; st[1] = t1; st[2] = r0; st[3] = r1;
  }
}

extern "C" int mfma_peak_launch(const float* src, float* dst, unsigned long long* stamps, int blocks, int iters, void* stream) {
  mfma_peak_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(src, dst, stamps, iters);
  return (int)hipGetLastError();
}
