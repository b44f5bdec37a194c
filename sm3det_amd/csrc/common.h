// common.h -- helpers shared by the gfx950 translation units of libsm3det_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sm3det_hip.h"

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Launch errors only (no synchronisation): the C ABI never blocks the host.
static inline int launch_status() { return hipGetLastError() == hipSuccess ? SM3_OK : SM3_ERR_LAUNCH; }

// Zero-fill as a KERNEL node.  hipMemsetAsync issued on the origin stream of a hipGraph capture did not stay ordered with
// the kernels around it on replay (ROCm 7.2: bias / depthwise gradients that are zeroed and then accumulated with atomics
// came back unzeroed or zeroed late; the same calls on a forked stream were fine), so the library never uses it.
__global__ static void sm3_zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline void sm3_zero_async(void* p, size_t bytes, hipStream_t st) {  // bytes: multiple of 4
  const size_t n = bytes / 4;
  if (n == 0) return;
  size_t blocks = (n + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  sm3_zero_words_kernel<<<(unsigned)blocks, 256, 0, st>>>((uint32_t*)p, n);
}

constexpr int kNumXCD = 8;  // MI355X: 8 XCDs x 32 CUs, block b runs on XCD b % 8 (speed only, never correctness)
constexpr int kNumCU = 256;
