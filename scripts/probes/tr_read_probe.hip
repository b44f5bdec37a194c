// What does ds_read_b64_tr_b16 return?  LDS holds lds16[i] = i; every lane passes its own byte address; the four 16-bit
// values each lane receives are printed for a few address patterns.   hipcc --offload-arch=gfx950 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const int* __restrict__ addr, uint16_t* __restrict__ out) {
  __shared__ uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)lds;  // LDS byte offset (address space 3 pointers are 32-bit offsets)
  unsigned a = base + (unsigned)addr[threadIdx.x];
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (uint16_t)(v[0] & 0xffff);
  out[threadIdx.x * 4 + 1] = (uint16_t)(v[0] >> 16);
  out[threadIdx.x * 4 + 2] = (uint16_t)(v[1] & 0xffff);
  out[threadIdx.x * 4 + 3] = (uint16_t)(v[1] >> 16);
}
int main() {
  int h_addr[64]; uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  const char* names[] = {"addr = 8*lane (contiguous rows of 4)", "addr = 32*lane (row stride 32 B)", "addr = 128*(lane%16) + 8*(lane/16)",
                         "addr = 64*(lane%16)+8*(lane/16)"};
  for (int p = 0; p < 4; p++) {
    for (int l = 0; l < 64; l++)
      h_addr[l] = p == 0 ? 8 * l : p == 1 ? 32 * l : p == 2 ? 128 * (l % 16) + 8 * (l / 16) : 64 * (l % 16) + 8 * (l / 16);
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d: %s (values = 16-bit element index in LDS; element index = byte address / 2)\n", p, names[p]);
    for (int l = 0; l < 64; l++)
      printf("  lane %2d addr %5d (elem %4d): %5d %5d %5d %5d\n", l, h_addr[l], h_addr[l] / 2, h_out[4 * l], h_out[4 * l + 1], h_out[4 * l + 2], h_out[4 * l + 3]);
  }
  return 0;
}
