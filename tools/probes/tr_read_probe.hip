// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value = its own element index.
// Each lane supplies a byte address; we print which 4 elements each lane receives for a few address patterns.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(int pattern, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;
  if (pattern == 0) addr = l * 8;                       // lane l -> elements 4l..4l+3
  else if (pattern == 1) addr = (l & 15) * 64 + (l >> 4) * 8;  // rows of 32 elements (64 B): row = l&15, col block = l>>4
  else if (pattern == 2) addr = (l & 15) * 8 + (l >> 4) * 512; // 16 consecutive 8-B items per 16-lane group, groups 512 B apart
  else addr = (l >> 4) * 8 + (l & 15) * 256;                  // row = l&15 with 256-B pitch, col block = l>>4
  addr += (uint32_t)(uintptr_t)lds;
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (uint16_t)(v & 0xffff);
  out[l * 4 + 1] = (uint16_t)((v >> 16) & 0xffff);
  out[l * 4 + 2] = (uint16_t)((v >> 32) & 0xffff);
  out[l * 4 + 3] = (uint16_t)((v >> 48) & 0xffff);
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int p = 0; p < 4; ++p) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, p, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d\n", p);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
      if (l % 4 == 3) printf("\n");
    }
  }
  return 0;
}
