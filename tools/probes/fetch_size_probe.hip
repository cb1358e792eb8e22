// FETCH_SIZE calibration probe: the guide says rocprofv3's FETCH_SIZE reports HALF the bytes of a wide coalesced streaming read
// (16 B per lane) on gfx950 and that other access widths are uncalibrated.  The level-0 kernels read their selection codes ONE byte per
// lane (and 2 / 8 bytes per lane elsewhere), so this probe streams a known byte count -- 1 GiB, far past the 256 MiB Infinity Cache -- once
// per access width, and the same again through the LDS-DMA; `rocprofv3 --pmc FETCH_SIZE` over it gives the factor per width.
// Build + run:  hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_size_probe.hip -o tools/bin/fetch_size_probe
//               rocprofv3 --pmc FETCH_SIZE -d /tmp/fp -o pmc -- tools/bin/fetch_size_probe  (tools/pmc_summary.py on the database)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <typename T> __global__ void read_kernel(const T* __restrict__ p, size_t n, unsigned* out) {      // n elements of T, grid-stride, coalesced
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = p[i];
    const unsigned char* b = reinterpret_cast<const unsigned char*>(&v);
    acc += b[0];
  }
  if (acc == 0xffffffffu) out[0] = acc;       // never true for the probe's data: keeps the loads
}
__global__ void read_dma16_kernel(const uint4* __restrict__ p, size_t n, unsigned* out) {                  // the same 16 B per lane through the LDS-DMA
  __shared__ uint4 buf[256];
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + i + threadIdx.x),
                                     (__attribute__((address_space(3))) void*)(buf + (threadIdx.x & ~63)), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc += buf[threadIdx.x].x & 1u;
  }
  if (acc == 0xffffffffu) out[0] = acc;
}
int main() {
  const size_t bytes = (size_t)1 << 30;
  void* p; unsigned* out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(p, 1, bytes);
  (void)hipDeviceSynchronize();
  const int grid = 256 * 8;
  hipLaunchKernelGGL(read_kernel<unsigned char>, dim3(grid), dim3(256), 0, 0, (const unsigned char*)p, bytes, out);
  hipLaunchKernelGGL(read_kernel<unsigned short>, dim3(grid), dim3(256), 0, 0, (const unsigned short*)p, bytes / 2, out);
  hipLaunchKernelGGL(read_kernel<unsigned int>, dim3(grid), dim3(256), 0, 0, (const unsigned int*)p, bytes / 4, out);
  hipLaunchKernelGGL(read_kernel<uint2>, dim3(grid), dim3(256), 0, 0, (const uint2*)p, bytes / 8, out);
  hipLaunchKernelGGL(read_kernel<uint4>, dim3(grid), dim3(256), 0, 0, (const uint4*)p, bytes / 16, out);
  hipLaunchKernelGGL(read_dma16_kernel, dim3(grid), dim3(256), 0, 0, (const uint4*)p, bytes / 16, out);
  if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "probe failed\n"); return 2; }
  printf("fetch_size_probe: every kernel read %zu bytes (1048576 KB) once\n", bytes);
  return 0;
}
