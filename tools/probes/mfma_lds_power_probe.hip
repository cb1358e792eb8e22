// Energy, not issue slots: how much of the matrix pipe's sustainable rate do LDS operand reads cost?  A stream of independent
// v_mfma_f32_16x16x32_bf16 on pseudo-random bf16 operands (as tools/probes/mfma_power_probe.hip) with R ds_read_b128 (conflict-free, random
// data, results consumed as the next MFMAs' B operands) per 16 MFMAs: R = 0, 3, 8, 16 = 0 / 0.19 / 0.5 / 1.0 reads per MFMA -- 0.5 is what
// conv_ws.hip's main loop does, 0.21 what a patch-row walk would do.  One and two waves per SIMD.  Reports wall-clock TF/s.
// Build + run: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/mfma_lds_power_probe.hip -o tools/bin/mfma_lds_power_probe && tools/bin/mfma_lds_power_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int N_IT = 8192;

__device__ __forceinline__ unsigned rnd_bf16_pair(unsigned& s) {
  s = s * 1664525u + 1013904223u;
  const unsigned lo = (s >> 4) & 0x80ffu, hi = (s >> 13) & 0x80ffu;
  return (lo | 0x3f00u) | ((hi | 0x3f00u) << 16);
}

template <int R>
__global__ __launch_bounds__(512, 1) void stream(float* sink) {
  __shared__ u32x4_t lds[512 * 4];          // 32 KB of random bf16
  const int tid = threadIdx.x;
  unsigned s = 12345u + (unsigned)tid * 7919u + blockIdx.x * 104729u;
  for (int k = 0; k < 4; ++k) lds[k * 512 + tid] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)};
  __syncthreads();
  u32x4_t a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)};
    b[i] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)};
  }
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4_t*)lds + (unsigned)(tid & 63) * 16u + (unsigned)(tid >> 6) * 1024u;
  u32x4_t r[R > 0 ? R : 1];
  for (int it = 0; it < N_IT; ++it) {
    // this iteration's reads land under its MFMAs and become B operands of the NEXT iteration (lane-linear 1 KB per read: conflict free)
#pragma unroll
    for (int k = 0; k < R; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(base), "n"((k % 4) * 8192));
#pragma unroll
    for (int i = 0; i < 16; ++i)
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i & 3]));
    if (R > 0) {
      if constexpr (R >= 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]));
#pragma unroll
      for (int k = 0; k < (R < 4 ? R : 4); ++k) b[k] = r[k];
    }
    if ((it & 63) == 63) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = acc[i] * 0.5f;
    }
  }
  float q = 0.f;
  for (int i = 0; i < 16; ++i) q += acc[i][0] + acc[i][3];
  if (q == 12345.678f) sink[tid] = q;
}

template <int R> void run(int waves, float* sink, hipEvent_t e0, hipEvent_t e1) {
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(stream<R>, dim3(256), dim3(256 * waves), 0, 0, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
  }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 16.0 * N_IT * 16384.0 * 256 * 4 * waves;
  printf("%d wave(s) per SIMD, %2d ds_read_b128 per 16 MFMAs (%.2f per MFMA): %8.1f us  %7.1f TF/s (%4.1f %% of 2.5 PF)\n", waves, R, R / 16.0, ms * 1e3, flop / (ms * 1e-3) / 1e12,
         flop / (ms * 1e-3) / 2.5e13);
}

int main() {
  float* sink;
  if (hipMalloc(&sink, 4096) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves = 1; waves <= 2; ++waves) {
    run<0>(waves, sink, e0, e1);
    run<3>(waves, sink, e0, e1);
    run<8>(waves, sink, e0, e1);
    run<16>(waves, sink, e0, e1);
  }
  return 0;
}
