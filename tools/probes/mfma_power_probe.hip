// What does the matrix pipe sustain on REAL operands?  The guide's DVFS note: zero-filled inputs ran +19 % TF/s at the same cycle count (the chip
// clocks to its power budget).  This probe runs a pure stream of independent v_mfma_f32_16x16x32_bf16 from W waves per SIMD on every CU and
// reports TF/s (wall clock) and shader cycles per MFMA, for operands that are all zeros, one constant, and pseudo-random bf16 in [-2, 2)
// (4 different A and 4 different B registers are cycled) -- the ceiling a dense kernel can reach on activations-like data.  Round 6: the same
// for v_mfma_f32_32x32x16_bf16 (twice the flops per operand register read).
// Build + run: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/mfma_power_probe.hip -o tools/bin/mfma_power_probe && tools/bin/mfma_power_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int N_IT = 8192;       // x 16 MFMAs per wave

__device__ __forceinline__ unsigned rnd_bf16_pair(unsigned& s) {      // two bf16 in [-2, 2): sign random, exponent 0x3e..0x3f, mantissa random
  s = s * 1664525u + 1013904223u;
  const unsigned lo = (s >> 4) & 0x80ffu, hi = (s >> 13) & 0x80ffu;
  return (lo | 0x3f00u) | ((hi | 0x3f00u) << 16);
}

__global__ __launch_bounds__(1024, 1) void mfma_stream(int data, float* sink, long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63;
  u32x4_t a[4], b[4];
  unsigned s = 12345u + (unsigned)tid * 7919u + blockIdx.x * 104729u;
  for (int i = 0; i < 4; ++i) {
    if (data == 0) { a[i] = u32x4_t{0u, 0u, 0u, 0u}; b[i] = a[i]; }
    else if (data == 1) { a[i] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; b[i] = a[i]; }
    else { a[i] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)}; b[i] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)}; }
  }
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < N_IT; it += 8) {
#pragma unroll
    for (int o = 0; o < 8; ++o) {              // compile-time operand rotation: no dynamic register indexing
#pragma unroll
      for (int i = 0; i < 16; ++i)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i + o) & 3]));      // volatile: program order, operands in registers
    }
    if (data == 2 && (it & 63) == 56) {          // keep the accumulators bounded: random walks, halved now and then
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = acc[i] * 0.5f;
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
  if (r == 12345.678f) sink[tid] = r;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
  (void)lane;
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
__global__ __launch_bounds__(512, 1) void mfma_stream32(int data, float* sink, long long* cyc) {
  const int tid = threadIdx.x;
  u32x4_t a[4], b[4];
  unsigned s = 12345u + (unsigned)tid * 7919u + blockIdx.x * 104729u;
  for (int i = 0; i < 4; ++i) {
    if (data == 0) { a[i] = u32x4_t{0u, 0u, 0u, 0u}; b[i] = a[i]; }
    else if (data == 1) { a[i] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; b[i] = a[i]; }
    else { a[i] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)}; b[i] = u32x4_t{rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s), rnd_bf16_pair(s)}; }
  }
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int it = 0; it < N_IT; it += 8) {
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[(i + o) & 3]));
    }
    if (data == 2 && (it & 63) == 56) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = acc[i] * 0.5f;
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
  if (r == 12345.678f) sink[tid] = r;
  if (blockIdx.x == 0 && tid == 0) cyc[0] = t1 - t0;
}

int main() {
  float* sink; long long* cyc;
  if (hipMalloc(&sink, 4096) != hipSuccess || hipMalloc(&cyc, 64) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"zeros", "constant 1.0", "pseudo-random bf16 in [-2, 2)"};
  for (int waves = 1; waves <= 4; waves *= 2)
    for (int data = 0; data < 3; ++data) {
      for (int rep = 0; rep < 2; ++rep) {          // first launch warms up
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_stream, dim3(256), dim3(256 * waves), 0, 0, data, sink, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      long long c = 0;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double nmfma = 16.0 * N_IT, flop = nmfma * 16384.0 * 256 * 4 * waves;
      printf("%d wave(s) per SIMD, %-30s  %8.1f us  %7.1f TF/s (%4.1f %% of 2.5 PF)  %.2f cycles per MFMA and wave  -> shader clock %.2f GHz\n", waves, names[data], ms * 1e3,
             flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e13, c / nmfma, c / (ms * 1e6));
    }
  for (int waves = 1; waves <= 2; waves *= 2)
    for (int data = 0; data < 3; ++data) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_stream32, dim3(256), dim3(256 * waves), 0, 0, data, sink, cyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
      }
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      long long c = 0;
      hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double nmfma = 8.0 * N_IT, flop = nmfma * 32768.0 * 256 * 4 * waves;
      printf("32x32x16: %d wave(s) per SIMD, %-30s  %8.1f us  %7.1f TF/s (%4.1f %% of 2.5 PF)  %.2f cycles per MFMA and wave  -> shader clock %.2f GHz\n", waves, names[data],
             ms * 1e3, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e13, c / nmfma, c / (ms * 1e6));
    }
  return 0;
}
