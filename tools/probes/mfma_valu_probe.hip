// Do MFMA and ordinary VALU instructions overlap on a gfx950 SIMD?  Chip-level, host-timed.
//   variant K: every wave runs 8 independent v_mfma_f32_16x16x32_bf16 + K independent v_add_u32 per MFMA, per iteration
//   variant 100+K: waves alternate roles -- even waves pure MFMA, odd waves pure VALU (8*K v_add per iteration)
// Build+run: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_probe.hip -o /tmp/mvp && /tmp/mvp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
#define N_IT 4096
template <int K, bool SPLIT> __global__ __launch_bounds__(512) void k(float* out, uint32_t seed) {
  f32x4_t acc[8];
  u32x4_t a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 11};
  uint32_t v[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; v[i] = threadIdx.x + i; }
  if (!SPLIT) {
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[(i + j) & 7]) : "v"(seed));
      }
    }
  } else if (((threadIdx.x >> 6) & 4) == 0) {      // waves w and w+4 share a SIMD: waves 0-3 MFMA only ...
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  } else {                                         // ... waves 4-7 VALU only (8 K v_add per iteration)
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[(i + j) & 7]) : "v"(seed));
      }
    }
  }
  float s = 0.f; uint32_t t = 0;
  for (int i = 0; i < 8; ++i) { s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]; t += v[i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)t;
}
template <int K, bool SPLIT> void run(float* out, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<K, SPLIT><<<256, 512>>>(out, 1); hipDeviceSynchronize();
  hipEventRecord(e0); k<K, SPLIT><<<256, 512>>>(out, 1); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_waves = SPLIT ? 4 : 8;
  const double flops = 256.0 * mfma_waves * N_IT * 8 * 2.0 * 16 * 16 * 32;
  const double valu = 256.0 * (SPLIT ? 4 : 8) * (double)N_IT * 8 * K;     // wave-instructions
  printf("%-34s %8.3f ms  %7.1f TF/s  (%.1f%% of 2500)   VALU %.2f Ginstr/s/SIMD-equivalent cycles/instr %.2f\n", name, ms,
         flops / ms / 1e9, flops / ms / 1e9 / 25.0, valu / ms / 1e6 / 1024.0, valu > 0 ? ms * 1e-3 * 2.4e9 * 1024.0 / valu : 0.0);
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run<0, false>(out, "MFMA only, 2 waves/SIMD");
  run<1, false>(out, "MFMA + 1 VALU each");
  run<2, false>(out, "MFMA + 2 VALU each");
  run<3, false>(out, "MFMA + 3 VALU each");
  run<4, false>(out, "MFMA + 4 VALU each");
  run<6, false>(out, "MFMA + 6 VALU each");
  run<0, true>(out, "split: wave A MFMA, wave B idle");
  run<2, true>(out, "split: A MFMA, B 2 VALU per slot");
  run<4, true>(out, "split: A MFMA, B 4 VALU per slot");
  run<8, true>(out, "split: A MFMA, B 8 VALU per slot");
  return 0;
}
