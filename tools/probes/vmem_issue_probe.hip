// What does a vector-memory instruction cost the OTHER wave of its SIMD?  (round 5: conv_ws.hip's memory instructions could not be hidden
// behind MFMAs by any placement)  One 8-wave workgroup per CU: waves 0-3 (one per SIMD) issue a stream of independent
// v_mfma_f32_16x16x32_bf16 and time it; waves 4-7 -- wave w + 4 shares its SIMD with wave w -- do one of
//   mode 0  nothing
//   mode 1  all four stream LDS-DMA pieces (global_load_lds_dwordx4, 1 KB per instruction, <= 8 in flight) from an L2-resident buffer
//   mode 3  only wave 4 does (SIMD of wave 0): waves 1-3 are the control
//   mode 4  only wave 4, global_store_dwordx4
//   mode 5  only wave 4, a VALU stream (v_fma_f32) -- the cost of a co-resident wave that issues plain vector instructions
// Output per mode: cycles per MFMA of each MFMA wave of workgroup 0 and the partner's instructions per MFMA.
// Build + run: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/probes/vmem_issue_probe.hip -o tools/bin/vmem_issue_probe && tools/bin/vmem_issue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int N_IT = 4096;      // x 16 MFMAs per MFMA wave

__global__ __launch_bounds__(512, 1) void probe(int mode, const u32x4_t* __restrict__ src, u32x4_t* __restrict__ dst, long long* out, volatile int* done_flag) {
  __shared__ u32x4_t lds[512];
  __shared__ int stop;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) stop = 0;
  __syncthreads();
  if (wave < 4) {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    u32x4_t a = {0x3f803f80u + (unsigned)lane, 1u, 2u, 3u}, b = {0x3f803f80u, 5u, 6u, 7u};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int it = 0; it < N_IT; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[i], 0, 0, 0);
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0];
    if (s == 12345.f) dst[tid] = a;
    if (lane == 0) {
      if (blockIdx.x == 0) out[wave] = t1 - t0;
      atomicAdd(&stop, 1);                  // the partner waves run until all four MFMA waves are done
    }
    return;
  }
  const bool active = mode == 1 || ((mode >= 3) && wave == 4);
  if (mode == 0 || !active) return;
  long long n = 0;
  const u32x4_t* p = src + (size_t)blockIdx.x * 4096 * 64 + lane;          // 4 MB per workgroup, walked in 1 KB steps (L2 resident after the first pass)
  u32x4_t* q = dst + (size_t)blockIdx.x * 4096 * 64 + lane;
  u32x4_t v = {1u, 2u, 3u, 4u};
  float f = 1.f;
  int k = 0;
  while (*(volatile int*)&stop < 4) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = ((k + u) & 4095) * 64;
      if (mode == 1 || mode == 3) {
        unsigned keep;
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4_t*)lds + (unsigned)(wave - 4) * 1024u));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "s"(base), "v"(p + idx) : "memory");
      } else if (mode == 4) {
        asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(q + idx), "v"(v) : "memory");
      } else {
        asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f));
      }
    }
    if (mode != 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    k += 8; n += 8;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (f == 12345.f) dst[tid] = v;
  if (lane == 0 && blockIdx.x == 0) out[8 + (wave - 4)] = n;
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  u32x4_t *src, *dst; long long* out; int* flag;
  const size_t bytes = (size_t)256 * 4096 * 1024;      // 1 GiB: 4 MB per workgroup
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&dst, bytes) != hipSuccess || hipMalloc(&out, 16 * 8) != hipSuccess || hipMalloc(&flag, 4) != hipSuccess) return 1;
  (void)hipMemset(src, 1, bytes);
  const char* names[6] = {"partner idle", "all four partners: global_load_lds_dwordx4", "(unused)", "wave 4 only: global_load_lds_dwordx4",
                          "wave 4 only: global_store_dwordx4", "wave 4 only: v_fma_f32 stream"};
  for (int mode = 0; mode < 6; ++mode) {
    if ((only >= 0 && mode != only) || mode == 2) continue;
    (void)hipMemset(out, 0, 16 * 8);
    hipLaunchKernelGGL(probe, dim3(256), dim3(512), 0, 0, mode, src, dst, out, flag);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "mode %d failed\n", mode); return 2; }
    long long h[16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = 16.0 * N_IT;
    printf("mode %d (%s): cycles per MFMA, waves 0-3: %.2f %.2f %.2f %.2f;  partner instructions per MFMA of its SIMD: %.3f %.3f %.3f %.3f\n", mode, names[mode],
           h[0] / nm, h[1] / nm, h[2] / nm, h[3] / nm, h[8] / nm, h[9] / nm, h[10] / nm, h[11] / nm);
    if (mode >= 2 && h[8] > 0)
      printf("        -> wave 0 lost %.1f cycles per partner instruction (control waves 1-3: %.2f cycles per MFMA)\n",
             (h[0] - (h[1] + h[2] + h[3]) / 3.0) / (double)h[8], (h[1] + h[2] + h[3]) / 3.0 / nm);
  }
  return 0;
}
