// VALU issue-rate probe for gfx950: time per wave64 instruction for the ops the attention softmax / dropout path uses,
// relative to v_fma_f32.  8 independent dependency chains per lane, inline asm so that nothing is folded.
// Build+run: hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate_probe.hip -o /tmp/valu_probe && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 2048
#define OP1(INS) asm volatile(INS " %0, %0, %1" : "+v"(a[i]) : "v"(c));
template <int OP> __global__ void k(uint32_t* out, uint32_t seed, long long* cyc) {
  uint32_t a[8]; uint32_t c = seed * 0x9E3779B1u + 12345u;
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 8 + i;
  long long t0 = clock64();
  for (int it = 0; it < N_IT; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) OP1("v_mul_lo_u32")
      if (OP == 1) OP1("v_add_u32")
      if (OP == 2) OP1("v_mul_u32_u24")
      if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 5) OP1("v_xor_b32")
      if (OP == 6) OP1("v_mul_hi_u32")
      if (OP == 7) OP1("v_max_f32")
      if (OP == 8) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 9) OP1("v_lshrrev_b32")
    }
    if (OP == 10) {   // packed fp32 fma on 4 register pairs = 8 element-ops per lane
      uint64_t* p = reinterpret_cast<uint64_t*>(a);
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i]));
    }
  }
  long long t1 = clock64();
  uint32_t s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
  uint32_t* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 64 * 8); hipMemset(cyc, 0, 64 * 8);
  const int T = 256;   // 4 waves on one CU = one wave per SIMD; T = 512: two waves per SIMD
  for (int threads = 256; threads <= 1024; threads *= 2) {
    k<0><<<1, threads>>>(out, 1, cyc); k<1><<<1, threads>>>(out, 1, cyc); k<2><<<1, threads>>>(out, 1, cyc); k<3><<<1, threads>>>(out, 1, cyc);
    k<4><<<1, threads>>>(out, 1, cyc); k<5><<<1, threads>>>(out, 1, cyc); k<6><<<1, threads>>>(out, 1, cyc); k<7><<<1, threads>>>(out, 1, cyc);
    k<8><<<1, threads>>>(out, 1, cyc); k<9><<<1, threads>>>(out, 1, cyc); k<10><<<1, threads>>>(out, 1, cyc);
    hipDeviceSynchronize();
    long long h[11]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[11] = {"v_mul_lo_u32", "v_add_u32", "v_mul_u32_u24", "v_exp_f32", "v_fma_f32", "v_xor_b32", "v_mul_hi_u32", "v_max_f32",
                          "v_cvt_pk_bf16_f32", "v_lshrrev_b32", "v_pk_fma_f32 (x4)"};
    printf("-- %d waves per SIMD\n", threads / 256);
    for (int i = 0; i < 11; ++i)
      printf("%-20s %7.3f ticks per instruction   (%.2f x v_fma_f32)\n", nm[i], (double)h[i] / (N_IT * (i == 10 ? 4.0 : 8.0)),
             (double)h[i] / (i == 10 ? 0.5 : 1.0) / (double)h[4]);
  }
  (void)T;
  // chip-level throughput: 256 CUs x 8 blocks x 256 threads = 8 waves per SIMD, host-timed
  printf("-- throughput, all CUs, 8 waves per SIMD (host timed; cycles assume 2.4 GHz)\n");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(OPN, NAME, PER)                                                                             \
  { k<OPN><<<2048, 256>>>(out, 1, cyc); hipDeviceSynchronize(); hipEventRecord(e0);                     \
    k<OPN><<<2048, 256>>>(out, 1, cyc); hipEventRecord(e1); hipEventSynchronize(e1);                    \
    float ms; hipEventElapsedTime(&ms, e0, e1);                                                         \
    double instr_per_simd = 2048.0 * 4 / (256 * 4) * N_IT * PER;                                        \
    printf("%-20s %7.3f cycles per wave64 instruction per SIMD\n", NAME, ms * 1e-3 * 2.4e9 / instr_per_simd); }
  RUN(0, "v_mul_lo_u32", 8) RUN(1, "v_add_u32", 8) RUN(2, "v_mul_u32_u24", 8) RUN(3, "v_exp_f32", 8) RUN(4, "v_fma_f32", 8)
  RUN(5, "v_xor_b32", 8) RUN(6, "v_mul_hi_u32", 8) RUN(7, "v_max_f32", 8) RUN(8, "v_cvt_pk_bf16_f32", 8) RUN(9, "v_lshrrev_b32", 8)
  RUN(10, "v_pk_fma_f32", 4)
  return 0;
}
