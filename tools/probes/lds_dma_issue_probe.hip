// What does a wave pay to ISSUE LDS-DMA pieces (global_load_lds_dwordx4), and does the M0 write between two pieces serialise them?
// (round 6: the grouped weight gradient's section timing showed ~1 100 - 1 500 cycles between a stage's barrier and its next one for 24 MFMAs
// and four DMA pieces per wave; tools/probes/vmem_issue_probe.hip had a streaming wave at one piece per ~340 cycles.)
//   rate kernel: one 4-wave workgroup per CU, every wave issues pieces of 1 KB from an L2-resident buffer, <= 8 in flight, and times them:
//     mode 0  the tree's idiom: save M0, set M0, s_nop, DMA, restore M0 -- per piece
//     mode 1  M0 set once before the loop (every piece to the same LDS slot)
//     mode 2  one M0 write per piece (a different slot each), no save / restore
//     mode 3  one M0 write per FOUR pieces; the four pieces differ by the instruction's immediate offset (0, 1024, 2048, 3072)
//   semantics kernel: where does a piece with `offset:1024` land, and what does it read?  (global: src + offset?  LDS: M0 + offset?)
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/probes/lds_dma_issue_probe.hip -o tools/bin/lds_dma_issue_probe && tools/bin/lds_dma_issue_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr int N_PIECES = 8192;

template <int INFLIGHT>
__global__ __launch_bounds__(1024, 1) void rate(int mode, const u32x4_t* __restrict__ src, long long* out) {
  __shared__ u32x4_t lds[4096];       // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32x4_t* p = src + (size_t)blockIdx.x * 4096 * 64 + lane;          // 4 MB per workgroup, walked in 1 KB steps
  const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(__attribute__((address_space(3))) u32x4_t*)lds + (unsigned)(wave & 3) * 16384u));
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
  if (mode == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base));
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
  for (int k = 0; k < N_PIECES; k += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u32x4_t* a = p + ((k + u) & 4095) * 64;
      const unsigned slot = base + (unsigned)(u & 3) * 1024u;
      if (mode == 0) {
        unsigned kp;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0" : "=&s"(kp) : "s"(slot), "v"(a) : "memory");
      } else if (mode == 1) {
        asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(a) : "memory");
      } else if (mode == 2) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(slot), "v"(a) : "memory");
      } else {
        if ((u & 3) == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(base));
        // the immediate moves the GLOBAL address too (semantics kernel): compensate in the register
        const char* ac = reinterpret_cast<const char*>(a);
        if ((u & 3) == 0) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(ac) : "memory");
        if ((u & 3) == 1) asm volatile("global_load_lds_dwordx4 %0, off offset:1024" ::"v"(ac - 1024) : "memory");
        if ((u & 3) == 2) asm volatile("global_load_lds_dwordx4 %0, off offset:2048" ::"v"(ac - 2048) : "memory");
        if ((u & 3) == 3) asm volatile("global_load_lds_dwordx4 %0, off offset:3072" ::"v"(ac - 3072) : "memory");
      }
    }
    if (INFLIGHT == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (INFLIGHT == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  asm volatile("s_mov_b32 m0, %0" ::"s"(keep));
  if (lds[tid].x == 0xdeadbeefu) out[15] = 1;
  if (lane == 0 && blockIdx.x == 0 && wave < 8) out[wave] = t1 - t0;
}

__global__ __launch_bounds__(64) void semantics(const unsigned* __restrict__ src, unsigned* out) {
  __shared__ unsigned lds[4096];      // 16 KB
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = 0xffffffffu;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)lds;
  const char* a = reinterpret_cast<const char*>(src) + lane * 16;          // dwords 4 lane .. 4 lane + 3
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off offset:1024\n\ts_waitcnt vmcnt(0)\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(base), "v"(a) : "memory");
  __syncthreads();
  for (int i = lane; i < 4096; i += 64) out[i] = lds[i];
}

int main() {
  u32x4_t* src; long long* out; unsigned* sem; unsigned* semsrc;
  const size_t bytes = (size_t)256 * 4096 * 1024;
  if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 16 * 8) != hipSuccess || hipMalloc(&sem, 4096 * 4) != hipSuccess || hipMalloc(&semsrc, 65536) != hipSuccess) return 1;
  (void)hipMemset(src, 1, bytes);
  {
    unsigned h[16384];
    for (int i = 0; i < 16384; ++i) h[i] = (unsigned)i;
    (void)hipMemcpy(semsrc, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, semsrc, sem);
    unsigned o[4096];
    (void)hipMemcpy(o, sem, sizeof(o), hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 4096; ++i) if (o[i] != 0xffffffffu) { first = i; break; }
    printf("semantics: `global_load_lds_dwordx4 v, off offset:1024` with M0 = LDS base, lane address = src + 16 lane:\n");
    if (first < 0) printf("  nothing landed in the first 16 KB\n");
    else printf("  first written LDS dword %d (byte %d) holds source dword %u (source byte %u); LDS dword %d holds %u\n", first, first * 4, o[first], o[first] * 4, first + 4, o[first + 4]);
  }
  const char* names[4] = {"save / set / DMA / restore M0 per piece (the tree's idiom)", "M0 set once, same slot", "one M0 write per piece", "one M0 write per four pieces (immediate offsets)"};
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipMemset(out, 0, 16 * 8);
      hipLaunchKernelGGL(rate<8>, dim3(256), dim3(256), 0, 0, mode, src, out);
      if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "mode %d failed\n", mode); return 2; }
    }
    long long h[16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d (%s): cycles per piece, waves 0-3: %.1f %.1f %.1f %.1f\n", mode, names[mode], h[0] / (double)N_PIECES, h[1] / (double)N_PIECES,
           h[2] / (double)N_PIECES, h[3] / (double)N_PIECES);
  }
  // is that the path or the latency?  mode 1 (no M0 traffic) with more waves per CU and more pieces in flight per wave
  for (int waves = 4; waves <= 16; waves *= 2)
    for (int depth = 8; depth <= 48; depth = depth == 8 ? 24 : (depth == 24 ? 48 : 99)) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(out, 0, 16 * 8);
        if (depth == 8) hipLaunchKernelGGL(rate<8>, dim3(256), dim3(64 * waves), 0, 0, 1, src, out);
        else if (depth == 24) hipLaunchKernelGGL(rate<24>, dim3(256), dim3(64 * waves), 0, 0, 1, src, out);
        else hipLaunchKernelGGL(rate<48>, dim3(256), dim3(64 * waves), 0, 0, 1, src, out);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "sweep failed\n"); return 2; }
      }
      long long h[16];
      (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
      const double cyc = h[0] / (double)N_PIECES;
      printf("%2d waves per CU, <= %2d pieces in flight per wave: %.1f cycles per piece and wave = %.1f bytes per clock and CU\n", waves, depth, cyc, waves * 1024.0 / cyc);
    }
  return 0;
}
