// Grid-wide barrier cost on gfx950 (8 XCDs, one L2 each): G co-resident workgroups, a monotonically increasing counter in
// global memory (agent-scope atomic add + spin on an agent-scope load), and the same with a 2 KB "activation" exchange through
// memory between the phases (each workgroup writes a row with write-through stores, after the barrier reads its neighbour's row
// with loads that bypass the non-coherent L2s) -- the pattern a persistent decode-step kernel would use between its phases.
// Build+run: hipcc --offload-arch=gfx950 -O3 tools/probes/grid_barrier_probe.hip -o /tmp/gbar && /tmp/gbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

template <int MODE> __global__ __launch_bounds__(256) void k(unsigned* ctr, float* buf, int iters, float* out, int* bad) {
  const int G = gridDim.x, b = blockIdx.x;
  float acc = 0.f;
  int wrong = 0;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {      // exchange: 256 floats per workgroup, coherent stores / loads (sc0 sc1 = system scope: bypass L2)
      float v = (float)(it * 1000 + b) + threadIdx.x * 0.001f;
      __hip_atomic_store(buf + (it & 1) * G * 256 + b * 256 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    grid_barrier(ctr, (unsigned)(it + 1) * G);
    if (MODE == 1) {
      const int nb = (b + 37) % G;
      const float v = __hip_atomic_load(buf + (it & 1) * G * 256 + nb * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float want = (float)(it * 1000 + nb) + threadIdx.x * 0.001f;
      if (v != want) ++wrong;
      acc += v;
    }
  }
  if (wrong) atomicAdd(bad, wrong);
  if (threadIdx.x == 0) out[b] = acc;
}

int main() {
  unsigned* ctr; float *buf, *out; int* bad;
  hipMalloc(&ctr, 4); hipMalloc(&buf, 2 * 1024 * 256 * 4); hipMalloc(&out, 4096); hipMalloc(&bad, 4);
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode)
    for (int G : {16, 32, 64, 128, 256, 512}) {
      float best = 1e30f;
      int hbad = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, ctr, buf, iters, out, bad);
        else hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, ctr, buf, iters, out, bad);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
      }
      printf("%s G=%3d: %.2f us per barrier  (mismatches %d)\n", mode ? "barrier + 1 KB exchange" : "barrier only          ", G, best * 1e3f / iters, hbad);
    }
  return 0;
}
