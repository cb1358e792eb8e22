// Development harness for the grouped weight-gradient contraction (asr_gemm_tn_grouped; csrc/gemm.hip tn256_body / tn256r_body): the linear
// layers of configs[1] (4 + 4 layers, d_model 512, inner 2048, 6400 encoder / 3200 decoder rows) in the two launches the training step
// issues, the equal-piece form (tuning TN_GROUP_TILE = 1; until round 6 also round 3's loop, removed since) against round 5's (operand reads of the next stage under the MFMAs): the same bits (the
// accumulation order per element is the same), interleaved timing.  `big` adds configs[3]'s row counts (12 720 / 1600).
// Build:  hipcc -O2 tools/tn_grouped_test.cpp -o tools/bin/tn_grouped_test -Iinclude -Lend2end-asr-pytorch_amd/asr_hip -lasr_hip \
//               -Wl,-rpath,'$ORIGIN/../../end2end-asr-pytorch_amd/asr_hip'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "asr_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { printf("asr error %d (%s) at %s:%d\n", r_, asr_strerror(r_), __FILE__, __LINE__); exit(2); } } while (0)

static uint32_t rng_state = 777u;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }
static uint16_t f2bf_rne(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }

struct Prob { int M, N, K; void* dy; void* x; float* dw; float* db; std::vector<uint16_t> hy, hx; };
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const bool big = argc > 1 && strcmp(argv[1], "big") == 0;
  const int Me = big ? 12720 : 6400, Md = big ? 1600 : 3200, LE = big ? 12 : 4, LD = big ? 6 : 4;
  std::vector<Prob> ps;
  auto add = [&](int M, int N, int K) { ps.push_back({M, N, K, nullptr, nullptr, nullptr, nullptr, {}, {}}); };
  // backward order: decoder (last layer first), vocabulary projection first of all; then the encoder
  add(Md, big ? 32 : 4416, 512);
  for (int l = 0; l < LD; ++l) { add(Md, 512, 2048); add(Md, 2048, 512); add(Md, 512, 512); add(Me, 1024, 512); add(Md, 512, 512); add(Md, 512, 512); add(Md, 1536, 512); }
  for (int l = 0; l < LE; ++l) { add(Me, 512, 2048); add(Me, 2048, 512); add(Me, 512, 512); add(Me, 1536, 512); }
  add(Me, 512, big ? 5120 : 2560);
  double flops = 0;
  for (auto& p : ps) {
    std::vector<uint16_t> h((size_t)p.M * p.N), g((size_t)p.M * p.K);
    for (auto& v : h) v = f2bf_rne((float)((int)(rnd() % 2001) - 1000) * 1e-3f);
    for (auto& v : g) v = f2bf_rne((float)((int)(rnd() % 2001) - 1000) * 1e-3f);
    CK(hipMalloc(&p.dy, h.size() * 2)); CK(hipMalloc(&p.x, g.size() * 2));
    CK(hipMalloc(&p.dw, (size_t)p.N * p.K * 4)); CK(hipMalloc(&p.db, (size_t)p.N * 4));
    CK(hipMemcpy(p.dy, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p.x, g.data(), g.size() * 2, hipMemcpyHostToDevice));
    flops += 2.0 * p.M * p.N * p.K;
    p.hy = std::move(h); p.hx = std::move(g);
  }
  printf("%zu problems, %.1f GFLOP per pass (%s)\n", ps.size(), flops * 1e-9, big ? "configs[3] rows" : "configs[1] rows");
  size_t group = 32;
  auto pass = [&]() {
    const size_t G = group;          // layers per launch: 32 with round 3's kernel (its limit), 48 with round 5's
    for (size_t base = 0; base < ps.size(); base += G) {
      const int n = (int)std::min<size_t>(G, ps.size() - base);
      const void* dy[48]; const void* x[48]; float* dw[48]; float* db[48]; int64_t ldy[48], ldx[48], ldw[48]; int M[48], N[48], K[48];
      for (int i = 0; i < n; ++i) {
        const Prob& p = ps[base + i];
        dy[i] = p.dy; x[i] = p.x; dw[i] = p.dw; db[i] = p.db; ldy[i] = p.N; ldx[i] = p.K; ldw[i] = p.K; M[i] = p.M; N[i] = p.N; K[i] = p.K;
      }
      AK(asr_gemm_tn_grouped(n, dy, ldy, x, ldx, dw, ldw, db, M, N, K, ASR_BF16, nullptr));
    }
  };
  // ---- the same bits
  std::vector<std::vector<float>> out[2];
  for (int rot = 0; rot < 2; ++rot) {
    AK(asr_set_tuning("TN_GROUP_TILE", rot ? 0 : 1));      // 0: the planner (whole blocks for the headline's list); 1: equal pieces, shared blocks meet in atomics
    group = rot ? 48 : 32;
    for (auto& p : ps) { CK(hipMemset(p.dw, 0, (size_t)p.N * p.K * 4)); CK(hipMemset(p.db, 0, (size_t)p.N * 4)); }
    pass();
    CK(hipDeviceSynchronize());
    for (auto& p : ps) {
      std::vector<float> h((size_t)p.N * p.K + p.N);
      CK(hipMemcpy(h.data(), p.dw, (size_t)p.N * p.K * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h.data() + (size_t)p.N * p.K, p.db, (size_t)p.N * 4, hipMemcpyDeviceToHost));
      out[rot].push_back(std::move(h));
    }
  }
  // against the definition: 48 elements and 4 bias sums per problem in fp64 on the host, error within 1e-5 of the sum of |products|
  size_t bad_ref[2] = {0, 0};
  double worst[2] = {0, 0};
  for (size_t i = 0; i < ps.size(); ++i) {
    const Prob& p = ps[i];
    for (int t = 0; t < 52; ++t) {
      const int n = (int)(rnd() % (uint32_t)p.N), k = (int)(rnd() % (uint32_t)p.K);
      double ref = 0, mag = 0;
      for (int m = 0; m < p.M; ++m) {
        const double a = bf2f(p.hy[(size_t)m * p.N + n]), b = t < 48 ? (double)bf2f(p.hx[(size_t)m * p.K + k]) : 1.0;
        ref += a * b; mag += fabs(a * b);
      }
      for (int rot = 0; rot < 2; ++rot) {
        const double got = t < 48 ? out[rot][i][(size_t)n * p.K + k] : out[rot][i][(size_t)p.N * p.K + n];
        const double e = fabs(got - ref) / (mag + 1e-30);
        worst[rot] = std::max(worst[rot], e);
        if (e > 1e-5) ++bad_ref[rot];
      }
    }
  }
  printf("  against fp64 on the host (52 samples per problem): round 3 %zu bad (worst %.2e of sum |products|), round 5 %zu bad (worst %.2e)\n",
         bad_ref[0], worst[0], bad_ref[1], worst[1]);
  size_t bad = 0, total = 0, bad_rel = 0;
  for (size_t i = 0; i < ps.size(); ++i)
    for (size_t e = 0; e < out[0][i].size(); ++e) {
      ++total;
      if (memcmp(&out[0][i][e], &out[1][i][e], 4) != 0) {
        ++bad;
        const float a = out[0][i][e], b = out[1][i][e];
        if (!(fabsf(a - b) <= 1e-4f * std::max(1.f, fabsf(a)))) ++bad_rel;
      }
    }
  // round 3's blocks shared between workgroups are summed with fp32 atomics in arrival order: those elements may differ in the last bits
  printf("  %zu of %zu elements differ in bits between the two kernels (round 3 sums shared blocks atomically), %zu beyond 1e-4 relative\n", bad, total, bad_rel);
  // ---- interleaved timing
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> t[2];
  for (int round = 0; round < 7; ++round)
    for (int rot = 0; rot < 2; ++rot) {
      AK(asr_set_tuning("TN_GROUP_TILE", rot ? 0 : 1));      // 0: the planner (whole blocks for the headline's list); 1: equal pieces, shared blocks meet in atomics
    group = rot ? 48 : 32;
      pass();
      const int iters = 6;
      CK(hipEventRecord(e0, nullptr));
      for (int i = 0; i < iters; ++i) pass();
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      t[rot].push_back(ms * 1e3 / iters);
    }
  for (int rot = 0; rot < 2; ++rot) {
    std::sort(t[rot].begin(), t[rot].end());
    printf("  %-40s median %7.1f us per pass (min %7.1f, max %7.1f) = %6.1f TF/s\n", rot ? "whole blocks, reads under MFMAs (r5)" : "equal pieces, atomics (round 3)",
           t[rot][3], t[rot][0], t[rot][6], flops / t[rot][3] * 1e-6);
  }
  AK(asr_clear_tuning("TN_GROUP_TILE"));
  const bool fail = bad_ref[0] || bad_ref[1];
  printf(fail ? "FAILED\n" : "OK\n");
  return fail ? 1 : 0;
}
