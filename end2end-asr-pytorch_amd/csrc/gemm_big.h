// Eight-wave linear-layer GEMMs (csrc/gemm_big.hip): one workgroup of 512 threads per 256 x 256 or 128 x 128 output block.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct BigGemmArgs {
  const void* A; const void* B; void* C;        // A (M, K) bf16 row-major; B: NT (N, K) bf16 row-major, NN (K, N) bf16 row-major
  const float* bias;                            // (N) fp32 or null
  const void* mask;                             // NN only: (M, N) bf16 laid out like C; C = 0 where mask <= 0 (ReLU backward)
  int64_t lda, ldb, ldc;                        // elements
  int M, N, K;
  float alpha;
  int relu, accumulate, out_f32;
  // NN only (asr_gemm_nn_rowdot): dot_out[(b H + h) T + q] = sum_{d < 64} C[b T + q][64 h + d] * O[b T + q][64 h + d] with the ROUNDED C;
  // O = dot_o32 (fp32, row stride N) if given, else dot_o (bf16, row stride N).  N = 64 H.
  const void* dot_o; const float* dot_o32; float* dot_out; int dot_T, dot_H;
  // NN only (asr_gemm_nn_poolbwd): C is NOT (M, N) but the un-pooled NHWC gradient (M / W2, 2 H2, 2 W2, pool_C); row m = (b, w2), column
  // n = (h2, c); every 16-byte piece goes through its 8 selection bytes pool_code[(m H2 + h2) pool_C + c] to the four window positions
  const uint8_t* pool_code; int pool_H2, pool_W2, pool_C;
};

// -> true when the shape / layout is taken (launched on `stream`), false when the caller should use the four-wave kernels.
bool asr_gemm_big_nt(const BigGemmArgs& p, hipStream_t stream);
bool asr_gemm_big_nn(const BigGemmArgs& p, hipStream_t stream);
