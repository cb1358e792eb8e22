// fp8 (OCP e4m3fn) path for the Low-Rank Transformer projections (BASELINE configs[4]):
//   asr_quant_fp8     bf16 / fp32 rows -> e4m3 bytes with ONE scale per tensor (amax / 448), two launches (amax, convert)
//   asr_gemm_nt_fp8   C[M,N] = sa sb sum_k A[m,k] B[n,k] (+ bias) (ReLU) on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales:
//                     the K = 128 block-scaled MFMA is the only fp8 matrix instruction of gfx950 that runs at twice the bf16 rate
//                     (the plain 16x16x32 fp8 form runs at the bf16 rate); E8M0 scale 127 = 2^0 turns it into a plain fp8 MFMA.
// Operand bytes are half of bf16: the r = 64 projections of the low-rank model are memory / latency bound (6400 x 64 x 512), which is
// where the halved traffic pays.  Tiles: 64 x 64 per 256-thread workgroup (4 waves as 2 x 2, 2 x 2 fragments each), K step 128 bytes
// staged through LDS with the same 16-byte-chunk XOR swizzle as the bf16 GEMMs; the last K step is zero filled in LDS, so any K
// that is a multiple of 16 works (K = 64: half of the one step is zeros).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

__device__ __forceinline__ unsigned f32_bits_abs(float v) { return __float_as_uint(v) & 0x7fffffffu; }

template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const T* __restrict__ x, int64_t ld, int M, int K, float* __restrict__ amax) {
  // non-negative floats order like their bit patterns: atomicMax on the uint view
  unsigned m = 0;
  const int64_t total = (int64_t)M * K;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / K, c = i - r * K;
    const unsigned b = f32_bits_abs(DT<T>::ld(x + r * ld + c));
    m = b > m ? b : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0 && m) atomicMax(reinterpret_cast<unsigned*>(amax), m);
}

// q[r, c] = e4m3(x[r, c] * 448 / amax) for c < K, 0 for K <= c < Kp (row stride ldq >= Kp, Kp = K rounded up to 16); scale[1] = amax / 448
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_kernel(const T* __restrict__ x, int64_t ld, int M, int K, int Kp,
                                                        uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
  const float amax = scale[0];
  const float mul = amax > 0.f ? 448.f / amax : 0.f;
  if (blockIdx.x == 0 && threadIdx.x == 0) scale[1] = amax > 0.f ? amax / 448.f : 0.f;
  const int q4 = Kp >> 2;                      // 4 outputs (one dword) per work item
  const int64_t total = (int64_t)M * q4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / q4;
    const int c = (int)(i - r * q4) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c + e < K ? DT<T>::ld(x + r * ld + c + e) * mul : 0.f;
    int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true);
    *reinterpret_cast<int*>(q + r * ldq + c) = p;
  }
}

struct Fp8Args {
  const uint8_t* A; const uint8_t* B; void* C; const float* bias; const float* sa; const float* sb;
  int64_t lda, ldb, ldc;
  int M, N, K, relu, tiles_n;
};

template <typename TO>
__global__ __launch_bounds__(256) void gemm_fp8_nt_kernel(Fp8Args p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 64 * 128];      // A tile | B tile: 64 rows x 128 bytes each
  unsigned char* sA = smem;
  unsigned char* sB = smem + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * 64, n0 = (tile % p.tiles_n) * 64;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += 128) {
    if (k0 > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, ch = c & 7;
      const int kc = k0 + ch * 16;
      uint4 va = make_uint4(0u, 0u, 0u, 0u), vb = make_uint4(0u, 0u, 0u, 0u);
      if (kc < p.K) {                           // K is a multiple of 16: whole chunks only
        if (m0 + row < p.M) va = *reinterpret_cast<const uint4*>(p.A + (int64_t)(m0 + row) * p.lda + kc);
        if (n0 + row < p.N) vb = *reinterpret_cast<const uint4*>(p.B + (int64_t)(n0 + row) * p.ldb + kc);
      }
      *reinterpret_cast<uint4*>(sA + row * 128 + ((ch ^ (row & 7)) << 4)) = va;
      *reinterpret_cast<uint4*>(sB + row * 128 + ((ch ^ (row & 7)) << 4)) = vb;
    }
    __syncthreads();
    // fragment of v_mfma_scale_f32_16x16x128_f8f6f4: lane (row lr, k block g) holds 32 consecutive bytes = chunks 2g, 2g+1 of its row
    i32x8_t fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ra = wm * 32 + i * 16 + lr, rb = wn * 32 + i * 16 + lr;
      const uint4 a0 = *reinterpret_cast<const uint4*>(sA + ra * 128 + (((2 * g) ^ (ra & 7)) << 4));
      const uint4 a1 = *reinterpret_cast<const uint4*>(sA + ra * 128 + (((2 * g + 1) ^ (ra & 7)) << 4));
      const uint4 b0 = *reinterpret_cast<const uint4*>(sB + rb * 128 + (((2 * g) ^ (rb & 7)) << 4));
      const uint4 b1 = *reinterpret_cast<const uint4*>(sB + rb * 128 + (((2 * g + 1) ^ (rb & 7)) << 4));
      fa[i] = i32x8_t{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
      fb[i] = i32x8_t{(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)     // formats 0 / 0 = e4m3 x e4m3; block scales 0x7F = 2^0
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], acc[i][j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  }
  // D[row = 4 g + r of the A fragment][col = lr of the B fragment]
  const float s = p.sa[1] * p.sb[1];
  TO* C = static_cast<TO*>(p.C);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int gn = n0 + wn * 32 + j * 16 + lr;
      const float bv = (p.bias && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + wm * 32 + i * 16 + g * 4 + r;
        if (gm < p.M && gn < p.N) {
          float v = acc[i][j][r] * s + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          DT<TO>::st(C + (int64_t)gm * p.ldc + gn, v);
        }
      }
    }
}

}  // namespace

extern "C" int asr_quant_fp8(const void* x, int64_t ld, int M, int K, int dtype, uint8_t* q, int64_t ldq, float* scale, hipStream_t s) {
  ASR_CHECK_ARG(x && q && scale && M >= 0 && K > 0 && ld >= K);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int Kp = (K + 15) / 16 * 16;
  ASR_CHECK_ARG(ldq >= Kp && ldq % 16 == 0 && aligned16(q));
  if (M == 0) return ASR_OK;
  if (hipMemsetAsync(scale, 0, 2 * sizeof(float), s) != hipSuccess) return ASR_ERUNTIME;
  const int64_t total = (int64_t)M * K;
  const unsigned grid = (unsigned)(total / 256 + 1 < 2048 ? total / 256 + 1 : 2048);
  if (dtype == ASR_F32) {
    hipLaunchKernelGGL((fp8_amax_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, ld, M, K, scale);
    hipLaunchKernelGGL((fp8_quant_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, ld, M, K, Kp, q, ldq, scale);
  } else {
    hipLaunchKernelGGL((fp8_amax_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, M, K, scale);
    hipLaunchKernelGGL((fp8_quant_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, M, K, Kp, q, ldq, scale);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_gemm_nt_fp8(const uint8_t* A, int64_t lda, const float* scale_a, const uint8_t* B, int64_t ldb, const float* scale_b,
                               void* C, int64_t ldc, const float* bias, int M, int N, int K, int relu, int out_dtype, hipStream_t s) {
  ASR_CHECK_ARG(A && B && C && scale_a && scale_b && M >= 0 && N >= 0 && K > 0);
  ASR_CHECK_ARG(out_dtype == ASR_F32 || out_dtype == ASR_BF16);
  if (K % 16 != 0 || lda % 16 != 0 || ldb % 16 != 0 || lda < K || ldb < K || !aligned16(A) || !aligned16(B)) return ASR_EUNSUPPORTED;
  if (M == 0 || N == 0) return ASR_OK;
  Fp8Args p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.sa = scale_a; p.sb = scale_b;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.relu = relu;
  p.tiles_n = (N + 63) / 64;
  const unsigned grid = (unsigned)(((M + 63) / 64) * p.tiles_n);
  AsrProfScope prof(ASR_OP_GEMM, s);
  if (out_dtype == ASR_F32) hipLaunchKernelGGL((gemm_fp8_nt_kernel<float>), dim3(grid), dim3(256), 0, s, p);
  else hipLaunchKernelGGL((gemm_fp8_nt_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
