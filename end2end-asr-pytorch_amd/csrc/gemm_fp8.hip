// fp8 (OCP e4m3fn) path for the Low-Rank Transformer projections (BASELINE configs[4]):
//   asr_quant_fp8     bf16 / fp32 rows -> e4m3 bytes with one scale PER ROW (row amax / 448): one launch, a wave per row, no atomics
//   asr_gemm_nt_fp8   C[m,n] = sa[m] sb[n] sum_k A[m,k] B[n,k] (+ bias) (ReLU) on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block
//                     scales: the K = 128 block-scaled MFMA is the only fp8 matrix instruction of gfx950 that runs at twice the bf16
//                     rate (the plain 16x16x32 fp8 form runs at the bf16 rate); E8M0 scale 127 = 2^0 makes it a plain fp8 MFMA.
// The r = 64 projections of the low-rank model are latency / memory bound (6400 x 64 x 512: 0.4 GFLOP over 3.3 MB): the kernel has
// no LDS stage and no barrier -- a lane's operand fragment (32 consecutive k bytes of one row) is two 16-byte global loads, the
// small operand comes from L2 -- waves are independent (a 32 x 32 output block each), K steps of 128 bytes, zero fill past K.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// one wave per row: amax over the row, then q[r, c] = e4m3(x[r, c] * 448 / amax) for c < K, 0 for K <= c < Kp; scale[r] = amax / 448
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const T* __restrict__ x, int64_t ld, int M, int K, int Kp,
                                                             uint8_t* __restrict__ q, int64_t ldq, float* __restrict__ scale) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const T* xr = x + (int64_t)row * ld;
  float m = 0.f;
  for (int c = lane * 4; c < K; c += 256)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < K) m = fmaxf(m, fabsf(DT<T>::ld(xr + c + e)));
  m = wave_max(m);
  const float mul = m > 0.f ? 448.f / m : 0.f;
  if (lane == 0) scale[row] = m > 0.f ? m / 448.f : 0.f;
  uint8_t* qr = q + (int64_t)row * ldq;
  for (int c = lane * 4; c < Kp; c += 256) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = c + e < K ? DT<T>::ld(xr + c + e) * mul : 0.f;
    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
    *reinterpret_cast<int*>(qr + c) = pk;
  }
}

struct Fp8Args {
  const uint8_t* A; const uint8_t* B; void* C; const float* bias; const float* sa; const float* sb;
  int64_t lda, ldb, ldc;
  int M, N, K, relu, tiles_n;
};

template <typename TO>
__global__ __launch_bounds__(256) void gemm_fp8_nt_kernel(Fp8Args p) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * 64 + wm * 32, n0 = (tile % p.tiles_n) * 64 + wn * 32;
  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // fragment of v_mfma_scale_f32_16x16x128_f8f6f4: lane (row lr, k block g) holds the 32 consecutive bytes k0 + 32 g .. of its row
  const uint8_t* pa[2];
  const uint8_t* pb[2];
  bool oka[2], okb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ra = m0 + i * 16 + lr, rb = n0 + i * 16 + lr;
    oka[i] = ra < p.M; okb[i] = rb < p.N;
    pa[i] = p.A + (int64_t)(oka[i] ? ra : 0) * p.lda + 32 * g;
    pb[i] = p.B + (int64_t)(okb[i] ? rb : 0) * p.ldb + 32 * g;
  }
  const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
  for (int k0 = 0; k0 < p.K; k0 += 128) {
    const bool k_lo = k0 + 32 * g < p.K, k_hi = k0 + 32 * g + 16 < p.K;       // K is a multiple of 16: whole chunks only
    i32x8_t fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4 a0 = (oka[i] && k_lo) ? *reinterpret_cast<const uint4*>(pa[i] + k0) : z4;
      const uint4 a1 = (oka[i] && k_hi) ? *reinterpret_cast<const uint4*>(pa[i] + k0 + 16) : z4;
      const uint4 b0 = (okb[i] && k_lo) ? *reinterpret_cast<const uint4*>(pb[i] + k0) : z4;
      const uint4 b1 = (okb[i] && k_hi) ? *reinterpret_cast<const uint4*>(pb[i] + k0 + 16) : z4;
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
  TO* C = static_cast<TO*>(p.C);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gn = n0 + j * 16 + lr;
    const float sb = gn < p.N ? p.sb[gn] : 0.f;
    const float bv = (p.bias && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = m0 + i * 16 + g * 4 + r;
        if (gm < p.M && gn < p.N) {
          float v = acc[i][j][r] * (p.sa[gm] * sb) + bv;
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
  const unsigned grid = (unsigned)((M + 3) / 4);
  if (dtype == ASR_F32) hipLaunchKernelGGL((fp8_quant_rows_kernel<float>), dim3(grid), dim3(256), 0, s, (const float*)x, ld, M, K, Kp, q, ldq, scale);
  else hipLaunchKernelGGL((fp8_quant_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ld, M, K, Kp, q, ldq, scale);
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
