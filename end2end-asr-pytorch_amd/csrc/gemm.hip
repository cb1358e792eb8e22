// NT GEMM on MFMA:  C[M,N] (op)= alpha * sum_k A[m,k] * B[n,k]  (+ bias[n]) (ReLU)
//
// Replaces every nn.Linear / Conv1d(k=1) call on the hot path (reference: models/common_layers.py:136-142,
// :181-187, :197; models/asr/transformer.py:172, :302) and, with explicitly transposed operands, their dgrad
// and wgrad.  Both operands are K-contiguous ("NT"), which is how nn.Linear stores its weight (N,K).
//
// Structure: 256 threads = 4 waves (2x2), tile BMxBN, LDS row = 128 data bytes (+16 pad) per tile row,
// register-staged global->LDS with the next tile's loads issued before the current tile's MFMAs.
#include "common.h"

namespace {

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* mask;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int k_per_split;     // multiple of BK
  float alpha;
  int relu, accumulate, atomic, vecA, vecB;
  int tiles_n;
};

constexpr int kPitch = 144;   // bytes per LDS tile row: 128 data + 16 pad (keeps 16-B alignment, breaks the 128-B stride)

template <typename T>
__device__ __forceinline__ uint4 load_chunk(const T* row, int64_t k, int64_t kend, bool row_ok, bool vec) {
  Chunk<T> c;
  c.v = make_uint4(0u, 0u, 0u, 0u);
  if (row_ok) {
    if (vec) {
      if (k < kend) c.v = *reinterpret_cast<const uint4*>(row + k);
    } else {
#pragma unroll
      for (int j = 0; j < DT<T>::EPC; ++j)
        if (k + j < kend) c.e[j] = row[k + j];
    }
  }
  return c.v;
}

template <typename TO> __device__ __forceinline__ void store_out(TO* p, float v, int accumulate, int atomic);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v, int accumulate, int atomic) {
  if (atomic) atomicAdd(p, v);
  else if (accumulate) *p += v;
  else *p = v;
}
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v, int accumulate, int) {
  if (accumulate) v += bf16_to_f32(*p);
  *p = f32_to_bf16(v);
}

template <typename T, typename TO, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int BK = 128 / (int)sizeof(T);
  constexpr int CA = BM * 8 / 256, CB = BN * 8 / 256;
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + BM * kPitch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_per_split;
  const int64_t kend = min((int64_t)p.K, kbeg + p.k_per_split);
  const T* A = static_cast<const T*>(p.A);
  const T* B = static_cast<const T*>(p.B);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[CA], rb[CB];
  auto gload = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      int gm = m0 + row;
      ra[i] = load_chunk<T>(A + (int64_t)gm * p.lda, k0 + kc * EPC, kend, gm < p.M, p.vecA);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      int gn = n0 + row;
      rb[i] = load_chunk<T>(B + (int64_t)gn * p.ldb, k0 + kc * EPC, kend, gn < p.N, p.vecB);
    }
  };
  auto swrite = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(sA + row * kPitch + kc * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(sB + row * kPitch + kc * 16) = rb[i];
    }
  };

  if (kbeg < kend) gload(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    swrite();
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      uint4 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        a[i] = *reinterpret_cast<const uint4*>(sA + (wm * WM + i * 16 + lr) * kPitch + (ms * 4 + g) * 16);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        b[j] = *reinterpret_cast<const uint4*>(sB + (wn * WN + j * 16 + lr) * kPitch + (ms * 4 + g) * 16);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma16<T>(acc[i][j], a[i], b[j]);
    }
    __syncthreads();
  }

  TO* C = static_cast<TO*>(p.C);
  const T* Msk = static_cast<const T*>(p.mask);
  const bool add_bias = p.bias != nullptr && blockIdx.z == 0;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = n0 + wn * WN + j * 16 + lr;
    if (col >= p.N) continue;
    const float bv = add_bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WM + i * 16 + g * 4 + r;
        if (row < p.M) {
          float v = acc[i][j][r] * p.alpha + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (Msk && !(DT<T>::ld(Msk + (int64_t)row * p.ldc + col) > 0.f)) v = 0.f;
          store_out<TO>(C + (int64_t)row * p.ldc + col, v, p.accumulate, p.atomic);
        }
      }
    }
  }
}

template <typename T, typename TO, int BM, int BN>
int launch(const GemmArgs& a, int splits, hipStream_t s) {
  GemmArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, (unsigned)splits);
  const size_t lds = (size_t)(BM + BN) * kPitch;
  hipLaunchKernelGGL((gemm_nt_kernel<T, TO, BM, BN>), grid, dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

template <typename T, typename TO>
int dispatch_tile(const GemmArgs& a, int splits, hipStream_t s) {
  const int64_t t128 = ceil_div64(a.M, 128) * ceil_div64(a.N, 128) * splits;
  const int64_t t12864 = ceil_div64(a.M, 128) * ceil_div64(a.N, 64) * splits;
  if (t128 >= 384 || (a.M > 64 && a.N > 64 && t12864 < 8)) return launch<T, TO, 128, 128>(a, splits, s);
  if (t12864 >= 384 && a.M > 64) return launch<T, TO, 128, 64>(a, splits, s);
  return launch<T, TO, 64, 64>(a, splits, s);
}

}  // namespace

extern "C" int asr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                           const float* bias, const void* relu_mask, int M, int N, int K, float alpha, int flags,
                           int splits, int in_dtype, int out_dtype, hipStream_t stream) {
  ASR_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0);
  if (M == 0 || N == 0) return ASR_OK;
  ASR_CHECK_ARG(in_dtype == ASR_F32 || in_dtype == ASR_BF16);
  ASR_CHECK_ARG(out_dtype == ASR_F32 || out_dtype == ASR_BF16);
  ASR_CHECK_ARG(!(in_dtype == ASR_F32 && out_dtype == ASR_BF16));
  const int esz = in_dtype == ASR_F32 ? 4 : 2, epc = 16 / esz, bk = 128 / esz;
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.mask = relu_mask;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.alpha = alpha;
  p.relu = (flags & ASR_GEMM_RELU) != 0;
  p.accumulate = (flags & ASR_GEMM_ACCUMULATE) != 0;
  if (splits < 1) splits = 1;
  int kps = (int)(ceil_div64(ceil_div64(K > 0 ? K : 1, splits), bk) * bk);
  splits = (int)ceil_div64(K > 0 ? K : 1, kps);
  p.k_per_split = kps;
  p.atomic = 0;
  if (splits > 1) {
    // split-K partial sums are combined with fp32 atomics: the destination must already hold the value to add to
    ASR_CHECK_ARG(out_dtype == ASR_F32 && p.accumulate && !p.relu && !relu_mask);
    p.atomic = 1;
  }
  p.vecA = aligned16(A) && (lda % epc == 0) && (K % epc == 0);
  p.vecB = aligned16(B) && (ldb % epc == 0) && (K % epc == 0);
  AsrProfScope prof(ASR_OP_GEMM, stream);
  if (in_dtype == ASR_F32) return dispatch_tile<float, float>(p, splits, stream);
  if (out_dtype == ASR_BF16) return dispatch_tile<bf16_t, bf16_t>(p, splits, stream);
  return dispatch_tile<bf16_t, float>(p, splits, stream);
}
