// Fused multi-head attention core, flash style (no T x T tensor in HBM):
//   O = dropout(softmax(mask(Q K^T * scale))) V      reference: models/common_layers.py:211-225 (+ permutes :185-195,
//   mask.repeat :190, mask construction :28-74 replaced by key_len / key_pad / causal arguments).
//
// Work split: block = 4 waves; in the forward and dQ kernels a wave owns 16 queries and walks 64-key tiles staged
// in LDS; in the dK/dV kernel a wave owns 16 keys and walks 64-query tiles.  All contractions use the "swapped"
// form so that the reduction axis of the softmax is lane-local:
//   S^T[key][q] = K . Q^T      A = K tile rows from LDS (natural [row][d] layout), B = Q rows held in registers
//   O^T[d][q]  += Vt . P^T     A = V tile TRANSPOSED in LDS ([d][key]),            B = P^T straight from the S^T
//                                                                                     accumulator registers
// C-fragment layout (col = lane&15, row = 4*(lane>>4)+reg) makes a lane hold 4 consecutive keys of ONE query, which
// is exactly a B-operand pack for the second contraction, so P never goes through LDS or cross-lane shuffles.
#include "attention.h"

using namespace asr_attn;

namespace {

template <typename T, int HD> struct AT {
  static constexpr int EPC = DT<T>::EPC;
  static constexpr int ESZ = (int)sizeof(T);
  static constexpr int KS = 4 * EPC;                    // k-range of one macro step
  static constexpr int NDS = (HD + KS - 1) / KS;        // macro steps over the head dim
  static constexpr int NDF = HD / 16;                   // 16-wide fragments over the head dim
  static constexpr int CPR = HD / EPC;                  // 16-B chunks per natural row
  static constexpr int PN = HD * ESZ + 16;              // pitch of a natural  [64][HD] tile
  static constexpr int PT = 64 * ESZ + 16;              // pitch of a transposed [HD][64] tile
  static constexpr int NMS = 64 / KS;                   // macro steps over a 64-wide tile axis
};

template <typename T>
__device__ __forceinline__ uint4 gload16(const T* p, bool ok, bool vec) {
  Chunk<T> c;
  c.v = make_uint4(0u, 0u, 0u, 0u);
  if (ok) {
    if (vec) c.v = *reinterpret_cast<const uint4*>(p);
    else {
#pragma unroll
      for (int j = 0; j < DT<T>::EPC; ++j) c.e[j] = p[j];
    }
  }
  return c.v;
}

// rows r0..r0+63 of a (rows, HD) slice -> LDS natural layout; rows >= nrows are zero
template <typename T, int HD>
__device__ __forceinline__ void stage_natural(unsigned char* lds, const T* g, int64_t st, int r0, int nrows, bool vec) {
  using A = AT<T, HD>;
  for (int c = threadIdx.x; c < 64 * A::CPR; c += 256) {
    const int row = c / A::CPR, ch = c % A::CPR;
    const uint4 v = gload16<T>(g + (int64_t)(r0 + row) * st + ch * A::EPC, r0 + row < nrows, vec);
    *reinterpret_cast<uint4*>(lds + row * A::PN + ch * 16) = v;
  }
}
// same slice -> LDS transposed layout [HD][64]
template <typename T, int HD>
__device__ __forceinline__ void stage_transposed(unsigned char* lds, const T* g, int64_t st, int r0, int nrows, bool vec) {
  using A = AT<T, HD>;
  for (int c = threadIdx.x; c < 64 * A::CPR; c += 256) {
    const int row = c & 63, ch = c >> 6;
    Chunk<T> v;
    v.v = gload16<T>(g + (int64_t)(r0 + row) * st + ch * A::EPC, r0 + row < nrows, vec);
#pragma unroll
    for (int j = 0; j < A::EPC; ++j)
      *reinterpret_cast<T*>(lds + (ch * A::EPC + j) * A::PT + row * A::ESZ) = v.e[j];
  }
}
// A-operand pack from a natural tile: row, macro step ds over the head dim
template <typename T, int HD>
__device__ __forceinline__ uint4 frag_nat(const unsigned char* lds, int row, int ds, int g) {
  using A = AT<T, HD>;
  const int col0 = ds * A::KS + g * A::EPC;
  if (col0 < HD) return *reinterpret_cast<const uint4*>(lds + row * A::PN + col0 * A::ESZ);
  return make_uint4(0u, 0u, 0u, 0u);
}
// A-operand pack from a transposed tile for macro step ms over the 64-wide axis, matching pack_c() below
template <typename T, int HD>
__device__ __forceinline__ uint4 frag_tr(const unsigned char* lds, int drow, int ms, int g);
// B-operand pack built from C-fragments: values v[f][r] = X[16 f + 4 g + r][col]
template <typename T> __device__ __forceinline__ uint4 pack_c(const f32x4_t* v, int ms);

template <> __device__ __forceinline__ uint4 pack_c<float>(const f32x4_t* v, int ms) {
  return make_uint4(__float_as_uint(v[ms][0]), __float_as_uint(v[ms][1]), __float_as_uint(v[ms][2]), __float_as_uint(v[ms][3]));
}
template <> __device__ __forceinline__ uint4 pack_c<bf16_t>(const f32x4_t* v, int ms) {
  const f32x4_t a = v[2 * ms], b = v[2 * ms + 1];
  uint4 r;
  r.x = (uint32_t)f32_to_bf16(a[0]) | ((uint32_t)f32_to_bf16(a[1]) << 16);
  r.y = (uint32_t)f32_to_bf16(a[2]) | ((uint32_t)f32_to_bf16(a[3]) << 16);
  r.z = (uint32_t)f32_to_bf16(b[0]) | ((uint32_t)f32_to_bf16(b[1]) << 16);
  r.w = (uint32_t)f32_to_bf16(b[2]) | ((uint32_t)f32_to_bf16(b[3]) << 16);
  return r;
}
#define ASR_FRAG_TR_F32(HD_)                                                                                            \
  template <> __device__ __forceinline__ uint4 frag_tr<float, HD_>(const unsigned char* lds, int drow, int ms, int g) { \
    return *reinterpret_cast<const uint4*>(lds + drow * AT<float, HD_>::PT + (16 * ms + 4 * g) * 4);                    \
  }
#define ASR_FRAG_TR_BF16(HD_)                                                                                            \
  template <> __device__ __forceinline__ uint4 frag_tr<bf16_t, HD_>(const unsigned char* lds, int drow, int ms, int g) { \
    const unsigned char* p = lds + drow * AT<bf16_t, HD_>::PT;                                                          \
    const uint2 lo = *reinterpret_cast<const uint2*>(p + (32 * ms + 4 * g) * 2);                                        \
    const uint2 hi = *reinterpret_cast<const uint2*>(p + (32 * ms + 16 + 4 * g) * 2);                                   \
    return make_uint4(lo.x, lo.y, hi.x, hi.y);                                                                          \
  }
ASR_FRAG_TR_F32(16) ASR_FRAG_TR_F32(32) ASR_FRAG_TR_F32(64)
ASR_FRAG_TR_BF16(16) ASR_FRAG_TR_BF16(32) ASR_FRAG_TR_BF16(64)

// ================================================================================================ forward
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  using A = AT<T, HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sVt = smem + 64 * A::PN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int q = blockIdx.x * 64 + wave * 16 + lr;
  const T* Qb = static_cast<const T*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const T* Kb = static_cast<const T*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const T* Vb = static_cast<const T*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;

  uint4 qf[A::NDS];
#pragma unroll
  for (int ds = 0; ds < A::NDS; ++ds) {
    const int col0 = ds * A::KS + g * A::EPC;
    qf[ds] = gload16<T>(Qb + (int64_t)q * p.q_st + col0, q < p.Tq && col0 < HD, p.vec);
  }
  f32x4_t o[A::NDF];
#pragma unroll
  for (int i = 0; i < A::NDF; ++i) o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  const int kend = key_end(p, b);
  int kstop = kend;
  if (p.causal) kstop = min(kend, (int)blockIdx.x * 64 + 64);

  for (int k0 = 0; k0 < kstop; k0 += 64) {
    __syncthreads();
    stage_natural<T, HD>(sK, Kb, p.k_st, k0, p.Tk, p.vec);
    stage_transposed<T, HD>(sVt, Vb, p.v_st, k0, p.Tk, p.vec);
    __syncthreads();
    f32x4_t s[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      s[kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < A::NDS; ++ds) mma16<T>(s[kf], frag_nat<T, HD>(sK, kf * 16 + lr, ds, g), qf[ds]);
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = k0 + kf * 16 + g * 4 + r;
        float v = s[kf][r] * p.scale;
        if (key_masked(p, b, kg, q, kend)) v = -INFINITY;
        s[kf][r] = v;
        tmax = fmaxf(tmax, v);
      }
    tmax = group_max(tmax);
    const float m_new = fmaxf(m, tmax);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = expf(m - m_safe);
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = expf(s[kf][r] - m_safe);
        psum += pv;
        if (p.thr) {
          const int kg = k0 + kf * 16 + g * 4 + r;
          pv = drop_keep(drop_row_key(seed, drop_row(p, b, h, q)), kg, p.thr) ? pv * p.inv_keep : 0.f;
        }
        s[kf][r] = pv;
      }
    psum = group_sum(psum);
    l = l * alpha + psum;
    m = m_new;
#pragma unroll
    for (int i = 0; i < A::NDF; ++i) o[i] *= alpha;
#pragma unroll
    for (int ms = 0; ms < A::NMS; ++ms) {
      const uint4 pb = pack_c<T>(s, ms);
#pragma unroll
      for (int df = 0; df < A::NDF; ++df) mma16<T>(o[df], frag_tr<T, HD>(sVt, df * 16 + lr, ms, g), pb);
    }
  }

  const float inv_l = l > 0.f ? 1.f / l : 0.f;
  if (q < p.Tq) {
    if (g == 0) p.lse[((int64_t)b * p.H + h) * p.Tq + q] = l > 0.f ? m + logf(l) : INFINITY;
    T* Ob = static_cast<T*>(p.Out) + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < A::NDF; ++df)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        DT<T>::st(Ob + df * 16 + g * 4 + r, o[df][r] * inv_l);
        if (p.Out32) p.Out32[(int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD + df * 16 + g * 4 + r] = o[df][r] * inv_l;
      }
  }
}

// ================================================================================================ delta = sum_d dO * O
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs p, int HD) {
  // one 16-lane group per (b,h,q); lanes stride over d
  const int64_t idx = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t total = (int64_t)p.B * p.H * p.Tq;
  const int sub = threadIdx.x & 15;
  float s = 0.f;
  if (idx < total) {
    const int q = (int)(idx % p.Tq);
    const int h = (int)((idx / p.Tq) % p.H);
    const int b = (int)(idx / ((int64_t)p.Tq * p.H));
    const T* o = static_cast<const T*>(p.O) + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD;
    const T* d = static_cast<const T*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD;
    const float* o32 = p.O32 ? p.O32 + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD : nullptr;
    for (int c = sub; c < HD; c += 16) s += (o32 ? o32[c] : DT<T>::ld(o + c)) * DT<T>::ld(d + c);
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (idx < total && sub == 0) p.delta[idx] = s;
}

// ================================================================================================ delta, consistent form (fp32 parity mode)
// delta[q] = sum_k P'[q,k] dA'[q,k] / sum_k P'[q,k]   (P' = exp(S - lse) as the two backward kernels recompute it, dA' = dropout-scaled
// dO . V_k from the same MFMA contractions; both sums in fp64).  Mathematically this is rowsum(dO * O) (the kernel above).  Numerically
// it makes sum_k dS[q,k] = sum_k P'(dA' - delta) vanish EXACTLY whatever common factor the recomputed P' carries (lse is rounded:
// sum P' = 1 + 2e-7), and every remaining rounding error enters multiplied by the small difference (dA' - delta).  With rowsum(dO * O)
// the forward's rounding of O leaves a rank-one residual (delta_true - delta) * P[q,k], which the query / key projections' gradients
// see multiplied by the MEAN key: when attention is nearly uniform over ~800 similar rows (the cross-attention of configs[3] at random
// init: the true gradient is a difference 1e-3 of its terms) that residual dominated -- measured at cfg3_b16: 1.1e-3 - 2.4e-3 relative
// against the fp64 oracle where the reference's own fp32 floor is 0.9e-4 - 2.4e-4; the un-normalised sum is WORSE (6.7e-3: the 2e-7 of
// lse times delta).  Reference: torch's softmax backward, grad * P - P * sum(grad * P) (models/common_layers.py:211-225).
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_delta_consistent_kernel(AttnArgs p) {
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  using A = AT<T, HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + 64 * A::PN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int q = blockIdx.x * 64 + wave * 16 + lr;
  const T* Qb = static_cast<const T*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const T* Kb = static_cast<const T*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const T* Vb = static_cast<const T*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const T* dOb = static_cast<const T*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)h * HD;
  uint4 qf[A::NDS], dof[A::NDS];
#pragma unroll
  for (int ds = 0; ds < A::NDS; ++ds) {
    const int col0 = ds * A::KS + g * A::EPC;
    qf[ds] = gload16<T>(Qb + (int64_t)q * p.q_st + col0, q < p.Tq && col0 < HD, p.vec);
    dof[ds] = gload16<T>(dOb + (int64_t)q * p.o_st + col0, q < p.Tq && col0 < HD, p.vec);
  }
  const int64_t sidx = ((int64_t)b * p.H + h) * p.Tq + q;
  const float lse = q < p.Tq ? p.lse[sidx] : INFINITY;
  const int kend = key_end(p, b);
  int kstop = kend;
  if (p.causal) kstop = min(kend, (int)blockIdx.x * 64 + 64);
  double acc = 0.0, accp = 0.0;
  for (int k0 = 0; k0 < kstop; k0 += 64) {
    __syncthreads();
    stage_natural<T, HD>(sK, Kb, p.k_st, k0, p.Tk, p.vec);
    stage_natural<T, HD>(sV, Vb, p.v_st, k0, p.Tk, p.vec);
    __syncthreads();
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < A::NDS; ++ds) {
        mma16<T>(s, frag_nat<T, HD>(sK, kf * 16 + lr, ds, g), qf[ds]);
        mma16<T>(dp, frag_nat<T, HD>(sV, kf * 16 + lr, ds, g), dof[ds]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = k0 + kf * 16 + g * 4 + r;
        float pv = 0.f;
        if (!key_masked(p, b, kg, q, kend)) pv = expf(s[r] * p.scale - lse);
        float da = dp[r];
        if (p.thr) da = drop_keep(drop_row_key(seed, drop_row(p, b, h, q)), kg, p.thr) ? da * p.inv_keep : 0.f;
        acc += (double)pv * (double)da;
        accp += (double)pv;
      }
    }
  }
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  accp += __shfl_xor(accp, 16, 64);
  accp += __shfl_xor(accp, 32, 64);
  if (q < p.Tq && g == 0) p.delta[sidx] = accp > 0.0 ? (float)(acc / accp) : 0.f;
}

// ================================================================================================ dQ
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnArgs p) {
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  using A = AT<T, HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned char* sV = smem + 64 * A::PN;
  unsigned char* sKt = smem + 2 * 64 * A::PN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int q = blockIdx.x * 64 + wave * 16 + lr;
  const T* Qb = static_cast<const T*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const T* Kb = static_cast<const T*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const T* Vb = static_cast<const T*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const T* dOb = static_cast<const T*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)h * HD;

  uint4 qf[A::NDS], dof[A::NDS];
#pragma unroll
  for (int ds = 0; ds < A::NDS; ++ds) {
    const int col0 = ds * A::KS + g * A::EPC;
    qf[ds] = gload16<T>(Qb + (int64_t)q * p.q_st + col0, q < p.Tq && col0 < HD, p.vec);
    dof[ds] = gload16<T>(dOb + (int64_t)q * p.o_st + col0, q < p.Tq && col0 < HD, p.vec);
  }
  const int64_t sidx = ((int64_t)b * p.H + h) * p.Tq + q;
  const float lse = q < p.Tq ? p.lse[sidx] : INFINITY;
  const float dlt = q < p.Tq ? p.delta[sidx] : 0.f;
  f32x4_t dq[A::NDF];
#pragma unroll
  for (int i = 0; i < A::NDF; ++i) dq[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int kend = key_end(p, b);
  int kstop = kend;
  if (p.causal) kstop = min(kend, (int)blockIdx.x * 64 + 64);

  for (int k0 = 0; k0 < kstop; k0 += 64) {
    __syncthreads();
    stage_natural<T, HD>(sK, Kb, p.k_st, k0, p.Tk, p.vec);
    stage_natural<T, HD>(sV, Vb, p.v_st, k0, p.Tk, p.vec);
    stage_transposed<T, HD>(sKt, Kb, p.k_st, k0, p.Tk, p.vec);
    __syncthreads();
    f32x4_t ds_[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < A::NDS; ++ds) {
        mma16<T>(s, frag_nat<T, HD>(sK, kf * 16 + lr, ds, g), qf[ds]);
        mma16<T>(dp, frag_nat<T, HD>(sV, kf * 16 + lr, ds, g), dof[ds]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = k0 + kf * 16 + g * 4 + r;
        float pv = 0.f;
        if (!key_masked(p, b, kg, q, kend)) pv = expf(s[r] * p.scale - lse);
        float da = dp[r];
        if (p.thr) da = drop_keep(drop_row_key(seed, drop_row(p, b, h, q)), kg, p.thr) ? da * p.inv_keep : 0.f;
        ds_[kf][r] = pv * (da - dlt);
      }
    }
#pragma unroll
    for (int ms = 0; ms < A::NMS; ++ms) {
      const uint4 pb = pack_c<T>(ds_, ms);
#pragma unroll
      for (int df = 0; df < A::NDF; ++df) mma16<T>(dq[df], frag_tr<T, HD>(sKt, df * 16 + lr, ms, g), pb);
    }
  }
  if (q < p.Tq) {
    T* o = static_cast<T*>(p.dQ) + (int64_t)b * p.q_sb + (int64_t)q * p.q_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < A::NDF; ++df)
#pragma unroll
      for (int r = 0; r < 4; ++r) DT<T>::st(o + df * 16 + g * 4 + r, dq[df][r] * p.scale);
  }
}

// ================================================================================================ dK, dV
// wave owns 16 keys (B-operand columns); loops over 64-query tiles.
//   S[q][key]   = Q . K^T      A = Q tile natural,  B = K rows in registers
//   dAd[q][key] = dO . V^T     A = dO tile natural, B = V rows in registers
//   dV^T[d][key] += dOt . Ad   A = dO tile transposed, B = Ad from registers
//   dK^T[d][key] += Qt . dS    A = Q tile transposed,  B = dS from registers
template <typename T, int HD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnArgs p) {
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  using A = AT<T, HD>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sQ = smem;
  unsigned char* sdO = smem + 64 * A::PN;
  unsigned char* sQt = smem + 2 * 64 * A::PN;
  unsigned char* sdOt = sQt + HD * A::PT;
  float* s_lse = reinterpret_cast<float*>(sdOt + HD * A::PT);
  float* s_dlt = s_lse + 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, g = lane >> 4;
  const int b = blockIdx.y / p.H, h = blockIdx.y % p.H;
  const int key = blockIdx.x * 64 + wave * 16 + lr;
  const T* Qb = static_cast<const T*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const T* Kb = static_cast<const T*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const T* Vb = static_cast<const T*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const T* dOb = static_cast<const T*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)h * HD;
  const int kend = key_end(p, b);

  uint4 kfr[A::NDS], vfr[A::NDS];
#pragma unroll
  for (int ds = 0; ds < A::NDS; ++ds) {
    const int col0 = ds * A::KS + g * A::EPC;
    kfr[ds] = gload16<T>(Kb + (int64_t)key * p.k_st + col0, key < p.Tk && col0 < HD, p.vec);
    vfr[ds] = gload16<T>(Vb + (int64_t)key * p.v_st + col0, key < p.Tk && col0 < HD, p.vec);
  }
  const bool key_dead = key >= kend;
  f32x4_t dk[A::NDF], dv[A::NDF];
#pragma unroll
  for (int i = 0; i < A::NDF; ++i) { dk[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  // causal: queries below the first key of this block never see it
  const int qstart = p.causal ? ((int)blockIdx.x * 64 / 64) * 64 : 0;

  for (int q0 = qstart; q0 < p.Tq; q0 += 64) {
    __syncthreads();
    stage_natural<T, HD>(sQ, Qb, p.q_st, q0, p.Tq, p.vec);
    stage_natural<T, HD>(sdO, dOb, p.o_st, q0, p.Tq, p.vec);
    stage_transposed<T, HD>(sQt, Qb, p.q_st, q0, p.Tq, p.vec);
    stage_transposed<T, HD>(sdOt, dOb, p.o_st, q0, p.Tq, p.vec);
    if (threadIdx.x < 64) {
      const int qq = q0 + threadIdx.x;
      const int64_t si = ((int64_t)b * p.H + h) * p.Tq + qq;
      s_lse[threadIdx.x] = qq < p.Tq ? p.lse[si] : INFINITY;
      s_dlt[threadIdx.x] = qq < p.Tq ? p.delta[si] : 0.f;
    }
    __syncthreads();
    f32x4_t ad[4], dsv[4];
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) {
      f32x4_t s = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < A::NDS; ++ds) {
        mma16<T>(s, frag_nat<T, HD>(sQ, qf * 16 + lr, ds, g), kfr[ds]);
        mma16<T>(dp, frag_nat<T, HD>(sdO, qf * 16 + lr, ds, g), vfr[ds]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = qf * 16 + g * 4 + r, qq = q0 + ql;
        float pv = 0.f;
        if (!key_dead && qq < p.Tq && !key_masked(p, b, key, qq, kend)) pv = expf(s[r] * p.scale - s_lse[ql]);
        float keepf = 1.f;
        if (p.thr) keepf = drop_keep(drop_row_key(seed, drop_row(p, b, h, qq)), key, p.thr) ? p.inv_keep : 0.f;
        ad[qf][r] = pv * keepf;
        dsv[qf][r] = pv * (dp[r] * keepf - s_dlt[ql]);
      }
    }
#pragma unroll
    for (int ms = 0; ms < A::NMS; ++ms) {
      const uint4 pa = pack_c<T>(ad, ms), pd = pack_c<T>(dsv, ms);
#pragma unroll
      for (int df = 0; df < A::NDF; ++df) {
        mma16<T>(dv[df], frag_tr<T, HD>(sdOt, df * 16 + lr, ms, g), pa);
        mma16<T>(dk[df], frag_tr<T, HD>(sQt, df * 16 + lr, ms, g), pd);
      }
    }
  }
  if (key < p.Tk) {
    T* ok = static_cast<T*>(p.dK) + (int64_t)b * p.k_sb + (int64_t)key * p.k_st + (int64_t)h * HD;
    T* ov = static_cast<T*>(p.dV) + (int64_t)b * p.v_sb + (int64_t)key * p.v_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < A::NDF; ++df)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        DT<T>::st(ok + df * 16 + g * 4 + r, dk[df][r] * p.scale);
        DT<T>::st(ov + df * 16 + g * 4 + r, dv[df][r]);
      }
  }
}

// ================================================================================================ probabilities
// Materialises the (H*B, Tq, Tk) post-dropout attention matrix the reference returns from
// MultiHeadAttention.forward (common_layers.py:200).  Only used when a caller asks for it.
template <typename T>
__global__ __launch_bounds__(256) void attn_probs_kernel(AttnArgs p, int HD) {
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  const int64_t total = (int64_t)p.B * p.H * p.Tq * p.Tk;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int k = (int)(i % p.Tk);
  const int q = (int)((i / p.Tk) % p.Tq);
  const int b = (int)((i / ((int64_t)p.Tk * p.Tq)) % p.B);
  const int h = (int)(i / ((int64_t)p.Tk * p.Tq * p.B));
  const T* qp = static_cast<const T*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)q * p.q_st + (int64_t)h * HD;
  const T* kp = static_cast<const T*>(p.K) + (int64_t)b * p.k_sb + (int64_t)k * p.k_st + (int64_t)h * HD;
  float s = 0.f;
  for (int c = 0; c < HD; ++c) s += DT<T>::ld(qp + c) * DT<T>::ld(kp + c);
  float pv = 0.f;
  if (!key_masked(p, b, k, q, key_end(p, b))) pv = expf(s * p.scale - p.lse[((int64_t)b * p.H + h) * p.Tq + q]);
  if (p.thr) pv = drop_keep(drop_row_key(seed, drop_row(p, b, h, q)), k, p.thr) ? pv * p.inv_keep : 0.f;
  p.attn_out[i] = pv;
}

template <typename T, int HD> size_t lds_fwd() { return (size_t)64 * AT<T, HD>::PN + (size_t)HD * AT<T, HD>::PT; }
template <typename T, int HD> size_t lds_dq() { return (size_t)2 * 64 * AT<T, HD>::PN + (size_t)HD * AT<T, HD>::PT; }
template <typename T, int HD> size_t lds_dkv() { return (size_t)2 * 64 * AT<T, HD>::PN + (size_t)2 * HD * AT<T, HD>::PT + 128 * sizeof(float); }

int launch_probs(const AttnArgs& p, int d, int dtype, hipStream_t s) {
  const int64_t total = (int64_t)p.B * p.H * p.Tq * p.Tk;
  if (dtype == ASR_F32)
    hipLaunchKernelGGL((attn_probs_kernel<float>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, s, p, d);
  else
    hipLaunchKernelGGL((attn_probs_kernel<bf16_t>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, s, p, d);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

template <typename T, int HD>
int run_fwd(const AttnArgs& p, hipStream_t s) {
  dim3 grid((p.Tq + 63) / 64, p.B * p.H);
  const size_t lds = lds_fwd<T, HD>();
  hipLaunchKernelGGL((attn_fwd_kernel<T, HD>), grid, dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  if (p.attn_out) {
    const int64_t total = (int64_t)p.B * p.H * p.Tq * p.Tk;
    hipLaunchKernelGGL((attn_probs_kernel<T>), dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, s, p, HD);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}
template <typename T, int HD>
int run_bwd(const AttnArgs& p, hipStream_t s) {
  const int64_t rows = (int64_t)p.B * p.H * p.Tq;
  const size_t l1 = lds_dq<T, HD>(), l2 = lds_dkv<T, HD>();
  if (p.parts & ASR_ATTN_DELTA) {
    if constexpr (sizeof(T) == 4)      // fp32 parity mode: the consistent form (one more pass over the keys; see the kernel's note)
      hipLaunchKernelGGL((attn_delta_consistent_kernel<T, HD>), dim3((p.Tq + 63) / 64, p.B * p.H), dim3(256), l1, s, p);
    else
      hipLaunchKernelGGL((attn_delta_kernel<T>), dim3((unsigned)ceil_div64(rows, 16)), dim3(256), 0, s, p, HD);
    ASR_LAUNCH_CHECK();
  }
  static bool granted = false;
  if (l2 > 48 * 1024 && !granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<T, HD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
    granted = true;
  }
  if (p.parts & ASR_ATTN_DQ) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<T, HD>), dim3((p.Tq + 63) / 64, p.B * p.H), dim3(256), l1, s, p);
    ASR_LAUNCH_CHECK();
  }
  if (p.parts & ASR_ATTN_DKV) {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, HD>), dim3((p.Tk + 63) / 64, p.B * p.H), dim3(256), l2, s, p);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}

template <typename T>
int dispatch(const AttnArgs& p, int d, bool bwd, hipStream_t s) {
  switch (d) {
    case 16: return bwd ? run_bwd<T, 16>(p, s) : run_fwd<T, 16>(p, s);
    case 32: return bwd ? run_bwd<T, 32>(p, s) : run_fwd<T, 32>(p, s);
    case 64: return bwd ? run_bwd<T, 64>(p, s) : run_fwd<T, 64>(p, s);
    default: return ASR_EUNSUPPORTED;
  }
}

int fill_common(AttnArgs& p, int B, int H, int Tq, int Tk, int d, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st,
                int64_t v_sb, int64_t v_st, int64_t o_sb, int64_t o_st, const int32_t* key_len, const uint8_t* key_pad,
                int64_t m_sb, int64_t m_sq, int causal, float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev,
                int dtype) {
  ASR_CHECK_ARG(B >= 0 && H > 0 && Tq >= 0 && Tk >= 0 && d > 0 && dropout_p >= 0.f && dropout_p < 1.f);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  p.B = B; p.H = H; p.Tq = Tq; p.Tk = Tk;
  p.q_sb = q_sb; p.q_st = q_st; p.k_sb = k_sb; p.k_st = k_st; p.v_sb = v_sb; p.v_st = v_st; p.o_sb = o_sb; p.o_st = o_st;
  p.key_len = key_len; p.key_pad = key_pad; p.m_sb = m_sb; p.m_sq = m_sq; p.causal = causal; p.scale = scale;
  // 16-bit dropout threshold; the rescale uses the QUANTISED probability so that E[dropout(x)] = x exactly
  p.thr = (uint32_t)(dropout_p * 65536.f + 0.5f);
  if (p.thr > 65535u) p.thr = 65535u;
  p.inv_keep = 1.f / (1.f - (float)p.thr / 65536.f); p.seed = seed; p.seed_dev = seed_dev;
  ASR_CHECK_ARG((int64_t)B * H * (Tq > 0 ? Tq : 1) < (int64_t)1 << 32);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  p.vec = (q_sb % epc == 0) && (q_st % epc == 0) && (k_sb % epc == 0) && (k_st % epc == 0) && (v_sb % epc == 0) &&
          (v_st % epc == 0) && (o_sb % epc == 0) && (o_st % epc == 0);
  return ASR_OK;
}

}  // namespace

extern "C" int asr_attn_fwd(const void* Q, const void* K, const void* V, void* O, float* o32, float* lse, float* attn_out, int B, int H,
                            int Tq, int Tk, int d, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb,
                            int64_t v_st, int64_t o_sb, int64_t o_st, const int32_t* key_len, const uint8_t* key_pad,
                            int64_t mask_sb, int64_t mask_sq, int causal, float scale, float dropout_p, uint64_t seed,
                            const uint64_t* seed_dev, int dtype, hipStream_t stream) {
  ASR_CHECK_ARG(Q && K && V && O && lse);
  AttnArgs p{};
  int rc = fill_common(p, B, H, Tq, Tk, d, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, key_len, key_pad, mask_sb, mask_sq, causal,
                       scale, dropout_p, seed, seed_dev, dtype);
  if (rc != ASR_OK) return rc;
  if (B == 0 || Tq == 0) return ASR_OK;
  p.Q = Q; p.K = K; p.V = V; p.Out = O; p.Out32 = o32; p.lse = lse; p.attn_out = attn_out;
  ASR_CHECK_ARG(!o32 || aligned16(o32));
  p.vec = p.vec && aligned16(Q) && aligned16(K) && aligned16(V);
  AsrProfScope prof(ASR_OP_ATTN_FWD, stream);
  {
    rc = attn_fast_fwd(p, d, dtype, stream);
    if (rc != ASR_EUNSUPPORTED) {
      if (rc == ASR_OK && attn_out) rc = launch_probs(p, d, dtype, stream);
      return rc;
    }
  }
  return dtype == ASR_F32 ? dispatch<float>(p, d, false, stream) : dispatch<bf16_t>(p, d, false, stream);
}

extern "C" int asr_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const float* o32, const void* dO, const float* lse,
                            float* delta, void* dQ, void* dK, void* dV, int B, int H, int Tq, int Tk, int d, int64_t q_sb,
                            int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st, int64_t o_sb, int64_t o_st,
                            const int32_t* key_len, const uint8_t* key_pad, int64_t mask_sb, int64_t mask_sq, int causal,
                            float scale, float dropout_p, uint64_t seed, const uint64_t* seed_dev, int parts, int dtype,
                            hipStream_t stream) {
  ASR_CHECK_ARG(Q && K && V && O && dO && lse && delta && dQ && dK && dV);
  ASR_CHECK_ARG(parts > 0 && parts <= (ASR_ATTN_DELTA | ASR_ATTN_DQ | ASR_ATTN_DKV));
  AttnArgs p{};
  int rc = fill_common(p, B, H, Tq, Tk, d, q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st, key_len, key_pad, mask_sb, mask_sq, causal,
                       scale, dropout_p, seed, seed_dev, dtype);
  if (rc != ASR_OK) return rc;
  if (B == 0 || Tq == 0 || Tk == 0) return ASR_OK;
  p.Q = Q; p.K = K; p.V = V; p.O = O; p.O32 = o32; p.dO = dO; p.lse = const_cast<float*>(lse); p.delta = delta;
  p.dQ = dQ; p.dK = dK; p.dV = dV; p.parts = parts;
  p.vec = p.vec && aligned16(Q) && aligned16(K) && aligned16(V) && aligned16(dO);
  AsrProfScope prof(ASR_OP_ATTN_BWD, stream);
  rc = attn_fast_bwd(p, d, dtype, stream);
  if (rc != ASR_EUNSUPPORTED) return rc;
  return dtype == ASR_F32 ? dispatch<float>(p, d, true, stream) : dispatch<bf16_t>(p, d, true, stream);
}
