// Attention kernels specialised for the reference configurations: bf16 storage, head dim 64, 16-byte aligned rows
// (models/common_layers.py:211-225 at dim_key = dim_value = 64).  Same algorithm and numerics as attention.hip
// (flash style, swapped contractions S^T = K Q^T, O^T = V^T P^T, fp32 softmax statistics); what changes is the data path:
//   * K / V tiles (64 keys x 64 dims, natural [key][d] layout) go HBM -> LDS with the LDS-DMA (global_load_lds, 16 B per
//     lane, XOR-swizzled 16-byte chunks), double buffered: the next tile is in flight while this one is consumed;
//   * the V^T operand of O^T += V^T P^T is read from the NATURAL V tile with ds_read_b64_tr_b16 (no transposed copy);
//   * a wave owns 32 queries (two B fragments), so every LDS operand read feeds two MFMAs;
//   * softmax in base 2 (v_exp_f32 directly, scale*log2(e) folded into one FMA), hardware bf16 packing, pair-wise dropout
//     hash, masks evaluated only on tiles that touch a boundary;
//   * workgroups that share K/V (the query blocks of one (b,h)) are placed on the same XCD.
#include "attention.h"

namespace asr_attn {
namespace {

constexpr int HD = 64;
constexpr int ROWB = 128;                 // bytes per LDS row (64 bf16)
constexpr int TILE = 64 * ROWB;           // one 64-row operand tile
constexpr float LOG2E = 1.4426950408889634f;

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ uint2 lds_read_tr16(const unsigned char* p) {
  const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)p);
  return __builtin_bit_cast(uint2, v);
}

// XCD-aware linear workgroup id: consecutive ids run on the same XCD (hardware deals blockIdx round-robin over 8 XCDs)
__device__ __forceinline__ int xcd_linear_id() {
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
}

// 64 rows x 128 B of a (rows, 64) bf16 slice -> LDS, chunk c of row r stored at slot c ^ (r & 7).  Rows past `nrows`
// re-read the last valid row (the LDS-DMA cannot zero-fill; such rows are masked / never stored by the callers).
__device__ __forceinline__ void stage_tile(unsigned char* lds, const bf16_t* g, int64_t st, int r0, int nrows, int tid, int wave) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 256 + tid, row = c >> 3, slot = c & 7;
    int gr = r0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    const bf16_t* src = g + (int64_t)gr * st + ((slot ^ (row & 7)) << 3);
    unsigned char* dst = lds + (i * 256 + wave * 64) * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}
// A-operand pack (one row, 8 consecutive d) from a natural tile: row, macro step ds over d, lane group g
__device__ __forceinline__ uint4 frag_rows(const unsigned char* tile, int row, int ds, int g) {
  return *reinterpret_cast<const uint4*>(tile + row * ROWB + (((ds * 4 + g) ^ (row & 7)) << 4));
}
// A-operand pack of the TRANSPOSED tile: for column (= output row) c0 + lr, the 8 tile rows 32 ms + 4 g + {0..3} and
// 32 ms + 16 + 4 g + {0..3} -- the k order of pack_p() below.  ds_read_b64_tr_b16: lane i of a 16-lane group supplies the
// 8-byte address of row p0 + (i >> 2), columns c0 + 4 (i & 3).., and receives rows p0..p0+3 of column c0 + i.
__device__ __forceinline__ uint4 frag_cols(const unsigned char* tile, int c0, int ms, int lr, int g) {
  const int row = 32 * ms + 4 * g + (lr >> 2), col = c0 + 4 * (lr & 3);
  const int chunk = col >> 3, half = (col >> 2) & 1;
  // the builtin (not inline asm) so that the compiler tracks lgkmcnt for the result registers itself
  const uint2 lo = lds_read_tr16(tile + row * ROWB + ((chunk ^ (row & 7)) << 4) + half * 8);
  const uint2 hi = lds_read_tr16(tile + (row + 16) * ROWB + ((chunk ^ ((row + 16) & 7)) << 4) + half * 8);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// B-operand pack from C fragments v[f][r] = X[16 f + 4 g + r][col]: k = 32 ms + 4 g + r (f = 2 ms), 32 ms + 16 + 4 g + r (f = 2 ms + 1)
__device__ __forceinline__ uint4 pack_p(const f32x4_t* v, int ms) {
  const f32x4_t a = v[2 * ms], b = v[2 * ms + 1];
  return make_uint4(pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3]));
}
__device__ __forceinline__ void mma(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
__device__ __forceinline__ uint4 load_row16(const bf16_t* base, int64_t st, int row, int nrows, int col0) {
  const int r = row < nrows ? row : nrows - 1;
  return *reinterpret_cast<const uint4*>(base + (int64_t)r * st + col0);
}

// Mask one 64-key tile of raw scores s[qi][kf][r] (key = k0 + 16 kf + 4 g + r, query = q_lane + 16 qi): straight-line
// selects, mask bytes read with clamped (always valid) addresses.
__device__ __forceinline__ void mask_scores(const AttnArgs& p, f32x4_t (*s)[4], int b, int k0, int g, int q_lane, int kend) {
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int q = q_lane + qi * 16;
    const uint8_t* mrow = p.key_pad ? p.key_pad + (int64_t)b * p.m_sb + (int64_t)(q < p.Tq ? q : p.Tq - 1) * p.m_sq : nullptr;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = k0 + kf * 16 + g * 4 + r;
        bool dead = kg >= kend;
        if (p.causal) dead = dead || kg > q;
        if (mrow) dead = dead || mrow[kg < p.Tk ? kg : p.Tk - 1] != 0;
        s[qi][kf][r] = dead ? -INFINITY : s[qi][kf][r];
      }
  }
}

// ================================================================================================ forward
// workgroup = 4 waves x 32 queries; grid = ceil(Tq / 128) * B * H (1-D, XCD-aware).
constexpr int FQ = 128;

__global__ __launch_bounds__(256) void attn_fwd_bf16_d64_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];      // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.Tq + FQ - 1) / FQ;
  const int vid = xcd_linear_id();
  const int bh = vid / nqb, qb = vid - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int qw = qb * FQ + wave * 32;                       // first query of this wave
  const bf16_t* Qb = static_cast<const bf16_t*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const bf16_t* Kb = static_cast<const bf16_t*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const bf16_t* Vb = static_cast<const bf16_t*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  const float c2 = p.scale * LOG2E;

  uint4 qf[2][2];
  uint32_t rkey[2];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) qf[qi][ds] = load_row16(Qb, p.q_st, qw + qi * 16 + lr, p.Tq, ds * 32 + g * 8);
    rkey[qi] = drop_row_key(seed, drop_row(p, b, h, qw + qi * 16 + lr));
  }
  f32x4_t o[2][4];
  float m[2], l[2];
#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    m[qi] = -INFINITY;
    l[qi] = 0.f;
#pragma unroll
    for (int df = 0; df < 4; ++df) o[qi][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int kend = key_end(p, b);
  int kstop = kend;
  if (p.causal) kstop = min(kend, qb * FQ + FQ);
  const int ntile = (kstop + 63) >> 6;

  if (ntile > 0) {
    stage_tile(smem, Kb, p.k_st, 0, p.Tk, tid, wave);
    stage_tile(smem + TILE, Vb, p.v_st, 0, p.Tk, tid, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int k0 = t << 6;
    const unsigned char* sK = smem + (t & 1) * 2 * TILE;
    const unsigned char* sV = sK + TILE;
    // ---- S^T[key][q] = K . Q^T
    f32x4_t s[2][4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      s[0][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      s[1][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const uint4 a = frag_rows(sK, kf * 16 + lr, ds, g);
        mma(s[0][kf], a, qf[0][ds]);
        mma(s[1][kf], a, qf[1][ds]);
      }
    }
    // V^T operand packs: issued now so that the transposing LDS reads complete under the softmax arithmetic
    uint4 vt[2][4];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int df = 0; df < 4; ++df) vt[ms][df] = frag_cols(sV, df * 16, ms, lr, g);
    // ---- masks only where a boundary crosses this tile (wave-uniform test)
    const bool need_mask = (k0 + 64 > kend) || p.key_pad != nullptr || (p.causal && k0 + 63 > qw);
    if (need_mask) mask_scores(p, s, b, k0, g, qw + lr, kend);
    // next tile: HBM -> LDS in flight under the softmax and the second contraction.  Issued AFTER the mask bytes were
    // consumed: ordinary loads and LDS-DMA loads share vmcnt, and waiting for the former with the latter in flight
    // was observed to return stale mask bytes.
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < ntile) {
      unsigned char* nb = smem + ((t + 1) & 1) * 2 * TILE;
      stage_tile(nb, Kb, p.k_st, k0 + 64, p.Tk, tid, wave);
      stage_tile(nb + TILE, Vb, p.v_st, k0 + 64, p.Tk, tid, wave);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- online softmax (base 2), dropout, pack P^T as the B operand
    uint4 pb[2][2];
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      float mx = -INFINITY;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qi][kf][r]);
      mx = group_max(mx);
      const float m_new = fmaxf(m[qi], mx);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f((m[qi] - m_safe) * c2);
      const float mc = m_safe * c2;
      float psum = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[qi][kf][r] * c2 - mc);
          psum += pv;
          s[qi][kf][r] = pv;
        }
      if (p.thr) {
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const uint32_t y = drop_pair_bits(rkey[qi], (uint32_t)(k0 + kf * 16 + g * 4 + pr * 2) >> 1);
            if ((y & 0xffffu) < p.thr) s[qi][kf][2 * pr] = 0.f;
            if ((y >> 16) < p.thr) s[qi][kf][2 * pr + 1] = 0.f;
          }
      }
      psum = group_sum(psum);
      l[qi] = l[qi] * alpha + psum;
      m[qi] = m_new;
#pragma unroll
      for (int df = 0; df < 4; ++df) o[qi][df] *= alpha;
      pb[qi][0] = pack_p(s[qi], 0);
      pb[qi][1] = pack_p(s[qi], 1);
    }
    // ---- O^T[d][q] += V^T . P^T
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        mma(o[0][df], vt[ms][df], pb[0][ms]);
        mma(o[1][df], vt[ms][df], pb[1][ms]);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

#pragma unroll
  for (int qi = 0; qi < 2; ++qi) {
    const int q = qw + qi * 16 + lr;
    if (q >= p.Tq) continue;
    const float inv_l = l[qi] > 0.f ? p.inv_keep / l[qi] : 0.f;      // the dropout rescale is a constant: applied once here
    if (g == 0) p.lse[((int64_t)b * p.H + h) * p.Tq + q] = l[qi] > 0.f ? m[qi] * p.scale + logf(l[qi]) : INFINITY;
    bf16_t* Ob = static_cast<bf16_t*>(p.Out) + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const f32x4_t v = o[qi][df] * inv_l;
      *reinterpret_cast<uint2*>(Ob + df * 16 + g * 4) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    }
  }
}

bool fast_ok(const AttnArgs& p, int d, int dtype) {
  return dtype == ASR_BF16 && d == HD && p.vec && p.Tq > 0 && p.Tk > 0 && getenv("ASR_ATTN_GENERIC") == nullptr;
}

}  // namespace

int attn_fast_fwd(const AttnArgs& p, int d, int dtype, hipStream_t s) {
  if (!fast_ok(p, d, dtype) || (((uintptr_t)p.Out) & 7) != 0 || p.o_st % 4 != 0 || p.o_sb % 4 != 0) return ASR_EUNSUPPORTED;
  const int nqb = (p.Tq + FQ - 1) / FQ;
  attn_fwd_bf16_d64_kernel<<<dim3((unsigned)(nqb * p.B * p.H)), dim3(256), 0, s>>>(p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

int attn_fast_bwd(const AttnArgs& p, int d, int dtype, hipStream_t s) {
  (void)p; (void)d; (void)dtype; (void)s;
  return ASR_EUNSUPPORTED;
}

}  // namespace asr_attn
