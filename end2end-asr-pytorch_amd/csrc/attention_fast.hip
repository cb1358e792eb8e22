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

// all-reduce over the 4 lane groups that share lane & 15, on the VALU (gfx950 lane-swap instructions) instead of two
// ds_bpermute round trips: v_permlane16_swap(v, v) -> {rows 0,0,2,2} / {rows 1,1,3,3}; v_permlane32_swap(v, v) -> {lo,lo} / {hi,hi}
__device__ __forceinline__ float group_max4(float v) {
  uint32_t u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1])));
  auto c = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ float group_sum4(float v) {
  uint32_t u = __float_as_uint(v);
  auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = __float_as_uint(__uint_as_float(a[0]) + __uint_as_float(a[1]));
  auto c = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}

// XCD-aware linear workgroup id: consecutive ids run on the same XCD (hardware deals blockIdx round-robin over 8 XCDs)
__device__ __forceinline__ int xcd_linear_id(int bid, int nwg) {
  const int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
}
__device__ __forceinline__ int xcd_linear_id() { return xcd_linear_id((int)blockIdx.x, (int)gridDim.x); }

// 64 rows x 128 B of a (rows, 64) bf16 slice -> LDS, chunk c of row r stored at slot c ^ (r & 7).  Rows past `nrows`
// re-read the last valid row (the LDS-DMA cannot zero-fill; such rows are masked / never stored by the callers).
//
// The LDS-DMA is issued BY HAND (inline asm).  Through the builtin the compiler counts these loads itself, and its waitcnt pass
// cannot tell a DMA's LDS write from the tile a later ds_read wants: it put s_waitcnt vmcnt(0) in front of the first LDS read that
// followed the prefetch of the next tile (the P.V operand reads) -- the "double buffering" only ever overlapped the softmax, and
// every tile exposed most of a global-memory round trip.  Hand-issued, nothing waits until the explicit s_waitcnt vmcnt(0) in front
// of the barrier at the end of the tile (every loop below has one); ordinary global loads issued while a DMA is in flight still
// wait for it (vmcnt retires in order), so the loops issue those BEFORE the prefetch.
__device__ __forceinline__ void lds_dma16(unsigned lds_wave_base, const void* src) {
  unsigned keep;      // M0 saved and restored: neutral for whatever the compiler keeps there
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}
// 4 bytes per lane (a row of 64 floats per wave)
__device__ __forceinline__ void lds_dma4(unsigned lds_wave_base, const void* src) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}
// The same with a wave-uniform base address (SGPR pair) + a per-thread 32-bit byte offset that does not change from tile to tile:
// the loops of the forward kernel hoist the offsets (tile_voff) and pay no vector instruction per prefetch (the per-tile 64-bit
// row * stride arithmetic was 24 of the 245 vector instructions of a forward tile).
__device__ __forceinline__ void lds_dma16_s(unsigned lds_wave_base, unsigned voff, const void* sbase) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(voff), "s"(sbase)
               : "memory");
}
// byte offset of this thread's i-th chunk of a tile relative to the tile's first row (row stride st elements; fast_ok(): < 2^22)
__device__ __forceinline__ unsigned tile_voff(int64_t st, int tid, int i) {
  const int c = i * 256 + tid, row = c >> 3, slot = c & 7;
  return (unsigned)(row * (int)st * 2 + ((slot ^ (row & 7)) << 4));
}
__device__ __forceinline__ void stage_tile(unsigned char* lds, const bf16_t* g, int64_t st, int r0, int nrows, int tid, int wave);
// full tiles (r0 + 64 <= nrows) go by base + hoisted offsets, the ragged last tile by the clamped per-row addresses
__device__ __forceinline__ void stage_tile_h(unsigned char* lds, const bf16_t* g, int64_t st, int r0, int nrows, const unsigned (&voff)[2],
                                             int tid, int wave) {
  if (r0 + 64 <= nrows) {
    const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const bf16_t* gb = g + (int64_t)r0 * st;
#pragma unroll
    for (int i = 0; i < 2; ++i) lds_dma16_s(base + (unsigned)((i * 256 + wave * 64) * 16), voff[i], gb);
  } else {
    stage_tile(lds, g, st, r0, nrows, tid, wave);
  }
}
__device__ __forceinline__ void stage_tile(unsigned char* lds, const bf16_t* g, int64_t st, int r0, int nrows, int tid, int wave) {
  const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 256 + tid, row = c >> 3, slot = c & 7;
    int gr = r0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    const bf16_t* src = g + (int64_t)gr * st + ((slot ^ (row & 7)) << 3);
    lds_dma16(base + (unsigned)((i * 256 + wave * 64) * 16), src);
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
  const uint2 lo = asr_lds_read_tr16(tile + row * ROWB + ((chunk ^ (row & 7)) << 4) + half * 8);
  const uint2 hi = asr_lds_read_tr16(tile + (row + 16) * ROWB + ((chunk ^ ((row + 16) & 7)) << 4) + half * 8);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// The same two reads with the LANE part of the address (row within its 16-row fragment, swizzled chunk) computed once per kernel:
// the row of a fragment read is 16 X + lr (rows) or 32 ms + 4 g + (lr >> 2) (+ 16) (columns), so row & 7 -- the swizzle -- does not
// depend on X / ms, and what is left per read is tile base + lane offset + a compile-time constant that fits the DS offset field.
// (Computed per read, the address arithmetic was ~20 of the ~100 vector instructions of a backward half tile.)
struct FragOff {
  unsigned rows[2];          // [ds]
  unsigned cols[4];          // [df]
  __device__ __forceinline__ void init(int lr, int g) {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) rows[ds] = (unsigned)(lr * ROWB + (((ds * 4 + g) ^ (lr & 7)) << 4));
    const int row = 4 * g + (lr >> 2);
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const int chunk = 2 * df + ((lr & 3) >> 1), half = lr & 1;       // column 16 df + 4 (lr & 3): 16-byte chunk, 8-byte half
      cols[df] = (unsigned)(row * ROWB + ((chunk ^ (row & 7)) << 4) + half * 8);
    }
  }
};
__device__ __forceinline__ uint4 frag_rows_at(const unsigned char* tile, unsigned off, int X) {
  return *reinterpret_cast<const uint4*>(tile + off + X * 16 * ROWB);
}
__device__ __forceinline__ uint4 frag_cols_at(const unsigned char* tile, unsigned off, int ms) {
  const uint2 lo = asr_lds_read_tr16(tile + off + 32 * ms * ROWB);
  const uint2 hi = asr_lds_read_tr16(tile + off + (32 * ms + 16) * ROWB);
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
__device__ __forceinline__ f32x4_t mma_z(const uint4& a, const uint4& b) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
__device__ __forceinline__ uint4 load_row16(const bf16_t* base, int64_t st, int row, int nrows, int col0) {
  const int r = row < nrows ? row : nrows - 1;
  return *reinterpret_cast<const uint4*>(base + (int64_t)r * st + col0);
}

// Buffer resource over the whole key-padding mask ((B, Tk) or (B, Tq, Tk) bytes; fast_ok() guarantees < 2^31): raw buffer, stride 0,
// out-of-range reads return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mask_rsrc(const AttnArgs& p) {
  const int64_t n = (int64_t)(p.B - 1) * p.m_sb + (int64_t)(p.Tq - 1) * p.m_sq + p.Tk;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(p.key_pad), 0, (int)n, 0x00020000);
}

// Mask NKF key fragments of raw scores s[qi][kf][r] (key = kbase + 16 kf + 4 g + r, query = q_lane + 16 qi): straight-line
// selects, mask bytes read with clamped (always valid) addresses.
template <int NKF, int NQ = 2>
__device__ __forceinline__ void mask_scores(const AttnArgs& p, f32x4_t (*s)[NKF], int b, int kbase, int g, int q_lane, int kend) {
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = q_lane + qi * 16;
    // every mask byte of this lane is loaded up front, unconditionally (a short-circuit `dead || byte` compiled to one branch, one
    // load and one s_waitcnt vmcnt(0) PER ELEMENT: 16 NQ serialised global round trips per tile) -- as BUFFER loads: one 32-bit
    // offset register + immediates instead of a 64-bit address per byte, and the hardware's range check (bytes past the end of
    // the mask read as 0) instead of per-element clamps; keys >= Tk are dead by kend anyway.
    uint32_t mbits = 0u;
    if (p.key_pad) {
      const int off = (int)((int64_t)b * p.m_sb + (int64_t)(q < p.Tq ? q : p.Tq - 1) * p.m_sq) + kbase + g * 4;
#pragma unroll
      for (int i = 0; i < NKF * 4; ++i)
        mbits |= (uint32_t)(__builtin_amdgcn_raw_buffer_load_b8(mask_rsrc(p), off + (i >> 2) * 16 + (i & 3), 0, 0) != 0) << i;
    }
#pragma unroll
    for (int kf = 0; kf < NKF; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = kbase + kf * 16 + g * 4 + r;
        const bool dead = (kg >= kend) | (p.causal != 0 & kg > q) | (((mbits >> (kf * 4 + r)) & 1u) != 0u);
        s[qi][kf][r] = dead ? -INFINITY : s[qi][kf][r];
      }
  }
}

// ================================================================================================ forward
// workgroup = 4 waves x 16 NQ queries; grid = ceil(Tq / (64 NQ)) * B * H (1-D, XCD-aware).  NQ = 2 (32 queries per wave) for long
// sequences; NQ = 1 for short ones (Tq <= 256, every attention of the benchmark model at T' = 200): measured at (32, 8, 200, 200)
// the tile loop costs 2.1 us per 64-key tile with two waves per SIMD -- one wave's S -> softmax -> P V chain cannot overlap its own
// phases -- against 2.7 us for everything else in the launch; half the queries per wave doubles the waves per SIMD.
constexpr int FQ = 128;

template <int NQ, bool DROP>
__global__ __launch_bounds__(256) void attn_fwd_bf16_d64_kernel(AttnArgs p) {
  constexpr int FQW = 64 * NQ;                               // queries per workgroup
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];      // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.Tq + FQW - 1) / FQW;
  const int vid = xcd_linear_id();
  const int bh = vid / nqb, qb = vid - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int qw = qb * FQW + wave * 16 * NQ;                 // first query of this wave
  const bf16_t* Qb = static_cast<const bf16_t*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const bf16_t* Kb = static_cast<const bf16_t*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const bf16_t* Vb = static_cast<const bf16_t*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  const float c2 = p.scale * LOG2E;
  FragOff fo;
  fo.init(lr, g);

  uint4 qf[NQ][2];
  uint32_t rkey[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) qf[qi][ds] = load_row16(Qb, p.q_st, qw + qi * 16 + lr, p.Tq, ds * 32 + g * 8);
    rkey[qi] = drop_row_key(seed, drop_row(p, b, h, qw + qi * 16 + lr));
  }
  f32x4_t o[NQ][4];
  float m[NQ], l[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    m[qi] = -INFINITY;
    l[qi] = 0.f;
#pragma unroll
    for (int df = 0; df < 4; ++df) o[qi][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  const int kend = key_end(p, b);
  int kstop = kend;
  if (p.causal) kstop = min(kend, qb * FQW + FQW);
  const int ntile = (kstop + 63) >> 6;

  const unsigned voK[2] = {tile_voff(p.k_st, tid, 0), tile_voff(p.k_st, tid, 1)};
  const unsigned voV[2] = {tile_voff(p.v_st, tid, 0), tile_voff(p.v_st, tid, 1)};
  if (ntile > 0) {
    stage_tile_h(smem, Kb, p.k_st, 0, p.Tk, voK, tid, wave);
    stage_tile_h(smem + TILE, Vb, p.v_st, 0, p.Tk, voV, tid, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int k0 = t << 6;
    const unsigned char* sK = smem + (t & 1) * 2 * TILE;
    const unsigned char* sV = sK + TILE;
    // ---- S^T[key][q] = K . Q^T
    f32x4_t s[NQ][4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) s[qi][kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        const uint4 a = frag_rows_at(sK, fo.rows[ds], kf);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) mma(s[qi][kf], a, qf[qi][ds]);
      }
    }
    // ---- masks only where a boundary crosses this tile (wave-uniform test)
    const bool need_mask = (k0 + 64 > kend) || p.key_pad != nullptr || (p.causal && k0 + 63 > qw);
    if (need_mask) mask_scores<4, NQ>(p, s, b, k0, g, qw + lr, kend);
    // next tile: HBM -> LDS in flight under the softmax and the second contraction.  Issued AFTER the mask bytes were
    // consumed: ordinary loads and LDS-DMA loads share vmcnt, and waiting for the former with the latter in flight
    // was observed to return stale mask bytes.
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < ntile) {
      unsigned char* nb = smem + ((t + 1) & 1) * 2 * TILE;
      stage_tile_h(nb, Kb, p.k_st, k0 + 64, p.Tk, voK, tid, wave);
      stage_tile_h(nb + TILE, Vb, p.v_st, k0 + 64, p.Tk, voV, tid, wave);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- online softmax (base 2), dropout, pack P^T as the B operand
    uint4 pb[NQ][2];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float mx = -INFINITY;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qi][kf][r]);
      mx = group_max4(mx);
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
      if (DROP) {                    // compile time: a run-time `if (p.thr)` here cost 8 register moves per query fragment even at p = 0
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const uint32_t y = drop_pair_bits(rkey[qi], (uint32_t)(k0 + kf * 16 + g * 4 + pr * 2) >> 1);
            if ((y & 0xffffu) < p.thr) s[qi][kf][2 * pr] = 0.f;
            if ((y >> 16) < p.thr) s[qi][kf][2 * pr + 1] = 0.f;
          }
      }
      psum = group_sum4(psum);
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
        const uint4 vt = frag_cols_at(sV, fo.cols[df], ms);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) mma(o[qi][df], vt, pb[qi][ms]);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = qw + qi * 16 + lr;
    if (q >= p.Tq) continue;
    const float inv_l = l[qi] > 0.f ? p.inv_keep / l[qi] : 0.f;      // the dropout rescale is a constant: applied once here
    if (g == 0) p.lse[((int64_t)b * p.H + h) * p.Tq + q] = l[qi] > 0.f ? m[qi] * p.scale + logf(l[qi]) : INFINITY;
    bf16_t* Ob = static_cast<bf16_t*>(p.Out) + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const f32x4_t v = o[qi][df] * inv_l;
      *reinterpret_cast<uint2*>(Ob + df * 16 + g * 4) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      if (p.Out32) *reinterpret_cast<f32x4_t*>(p.Out32 + (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD + df * 16 + g * 4) = v;
    }
  }
}

// (A software-pipelined form of the forward -- S(t+1) issued before the softmax of S(t), interleaved instruction by instruction -- was
// measured slower at every shape in round 2, profiles/r02_ab_attn_pipe.txt, and removed in round 3.)

// ================================================================================================ backward: dQ
// Same work split as the forward (wave = 32 queries, 64-key K / V tiles double buffered in LDS):
//   S^T = K Q^T, dP^T = V dO^T (A operands: K / V rows from LDS; B: Q / dO rows in registers)
//   P = exp2(S c - lse c'), dS = P (keep/(1-p) dP - delta)
//   dQ^T[d][q] += K^T[d][key] dS^T[key][q]   (A: transposing reads of the natural K tile; B: dS from the accumulators)
// NQ as in the forward kernel: 16 NQ queries per wave.  In the two backward kernels 16-row waves win at every length (measured:
// (32, 8, 800, 800) p = 0.1 backward 324 -> 301 us, (16, 8, 795, 795) 185 -> 166 us; the forward loses 4 - 9 % there), so they are the default.
template <int NQ>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnArgs& p, const int vid, unsigned char* smem) {
  constexpr int FQW = 64 * NQ;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = (p.Tq + FQW - 1) / FQW;
  const int bh = vid / nqb, qb = vid - bh * nqb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int qw = qb * FQW + wave * 16 * NQ;
  const bf16_t* Qb = static_cast<const bf16_t*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const bf16_t* Kb = static_cast<const bf16_t*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const bf16_t* Vb = static_cast<const bf16_t*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const bf16_t* dOb = static_cast<const bf16_t*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)h * HD;
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  const float c2 = p.scale * LOG2E;
  const float keep_prob = p.thr ? 1.f / p.inv_keep : 1.f, out_scale = p.thr ? p.scale * p.inv_keep : p.scale;
  const uint32_t thr_hi = p.thr << 16;            // field >= thr  <=>  (field << 16) >= thr_hi (thr <= 0xffff)
  const unsigned voK[2] = {tile_voff(p.k_st, tid, 0), tile_voff(p.k_st, tid, 1)};
  const unsigned voV[2] = {tile_voff(p.v_st, tid, 0), tile_voff(p.v_st, tid, 1)};
  FragOff fo;
  fo.init(lr, g);

  uint4 qf[NQ][2], dof[NQ][2];
  uint32_t rkey[NQ];
  float lse2[NQ], dlt[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = qw + qi * 16 + lr;
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      qf[qi][ds] = load_row16(Qb, p.q_st, q, p.Tq, ds * 32 + g * 8);
      dof[qi][ds] = load_row16(dOb, p.o_st, q, p.Tq, ds * 32 + g * 8);
    }
    // first hash stage of the dropout mask as a running sum: (row key + key pair) * DROP_C1 is linear modulo 2^32, the lane part
    // (row, 2 g) is multiplied once here, a tile adds 32 * DROP_C1, the (half, fragment, pair) part is a literal
    rkey[qi] = (drop_row_key(seed, drop_row(p, b, h, q)) + 2u * (uint32_t)g) * DROP_C1;
    const int64_t si = ((int64_t)b * p.H + h) * p.Tq + (q < p.Tq ? q : p.Tq - 1);
    lse2[qi] = p.lse[si] * LOG2E;                 // +inf for fully masked rows: exp2(-inf) = 0
    // dS = P (keep / (1 - p) dP - delta) = 1 / (1 - p) * P (keep dP - (1 - p) delta): the rescale leaves the loop (it joins p.scale
    // in the epilogue), the mask becomes a plain select of dP
    dlt[qi] = p.delta[si] * keep_prob;
  }
  f32x4_t dq[NQ][4];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
    for (int df = 0; df < 4; ++df) dq[qi][df] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int kend = key_end(p, b);
  int kstop = kend;
  if (p.causal) kstop = min(kend, qb * FQW + FQW);
  const int ntile = (kstop + 63) >> 6;

  if (ntile > 0) {
    stage_tile_h(smem, Kb, p.k_st, 0, p.Tk, voK, tid, wave);
    stage_tile_h(smem + TILE, Vb, p.v_st, 0, p.Tk, voV, tid, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int k0 = t << 6;
    const unsigned char* sK = smem + (t & 1) * 2 * TILE;
    const unsigned char* sV = sK + TILE;
    const bool need_mask = (k0 + 64 > kend) || p.key_pad != nullptr || (p.causal && k0 + 63 > qw);
    // the tile is consumed in two halves of 32 keys (one macro step of the dQ contraction each): half the live accumulators
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      f32x4_t s[NQ][2], dp[NQ][2];
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
          s[qi][k2] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          dp[qi][k2] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          const uint4 ak = frag_rows_at(sK, fo.rows[ds], 2 * ms + k2);
          const uint4 av = frag_rows_at(sV, fo.rows[ds], 2 * ms + k2);
#pragma unroll
          for (int qi = 0; qi < NQ; ++qi) {
            mma(s[qi][k2], ak, qf[qi][ds]);
            mma(dp[qi][k2], av, dof[qi][ds]);
          }
        }
      }
      if (need_mask) mask_scores<2, NQ>(p, s, b, k0 + 32 * ms, g, qw + lr, kend);
      if (ms == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntile) {
          unsigned char* nb = smem + ((t + 1) & 1) * 2 * TILE;
          stage_tile_h(nb, Kb, p.k_st, k0 + 64, p.Tk, voK, tid, wave);
          stage_tile_h(nb + TILE, Vb, p.v_st, k0 + 64, p.Tk, voV, tid, wave);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      uint4 pb[NQ];
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        if (p.thr) {
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const uint32_t y = drop_pair_mix(rkey[qi] + (uint32_t)((2 * ms + k2) * 8 + pr) * DROP_C1);
              dp[qi][k2][2 * pr] = (y << 16) < thr_hi ? 0.f : dp[qi][k2][2 * pr];
              dp[qi][k2][2 * pr + 1] = y < thr_hi ? 0.f : dp[qi][k2][2 * pr + 1];
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int r = 0; r < 4; r += 2) {      // two scores per packed instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32)
            const f32x2_t e = f32x2_t{s[qi][k2][r], s[qi][k2][r + 1]} * c2 - lse2[qi];       // masked: s = -inf -> exp2 = 0
            const f32x2_t pv = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            const f32x2_t ds = pv * (f32x2_t{dp[qi][k2][r], dp[qi][k2][r + 1]} - dlt[qi]);
            s[qi][k2][r] = ds[0];
            s[qi][k2][r + 1] = ds[1];
          }
        pb[qi] = pack_p(s[qi], 0);
      }
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const uint4 kt = frag_cols_at(sK, fo.cols[df], ms);
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) mma(dq[qi][df], kt, pb[qi]);
      }
    }
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) rkey[qi] += 32u * DROP_C1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const int q = qw + qi * 16 + lr;
    if (q >= p.Tq) continue;
    bf16_t* o = static_cast<bf16_t*>(p.dQ) + (int64_t)b * p.q_sb + (int64_t)q * p.q_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const f32x4_t v = dq[qi][df] * out_scale;
      *reinterpret_cast<uint2*>(o + df * 16 + g * 4) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    }
  }
}

// ================================================================================================ backward: dK, dV
// wave = 32 keys (two B fragments, K / V rows in registers); 64-query Q / dO tiles double buffered in LDS with their
// lse / delta rows.   S = Q K^T, dP = dO V^T (A: Q / dO rows from LDS), P, dS as above, then
//   dV^T[d][key] += dO^T[d][q] Pd[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]   (A: transposing reads of dO / Q tiles)
// NK: 16 NK keys per wave (NK = 1 for Tk <= 256, as in the other two kernels).
template <int NK>
__device__ __forceinline__ void attn_bwd_dkv_body(const AttnArgs& p, const int vid, unsigned char* smem, float (*s_stat)[2][64]) {
  constexpr int FKW = 64 * NK;                               // keys per workgroup;  s_stat: [buffer][lse*log2e | delta][q]
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nkb = (p.Tk + FKW - 1) / FKW;
  const int bh = vid / nkb, kb = vid - bh * nkb;
  const int b = bh / p.H, h = bh - b * p.H;
  const int kw = kb * FKW + wave * 16 * NK;                 // first key of this wave
  const bf16_t* Qb = static_cast<const bf16_t*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const bf16_t* Kb = static_cast<const bf16_t*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const bf16_t* Vb = static_cast<const bf16_t*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const bf16_t* dOb = static_cast<const bf16_t*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)h * HD;
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  const float c2 = p.scale * LOG2E;
  const int kend = key_end(p, b);
  const int64_t stat0 = ((int64_t)b * p.H + h) * p.Tq;
  const uint32_t field_sh = (uint32_t)(lr & 1) << 4;        // this lane's keys are kw + 16 ki + lr with kw a multiple of 16: key & 1 = lr & 1
  // (no hoisted DMA offsets / column-fragment addresses here, unlike the dQ half: this half has to fit 128 registers for the fourth wave per
  //  SIMD, and its loop is bound by the dependent chain of a wave, not by the vector instruction count -- profiles/r04_attention_bwd_ablation.txt)
  FragOff fo;
  fo.init(lr, g);

  uint4 kfr[NK][2], vfr[NK][2];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki)
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      kfr[ki][ds] = load_row16(Kb, p.k_st, kw + ki * 16 + lr, p.Tk, ds * 32 + g * 8);
      vfr[ki][ds] = load_row16(Vb, p.v_st, kw + ki * 16 + lr, p.Tk, ds * 32 + g * 8);
    }
  f32x4_t dk[NK][4], dv[NK][4];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki)
#pragma unroll
    for (int df = 0; df < 4; ++df) { dk[ki][df] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[ki][df] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  // causal: query tiles that end before the first key of this workgroup never see it
  const int t0 = p.causal ? (kb * FKW) >> 6 : 0;
  const int ntile = (p.Tq + 63) >> 6;
  // first hash stage of the dropout mask, (row key + key pair) * DROP_C1, as a running sum (linear modulo 2^32; the row key is linear
  // in the row, + DROP_ROW_C per row): the lane part (row 4 g + 2 (lr & 1) of tile t0, this lane's key pair) is multiplied once, a
  // tile adds 64 rows, the (half, fragment) part and the second row of the lane are literals
  constexpr uint32_t DROP_ROW_C = 0x9E3779B1u;
  uint32_t ybase[NK];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki)
    ybase[ki] = (drop_row_key(seed, drop_row(p, b, h, (t0 << 6) + g * 4 + 2 * (lr & 1))) + ((uint32_t)(kw + ki * 16 + lr) >> 1)) * DROP_C1;

  // lse / delta of query tile t (64 floats each).  Full tiles: part of the tile's DMA request (wave 0 fetches the lse row, wave 1
  // the delta row, 4 bytes per lane) -- as ordinary loads stored to LDS by the threads, every tile exposed a global-memory round
  // trip: the value has to be CONSUMED before the hand-issued DMA of that tile goes out (vmcnt retires in order: waiting for it
  // later would wait for the DMA as well), i.e. half a tile after it was requested.  The ragged last tile (rows past Tq must read
  // lse = +inf: P = exp2(-inf) = 0) and the first tile keep the load / store path.
  auto load_stat = [&](int t) __attribute__((always_inline)) -> float {
    const int ql = tid & 63, qq = (t << 6) + ql;
    if (tid < 64) return qq < p.Tq ? p.lse[stat0 + qq] : INFINITY;
    return (tid < 128 && qq < p.Tq) ? p.delta[stat0 + qq] : 0.f;
  };
  auto dma_stats = [&](int t, int buf) __attribute__((always_inline)) {
    if (wave < 2) {
      const float* src = (wave == 0 ? p.lse : p.delta) + stat0 + (t << 6) + lane;
      lds_dma4((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)&s_stat[buf][wave][0], src);
    }
  };
  auto stage_stats = [&](int t, int buf) __attribute__((always_inline)) {
    const float v = load_stat(t);
    if (tid < 128) s_stat[buf][tid >> 6][tid & 63] = v;
  };
  if (t0 < ntile) {
    stage_stats(t0, t0 & 1);
    stage_tile(smem, Qb, p.q_st, t0 << 6, p.Tq, tid, wave);
    stage_tile(smem + TILE, dOb, p.o_st, t0 << 6, p.Tq, tid, wave);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = t0; t < ntile; ++t) {
    const int q0 = t << 6;
    const unsigned char* sQ = smem + ((t - t0) & 1) * 2 * TILE;
    const unsigned char* sdO = sQ + TILE;
    const float* st_lse = s_stat[t & 1][0];
    const float* st_dlt = s_stat[t & 1][1];
    const bool need_mask = p.key_pad != nullptr || (p.causal && kw + 16 * NK - 1 > q0) || (kw + 16 * NK > kend);
    const bool next_ragged = q0 + 128 > p.Tq;
    const float next_stat = (t + 1 < ntile && next_ragged) ? load_stat(t + 1) : 0.f;
    // the query tile is consumed in two halves of 32 queries (one macro step of the dV / dK contractions each)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      f32x4_t s[NK][2], dp[NK][2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
        for (int ki = 0; ki < NK; ++ki) {
          s[ki][q2] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          dp[ki][q2] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          const uint4 aq = frag_rows_at(sQ, fo.rows[ds], 2 * ms + q2);
          const uint4 ao = frag_rows_at(sdO, fo.rows[ds], 2 * ms + q2);
#pragma unroll
          for (int ki = 0; ki < NK; ++ki) {
            mma(s[ki][q2], aq, kfr[ki][ds]);
            mma(dp[ki][q2], ao, vfr[ki][ds]);
          }
        }
      }
      // per-row statistics of the 8 queries this lane sees in this half: q = q0 + 32 ms + 16 q2 + 4 g + r
      f32x4_t lse2[2], dlt[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        lse2[q2] = *reinterpret_cast<const f32x4_t*>(st_lse + (2 * ms + q2) * 16 + g * 4) * LOG2E;
        dlt[q2] = *reinterpret_cast<const f32x4_t*>(st_dlt + (2 * ms + q2) * 16 + g * 4);
      }
      if (ms == 0) {
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < ntile) {
          unsigned char* nb = smem + ((t + 1 - t0) & 1) * 2 * TILE;
          if (!next_ragged) dma_stats(t + 1, (t + 1) & 1);
          else if (tid < 128) s_stat[(t + 1) & 1][tid >> 6][tid & 63] = next_stat;
          stage_tile(nb, Qb, p.q_st, q0 + 64, p.Tq, tid, wave);
          stage_tile(nb + TILE, dOb, p.o_st, q0 + 64, p.Tq, tid, wave);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      uint4 pa[NK], pd[NK];
#pragma unroll
      for (int ki = 0; ki < NK; ++ki) {
        const int key = kw + ki * 16 + lr;
        if (need_mask) {
          uint32_t mbits = 0u;                      // mask bytes loaded up front, unconditionally (see mask_scores)
          if (p.key_pad) {
            const int col = (int)((int64_t)b * p.m_sb) + (key < p.Tk ? key : p.Tk - 1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int qq = q0 + (2 * ms + (i >> 2)) * 16 + g * 4 + (i & 3);
              mbits |= (uint32_t)(__builtin_amdgcn_raw_buffer_load_b8(mask_rsrc(p), col + (qq < p.Tq ? qq : p.Tq - 1) * (int)p.m_sq, 0, 0) != 0) << i;
            }
          }
#pragma unroll
          for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int qq = q0 + (2 * ms + q2) * 16 + g * 4 + r;
              const bool dead = (key >= kend) | (p.causal != 0 & key > qq) | (((mbits >> (q2 * 4 + r)) & 1u) != 0u);
              s[ki][q2][r] = dead ? -INFINITY : s[ki][q2][r];
            }
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          // Dropout mask of this lane's 4 consecutive query rows at ONE key.  The hash word of (row, key pair) holds the fields of keys
          // 2j and 2j + 1, i.e. of this lane and of lane ^ 1: the even lane of a pair hashes rows 0 and 1, the odd lane rows 2 and 3,
          // and both read all four words through quad permutes -- two hashes and four v_mov_dpp per lane instead of four hashes
          // (the row key is linear in the row: + 0x9E3779B1 per row, attention.h).
          uint32_t yw[4] = {0u, 0u, 0u, 0u};
          if (p.thr) {
            const uint32_t ya = drop_pair_mix(ybase[ki] + (uint32_t)((2 * ms + q2) * 16) * DROP_ROW_C * DROP_C1);
            const uint32_t yb = drop_pair_mix(ybase[ki] + (uint32_t)((2 * ms + q2) * 16 + 1) * DROP_ROW_C * DROP_C1);
            yw[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)ya, 0xA0, 0xf, 0xf, true);        // quad_perm [0,0,2,2]: the pair's even lane
            yw[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xA0, 0xf, 0xf, true);
            yw[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)ya, 0xF5, 0xf, 0xf, true);        // quad_perm [1,1,3,3]: the pair's odd lane
            yw[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xF5, 0xf, 0xf, true);
          }
#pragma unroll
          for (int r = 0; r < 4; r += 2) {      // two scores per packed instruction
            const f32x2_t e = f32x2_t{s[ki][q2][r], s[ki][q2][r + 1]} * c2 - f32x2_t{lse2[q2][r], lse2[q2][r + 1]};
            const f32x2_t pv = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            // field of key & 1 (v_bfe_u32).  No test of p.thr here: with dropout off thr = 0 never exceeds a field and inv_keep = 1
            const f32x2_t keepf = {((yw[r] >> field_sh) & 0xffffu) < p.thr ? 0.f : p.inv_keep,
                                   ((yw[r + 1] >> field_sh) & 0xffffu) < p.thr ? 0.f : p.inv_keep};
            const f32x2_t pd = pv * keepf;                                                     // dropped / rescaled probabilities (for dV)
            const f32x2_t ds = pv * (f32x2_t{dp[ki][q2][r], dp[ki][q2][r + 1]} * keepf - f32x2_t{dlt[q2][r], dlt[q2][r + 1]});   // dS (for dK)
            s[ki][q2][r] = pd[0];
            s[ki][q2][r + 1] = pd[1];
            dp[ki][q2][r] = ds[0];
            dp[ki][q2][r + 1] = ds[1];
          }
        }
        pa[ki] = pack_p(s[ki], 0);
        pd[ki] = pack_p(dp[ki], 0);
      }
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const uint4 aot = frag_cols(sdO, df * 16, ms, lr, g);        // (addresses computed per read: this half lives at 128 registers)
        const uint4 aqt = frag_cols(sQ, df * 16, ms, lr, g);
#pragma unroll
        for (int ki = 0; ki < NK; ++ki) {
          mma(dv[ki][df], aot, pa[ki]);
          mma(dk[ki][df], aqt, pd[ki]);
        }
      }
    }
#pragma unroll
    for (int ki = 0; ki < NK; ++ki) ybase[ki] += 64u * DROP_ROW_C * DROP_C1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int ki = 0; ki < NK; ++ki) {
    const int key = kw + ki * 16 + lr;
    if (key >= p.Tk) continue;
    bf16_t* ok = static_cast<bf16_t*>(p.dK) + (int64_t)b * p.k_sb + (int64_t)key * p.k_st + (int64_t)h * HD;
    bf16_t* ov = static_cast<bf16_t*>(p.dV) + (int64_t)b * p.v_sb + (int64_t)key * p.v_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const f32x4_t a = dk[ki][df] * p.scale, c = dv[ki][df];
      *reinterpret_cast<uint2*>(ok + df * 16 + g * 4) = make_uint2(pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]));
      *reinterpret_cast<uint2*>(ov + df * 16 + g * 4) = make_uint2(pack_bf16(c[0], c[1]), pack_bf16(c[2], c[3]));
    }
  }
}

// ---- short key sequences (Tk <= 256: every attention of the benchmark model, T' = 200 / 100): the WHOLE backward of one (batch, head) in
// one workgroup and ONE pass over the scores (round 5).  The two-half form above recomputes S, the softmax, the dropout mask and dP in
// both halves (the dQ workgroups and the dK / dV workgroups): 7 matrix products and two softmax passes per (query, key) pair, at 12.7
// vector instructions per MFMA the launch is bound by that vector work.  Here 8 waves own 32 keys each (all 256 of the head), walk the
// query tiles once as the dK / dV half does -- S^T, P^T, dP^T, dS^T in registers, dK / dV accumulated in registers -- and the data
// gradient of the queries comes from the same dS: every wave writes its dS^T rows (key-major, the standard 64 x 64 tile image) to LDS,
// and after one barrier the 8 waves contract dQ^T[d][q] = K^T[d][key] dS^T[key][q] over all 256 keys, 2 of the 16 output fragments each
// (both operands by transposing reads: K staged once per workgroup).  5 matrix products and one softmax pass per pair; no atomics.
template <int NK, int NW>
__device__ __forceinline__ void attn_bwd_fused_body(const AttnArgs& p, const int bh, unsigned char* smem) {
  // 16 NK keys per wave, NW waves: (2, 8) = 256 keys; (1, 8) = 128 keys (the decoder's self-attention: every wave has live keys);
  // (1, 16) = 256 keys on 1024 threads (<= 128 registers: a few spills, whose reloads wait for every load in flight -- measured, not used)
  constexpr int KT = NK * NW / 4, NQF = 16 / NW;             // key tiles of 64; dQ fragments per wave
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int htid = tid & 255, hwave = wave & 3, grp = wave >> 2;            // staging is done by groups of 4 waves, one tile each
  constexpr int NGRP = NW / 4;
  const int b = bh / p.H, h = bh - b * p.H;
  const int kw = wave * 16 * NK;
  // LDS: [Q | dO] x 2 buffers, K (4 tiles), dS (4 tiles), lse / delta rows x 2 buffers
  unsigned char* sK = smem + 4 * TILE;
  unsigned char* sdS = smem + 8 * TILE;
  float (*s_stat)[2][64] = reinterpret_cast<float (*)[2][64]>(smem + 12 * TILE);
  const bf16_t* Qb = static_cast<const bf16_t*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const bf16_t* Kb = static_cast<const bf16_t*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const bf16_t* Vb = static_cast<const bf16_t*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  const bf16_t* dOb = static_cast<const bf16_t*>(p.dO) + (int64_t)b * p.o_sb + (int64_t)h * HD;
  const uint64_t seed = asr_mix_seed(p.seed, p.seed_dev);
  const float c2 = p.scale * LOG2E;
  const int kend = key_end(p, b);
  const int64_t stat0 = ((int64_t)b * p.H + h) * p.Tq;
  const uint32_t field_sh = (uint32_t)(lr & 1) << 4;
  FragOff fo;
  fo.init(lr, g);

  uint4 kfr[NK][2], vfr[NK][2];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki)
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      kfr[ki][ds] = load_row16(Kb, p.k_st, kw + ki * 16 + lr, p.Tk, ds * 32 + g * 8);
      vfr[ki][ds] = load_row16(Vb, p.v_st, kw + ki * 16 + lr, p.Tk, ds * 32 + g * 8);
    }
  f32x4_t dk[NK][4], dv[NK][4];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki)
#pragma unroll
    for (int df = 0; df < 4; ++df) { dk[ki][df] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[ki][df] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  const int ntile = (p.Tq + 63) >> 6, nkt = (p.Tk + 63) >> 6;
  constexpr uint32_t DROP_ROW_C = 0x9E3779B1u;
  uint32_t ybase[NK];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki)
    ybase[ki] = (drop_row_key(seed, drop_row(p, b, h, g * 4 + 2 * (lr & 1))) + ((uint32_t)(kw + ki * 16 + lr) >> 1)) * DROP_C1;
  // where this lane's dS^T values go: row = its key inside the key tile, the 4 consecutive queries 32 ms + 16 q2 + 4 g .. + 3 = 8 bytes
  unsigned ds_row[NK];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki) {
    const int key = kw + ki * 16 + lr, row = key & 63;
    ds_row[ki] = (unsigned)((key >> 6) * TILE + row * ROWB + (g & 1) * 8) | ((unsigned)(row & 7) << 28);      // swizzle key in the top bits
  }

  // a key that is dead for EVERY query (past the key length, or padded in a (B, Tk) mask: m_sq == 0) is known once per kernel; what is
  // left per element is the causal comparison in the tiles on the diagonal, and the bytes of a (B, Tq, Tk) mask
  const bool mask3d = p.key_pad != nullptr && p.m_sq != 0;
  bool kd[NK];
#pragma unroll
  for (int ki = 0; ki < NK; ++ki) {
    const int key = kw + ki * 16 + lr;
    kd[ki] = key >= kend;
    if (p.key_pad != nullptr && p.m_sq == 0) kd[ki] |= key < p.Tk && p.key_pad[(int64_t)b * p.m_sb + key] != 0;
  }
  auto load_stat = [&](int t) __attribute__((always_inline)) -> float {
    const int ql = tid & 63, qq = (t << 6) + ql;
    if (tid < 64) return qq < p.Tq ? p.lse[stat0 + qq] : INFINITY;
    return (tid < 128 && qq < p.Tq) ? p.delta[stat0 + qq] : 0.f;
  };
  auto stage_stats = [&](int t, int buf) __attribute__((always_inline)) {
    const float v = load_stat(t);
    if (tid < 128) s_stat[buf][tid >> 6][tid & 63] = v;
  };
  auto stage_q_do = [&](unsigned char* buf, int q0) __attribute__((always_inline)) {
    if (grp == 0) stage_tile(buf, Qb, p.q_st, q0, p.Tq, htid, hwave);
    else if (grp == 1) stage_tile(buf + TILE, dOb, p.o_st, q0, p.Tq, htid, hwave);
  };
  // K (all keys of the head; rows past Tk repeat the last key: their dS is 0) for the dQ contraction, the first query tile, its statistics
#pragma unroll
  for (int j = 0; j < (KT + NGRP - 1) / NGRP; ++j)
    if ((KT + NGRP - 1) / NGRP * grp + j < KT)
      stage_tile(sK + ((KT + NGRP - 1) / NGRP * grp + j) * TILE, Kb, p.k_st, ((KT + NGRP - 1) / NGRP * grp + j) * 64, p.Tk, htid, hwave);
  // dS^T rows of key fragments that are dead for a whole query tile (past the key length; causal: every key after the tile's last query)
  // are never written: zero once (a causal fragment only ever goes from dead to live as the query tiles advance)
#pragma unroll
  for (int i = 0; i < KT * TILE / (16 * 64 * NW); ++i)
    *reinterpret_cast<uint4*>(sdS + (size_t)(i * 64 * NW + tid) * 16) = make_uint4(0u, 0u, 0u, 0u);
  stage_stats(0, 0);
  stage_q_do(smem, 0);
  // lse / delta of the tile after next travel in a register for a whole tile: no global round trip is ever waited for inside a tile
  float stat_next = ntile > 1 ? load_stat(1) : 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int q0 = t << 6;
    const unsigned char* sQ = smem + (t & 1) * 2 * TILE;
    const unsigned char* sdO = sQ + TILE;
    const float* st_lse = s_stat[t & 1][0];
    const float* st_dlt = s_stat[t & 1][1];
    bool live[NK], need_mask[NK];
#pragma unroll
    for (int ki = 0; ki < NK; ++ki) {
      const int k0 = kw + ki * 16;
      live[ki] = k0 < kend && !(p.causal && k0 > q0 + 63);
      need_mask[ki] = p.key_pad != nullptr || (p.causal && k0 + 15 > q0) || (k0 + 16 > kend);     // (a 2-D pad mask: any lane may be dead)
    }
    // the tile's key-padding bytes, both halves, requested first and packed before the next tile's DMA goes out: the waits the compiler
    // puts in front of their use count only the loads it knows -- behind the hand-issued DMA they waited for its round trip as well
    uint32_t mb[2][NK];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int ki = 0; ki < NK; ++ki) {
        mb[ms][ki] = 0u;
        if (mask3d) {
          const int key = kw + ki * 16 + lr;
          const int col = (int)((int64_t)b * p.m_sb) + (key < p.Tk ? key : p.Tk - 1);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int qq = q0 + (2 * ms + (i >> 2)) * 16 + g * 4 + (i & 3);
            mb[ms][ki] |= (uint32_t)(__builtin_amdgcn_raw_buffer_load_b8(mask_rsrc(p), col + (qq < p.Tq ? qq : p.Tq - 1) * (int)p.m_sq, 0, 0) != 0) << i;
          }
        }
      }
    const float next_stat = stat_next;                        // tile t + 1's, requested a tile ago
    if (t + 2 < ntile) stat_next = load_stat(t + 2);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      f32x4_t s[NK][2], dp[NK][2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          const uint4 aq = frag_rows_at(sQ, fo.rows[ds], 2 * ms + q2);
          const uint4 ao = frag_rows_at(sdO, fo.rows[ds], 2 * ms + q2);
#pragma unroll
          for (int ki = 0; ki < NK; ++ki) {
            if (!live[ki]) continue;
            if (ds == 0) {                                      // the first product takes a literal 0 as its addend: no zeroing moves
              s[ki][q2] = mma_z(aq, kfr[ki][ds]);
              dp[ki][q2] = mma_z(ao, vfr[ki][ds]);
            } else {
              mma(s[ki][q2], aq, kfr[ki][ds]);
              mma(dp[ki][q2], ao, vfr[ki][ds]);
            }
          }
        }
      }
      f32x4_t lse2[2], dlt[2];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        lse2[q2] = *reinterpret_cast<const f32x4_t*>(st_lse + (2 * ms + q2) * 16 + g * 4) * LOG2E;
        dlt[q2] = *reinterpret_cast<const f32x4_t*>(st_dlt + (2 * ms + q2) * 16 + g * 4);
      }
      if (ms == 0 && t + 1 < ntile) {            // the next query tile and its statistics into the other buffers
#pragma unroll
        for (int ki = 0; ki < NK; ++ki) asm volatile("" : "+v"(mb[0][ki]), "+v"(mb[1][ki]));      // packed (= loaded) by here
        __builtin_amdgcn_sched_barrier(0);
        if (tid < 128) s_stat[(t + 1) & 1][tid >> 6][tid & 63] = next_stat;
        stage_q_do(smem + ((t + 1) & 1) * 2 * TILE, q0 + 64);
        __builtin_amdgcn_sched_barrier(0);
      }
      uint4 pa[NK], pd[NK];
#pragma unroll
      for (int ki = 0; ki < NK; ++ki) {
        if (!live[ki]) continue;
        const int key = kw + ki * 16 + lr;
        if (need_mask[ki]) {
          if (mask3d || (p.causal && kw + ki * 16 + 15 > q0)) {
            const uint32_t mbits = mb[ms][ki];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int qq = q0 + (2 * ms + q2) * 16 + g * 4 + r;
                const bool dead = kd[ki] | (p.causal != 0 & key > qq) | (((mbits >> (q2 * 4 + r)) & 1u) != 0u);
                s[ki][q2][r] = dead ? -INFINITY : s[ki][q2][r];
              }
          } else {                                              // the lane's key is dead for all 8 queries or for none
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
              for (int r = 0; r < 4; ++r) s[ki][q2][r] = kd[ki] ? -INFINITY : s[ki][q2][r];
          }
        }
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
          uint32_t yw[4] = {0u, 0u, 0u, 0u};
          if (p.thr) {
            const uint32_t ya = drop_pair_mix(ybase[ki] + (uint32_t)((2 * ms + q2) * 16) * DROP_ROW_C * DROP_C1);
            const uint32_t yb = drop_pair_mix(ybase[ki] + (uint32_t)((2 * ms + q2) * 16 + 1) * DROP_ROW_C * DROP_C1);
            yw[0] = (uint32_t)__builtin_amdgcn_mov_dpp((int)ya, 0xA0, 0xf, 0xf, true);
            yw[1] = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xA0, 0xf, 0xf, true);
            yw[2] = (uint32_t)__builtin_amdgcn_mov_dpp((int)ya, 0xF5, 0xf, 0xf, true);
            yw[3] = (uint32_t)__builtin_amdgcn_mov_dpp((int)yb, 0xF5, 0xf, 0xf, true);
          }
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2_t e = f32x2_t{s[ki][q2][r], s[ki][q2][r + 1]} * c2 - f32x2_t{lse2[q2][r], lse2[q2][r + 1]};
            const f32x2_t pv = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
            const f32x2_t keepf = {((yw[r] >> field_sh) & 0xffffu) < p.thr ? 0.f : p.inv_keep,
                                   ((yw[r + 1] >> field_sh) & 0xffffu) < p.thr ? 0.f : p.inv_keep};
            const f32x2_t pdv = pv * keepf;
            const f32x2_t dsv = pv * (f32x2_t{dp[ki][q2][r], dp[ki][q2][r + 1]} * keepf - f32x2_t{dlt[q2][r], dlt[q2][r + 1]});
            s[ki][q2][r] = pdv[0];
            s[ki][q2][r + 1] = pdv[1];
            dp[ki][q2][r] = dsv[0];
            dp[ki][q2][r + 1] = dsv[1];
          }
        }
        pa[ki] = pack_p(s[ki], 0);
        pd[ki] = pack_p(dp[ki], 0);
        // dS^T rows of this wave for the dQ contraction: queries 32 ms + 4 g .. + 3 (q2 = 0) and + 16 (q2 = 1) of key row `key`
        const unsigned base = ds_row[ki] & 0x0fffffffu, rk = ds_row[ki] >> 28;
        *reinterpret_cast<uint2*>(sdS + base + (((unsigned)(4 * ms + (g >> 1)) ^ rk) << 4)) = make_uint2(pd[ki].x, pd[ki].y);
        *reinterpret_cast<uint2*>(sdS + base + (((unsigned)(4 * ms + 2 + (g >> 1)) ^ rk) << 4)) = make_uint2(pd[ki].z, pd[ki].w);
      }
#pragma unroll
      for (int df = 0; df < 4; ++df) {
        const uint4 aot = frag_cols(sdO, df * 16, ms, lr, g);
        const uint4 aqt = frag_cols(sQ, df * 16, ms, lr, g);
#pragma unroll
        for (int ki = 0; ki < NK; ++ki) {
          if (!live[ki]) continue;
          mma(dv[ki][df], aot, pa[ki]);
          mma(dk[ki][df], aqt, pd[ki]);
        }
      }
    }
#pragma unroll
    for (int ki = 0; ki < NK; ++ki) ybase[ki] += 64u * DROP_ROW_C * DROP_C1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // every wave's dS^T rows of this query tile are in LDS
    asm volatile("" ::: "memory");
    {
      // dQ^T[d][q] of this tile: wave -> d fragment (wave & 3) and query fragments NK (wave >> 2) ..; contraction over the key tiles
      const int df = wave & 3, qf0 = NQF * (wave >> 2);
      f32x4_t dq[NQF];
#pragma unroll
      for (int j = 0; j < NQF; ++j) dq[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      // key tiles that are dead for this whole query tile add nothing (their dS^T rows are zero): the live range ends at the key
      // length, and under the causal mask at the tile's last query
      const int klive = p.causal ? min(kend, q0 + 64) : kend;
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        if (kt * 64 >= klive) break;
        uint4 akt[2], bdS[2][NQF];
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
          akt[ms] = frag_cols(sK + kt * TILE, df * 16, ms, lr, g);
#pragma unroll
          for (int j = 0; j < NQF; ++j) bdS[ms][j] = frag_cols(sdS + kt * TILE, (qf0 + j) * 16, ms, lr, g);
        }
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
          for (int j = 0; j < NQF; ++j) mma(dq[j], akt[ms], bdS[ms][j]);
      }
      // the next tile's DMA (and the statistics behind it) has landed -- awaited HERE, before this tile's stores go out: vmcnt retires
      // in order, a wait behind the stores would wait for their round trip as well (it did: 8 us per query tile instead of 4)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < NQF; ++j) {
        const int q = q0 + (qf0 + j) * 16 + lr;
        if (q < p.Tq) {
          bf16_t* oq = static_cast<bf16_t*>(p.dQ) + (int64_t)b * p.q_sb + (int64_t)q * p.q_st + (int64_t)h * HD + df * 16 + g * 4;
          const f32x4_t a = dq[j] * p.scale;
          *reinterpret_cast<uint2*>(oq) = make_uint2(pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]));
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                              // next tile landed (every wave waited above); everybody is done with this tile's dS^T rows
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int ki = 0; ki < NK; ++ki) {
    const int key = kw + ki * 16 + lr;
    if (key >= p.Tk) continue;
    bf16_t* ok = static_cast<bf16_t*>(p.dK) + (int64_t)b * p.k_sb + (int64_t)key * p.k_st + (int64_t)h * HD;
    bf16_t* ov = static_cast<bf16_t*>(p.dV) + (int64_t)b * p.v_sb + (int64_t)key * p.v_st + (int64_t)h * HD;
#pragma unroll
    for (int df = 0; df < 4; ++df) {
      const f32x4_t a = dk[ki][df] * p.scale, c = dv[ki][df];
      *reinterpret_cast<uint2*>(ok + df * 16 + g * 4) = make_uint2(pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]));
      *reinterpret_cast<uint2*>(ov + df * 16 + g * 4) = make_uint2(pack_bf16(c[0], c[1]), pack_bf16(c[2], c[3]));
    }
  }
}
constexpr int FUSED_BWD_LDS = 12 * TILE + 2 * 2 * 64 * 4;
template <int NK, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_fused_bf16_d64_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsmem[];
  attn_bwd_fused_body<NK, NW>(p, xcd_linear_id(), fsmem);
}

template <int NQ>
__global__ __launch_bounds__(256) void attn_bwd_dq_bf16_d64_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];
  attn_bwd_dq_body<NQ>(p, xcd_linear_id(), smem);
}
template <int NK>
__global__ __launch_bounds__(256, NK == 1 ? 4 : 2) void attn_bwd_dkv_bf16_d64_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];
  __shared__ float s_stat[2][2][64];
  attn_bwd_dkv_body<NK>(p, xcd_linear_id(), smem, s_stat);
}
// Both halves of the backward in ONE launch: the first n_dq workgroups are dQ workgroups, the others dK / dV workgroups (they are
// independent; both only need delta).  Two launches on two streams cost a fork and a join -- 0.16 ms of idle time per step over the
// 12 attention blocks of the benchmark model -- for the same overlap.
template <int N>
__global__ __launch_bounds__(256, N == 1 ? 4 : 2) void attn_bwd_both_bf16_d64_kernel(AttnArgs p, int n_dq) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * TILE];
  __shared__ float s_stat[2][2][64];
  const int bid = (int)blockIdx.x;
  if (bid < n_dq) attn_bwd_dq_body<N>(p, xcd_linear_id(bid, n_dq), smem);
  else attn_bwd_dkv_body<N>(p, xcd_linear_id(bid - n_dq, (int)gridDim.x - n_dq), smem, s_stat);
}

__global__ __launch_bounds__(256) void attn_delta_bf16_d64_kernel(AttnArgs p) {
  // delta[b,h,q] = sum_d dO * O : one 8-lane group per row (8 x 16 B = one 128-byte head row)
  const int64_t idx = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int64_t total = (int64_t)p.B * p.H * p.Tq;
  const int sub = threadIdx.x & 7;
  float acc = 0.f;
  if (idx < total) {
    const int q = (int)(idx % p.Tq);
    const int h = (int)((idx / p.Tq) % p.H);
    const int b = (int)(idx / ((int64_t)p.Tq * p.H));
    const int64_t off = (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD + sub * 8;
    Chunk<bf16_t> o, d;
    d.v = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.dO) + off);
    if (p.O32) {       // un-rounded forward output: the rounding error of a bf16 O does not cancel against dP (see asr_hip.h)
      const f32x4_t o0 = *reinterpret_cast<const f32x4_t*>(p.O32 + off), o1 = *reinterpret_cast<const f32x4_t*>(p.O32 + off + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += o0[j] * bf16_to_f32(d.e[j]) + o1[j] * bf16_to_f32(d.e[4 + j]);
    } else {
      o.v = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.O) + off);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += bf16_to_f32(o.e[j]) * bf16_to_f32(d.e[j]);
    }
  }
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 1, 64);
  if (idx < total && sub == 0) p.delta[idx] = acc;
}

bool fast_ok(const AttnArgs& p, int d, int dtype) {
  if (p.q_st >= (1 << 22) || p.k_st >= (1 << 22) || p.v_st >= (1 << 22) || p.o_st >= (1 << 22)) return false;   // 32-bit in-tile byte offsets
  if (p.key_pad && (int64_t)(p.B - 1) * p.m_sb + (int64_t)(p.Tq - 1) * p.m_sq + p.Tk >= ((int64_t)1 << 31)) return false;   // 32-bit mask offsets
  return dtype == ASR_BF16 && d == HD && p.vec && p.Tq > 0 && p.Tk > 0 && asr_tuning("ATTN_GENERIC", 0) == 0;
}

}  // namespace

int attn_fast_fwd(const AttnArgs& p, int d, int dtype, hipStream_t s) {
  if (!fast_ok(p, d, dtype) || (((uintptr_t)p.Out) & 7) != 0 || p.o_st % 4 != 0 || p.o_sb % 4 != 0) return ASR_EUNSUPPORTED;
  if (!p.causal && p.Tk >= asr_tuning("ATTN_PP_MIN", 384) && asr_tuning("ATTN_PP", 1) != 0 && attn_pp_fwd(p, s) == ASR_OK) return ASR_OK;
  if (p.Tq <= asr_tuning("ATTN_SHORT", 256)) {
    const dim3 grid((unsigned)(((p.Tq + 63) / 64) * p.B * p.H));
    if (p.thr) attn_fwd_bf16_d64_kernel<1, true><<<grid, dim3(256), 0, s>>>(p);
    else attn_fwd_bf16_d64_kernel<1, false><<<grid, dim3(256), 0, s>>>(p);
  } else {
    const dim3 grid((unsigned)(((p.Tq + FQ - 1) / FQ) * p.B * p.H));
    if (p.thr) attn_fwd_bf16_d64_kernel<2, true><<<grid, dim3(256), 0, s>>>(p);
    else attn_fwd_bf16_d64_kernel<2, false><<<grid, dim3(256), 0, s>>>(p);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

int attn_fast_bwd(const AttnArgs& p, int d, int dtype, hipStream_t s) {
  if (!fast_ok(p, d, dtype)) return ASR_EUNSUPPORTED;
  for (const void* ptr : {(const void*)p.dQ, (const void*)p.dK, (const void*)p.dV, p.O})
    if ((((uintptr_t)ptr) & 15) != 0) return ASR_EUNSUPPORTED;
  const int64_t rows = (int64_t)p.B * p.H * p.Tq;
  if (p.parts & ASR_ATTN_DELTA) {
    attn_delta_bf16_d64_kernel<<<dim3((unsigned)ceil_div64(rows, 32)), dim3(256), 0, s>>>(p);
    ASR_LAUNCH_CHECK();
  }
  // short key sequences: the whole backward of a (batch, head) in one workgroup, one pass over the scores (attn_bwd_fused_body)
  if ((p.parts & ASR_ATTN_DQ) && (p.parts & ASR_ATTN_DKV) && p.Tk <= 256 && asr_tuning("ATTN_BWD_FUSED", 1) != 0) {
    static bool granted = false;
    if (!granted) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_bf16_d64_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_BWD_LDS);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_fused_bf16_d64_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, FUSED_BWD_LDS);
      granted = true;
    }
    // 8 waves x 16 keys up to 128 keys, x 32 keys up to 256
    if (p.Tk <= 128) attn_bwd_fused_bf16_d64_kernel<1, 8><<<dim3((unsigned)(p.B * p.H)), dim3(512), FUSED_BWD_LDS, s>>>(p);
    else attn_bwd_fused_bf16_d64_kernel<2, 8><<<dim3((unsigned)(p.B * p.H)), dim3(512), FUSED_BWD_LDS, s>>>(p);
    ASR_LAUNCH_CHECK();
    return ASR_OK;
  }
  if ((p.parts & ASR_ATTN_DQ) && (p.parts & ASR_ATTN_DKV) && asr_tuning("ATTN_BOTH", 1) != 0) {
    const bool short_q = p.Tq <= asr_tuning("ATTN_SHORT_BWD", 1 << 30), short_k = p.Tk <= asr_tuning("ATTN_SHORT_BWD", 1 << 30);
    if (short_q == short_k) {
      const int rows = short_q ? 64 : FQ;
      const int n_dq = ((p.Tq + rows - 1) / rows) * p.B * p.H, n_dkv = ((p.Tk + rows - 1) / rows) * p.B * p.H;
      if (short_q) attn_bwd_both_bf16_d64_kernel<1><<<dim3((unsigned)(n_dq + n_dkv)), dim3(256), 0, s>>>(p, n_dq);
      else attn_bwd_both_bf16_d64_kernel<2><<<dim3((unsigned)(n_dq + n_dkv)), dim3(256), 0, s>>>(p, n_dq);
      ASR_LAUNCH_CHECK();
      return ASR_OK;
    }
  }
  if (p.parts & ASR_ATTN_DQ) {
    if (p.Tq <= asr_tuning("ATTN_SHORT_BWD", 1 << 30))
      attn_bwd_dq_bf16_d64_kernel<1><<<dim3((unsigned)(((p.Tq + 63) / 64) * p.B * p.H)), dim3(256), 0, s>>>(p);
    else
      attn_bwd_dq_bf16_d64_kernel<2><<<dim3((unsigned)(((p.Tq + FQ - 1) / FQ) * p.B * p.H)), dim3(256), 0, s>>>(p);
    ASR_LAUNCH_CHECK();
  }
  if (p.parts & ASR_ATTN_DKV) {
    if (p.Tk <= asr_tuning("ATTN_SHORT_BWD", 1 << 30))
      attn_bwd_dkv_bf16_d64_kernel<1><<<dim3((unsigned)(((p.Tk + 63) / 64) * p.B * p.H)), dim3(256), 0, s>>>(p);
    else
      attn_bwd_dkv_bf16_d64_kernel<2><<<dim3((unsigned)(((p.Tk + FQ - 1) / FQ) * p.B * p.H)), dim3(256), 0, s>>>(p);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}

}  // namespace asr_attn
