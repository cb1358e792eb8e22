// Long-sequence attention forward for bf16 / head dim 64 (models/common_layers.py:211-225 at dim_key = dim_value = 64; the encoder
// self-attention of BASELINE.json's north-star shape B H = 256, T = 800 and of configs[3], T' = 795).  Round 3 redesign of
// attention_fast.hip's forward kernel; what changed and why (MI355X_MICROARCH.md "Two waves per SIMD", measured split of the old
// kernel in profiles/r02_attention_d64_pmc.txt: 8.6 vector instructions per 16x16x32 MFMA, 26 % of a wave's life parked at
// s_waitcnt / s_barrier, 2.2 waves per SIMD):
//   * v_mfma_f32_32x32x16_bf16 with swapped contractions S^T = K Q^T, O^T = V^T P^T: a lane owns ONE query column, its 32 scores of
//     a 64-key tile sit in its own registers -> row maximum and row sum are lane-local (one v_permlane32_swap per tile for the
//     maximum, none for the sum until the epilogue), and the register -> key map of an S^T block IS the k order of the second
//     contraction, so P goes from the softmax straight into the MFMA operand (no LDS, no lane permutes);
//   * the S^T accumulators start at -reference instead of 0, so a probability is exp2(c2 * s): half a packed multiply and one
//     v_exp_f32 per score, no subtraction;
//   * the running maximum is only moved when a tile's maximum exceeds it by more than 2^THR (wave-uniform branch): the O(d) rescale
//     of the accumulators leaves the tile loop; everything still at the old maximum (O and l) is rescaled at that one point, before
//     the tile's probabilities are formed, and the previous tile's P V is complete by then;
//   * phases: every operand of the NEXT matrix phase (K(t+1) fragments and V(t) fragments) is read from LDS into registers at the top
//     of the softmax phase, so the matrix phase is 16 back-to-back MFMAs on registers with no wait in it, and the softmax phase is
//     pure vector work; two 4-wave workgroups per CU (one wave of each per SIMD, 256 registers each) run these phases against each
//     other -- matrix pipe and vector pipe of a SIMD are fed by different waves;
//   * K / V tiles arrive by hand-issued LDS-DMA into a 3-deep ring, ONE barrier per tile; a tile's DMA is issued a whole iteration
//     before it is awaited, so the s_waitcnt vmcnt(0) in front of the barrier finds it landed;
//   * work items: 128-query chunks; the queries left over after the last full chunk (800 = 6 x 128 + 32) run in TAIL workgroups
//     whose four waves split the KEY range of one 32-query block and combine through LDS -- the launch is 3 full rounds of the
//     512 workgroup slots plus short tails instead of 3.5 -> 4 rounds.
#include "attention.h"

namespace asr_attn {
namespace {

constexpr int HD = 64;
constexpr int ROWB = 128;                  // bytes per LDS row (64 bf16)
constexpr int KT = 64;                     // keys per tile
constexpr int TILE = KT * ROWB;            // 8 KB
constexpr int NS = 3;                      // ring depth
constexpr float LOG2E = 1.4426950408889634f;
constexpr float THR = 8.f;                 // deferred maximum: P <= 2^THR
constexpr float M_INIT = -1e30f;

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float max_halves(float v) {
  const uint32_t u = __float_as_uint(v);
  auto c = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(c[0]), __uint_as_float(c[1]));
}
__device__ __forceinline__ float sum_halves(float v) {
  const uint32_t u = __float_as_uint(v);
  auto c = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(c[0]) + __uint_as_float(c[1]);
}
__device__ __forceinline__ int xcd_linear(int bid, int nwg) {
  const int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
}
__device__ __forceinline__ void mma32(f32x16_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// LDS-DMA, hand issued (the compiler must not count it: see attention_fast.hip lds_dma16)
__device__ __forceinline__ void dma16(unsigned lds_wave_base, const void* src) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}
__device__ __forceinline__ void dma16_s(unsigned lds_wave_base, unsigned voff, const void* sbase) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(voff), "s"(sbase)
               : "memory");
}

// LDS image of a 64-key tile: row = key, 8 chunks of 16 B, chunk c of row r stored at slot c ^ key(r).
//   K: key(r) = (r >> 1) & 7  -- ds_read_b128 by 32 lanes = 32 consecutive rows at one logical chunk: conflict free with two rows
//      per 256-byte bank line (lane groups of MI355X_MICROARCH.md "LDS");
//   V: key(r) = ((r >> 1) & 1) << 2 -- ds_read_b64_tr_b16 by 32 lanes = 4 consecutive rows x 64 bytes: rows r and r + 2 share a bank
//      line half, the key moves them to the other 64-byte group.
__device__ __forceinline__ int swz_k(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int swz_v(int r) { return ((r >> 1) & 1) << 2; }

template <int NW>
struct Stager {          // this thread's share of a tile: 512 / (64 NW) 16-byte chunks of K and as many of V (8 KB each)
  static constexpr int NC = 8 / NW;          // chunks per thread and operand: 2 (4 waves) or 1 (8 waves)
  unsigned voK[NC], voV[NC];    // byte offsets inside a FULL tile (row * stride + swizzled chunk), hoisted out of the tile loop
  unsigned piece;               // wave-uniform: LDS byte offset of this wave's first 1-KB piece inside a tile
  const bf16_t *Kb, *Vb;
  int k_st, v_st, Tk, tid;
  __device__ __forceinline__ void init(const AttnArgs& p, const bf16_t* Kb_, const bf16_t* Vb_, int tid_, int wave) {
    Kb = Kb_; Vb = Vb_; k_st = (int)p.k_st; v_st = (int)p.v_st; Tk = p.Tk; tid = tid_;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      const int c = i * 64 * NW + tid, row = c >> 3, slot = c & 7;
      voK[i] = (unsigned)(row * k_st * 2 + ((slot ^ swz_k(row)) << 4));
      voV[i] = (unsigned)(row * v_st * 2 + ((slot ^ swz_v(row)) << 4));
    }
    piece = (unsigned)(wave * 64 * 16);
  }
  // tile starting at key k0 -> stage at LDS byte address `stage` ([K | V]); rows past Tk re-read the last valid row (the LDS-DMA cannot
  // zero-fill; such keys are masked by the caller)
  __device__ __forceinline__ void issue(unsigned stage, int k0) const {
    const bf16_t* kb = Kb + (int64_t)k0 * k_st;
    const bf16_t* vb = Vb + (int64_t)k0 * v_st;
    if (k0 + KT <= Tk) {
#pragma unroll
      for (int i = 0; i < NC; ++i) dma16_s(stage + piece + i * NW * 1024, voK[i], kb);
#pragma unroll
      for (int i = 0; i < NC; ++i) dma16_s(stage + TILE + piece + i * NW * 1024, voV[i], vb);
    } else {
      const int last = Tk - 1 - k0;
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int c = i * 64 * NW + tid, row = c >> 3, slot = c & 7, gr = row < last ? row : last;
        dma16_s(stage + piece + i * NW * 1024, (unsigned)(gr * k_st * 2 + ((slot ^ swz_k(row)) << 4)), kb);
        dma16_s(stage + TILE + piece + i * NW * 1024, (unsigned)(gr * v_st * 2 + ((slot ^ swz_v(row)) << 4)), vb);
      }
    }
  }
};

// register file of one wave-tile: 32 queries (lane & 31 = query, lane >> 5 = half)
struct Frags {
  uint4 kf[2][4];        // K(t): [key block][d step]      A operand of S^T
  uint4 vf[2][2][2];     // V(t): [d block][key block][k step]  A operand of O^T (transposing reads)
};

// Fragment reads.  `stage` is a compile-time LDS byte offset in the unrolled loop; the per-lane part is made opaque per call so
// that the (loop-invariant) sums lane offset + constant are NOT hoisted into one address register each -- they fold into the
// instructions' immediate offsets instead (hoisted, they cost 24 registers per ring stage).
typedef __attribute__((address_space(3))) const u32x4_t* lds_u4_ptr;
__device__ __forceinline__ void read_k(uint4 (&kf)[2][4], unsigned stage, const unsigned (&koff)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    unsigned a = koff[ks];
    asm volatile("" : "+v"(a));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const u32x4_t v = *(lds_u4_ptr)(uintptr_t)(a + stage + kb * 32 * ROWB);
      kf[kb][ks] = make_uint4(v[0], v[1], v[2], v[3]);
    }
  }
}
__device__ __forceinline__ uint2 lds_tr16(unsigned a) {
  const asr_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) asr_s16x4_t*)(uintptr_t)a);
  return __builtin_bit_cast(uint2, v);
}
__device__ __forceinline__ void read_v(uint4 (&vf)[2][2][2], unsigned stage, const unsigned (&voff)[2]) {
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    unsigned a = voff[db];
    asm volatile("" : "+v"(a));
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint2 lo = lds_tr16(a + stage + (kb * 32 + j * 16) * ROWB);
        const uint2 hi = lds_tr16(a + stage + (kb * 32 + j * 16 + 8) * ROWB);
        vf[db][kb][j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
  }
}

// key index of score register r of key block kb in the lane's half: the C layout of v_mfma_f32_32x32x16 (row = (r & 3) + 8 (r >> 2) + 4 half)
__device__ __forceinline__ int key_of(int kb, int r, int half) { return kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; }

// Per-lane softmax state of one query.  Scores leave the first contraction already RELATIVE to the reference `ref` (the accumulators
// of S^T start at nref = -ref in every register instead of 0): a probability is exp2(c2 * s), one packed multiply per two scores and
// one v_exp_f32 per score.  `ref` (raw score units) follows the running maximum lazily: it moves when a tile's maximum exceeds it by
// more than THR / c2, or at the first tile that has a live key (`gate` = -FLT_MAX until then, THR / c2 afterwards: one compare
// covers both).  nref is 0 until then, never -(-inf).
// (Round 3 first folded c2 = scale * log2(e) into a bf16 re-rounding of Q: one instruction per score less, but the second rounding
//  of Q costs 0.25 % of a score -- 0.1 in the exponent at |score| = 40, measured by tools/ab/ab_attn_pp.py "spiked keys" -- where
//  the first-generation kernel and the reference are exact in fp32.  Not kept.)
struct Soft {
  float ref, gate;
  f32x2_t la, lb;     // four partial row sums (two independent packed chains)
  f32x16_t nref;
};

// key-length mask of the tile that holds the boundary (key padding BYTES are not this kernel's: attn_pp_fwd refuses them)
__device__ __forceinline__ void mask_tile(f32x16_t (&s)[2], int k0, int half, int kend) {
  const int lim = kend - k0 - 4 * half;          // key_of(kb, r, 0) >= lim  <=>  key >= kend
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[kb][r] = key_of(kb, r, 0) >= lim ? -INFINITY : s[kb][r];
}

// softmax phase: relative scores (log2 units) -> P as the packed B operand pk[key block][k step].  Everything still at the old
// reference (O, l, and this tile's scores) is moved at ONE point, before the tile's probabilities are formed; the previous tile's
// P V is complete by then (it was issued before this tile's S in program order).
// part 1: tile maximum, and (rare, wave-uniform) the move of the reference with everything that still sits at the old one
__device__ __forceinline__ void softmax_decide(f32x16_t (&s)[2], Soft& st, f32x16_t (&o)[2], float c2) {
  float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s[0][r]), s[1][r]);
  mx = max_halves(mx);
  const bool move = mx > st.gate;
  if (__builtin_amdgcn_ballot_w64(move) != 0ull) {
    // lanes that do not move: delta = 0, alpha = 1.  A first move (gate = -FLT_MAX) has O = l = 0: alpha is kept finite.
    const float delta = (move && mx > -INFINITY) ? mx : 0.f;
    const float alpha = __builtin_amdgcn_exp2f(fminf(-delta * c2, 100.f));
    st.ref += delta;
    st.gate = (move && mx > -INFINITY) ? THR / c2 : st.gate;
    st.la *= alpha;
    st.lb *= alpha;
    o[0] *= alpha;
    o[1] *= alpha;
    s[0] -= delta;
    s[1] -= delta;
    st.nref -= delta;
  }
}
// part 2: probabilities, row sums, dropout, packed B operand
template <bool DROP>
__device__ __forceinline__ void softmax_form(f32x16_t (&s)[2], Soft& st, uint4 (&pk)[2][2], uint32_t rkey, int k0, int half, uint32_t thr,
                                             float c2) {
  const uint32_t thrm1 = (thr - 1u) * 0x10001u;        // (thr - 1) in both halves (DROP: thr >= 1)
  // rkey is the lane's hash base ALREADY multiplied out: (row key + 2 half) * DROP_C1 (attn_fwd_pp_body); the pair index of register r
  // of key block kb is k0 / 2 + 2 half + kb 16 + (r & 3) / 2 + 4 (r >> 2), so a pair's first-stage value is tile base + a literal
  const uint32_t ytile = rkey + (uint32_t)(k0 >> 1) * DROP_C1;
  s[0] *= c2;                                          // v_pk_mul_f32: log2 units
  s[1] *= c2;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const float a0 = __builtin_amdgcn_exp2f(s[0][r]), a1 = __builtin_amdgcn_exp2f(s[0][r + 1]);
    const float b0 = __builtin_amdgcn_exp2f(s[1][r]), b1 = __builtin_amdgcn_exp2f(s[1][r + 1]);
    st.la += f32x2_t{a0, a1};                          // v_pk_add_f32, two independent chains
    st.lb += f32x2_t{b0, b1};
    s[0][r] = a0; s[0][r + 1] = a1;
    s[1][r] = b0; s[1][r + 1] = b1;
  }
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 8 * j + 2 * e;
        w[e] = pack_bf16(s[kb][r], s[kb][r + 1]);
        if (DROP) {
          // the pair's two 16-bit uniform fields against the threshold, both at once on the packed-16 ALU (drop_apply_pk, attention.h)
          const uint32_t y = drop_pair_mix(ytile + (uint32_t)(key_of(kb, r, 0) >> 1) * DROP_C1);
          w[e] = drop_apply_pk(w[e], y, thrm1, 0x10001u);
        }
      }
      pk[kb][j] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

template <bool DROP>
__device__ __forceinline__ void softmax_tile(f32x16_t (&s)[2], Soft& st, f32x16_t (&o)[2], uint4 (&pk)[2][2], uint32_t rkey, int k0,
                                             int half, uint32_t thr, float c2) {
  softmax_decide(s, st, o, c2);
  softmax_form<DROP>(s, st, pk, rkey, k0, half, thr, c2);
}

// matrix work: O^T += V(t)^T P(t)^T (two accumulator chains) and S^T(t+1) = K(t+1) Q^T - ref (two chains)
template <int KB>
__device__ __forceinline__ void pv_phase(f32x16_t (&o)[2], const Frags& f, const uint4 (&pk)[2][2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    mma32(o[0], f.vf[0][KB][j], pk[KB][j]);
    mma32(o[1], f.vf[1][KB][j], pk[KB][j]);
  }
}
__device__ __forceinline__ void qk_phase(f32x16_t (&s)[2], const Frags& f, const uint4 (&qf)[4], const f32x16_t& nref) {
  s[0] = nref;
  s[1] = nref;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    mma32(s[0], f.kf[0][i], qf[i]);
    mma32(s[1], f.kf[1][i], qf[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------------- the kernel
// grid: n_full * B * H chunk workgroups (128 queries each: wave w owns queries 32 w ..), then n_tail * B * H tail workgroups (one
// 32-query block each, the four waves take interleaved key tiles and combine).  blockIdx -> work item is XCD-aware: the workgroups
// of one (b, h) share an L2.  The two modes are separate bodies (no merged register state between them).
struct Ctx {             // loop-invariant per-lane / per-wave values
  unsigned koff[4], voff[2];
  unsigned lds0;
  int half, kend, ntile, prio;
  uint32_t rkey;
  float c2;            // scale * log2(e)
};

// one iteration of the chunk loop on ring stage STG (compile time: every LDS offset of the iteration is an instruction immediate)
template <bool DROP, int STG, int NW>
__device__ __forceinline__ void chunk_iter(const AttnArgs& p, const Ctx& c, const Stager<NW>& sg, unsigned char* smem, int t, f32x16_t (&s)[2],
                                           f32x16_t (&o)[2], Soft& st, Frags& f, const uint4 (&qf)[4], uint4 (&pk)[2][2]) {
  constexpr unsigned stg_t = STG * 2 * TILE, stg_n = ((STG + 1) % NS) * 2 * TILE;
  const bool more = t + 1 < c.ntile;
  read_v(f.vf, stg_t + TILE, c.voff);
  __builtin_amdgcn_sched_barrier(0);
  const int k0 = t * KT;
  if (k0 + KT > c.kend) mask_tile(s, k0, c.half, c.kend);
  softmax_tile<DROP>(s, st, o, pk, c.rkey, k0, c.half, p.thr, c.c2);
  __builtin_amdgcn_sched_barrier(0);
  if (c.prio == 1) __builtin_amdgcn_s_setprio(1);
  // K(t+1) fragments are read behind the first half of P V (the registers of V's first key block are free by then) and land under
  // its second half
  pv_phase<0>(o, f, pk);
  __builtin_amdgcn_sched_barrier(0);
  if (more) read_k(f.kf, stg_n, c.koff);
  pv_phase<1>(o, f, pk);
  if (more) qk_phase(s, f, qf, st.nref);
  if (c.prio == 1) __builtin_amdgcn_s_setprio(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (t + 3 < c.ntile) sg.issue(c.lds0 + (unsigned)(STG * 2 * TILE), (t + 3) * KT);
}

template <bool DROP, bool TAIL, int NW>
__device__ __forceinline__ void attn_fwd_pp_body(const AttnArgs& p, unsigned char* smem, int bh, int q0, int prio) {
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, half = lane >> 5, l16 = lane & 15, dh = (lane >> 4) & 1;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = bh / p.H, h = bh - b * p.H;
  const int q = q0 + n;
  const bf16_t* Qb = static_cast<const bf16_t*>(p.Q) + (int64_t)b * p.q_sb + (int64_t)h * HD;
  const bf16_t* Kb = static_cast<const bf16_t*>(p.K) + (int64_t)b * p.k_sb + (int64_t)h * HD;
  const bf16_t* Vb = static_cast<const bf16_t*>(p.V) + (int64_t)b * p.v_sb + (int64_t)h * HD;
  Ctx c;
  c.rkey = 0u;
  // (multiplied out, with the lane half's two key pairs folded in: softmax_form adds the tile's and the register's part)
  if (DROP) c.rkey = (drop_row_key(asr_mix_seed(p.seed, p.seed_dev), drop_row(p, b, h, q)) + 2u * (uint32_t)half) * DROP_C1;
  c.lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  c.half = half; c.prio = prio; c.c2 = p.scale * LOG2E;
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) c.koff[ks] = c.lds0 + (unsigned)(n * ROWB + (((2 * ks + half) ^ swz_k(n)) << 4));
#pragma unroll
  for (int db = 0; db < 2; ++db)
    c.voff[db] = c.lds0 + (unsigned)((4 * half + (l16 >> 2)) * ROWB + (((4 * (db ^ ((l16 >> 3) & 1)) + 2 * dh + ((l16 & 3) >> 1))) << 4) + (l16 & 1) * 8);
  c.kend = key_end(p, b);
  c.ntile = (c.kend + KT - 1) / KT;
  const int ntile = c.ntile;
  Stager<NW> sg;
  sg.init(p, Kb, Vb, tid, wave);

  // Q^T fragments (B operand)
  uint4 qf[4];
  {
    const int qr = q < p.Tq ? q : p.Tq - 1;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const uint4*>(Qb + (int64_t)qr * p.q_st + ks * 16 + half * 8);
  }
  const f32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16_t o[2] = {zero16, zero16}, s[2] = {zero16, zero16};
  Soft st;
  st.ref = 0.f; st.la = f32x2_t{0.f, 0.f}; st.lb = st.la; st.gate = -3.0e38f; st.nref = zero16;
  Frags f;
  uint4 pk[2][2];

  // ---- prologue: the ring is filled (Q was loaded first: the compiler's own waits for it do not have to cover the DMA)
  if (ntile > 0) sg.issue(c.lds0, 0);
  if (ntile > 1) sg.issue(c.lds0 + 2 * TILE, KT);
  if (ntile > 2) sg.issue(c.lds0 + 2 * 2 * TILE, 2 * KT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  if (!TAIL) {
    // every wave computes every tile.  Iteration t: V(t) fragments -> softmax of S(t) -> K(t+1) fragments -> O += P(t) V(t) -> S(t+1); the
    // barrier at its end frees tile t's stage (refilled with tile t + 3) and publishes tile t + 2.
    if (ntile > 0) {
      read_k(f.kf, 0u, c.koff);
      qk_phase(s, f, qf, st.nref);
    }
    for (int t = 0; t < ntile; t += 3) {
      chunk_iter<DROP, 0, NW>(p, c, sg, smem, t, s, o, st, f, qf, pk);
      if (t + 1 >= ntile) break;
      chunk_iter<DROP, 1, NW>(p, c, sg, smem, t + 1, s, o, st, f, qf, pk);
      if (t + 2 >= ntile) break;
      chunk_iter<DROP, 2, NW>(p, c, sg, smem, t + 2, s, o, st, f, qf, pk);
    }
  } else {
    // wave w < 4 computes tiles w, w + 4, ...; the workgroup still stages EVERY tile in order (the ring and the barriers are common;
    // the waves 4 .. 7 of an 8-wave workgroup only stage)
    for (int t = 0; t < ntile; ++t) {
      if ((t & 3) == wave) {
        const unsigned stg_t = (unsigned)((t % NS) * 2 * TILE);
        read_k(f.kf, stg_t, c.koff);
        read_v(f.vf, stg_t + TILE, c.voff);
        qk_phase(s, f, qf, st.nref);
        const int k0 = t * KT;
        if (k0 + KT > c.kend) mask_tile(s, k0, half, c.kend);
        softmax_tile<DROP>(s, st, o, pk, c.rkey, k0, half, p.thr, c.c2);
        pv_phase<0>(o, f, pk);
        pv_phase<1>(o, f, pk);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t + 3 < ntile) sg.issue(c.lds0 + (unsigned)((t % NS) * 2 * TILE), (t + 3) * KT);
    }
  }
  float l = (st.la[0] + st.la[1]) + (st.lb[0] + st.lb[1]);
  float ref = st.gate > 0.f ? st.ref : M_INIT;       // no live key seen: "minus infinity"
  if (TAIL) {
    // combine the four key ranges of the block (wave 0 collects); the loop's last barrier has retired every ring access
    float* red = reinterpret_cast<float*>(smem);       // [wave][34][64]: ref, l, o[0][16], o[1][16] per lane
    if (wave < 4) red[(wave * 34 + 0) * 64 + lane] = ref;
    __syncthreads();
    const float mg = fmaxf(fmaxf(red[(0 * 34) * 64 + lane], red[(1 * 34) * 64 + lane]), fmaxf(red[(2 * 34) * 64 + lane], red[(3 * 34) * 64 + lane]));
    const float a = __builtin_amdgcn_exp2f((ref - mg) * c.c2);        // all at M_INIT (no key seen by anyone): exp2(0) = 1 on zeros
    if (wave < 4) {
      red[(wave * 34 + 1) * 64 + lane] = l * a;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[(wave * 34 + 2 + r) * 64 + lane] = o[0][r] * a;
        red[(wave * 34 + 18 + r) * 64 + lane] = o[1][r] * a;
      }
    }
    __syncthreads();
    if (wave != 0) return;
    ref = mg;
    l = red[(0 * 34 + 1) * 64 + lane] + red[(1 * 34 + 1) * 64 + lane] + red[(2 * 34 + 1) * 64 + lane] + red[(3 * 34 + 1) * 64 + lane];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o[0][r] = red[(0 * 34 + 2 + r) * 64 + lane] + red[(1 * 34 + 2 + r) * 64 + lane] + red[(2 * 34 + 2 + r) * 64 + lane] + red[(3 * 34 + 2 + r) * 64 + lane];
      o[1][r] = red[(0 * 34 + 18 + r) * 64 + lane] + red[(1 * 34 + 18 + r) * 64 + lane] + red[(2 * 34 + 18 + r) * 64 + lane] + red[(3 * 34 + 18 + r) * 64 + lane];
    }
  }
  const float l_tot = sum_halves(l);

  // ---- epilogue: O = O^T / l, stored as 16-byte row pieces (half 0 and half 1 of a query exchange 8-byte pieces)
  if (q0 >= p.Tq) return;
  const float inv_l = l_tot > 0.f ? p.inv_keep / l_tot : 0.f;
  if (half == 0 && q < p.Tq) p.lse[((int64_t)b * p.H + h) * p.Tq + q] = l_tot > 0.f ? ref * p.scale + __logf(l_tot) : INFINITY;
  const int64_t orow = (int64_t)b * p.o_sb + (int64_t)q * p.o_st + (int64_t)h * HD;
  bf16_t* Ob = static_cast<bf16_t*>(p.Out) + orow;
#pragma unroll
  for (int db = 0; db < 2; ++db) {
    const f32x16_t v = o[db] * inv_l;
    if (p.Out32 && q < p.Tq) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
        *reinterpret_cast<f32x4_t*>(p.Out32 + orow + db * 32 + rg * 8 + half * 4) = f32x4_t{v[4 * rg], v[4 * rg + 1], v[4 * rg + 2], v[4 * rg + 3]};
    }
#pragma unroll
    for (int pg = 0; pg < 2; ++pg) {
      // pieces rg = 2 pg (d = 8 rg + 4 half ..) and rg + 1: after the swap half 0 holds d = 8 rg .. 8 rg + 7, half 1 d = 8 (rg + 1) ..
      uint2 a = make_uint2(pack_bf16(v[8 * pg + 0], v[8 * pg + 1]), pack_bf16(v[8 * pg + 2], v[8 * pg + 3]));
      uint2 cc = make_uint2(pack_bf16(v[8 * pg + 4], v[8 * pg + 5]), pack_bf16(v[8 * pg + 6], v[8 * pg + 7]));
      auto r0 = __builtin_amdgcn_permlane32_swap(a.x, cc.x, false, false);
      auto r1 = __builtin_amdgcn_permlane32_swap(a.y, cc.y, false, false);
      if (q < p.Tq) *reinterpret_cast<uint4*>(Ob + db * 32 + (2 * pg + half) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
    }
  }
}

template <bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd_pp_bf16_d64_kernel(AttnArgs p, int n_full, int n_tail, int prio) {
  __shared__ __attribute__((aligned(256))) unsigned char smem[NS * 2 * TILE];
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int BH = p.B * p.H;
  const int n_main = n_full * BH;
  const int bid = (int)blockIdx.x;
  // Stagger (prio bits 8..): the two workgroups that share a CU start together and run phases of equal length, i.e. in lockstep --
  // both in their softmax phase, then both in their matrix phase -- which is the one arrangement in which the SIMD's vector and
  // matrix pipes never overlap.  Every other workgroup (by the dispatch order's CU round: bit 8 of the block id, or bit 3) starts
  // `stagger` x 64 cycles late; equal run times keep the offset for the rest of the launch.
  {
    const int stagger = (prio >> 8) & 0xff, sel = (prio >> 16) & 1;
    if (stagger && ((bid >> (sel ? 3 : 8)) & 1))
      for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);
    prio &= 0xff;
  }
  if (bid < n_main) {
    const int vid = xcd_linear(bid, n_main);
    const int bh = vid / n_full;
    attn_fwd_pp_body<DROP, false, NW>(p, smem, bh, (vid - bh * n_full) * (32 * NW) + wave * 32, prio);
  } else {
    const int vid = xcd_linear(bid - n_main, n_tail * BH);
    const int bh = vid / n_tail;
    attn_fwd_pp_body<DROP, true, NW>(p, smem, bh, n_full * (32 * NW) + (vid - bh * n_tail) * 32, prio);
  }
}

}  // namespace

// Entry: ASR_EUNSUPPORTED unless the long-sequence kernel applies (bf16, d = 64, no causal mask, no padding bytes -- key lengths only --, 16-byte aligned rows -- checked by
// the caller's fast_ok() -- and at least `min_keys` keys).
int attn_pp_fwd(const AttnArgs& p, hipStream_t s) {
  if (p.causal || p.key_pad) return ASR_EUNSUPPORTED;
  if ((((uintptr_t)p.Out) & 15) != 0 || p.o_st % 8 != 0 || p.o_sb % 8 != 0) return ASR_EUNSUPPORTED;
  if (p.Out32 && (((uintptr_t)p.Out32) & 15) != 0) return ASR_EUNSUPPORTED;
  const int BH = p.B * p.H;
  constexpr int nw = 4;
  const int n_full = p.Tq / (32 * nw);
  const int rest = p.Tq - n_full * 32 * nw;
  int n_tail = (rest + 31) / 32;
  int nf = n_full;
  if (asr_tuning("ATTN_PP_TAIL", 1) == 0 && rest > 0) { nf = n_full + 1; n_tail = 0; }      // A/B: leftover queries as one more (partly idle) chunk
  const dim3 grid((unsigned)((nf + n_tail) * BH));
  // measured (profiles/r03_attention_pp_stagger_pipe_ab.txt): any delay of 4 .. 12 x 64 cycles on block-id bit 3 takes the north-star
  // shape from 71.8 to 65.8 us at p = 0 and changes nothing elsewhere
  const int prio = (int)asr_tuning("ATTN_PP_PRIO", 0) | ((int)asr_tuning("ATTN_PP_STAGGER", 8) << 8) |
                   ((int)asr_tuning("ATTN_PP_STAGGER_SEL", 1) << 16);
  if (p.thr) attn_fwd_pp_bf16_d64_kernel<true, nw><<<grid, dim3(64 * nw), 0, s>>>(p, nf, n_tail, prio);
  else attn_fwd_pp_bf16_d64_kernel<false, nw><<<grid, dim3(64 * nw), 0, s>>>(p, nf, n_tail, prio);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

}  // namespace asr_attn
