// Shared pieces of the persistent 64 -> 64 channel kernels (conv_c64.hip: conv.2 forward / data gradient on stored activations;
// conv_level0.hip: the same contractions with the halo patch PRODUCED in LDS -- conv.0 recomputed from the log-mel frames, or the
// pooled gradient expanded through its selection codes).  See conv_c64.hip for the design notes.
#pragma once
#include "common.h"

#include <utility>

namespace {

__device__ const uint4 c64_zero_page = {0u, 0u, 0u, 0u};

typedef __attribute__((ext_vector_type(2))) float c64_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 c64_bf16x2_t;
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {      // one v_cvt_pk_bf16_f32
  const c64_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, c64_bf16x2_t));
}

__device__ __forceinline__ void lds_read16(u32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
}

// k step S = (tap, 32-channel half): the 4 pixel-fragment operands of the step.  Address = the lane's register for this
// (dx, half) -- buffer base + wave / lane part + swizzled chunk -- plus a compile-time (fragment, tap) offset (instruction immediate).
template <int S, int PW, int CB>
__device__ __forceinline__ void c64_issue(u32x4_t (&dst)[4], const unsigned (&pbd)[3][2]) {
  constexpr int tap = S >> 1, ms = S & 1, dy = tap / 3, dx = tap % 3;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    asm volatile("ds_read_b128 %0, %1 offset:%2"
                 : "=v"(dst[i])
                 : "v"(pbd[dx][ms]), "n"((((i / CB) + dy) * PW + (i % CB) * 16 + dx) * 128));
}

// FLIP = false: A = weights (rows = output channels), B = pixels: a lane owns 4 channels of ONE pixel (the NHWC store epilogue).
// FLIP = true : A = pixels, B = weights: a lane owns 4 consecutive PIXELS of one channel -- the accumulator fragment is then itself the
// B operand (k = pixel) of a 16x16x16 contraction over the pixels (conv_level0.hip: the first layer's weight gradient from the tile).
template <int S, int PW, int CB, bool FLIP = false>
__device__ __forceinline__ void c64_step(f32x4_t (&acc)[4][2], u32x4_t (&a)[2][4], const u32x4_t (&wB)[9][2][2],
                                         const unsigned (&pbd)[3][2], u32x4_t (&bq)[2]) {
  u32x4_t(&cur)[4] = a[S & 1];
  if constexpr (S == 0) {
    c64_issue<S + 1, PW, CB>(a[(S + 1) & 1], pbd);
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(bq[0]), "+v"(bq[1]));
  } else if constexpr (S + 1 < 18) {
    c64_issue<S + 1, PW, CB>(a[(S + 1) & 1], pbd);           // next step's operands in flight under this step's MFMAs
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]));
  }
  constexpr int tap = S >> 1, ms = S & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)      // the first step starts every accumulator from the bias of its 4 output channels
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, FLIP ? cur[i] : wB[tap][ms][j]),
                                                          __builtin_bit_cast(bf16x8_t, FLIP ? wB[tap][ms][j] : cur[i]),
                                                          S == 0 ? __builtin_bit_cast(f32x4_t, bq[j]) : acc[i][j], 0, 0, 0);
}

template <int PW, int CB, bool FLIP = false, int... S>
__device__ __forceinline__ void c64_steps(std::integer_sequence<int, S...>, f32x4_t (&acc)[4][2], u32x4_t (&a)[2][4],
                                          const u32x4_t (&wB)[9][2][2], const unsigned (&pbd)[3][2], u32x4_t (&bq)[2]) {
  (c64_step<S, PW, CB, FLIP>(acc, a, wB, pbd, bq), ...);
}

// ---- the same contraction ordered by PATCH ROW (round 6, conv_level0.hip).  For a fixed (tap column dx, channel half) the operand of patch
// row r serves every (tap row dy, tile row i) with i + dy = r: 6 row operands instead of 12 step operands, 36 ds_read_b128 per wave and tile
// instead of 72.  That matters because the step form runs the LDS exactly as long as the matrix pipe: 4 reads of 1 KB feed 8 MFMAs of 16
// cycles per wave, eight waves share the CU's 128 B / clk = 256 LDS cycles against 256 MFMA cycles per SIMD -- every bank conflict and every
// other LDS instruction of the kernel (patch generation, expansion) then stalls the MFMA stream.  Item K = (group G = K / 6: dx = G / 2,
// half = G % 2; row r = K % 6); reads run D items ahead in a ring of D + 1 registers quads (the step form held 8 quads).
template <int K, int PW>
__device__ __forceinline__ void c64_row_issue(u32x4_t& dst, const unsigned (&pbd)[3][2]) {
  constexpr int G = K / 6, r = K % 6, dx = G >> 1, ms = G & 1;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(pbd[dx][ms]), "n"((r * PW + dx) * 128));
}
template <int K, int PW, bool FLIP, int D>
__device__ __forceinline__ void c64_row_item(f32x4_t (&acc)[4][2], u32x4_t (&ring)[D + 1], const u32x4_t (&wB)[9][2][2],
                                             const unsigned (&pbd)[3][2], u32x4_t (&bq)[2]) {
  constexpr int G = K / 6, r = K % 6, dx = G >> 1, ms = G & 1;
  if constexpr (K + D < 36) c64_row_issue<K + D, PW>(ring[(K + D) % (D + 1)], pbd);
  constexpr int pending = (K + D < 36) ? D : (35 - K);      // reads issued after read K (the LDS returns in order)
  u32x4_t& cur = ring[K % (D + 1)];
  if constexpr (K == 0) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(cur), "+v"(bq[0]), "+v"(bq[1]) : "n"(pending));
  else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"(pending));
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int i = r - dy;
    if (i < 0 || i > 3) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)      // the first touch of an accumulator (group 0, tap row 0) starts it from the bias of its output channels
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, FLIP ? cur : wB[dy * 3 + dx][ms][j]),
                                                          __builtin_bit_cast(bf16x8_t, FLIP ? wB[dy * 3 + dx][ms][j] : cur),
                                                          (G == 0 && dy == 0) ? __builtin_bit_cast(f32x4_t, bq[j]) : acc[i][j], 0, 0, 0);
  }
}
template <int PW, bool FLIP, int D, int... K>
__device__ __forceinline__ void c64_rows_seq(std::integer_sequence<int, K...>, f32x4_t (&acc)[4][2], u32x4_t (&ring)[D + 1],
                                             const u32x4_t (&wB)[9][2][2], const unsigned (&pbd)[3][2], u32x4_t (&bq)[2]) {
  (c64_row_item<K, PW, FLIP, D>(acc, ring, wB, pbd, bq), ...);
}
template <int PW, int D, int... P>
__device__ __forceinline__ void c64_rows_prologue(std::integer_sequence<int, P...>, u32x4_t (&ring)[D + 1], const unsigned (&pbd)[3][2]) {
  (c64_row_issue<P, PW>(ring[P], pbd), ...);
}
template <int PW, bool FLIP = false, int D = 3>
__device__ __forceinline__ void c64_rows(f32x4_t (&acc)[4][2], const u32x4_t (&wB)[9][2][2], const unsigned (&pbd)[3][2], u32x4_t (&bq)[2]) {
  u32x4_t ring[D + 1];
  c64_rows_prologue<PW, D>(std::make_integer_sequence<int, D>{}, ring, pbd);
  c64_rows_seq<PW, FLIP, D>(std::make_integer_sequence<int, 36>{}, acc, ring, wB, pbd, bq);
}

// 2 bf16 of `o` zeroed where the mask element is not > 0 (packed 16-bit integer ops: a bf16 is > 0 iff its bits are > 0 as int16;
// op_sel_hi:[0,1]: the high lane takes the shift count from the LOW half of the inline constant too)
__device__ __forceinline__ uint32_t c64_mask2(uint32_t o, uint32_t m) {
  uint32_t t;
  asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_sub_i16 %0, 0, %0\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(t) : "v"(m));
  return o & t;
}

}  // namespace
