// Shared device helpers for libasr_hip.so (gfx950 / MI355X only).
//
// Conventions used by every kernel in this directory
//   * storage dtype T is `float` (parity mode) or `bf16_t` (= unsigned short, perf mode); accumulation is fp32.
//   * MFMA "pack": one lane holds KPACK consecutive k of one row/col of a 16x16 fragment:
//       bf16: KPACK = 8  -> one v_mfma_f32_16x16x32_bf16 per pack  (k-range of a macro step = 32)
//       f32 : KPACK = 4  -> four v_mfma_f32_16x16x4_f32 per pack   (k-range of a macro step = 16)
//     lane l: row/col = l & 15, k-block g = l >> 4, so a pack is always one aligned 16-byte read.
//     Both operands use the same k <-> (g, j) map, so the contraction is exact whatever the map is.
//   * C/D fragment (both dtypes): col = l & 15, row = 4 * (l >> 4) + reg   (reg = 0..3).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/asr_hip.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

#define ASR_WAVE 64

// ---------------------------------------------------------------------------------------------- conversions
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round to nearest even, NaN -> quiet NaN: the hardware conversion (v_cvt_pk_bf16_f32), one instruction instead of the five of
// the integer formulation (the same values for every finite input and for infinities)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }

template <typename T> struct DT;
template <> struct DT<float> {
  static constexpr int kDtype = ASR_F32;
  static constexpr int EPC = 4;          // elements per 16-byte chunk (= KPACK)
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float to(float v) { return v; }
  __device__ static __forceinline__ float from(float v) { return v; }
};
template <> struct DT<bf16_t> {
  static constexpr int kDtype = ASR_BF16;
  static constexpr int EPC = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ bf16_t to(float v) { return f32_to_bf16(v); }
  __device__ static __forceinline__ float from(bf16_t v) { return bf16_to_f32(v); }
};

// 16-byte chunk viewed as elements
template <typename T> union Chunk {
  uint4 v;
  T e[16 / sizeof(T)];
};

// ---------------------------------------------------------------------------------------------- MFMA wrapper
// acc += A_pack (x) B_pack for one 16x16 fragment over one macro step.
template <typename T> __device__ __forceinline__ void mma16(f32x4_t& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4_t& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------- transposing LDS read
// ds_read_b64_tr_b16 through the compiler builtin (NOT inline asm: the compiler then tracks lgkmcnt for the result and is
// free to issue several reads back to back and to schedule MFMAs under them).  In each 16-lane group, lane i supplying the
// address of T[p0 + (i >> 2)][c0 + 4 (i & 3)] (8 bytes) receives T[p0 .. p0+3][c0 + i]  (tools/probes/tr_read_probe.hip).
typedef __attribute__((ext_vector_type(4))) short asr_s16x4_t;
__device__ __forceinline__ uint2 asr_lds_read_tr16(const unsigned char* p) {
  const asr_s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) asr_s16x4_t*)(uintptr_t)p);
  return __builtin_bit_cast(uint2, v);
}

// Pixels 2 .. 9 of a run of 12 bf16 (r0 = 0 .. 3, r1 = 4 .. 7, r2 = 8 .. 11) as one MFMA operand: (r0.y, r1.x | r1.y, r2.x).  An operand must start
// on an even register and r0.y never does, so the compiler spends four v_mov on it; v_pk_mov_b32 moves two dwords at once and picks a half of
// each 64-bit source (low result = src0's half by op_sel[0], high result = src1's half by op_sel_hi[1]).
__device__ __forceinline__ bf16x8_t asr_shift2_of12(const uint2& r0, const uint2& r1, const uint2& r2) {
  uint2 lo, hi;
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(lo) : "v"(r0), "v"(r1));
  asm("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0]" : "=v"(hi) : "v"(r1), "v"(r2));
  return __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

// acc += the 8 bf16 of v (fp32 accumulation): v_dot2c_f32_bf16 against (1, 1), one instruction per two values.  One asm block: the
// accumulating chain is hazard-free, but a DOT result needs 3 wait states before any OTHER vector instruction reads it (4 before one
// overwrites it) and the compiler cannot see the opcode inside an asm -- hence the trailing s_nop.  (__builtin_amdgcn_fdot2_f32_bf16 picked one dword of the operand four
// times when it was tried here, hipcc 7.0.)
__device__ __forceinline__ void asr_sum8_bf16(float& acc, const bf16x8_t& v) {
  const u32x4_t u = __builtin_bit_cast(u32x4_t, v);
  asm("v_dot2c_f32_bf16 %0, %1, %2\n\tv_dot2c_f32_bf16 %0, %1, %3\n\tv_dot2c_f32_bf16 %0, %1, %4\n\tv_dot2c_f32_bf16 %0, %1, %5\n\ts_nop 3"
      : "+v"(acc) : "s"(0x3F803F80u), "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]));
}

// ---------------------------------------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------- dropout RNG
// Counter based: keep(seed, idx) is a pure function so backward regenerates the forward mask exactly.
// (The reference uses torch's Philox stream; RNG streams cannot match across implementations, SURVEY.md 4.3 --
//  parity is checked with dropout = 0, dropout itself by its statistics and fwd/bwd mask consistency.)
__device__ __forceinline__ uint32_t asr_hash32(uint64_t seed, uint64_t idx) {
  // 32-bit arithmetic only (full-rate multiplies; a 64-bit multiply costs 3-4 of them): the 64-bit counter and seed are
  // folded linearly, then two multiply / xor-shift rounds mix.  Quality is checked statistically (tests: keep rate and
  // neighbour correlations of the dropout masks).
  uint32_t x = (uint32_t)idx * 0x9E3779B1u + (uint32_t)(idx >> 32) * 0x85EBCA6Bu + (uint32_t)seed + (uint32_t)(seed >> 32) * 0xC2B2AE35u;
  x ^= x >> 15; x *= 0x2C1B3C6Du;
  x ^= x >> 12; x *= 0x297A2D39u;
  x ^= x >> 15;
  return x;
}
// Per-step seed: host part (distinct per dropout site) mixed with a DEVICE counter that a captured hipGraph advances
// on every replay (asr_step_advance), so replays do not repeat their dropout masks.
__device__ __forceinline__ uint64_t asr_mix_seed(uint64_t seed, const uint64_t* seed_dev) {
  return seed_dev ? seed + seed_dev[0] * 0xD1B54A32D192ED03ull : seed;
}
// threshold = (uint32)(p * 2^32); keep iff hash >= threshold
__device__ __forceinline__ bool asr_keep(uint64_t seed, uint64_t idx, uint32_t thr) { return asr_hash32(seed, idx) >= thr; }
static inline uint32_t asr_drop_threshold(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// ---------------------------------------------------------------------------------------------- host helpers
#define ASR_CHECK_ARG(cond)                 \
  do {                                      \
    if (!(cond)) return ASR_EINVAL;         \
  } while (0)
#define ASR_LAUNCH_CHECK()                                     \
  do {                                                         \
    if (hipGetLastError() != hipSuccess) return ASR_ELAUNCH;   \
  } while (0)

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// tuning switch set through asr_set_tuning (prof.hip), or `dflt`; the library itself reads no environment variables
int64_t asr_tuning(const char* name, int64_t dflt);

// profiling hook (prof.hip): brackets one launch with hipEvents when profiling of `op` is enabled.
struct AsrProfScope {
  int op;
  hipStream_t s;
  void* slot;
  AsrProfScope(int op, hipStream_t s);
  ~AsrProfScope();
};
