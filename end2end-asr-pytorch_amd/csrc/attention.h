// Shared declarations of the attention kernels (attention.hip: generic fp32/bf16 kernels for d in {16,32,64};
// attention_fast.hip: the bf16 / d = 64 kernels used by the reference configurations).
#pragma once
#include "common.h"

namespace asr_attn {

struct AttnArgs {
  const void *Q, *K, *V, *O, *dO;
  void *Out, *dQ, *dK, *dV;
  float* lse; float* delta; float* attn_out;
  float* Out32; const float* O32;   // optional fp32 copy of the output (same element strides as O): delta = rowsum(dO * O32)
  int B, H, Tq, Tk;
  int64_t q_sb, q_st, k_sb, k_st, v_sb, v_st, o_sb, o_st;
  const int32_t* key_len; const uint8_t* key_pad; int64_t m_sb, m_sq;
  int causal; float scale; uint32_t thr /* 16-bit dropout threshold */; float inv_keep; uint64_t seed; const uint64_t* seed_dev;
  int vec;     // all pointers 16-B aligned and all strides multiples of EPC
  int parts;   // backward: ASR_ATTN_DELTA | ASR_ATTN_DQ | ASR_ATTN_DKV (which of its three kernels this call launches)
};


// ---------------------------------------------------------------------------------------------- dropout on P
// Counter-based and cheap enough to sit inside the softmax loop: one 32-bit hash per PAIR of adjacent keys gives two 16-bit
// uniform fields; key k of row (b,h,q) is kept iff its field >= thr16 = round(p * 65536).  Forward, both backward kernels and the
// probability dump evaluate the same function, so the mask is identical everywhere (the reference draws from torch's Philox
// stream; RNG streams cannot match across implementations).
__device__ __forceinline__ uint32_t drop_row_key(uint64_t seed, uint32_t row) {
  // linear in the row (one multiply-add per row even where rows change per element); drop_pair_bits does the mixing
  return (uint32_t)seed + (uint32_t)(seed >> 32) * 0x85EBCA6Bu + row * 0x9E3779B1u;
}
constexpr uint32_t DROP_C1 = 0xC2B2AE35u;
// second half of drop_pair_bits, for callers that keep y0 = (row_key + key_pair) * DROP_C1 as a running sum (the product is linear
// modulo 2^32: one add per key pair instead of an add and a multiply)
__device__ __forceinline__ uint32_t drop_pair_mix(uint32_t y) {
  y ^= y >> 15;
  y *= 0x27D4EB2Fu;
  y ^= y >> 13;
  return y;
}
__device__ __forceinline__ uint32_t drop_pair_bits(uint32_t row_key, uint32_t key_pair) {
  return drop_pair_mix((row_key + key_pair) * DROP_C1);
}
// The keep mask applied to a packed bf16 pair w whose two uniform fields are y: field >= thr  <=>  saturating (field - (thr - 1)) != 0;
// min(., 1) is then 0 / 1 per half and multiplies the bit pattern.  Three packed 16-bit instructions, written out because the compiler
// turns the equivalent builtins into two compares, two selects and a byte permute.  thrm1x2 = (thr - 1) in both halves (thr >= 1).
__device__ __forceinline__ uint32_t drop_apply_pk(uint32_t w, uint32_t y, uint32_t thrm1x2, uint32_t ones) {
  uint32_t t, m, r;
  asm("v_pk_sub_u16 %0, %1, %2 clamp" : "=v"(t) : "v"(y), "v"(thrm1x2));
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(t), "v"(ones));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(m));
  return r;
}
__device__ __forceinline__ bool drop_keep(uint32_t row_key, int k, uint32_t thr16) {
  const uint32_t y = drop_pair_bits(row_key, (uint32_t)k >> 1);
  return ((k & 1) ? (y >> 16) : (y & 0xffffu)) >= thr16;
}
__device__ __forceinline__ uint32_t drop_row(const AttnArgs& p, int b, int h, int q) {
  return (uint32_t)((h * p.B + b) * p.Tq + q);
}

__device__ __forceinline__ float group_max(float v) {   // across the 4 lane groups that share lane&15
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ bool key_masked(const AttnArgs& p, int b, int kg, int q, int kend) {
  if (kg >= kend) return true;
  if (p.key_pad && p.key_pad[(int64_t)b * p.m_sb + (int64_t)(q < p.Tq ? q : p.Tq - 1) * p.m_sq + kg]) return true;
  if (p.causal && kg > q) return true;
  return false;
}
__device__ __forceinline__ int key_end(const AttnArgs& p, int b) {
  int kend = p.Tk;
  if (p.key_len) { int kl = p.key_len[b]; kend = kl < kend ? (kl < 0 ? 0 : kl) : kend; }
  return kend;
}


// fast path entry points (attention_fast.hip); return ASR_EUNSUPPORTED when the shape / layout is not theirs
int attn_fast_fwd(const AttnArgs& p, int d, int dtype, hipStream_t s);
int attn_fast_bwd(const AttnArgs& p, int d, int dtype, hipStream_t s);
// long-sequence forward (attention_pp.hip): bf16, d = 64, no causal mask; the caller has checked fast_ok()
int attn_pp_fwd(const AttnArgs& p, hipStream_t s);

}  // namespace asr_attn
