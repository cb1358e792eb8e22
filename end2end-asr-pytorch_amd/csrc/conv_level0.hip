// The full-resolution level of vgg_cnn without its full-resolution activations (reference: models/asr/transformer.py:42-47 --
// conv.0 (1 -> 64) + ReLU, conv.2 (64 -> 64) + ReLU, MaxPool2d(2, 2) -- and their autograd).
//
// At the benchmark shape (B = 32, 161 x 800) every 64-channel full-resolution tensor is 528 MB in bf16, and the level used to move
// 5.3 GB per step through HBM: conv.0 wrote y1, conv.2 read it, the pooling backward expanded the 131 MB pooled gradient into 528 MB,
// conv.2's weight gradient read that and y1, its data gradient read it again plus y1 as the ReLU mask and wrote dy1, and conv.0's
// weight gradient read dy1 (the data gradient alone: 1.73 GB in 307 us = 5.6 TB/s, HBM-bound at 39 % of the MFMA peak).  None of those
// tensors carries information that is not in the 16.5 MB of log-mel frames, the 131 MB pooled tensor and its selection codes:
//
//   forward  (vgg_level0_fwd_kernel)    src -> [conv.0 + ReLU as a K = 32 MFMA per 16 pixels, straight into the LDS halo patch]
//                                           -> conv.2 (register-resident weights, conv_c64_core.h) -> ReLU -> pool + selection codes
//   dgrad    (vgg_level0_dgrad_kernel)  pooled gradient + codes -> [expanded into the LDS halo patch] -> conv.2 data gradient
//                                           -> ReLU mask of conv.0 RECOMPUTED for the tile -> dW0 / db0 contracted from the tile in
//                                           registers (16x16x16 MFMAs over the pixels); dy1 is never stored
//   wgrad    (vgg_level0_wgrad_kernel)  src -> [conv.0 + ReLU into the LDS patch], pooled gradient + codes -> [expanded dY tile]
//                                           -> the 64 x 64 x 9 weight-gradient block of conv_wgrad_dma.hip
//
// conv.0 as ONE MFMA per 16 pixels x 16 channels (K = 32 = lane group x 8 slots): the fp32 frame value is split x = hi + lo into two bf16
// (hi = truncation, lo = round(x - hi): 16 mantissa bits), the weight is rounded to bf16 twice (w = whi + wlo), and the 27 products
// hi.whi + lo.whi + hi.wlo of a pixel's 3 x 3 window sit in the 32 slots -- lane group ky < 3 holds tap row ky as
// {hi0, hi1, hi2, hi0 | lo0, lo1, lo2, hi1} against {whi0, whi1, whi2, wlo0 | whi0, whi1, whi2, wlo1}, lane group 3 holds the third
// column {hi2 of rows 0, 1, 2} against {wlo2 of rows 0, 1, 2} (round 6; rounds 4 - 5 spent a second MFMA on the wlo terms and left
// group 3 empty).  Bias = the accumulator's initial value: products exact to 2^-16, fp32 accumulation -- the fp32 vector-ALU kernel it
// replaces (conv1_fwd) agrees to rounding of the bf16 result.  L0_WSPLIT = 0 (tuning) drops the wlo terms.
// The frame patch (12 x 20 values per 8 x 16 tile) arrives by 4-byte LDS-DMA and is split in place by the thread that fetched it.
#include "common.h"
#include "conv_c64_core.h"
#include "conv_level0.h"
#include "conv_wgrad_dma.h"

#include <cstdio>
#include <type_traits>
#include <utility>

extern "C" int asr_conv3x3_wgrad_reduce(const float* workspace, float* dw, int B, int H, int W, int Cin, int Cout, hipStream_t s);

namespace {

constexpr int L0_PB = 180 * 128;        // halo patch of 64 bf16 channels: 10 x 18 pixels, 16-B chunk c of the pixel in patch column x in slot c ^ (x & 7)
constexpr int L0_SS = 1536;             // frame patch buffer: 12 x 20 packed (hi << 16 | lo) dwords (960 B, padded to 1024), 256 B of (1.0 | 0), 256 B of zeros
constexpr int L0_WM = 4096;             // conv.0 weights as MFMA operands: [channel fragment 4][64 lanes][16 B]

typedef __attribute__((ext_vector_type(4))) short l0_s16x4_t;

#define L0_FENCE() asm volatile("" ::: "memory")
#ifdef L0_TIMING       // tuning builds only (ASR_HIPCC_EXTRA=-DL0_TIMING): s_memtime stamps at the forward kernel's section boundaries
#define L0_STAMP(K) { const long long now_ = clock64(); tsec[K] += now_ - tlast; tlast = now_; }
#else
#define L0_STAMP(K)
#endif

// LDS-DMA issued by hand (M0 saved / restored; the compiler neither counts these loads nor drains them before its own LDS reads)
__device__ __forceinline__ void l0_dma4(unsigned lds_wave_base, const void* src) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_wave_base), "v"(src) : "memory");
}
__device__ __forceinline__ void l0_dma16(unsigned lds_wave_base, const void* src) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_wave_base), "v"(src) : "memory");
}

// fp32 frame value -> (hi << 16) | lo, hi = the top 16 bits (a bf16, exact), lo = bf16(x - hi)
__device__ __forceinline__ uint32_t l0_split(float x) {
  const uint32_t xb = __float_as_uint(x), hb = xb & 0xffff0000u;
  return hb | (uint32_t)f32_to_bf16(x - __uint_as_float(hb));
}

// One conv.0 operand of a pixel: three packed frame values at p, p + stride, p + 2 stride -- lane groups 0 .. 2: the tap row's three
// columns (stride 4), lane group 3: the third column of the three tap rows (stride 80 = one frame-patch row) -- as the slots
// {hi0, hi1, hi2, hi0 | lo0, lo1, lo2, hi1}
__device__ __forceinline__ u32x4_t l0_frame_operand(const unsigned char* p, int stride) {
  const uint32_t d0 = *reinterpret_cast<const uint32_t*>(p), d1 = *reinterpret_cast<const uint32_t*>(p + stride),
                 d2 = *reinterpret_cast<const uint32_t*>(p + 2 * stride);
  u32x4_t r;
  r[0] = __builtin_amdgcn_perm(d1, d0, 0x07060302u);
  r[1] = __builtin_amdgcn_perm(d0, d2, 0x07060302u);
  r[2] = __builtin_amdgcn_perm(d1, d0, 0x05040100u);
  r[3] = __builtin_amdgcn_perm(d1, d2, 0x07060100u);
  return r;
}
// the lane's part of a frame-patch address: tap row g of the pixel for g < 3, row 0 / third column for lane group 3; and its stride
__device__ __forceinline__ int l0_frame_lane_off(int g) { return g < 3 ? g * 80 : 8; }
__device__ __forceinline__ int l0_frame_lane_stride(int g) { return g < 3 ? 4 : 80; }

// conv.0 weights of output channel `co` as the matching operand of lane group g
__device__ __forceinline__ u32x4_t l0_weight_operand(const float* w0, int co, int g, bool split) {
  uint32_t h[9], l[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float w = w0[co * 9 + t];
    h[t] = f32_to_bf16(w);
    l[t] = split ? (uint32_t)f32_to_bf16(w - bf16_to_f32((bf16_t)h[t])) : 0u;
  }
  u32x4_t r = u32x4_t{l[2] | (l[5] << 16), l[8], 0u, 0u};
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (g == q) r = u32x4_t{h[3 * q] | (h[3 * q + 1] << 16), h[3 * q + 2] | (l[3 * q] << 16), h[3 * q] | (h[3 * q + 1] << 16), h[3 * q + 2] | (l[3 * q + 1] << 16)};
  return r;
}

// tile walk of the persistent kernels: origins advance by a fixed (images, tile rows, tile columns) step with carries (conv_c64.hip)
struct L0Org { int b, h0, w0; };
struct L0Walk {
  int dtw, dth, db, wlim, hlim;
  __device__ __forceinline__ void init(int nwg, int tiles_w, int tiles_h) {
    dtw = (nwg % tiles_w) * 16;
    const int q1 = nwg / tiles_w;
    dth = (q1 % tiles_h) * 8;
    db = q1 / tiles_h;
    wlim = tiles_w * 16; hlim = tiles_h * 8;
  }
  __device__ __forceinline__ void advance(L0Org& o) const {
    o.w0 += dtw;
    const bool c1 = o.w0 >= wlim;
    o.w0 -= c1 ? wlim : 0;
    o.h0 += dth + (c1 ? 8 : 0);
    const bool c2 = o.h0 >= hlim;
    o.h0 -= c2 ? hlim : 0;
    o.b += db + (c2 ? 1 : 0);
  }
  __device__ __forceinline__ void first(L0Org& o, int t, int tiles_w, int tiles_h) const {
    o.w0 = (t % tiles_w) * 16; t /= tiles_w;
    o.h0 = (t % tiles_h) * 8; o.b = t / tiles_h;
  }
};

// the frame patch of a tile (rows h0 - 2 .. h0 + 9, columns w0 - 2 .. w0 + 17; zeros outside the image): one 4-byte DMA per thread
__device__ __forceinline__ void l0_stage_frames(const L0Args& p, const L0Org& o, unsigned sbuf_lds, int tid, int wave_u) {
  if (tid < 240) {
    const int sr = tid / 20, sc = tid - sr * 20;
    const int gy = o.h0 + sr - 2, gx = o.w0 + sc - 2;
    const bool in = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
    const float* s = in ? p.src + ((int64_t)o.b * p.H + gy) * p.W + gx : reinterpret_cast<const float*>(&c64_zero_page);
    l0_dma4(sbuf_lds + (unsigned)wave_u * 256u, s);
  }
}
__device__ __forceinline__ void l0_split_frames(unsigned char* sbuf, int tid) {      // after the thread's own DMA has landed
  if (tid < 240) {
    uint32_t* q = reinterpret_cast<uint32_t*>(sbuf) + tid;
    *q = l0_split(__uint_as_float(*q));
  }
}

// ---------------------------------------------------------------------------------------------- conv.0 + ReLU into a halo patch
// 180 halo pixels = 12 fragments of 16 (the last one 4 pixels); fragment f, lane lr <-> halo pixel hp = 16 f + lr = patch (hp / 18, hp % 18).
// Swapped operands (A = weights): a lane gets 4 consecutive channels of ONE pixel = 8 bytes of the pixel's 128-byte patch row.
struct L0GenLane {          // per (lane, fragment) addressing, recomputed per tile (a dozen vector instructions; 36 registers if kept)
  int gaddr;                // frame-patch byte offset of the lane's tap row for this pixel
  int waddr;                // patch byte offset of the lane's 8 output bytes for channel fragment 0 (other fragments: ^ (cf << 5))
  int prc;                  // patch row | patch column << 8 | (hp < 180) << 16
  __device__ __forceinline__ void init(int f, int lr, int g) {
    const int hp = f * 16 + lr, hpc = hp < 180 ? hp : 179;
    const int pr = (hpc * 3641) >> 16, pc = hpc - pr * 18;          // hpc / 18 for hpc < 192
    gaddr = (hpc + 2 * pr) * 4 + l0_frame_lane_off(g);              // (pr * 20 + pc) * 4 + the lane group's tap row / column
    waddr = hpc * 128 + ((((g >> 1) ^ (pc & 7))) << 4) + (g & 1) * 8;
    prc = pr | (pc << 8) | ((hp < 180 ? 1 : 0) << 16);
  }
};

// One unit = (16 halo pixels) x (16 channels), in two stages so that a caller can run the MFMAs of several units back to back and
// convert / store behind them (a unit alone is a chain MFMA -> MFMA -> convert -> ReLU -> store of ~190 cycles; twelve of them in
// sequence cost as much as the tile's 144 main MFMAs -- in-kernel section timing, profiles/r04_level0_structure_ab.txt).
__device__ __forceinline__ f32x4_t l0_gen_mfma(const u32x4_t& bop, const u32x4_t& w, const f32x4_t& bias) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, bop), bias, 0, 0, 0);
}
// `wa`: the lane's output offset for channel fragment 0 -- L0GenLane::waddr, or the offset of a 128-byte dump area for the lanes of the
// last fragment that lie past the patch (an address select instead of a branch keeps the units in ONE basic block)
__device__ __forceinline__ void l0_gen_store(unsigned char* ypatch, int wa, const f32x4_t& a, int cf, uint32_t vmask) {
  // halo pixels outside the image are conv.2's zero padding (vmask), not conv.0 of padded frames
  uint32_t pa = pack_bf16(a[0], a[1]), pb = pack_bf16(a[2], a[3]);
  asm("v_pk_max_i16 %0, %0, 0" : "+v"(pa));          // ReLU on the bf16 bit patterns (conv_c64.hip); plain asm: free to be scheduled
  asm("v_pk_max_i16 %0, %0, 0" : "+v"(pb));
  *reinterpret_cast<uint2*>(ypatch + (wa ^ (cf << 5))) = make_uint2(pa & vmask, pb & vmask);
}
__device__ __forceinline__ void l0_gen_store_inside(unsigned char* ypatch, int wa, const f32x4_t& a, int cf) {
  uint32_t pa = pack_bf16(a[0], a[1]), pb = pack_bf16(a[2], a[3]);
  asm("v_pk_max_i16 %0, %0, 0" : "+v"(pa));
  asm("v_pk_max_i16 %0, %0, 0" : "+v"(pb));
  *reinterpret_cast<uint2*>(ypatch + (wa ^ (cf << 5))) = make_uint2(pa, pb);
}
__device__ __forceinline__ uint32_t l0_halo_valid(const L0Args& p, const L0Org& o, int prc) {
  const int gy = o.h0 + (prc & 0xff) - 1, gx = o.w0 + ((prc >> 8) & 0xff) - 1;
  return ((unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? 0xffffffffu : 0u;
}

// pooled 2 x 2 max + selection codes from the accumulators of an 8 x 16 tile (the POOL epilogue of conv3x3_c64_kernel, y never stored).
// A lane holds 4 channels x 2 channel fragments of ONE pixel per tile row; the two pixels of a pooling window sit in a lane pair.  The
// pair first trades fragments -- the even lane keeps fragment 0 of both pixels, the odd lane fragment 1 (one v_cndmask with a DPP source
// per dword) -- so that every lane then owns WHOLE windows of half the channels: maxima and codes are computed once per window instead
// of twice per lane pair (round 6: 217 -> ~130 vector instructions per wave and tile).  ReLU comes after the maximum: the int16 order of
// bf16 bit patterns is the value order wherever a positive value exists, and a window without one pools to 0 / code 0 either way.
// (k0[d], k1[d]: this lane's pixel, fragments 0 / 1.  L = fragment 0 of the even lane's pixel on the even lane, fragment 1 of it on the odd
// lane; R = the odd lane's pixel likewise.  v_cndmask_b32_dpp takes VCC implicitly, so the four selects of a tile row share one asm block;
// the leading s_nop covers the VALU-write -> DPP-read wait states the compiler cannot see inside an asm.)
__device__ __forceinline__ void l0_pair_trade(const uint32_t (&k0)[2], const uint32_t (&k1)[2], uint64_t even_lanes, uint32_t (&L)[2], uint32_t (&R)[2]) {
  asm("s_nop 1\n\ts_mov_b64 vcc, %8\n\t"
      "v_cndmask_b32_dpp %0, %5, %4, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %1, %7, %6, vcc quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_not_b64 vcc, vcc\n\t"
      "v_cndmask_b32_dpp %2, %4, %5, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
      "v_cndmask_b32_dpp %3, %6, %7, vcc quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf"
      : "=&v"(L[0]), "=&v"(L[1]), "=&v"(R[0]), "=&v"(R[1]) : "v"(k0[0]), "v"(k1[0]), "v"(k0[1]), "v"(k1[1]), "s"(even_lanes) : "vcc", "scc");
}
// lane_poff: the lane's element offset inside the tile's pooled block -- ((2 wm + (g & 1)) W/2 + lr / 2) 64 + its 8 channels -- fixed for
// the kernel; the tile's own offset is wave-uniform (scalar registers, folded into the store's base address).
__device__ __forceinline__ void l0_pool_epilogue(const L0Args& p, f32x4_t (&acc)[4][2], int tl, int b, int h0, int w0, unsigned lane_poff) {
  const uint64_t even_lanes = 0x5555555555555555ull;
  uint32_t L[4][2], R[4][2];      // [tile row][channel pair]: left / right pixel of the window column, this lane's fragment
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t k0[2], k1[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      k0[d] = pack_bf16(acc[i][0][2 * d], acc[i][0][2 * d + 1]);
      k1[d] = pack_bf16(acc[i][1][2 * d], acc[i][1][2 * d + 1]);
    }
    l0_pair_trade(k0, k1, even_lanes, L[i], R[i]);
  }
  const uint32_t one = 0x00010001u;
  uint32_t pm[2][2], cb[2];
#pragma unroll
  for (int pr = 0; pr < 2; ++pr) {
    uint32_t cw[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const uint32_t v0 = L[2 * pr][d], v1 = R[2 * pr][d], v2 = L[2 * pr + 1][d], v3 = R[2 * pr + 1][d];
      uint32_t ma, mb, m;
      asm("v_pk_max_i16 %0, %1, %2" : "=v"(ma) : "v"(v0), "v"(v1));
      asm("v_pk_max_i16 %0, %1, %2" : "=v"(mb) : "v"(v2), "v"(v3));
      asm("v_pk_max_i16 %0, %0, %1" : "+v"(ma) : "v"(mb));
      asm("v_pk_max_i16 %0, %1, 0" : "=v"(m) : "v"(ma));          // ReLU (conv_c64.hip): the pooled value
      // code = 0 where the maximum is 0, else 1 + the first window position (row-major) that holds it
      uint32_t n0 = v0 ^ m, n1 = v1 ^ m, n2 = v2 ^ m, nz;
      asm("v_pk_min_u16 %0, %0, %1" : "+v"(n0) : "v"(one));
      asm("v_pk_min_u16 %0, %0, %1" : "+v"(n1) : "v"(one));
      asm("v_pk_min_u16 %0, %0, %1" : "+v"(n2) : "v"(one));
      asm("v_pk_min_u16 %0, %1, %2" : "=v"(nz) : "v"(m), "v"(one));
      const uint32_t n01 = n0 & n1;
      uint32_t t1, t2, c;
      asm("v_pk_mad_u16 %0, %1, %2, %1" : "=v"(t1) : "v"(n0), "v"(n1));            // n0 + n0 n1
      asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(t2) : "v"(n01), "v"(n2), "v"(t1));   // + n0 n1 n2
      asm("v_pk_mad_u16 %0, %1, %2, %2" : "=v"(c) : "v"(t2), "v"(nz));              // (1 + ...) where the maximum is not 0
      pm[pr][d] = m;
      cw[d] = c;
    }
    cb[pr] = __builtin_amdgcn_perm(cw[1], cw[0], 0x06040200u);      // 4 selection bytes of channels 4 g .. 4 g + 3
  }
  // lane rows g, g ^ 1 trade pooled rows: the even row keeps pooled row 0 of 8 consecutive channels, the odd row pooled row 1
  const auto s0 = __builtin_amdgcn_permlane16_swap(pm[0][0], pm[1][0], false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(pm[0][1], pm[1][1], false, false);
  const auto sc = __builtin_amdgcn_permlane16_swap(cb[0], cb[1], false, false);
  const int H2 = p.H >> 1, W2 = p.W >> 1;
  const int64_t tile_off = (((int64_t)b * H2 + (h0 >> 1)) * W2 + (w0 >> 1)) * 64;
  bool st = true;
  if (h0 + 8 > 2 * H2 || w0 + 16 > 2 * W2) {       // (wave-uniform: a tile on the lower / right edge)
    const int g = (tl >> 4) & 3;
    st = (h0 >> 1) + (tl >> 7) * 2 + (g & 1) < H2 && (w0 >> 1) + ((tl & 15) >> 1) < W2;
  }
  if (st) {
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.pool + tile_off) + lane_poff * 2u) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    *reinterpret_cast<uint2*>(p.code + tile_off + lane_poff) = make_uint2(sc[0], sc[1]);
  }
}

// ================================================================================================ forward
__global__ __launch_bounds__(256, 2) void vgg_level0_fwd_kernel(L0Args p) {
  constexpr int PW = 18, CB = 1;
  constexpr int S_OFF = 2 * L0_PB, WM_OFF = S_OFF + 2 * L0_SS, B0_OFF = WM_OFF + L0_WM, B2_OFF = B0_OFF + 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nwg = gridDim.x;
  const int vid = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;
  const int cnt = vid < p.ntiles ? (p.ntiles - vid + nwg - 1) / nwg : 0;
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // conv.2 weights: A operand of every main MFMA, resident in registers
  u32x4_t wB[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wB[tap][ms][j] = *reinterpret_cast<const u32x4_t*>(p.wk + ((int64_t)(wn * 32 + j * 16 + lr) * 9 + tap) * 64 + ms * 32 + g * 8);
  {   // conv.0 operands (wave = channel fragment) and both biases into LDS
    *reinterpret_cast<u32x4_t*>(smem + WM_OFF + (wave * 64 + lane) * 16) = l0_weight_operand(p.w0, wave * 16 + lr, g, p.wsplit != 0);
    if (tid < 64) {
      reinterpret_cast<float*>(smem + B0_OFF)[tid] = p.b0 ? p.b0[tid] : 0.f;
      reinterpret_cast<float*>(smem + B2_OFF)[tid] = p.b2 ? p.b2[tid] : 0.f;
    }
  }
  L0Walk walk;
  walk.init(nwg, p.tiles_w, p.tiles_h);
  L0Org org[3];
  walk.first(org[0], vid, p.tiles_w, p.tiles_h);
  org[1] = org[0]; walk.advance(org[1]);
  org[2] = org[1]; walk.advance(org[2]);

  const unsigned lane_poff = (unsigned)(((wm * 2 + (g & 1)) * (p.W >> 1) + (lr >> 1)) * 64 + wn * 32 + (lr & 1) * 16 + (g & 2) * 4);

  // the wave's three pixel fragments: frame-patch and halo-patch offsets of the lane, fixed for the kernel (round 6: held in registers --
  // recomputing them per tile was 60 of the generation's 180 vector instructions)
  int g_addr[3], w_addr[3], g_prc[3];
  const int gstride = l0_frame_lane_stride(g);
#pragma unroll
  for (int fi = 0; fi < 3; ++fi) {
    L0GenLane gl;
    gl.init(wave * 3 + fi, lr, g);
    g_addr[fi] = gl.gaddr;
    w_addr[fi] = (gl.prc & 0x10000) ? gl.waddr : -1;      // -1: a lane of the last fragment past the patch (dump area)
    g_prc[fi] = gl.prc;
  }
  auto generate = [&](int yb, int sb, const L0Org& o) __attribute__((always_inline)) {
    unsigned char* yp = smem + yb * L0_PB;
    const unsigned char* sp = smem + S_OFF + sb * L0_SS;
    const bool inside = o.h0 >= 1 && o.w0 >= 1 && o.h0 + 9 <= p.H && o.w0 + 17 <= p.W;
    u32x4_t bop[3];
    int wa[3];
#pragma unroll
    for (int fi = 0; fi < 3; ++fi) {
      bop[fi] = l0_frame_operand(sp + g_addr[fi], gstride);
      wa[fi] = w_addr[fi] >= 0 ? w_addr[fi] : (S_OFF + 1024 - yb * L0_PB);      // dump area: the unused tail of frame buffer 0
    }
    // two halves of two channel fragments: every LDS read of a half first, then its 6 MFMAs back to back, then the six convert / store tails
    auto half = [&](int h, auto masked) __attribute__((always_inline)) {
      u32x4_t wh[2];
      f32x4_t acc6[2][3];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int cf = 2 * h + c;
        wh[c] = *reinterpret_cast<const u32x4_t*>(smem + WM_OFF + (cf * 64 + lane) * 16);
        acc6[c][0] = acc6[c][1] = acc6[c][2] = *reinterpret_cast<const f32x4_t*>(smem + B0_OFF + (cf * 16 + 4 * g) * 4);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) acc6[c][fi] = l0_gen_mfma(bop[fi], wh[c], acc6[c][fi]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int fi = 0; fi < 3; ++fi) {
          if constexpr (decltype(masked)::value) l0_gen_store(yp, wa[fi], acc6[c][fi], 2 * h + c, l0_halo_valid(p, o, g_prc[fi]));
          else l0_gen_store_inside(yp, wa[fi], acc6[c][fi], 2 * h + c);
        }
    };
    if (inside) {       // (wave-uniform) no halo pixel of the tile lies outside the image: nothing to mask
      half(0, std::false_type{});
      half(1, std::false_type{});
    } else {
      half(0, std::true_type{});
      half(1, std::true_type{});
    }
  };

  if (cnt > 0) l0_stage_frames(p, org[0], smem_base + S_OFF, tid, wave_u);
  if (cnt > 1) l0_stage_frames(p, org[1], smem_base + S_OFF + L0_SS, tid, wave_u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  l0_split_frames(smem + S_OFF, tid);
  l0_split_frames(smem + S_OFF + L0_SS, tid);
  __syncthreads();
  if (cnt > 0) generate(0, 0, org[0]);

  unsigned pbd[3][2];           // operand addresses of the CURRENT tile's patch buffer (toggled per tile, not re-derived from a second copy)
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
      pbd[dx][ms] = smem_base + (unsigned)((((wm * 4) / CB) * PW + ((wm * 4) % CB) * 16 + lr) * 128) +
                    (unsigned)(((ms * 4 + g) ^ ((lr + dx) & 7)) << 4);

#ifdef L0_TIMING
  long long tsec[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  for (int n = 0; n < cnt; ++n) {
    L0_STAMP(5)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's patch / frame-split writes are in LDS
    __builtin_amdgcn_s_barrier();       // patch n complete, frame patch n + 1 split; everybody is done with tile n - 1
    L0_FENCE();
    L0_STAMP(0)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (n + 2 < cnt) l0_stage_frames(p, org[2], smem_base + S_OFF + (unsigned)((n & 1) * L0_SS), tl, wave_u);
    if (n + 1 < cnt) generate((n + 1) & 1, (n + 1) & 1, org[1]);
    L0_FENCE();
    L0_STAMP(1)

    u32x4_t bq[2];
    const unsigned bias_addr = smem_base + (unsigned)(B2_OFF + (((tl >> 6) & 1) * 32 + 4 * ((tl >> 4) & 3)) * 4);
    lds_read16(bq[0], bias_addr);
    lds_read16(bq[1], bias_addr + 64);
    f32x4_t acc[4][2];
    c64_rows<PW>(acc, wB, pbd, bq);

    L0_FENCE();
    L0_STAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // frame patch n + 2 (issued a whole tile ago) -- and last tile's stores
    L0_STAMP(3)
    if (n + 2 < cnt) l0_split_frames(smem + S_OFF + (n & 1) * L0_SS, tl);
    l0_pool_epilogue(p, acc, tl, org[0].b, org[0].h0, org[0].w0, lane_poff);
    L0_STAMP(4)
    org[0] = org[1]; org[1] = org[2];
    walk.advance(org[2]);
    const unsigned flip = (n & 1) ? (unsigned)(-L0_PB) : (unsigned)L0_PB;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) pbd[dx][ms] += flip;
  }
#ifdef L0_TIMING
  if (p.dbg && blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 6; ++k) p.dbg[wave * 8 + k] = tsec[k];
    p.dbg[wave * 8 + 7] = cnt;
  }
#endif
}

// ================================================================================================ data gradient + dW0 / db0
// Pooled gradient patch of an 8 x 16 tile's halo: pooled rows h0/2 - 1 .. h0/2 + 4, columns w0/2 - 1 .. w0/2 + 8 (60 pooled pixels):
// values 60 x 128 B and selection codes 60 x 64 B by 16-byte DMA, then every (pooled pixel, 8-channel chunk) item is expanded into
// its four window positions of the halo patch: out[k] = value where code == 1 + k.
constexpr int L0_PV = 60 * 128, L0_PST = 60 * 192;      // staged bytes: values, values + codes

__device__ __forceinline__ void l0_stage_pooled(const L0Args& p, const L0Org& o, unsigned pst_lds, int t, int wave_u) {
  const int H2 = p.H >> 1, W2 = p.W >> 1;
  const int ph0 = (o.h0 >> 1) - 1, pw0 = (o.w0 >> 1) - 1;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(&c64_zero_page);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ci = t + it * 256;
    if (ci < 480) {
      const int q = ci >> 3, c = ci & 7, qr = q / 10, qc = q - qr * 10;
      const int ph = ph0 + qr, pw = pw0 + qc;
      const bool in = (unsigned)ph < (unsigned)H2 && (unsigned)pw < (unsigned)W2;
      const unsigned char* s = in ? reinterpret_cast<const unsigned char*>(p.dpool) + ((((int64_t)o.b * H2 + ph) * W2 + pw) * 64 + c * 8) * 2 : zero;
      l0_dma16(pst_lds + (unsigned)(it * 4096) + (unsigned)wave_u * 1024u, s);
    }
  }
  if (t < 240) {
    const int q = t >> 2, qr = q / 10, qc = q - qr * 10;
    const int ph = ph0 + qr, pw = pw0 + qc;
    const bool in = (unsigned)ph < (unsigned)H2 && (unsigned)pw < (unsigned)W2;
    const unsigned char* s = in ? p.code + (((int64_t)o.b * H2 + ph) * W2 + pw) * 64 + (t & 3) * 16 : zero;
    l0_dma16(pst_lds + (unsigned)L0_PV + (unsigned)wave_u * 1024u, s);
  }
}

// selection codes of 8 channels (8 bytes) -> for window position k the 16-bit keep masks of the 8 channels (4 dwords)
struct L0CodeMasks {
  uint32_t oh[4];           // per 16-bit half: 1 << code
  __device__ __forceinline__ void init(uint2 k) {
    uint32_t c16[4];
    c16[0] = __builtin_amdgcn_perm(0u, k.x, 0x0C010C00u);
    c16[1] = __builtin_amdgcn_perm(0u, k.x, 0x0C030C02u);
    c16[2] = __builtin_amdgcn_perm(0u, k.y, 0x0C010C00u);
    c16[3] = __builtin_amdgcn_perm(0u, k.y, 0x0C030C02u);
    const uint32_t one = 0x00010001u;
#pragma unroll
    for (int d = 0; d < 4; ++d) asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(oh[d]) : "v"(c16[d]), "v"(one));
  }
  template <int K>          // K = 1 + window position: bit K of the one-hot word, spread over the half
  __device__ __forceinline__ uint32_t mask(int d) const {
    uint32_t m;
    asm("v_pk_lshlrev_b16 %0, %2, %1 op_sel_hi:[0,1]\n\tv_pk_ashrrev_i16 %0, 15, %0 op_sel_hi:[0,1]" : "=&v"(m) : "v"(oh[d]), "n"(15 - K));
    return m;
  }
};

__global__ __launch_bounds__(256, 2) void vgg_level0_dgrad_kernel(L0Args p) {
  constexpr int PW = 18, CB = 1;
  constexpr int PST_OFF = 2 * L0_PB, S_OFF = PST_OFF + 2 * L0_PST, WM_OFF = S_OFF + 2 * L0_SS, B0_OFF = WM_OFF + L0_WM, DUMP_OFF = B0_OFF + 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int nwg = gridDim.x;
  const int vid = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;
  const int cnt = vid < p.ntiles ? (p.ntiles - vid + nwg - 1) / nwg : 0;
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;

  // tap-flipped conv.2 weights (ci, flipped tap, co): here the B operand (columns = input channels of conv.2 = this kernel's outputs)
  u32x4_t wB[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wB[tap][ms][j] = *reinterpret_cast<const u32x4_t*>(p.wk + ((int64_t)(wn * 32 + j * 16 + lr) * 9 + tap) * 64 + ms * 32 + g * 8);
  {
    *reinterpret_cast<u32x4_t*>(smem + WM_OFF + (wave * 64 + lane) * 16) = l0_weight_operand(p.w0, wave * 16 + lr, g, p.wsplit != 0);
    if (tid < 64) {
      reinterpret_cast<float*>(smem + B0_OFF)[tid] = p.b0 ? p.b0[tid] : 0.f;
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        reinterpret_cast<uint32_t*>(smem + S_OFF + sb * L0_SS + 1024)[tid] = 0x3F800000u;      // (hi, lo) = (1.0, 0): the bias-gradient row
        reinterpret_cast<uint32_t*>(smem + S_OFF + sb * L0_SS + 1280)[tid] = 0u;
        if (tid < 16) reinterpret_cast<uint32_t*>(smem + S_OFF + sb * L0_SS + 960)[tid] = 0u;
      }
    }
  }
  L0Walk walk;
  walk.init(nwg, p.tiles_w, p.tiles_h);
  L0Org org[3];
  walk.first(org[0], vid, p.tiles_w, p.tiles_h);
  org[1] = org[0]; walk.advance(org[1]);
  org[2] = org[1]; walk.advance(org[2]);

  // expansion items of this thread: (pooled pixel q, chunk c) = tid and tid + 256 (< 480).  xa[it][dx]: patch byte offset of window
  // column dx in window row 0 (row 1: + 18 * 128); flags: bit 0/1 = window row 0/1 inside the patch, bit 2/3 = column 0/1
  // xa[it][dx] (multiples of 16) carry the validity flags in their low bits: xa[it][0] bit 0/1 = window row 0/1 inside the patch,
  // bit 2 = window column 0 inside; xa[it][1] bit 0 = window column 1 inside, bit 1 = the item exists
  int xa[2][2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = tid + it * 256, q = item >> 3, c = item & 7, qr = q / 10, qc = q - qr * 10;
    const int rowbase = (2 * qr - 1) * (18 * 128);
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int x = 2 * qc + dx - 1;
      xa[it][dx] = rowbase + x * 128 + ((c ^ (x & 7)) << 4);
    }
    const int ok = item < 480 ? 1 : 0;
    xa[it][0] |= ok * ((qr > 0 ? 1 : 0) | (qr < 5 ? 2 : 0) | (qc > 0 ? 4 : 0));
    xa[it][1] |= ok * ((qc < 9 ? 1 : 0) | 2);
  }
  // (window positions that fall outside the patch -- and the second item of the threads that have none -- go to a 16-byte dump slot by
  // an address select: no branches, the 8 stores of a thread issue back to back)
  auto expand = [&](int yb, int ps) __attribute__((always_inline)) {
    unsigned char* yp = smem + yb * L0_PB;
    const unsigned char* pp = smem + PST_OFF + ps * L0_PST;
    const int dump = DUMP_OFF - yb * L0_PB;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it == 0 ? tid : (tid < 224 ? tid + 256 : tid);
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(pp + item * 16);
      const uint2 k = *reinterpret_cast<const uint2*>(pp + L0_PV + item * 8);
      L0CodeMasks cm;
      cm.init(k);
      u32x4_t o1, o2, o3, o4;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        o1[d] = v[d] & cm.mask<1>(d); o2[d] = v[d] & cm.mask<2>(d);
        o3[d] = v[d] & cm.mask<3>(d); o4[d] = v[d] & cm.mask<4>(d);
      }
      const int f0 = xa[it][0] & 15, f1 = xa[it][1] & 15, a0 = xa[it][0] & ~15, a1 = xa[it][1] & ~15;
      const bool c0 = (f0 & 4) != 0, c1 = (f1 & 3) == 3;
      *reinterpret_cast<u32x4_t*>(yp + (((f0 & 1) && c0) ? a0 : dump)) = o1;
      *reinterpret_cast<u32x4_t*>(yp + (((f0 & 1) && c1) ? a1 : dump)) = o2;
      *reinterpret_cast<u32x4_t*>(yp + (((f0 & 2) && c0) ? a0 + 18 * 128 : dump)) = o3;
      *reinterpret_cast<u32x4_t*>(yp + (((f0 & 2) && c1) ? a1 + 18 * 128 : dump)) = o4;
    }
  };

  // ---- prologue: pooled patches 0 (expanded now) and 1, frame patch 0
  if (cnt > 0) {
    l0_stage_pooled(p, org[0], smem_base + PST_OFF, tid, wave_u);
    l0_stage_frames(p, org[0], smem_base + S_OFF, tid, wave_u);
  }
  if (cnt > 1) l0_stage_pooled(p, org[1], smem_base + PST_OFF + L0_PST, tid, wave_u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  l0_split_frames(smem + S_OFF, tid);
  __syncthreads();
  if (cnt > 0) expand(0, 0);

  unsigned offk[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
      offk[dx][ms] = smem_base + (unsigned)((((wm * 4) / CB) * PW + ((wm * 4) % CB) * 16 + lr) * 128) +
                     (unsigned)(((ms * 4 + g) ^ ((lr + dx) & 7)) << 4);
  // epilogue addressing (frame patch, packed dwords): conv.0 recomputed for the tile pixels of this wave (pixel column lr, tap row g),
  // and the frame values under every tap for 4 consecutive pixels 4 g .. 4 g + 3 (tap = lr; 9 = the row of ones, above = zeros)
  const int mbase = ((wm * 4 + 1) * 20 + lr + 1) * 4 + l0_frame_lane_off(g), mstride = l0_frame_lane_stride(g);
  const int tbase = lr < 9 ? ((wm * 4 + 1 + lr / 3) * 20 + 4 * g + 1 + lr % 3) * 4 : (lr == 9 ? 1024 : 1280);
  f32x4_t dw0[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};      // [tap rows 4 g + r][channel lr of fragment j]

  for (int n = 0; n < cnt; ++n) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // patch n expanded, pooled patch n + 1 landed, frame patch n split
    L0_FENCE();
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (n + 2 < cnt) l0_stage_pooled(p, org[2], smem_base + PST_OFF + (unsigned)((n & 1) * L0_PST), tl, wave_u);
    if (n + 1 < cnt) {
      l0_stage_frames(p, org[1], smem_base + S_OFF + (unsigned)(((n + 1) & 1) * L0_SS), tl, wave_u);
      expand((n + 1) & 1, (n + 1) & 1);
    }
    L0_FENCE();

    u32x4_t bq[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
    f32x4_t acc[4][2];
    unsigned pbd[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) pbd[dx][ms] = offk[dx][ms] + (unsigned)((n & 1) * L0_PB);
    c64_rows<PW, true>(acc, wB, pbd, bq);   // acc[i][j]: rows = pixels 4 g + r of tile row 4 wm + i, column = channel

    L0_FENCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // pooled patch n + 2 and frame patch n + 1
    if (n + 1 < cnt) l0_split_frames(smem + S_OFF + ((n + 1) & 1) * L0_SS, tl);

    // ---- epilogue: ReLU mask of conv.0 recomputed, then dW0 / db0 += frames^T . dy1 over the tile's pixels
    const unsigned char* sp = smem + S_OFF + (n & 1) * L0_SS;
    const int h0 = org[0].h0, w0 = org[0].w0;
    const bool whole = h0 + 8 <= p.H && w0 + 16 <= p.W;
    float bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bj[j] = *reinterpret_cast<const float*>(smem + B0_OFF + (wn * 32 + j * 16 + lr) * 4);
    // (a tile on the lower / right edge -- wave-uniform, 1 in 10 at the benchmark shape -- first clears the gradient of its pixels outside the
    // image; the mask itself is then conv.0's sign alone: one compare and one select per element.  Folded into ONE condition the row /
    // column tests cost the compiler six vector instructions per element, round 6.)
    if (!whole) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool rowok = wm * 4 + i < p.H - h0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool ok = rowok && 4 * g + r < p.W - w0;
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j][r] = ok ? acc[i][j][r] : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4_t am = l0_frame_operand(sp + mbase + i * 80, mstride);
      const uint32_t d0 = *reinterpret_cast<const uint32_t*>(sp + tbase + i * 80), d1 = *reinterpret_cast<const uint32_t*>(sp + tbase + i * 80 + 4),
                     d2 = *reinterpret_cast<const uint32_t*>(sp + tbase + i * 80 + 8), d3 = *reinterpret_cast<const uint32_t*>(sp + tbase + i * 80 + 12);
      const uint2 fhi = make_uint2(__builtin_amdgcn_perm(d1, d0, 0x07060302u), __builtin_amdgcn_perm(d3, d2, 0x07060302u));
      const uint2 flo = make_uint2(__builtin_amdgcn_perm(d1, d0, 0x05040100u), __builtin_amdgcn_perm(d3, d2, 0x05040100u));
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // (the conv.0 operands are re-read per use: 16 registers held across the epilogue would not fit beside the 144 of the weights)
        const u32x4_t wh = *reinterpret_cast<const u32x4_t*>(smem + WM_OFF + ((wn * 2 + j) * 64 + lane) * 16);
        const f32x4_t y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, am), __builtin_bit_cast(bf16x8_t, wh),
                                                                 f32x4_t{bj[j], bj[j], bj[j], bj[j]}, 0, 0, 0);
        float dy[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) dy[r] = y[r] > 0.f ? acc[i][j][r] : 0.f;
        const uint2 bd = make_uint2(pack_bf16(dy[0], dy[1]), pack_bf16(dy[2], dy[3]));
        // K = 32 = (4 pixels of the lane group) x (hi, lo): the masked gradient twice against the two halves of the frame values
        dw0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, u32x4_t{fhi.x, fhi.y, flo.x, flo.y}),
                                                         __builtin_bit_cast(bf16x8_t, u32x4_t{bd.x, bd.y, bd.x, bd.y}), dw0[j], 0, 0, 0);
      }
    }
    org[0] = org[1]; org[1] = org[2];
    walk.advance(org[2]);
  }
  // partial sums of the workgroup: the two waves that own a channel (tile rows 0 - 3 / 4 - 7) meet in LDS -> ws[workgroup][tap row 0 .. 9][64]
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; ++j) *reinterpret_cast<f32x4_t*>(red + (wave * 2 + j) * 256 + lane * 4) = dw0[j];
  __syncthreads();
  for (int o = tid; o < 640; o += 256) {
    const int tap = o >> 6, co = o & 63;
    const int wn2 = co >> 5, j = (co >> 4) & 1, ln = (tap >> 2) * 16 + (co & 15), r = tap & 3;
    p.ws[(int64_t)blockIdx.x * 640 + o] = red[((0 * 2 + wn2) * 2 + j) * 256 + ln * 4 + r] + red[((1 * 2 + wn2) * 2 + j) * 256 + ln * 4 + r];
  }
}

// dW0[co][tap] += / db0[co] += the workgroups' partial sums, fixed order (block = tap row 0 .. 9, thread = (channel, quarter of the workgroups))
__global__ __launch_bounds__(1024) void vgg_level0_dw0_reduce_kernel(const float* __restrict__ ws, int nwg, float* dw0, float* db0) {
  __shared__ float red[16][64];
  const int tap = blockIdx.x, co = threadIdx.x & 63, part = threadIdx.x >> 6;
  float s = 0.f;
  for (int w = part; w < nwg; w += 128) {          // 8 loads in flight per thread
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (w + u * 16 < nwg) ? ws[(int64_t)(w + u * 16) * 640 + tap * 64 + co] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  red[part][co] = s;
  __syncthreads();
  if (part != 0) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][co];
  if (tap < 9) dw0[co * 9 + tap] += s;
  else db0[co] += s;
}

// ================================================================================================ weight gradient of conv.2
// The block walk, operand reads and partial-block layout of conv3x3_wgrad_dma_kernel (conv_wgrad_dma.hip) with both stage halves
// produced in LDS: X = ReLU(conv.0(frames)) on the 10 x 18 halo patch (wave = channel fragment, all 12 pixel fragments), dY = the
// pooled gradient of the tile's 4 x 8 pooled pixels expanded through the selection codes (one (pixel, chunk) item per thread, its
// 16 + 8 bytes fetched into registers one patch ahead).
constexpr int L0_XB = 180 * 128, L0_DB = 128 * 128, L0_STAGE = L0_XB + L0_DB;

__device__ __forceinline__ bf16x8_t l0_read_tr(const unsigned char* lo, const unsigned char* hi) {
  const uint2 a = asr_lds_read_tr16(lo), b = asr_lds_read_tr16(hi);
  return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}

__global__ __launch_bounds__(256, 2) void vgg_level0_wgrad_kernel(L0Args p) {
  constexpr int S_OFF = 2 * L0_STAGE, B0_OFF = S_OFF + 2 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  int vid = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x, xcd = vid & 7, qn = nwg >> 3, rn = nwg & 7;
    vid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (vid >> 3);
  }
  const int npatch = p.ntiles;
  const int p_beg = vid * p.patches_per_wg, p_end = min(npatch, p_beg + p.patches_per_wg);
  const int np = p_end - p_beg;
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int H2 = p.H >> 1, W2 = p.W >> 1;

  // conv.0 operands of this wave's channel fragment (registers) and its bias
  const u32x4_t wop = l0_weight_operand(p.w0, wave * 16 + lr, g, p.wsplit != 0);
  if (tid < 64) reinterpret_cast<float*>(smem + B0_OFF)[tid] = p.b0 ? p.b0[tid] : 0.f;
  __syncthreads();
  const f32x4_t bias0 = *reinterpret_cast<const f32x4_t*>(smem + B0_OFF + (wave * 16 + 4 * g) * 4);

  // patch origins (this kernel walks CONSECUTIVE patches: plain carries)
  L0Org org[3];
  {
    int t = p_beg;
    org[0].w0 = (t % p.tiles_w) * 16; t /= p.tiles_w;
    org[0].h0 = (t % p.tiles_h) * 8; org[0].b = t / p.tiles_h;
  }
  auto next = [&](L0Org o) __attribute__((always_inline)) {
    o.w0 += 16;
    if (o.w0 >= p.tiles_w * 16) { o.w0 = 0; o.h0 += 8; if (o.h0 >= p.tiles_h * 8) { o.h0 = 0; ++o.b; } }
    return o;
  };
  org[1] = next(org[0]);
  org[2] = next(org[1]);

  // the thread's expansion item: pooled pixel (qr, qc) of the tile's 4 x 8, chunk c
  const int iq = tid >> 3, ic = tid & 7, iqr = iq >> 3, iqc = iq & 7;
  const int da0 = L0_XB + ((2 * iqr) * 16 + 2 * iqc) * 128 + ((ic ^ wgd_key(2 * iqc)) << 4);        // window (0, 0); (0, 1) = (da0 + 128) ^ 16; row 1: + 16 * 128
  u32x4_t pv = u32x4_t{0u, 0u, 0u, 0u};
  uint2 pk = make_uint2(0u, 0u);
  auto load_pooled = [&](const L0Org& o) __attribute__((always_inline)) {
    const int ph = (o.h0 >> 1) + iqr, pw = (o.w0 >> 1) + iqc;
    pv = u32x4_t{0u, 0u, 0u, 0u};
    pk = make_uint2(0u, 0u);
    if (ph < H2 && pw < W2) {
      const int64_t e = (((int64_t)o.b * H2 + ph) * W2 + pw) * 64 + ic * 8;
      pv = *reinterpret_cast<const u32x4_t*>(p.dpool + e);
      pk = *reinterpret_cast<const uint2*>(p.code + e);
    }
  };
  auto expand = [&](int st) __attribute__((always_inline)) {
    unsigned char* dp = smem + st * L0_STAGE;
    L0CodeMasks cm;
    cm.init(pk);
    u32x4_t o1, o2, o3, o4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      o1[d] = pv[d] & cm.mask<1>(d); o2[d] = pv[d] & cm.mask<2>(d);
      o3[d] = pv[d] & cm.mask<3>(d); o4[d] = pv[d] & cm.mask<4>(d);
    }
    const int a0 = da0, a1 = (da0 + 128) ^ 16;
    *reinterpret_cast<u32x4_t*>(dp + a0) = o1;
    *reinterpret_cast<u32x4_t*>(dp + a1) = o2;
    *reinterpret_cast<u32x4_t*>(dp + a0 + 16 * 128) = o3;
    *reinterpret_cast<u32x4_t*>(dp + a1 + 16 * 128) = o4;
  };
  // conv.0 for the 10 x 18 halo patch, this wave's 16 channels.  Pixel fragments by PATCH ROW (round 6): fragment f < 10 = columns 0 .. 15 of
  // row f, so its frame-patch and halo-patch addresses are one lane constant each plus a compile-time offset (instruction immediates);
  // fragments 10 / 11 = columns 16, 17 of rows 0 .. 7 / 8, 9 (lane lr <-> row lr / 2, column 16 + (lr & 1)); the 12 lanes of fragment 11
  // that have no pixel repeat fragment 10's (same value to the same place).  The linear numbering it replaces (pixel = 16 f + lr) cost two
  // divisions by 18 per fragment and lane: ~300 of the ~530 vector instructions of a wave and patch.
  const int gstride = l0_frame_lane_stride(g);
  const int swz = ((g >> 1) ^ (wave << 1)) << 4, sub8 = (g & 1) * 8;
  const int ga1 = lr * 4 + l0_frame_lane_off(g), wa1 = lr * 128 + (swz ^ (wgd_key(lr) << 4)) + sub8;
  const int pr2 = lr >> 1, pc2 = 16 + (lr & 1);
  const int ga2 = (pr2 * 20 + pc2) * 4 + l0_frame_lane_off(g), wa2 = (pr2 * 18 + pc2) * 128 + (swz ^ (wgd_key(pc2) << 4)) + sub8;
  const int ga3 = lr < 4 ? ga2 + 8 * 80 : ga2, wa3 = lr < 4 ? wa2 + 8 * 18 * 128 : wa2, pr3 = lr < 4 ? pr2 + 8 : pr2;
  auto generate = [&](int st, int sb, const L0Org& o) __attribute__((always_inline)) {
    unsigned char* xp = smem + st * L0_STAGE;
    const unsigned char* sp = smem + S_OFF + sb * 1024;
    const bool inside = o.h0 >= 1 && o.w0 >= 1 && o.h0 + 9 <= p.H && o.w0 + 17 <= p.W;
    auto frag_ga = [&](int f) __attribute__((always_inline)) { return f < 10 ? ga1 + f * 80 : (f == 10 ? ga2 : ga3); };
    auto frag_wa = [&](int f) __attribute__((always_inline)) { return f < 10 ? wa1 + f * (18 * 128) : (f == 10 ? wa2 : wa3); };
    // software pipeline over groups of 3 pixel fragments: the 6 MFMAs of group k issued, then group k - 1 converted / stored
    auto run = [&](auto masked) __attribute__((always_inline)) {
      uint32_t colm = 0xffffffffu, colm2 = 0xffffffffu;
      if constexpr (decltype(masked)::value) {       // halo pixels outside the image are conv.2's zero padding
        colm = (unsigned)(o.w0 + lr - 1) < (unsigned)p.W ? 0xffffffffu : 0u;
        colm2 = (unsigned)(o.w0 + pc2 - 1) < (unsigned)p.W ? 0xffffffffu : 0u;
      }
      f32x4_t acc3[2][3];
      auto finish = [&](int k) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const int f = k * 3 + u;
          if constexpr (decltype(masked)::value) {
            uint32_t vm;
            if (f < 10) vm = (unsigned)(o.h0 + f - 1) < (unsigned)p.H ? colm : 0u;
            else vm = (unsigned)(o.h0 + (f == 10 ? pr2 : pr3) - 1) < (unsigned)p.H ? colm2 : 0u;
            l0_gen_store(xp, frag_wa(f), acc3[k & 1][u], 0, vm);
          } else {
            l0_gen_store_inside(xp, frag_wa(f), acc3[k & 1][u], 0);
          }
        }
      };
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int u = 0; u < 3; ++u) acc3[k & 1][u] = l0_gen_mfma(l0_frame_operand(sp + frag_ga(k * 3 + u), gstride), wop, bias0);
        if (k > 0) finish(k - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      finish(3);
    };
    if (inside) run(std::false_type{});       // (wave-uniform)
    else run(std::true_type{});
  };

  // ---- per-lane operand offsets inside a stage (conv_wgrad_dma.hip)
  const int colb = 8 * (g & 1) + (lr >> 2), rowb = g >> 1, sub = 8 * (lr & 1), cpair = (lr & 3) >> 1;
  int xrun[3], dlo[4], dhi[4];          // (the stage's swizzle key: wgd_key, conv_wgrad_dma.h)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = colb + 4 * q, ch = wave * 2 + cpair;
    xrun[q] = (rowb * 18 + c) * 128 + ((ch ^ wgd_key(c)) << 4) + sub;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = 2 * i + cpair;
    dlo[i] = L0_XB + (rowb * 16 + colb) * 128 + ((ch ^ wgd_key(colb)) << 4) + sub;
    dhi[i] = L0_XB + (rowb * 16 + colb + 4) * 128 + ((ch ^ wgd_key(colb + 4)) << 4) + sub;
  }
  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // db2: every wave sums ONE channel fragment of the dY operands it reads anyway (fragment = wave; asr_sum8_bf16: one instruction per two
  // values).  Round 5 had wave 0 sum all four with shift / mask / add: 192 vector instructions on one wave of a
  // barrier-synchronised four.
  float bsum = 0.f;
  const bool do_bias = p.db != nullptr;

  // ---- prologue: frame patches 0 and 1, stage 0 complete
  if (np > 0) { l0_stage_frames(p, org[0], smem_base + S_OFF, tid, wave_u); load_pooled(org[0]); }
  if (np > 1) l0_stage_frames(p, org[1], smem_base + S_OFF + 1024, tid, wave_u);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  l0_split_frames(smem + S_OFF, tid);
  l0_split_frames(smem + S_OFF + 1024, tid);
  __syncthreads();
  if (np > 0) { generate(0, 0, org[0]); expand(0); }

  for (int j = 0; j < np; ++j) {
    const int buf = j & 1;
    __syncthreads();          // stage `buf` complete for every wave; everybody is done with the other stage; frame patch j + 1 split
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (j + 2 < np) l0_stage_frames(p, org[2], smem_base + S_OFF + (unsigned)(buf * 1024), tl, wave_u);
    if (j + 1 < np) {
      load_pooled(org[1]);
      generate(buf ^ 1, buf ^ 1, org[1]);
    }
    const unsigned char* sb = smem + buf * L0_STAGE;
    // 12 steps = (macro step ms: two patch rows of dY) x (kernel row ky); the X run of a step is read ONE STEP AHEAD of its MFMAs (round 6: read
    // and consumed inside one step, every step began with the LDS latency -- 12 exposed round trips per patch and wave)
    uint2 xr[2][3];
    auto xread = [&](uint2 (&r)[3], int st) __attribute__((always_inline)) {
      const int off = ((2 * (st / 3) + st % 3) * 18) * 128;
#pragma unroll
      for (int q = 0; q < 3; ++q) r[q] = asr_lds_read_tr16(sb + xrun[q] + off);
    };
    xread(xr[0], 0);
    bf16x8_t a[4];
#pragma unroll
    for (int st = 0; st < 12; ++st) {
      const int ms = st / 3, ky = st % 3;
      if (ky == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = l0_read_tr(sb + dlo[i] + ms * 4096, sb + dhi[i] + ms * 4096);
        if (do_bias) {
          switch (wave_u) {       // (wave-uniform: scalar branches)
            case 0: asr_sum8_bf16(bsum, a[0]); break;
            case 1: asr_sum8_bf16(bsum, a[1]); break;
            case 2: asr_sum8_bf16(bsum, a[2]); break;
            default: asr_sum8_bf16(bsum, a[3]); break;
          }
        }
      }
      if (st + 1 < 12) xread(xr[(st + 1) & 1], st + 1);
      // one run of 12 pixels per kernel row, the kx = 1, 2 operands by shifting (conv_wgrad_dma.hip)
      const uint2 r0 = xr[st & 1][0], r1 = xr[st & 1][1], r2 = xr[st & 1][2];
      const bf16x8_t b0 = __builtin_bit_cast(bf16x8_t, make_uint4(r0.x, r0.y, r1.x, r1.y));
      const bf16x8_t b1 = __builtin_bit_cast(bf16x8_t, make_uint4(__builtin_amdgcn_alignbit(r0.y, r0.x, 16), __builtin_amdgcn_alignbit(r1.x, r0.y, 16),
                                                                   __builtin_amdgcn_alignbit(r1.y, r1.x, 16), __builtin_amdgcn_alignbit(r2.x, r1.y, 16)));
      const bf16x8_t b2 = asr_shift2_of12(r0, r1, r2);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[3 * ky][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b0, acc[3 * ky][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[3 * ky + 1][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b1, acc[3 * ky + 1][i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[3 * ky + 2][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b2, acc[3 * ky + 2][i], 0, 0, 0);
      L0_FENCE();
    }
    if (j + 1 < np) expand(buf ^ 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (j + 2 < np) l0_split_frames(smem + S_OFF + buf * 1024, tl);
    org[0] = org[1]; org[1] = org[2]; org[2] = next(org[2]);
  }

  float* part = p.ws + (int64_t)vid * (9 * 64 * 64);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(t * 64 + i * 16 + g * 4 + r) * 64 + wave * 16 + lr] = acc[t][i][r];
  if (do_bias) {
    float v = bsum;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0) atomicAdd(p.db + wave * 16 + lr, v);
  }
}

int l0_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus = n;
  }
  return cus;
}
// One flag per KERNEL (the six kernels share the type void (*)(L0Args): a template over the type would share one flag between them,
// ADVICE r5); the first (eager / warm-up) launch does the grant, never a captured one.
template <void (*Kern)(L0Args)> int l0_grant(size_t lds) {
  static bool granted = false;
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      (void)hipGetLastError();          // a refused grant is reported by the return value, not left behind for the next launch check
      return ASR_ELAUNCH;
    }
    granted = true;
  }
  return ASR_OK;
}
bool l0_shape_ok(int B, int H, int W) {
  return B >= 0 && H >= 2 && W >= 2 && (int64_t)B * H * W * 128 < ((int64_t)1 << 32);
}
void l0_wgrad_grid(int B, int H, int W, int* wgx, int* ppw) {      // the grid conv.hip's reduce will fold (one 64 x 64 block)
  int blocks_y = 0;
  asr_conv3x3_wgrad_grid(B, H, W, 64, 64, wgx, &blocks_y, ppw);
}

}  // namespace

extern "C" int asr_vgg_level0_fwd(const float* src, const float* w0, const float* b0, const void* wk2, const float* b2, void* pool,
                                  uint8_t* code, int B, int H, int W, hipStream_t s) {
  ASR_CHECK_ARG(src && w0 && b0 && wk2 && b2 && pool && code);
  if (!l0_shape_ok(B, H, W) || !aligned16(wk2) || !aligned16(pool) || (((uintptr_t)code) & 7) != 0) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  L0Args a{};
  a.src = src; a.w0 = w0; a.b0 = b0; a.wk = static_cast<const bf16_t*>(wk2); a.b2 = b2;
  a.pool = static_cast<bf16_t*>(pool); a.code = code; a.B = B; a.H = H; a.W = W;
  a.tiles_h = (H + 7) / 8; a.tiles_w = (W + 15) / 16; a.ntiles = B * a.tiles_h * a.tiles_w;
  const size_t lds = 2 * L0_PB + 2 * L0_SS + L0_WM + 512;
  a.wsplit = asr_tuning("L0_WSPLIT", 1) != 0 ? 1 : 0;
  {
    // The caller decides HERE whether the level runs on these kernels (EUNSUPPORTED -> the stored-activation launch chain): the two
    // backward kernels need more LDS than this one, so their grants are part of the decision -- a backward pass cannot fall back.
    const size_t lds_d = 2 * L0_PB + 2 * L0_PST + 2 * L0_SS + L0_WM + 256 + 16, lds_w = 2 * L0_STAGE + 2 * 1024 + 256;
    const int r0 = l0_grant<vgg_level0_fwd_kernel>(lds), r1 = l0_grant<vgg_level0_dgrad_kernel>(lds_d), r2 = l0_grant<vgg_level0_wgrad_kernel>(lds_w);
    if (r0 != ASR_OK || r1 != ASR_OK || r2 != ASR_OK) return ASR_EUNSUPPORTED;
  }
  const int64_t slots = (int64_t)l0_cus() * 2;
  const unsigned grid = (unsigned)(a.ntiles < slots ? a.ntiles : slots);
#ifdef L0_TIMING
  static long long* dbg = nullptr;
  if (!dbg) (void)hipMalloc(&dbg, 64 * 8);
  (void)hipMemsetAsync(dbg, 0, 64 * 8, s);
  a.dbg = dbg;
#endif
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  hipLaunchKernelGGL(vgg_level0_fwd_kernel, dim3(grid), dim3(256), lds, s, a);
  ASR_LAUNCH_CHECK();
#ifdef L0_TIMING
  {
    long long h[64];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    static int shown = 0;
    if (shown++ == 8)       // a warm launch
      for (int w = 0; w < 4; ++w)
        fprintf(stderr, "level0 fwd timing wave %d tiles %lld: barrier %lld generation %lld main %lld dmawait %lld epilogue %lld looptop %lld (s_memtime ticks)\n",
                w, h[w * 8 + 7], h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5]);
  }
#endif
  return ASR_OK;
}

extern "C" int64_t asr_vgg_level0_bwd_workspace(int B, int H, int W) {
  if (!l0_shape_ok(B, H, W)) return 0;
  int wgx, ppw;
  l0_wgrad_grid(B, H, W, &wgx, &ppw);
  const int64_t a = (int64_t)wgx * 9 * 64 * 64, b = (int64_t)l0_cus() * 2 * 640;
  return a > b ? a : b;
}

extern "C" int asr_vgg_level0_dgrad(const void* dpool, const uint8_t* code, const float* src, const float* w0, const float* b0,
                                    const void* wd2, float* dw0, float* db0, float* workspace, int64_t workspace_floats, int B, int H,
                                    int W, hipStream_t s) {
  ASR_CHECK_ARG(dpool && code && src && w0 && b0 && wd2 && dw0 && db0 && workspace);
  if (!l0_shape_ok(B, H, W) || !aligned16(wd2) || !aligned16(dpool) || !aligned16(code) || !aligned16(workspace)) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  L0Args a{};
  a.src = src; a.w0 = w0; a.b0 = b0; a.wk = static_cast<const bf16_t*>(wd2);
  a.dpool = static_cast<const bf16_t*>(dpool); a.code = const_cast<uint8_t*>(code); a.ws = workspace; a.B = B; a.H = H; a.W = W;
  a.tiles_h = (H + 7) / 8; a.tiles_w = (W + 15) / 16; a.ntiles = B * a.tiles_h * a.tiles_w;
  const size_t lds = 2 * L0_PB + 2 * L0_PST + 2 * L0_SS + L0_WM + 256 + 16;
  a.wsplit = asr_tuning("L0_WSPLIT", 1) != 0 ? 1 : 0;
  const int rc = l0_grant<vgg_level0_dgrad_kernel>(lds);
  if (rc != ASR_OK) return rc;
  const int64_t slots = (int64_t)l0_cus() * 2;
  const unsigned grid = (unsigned)(a.ntiles < slots ? a.ntiles : slots);
  if (workspace_floats < (int64_t)grid * 640) return ASR_EINVAL;
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  hipLaunchKernelGGL(vgg_level0_dgrad_kernel, dim3(grid), dim3(256), lds, s, a);
  ASR_LAUNCH_CHECK();
  hipLaunchKernelGGL(vgg_level0_dw0_reduce_kernel, dim3(10), dim3(1024), 0, s, workspace, (int)grid, dw0, db0);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_vgg_level0_wgrad(const float* src, const float* w0, const float* b0, const void* dpool, const uint8_t* code, float* dw2,
                                    float* db2, float* workspace, int64_t workspace_floats, int B, int H, int W, hipStream_t s) {
  ASR_CHECK_ARG(src && w0 && b0 && dpool && code && dw2 && workspace);
  if (!l0_shape_ok(B, H, W) || !aligned16(dpool) || (((uintptr_t)code) & 7) != 0 || !aligned16(workspace)) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  L0Args a{};
  a.src = src; a.w0 = w0; a.b0 = b0; a.dpool = static_cast<const bf16_t*>(dpool); a.code = const_cast<uint8_t*>(code);
  a.ws = workspace; a.db = db2; a.B = B; a.H = H; a.W = W;
  a.tiles_h = (H + 7) / 8; a.tiles_w = (W + 15) / 16; a.ntiles = B * a.tiles_h * a.tiles_w;
  int wgx;
  l0_wgrad_grid(B, H, W, &wgx, &a.patches_per_wg);
  if (workspace_floats < (int64_t)wgx * 9 * 64 * 64) return ASR_EINVAL;
  const size_t lds = 2 * L0_STAGE + 2 * 1024 + 256;
  a.wsplit = asr_tuning("L0_WSPLIT", 1) != 0 ? 1 : 0;
  const int rc = l0_grant<vgg_level0_wgrad_kernel>(lds);
  if (rc != ASR_OK) return rc;
  {
    AsrProfScope prof(ASR_OP_CONV_WGRAD, s);
    hipLaunchKernelGGL(vgg_level0_wgrad_kernel, dim3((unsigned)wgx), dim3(256), lds, s, a);
    ASR_LAUNCH_CHECK();
  }
  return asr_conv3x3_wgrad_reduce(workspace, dw2, B, H, W, 64, 64, s);
}
