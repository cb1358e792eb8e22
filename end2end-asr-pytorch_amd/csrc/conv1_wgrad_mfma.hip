// Weight / bias gradient of vgg_cnn's first layer (1 -> 64 channels, 3x3; reference: autograd of models/asr/transformer.py:43-44)
// on the matrix cores, bf16 storage mode.  dW[co][tap] = sum_px dY[px][co] * x[px + tap] is a GEMM whose contraction index is the
// PIXEL: M = 64 channels, N = taps, K = B*H*W.  The vector-ALU kernel of conv.hip spends 9 FMA per (pixel, channel) and runs at
// 2.1 TB/s of dY; here:
//   * dY tiles (128 consecutive pixels of one image row = 16 KB) travel HBM -> LDS by hand-issued LDS-DMA, four stages, 16-byte
//     chunk c of pixel p in slot c ^ (p & 7); the A operands (8 consecutive pixels per lane) are built by ds_read_b64_tr_b16;
//   * the fp32 input keeps (almost) its precision: x = xh + xl, two bf16 -- the B operand has the 9 taps of xh in columns 0..8 of
//     one fragment and the 9 taps of xl in columns 0..8 of a second one; column 9 of the first is the constant 1, so the bias
//     gradient sum_px dY[px][co] comes out of the same MFMAs;
//   * a wave contracts one 32-pixel macro step of every tile against all 64 channels: 8 MFMAs per 32 pixels -- the kernel is bound
//     by reading dY;
//   * per-wave partial sums meet in LDS, then one atomic per (workgroup, element).
#include "common.h"
#include "conv1_wgrad_mfma.h"

#include <stdlib.h>

namespace {

__device__ const uint4 c1w_zero_page = {0u, 0u, 0u, 0u};

typedef __attribute__((ext_vector_type(2))) float c1w_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 c1w_bf16x2_t;
__device__ __forceinline__ uint32_t c1w_pack(float a, float b) {      // one v_cvt_pk_bf16_f32: a -> low half
  const c1w_f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, c1w_bf16x2_t));
}

// hand-issued LDS-DMA piece (see conv_wgrad_dma.hip: the compiler must not see a DMA in flight, and does not count it)
__device__ __forceinline__ void c1w_dma(unsigned lds_wave_base, const unsigned char* src) {
  unsigned keep;      // M0 is saved and restored: the statement is neutral for whatever the compiler keeps there
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}

// 4 bytes per lane (the fp32 input rows: only dword alignment is guaranteed)
__device__ __forceinline__ void c1w_dma4(unsigned lds_wave_base, const unsigned char* src) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}

constexpr int TILE = 128;              // pixels per tile (one image row segment)
constexpr int DYB = TILE * 128;        // dY tile bytes
constexpr int XW = TILE + 8;           // staged input row: columns x0 - 4 .. x0 + TILE + 3 (16-byte aligned start)
constexpr int XB = 2048;               // three input rows, fp32 (3 x 136 x 4 = 1632 B) padded to the 512 DMA lanes
constexpr int STAGE = DYB + XB;
constexpr int C1W_STAGES = 4;          // LDS stages (3 tiles ahead)

__global__ __launch_bounds__(256) void conv1_wgrad_mfma_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                               float* dw, float* db, int B, int H, int W, int tiles_w, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wave_lds = smem_base + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
  const unsigned char* DY = reinterpret_cast<const unsigned char*>(dy);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(&c1w_zero_page);

  // tile -> (image row, first column)
  auto origin = [&](int tile, int64_t& rowpix, int& yy, int64_t& img, int& x0) __attribute__((always_inline)) {
    const int row = tile / tiles_w;
    x0 = (tile - row * tiles_w) * TILE;
    yy = row % H;
    img = row / H;
    rowpix = (int64_t)row * W;
  };
  // dY tile -> stage `buf` by DMA (pixels past the row end come from the zero page); this thread's 2 input values -> registers
  auto stage_dy = [&](int tile, int buf) __attribute__((always_inline)) {
    int64_t rowpix, img; int yy, x0;
    origin(tile, rowpix, yy, img, x0);
#pragma unroll
    for (int i = 0; i < DYB / 4096; ++i) {
      const int c = tid + i * 256, px = c >> 3, ch = (c & 7) ^ (px & 7);
      const unsigned char* src = x0 + px < W ? DY + ((rowpix + x0 + px) * 128 + ch * 16) : zero;
      c1w_dma(wave_lds + (unsigned)(buf * STAGE + i * 4096), src);
    }
  };
  // the three input rows of the tile (columns x0 - 4 .. x0 + TILE + 3, fp32) by 4-byte DMA pieces: element e of the 3 x XW block sits
  // at LDS offset 4 e; rows / columns outside the image come from the zero page
  const unsigned wave_lds4 = smem_base + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 256u;
  auto stage_x = [&](int tile, int buf) __attribute__((always_inline)) {
    int64_t rowpix, img; int yy, x0;
    origin(tile, rowpix, yy, img, x0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int e = tid + i * 256;                 // (e >= 3 XW: lands in the unused tail of the block, XB is padded to 2 KB)
      const int r = e / XW, c = e - r * XW;
      const int ry = yy + r - 1, cx = x0 - 4 + c;
      const bool ok = e < 3 * XW && ry >= 0 && ry < H && cx >= 0 && cx < W;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(x + ((img * H + ry) * (int64_t)W + cx)) : zero;
      c1w_dma4(wave_lds4 + (unsigned)(buf * STAGE + DYB + i * 1024), src);
    }
  };

  // accumulators: D[co = 16 i + 4 g + r][n = lr], n-fragment 0 = {9 taps of xh, ones}, 1 = {9 taps of xl}
  f32x4_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // this lane's tap (column lr of the B operand): input offset of tap (dy, dx) relative to the pixel, in the staged block
  const int tap_dy = lr / 3, tap_dx = lr - tap_dy * 3;
  const int tap_off = lr < 9 ? tap_dy * XW + tap_dx + 3 : 0;      // block column of pixel p is p + 4; tap column = p + dx - 1 + 4
  const float is_tap = lr < 9 ? 1.f : 0.f, is_one = lr == 9 ? 1.f : 0.f;
  // A-operand addressing inside a stage: pixels ms*32 + 8 g + j, channels 16 i + lr
  const int prow = 8 * g + (lr >> 2), sub = 8 * (lr & 1), cpair = (lr & 3) >> 1;

  // NS stages per workgroup, everything staged by hand-issued DMA (no compiler-visible loads in the loop): reading at HBM speed
  // needs ~80 KB in flight per CU (21 KB/us per CU x 3-4 us of latency) = 2 workgroups x 3 tiles of 18 KB ahead.  Every wave issues
  // exactly 6 pieces per tile and loads retire in order, so "at most 6 k outstanding" means the k newest tiles may still be in flight.
  constexpr int NS = C1W_STAGES;
  const int step = gridDim.x;
  int tile = blockIdx.x;
#pragma unroll
  for (int d = 0; d < NS - 1; ++d)
    if (tile + d * step < ntiles) { stage_dy(tile + d * step, d); stage_x(tile + d * step, d); }
  for (int n = 0; tile < ntiles; tile += step, ++n) {
    const int buf = n % NS;
    const int ahead = min(NS - 2, (ntiles - 1 - tile) / step);        // tiles after this one that are already in flight
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                      // the tile is complete for every wave; everybody is done with tile n-1's stage
    const int next = tile + (NS - 1) * step;
    if (next < ntiles) { stage_dy(next, (n + NS - 1) % NS); stage_x(next, (n + NS - 1) % NS); }
    const unsigned char* sd = smem + buf * STAGE;
    const float* sx = reinterpret_cast<const float*>(sd + DYB);
    // ---- this wave's macro step: pixels wave*32 .. +31 of the tile
    const int p0 = wave * 32 + prow;
    bf16x8_t a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ch = 2 * i + cpair;
      const uint2 lo = asr_lds_read_tr16(sd + p0 * 128 + ((ch ^ (p0 & 7)) << 4) + sub);
      const uint2 hi = asr_lds_read_tr16(sd + (p0 + 4) * 128 + ((ch ^ ((p0 + 4) & 7)) << 4) + sub);
      a[i] = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
    // B operand: k = 8 g + e  <->  pixel wave*32 + 8 g + e; value = x at this lane's tap of that pixel
    float xv[8];
    const float* xp = sx + tap_off + wave * 32 + 8 * g;
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[e] = xp[e];
    uint32_t bh[4], bl[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v0 = xv[2 * e] * is_tap, v1 = xv[2 * e + 1] * is_tap;
      const float h0 = __uint_as_float(__float_as_uint(v0) & 0xffff0000u), h1 = __uint_as_float(__float_as_uint(v1) & 0xffff0000u);
      bh[e] = c1w_pack(h0 + is_one, h1 + is_one);
      bl[e] = c1w_pack(v0 - h0, v1 - h1);
    }
    const bf16x8_t bhv = __builtin_bit_cast(bf16x8_t, make_uint4(bh[0], bh[1], bh[2], bh[3]));
    const bf16x8_t blv = __builtin_bit_cast(bf16x8_t, make_uint4(bl[0], bl[1], bl[2], bl[3]));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], bhv, acc[i][0], 0, 0, 0);
      acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], blv, acc[i][1], 0, 0, 0);
    }
  }

  // ---- fold the 4 waves in LDS, then one atomic per (workgroup, element): dw[co][tap] (n < 9), db[co] (n == 9)
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);            // [4 waves][64 co][16 n]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 64 + i * 16 + 4 * g + r) * 16 + lr] = acc[i][0][r] + acc[i][1][r];
  __syncthreads();
  for (int e = tid; e < 64 * 16; e += 256) {
    const int co = e >> 4, nn = e & 15;
    if (nn > 9) continue;
    const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
    if (nn < 9) atomicAdd(dw + co * 9 + nn, v);
    else if (db) atomicAdd(db + co, v);
  }
}

}  // namespace

int asr_conv1_wgrad_mfma_launch(const float* x, const bf16_t* dy, float* dw, float* db, int B, int H, int W, hipStream_t s) {
  const int tiles_w = (W + TILE - 1) / TILE;
  const int64_t nt = (int64_t)B * H * tiles_w;
  if (nt >= ((int64_t)1 << 31)) return ASR_EUNSUPPORTED;
  const size_t lds = (size_t)C1W_STAGES * STAGE;
  static bool granted = false;          // the first (eager / warm-up) launch does it, never a captured one
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv1_wgrad_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return ASR_ELAUNCH;
    granted = true;
  }
  const int per_cu = (int)asr_tuning("CONV1_WGRAD_WGS", 2);
  const unsigned grid = (unsigned)(nt < 256 * per_cu ? nt : 256 * per_cu);     // every workgroup ends with 640 atomics
  hipLaunchKernelGGL(conv1_wgrad_mfma_kernel, dim3(grid), dim3(256), lds, s, x, dy, dw, db, B, H, W, tiles_w, (int)nt);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
