// 3x3 convolution weight gradient, bf16 NHWC, LDS-DMA pipelined variant of conv3x3_wgrad_nhwc_kernel (conv.hip; reference:
// the autograd of models/asr/transformer.py:44-52).  Same decomposition -- a workgroup owns a 64 co x 64 ci x 9 tap block of dW
// and walks 8 x 16 pixel patches, wave w holds the 9 x 4 accumulator fragments of ci slice w -- but the operands no longer pass
// through registers on their way to LDS and the vector ALU no longer does per-patch address arithmetic:
//   * halo patch of X (10 x 18 px x 64 ci) and dY tile (8 x 16 px x 64 co) of patch n+1 travel HBM -> LDS by the LDS-DMA (two
//     stages) while patch n is contracted; per-thread source offsets are computed once, a patch costs one scalar base update
//     (patches on the image border take a slower path: per-chunk bounds test, outside pixels from a 16-byte zero page);
//   * LDS image = unpadded 128-byte pixel rows, 16-B chunk c of the pixel in patch column x in slot c ^ wgd_key(x) (conv_wgrad_dma.h); the MFMA operands
//     (8 CONSECUTIVE PIXELS per lane) are built by ds_read_b64_tr_b16 (see common.h), issued by hand two k steps ahead -- the
//     compiler would drain the DMA counter before every LDS read it can see -- with the (macro step, tap) part of the address in
//     the instruction's immediate offset;
//   * two 4-wave workgroups per CU, out of phase: one's DMA issue / epilogue overlaps the other's MFMAs.
// Per-workgroup partial dW blocks go to the caller's workspace and are folded by wgrad_reduce_kernel (conv.hip).
#include "common.h"
#include "conv_wgrad_dma.h"

#include <utility>

namespace {

__device__ const uint4 wgd_zero_page = {0u, 0u, 0u, 0u};

#define WGD_FENCE() asm volatile("" ::: "memory")

constexpr int XB = 180 * 128;          // halo patch bytes
constexpr int DB = 128 * 128;          // dY tile bytes
constexpr int STAGE = XB + DB;

// One 16-byte-per-lane LDS-DMA piece, issued by hand: the compiler must not know that a DMA is in flight, or it drains the
// VMEM counter before every LDS read of the patch being contracted.  (It then also cannot count these loads: the kernel has no
// compiler-visible vector memory loads while a DMA is outstanding, and the patch loop waits for vmcnt(0) by hand.)
__device__ __forceinline__ void wgd_dma(unsigned lds_wave_base, const unsigned char* base, unsigned off) {
  unsigned keep;      // M0 is saved and restored: the statement is neutral for whatever the compiler keeps there
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(off), "s"(base)
               : "memory");
}
__device__ __forceinline__ void wgd_dma(unsigned lds_wave_base, const unsigned char* src) {
  unsigned keep;      // M0 is saved and restored: the statement is neutral for whatever the compiler keeps there
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}

// transposed 8-pixel operand: two ds_read_b64_tr_b16 (pixels 0..3 and 4..7 of the lane group's 8)
__device__ __forceinline__ bf16x8_t wgd_read(const unsigned char* lo, const unsigned char* hi) {
  const uint2 a = asr_lds_read_tr16(lo), b = asr_lds_read_tr16(hi);
  return __builtin_bit_cast(bf16x8_t, make_uint4(a.x, a.y, b.x, b.y));
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_dma_kernel(WgdArgs p) {
  constexpr int NXC = 180 * 8, NDC = 128 * 8;          // 16-byte chunks of the X patch / dY tile
  constexpr int RX = (NXC + 255) / 256, RD = NDC / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  // The (co block, ci block) workgroups of one patch range read the same X patches / dY tiles: they are made NEIGHBOURS in an id that
  // is dealt so that consecutive ids run on one XCD (blockIdx round-robins over the 8 XCDs, each with its own L2) -- the shared slice
  // comes from HBM once per XCD instead of once per sibling (a 2-D grid put the siblings wgx ids apart: on whatever XCDs).
  int vid = (int)blockIdx.x;
  if (p.xcd_order) {
    const int nwg = (int)gridDim.x, xcd = vid & 7, qn = nwg >> 3, rn = nwg & 7;
    vid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (vid >> 3);
  }
  // (wave-uniform by construction; readfirstlane tells the compiler so: the DMA below takes its base pointers in scalar registers)
  const int by = __builtin_amdgcn_readfirstlane(p.xcd_order ? vid % p.blocks_y : vid / p.wgx);
  const int bx = __builtin_amdgcn_readfirstlane(p.xcd_order ? vid / p.blocks_y : vid % p.wgx);
  const int co0 = (by / p.nci) * 64, ci0 = (by % p.nci) * 64;
  const unsigned char* X = reinterpret_cast<const unsigned char*>(p.x) + ci0 * 2;
  const unsigned char* DY = reinterpret_cast<const unsigned char*>(p.dy) + co0 * 2;
  const int p_beg = bx * p.patches_per_wg, p_end = min(p.npatch, p_beg + p.patches_per_wg);
  const int xrow = p.Cin * 2, drow = p.Cout * 2;       // bytes per pixel

  // ---- per-thread DMA source offsets relative to the patch's first pixel
  int relx[RX], reld[RD];
#pragma unroll
  for (int i = 0; i < RX; ++i) {
    const int c = tid + i * 256, hp = c >> 3, col = hp % 18;
    relx[i] = ((hp / 18 - 1) * p.W + col - 1) * xrow + (((c & 7) ^ wgd_key(col)) << 4);
  }
#pragma unroll
  for (int i = 0; i < RD; ++i) {
    const int c = tid + i * 256, px = c >> 3, col = px & 15;
    reld[i] = ((px >> 4) * p.W + col) * drow + (((c & 7) ^ wgd_key(col)) << 4);
  }
  // patch origin, advanced without divisions
  int b, h0, w0;
  {
    int t = p_beg;
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    w0 = tw * 16; h0 = (t % p.tiles_h) * 8; b = t / p.tiles_h;
  }
  auto advance = [&]() __attribute__((always_inline)) {
    w0 += 16;
    if (w0 >= p.tiles_w * 16) { w0 = 0; h0 += 8; if (h0 >= p.tiles_h * 8) { h0 = 0; ++b; } }
  };
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wave_lds = smem_base + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;      // + piece * 4096 + stage
  auto stage = [&](int buf, int t_) __attribute__((always_inline)) {
    const unsigned sx = wave_lds + (unsigned)(buf * STAGE);
    const unsigned pix = ((unsigned)b * (unsigned)p.H + (unsigned)h0) * (unsigned)p.W + (unsigned)w0;
    const unsigned bx = pix * (unsigned)xrow, bd = pix * (unsigned)drow;            // < 4 GB (launcher)
    const bool inside = h0 >= 1 && w0 >= 1 && h0 + 9 <= p.H && w0 + 17 <= p.W;
    if (inside) {
#pragma unroll
      for (int i = 0; i < RX; ++i)
        if (i < RX - 1 || tid + i * 256 < NXC) wgd_dma(sx + i * 4096, X, bx + (unsigned)relx[i]);
#pragma unroll
      for (int i = 0; i < RD; ++i) wgd_dma(sx + XB + i * 4096, DY, bd + (unsigned)reld[i]);
    } else {
      // (everything from the laundered thread index `t_`: this arithmetic must stay inside the patch loop, not in 20 hoisted registers)
      const unsigned char* zero = reinterpret_cast<const unsigned char*>(&wgd_zero_page);
#pragma unroll
      for (int i = 0; i < RX; ++i) {
        const int c = t_ + i * 256;
        if (i < RX - 1 || c < NXC) {
          const int hp = c >> 3, col = hp % 18, gy = h0 + hp / 18 - 1, gx = w0 + col - 1;
          const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          const unsigned off = (((unsigned)b * (unsigned)p.H + (unsigned)gy) * (unsigned)p.W + (unsigned)gx) * (unsigned)xrow +
                               (unsigned)(((c & 7) ^ wgd_key(col)) << 4);
          wgd_dma(sx + i * 4096, in ? X + off : zero);
        }
      }
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        const int c = t_ + i * 256, px = c >> 3, col = px & 15, gy = h0 + (px >> 4), gx = w0 + col;
        const bool in = gy < p.H && gx < p.W;
        const unsigned off = (((unsigned)b * (unsigned)p.H + (unsigned)gy) * (unsigned)p.W + (unsigned)gx) * (unsigned)drow +
                             (unsigned)(((c & 7) ^ wgd_key(col)) << 4);
        wgd_dma(sx + XB + i * 4096, in ? DY + off : zero);
      }
    }
  };

  // ---- per-lane operand offsets inside a stage (k = 8 g + j <-> patch row 2 ms + (g >> 1), column 8 (g & 1) + j); the
  // (macro step, tap) part of an address is a compile-time constant that lands in the read's immediate offset
  const int colb = 8 * (g & 1) + (lr >> 2), rowb = g >> 1, sub = 8 * (lr & 1), cpair = (lr & 3) >> 1;
  int xrun[3], dlo[4], dhi[4];          // xrun: pixels colb .. + 3, + 4 .. + 7, + 8 .. + 11 of the lane's patch row (the kx = 1, 2 operands are shifts of the run)
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int c = colb + 4 * q, ch = wave * 2 + cpair;
    xrun[q] = (rowb * 18 + c) * 128 + ((ch ^ wgd_key(c)) << 4) + sub;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = 2 * i + cpair;
    dlo[i] = XB + (rowb * 16 + colb) * 128 + ((ch ^ wgd_key(colb)) << 4) + sub;
    dhi[i] = XB + (rowb * 16 + colb + 4) * 128 + ((ch ^ wgd_key(colb + 4)) << 4) + sub;
  }

  f32x4_t acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // db: sum of dY over the pixels, in the ci block 0 workgroups.  Every wave sums ONE channel fragment of the dY operands it reads anyway
  // (fragment = wave, one v_dot2c per two values); until round 6 wave 0 summed all four with shift / mask / add -- 192 vector
  // instructions per patch on one wave of a barrier-synchronised four.
  float bsum = 0.f;
  const bool do_bias = p.db != nullptr && ci0 == 0;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);

  if (p_beg < p_end) stage(0, tid);
  for (int patch = p_beg; patch < p_end; ++patch) {
    const int buf = (patch - p_beg) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the patch have landed (the compiler does not count them)
    __syncthreads();          // the patch is complete for every wave; everybody is done with the other stage
    int tl = tid;
    asm volatile("" : "+v"(tl));
    advance();
    if (patch + 1 < p_end) stage(buf ^ 1, tl);
    const unsigned char* sb = smem + buf * STAGE;
    // 12 steps = (macro step ms: two patch rows of dY) x (kernel row ky).  The three taps of a kernel row read the same patch row at columns
    // c .. c + 7, c + 1 .. c + 8, c + 2 .. c + 9: ONE run of 12 pixels (three transposing reads instead of six) and the kx = 1, 2 operands by
    // shifting -- kx = 2 two v_pk_mov, kx = 1 four v_alignbit.  (Pixels 10, 11 of the run are not used: for the right half they lie in the
    // next patch row / behind the patch, inside the stage.)  The run of a step is read ONE STEP AHEAD of its MFMAs (round 6: read and
    // consumed inside one step, every step began with the LDS latency -- 12 exposed round trips per patch and wave).
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
        for (int i = 0; i < 4; ++i) a[i] = wgd_read(sb + dlo[i] + ms * 4096, sb + dhi[i] + ms * 4096);
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
      WGD_FENCE();      // bounds how far ahead the scheduler hoists operand reads (and their registers): one step
    }
  }

  // ---- partial dW block -> workspace [block of dW][patch range][tap][co][ci]
  float* part = p.ws + ((int64_t)by * p.wgx + bx) * (9 * 64 * 64);
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
    if (g == 0) atomicAdd(p.db + co0 + wave * 16 + lr, v);
  }
}

}  // namespace

int asr_conv3x3_wgrad_dma_launch(const WgdArgs& p, unsigned wgx, unsigned blocks_y, hipStream_t s) {
  const size_t lds = 2 * (size_t)STAGE;
  static bool granted = false;          // the first (eager / warm-up) launch does it, never a captured one
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wgrad_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return ASR_ELAUNCH;
    granted = true;
  }
  WgdArgs q = p;
  q.wgx = (int)wgx; q.blocks_y = (int)blocks_y; q.xcd_order = asr_tuning("WGRAD_XCD", 1) != 0;
  hipLaunchKernelGGL(conv3x3_wgrad_dma_kernel, dim3(wgx * blocks_y), dim3(256), lds, s, q);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
