// 3x3 convolution, 64 -> 64 channels, bf16 NHWC: the full-resolution layer of vgg_cnn (reference: models/asr/transformer.py:44-47,
// conv2 forward and its dgrad), the largest single kernel of the training step.  A persistent, software-pipelined variant of the
// implicit GEMM of conv.hip, possible because ALL of the layer's weights (64 co x 9 taps x 64 ci = 72 KB) fit in the register
// files of a workgroup:
//   * persistent workgroups (default: TWO 4-wave workgroups per CU, which run out of phase so that one's address / epilogue work
//     overlaps the other's MFMAs) walk tiles of 128 output pixels (8 x 16) x 64 co; wave (wm, wn) owns 4 pixel fragments x 2 co
//     fragments and keeps its 32 co x 576 k weights in 144 VGPRs for the whole kernel: no weight traffic, no per-tap barrier
//     (8-wave workgroups on 16 x 16 / 8 x 32 tiles remain as ASR_C64_SHAPE=1/2);
//   * the halo patch of tile n+2 travels HBM -> LDS by the LDS-DMA into one of THREE patch buffers (two in the masked variant, where
//     the mask stash needs the LDS) while tile n is contracted: one s_barrier per tile, the wait on the DMA counter sits after the
//     MFMAs of a whole tile and is COUNTED (loads retire in order: "at most N outstanding" proves everything older has landed);
//   * the vector ALU does almost nothing per tile (an MFMA leaves room for about two other vector instructions): operand
//     addresses = 6 per-lane registers + instruction immediates (swizzle keyed on the patch COLUMN), per-thread DMA offsets computed
//     once + a scalar tile base, tile origins advanced with carries, branch-free epilogue;
//   * pixels outside the image are DMA'd from a 16-byte zero page (no zero-fill pass);
//   * operand reads are hand-issued ds_read_b128 (inline asm, double buffered per k step) -- the compiler would otherwise drain
//     the DMA counter before every LDS read it can see;
//   * epilogue from the accumulators (co rows x pixel columns): bias / ReLU / bf16 / v_permlane16_swap -> one 16-byte chunk of a
//     pixel's NHWC row per lane; the dgrad's ReLU-mask chunks are prefetched per lane through a private LDS stash by the same DMA.
// MFMA ~ HBM bound: 2*9*64*64 flop per output pixel; HBM bytes per pixel = 2 * 64 * 2 (+ halo overlap on the read side).
// Measured (MI355X, B=32 161x800): 270 us forward / 310 us dgrad+mask; the MFMAs alone 144 us, the memory side alone 219 us.
#include "common.h"
#include "conv_c64.h"
#include "conv_c64_core.h"

#include <stdio.h>
#include <stdlib.h>

#include <utility>

namespace {

#define C64_COMPILER_FENCE() asm volatile("" ::: "memory")
// ablation hooks (ASR_C64_ABLATE: 1 = no patch DMA, 2 = no operand reads / MFMAs, 4 = no stores) exist in -DASR_TUNE_ABLATE builds only
#ifdef ASR_TUNE_ABLATE
#define C64_ABL(P, BIT) (((P).ablate & (BIT)) != 0)
#else
#define C64_ABL(P, BIT) false
#endif
#ifdef C64_TIMING      // tuning builds only: s_memtime stamps at section boundaries (each one drains lgkmcnt)
#define C64_STAMP(K) { const long long now_ = clock64(); tsec[K] += now_ - tlast; tlast = now_; }
#else
#define C64_STAMP(K)
#endif

template <int TW, int TH, bool MASK, int NBUF, bool POOL = false>
__global__ __launch_bounds__(TW * TH * 2, 2) void conv3x3_c64_kernel(C64Args p) {
  constexpr int CB = TW / 16;                         // 16-pixel column blocks per tile row
  constexpr int WM = TW * TH / 64;                    // wave rows (4 pixel fragments each); waves = WM x 2 (32 co each)
  constexpr int NT = WM * 128;                        // threads
  constexpr int PW = TW + 2, NHALO = (TH + 2) * PW;   // halo patch
  constexpr int PBYTES = NHALO * 128;                 // 64 bf16 channels per pixel, 16-B chunk c of the pixel in patch column x in slot c ^ (x & 7)
  constexpr int NCH = NHALO * 8;                      // 16-byte chunks of a patch
  constexpr int PIT = (NCH + NT - 1) / NT;            // DMA instructions per thread and patch (the last one partial)
  constexpr int DIST = NBUF - 1;                      // tiles the patch DMA runs ahead
  constexpr int STASH = MASK ? WM * 2 * 4096 : 0;     // MASK: 4 chunks per lane
  constexpr int BIAS_OFF = NBUF * PBYTES + STASH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* stash = smem + NBUF * PBYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware walk: the workgroups of one XCD (blockIdx % 8) take consecutive tiles, so most halo rows are shared in ONE L2
  const int nwg = gridDim.x;
  const int vid = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;
  const int cnt = vid < p.ntiles ? (p.ntiles - vid + nwg - 1) / nwg : 0;
  const unsigned char* X = reinterpret_cast<const unsigned char*>(p.x);

  // ---- weights: A operand of every MFMA, resident in registers
  u32x4_t wB[9][2][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wB[tap][ms][j] = *reinterpret_cast<const u32x4_t*>(p.wk + ((int64_t)(wn * 32 + j * 16 + lr) * 9 + tap) * 64 + ms * 32 + g * 8);
  // bias: parked in LDS (256 B), re-read at the start of every tile as the initial accumulator
  if (tid < 64) reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = p.bias ? p.bias[tid] : 0.f;
  __syncthreads();

  // tile origins (image, first row, first column) without per-tile integer divisions: the walk advances by gridDim.x tiles per
  // iteration, i.e. by a fixed (images, tile rows, tile columns) step with carries; org[k] = origin of tile n + k
  struct Org { int b, h0, w0; };
  const int dtw = (nwg % p.tiles_w) * TW, q1 = nwg / p.tiles_w, dth = (q1 % p.tiles_h) * TH, db = q1 / p.tiles_h;
  const int wlim = p.tiles_w * TW, hlim = p.tiles_h * TH;
  auto advance = [&](Org& o) __attribute__((always_inline)) {
    o.w0 += dtw;
    const bool c1 = o.w0 >= wlim;
    o.w0 -= c1 ? wlim : 0;
    o.h0 += dth + (c1 ? TH : 0);
    const bool c2 = o.h0 >= hlim;
    o.h0 -= c2 ? hlim : 0;
    o.b += db + (c2 ? 1 : 0);
  };
  Org org[DIST + 1];
  {
    int t = vid;
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    org[0].w0 = tw * TW; org[0].h0 = (t % p.tiles_h) * TH; org[0].b = t / p.tiles_h;
#pragma unroll
    for (int k = 1; k <= DIST; ++k) { org[k] = org[k - 1]; advance(org[k]); }
  }
  // per-thread byte offsets relative to the tile's first pixel: rel = the thread's PIT patch chunks, relo = its 4 output chunks
  // (pixel of fragment f = 4 wm + i: row f / CB, column block f % CB, column lr; channels co0 .. co0 + 7)
  const int co0 = wn * 32 + (g & 1) * 16 + (g & 2) * 4;
  int rel[PIT], relo[4];
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int c = tid + it * NT, hp = c >> 3, px = hp % PW;
    rel[it] = ((hp / PW - 1) * p.W + px - 1) * 128 + (((c & 7) ^ (px & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = wm * 4 + i;
    relo[i] = ((f / CB) * p.W + (f % CB) * 16 + lr) * (MASK ? 128 : p.ypix) + co0 * 2;
  }
  // halo patch of tile n -> buffer n % NBUF.  Tiles whose halo lies inside the image: one add per chunk.  Border tiles: per-chunk
  // bounds test, outside pixels come from the zero page (`t_` = the thread index, laundered per tile by the caller so that this
  // arithmetic is redone every tile instead of being hoisted out of the tile loop into registers that the weights leave no room for)
  auto stage = [&](int n, const Org& o_, int t_) __attribute__((always_inline)) {
    const int b = o_.b, h0 = o_.h0, w0 = o_.w0;
    unsigned char* buf = smem + (n % NBUF) * PBYTES;
    const unsigned base = (((unsigned)b * (unsigned)p.H + (unsigned)h0) * (unsigned)p.W + (unsigned)w0) * 128u;   // < 4 GB (launcher)
    const bool inside = h0 >= 1 && w0 >= 1 && h0 + TH + 1 <= p.H && w0 + TW + 1 <= p.W;
    if (inside) {
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        if (it < PIT - 1 || tid + it * NT < NCH) {
          unsigned char* dst = buf + (it * NT + (tid & ~63)) * 16;       // wave-uniform; the DMA adds lane * 16
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (base + (unsigned)rel[it])),
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < PIT; ++it) {
        const int c = t_ + it * NT;
        if (it < PIT - 1 || c < NCH) {
          const int hp = c >> 3, px = hp % PW;
          const int gy = h0 + hp / PW - 1, gx = w0 + px - 1;
          const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          const unsigned char* src = in ? X + (base + (unsigned)rel[it]) : reinterpret_cast<const unsigned char*>(&c64_zero_page);
          unsigned char* dst = buf + (it * NT + (t_ & ~63)) * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
    }
  };

#pragma unroll
  for (int d = 0; d < DIST; ++d)
    if (cnt > d) stage(d, org[d], tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // per-lane operand addressing: LDS byte address of the lane's chunk of (fragment 0, tap row 0) for column shift dx and channel half
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned offk[3][2];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
      offk[dx][ms] = smem_base + (unsigned)((((wm * 4) / CB) * PW + ((wm * 4) % CB) * 16 + lr) * 128) +
                     (unsigned)(((ms * 4 + g) ^ ((lr + dx) & 7)) << 4);
  const unsigned bias_addr = smem_base + (unsigned)(BIAS_OFF + (wn * 32 + 4 * g) * 4);
  const unsigned stash_addr = smem_base + (unsigned)(NBUF * PBYTES + (wave * 4 * 64 + lane) * 16);

#ifdef C64_TIMING
  long long tsec[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  for (int n = 0; n < cnt; ++n) {
    C64_STAMP(5)
    C64_COMPILER_FENCE();
    __builtin_amdgcn_s_barrier();       // patch n landed for every wave; everybody is done with tile n-1 (its buffer is free)
    C64_COMPILER_FENCE();
    C64_STAMP(0)
    const bool more = n + DIST < cnt;
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int b = org[0].b, h0 = org[0].h0, w0 = org[0].w0;
    const unsigned obase = (((unsigned)b * (unsigned)p.H + (unsigned)h0) * (unsigned)p.W + (unsigned)w0) * (MASK ? 128u : (unsigned)p.ypix);
    const bool whole = h0 + TH <= p.H && w0 + TW <= p.W;          // every output pixel of the tile is inside the image
    if (MASK) {                        // this tile's mask chunks -> the lane's private stash (pixels outside the image: clamped)
      const unsigned char* Mk = reinterpret_cast<const unsigned char*>(p.mask);
      unsigned moff[4];
      if (whole) {
#pragma unroll
        for (int i = 0; i < 4; ++i) moff[i] = obase + (unsigned)relo[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int f = (tl >> 7) * 4 + i;
          const int gy = min(h0 + f / CB, p.H - 1), gx = min(w0 + (f % CB) * 16 + (tl & 15), p.W - 1);
          moff[i] = (((unsigned)b * (unsigned)p.H + (unsigned)gy) * (unsigned)p.W + (unsigned)gx) * 128u + (unsigned)co0 * 2u;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned char* dst = stash + ((wave * 4 + i) * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Mk + moff[i]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
    if (more && !C64_ABL(p, 1)) stage(n + DIST, org[DIST], tl);
    C64_COMPILER_FENCE();

    C64_STAMP(1)
    u32x4_t bq[2];                       // bias: issued ahead of the first operand reads, covered by the first step's wait (in order)
    lds_read16(bq[0], bias_addr);
    lds_read16(bq[1], bias_addr + 64);
    f32x4_t acc[4][2];
    unsigned pbd[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) pbd[dx][ms] = offk[dx][ms] + (unsigned)((n % NBUF) * PBYTES);
    if (!C64_ABL(p, 2)) {
      if constexpr (CB == 1) {        // one column block per tile row: the row-ordered form (half the operand reads; conv_c64_core.h)
        c64_rows<PW>(acc, wB, pbd, bq);
      } else {
        u32x4_t a[2][4];
        c64_issue<0, PW, CB>(a[0], pbd);
        c64_steps<PW, CB>(std::make_integer_sequence<int, 18>{}, acc, a, wB, pbd, bq);
      }
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]));
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = __builtin_bit_cast(f32x4_t, bq[0]);
    }

    C64_STAMP(2)
    // patch n+1 (and this tile's mask chunks) must have landed before the next barrier; with NBUF = 3, patch n+2 (just issued) may
    // stay in flight: loads complete in order, so "at most PIT-1 outstanding" implies that everything older than it is done
    C64_COMPILER_FENCE();
    if (DIST == 2 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIT - 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    C64_STAMP(3)
    // ---- epilogue: bf16 pairs, ReLU / mask on the packed halves, lane-group exchange, one 16-byte store per fragment
    u32x4_t mk[4];
    if (MASK) {
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(mk[i]) : "v"(stash_addr), "n"(i * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]), "+v"(mk[2]), "+v"(mk[3]));
    }
    // (branch-free: ReLU = packed signed max with 0, "no ReLU" = max with the most negative int16; validity of a pixel by compares)
    const uint32_t floor2 = p.relu ? 0u : 0x80008000u;
    const int rows_ok = p.H - h0, cols_ok = p.W - w0 - (tl & 15);
    uint32_t keep[POOL ? 4 : 1][2][2];        // POOL: the fragments' packed (ReLU'd) channel pairs, [row][channel fragment][pair]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t lo[2], hi[2];
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        uint32_t pa = pack_bf16(acc[i][0][2 * d], acc[i][0][2 * d + 1]);
        uint32_t pb2 = pack_bf16(acc[i][1][2 * d], acc[i][1][2 * d + 1]);
        asm("v_pk_max_i16 %0, %0, %1" : "+v"(pa) : "s"(floor2));     // max(x, 0) on bf16 bits = signed 16-bit max (rounding keeps the sign)
        asm("v_pk_max_i16 %0, %0, %1" : "+v"(pb2) : "s"(floor2));
        if (POOL) { keep[i][0][d] = pa; keep[i][1][d] = pb2; }
        // (a, b) -> a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3}
        auto sw = __builtin_amdgcn_permlane16_swap(pa, pb2, false, false);
        lo[d] = sw[0]; hi[d] = sw[1];
      }
      uint4 o = make_uint4(lo[0], lo[1], hi[0], hi[1]);
      if (MASK) {
        o.x = c64_mask2(o.x, mk[i][0]); o.y = c64_mask2(o.y, mk[i][1]);
        o.z = c64_mask2(o.z, mk[i][2]); o.w = c64_mask2(o.w, mk[i][3]);
      }
      const int f = (tl >> 7) * 4 + i;
      const bool ok = (f / CB < rows_ok) & ((f % CB) * 16 < cols_ok) & !C64_ABL(p, 4);
      if (ok && (!POOL || p.y != nullptr)) *reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(p.y) + (obase + (unsigned)relo[i])) = o;
    }
    if (POOL) {
      // 2x2 / stride 2 max-pool of the tile (8 x 16 -> 4 x 8 pixels) from the packed halves: the values are ReLU outputs (>= 0), so
      // the signed 16-bit maximum IS the bf16 maximum.  Rows: fragments (0,1) and (2,3) of this wave; columns: lane pairs (lr, lr^1)
      // by a quad-permute DPP move; the even lanes then own a pooled pixel and the usual lane-group exchange builds its 16-byte chunk.
      static_assert(!POOL || (CB == 1 && TH == 8), "pooled epilogue: 8 x 16 tiles");
      const int H2 = p.H >> 1, W2 = p.W >> 1;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          uint32_t va = keep[2 * pr][0][d], vb = keep[2 * pr][1][d];
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(va) : "v"(keep[2 * pr + 1][0][d]));
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(vb) : "v"(keep[2 * pr + 1][1][d]));
          const uint32_t na = (uint32_t)__builtin_amdgcn_mov_dpp((int)va, 0xB1, 0xf, 0xf, true);      // quad_perm [1,0,3,2]: lane ^ 1
          const uint32_t nb = (uint32_t)__builtin_amdgcn_mov_dpp((int)vb, 0xB1, 0xf, 0xf, true);
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(va) : "v"(na));
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(vb) : "v"(nb));
          auto sw = __builtin_amdgcn_permlane16_swap(va, vb, false, false);
          lo[d] = sw[0]; hi[d] = sw[1];
        }
        const int prow = (h0 >> 1) + (tl >> 7) * 2 + pr, pcol = (w0 >> 1) + ((tl & 15) >> 1);
        const int64_t pidx = (((int64_t)b * H2 + prow) * W2 + pcol) * 64 + co0;
        if (!(tl & 1) && prow < H2 && pcol < W2)
          *reinterpret_cast<uint4*>(p.pool + pidx) = make_uint4(lo[0], lo[1], hi[0], hi[1]);
        if (p.code != nullptr) {
          // selection byte per pooled element: 0 where the maximum is 0 (a ReLU output: no gradient), else 1 + position of the FIRST
          // maximum in scan order (row 0: this lane pair's even / odd lane, row 1 the same).  Packed 16-bit arithmetic on the bf16 bit
          // patterns (non-negative values: equal numbers <=> equal bits): ne_k = min(v_k xor m, 1) per half,
          // code = (1 + ne0 + ne0 ne1 + ne0 ne1 ne2) * min(m, 1).  Every lane computes its pair's code; the even lane stores.
          uint32_t clo[2], chi[2];
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            uint32_t cw[2];
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) {
              const uint32_t mine0 = keep[2 * pr][cf][d], mine1 = keep[2 * pr + 1][cf][d];
              const uint32_t oth0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine0, 0xB1, 0xf, 0xf, true);      // lane ^ 1
              const uint32_t oth1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine1, 0xB1, 0xf, 0xf, true);
              const bool odd = (tl & 1) != 0;
              const uint32_t v0 = odd ? oth0 : mine0, v1 = odd ? mine0 : oth0, v2 = odd ? oth1 : mine1, v3 = odd ? mine1 : oth1;
              uint32_t m = v0;
              asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v1));
              asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v2));
              asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v3));
              const uint32_t one = 0x00010001u;
              uint32_t n0 = v0 ^ m, n1 = v1 ^ m, n2 = v2 ^ m, nz = m;
              asm("v_pk_min_u16 %0, %0, %1" : "+v"(n0) : "v"(one));
              asm("v_pk_min_u16 %0, %0, %1" : "+v"(n1) : "v"(one));
              asm("v_pk_min_u16 %0, %0, %1" : "+v"(n2) : "v"(one));
              asm("v_pk_min_u16 %0, %0, %1" : "+v"(nz) : "v"(one));
              const uint32_t n01 = n0 & n1, n012 = n01 & n2;
              uint32_t c = one + n0 + n01 + n012;            // halves stay <= 4: no carry between them
              asm("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(c) : "v"(nz));
              cw[cf] = c;
            }
            auto sw = __builtin_amdgcn_permlane16_swap(cw[0], cw[1], false, false);    // the same lane-group exchange as the values
            clo[d] = sw[0]; chi[d] = sw[1];
          }
          if (!(tl & 1) && prow < H2 && pcol < W2) {
            // 8 codes in channel order as 16-bit halves of (clo[0], clo[1], chi[0], chi[1]) -> 8 bytes
            const uint32_t b0 = __builtin_amdgcn_perm(clo[1], clo[0], 0x06040200u), b1 = __builtin_amdgcn_perm(chi[1], chi[0], 0x06040200u);
            *reinterpret_cast<uint2*>(p.code + pidx) = make_uint2(b0, b1);
          }
        }
      }
    }
    C64_STAMP(4)
#pragma unroll
    for (int k = 0; k < DIST; ++k) org[k] = org[k + 1];
    advance(org[DIST]);
  }
#ifdef C64_TIMING
  if (p.dbg && blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 6; ++k) p.dbg[wave * 8 + k] = tsec[k];
  if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[7] = cnt;
#endif
}

template <int TW, int TH, bool MASK, int NBUF, bool POOL = false>
int launch_t(C64Args p, hipStream_t s) {
  p.tiles_h = (p.H + TH - 1) / TH;
  p.tiles_w = (p.W + TW - 1) / TW;
  const int64_t nt = (int64_t)p.B * p.tiles_h * p.tiles_w;
  if (nt >= ((int64_t)1 << 31)) return ASR_EUNSUPPORTED;
  p.ntiles = (int)nt;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  constexpr int WM = TW * TH / 64;
  const size_t lds = (size_t)NBUF * (TH + 2) * (TW + 2) * 128 + (MASK ? WM * 2 * 4096 : 0) + 256;
  const int per_cu = (int)(163840 / lds) < 8 / (WM * 2) ? (int)(163840 / lds) : 8 / (WM * 2);   // LDS- and register-limited
  static bool granted = false;          // per instantiation; the first (eager / warm-up) launch does it, never a captured one
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_c64_kernel<TW, TH, MASK, NBUF, POOL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ASR_ELAUNCH;
    granted = true;
  }
  const int pc = (int)asr_tuning("C64_PER_CU", 0);    // tuning: workgroups per CU (0 = what the occupancy query says)
  const int64_t slots = (int64_t)cus * (pc > 0 ? pc : (per_cu > 0 ? per_cu : 1));
  const unsigned grid = (unsigned)(nt < slots ? nt : slots);
#ifdef C64_TIMING
  static long long* dbg = nullptr;
  if (!dbg) { (void)hipMalloc(&dbg, 64 * 8); }
  (void)hipMemset(dbg, 0, 64 * 8);
  p.dbg = dbg;
#endif
  hipLaunchKernelGGL((conv3x3_c64_kernel<TW, TH, MASK, NBUF, POOL>), dim3(grid), dim3(WM * 128), lds, s, p);
  ASR_LAUNCH_CHECK();
#ifdef C64_TIMING
  {
    long long h[64];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
    static int shown = 0;
    if (shown++ < 2)
      for (int w = 0; w < WM * 2; ++w)
        fprintf(stderr, "c64 timing wave %d tiles %lld: barrier %lld stage %lld mfma %lld dmawait %lld epilogue %lld looptop %lld (100 MHz ticks)\n", w,
                h[7], h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5]);
  }
#endif
  return ASR_OK;
}

}  // namespace

int asr_conv3x3_c64_launch(const C64Args& a_, hipStream_t s) {
  C64Args a = a_;
  if (a.ypix == 0) a.ypix = 128;
  if (a.ypix != 128 && (a.mask || a.pool)) return ASR_EUNSUPPORTED;
  a.ablate = (int)asr_tuning("C64_ABLATE", 0);
  // shape (ASR_C64_SHAPE, tuning): 0 = two 4-wave workgroups per CU on 8 x 16 pixel tiles (default: the two workgroups are not in
  // phase, so one's address / epilogue VALU work overlaps the other's MFMAs); 1 / 2 = one 8-wave workgroup on 16 x 16 / 8 x 32 tiles
  const int shape = (int)asr_tuning("C64_SHAPE", 0);
  if (a.pool) {                      // pooled epilogue: forward with ReLU on the default shape only
    if (a.mask || !a.relu) return ASR_EUNSUPPORTED;
    return launch_t<16, 8, false, 3, true>(a, s);
  }
  if (shape == 1) return a.mask ? launch_t<16, 16, true, 3>(a, s) : launch_t<16, 16, false, 3>(a, s);
  if (shape == 2) return a.mask ? launch_t<32, 8, true, 3>(a, s) : launch_t<32, 8, false, 3>(a, s);
  return a.mask ? launch_t<16, 8, true, 2>(a, s) : launch_t<16, 8, false, 3>(a, s);
}
