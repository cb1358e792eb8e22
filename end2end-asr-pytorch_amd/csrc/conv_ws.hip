// 3x3 convolution, bf16 NHWC, weight-stationary: the four launches of vgg_cnn's second level that are not weight gradients
// (reference: models/asr/transformer.py:48-52 -- conv.5 forward (64 -> 128, one pass), conv.7 forward with its ReLU + MaxPool2d + the
// (B, T', C F') view / transpose of :74-76, conv.7's data gradient with conv.5's ReLU mask, conv.5's data gradient).  The notes below
// describe the 128-input-channel form; WsGeo<64, 4> is the same kernel with 144 registers of weights per wave and two workgroups per CU.
//
// The generic implicit GEMM (conv.hip) re-reads all 9 x 128 x Cout weights (295 KB at Cout = 128) from L2 for every 16 x 16-pixel
// workgroup tile, through registers into a single LDS buffer with two barriers per tap: its MFMA loop alone runs at 55 % of the peak,
// the shipped kernel at 37 - 40 % (profiles/r04_igemm_ablation.txt).  Here the recipe of conv_c64.hip is taken to 128 input channels:
//   * ONE persistent 4-wave workgroup per CU (one wave per SIMD, the whole 512-entry register file each).  Wave wn owns 32 output
//     channels and keeps their 32 x 9 x 128 weights -- 72 MFMA operands, 288 registers, most of them in the accumulation half of the
//     unified file, where an MFMA reads its A operand just as well -- for the lifetime of the kernel: no weight traffic, no per-tap
//     barrier, one s_barrier per tile.  Cout = 128: the four waves share all 8 pixel fragments of a tile (4 x 32 channels);
//     Cout = 64: two waves per channel half, each 4 pixel fragments.
//   * tiles of 8 x 16 output pixels; the 10 x 18 x 128-channel halo patch (46 KB) of tile n + 1 travels HBM -> LDS by the LDS-DMA
//     while tile n is contracted (two patch buffers; a tile is ~4 us of MFMAs, one tile ahead covers the latency).
//   * LDS image: 16-byte chunk c of the pixel in patch column x sits in slot c ^ (x & 15) of the pixel's 256 bytes; fragment COLUMN l
//     of an MFMA is pixel pix(l) of the 16-pixel row segment (even pixels in lanes 0-3 / 12-15, the odd neighbour in lane l ^ 8) and
//     lane group g contracts channel chunk 4 ms + {0, 2, 1, 3}[g]: every ds_read_b128 lane group (MI355X_MICROARCH.md, LDS) then touches
//     16 different slots whatever the tap's column shift -- and the 2 x 2 pooling partner of a lane is a row rotation by 8 (one DPP).
//   * operand reads are hand-issued ds_read_b128 with instruction-immediate (fragment row, tap row) offsets, two units (4 fragments x
//     2 channel fragments = 8 MFMAs each) ahead of the MFMAs that consume them.
//   * epilogues from the accumulators: NHWC rows as in conv_c64.hip (bias / ReLU / mask on packed bf16, v_permlane16_swap -> one
//     16-byte chunk per lane); pooled form: 2 x 2 maximum over row pairs (registers) and lane pairs (DPP), selection byte per pooled
//     element (packed 16-bit arithmetic), and -- a wave holds all 8 rows of a tile -- the 4 pooled rows of a (column, channel) as one
//     8-byte run of the (B, W/2, C, H/2) encoder layout straight from registers (no LDS staging, no barrier); with H % 16 == 0 the tiles
//     are walked in vertical pairs and a run leaves as one 16-byte + one 8-byte store (EP = 2).
// What bounds it: the contraction runs at 16.4 cycles per MFMA, the matrix pipe's own rate; the tile's 20 - 36
// memory instructions and its epilogue are serial to it at one wave per SIMD (61 % MFMA-busy in cycles) -- and the chip's power budget: a
// pure MFMA stream sustains 1.8 PF on activations-like operands (tools/probes/mfma_power_probe.hip), 1.69 PF with this loop's 0.5 LDS
// operand reads per MFMA beside it (mfma_lds_power_probe.hip); the 64-channel form's two workgroups per CU double the work per cycle and
// run at 0.73 x the clock.
#include "common.h"
#include "conv_c64_core.h"
#include "conv_ws.h"

#include <stdio.h>

#include <utility>

namespace {

#define WS_FENCE() asm volatile("" ::: "memory")

constexpr int WS_TW = 16, WS_PW = 18, WS_NBUF = 2;
// geometry of a (CI input channels, TH-row tile) instantiation: CI = 128 / TH = 8 -- the three 128-channel launches, one workgroup per
// CU; CI = 64 / TH = 4 -- conv.5's forward (64 -> 128) in ONE pass, two workgroups per CU (144 registers of weights per wave)
template <int CI, int TH> struct WsGeo {
  static constexpr int KS = CI / 32;                    // k steps per tap
  static constexpr int CPP = CI / 8;                    // 16-byte chunks per pixel
  static constexpr int PSZ = CI * 2;                    // bytes per pixel
  static constexpr int NHALO = (TH + 2) * WS_PW;        // halo pixels of a tile (180 / 108)
  static constexpr int PB = NHALO * PSZ;                // bytes of one patch buffer
  static constexpr int NCH = NHALO * CPP;               // 16-byte chunks of a patch
  static constexpr int PIT = (NCH + 255) / 256;         // DMA instructions per thread and patch (the last one partial)
  // swizzle key of patch column x: the slot of chunk c is c ^ key(x).  256-byte pixels: x & 15; 128-byte pixels (two per bank line):
  // (x >> 1) & 7 -- either way the 16 lanes of a ds_read_b128 lane group touch 16 different 16-byte slots of a bank line
  __device__ static __forceinline__ int key(int x) { return CI == 128 ? (x & 15) : ((x >> 1) & 7); }
};

// The contraction of a tile, ordered by PATCH ROW (round 6; conv_c64_core.h has the 64-channel twin).  For a fixed (tap column dx, k step)
// the operand of patch row r -- 16 pixels x 32 channels, one ds_read_b128 -- serves every (tap row dy, tile row i) with i + dy = r and both
// channel fragments of the wave: FM + 2 reads for 6 FM MFMAs instead of the 3 FM reads of the (tap, k step, 4 rows) units this replaces
// (0.21 reads per MFMA at FM = 8, 0.25 at FM = 4, against 0.5).  The kernels are bound by the chip's power budget, and LDS operand reads
// beside the MFMAs are energy: mfma_lds_power_probe.hip prices 0.5 reads per MFMA at 10 % of the sustained rate and 0.19 at 5 %.
// Item K = (group G = K / (FM + 2): dx = G / KS, k step = G % KS; row r = K % (FM + 2)); reads run D items ahead in a ring of D + 1 quads.
template <int K, int FM, int KS, int PSZ>
__device__ __forceinline__ void ws_row_issue(u32x4_t& dst, const unsigned (&pbd)[3][KS]) {
  constexpr int R = FM + 2, G = K / R, r = K % R, dx = G / KS, ms = G % KS;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(pbd[dx][ms]), "n"(r * WS_PW * PSZ));
}
template <int K, int FM, int D, int KS, int PSZ>
__device__ __forceinline__ void ws_row_item(f32x4_t (&acc)[FM][2], u32x4_t (&ring)[D + 1], const u32x4_t (&wB)[9][KS][2],
                                            const unsigned (&pbd)[3][KS], u32x4_t (&bq)[2]) {
  constexpr int R = FM + 2, NI = 3 * KS * R, G = K / R, r = K % R, dx = G / KS, ms = G % KS;
  if constexpr (K + D < NI) ws_row_issue<K + D, FM, KS, PSZ>(ring[(K + D) % (D + 1)], pbd);
  constexpr int pending = (K + D < NI) ? D : (NI - 1 - K);      // reads issued after read K (they return in order)
  u32x4_t& cur = ring[K % (D + 1)];
  if constexpr (K == 0) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(cur), "+v"(bq[0]), "+v"(bq[1]) : "n"(pending));
  else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"(pending));
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int i = r - dy;
    if (i < 0 || i >= FM) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)       // the first touch of an accumulator (group 0, tap row 0) starts it from the bias of its 4 output channels
      acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wB[dy * 3 + dx][ms][j]), __builtin_bit_cast(bf16x8_t, cur),
                                                          (G == 0 && dy == 0) ? __builtin_bit_cast(f32x4_t, bq[j]) : acc[i][j], 0, 0, 0);
  }
}
template <int FM, int D, int KS, int PSZ, int... K>
__device__ __forceinline__ void ws_rows_seq(std::integer_sequence<int, K...>, f32x4_t (&acc)[FM][2], u32x4_t (&ring)[D + 1],
                                            const u32x4_t (&wB)[9][KS][2], const unsigned (&pbd)[3][KS], u32x4_t (&bq)[2]) {
  (ws_row_item<K, FM, D, KS, PSZ>(acc, ring, wB, pbd, bq), ...);
}
template <int FM, int D, int KS, int PSZ, int... P>
__device__ __forceinline__ void ws_rows_prologue(std::integer_sequence<int, P...>, u32x4_t (&ring)[D + 1], const unsigned (&pbd)[3][KS]) {
  (ws_row_issue<P, FM, KS, PSZ>(ring[P], pbd), ...);
}
template <int FM, int KS, int PSZ, int D = 3>
__device__ __forceinline__ void ws_rows(f32x4_t (&acc)[FM][2], const u32x4_t (&wB)[9][KS][2], const unsigned (&pbd)[3][KS], u32x4_t (&bq)[2]) {
  u32x4_t ring[D + 1];
  ws_rows_prologue<FM, D, KS, PSZ>(std::make_integer_sequence<int, D>{}, ring, pbd);
  ws_rows_seq<FM, D, KS, PSZ>(std::make_integer_sequence<int, 3 * KS * (FM + 2)>{}, acc, ring, wB, pbd, bq);
}

// EP = 0: y (B, H, W, CO) NHWC, optional ReLU / mask.  EP = 1: ReLU + 2x2 max-pool + selection codes in the (B, W/2, CO, H/2) layout.
// EP = 2: the same with the tiles walked in VERTICAL PAIRS (H % 16 == 0): the first tile's 4 pooled rows wait in registers for the second's, and a
// (column, channel) run leaves as ONE 16-byte + ONE 8-byte store instead of two 8-byte + two 4-byte ones -- a store instruction costs the SIMD
// ~150 cycles whatever its width, the pooled epilogue was 16 of them per tile.
// TM = true (tuning WS_DBG = device address of 64 int64): every wave of workgroup 0 stamps the shader clock at the section boundaries
// of a tile and leaves its totals {barrier, staging issue, contraction, DMA wait, epilogue, tiles} in dbg[wave * 8 ..]
template <int CI, int TH, int CO, int MK, int EP, bool TM = false>
__global__ __launch_bounds__(256, CI == 64 ? 2 : 1) void conv3x3_ws128_kernel(WsArgs p) {
  using G = WsGeo<CI, TH>;
  constexpr int KS = G::KS, CPP = G::CPP, PSZ = G::PSZ, WS_PB = G::PB, WS_NCH = G::NCH, WS_PIT = G::PIT, WS_TH = TH;
  static_assert(CO == 64 || CO == 128, "output channels");
  // MK: 0 = no mask; 1 = output zeroed where the bf16 tensor p.mask is <= 0 (the lane's 16-byte mask chunks by LDS-DMA into a private
  // stash); 2 = the same from ONE BIT per element (p.bits_in, layout below): a single 8-byte load per lane and tile instead of 8 DMA
  // pieces, 16 MB of mask instead of 262 MB; 3 = no mask, but the epilogue WRITES such bits for its ReLU output (p.bits_out: conv.5
  // forward for conv.7's data gradient).  Bit layout (asr_hip.h: asr_relu_bits_bytes): one dword per (4-row x 16-column tile, wave, lane) in
  // THIS kernel's lane order -- [b][h / 4][w / 16][co / 32][lane] with byte r = row 4 (h / 4) + r, bit k = channel co0(lane) + k of pixel
  // column pix(lane) -- so that the producer's store and the consumer's two loads are 256 contiguous bytes per wave.  (A pixel-major
  // layout, [b][h / 8][w][c / 8][h % 8], was measured first: its 4-byte stores at 8-byte stride cost the producer 160 us instead of 125,
  // profiles/r05_relu_bits.txt -- partial-sector writes, not instructions: the same kernel WITHOUT the store ran in 119.)
  constexpr bool MASK = MK == 1, BITS_IN = MK == 2, BITS_OUT = MK == 3;
  static_assert(EP == 0 || (CO == 128 && MK == 0 && TH == 8 && CI == 128), "pooled epilogue: conv.7 forward");
  static_assert(!BITS_IN || (TH == 8 && CO == 128), "bit mask in: 8-row tiles, a wave holds all 8 rows");
  static_assert(!BITS_OUT || (TH == 4 && CO == 128 && EP == 0), "bit mask out: the 4-row form");
  constexpr bool POOLED = EP != 0, PAIR = EP == 2;
  constexpr int WN = CO / 32, WM = 4 / WN, FM = TH / WM;     // waves along channels / pixel rows, pixel fragments (tile rows) per wave
  static_assert(FM == 4 || FM == 8, "4 or 8 tile rows per wave");
  constexpr int NA = CI == 128 ? 64 : 32;                    // weight operands that live in the accumulation half of the register file
  constexpr int STASH = MASK ? 4 * FM * 1024 : 0;            // the lane's FM mask chunks of the tile
  constexpr int BIAS_OFF = WS_NBUF * WS_PB + STASH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* stash = smem + WS_NBUF * WS_PB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  // fragment column -> pixel of the row segment, lane group -> channel chunk of a k step (see the header)
  const int la = lr >> 2, pix = 8 * (la & 1) + 2 * (lr & 3) + (((la >> 1) ^ la) & 1);
  const int gs = ((g & 1) << 1) | (g >> 1);
  const int nwg = gridDim.x;
  const int vid = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;      // consecutive tiles on one XCD
  // PAIR: the unit of the walk is a pair of vertically adjacent tiles (2 q, 2 q + 1 in the row-tiles-fastest order; tiles_h is even)
  const int nitems = PAIR ? p.ntiles / 2 : p.ntiles;
  const int cnt = (vid < nitems ? (nitems - vid + nwg - 1) / nwg : 0) * (PAIR ? 2 : 1);
  const unsigned char* X = reinterpret_cast<const unsigned char*>(p.x);

  // ---- weights: A operand of every MFMA, resident in registers for the whole kernel
  u32x4_t wB[9][KS][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ms = 0; ms < KS; ++ms)
#pragma unroll
      for (int j = 0; j < 2; ++j)
      {
        // 64 of the 72 operands are loaded STRAIGHT into the accumulation half of the register file ("=a"): the MFMA reads its A operand
        // from there directly.  Left to itself the allocator treats that half as spill space and pays four v_accvgpr_read per use.
        const bf16_t* src = p.wk + ((int64_t)(wn * 32 + j * 16 + lr) * 9 + tap) * CI + (ms * 4 + gs) * 8;
        if ((tap * KS + ms) * 2 + j < NA) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wB[tap][ms][j]) : "v"(src));
        else wB[tap][ms][j] = *reinterpret_cast<const u32x4_t*>(src);
      }
  if (tid < CO) reinterpret_cast<float*>(smem + BIAS_OFF)[tid] = p.bias ? p.bias[tid] : 0.f;

  // tile n of this workgroup -> (image, first row, first column).  EP = 1 walks the row tiles fastest: the pieces of a
  // (column, channel) run of the transposed output then meet in one L2 before the line is written back.
  struct Org { int b, h0, w0; };
  auto origin = [&](int n) __attribute__((always_inline)) {
    int t = PAIR ? 2 * (vid + (n >> 1) * nwg) + (n & 1) : vid + n * nwg;
    Org o;
    if (POOLED) {
      o.h0 = (t % p.tiles_h) * WS_TH; t /= p.tiles_h;
      o.w0 = (t % p.tiles_w) * WS_TW; o.b = t / p.tiles_w;
    } else {
      o.w0 = (t % p.tiles_w) * WS_TW; t /= p.tiles_w;
      o.h0 = (t % p.tiles_h) * WS_TH; o.b = t / p.tiles_h;
    }
    return o;
  };
  // per-thread byte offsets relative to the tile's first pixel: rel = the thread's patch chunks, relo = its output chunk of fragment 0
  // (pixel column pix, channels co0 .. co0 + 7 after the lane-group exchange of the epilogue)
  const int co0 = wn * 32 + (g & 1) * 16 + (g & 2) * 4;
  int rel[WS_PIT];
#pragma unroll
  for (int it = 0; it < WS_PIT; ++it) {
    const int c = tid + it * 256, hp = c / CPP, px = hp % WS_PW;
    rel[it] = ((hp / WS_PW - 1) * p.W + px - 1) * PSZ + (((c % CPP) ^ G::key(px)) << 4);
  }
  const int relo = ((wm * FM) * p.W + pix) * (CO * 2) + co0 * 2;
  auto stage = [&](int n, const Org& o_, int t_) __attribute__((always_inline)) {
    const int b = o_.b, h0 = o_.h0, w0 = o_.w0;
    unsigned char* buf = smem + (n % WS_NBUF) * WS_PB;
    const unsigned base = (((unsigned)b * (unsigned)p.H + (unsigned)h0) * (unsigned)p.W + (unsigned)w0) * (unsigned)PSZ;   // < 4 GB (launcher)
    const bool inside = h0 >= 1 && w0 >= 1 && h0 + WS_TH + 1 <= p.H && w0 + WS_TW + 1 <= p.W;
    if (inside) {
#pragma unroll
      for (int it = 0; it < WS_PIT; ++it) {
        if (it < WS_PIT - 1 || tid + it * 256 < WS_NCH) {
          unsigned char* dst = buf + (it * 256 + (tid & ~63)) * 16;       // wave-uniform; the DMA adds lane * 16
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (base + (unsigned)rel[it])),
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
    } else {        // border tile: per-chunk bounds test, outside pixels come from the zero page (the DMA cannot zero-fill)
#pragma unroll
      for (int it = 0; it < WS_PIT; ++it) {
        const int c = t_ + it * 256;
        if (it < WS_PIT - 1 || c < WS_NCH) {
          const int hp = c / CPP, px = hp % WS_PW;
          const int gy = h0 + hp / WS_PW - 1, gx = w0 + px - 1;
          const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
          const unsigned char* src = in ? X + (base + (unsigned)rel[it]) : reinterpret_cast<const unsigned char*>(&c64_zero_page);
          unsigned char* dst = buf + (it * 256 + (t_ & ~63)) * 16;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        }
      }
    }
  };

  if (cnt > 0) stage(0, origin(0), tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // every kernel argument the tile loop uses is consumed once HERE: the compiler otherwise waits for its scalar loads (lgkmcnt(0) --
  // the counter our hand-issued operand reads use) at their first use inside the loop, draining the read pipeline once per tile
  asm volatile("" ::"s"(p.y), "s"(p.pool), "s"(p.code), "s"(p.mask), "s"(p.H), "s"(p.W), "s"(p.relu), "s"(p.tiles_h), "s"(p.tiles_w), "s"(p.code_cl));

  // per-lane operand addressing: LDS byte address of the lane's chunk of (fragment 0, tap row 0) for column shift dx and k step ms
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned offk[3][KS];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ms = 0; ms < KS; ++ms)
      offk[dx][ms] = smem_base + (unsigned)(((wm * FM) * WS_PW + pix + dx) * PSZ) + (unsigned)(((ms * 4 + gs) ^ G::key(pix + dx)) << 4);
  const unsigned bias_addr = smem_base + (unsigned)(BIAS_OFF + (wn * 32 + 4 * g) * 4);
  const unsigned stash_addr = smem_base + (unsigned)(WS_NBUF * WS_PB + (wave * FM * 64 + lane) * 16);

  uint2 holdv[PAIR ? 8 : 1];          // PAIR: the upper tile's pooled rows and selection bytes of the lane's 8 channels
  uint32_t holdc[PAIR ? 8 : 1];
  long long tsec[5] = {0, 0, 0, 0, 0}, tlast = TM ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define WS_STAMP(K) if (TM) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); tsec[K] += now_ - tlast; tlast = now_; }
  for (int n = 0; n < cnt; ++n) {
    WS_FENCE();
    __builtin_amdgcn_s_barrier();       // patch n landed for every wave; everybody is done with tile n - 1 (its buffer is free)
    WS_FENCE();
    WS_STAMP(0)
    const bool more = n + 1 < cnt;
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const Org og = origin(n);
    const int b = og.b, h0 = og.h0, w0 = og.w0;
    const unsigned obase = (((unsigned)b * (unsigned)p.H + (unsigned)h0) * (unsigned)p.W + (unsigned)w0) * (unsigned)(CO * 2);
    const bool whole = h0 + WS_TH <= p.H && w0 + WS_TW <= p.W;          // every output pixel of the tile is inside the image
    if (MASK) {                        // this tile's mask chunks -> the lane's private stash (pixels outside the image: clamped)
      const unsigned char* Mk = reinterpret_cast<const unsigned char*>(p.mask);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        unsigned moff;
        if (whole) {
          moff = obase + (unsigned)relo + (unsigned)(i * p.W * (CO * 2));
        } else {
          const int gy = min(h0 + wm * FM + i, p.H - 1), gx = min(w0 + pix, p.W - 1);
          moff = (((unsigned)b * (unsigned)p.H + (unsigned)gy) * (unsigned)p.W + (unsigned)gx) * (unsigned)(CO * 2) + (unsigned)co0 * 2u;
        }
        unsigned char* dst = stash + ((wave * FM + i) * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Mk + moff),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
    uint2 mbits = make_uint2(0u, 0u);
    if constexpr (BITS_IN) {           // the lane's 8 mask bytes of the tile (rows 0 .. 7 of its pixel column, its 8-channel chunk)
      const uint8_t* src = p.bits_in + ((((size_t)b * (2 * ((p.H + 7) >> 3)) + (h0 >> 2)) * p.tiles_w + (w0 >> 4)) * 256 + tl) * 4;
      asm volatile("global_load_dword %0, %1, off" : "=v"(mbits.x) : "v"(src));                        // rows 0 .. 3
      asm volatile("global_load_dword %0, %1, off" : "=v"(mbits.y) : "v"(src + (size_t)p.tiles_w * 1024));     // rows 4 .. 7: the 4-row tile below
    }
    if (more) stage(n + 1, origin(n + 1), tl);
    WS_FENCE();
    WS_STAMP(1)

    u32x4_t bq[2];                       // bias: issued ahead of the first operand reads, covered by the first unit's wait (in order)
    lds_read16(bq[0], bias_addr);
    lds_read16(bq[1], bias_addr + 64);
    f32x4_t acc[FM][2];
    unsigned pbd[3][KS];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ms = 0; ms < KS; ++ms) pbd[dx][ms] = offk[dx][ms] + (unsigned)((n % WS_NBUF) * WS_PB);
    ws_rows<FM, KS, PSZ>(acc, wB, pbd, bq);

    // patch n + 1 (and this tile's mask chunks) must have landed before the next barrier
    WS_FENCE();
    WS_STAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (BITS_IN) asm volatile("" : "+v"(mbits));       // valid from here on (the wait above is this load's too)
    WS_STAMP(3)

    if constexpr (EP == 0) {
      // ---- NHWC epilogue: bf16 pairs, ReLU / mask on the packed halves, lane-group exchange, one 16-byte store per fragment
      u32x4_t mk[FM];
      if (MASK) {
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(mk[i]) : "v"(stash_addr), "n"(i * 1024));
#pragma unroll
        for (int i = 0; i < FM; i += 4)
          asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(mk[i]), "+v"(mk[i + 1]), "+v"(mk[i + 2]), "+v"(mk[i + 3]) : "n"(FM - 4 - i));
      }
      const uint32_t floor2 = p.relu ? 0u : 0x80008000u;       // ReLU = packed signed max with 0, "no ReLU" = max with the most negative int16
      uint32_t rowbits[BITS_OUT ? FM : 1];
      const int rows_ok = p.H - h0 - (tl >> 6) / WN * FM;
      const bool col_ok = w0 + pix < p.W;
      unsigned char* yb = reinterpret_cast<unsigned char*>(p.y) + (obase + (unsigned)relo);
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          uint32_t pa = pack_bf16(acc[i][0][2 * d], acc[i][0][2 * d + 1]);
          uint32_t pb2 = pack_bf16(acc[i][1][2 * d], acc[i][1][2 * d + 1]);
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(pa) : "s"(floor2));
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(pb2) : "s"(floor2));
          // (a, b) -> a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3}
          auto sw = __builtin_amdgcn_permlane16_swap(pa, pb2, false, false);
          lo[d] = sw[0]; hi[d] = sw[1];
        }
        uint4 o = make_uint4(lo[0], lo[1], hi[0], hi[1]);
        if (MASK) {
          o.x = c64_mask2(o.x, mk[i][0]); o.y = c64_mask2(o.y, mk[i][1]);
          o.z = c64_mask2(o.z, mk[i][2]); o.w = c64_mask2(o.w, mk[i][3]);
        }
        if constexpr (BITS_IN) {         // byte i of the lane's 8: bit k = channel co0 + k kept; bit -> 16-bit mask by a signed bit-field extract
          const uint32_t by = ((i < 4 ? mbits.x : mbits.y) >> (8 * (i & 3)));
          uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const uint32_t l = (uint32_t)__builtin_amdgcn_sbfe((int)by, 2 * d, 1), h = (uint32_t)__builtin_amdgcn_sbfe((int)by, 2 * d + 1, 1);
            ow[d] &= __builtin_amdgcn_perm(h, l, 0x07060100u);
          }
        }
        if constexpr (BITS_OUT) {        // the chunk's 8 sign tests (ReLU output: > 0 <=> bits != 0) -> byte i of the lane's dword
          const uint32_t one = 0x00010001u;
          uint32_t t[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int d = 0; d < 4; ++d) asm("v_pk_min_u16 %0, %0, %1" : "+v"(t[d]) : "v"(one));       // 1 per non-zero half
          uint32_t e = t[0] | (t[1] << 2);                      // channels 0, 2, 4, 6 at bits 0, 2, 4, 6; 1, 3, 5, 7 at bits 16, 18, 20, 22
          e |= t[2] << 4;
          e |= t[3] << 6;
          rowbits[i] = e | (e >> 15);                           // low byte = the mask byte (the rest is dropped by the byte pick below)
        }
        if (col_ok && i < rows_ok) *reinterpret_cast<uint4*>(yb + (size_t)i * p.W * (CO * 2)) = o;
      }
      if constexpr (BITS_OUT) {
        const uint32_t obits = __builtin_amdgcn_perm(__builtin_amdgcn_perm(rowbits[3], rowbits[2], 0x0c0c0400u),
                                                     __builtin_amdgcn_perm(rowbits[1], rowbits[0], 0x0c0c0400u), 0x05040100u);
        *reinterpret_cast<uint32_t*>(p.bits_out + ((((size_t)b * (2 * ((p.H + 7) >> 3)) + (h0 >> 2)) * p.tiles_w + (w0 >> 4)) * 256 + tl) * 4) = obits;
      }
    } else {
      // ---- pooled epilogue (launcher: H % 8 == 0, W % 16 == 0, so every tile is whole).  Values are ReLU outputs (>= 0): the
      // unsigned 16-bit maximum of the bf16 bit patterns IS the bf16 maximum and equal numbers have equal bits.  Window scan order
      // = (row 0: even, odd column; row 1: even, odd): v0 .. v3 of the EVEN pixel's lane (the odd one computes don't-cares).
      const uint32_t one = 0x00010001u;
      uint32_t mx[4][2][2], cd[4][2][2];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            uint32_t v0 = pack_bf16(acc[2 * pr][j][2 * d], acc[2 * pr][j][2 * d + 1]);
            uint32_t v2 = pack_bf16(acc[2 * pr + 1][j][2 * d], acc[2 * pr + 1][j][2 * d + 1]);
            asm("v_pk_max_i16 %0, %0, 0" : "+v"(v0));
            asm("v_pk_max_i16 %0, %0, 0" : "+v"(v2));
            const uint32_t v1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v0, 0x128, 0xf, 0xf, true);      // row_ror:8 = lane ^ 8
            const uint32_t v3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)v2, 0x128, 0xf, 0xf, true);
            uint32_t m = v0;
            asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v1));
            asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v2));
            asm("v_pk_max_u16 %0, %0, %1" : "+v"(m) : "v"(v3));
            uint32_t n0 = v0 ^ m, n1 = v1 ^ m, n2 = v2 ^ m, nz = m;
            asm("v_pk_min_u16 %0, %0, %1" : "+v"(n0) : "v"(one));
            asm("v_pk_min_u16 %0, %0, %1" : "+v"(n1) : "v"(one));
            asm("v_pk_min_u16 %0, %0, %1" : "+v"(n2) : "v"(one));
            asm("v_pk_min_u16 %0, %0, %1" : "+v"(nz) : "v"(one));
            const uint32_t n01 = n0 & n1, n012 = n01 & n2;
            uint32_t c = one + n0 + n01 + n012;            // halves stay <= 4: no carry between them
            asm("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(c) : "v"(nz));
            mx[pr][j][d] = m; cd[pr][j][d] = c;
          }
      // selection bytes channel last (p.code_cl; B, W/2, H/2, 128): the lane's 4 + 4 bytes of a pooled row are exchanged between the
      // lane groups like the NHWC epilogue's values -- 8 consecutive channels, one 8-byte store per pooled row, the four lane groups of
      // a pixel one whole 32-byte sector.  (All lanes take part in the exchange; the even pixel's lanes store.)
      uint2 ccl[4];
      if (p.code_cl) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const uint32_t ca = __builtin_amdgcn_perm(cd[pr][0][1], cd[pr][0][0], 0x06040200u);
          const uint32_t cb = __builtin_amdgcn_perm(cd[pr][1][1], cd[pr][1][0], 0x06040200u);
          auto sw = __builtin_amdgcn_permlane16_swap(ca, cb, false, false);
          ccl[pr] = make_uint2(sw[0], sw[1]);
        }
      }
      if ((pix & 1) == 0) {
        const int H2 = p.H >> 1, W2 = p.W >> 1;
        if (p.code_cl) {
          uint8_t* cq = p.code + ((((int64_t)b * W2 + ((w0 + pix) >> 1)) * H2 + (h0 >> 1)) * 128 + co0);
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) *reinterpret_cast<uint2*>(cq + pr * 128) = ccl[pr];
        }
        const int64_t e0 = (((int64_t)b * W2 + ((w0 + pix) >> 1)) * 128 + wn * 32 + 4 * g) * H2 + (h0 >> 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = r >> 1;
            const uint32_t selv = (r & 1) ? 0x07060302u : 0x05040100u;       // the channel's half of (second, first) pooled row
            const uint32_t selc = (r & 1) ? 0x0c0c0602u : 0x0c0c0400u;       // its code bytes
            const uint32_t o01 = __builtin_amdgcn_perm(mx[1][j][d], mx[0][j][d], selv);
            const uint32_t o23 = __builtin_amdgcn_perm(mx[3][j][d], mx[2][j][d], selv);
            const uint32_t c01 = __builtin_amdgcn_perm(cd[1][j][d], cd[0][j][d], selc);
            const uint32_t c23 = __builtin_amdgcn_perm(cd[3][j][d], cd[2][j][d], selc);
            const int64_t e = e0 + (int64_t)(j * 16 + r) * H2;
            if constexpr (PAIR) {
              if ((n & 1) == 0) {
                holdv[j * 4 + r] = make_uint2(o01, o23);
                holdc[j * 4 + r] = c01 | (c23 << 16);
              } else {            // e - 4 = the upper tile's first pooled row: 8 rows = 16 bytes of values, 8 selection bytes
                *reinterpret_cast<uint4*>(p.pool + e - 4) = make_uint4(holdv[j * 4 + r].x, holdv[j * 4 + r].y, o01, o23);
                if (!p.code_cl) *reinterpret_cast<uint2*>(p.code + e - 4) = make_uint2(holdc[j * 4 + r], c01 | (c23 << 16));
              }
            } else {
              *reinterpret_cast<uint2*>(p.pool + e) = make_uint2(o01, o23);
              if (!p.code_cl) *reinterpret_cast<uint32_t*>(p.code + e) = c01 | (c23 << 16);
            }
          }
      }
    }
    WS_STAMP(4)
  }
#undef WS_STAMP
  if (TM && p.dbg && blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 5; ++k) p.dbg[wave * 8 + k] = tsec[k];
    p.dbg[wave * 8 + 5] = cnt;
  }
}

template <int CI, int TH, int CO, int MK, int EP, bool TM = false>
int ws_launch_t(WsArgs p, hipStream_t s) {
  using G = WsGeo<CI, TH>;
  constexpr int WS_TH = TH, WS_PB = G::PB;
  p.tiles_h = (p.H + WS_TH - 1) / WS_TH;
  p.tiles_w = (p.W + WS_TW - 1) / WS_TW;
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
  constexpr int FM = TH / (4 / (CO / 32));
  const size_t lds = (size_t)WS_NBUF * WS_PB + (MK == 1 ? 4 * FM * 1024 : 0) + CO * 4;
  static bool granted = false;          // per instantiation; the first (eager / warm-up) launch does it, never a captured one
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws128_kernel<CI, TH, CO, MK, EP, TM>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return ASR_EUNSUPPORTED;
    granted = true;
  }
  const int per_cu = CI == 64 ? (int)asr_tuning("WS64_PER_CU", 2) : 1;
  const int64_t slots = (int64_t)cus * (per_cu > 0 ? per_cu : 1);         // 64 input channels: 144 registers of weights per wave, two workgroups per CU
  const int64_t items = EP == 2 ? nt / 2 : nt;
  const unsigned grid = (unsigned)(items < slots ? items : slots);
  hipLaunchKernelGGL((conv3x3_ws128_kernel<CI, TH, CO, MK, EP, TM>), dim3(grid), dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

}  // namespace

int asr_conv3x3_ws128_launch(const WsArgs& a, hipStream_t s) {
  if (a.Cout != 64 && a.Cout != 128) return ASR_EUNSUPPORTED;
  // 32-bit byte offsets inside the kernel
  if ((int64_t)a.B * a.H * a.W * 256 >= ((int64_t)1 << 32)) return ASR_EUNSUPPORTED;
  if (a.bits_in && (a.Cin != 128 || a.Cout != 128 || a.mask || a.pool || a.bits_out)) return ASR_EUNSUPPORTED;
  if (a.bits_out && (a.Cin != 64 || a.Cout != 128 || a.mask || a.pool || !a.relu)) return ASR_EUNSUPPORTED;
  if (a.Cin == 64) {               // conv.5 forward (64 -> 128) in one pass: 4-row tiles, two workgroups per CU
    if (a.Cout != 128 || a.mask || a.pool) return ASR_EUNSUPPORTED;
    if (a.bits_out) return ws_launch_t<64, 4, 128, 3, 0>(a, s);        // ... also writing the ReLU bit mask of its output
#ifdef ASR_TUNE_ABLATE
    if (const int64_t dbg = asr_tuning("WS_DBG", 0)) {
      WsArgs t = a;
      t.dbg = reinterpret_cast<long long*>(dbg);
      return ws_launch_t<64, 4, 128, 0, 0, true>(t, s);
    }
#endif
    return ws_launch_t<64, 4, 128, 0, 0>(a, s);
  }
  if (a.Cin != 128) return ASR_EUNSUPPORTED;
#ifdef ASR_TUNE_ABLATE
  // development builds only (ASR_HIPCC_EXTRA=-DASR_TUNE_ABLATE): per-section clock totals of workgroup 0 written through a device address
  // handed in as tuning value WS_DBG (tools/conv_ws_test.cpp).  Never in the shipped library: asr_hip/lib.py forwards ASR_* environment
  // variables to asr_set_tuning, and a stray value must not become a pointer the production kernel writes through.
  if (const int64_t dbg = asr_tuning("WS_DBG", 0)) {
    WsArgs t = a;
    t.dbg = reinterpret_cast<long long*>(dbg);
    if (a.pool && a.Cout == 128 && !a.mask && a.code && a.H % 8 == 0 && a.W % 16 == 0) return ws_launch_t<128, 8, 128, 0, 1, true>(t, s);
    if (!a.pool && a.Cout == 128 && a.mask) return ws_launch_t<128, 8, 128, 1, 0, true>(t, s);
    if (!a.pool && a.Cout == 64 && !a.mask) return ws_launch_t<128, 8, 64, 0, 0, true>(t, s);
  }
#endif
  if (a.pool) {
    if (a.Cout != 128 || a.mask || !a.code || a.H % 8 != 0 || a.W % 16 != 0) return ASR_EUNSUPPORTED;
    // vertical tile pairs: half the store instructions.  Single tiles where H % 16 == 8 -- and under tuning WS_PAIR = 0, the arm
    // tests/test_gpu_conv_ws.py holds the paired epilogue against
    if (a.H % 16 == 0 && asr_tuning("WS_PAIR", 1) != 0) return ws_launch_t<128, 8, 128, 0, 2>(a, s);
    return ws_launch_t<128, 8, 128, 0, 1>(a, s);
  }
  if (a.bits_in) return ws_launch_t<128, 8, 128, 2, 0>(a, s);           // conv.7's data gradient with conv.5's ReLU mask as bits
  if (a.Cout == 128) return a.mask ? ws_launch_t<128, 8, 128, 1, 0>(a, s) : ws_launch_t<128, 8, 128, 0, 0>(a, s);
  return a.mask ? ws_launch_t<128, 8, 64, 1, 0>(a, s) : ws_launch_t<128, 8, 64, 0, 0>(a, s);
}
