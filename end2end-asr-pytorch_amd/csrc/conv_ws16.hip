// 3x3 convolution with 128 input channels, bf16 NHWC, weight-stationary, second form: TWO independent 4-wave workgroups per CU, each wave
// holding the weights of 16 output channels (reference: models/asr/transformer.py:48-52 -- conv.7 forward + ReLU + MaxPool2d + the
// (B, T', C F') view of :74-76, conv.7's data gradient with conv.5's ReLU mask, conv.5's data gradient).
//
// conv_ws.hip (one workgroup per CU, one wave per SIMD with 32 channels' weights) contracts a tile at the matrix pipe's own rate --
// 16.4 cycles per MFMA -- but a lone wave per SIMD hides nothing: its 20 - 36 vector-memory instructions per tile block it ~150 cycles
// each wherever they are issued (profiles/r05_conv_ws_sections_*.txt), 40 % on top of the contraction.  Two waves per SIMD hide them
// behind each other's MFMAs, and 256 registers per wave hold 16 channels x 9 taps x 128 inputs = 36 MFMA operands (144 registers, 128 of
// them in the accumulation half of the file).  A 16-channel wave would read one pixel operand from LDS per MFMA -- exactly the LDS's
// bandwidth at the matrix pipe's rate -- so the loop walks the PATCH ROWS instead of the output rows: one 16-pixel row operand of the
// (column shift, k step) group feeds the MFMAs of all three vertical taps (output rows r, r - 1, r - 2): 6 reads for 12 MFMAs.
//   * a workgroup owns 64 output channels (Cout = 128: workgroups of even / odd parity take the two halves and walk the same tiles);
//     tiles of 4 x 16 pixels, the 6 x 18 x 128-channel halo patch (27 KB) of tile n + 1 by LDS-DMA under tile n (two buffers);
//   * LDS image, fragment column -> pixel and lane group -> channel chunk maps as in conv_ws.hip (conflict-free ds_read_b128 at every
//     column shift; the pooling partner of a lane is a row rotation by 8);
//   * epilogues from the accumulators: two tile rows exchange lane groups (v_permlane16_swap) so that a lane owns one 16-byte chunk
//     (8 channels) of one pixel; pooled form: 2 pooled rows of a (column, channel) as a 4-byte run + 2 selection bytes.
#include "common.h"
#include "conv_c64_core.h"
#include "conv_ws.h"

#include <utility>

namespace {

#define V_FENCE() asm volatile("" ::: "memory")

constexpr int V_TH = 4, V_TW = 16, V_PW = 18;
constexpr int V_NHALO = (V_TH + 2) * V_PW;      // 108 halo pixels
constexpr int V_PB = V_NHALO * 256;             // bytes of one patch buffer
constexpr int V_NCH = V_NHALO * 16;             // 16-byte chunks of a patch (1728)
constexpr int V_PIT = (V_NCH + 255) / 256;      // DMA instructions per thread and patch (7; the last one: 192 threads)
constexpr int V_NBUF = 2;
constexpr int V_PD = 6;                         // row operands in flight ahead of their MFMAs
constexpr int V_NQ = 72;                        // row steps per tile: 12 (column shift, k step) groups x 6 patch rows

template <typename F, int... I>
__device__ __forceinline__ void v_for(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}

template <int Q> struct VStep { static constexpr int G = Q / 6, r = Q % 6, dx = G / 4, ms = G % 4; };

template <int Q>
__device__ __forceinline__ void v_issue(u32x4_t& dst, const unsigned (&pbd)[3][4]) {
  using K = VStep<Q>;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(pbd[K::dx][K::ms]), "n"(K::r * V_PW * 256));
}

// row step Q: patch row r of group (dx, ms) is the B operand of output rows r (top tap), r - 1 (middle), r - 2 (bottom)
template <int Q>
__device__ __forceinline__ void v_step(f32x4_t (&acc)[4], u32x4_t (&a)[V_PD + 1], const u32x4_t (&wB)[9][4], const unsigned (&pbd)[3][4]) {
  using K = VStep<Q>;
  u32x4_t& cur = a[Q % (V_PD + 1)];
  if constexpr (Q + V_PD < V_NQ) v_issue<Q + V_PD>(a[(Q + V_PD) % (V_PD + 1)], pbd);
  constexpr int ahead = (Q + V_PD < V_NQ) ? V_PD : V_NQ - 1 - Q;      // reads that may stay in flight (they return in order)
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(cur) : "n"(ahead));
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    constexpr int r = K::r;
    const int i = r - dy;
    if (i >= 0 && i < 4)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wB[dy * 3 + K::dx][K::ms]), __builtin_bit_cast(bf16x8_t, cur),
                                                       acc[i], 0, 0, 0);
  }
}

// EP = 0: y (B, H, W, CO) NHWC, optional ReLU / mask.  EP = 1: ReLU + 2x2 max-pool + selection codes in the (B, W/2, CO, H/2) layout.
// TM (tuning WS_DBG): per-section shader-clock totals of workgroup 0, as in conv_ws.hip.
template <bool MASK, int EP, bool TM = false>
__global__ __launch_bounds__(256, 2) void conv3x3_ws16_kernel(WsArgs p) {
  constexpr int STASH = MASK ? 4 * 2 * 1024 : 0;            // per wave: 2 DMA pieces of mask chunks (one per pair of tile rows)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* stash = smem + V_NBUF * V_PB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  // fragment column -> pixel of the row segment, lane group -> channel chunk of a k step (conv_ws.hip)
  const int la = lr >> 2, pix = 8 * (la & 1) + 2 * (lr & 3) + (((la >> 1) ^ la) & 1);
  const int gs = ((g & 1) << 1) | (g >> 1);
  const int CO = p.Cout, nhalf = CO >> 6;
  const int nwg = gridDim.x;
  const int vid = (nwg % 8 == 0) ? (blockIdx.x % 8) * (nwg / 8) + blockIdx.x / 8 : blockIdx.x;      // consecutive tasks on one XCD
  const int half = vid % nhalf, lane_id = vid / nhalf, nlanes = nwg / nhalf;        // (nwg is a multiple of nhalf: launcher)
  const int cnt = lane_id < p.ntiles ? (p.ntiles - lane_id + nlanes - 1) / nlanes : 0;
  const int co_base = half * 64 + wave * 16;
  const unsigned char* X = reinterpret_cast<const unsigned char*>(p.x);

  // ---- weights: A operand of every MFMA, resident in registers for the whole kernel (32 of the 36 operands straight into the
  // accumulation half of the file, where the MFMA reads them directly)
  u32x4_t wB[9][4];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int ms = 0; ms < 4; ++ms) {
      const bf16_t* src = p.wk + ((int64_t)(co_base + lr) * 9 + tap) * 128 + (ms * 4 + gs) * 8;
      if (tap * 4 + ms < 32) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wB[tap][ms]) : "v"(src));
      else wB[tap][ms] = *reinterpret_cast<const u32x4_t*>(src);
    }
  const f32x4_t bq = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + co_base + 4 * g) : f32x4_t{0.f, 0.f, 0.f, 0.f};

  struct Org { int b, h0, w0; };
  auto origin = [&](int n) __attribute__((always_inline)) {
    int t = lane_id + n * nlanes;
    Org o;
    if (EP == 1) {        // row tiles fastest: the pieces of a (column, channel) run of the transposed output meet in one L2
      o.h0 = (t % p.tiles_h) * V_TH; t /= p.tiles_h;
      o.w0 = (t % p.tiles_w) * V_TW; o.b = t / p.tiles_w;
    } else {
      o.w0 = (t % p.tiles_w) * V_TW; t /= p.tiles_w;
      o.h0 = (t % p.tiles_h) * V_TH; o.b = t / p.tiles_h;
    }
    return o;
  };
  // halo patch of tile n -> buffer n % 2: chunk c = (halo pixel, slot) of the lane-linear LDS image takes source chunk
  // slot ^ (patch column & 15); pixels outside the image come from the zero page (the DMA cannot zero-fill)
  auto stage = [&](int n, int t_) __attribute__((always_inline)) {
    const Org o = origin(n);
    unsigned char* buf = smem + (n % V_NBUF) * V_PB;
    const unsigned base = (((unsigned)o.b * (unsigned)p.H + (unsigned)o.h0) * (unsigned)p.W + (unsigned)o.w0) * 256u;   // < 4 GB (launcher)
    const bool inside = o.h0 >= 1 && o.w0 >= 1 && o.h0 + V_TH + 1 <= p.H && o.w0 + V_TW + 1 <= p.W;
#pragma unroll
    for (int it = 0; it < V_PIT; ++it) {
      const int c = t_ + it * 256;
      if (it < V_PIT - 1 || c < V_NCH) {
        const int hp = c >> 4, py = hp / V_PW, px = hp - py * V_PW;
        const unsigned rel = (unsigned)(((py - 1) * p.W + px - 1) * 256 + (((c & 15) ^ (px & 15)) << 4));
        unsigned char* dst = buf + (it * 256 + (t_ & ~63)) * 16;       // wave-uniform; the DMA adds lane * 16
        const unsigned char* src = X + (base + rel);
        if (!inside) {
          const int gy = o.h0 + py - 1, gx = o.w0 + px - 1;
          if (!(gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)) src = reinterpret_cast<const unsigned char*>(&c64_zero_page);
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  if (cnt > 0) stage(0, tid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // every kernel argument the tile loop uses is consumed once HERE (see conv_ws.hip: the compiler's lgkmcnt(0) for its scalar loads)
  asm volatile("" ::"s"(p.y), "s"(p.pool), "s"(p.code), "s"(p.mask), "s"(p.H), "s"(p.W), "s"(p.relu), "s"(p.tiles_h), "s"(p.tiles_w));

  // per-lane operand addressing: LDS byte address of the lane's chunk of patch row 0 for column shift dx and k step ms
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned offk[3][4];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
      offk[dx][ms] = smem_base + (unsigned)((pix + dx) * 256) + (unsigned)(((ms * 4 + gs) ^ ((pix + dx) & 15)) << 4);
  const unsigned stash_addr = smem_base + (unsigned)(V_NBUF * V_PB + (wave * 2 * 64 + lane) * 16);
  // the lane's output chunk after the lane-group exchange: 8 channels from (g >> 1) * 8 of the pixel in tile row 2 pp + (g & 1)
  const int orow = g & 1, och = co_base + (g >> 1) * 8;

  long long tsec[5] = {0, 0, 0, 0, 0}, tlast = TM ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define V_STAMP(K) if (TM) { const long long now_ = (long long)__builtin_amdgcn_s_memtime(); tsec[K] += now_ - tlast; tlast = now_; }
  for (int n = 0; n < cnt; ++n) {
    V_FENCE();
    __builtin_amdgcn_s_barrier();       // patch n landed for every wave; everybody is done with tile n - 1 (its buffer is free)
    V_FENCE();
    V_STAMP(0)
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const Org og = origin(n);
    const int b = og.b, h0 = og.h0, w0 = og.w0;
    // byte offset of the lane's chunk of tile row pair 0 in y / the mask (both (B, H, W, CO)); clamped copies for the mask loads
    const int gx = w0 + pix;
    if (MASK) {                        // this tile's mask chunks -> the lane's private stash (pixels outside the image: clamped)
      const unsigned char* Mk = reinterpret_cast<const unsigned char*>(p.mask);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const int gy = min(h0 + 2 * pp + orow, p.H - 1), gxc = min(gx, p.W - 1);
        const unsigned moff = ((((unsigned)b * (unsigned)p.H + (unsigned)gy) * (unsigned)p.W + (unsigned)gxc) * (unsigned)CO + (unsigned)och) * 2u;
        unsigned char* dst = stash + ((wave * 2 + pp) * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Mk + moff),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
    if (n + 1 < cnt) stage(n + 1, tl);
    V_FENCE();
    V_STAMP(1)

    f32x4_t acc[4] = {bq, bq, bq, bq};            // every accumulator starts from the bias of its 4 output channels
    unsigned pbd[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int ms = 0; ms < 4; ++ms) pbd[dx][ms] = offk[dx][ms] + (unsigned)((n % V_NBUF) * V_PB);
    u32x4_t a[V_PD + 1];
    v_for(std::make_integer_sequence<int, V_PD>{}, [&](auto qc) __attribute__((always_inline)) { v_issue<decltype(qc)::value>(a[decltype(qc)::value], pbd); });
    v_for(std::make_integer_sequence<int, V_NQ>{}, [&](auto qc) __attribute__((always_inline)) { v_step<decltype(qc)::value>(acc, a, wB, pbd); });

    // patch n + 1 (and this tile's mask chunks) must have landed before the next barrier
    V_FENCE();
    V_STAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    V_STAMP(3)

    if constexpr (EP == 0) {
      // ---- NHWC epilogue: bf16 pairs, ReLU on the packed halves, two tile rows exchange lane groups -> one 16-byte chunk per lane and row pair
      u32x4_t mk[2];
      if (MASK) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(mk[0]) : "v"(stash_addr));
        asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(mk[1]) : "v"(stash_addr));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(mk[0]), "+v"(mk[1]));
      }
      const uint32_t floor2 = p.relu ? 0u : 0x80008000u;       // ReLU = packed signed max with 0, "no ReLU" = max with the most negative int16
      unsigned char* Y = reinterpret_cast<unsigned char*>(p.y);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          uint32_t pa = pack_bf16(acc[2 * pp][2 * d], acc[2 * pp][2 * d + 1]);
          uint32_t pb2 = pack_bf16(acc[2 * pp + 1][2 * d], acc[2 * pp + 1][2 * d + 1]);
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(pa) : "s"(floor2));
          asm("v_pk_max_i16 %0, %0, %1" : "+v"(pb2) : "s"(floor2));
          // (a, b) -> a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3}  (row = lane group)
          auto sw = __builtin_amdgcn_permlane16_swap(pa, pb2, false, false);
          lo[d] = sw[0]; hi[d] = sw[1];
        }
        uint4 o = make_uint4(lo[0], lo[1], hi[0], hi[1]);
        if (MASK) {
          o.x = c64_mask2(o.x, mk[pp][0]); o.y = c64_mask2(o.y, mk[pp][1]);
          o.z = c64_mask2(o.z, mk[pp][2]); o.w = c64_mask2(o.w, mk[pp][3]);
        }
        const int gy = h0 + 2 * pp + orow;
        if (gy < p.H && gx < p.W)
          *reinterpret_cast<uint4*>(Y + ((((size_t)b * p.H + gy) * p.W + gx) * CO + och) * 2) = o;
      }
    } else {
      // ---- pooled epilogue (launcher: H % 4 == 0, W % 16 == 0: every tile is whole).  Values are ReLU outputs (>= 0): the unsigned
      // 16-bit maximum of the bf16 bit patterns IS the bf16 maximum and equal numbers have equal bits.  Window scan order = (row 0: even,
      // odd column; row 1: even, odd): v0 .. v3 of the EVEN pixel's lane (the odd one computes don't-cares).
      const uint32_t one = 0x00010001u;
      uint32_t mx[2][2], cd[2][2];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          uint32_t v0 = pack_bf16(acc[2 * pr][2 * d], acc[2 * pr][2 * d + 1]);
          uint32_t v2 = pack_bf16(acc[2 * pr + 1][2 * d], acc[2 * pr + 1][2 * d + 1]);
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
          mx[pr][d] = m; cd[pr][d] = c;
        }
      if ((pix & 1) == 0) {
        const int H2 = p.H >> 1, W2 = p.W >> 1;
        const int64_t e0 = (((int64_t)b * W2 + (gx >> 1)) * 128 + co_base + 4 * g) * H2 + (h0 >> 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = r >> 1;
          const uint32_t selv = (r & 1) ? 0x07060302u : 0x05040100u;       // the channel's half of (second, first) pooled row
          const uint32_t selc = (r & 1) ? 0x0c0c0602u : 0x0c0c0400u;       // its code bytes
          const int64_t e = e0 + (int64_t)r * H2;
          *reinterpret_cast<uint32_t*>(p.pool + e) = __builtin_amdgcn_perm(mx[1][d], mx[0][d], selv);
          *reinterpret_cast<uint16_t*>(p.code + e) = (uint16_t)__builtin_amdgcn_perm(cd[1][d], cd[0][d], selc);
        }
      }
    }
    V_STAMP(4)
  }
#undef V_STAMP
  if (TM && p.dbg && blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 5; ++k) p.dbg[wave * 8 + k] = tsec[k];
    p.dbg[wave * 8 + 5] = cnt;
  }
}

template <bool MASK, int EP, bool TM = false>
int v_launch_t(WsArgs p, hipStream_t s) {
  p.tiles_h = (p.H + V_TH - 1) / V_TH;
  p.tiles_w = (p.W + V_TW - 1) / V_TW;
  const int64_t nt = (int64_t)p.B * p.tiles_h * p.tiles_w;
  if (nt >= ((int64_t)1 << 30)) return ASR_EUNSUPPORTED;
  p.ntiles = (int)nt;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  const size_t lds = (size_t)V_NBUF * V_PB + (MASK ? 4 * 2 * 1024 : 0);
  static bool granted = false;          // per instantiation; the first (eager / warm-up) launch does it, never a captured one
  if (!granted) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_ws16_kernel<MASK, EP, TM>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return ASR_EUNSUPPORTED;
    granted = true;
  }
  const int nhalf = p.Cout / 64;
  const int per_cu = (int)asr_tuning("WS16_PER_CU", 2);
  int64_t grid = (int64_t)cus * (per_cu > 0 ? per_cu : 2);            // two workgroups per CU (63 KB of LDS, 256 registers each)
  if (grid > nt * nhalf) grid = nt * nhalf;
  grid -= grid % nhalf;
  hipLaunchKernelGGL((conv3x3_ws16_kernel<MASK, EP, TM>), dim3((unsigned)grid), dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

}  // namespace

int asr_conv3x3_ws16_launch(const WsArgs& a, hipStream_t s) {
  if (a.Cout != 64 && a.Cout != 128) return ASR_EUNSUPPORTED;
  // 32-bit byte offsets inside the kernel
  if ((int64_t)a.B * a.H * a.W * 256 >= ((int64_t)1 << 32)) return ASR_EUNSUPPORTED;
  const bool pooled = a.pool != nullptr;
  if (pooled && (a.Cout != 128 || a.mask || !a.code || a.H % 4 != 0 || a.W % 16 != 0)) return ASR_EUNSUPPORTED;
  if (const int64_t dbg = asr_tuning("WS_DBG", 0)) {        // development: per-section clock totals of workgroup 0 (tools/conv_ws_test.cpp)
    WsArgs t = a;
    t.dbg = reinterpret_cast<long long*>(dbg);
    if (pooled) return v_launch_t<false, 1, true>(t, s);
    return a.mask ? v_launch_t<true, 0, true>(t, s) : v_launch_t<false, 0, true>(t, s);
  }
  if (pooled) return v_launch_t<false, 1>(a, s);
  return a.mask ? v_launch_t<true, 0>(a, s) : v_launch_t<false, 0>(a, s);
}
