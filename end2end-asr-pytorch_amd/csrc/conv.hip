// vgg_cnn front end (reference: models/asr/transformer.py:42-53 applied at :70-71, reshape :74-76).
// Activations are NHWC (B, H=F, W=T, C): the contraction axis of every 3x3 convolution (taps x channels) is then
// made of 9 shifted, channel-contiguous rows, i.e. an implicit GEMM whose A operand is read from ONE halo patch
// staged in LDS per workgroup.
//
//   asr_conv3x3_igemm : forward (+bias+ReLU) and dgrad (tap-flipped weights, ReLU mask of the consumer's input).
//                       MFMA-bound: 2*9*Cin*Cout flop per output pixel; HBM bytes per pixel = (Cin + Cout)*sizeof(T)
//                       (+ halo overlap 1.4x on the read side, served by L2).
//   asr_conv3x3_wgrad_nhwc : dW = dY^T . shift(X) over the B*H*W pixels straight from the NHWC tensors (transposing LDS
//                       reads build the pixel-major MFMA operands); per-workgroup partial dW blocks meet in a workspace.
//   conv1 / pooling kernels are HBM-bound streaming kernels.
#include <stdlib.h>

#include "common.h"
#include "conv1_wgrad_mfma.h"
#include "conv_c64.h"
#include "conv_ws.h"
#include "conv_wgrad_dma.h"

namespace {

// Ablation hooks (ASR_IGEMM_ABLATE / ASR_WGRAD_ABLATE) exist only in -DASR_TUNE_ABLATE builds: a run-time test inside the MFMA
// loops costs scalar branches per step and blocks unrolling.
#ifdef ASR_TUNE_ABLATE
#define ASR_ABL(P, BIT) (((P).ablate & (BIT)) != 0)
#else
#define ASR_ABL(P, BIT) false
#endif

// ================================================================================================ conv1 (Cin = 1)
// Direct convolution on the vector ALU, HBM bound (528 MB of bf16 activations written / read at B = 32).  A thread owns EPC
// output channels (taps + bias in registers) and walks QUADS of 4 horizontally adjacent pixels: the 3 x 6 input window of a
// quad is loaded once (4.5 instead of 9 input loads per pixel, bounds handled by clamped addresses + selects), channel
// pairs are packed fp32 (v_pk_fma_f32).  The C0/EPC threads of a quad are adjacent lanes, so a pixel's NHWC row is one
// contiguous C0*sizeof(T) run.
typedef __attribute__((ext_vector_type(2))) float asr_f32x2_t;
constexpr int C1_PW = 4;

__device__ __forceinline__ void conv1_window(const float* __restrict__ x, int64_t b, int yh, int x0, int H, int W,
                                             float (*in)[C1_PW + 2]) {
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = yh + ky - 1;
    const bool rowok = yy >= 0 && yy < H;
    const float* xr = x + (b * H + (rowok ? yy : yh)) * (int64_t)W;
#pragma unroll
    for (int k = 0; k < C1_PW + 2; ++k) {
      const int xx = x0 + k - 1;
      const bool ok = rowok && xx >= 0 && xx < W;
      const float v = xr[xx < 0 ? 0 : (xx < W ? xx : W - 1)];
      in[ky][k] = ok ? v : 0.f;
    }
  }
}

// The two halves of conv1_window: the clamped loads alone (issued early), and the zeroing of out-of-image taps (applied when the
// values are consumed) -- a select right behind the load would make the compiler wait for the load where it is issued.
__device__ __forceinline__ void conv1_window_load(const float* __restrict__ x, int64_t b, int yh, int x0, int H, int W,
                                                  float (*raw)[C1_PW + 2]) {
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = yh + ky - 1;
    const float* xr = x + (b * H + ((yy >= 0 && yy < H) ? yy : yh)) * (int64_t)W;
#pragma unroll
    for (int k = 0; k < C1_PW + 2; ++k) {
      const int xx = x0 + k - 1;
      raw[ky][k] = xr[xx < 0 ? 0 : (xx < W ? xx : W - 1)];
    }
  }
}
__device__ __forceinline__ void conv1_window_mask(int yh, int x0, int H, int W, const float (*raw)[C1_PW + 2], float (*in)[C1_PW + 2]) {
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int yy = yh + ky - 1;
    const bool rowok = yy >= 0 && yy < H;
#pragma unroll
    for (int k = 0; k < C1_PW + 2; ++k) {
      const int xx = x0 + k - 1;
      in[ky][k] = (rowok && xx >= 0 && xx < W) ? raw[ky][k] : 0.f;
    }
  }
}

// FULL: W is a multiple of the quad width, every quad stores exactly C1_PW chunks -- the compiler then KNOWS how many stores follow
// the prefetch loads and can wait for the loads alone (s_waitcnt vmcnt(C1_PW)); behind a conditional store it has to assume none.
template <typename T, bool FULL>
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, T* __restrict__ y, int B, int H,
                                                        int W, int C0) {
  constexpr int EPC = DT<T>::EPC, NP = EPC / 2;
  const int groups = C0 / EPC;                // 256 % groups == 0
  const int cg = threadIdx.x % groups;
  asr_f32x2_t wr[NP][9], br[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    br[j] = asr_f32x2_t{bias[cg * EPC + 2 * j], bias[cg * EPC + 2 * j + 1]};
#pragma unroll
    for (int t = 0; t < 9; ++t) wr[j][t] = asr_f32x2_t{w[(cg * EPC + 2 * j) * 9 + t], w[(cg * EPC + 2 * j + 1) * 9 + t]};
  }
  // a block walks whole image rows (b, yh); its threads cover the row's quads: no 64-bit division per work item (an emulated
  // int64 div/mod costs more instructions than the 288 FMAs of a quad)
  const int wq = (W + C1_PW - 1) / C1_PW;
  const int qpb = 256 / groups;
  // the weights have landed before the loops start: left pending, their first use INSIDE the quad loop carries an
  // s_waitcnt vmcnt(<prefetch loads>) that every iteration then pays by waiting for the previous iteration's stores
#pragma unroll
  for (int j = 0; j < NP; ++j) {             // (an empty asm that "reads" every weight register: the waits happen here)
    asm volatile("" ::"v"(br[j][0]), "v"(br[j][1]));
#pragma unroll
    for (int t = 0; t < 9; ++t) asm volatile("" ::"v"(wr[j][t][0]), "v"(wr[j][t][1]));
  }
  for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
   const int yh = row % H;
   const int64_t b = row / H;
   // The NEXT quad's window is loaded before this quad's stores are issued: loads and stores retire through one in-order counter
   // (vmcnt), so a window loaded AFTER the stores could only be waited for together with them -- every iteration then exposed a
   // full store round trip to HBM and the 60 us of arithmetic never overlapped the 114 us of stores.
   float in[3][C1_PW + 2], nxt[3][C1_PW + 2];
   int qx = threadIdx.x / groups;
   if (qx < wq) conv1_window(x, b, yh, qx * C1_PW, H, W, in);
   for (; qx < wq; qx += qpb) {
    const int x0 = qx * C1_PW;
    const int qn = min(qx + qpb, wq - 1);            // always issued (clamped): no branch between the prefetch and the stores
    conv1_window_load(x, b, yh, qn * C1_PW, H, W, nxt);
    __builtin_amdgcn_sched_barrier(0);               // (the scheduler otherwise sinks the loads below the first stores)
#pragma unroll
    for (int px = 0; px < C1_PW; ++px) {
      if (!FULL && x0 + px >= W) break;
      Chunk<T> o;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        asr_f32x2_t a = br[j];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v = in[ky][px + kx];
            a += wr[j][ky * 3 + kx] * asr_f32x2_t{v, v};
          }
        o.e[2 * j] = DT<T>::to(fmaxf(a[0], 0.f));
        o.e[2 * j + 1] = DT<T>::to(fmaxf(a[1], 0.f));
      }
      *reinterpret_cast<uint4*>(y + (((b * H + yh) * (int64_t)W) + x0 + px) * C0 + cg * EPC) = o.v;
    }
    __builtin_amdgcn_sched_barrier(0);               // consume the prefetch only here: C1_PW stores are younger, s_waitcnt vmcnt(C1_PW) suffices
    conv1_window_mask(yh, qn * C1_PW, H, W, nxt, in);
   }
  }
}

// dw[c][tap] += sum_px dy[px][c] * x[px + tap], db[c] += sum_px dy[px][c]: same quad walk, per-thread accumulators for
// its EPC channels, then LDS atomics per block and one global atomic per (block, element).
template <typename T>
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const T* __restrict__ dy,
                                                          float* dw, float* db, int B, int H, int W, int C0) {
  constexpr int EPC = DT<T>::EPC, NP = EPC / 2;
  extern __shared__ float sacc[];     // [C0*10]
  for (int i = threadIdx.x; i < C0 * 10; i += 256) sacc[i] = 0.f;
  __syncthreads();
  const int groups = C0 / EPC;        // 256 % groups == 0 -> a thread keeps its channel group across the loop
  const int cg = threadIdx.x % groups;
  asr_f32x2_t aw[NP][9], ab[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    ab[j] = asr_f32x2_t{0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) aw[j][t] = asr_f32x2_t{0.f, 0.f};
  }
  const int wq = (W + C1_PW - 1) / C1_PW;
  const int qpb = 256 / groups;
  for (int row = blockIdx.x; row < B * H; row += gridDim.x) {
   const int yh = row % H;
   const int64_t b = row / H;
   for (int qx = threadIdx.x / groups; qx < wq; qx += qpb) {
    const int x0 = qx * C1_PW;
    Chunk<T> d[C1_PW];
#pragma unroll
    for (int px = 0; px < C1_PW; ++px) {          // unconditional loads (clamped address + select): one round trip for all
      const int xx = x0 + px < W ? x0 + px : W - 1;
      const uint4 v = *reinterpret_cast<const uint4*>(dy + (((b * H + yh) * (int64_t)W) + xx) * C0 + cg * EPC);
      d[px].v = x0 + px < W ? v : make_uint4(0u, 0u, 0u, 0u);
    }
    float in[3][C1_PW + 2];
    conv1_window(x, b, yh, x0, H, W, in);
#pragma unroll
    for (int px = 0; px < C1_PW; ++px)
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const asr_f32x2_t g = asr_f32x2_t{DT<T>::from(d[px].e[2 * j]), DT<T>::from(d[px].e[2 * j + 1])};
        ab[j] += g;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const float v = in[ky][px + kx];
            aw[j][ky * 3 + kx] += g * asr_f32x2_t{v, v};
          }
      }
   }
  }
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = cg * EPC + 2 * j + h;
      atomicAdd(&sacc[C0 * 9 + c], ab[j][h]);
#pragma unroll
      for (int t = 0; t < 9; ++t) atomicAdd(&sacc[c * 9 + t], aw[j][t][h]);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < C0 * 10; i += 256) {
    if (i < C0 * 9) atomicAdd(dw + i, sacc[i]);
    else atomicAdd(db + (i - C0 * 9), sacc[i]);
  }
}

// ================================================================================================ weight packing
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wk, T* __restrict__ wd, int Cout, int Cin) {
  const int64_t total = (int64_t)Cout * Cin * 9;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 9);
    const int ci = (int)((i / 9) % Cin);
    const int co = (int)(i / (9 * (int64_t)Cin));
    const float v = w[i];
    if (wk) DT<T>::st(wk + ((int64_t)co * 9 + tap) * Cin + ci, v);
    if (wd) DT<T>::st(wd + ((int64_t)ci * 9 + (8 - tap)) * Cout + co, v);   // tap flip: (2-ky)*3+(2-kx) = 8 - tap
  }
}

// up to 8 weight tensors in one launch (the conv stack's three packs were three 5 us launches per step)
struct PackMulti {
  const float* w[8]; void* wk[8]; void* wd[8];
  int cout[8], cin[8];
  int64_t start[9];        // element ranges of the tensors in the launch's flat index
  int n;
};
template <typename T>
__global__ void pack_weight_multi_kernel(PackMulti a) {
  const int64_t total = a.start[a.n];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int k = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) k += (j < a.n && i >= a.start[j]) ? 1 : 0;
    const int64_t e = i - a.start[k];
    const int Cin = a.cin[k], Cout = a.cout[k];
    const int tap = (int)(e % 9);
    const int ci = (int)((e / 9) % Cin);
    const int co = (int)(e / (9 * (int64_t)Cin));
    const float v = a.w[k][e];
    T* wk = static_cast<T*>(a.wk[k]);
    T* wd = static_cast<T*>(a.wd[k]);
    if (wk) DT<T>::st(wk + ((int64_t)co * 9 + tap) * Cin + ci, v);
    if (wd) DT<T>::st(wd + ((int64_t)ci * 9 + (8 - tap)) * Cout + co, v);
  }
}

// ================================================================================================ implicit GEMM 3x3
__device__ const uint4 conv_zero_page = {0u, 0u, 0u, 0u};      // source of halo pixels outside the image

struct ConvArgs {
  const void* x; const void* wk; const float* bias; const void* mask_src; void* y;
  void* pool; uint8_t* code;    // PT kernels: (B, W/2, Cout, H/2) pooled output + its selection bytes instead of y
  int xcd_order;                // consecutive tiles on one XCD (tuning IGEMM_XCD, default 1)
  int B, H, W, Cin, Cout, relu, tiles_h, tiles_w;
  int ablate;   // tuning only (ASR_IGEMM_ABLATE in -DASR_TUNE_ABLATE builds): 1 = no patch loads, 2 = no weight loads, 4 = no stores, 8 = no MFMAs
};

// Workgroup tile = TH x 16 pixels x NCO output channels, K step = (tap, 64-channel slice).  The TH+2 x 18 halo patch of a
// channel slice is staged once and read at 9 shifted positions; the tap's weight rows are double buffered (register
// prefetch).  Waves: (TH/4) along pixel rows x WN along Cout, each wave 4 pixel-row fragments x FN = NCO/(16 WN) Cout
// fragments.  TH = 16 gives 4 x 4 (NCO 64) / 4 x 8 (NCO 128) fragments per wave: 2 / 2.7 MFMAs per LDS operand read
// instead of 1.3 / 2 with TH = 8.  What bounds the loop is instruction issue around the MFMAs (an MFMA leaves room for about two
// other vector instructions, tools/probes/mfma_valu_probe.hip; this loop carries 1.8 - 3.4) and the two barriers per tap.
template <typename T, int NCO, int TH, int TPS, int WBUF, bool PT = false>
__global__ __launch_bounds__(256) void conv3x3_igemm_kernel(ConvArgs p) {
  constexpr int EPC = DT<T>::EPC, ESZ = (int)sizeof(T);
  constexpr int CPP = 64 / EPC;            // 16-B chunks per 64-channel pixel slice
  // LDS rows (pixel slices / weight rows): unpadded rows of 64 channels, 16-B chunk c of row r in slot c ^ (r & 7) (conflict-free
  // operand reads; the image is lane-linear, so the halo patch can be filled by the LDS-DMA with the swizzle applied on the
  // source address).  fp32 rows are 256 B = 16 chunks: the XOR only permutes the low 3 bits of the chunk index.
  constexpr bool SWZ = true;
  constexpr int PP = 64 * ESZ;
#define ASR_SLOT(ROW, CH) ((SWZ ? ((CH) ^ ((ROW) & 7)) : (CH)) << 4)
  constexpr int NMS = 64 / (4 * EPC);      // macro steps per 64 channels
  constexpr int WM = TH / 4, WN = 4 / WM;  // wave grid
  constexpr int FN = NCO / (16 * WN);      // cout fragments per wave
  constexpr int WROWS = TPS * NCO;         // weight rows per step (TPS taps)
  constexpr int WCH = WROWS * CPP / 256;   // weight chunks per thread
  constexpr int SPS = (9 + TPS - 1) / TPS; // steps per 64-channel slice
  constexpr int NHALO = (TH + 2) * 18;     // halo pixels
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sP = smem;
  unsigned char* sW0 = smem + NHALO * PP;         // weight tile, buffer 0
  unsigned char* sW1 = WBUF == 2 ? sW0 + WROWS * PP : sW0;   // buffer 1 (WBUF == 1: one weight buffer, an extra barrier per step)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int wm = wave / WN, wn = wave % WN;
  int t = blockIdx.x, tw, th, b;
  if (PT || p.xcd_order) {
    // consecutive tile ids on ONE XCD (blockIdx is dealt round-robin over the 8 XCDs, each with its own L2): neighbouring tiles share
    // their halo rows / columns in that L2 (PMC, conv.7 forward: 167.5 -> 131.1 MB fetched)
    const int nwg = gridDim.x, xcd = t & 7, qn = nwg >> 3, rn = nwg & 7;
    t = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (t >> 3);
  }
  if constexpr (PT) {
    // pooled epilogue: the five row tiles of a column strip each write 16 bytes of every 80-byte (column, channel) run of the encoder
    // layout.  Row tile fastest: the pieces of a cache line then meet in one L2 before it is written back, instead of five partial
    // write-backs from five L2s (PMC WRITE_SIZE 256 000 -> 100 414 KB).
    th = t % p.tiles_h; t /= p.tiles_h;
    tw = t % p.tiles_w;
    b = t / p.tiles_w;
  } else {
    tw = t % p.tiles_w; t /= p.tiles_w;
    th = t % p.tiles_h;
    b = t / p.tiles_h;
  }
  const int h0 = th * TH, w0 = tw * 16;
  const T* X = static_cast<const T*>(p.x);
  const T* Wk = static_cast<const T*>(p.wk);
  const int nchunk = p.Cin / 64;
  const int nsteps = nchunk * SPS;

  f32x4_t acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // weight tile (TPS taps, 64-channel slice): global -> registers -> LDS; the register copy lives for one iteration only.
  // Row tt * NCO + co of the tile holds tap (first tap of the step + tt); a step past tap 8 loads nothing for that row.
#define ASR_WLOAD(RW, STEP)                                                                                   \
  {                                                                                                           \
    const int cc_ = (STEP) / SPS, tap0_ = ((STEP) % SPS) * TPS;                                               \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                                         \
      const int c = tid + i * 256, row = c / CPP, ch = c % CPP, tt = row / NCO, co = row % NCO;               \
      if (TPS == 1 || tap0_ + tt < 9)                                                                         \
        RW[i] = *reinterpret_cast<const u32x4_t*>(Wk + ((int64_t)co * 9 + tap0_ + tt) * p.Cin + cc_ * 64 + ch * EPC); \
    }                                                                                                         \
  }
#define ASR_WWRITE(RW, DST)                                                                                   \
  {                                                                                                           \
    _Pragma("unroll") for (int i = 0; i < WCH; ++i) {                                                         \
      const int c = tid + i * 256, row = c / CPP, ch = c % CPP;                                               \
      *reinterpret_cast<u32x4_t*>((DST) + row * PP + ASR_SLOT(row, ch)) = RW[i];                                        \
    }                                                                                                         \
  }
  // halo patch of one 64-channel slice, HBM -> LDS by the LDS-DMA (no registers, one round trip): chunk c = (pixel hp, slot) of the
  // lane-linear image takes source chunk slot ^ (hp & 7); pixels outside the image are read from a 16-byte zero page.
  constexpr int PIT = (NHALO * CPP + 255) / 256;
  auto pstage = [&](int cc) __attribute__((always_inline)) {
#pragma unroll 1                         // rolled on purpose: a DMA is fire-and-forget, unrolling only pins 2 address registers per pass
    for (int it = 0; it < PIT; ++it) {
      const int c = tid + it * 256;
      if (c < NHALO * CPP) {
        const int hp = c / CPP, slot = c % CPP, ch = SWZ ? (slot ^ (hp & 7)) : slot;
        const int gy = h0 + hp / 18 - 1, gx = w0 + hp % 18 - 1;
        const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        const T* src = in ? X + (((int64_t)b * p.H + gy) * p.W + gx) * p.Cin + cc * 64 + ch * EPC
                          : reinterpret_cast<const T*>(&conv_zero_page);       // the DMA cannot zero-fill: outside pixels read zeros
        unsigned char* dst = sP + (it * 256 + (tid & ~63)) * 16;      // wave-uniform; the DMA adds lane * 16
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };
  {
    u32x4_t rw0[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) rw0[i] = u32x4_t{0u, 0u, 0u, 0u};
    ASR_WLOAD(rw0, 0)
    ASR_WWRITE(rw0, sW0)
  }
#pragma unroll 1
  for (int step = 0; step < nsteps; ++step) {
    const int sstep = step % SPS;
    if (sstep == 0) {
      if (step > 0) __syncthreads();      // everybody is done with the previous channel slice of the patch
      if (!ASR_ABL(p, 1)) pstage(step / SPS);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                     // patch + weight buffer (step&1) visible
    }
    const bool has_next = step + 1 < nsteps;
    u32x4_t rw[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) rw[i] = u32x4_t{0u, 0u, 0u, 0u};
    if (has_next && !ASR_ABL(p, 2)) ASR_WLOAD(rw, step + 1)
    const unsigned char* sW = (step & 1) ? sW1 : sW0;
#pragma unroll
    for (int tt = 0; tt < TPS; ++tt) {
      const int tap = sstep * TPS + tt;
      if (TPS > 1 && tap >= 9) break;
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int ms = 0; ms < NMS; ++ms) {
        uint4 a[4], bfr[FN];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int hp = (wm * 4 + i + dy) * 18 + lr + dx;
          a[i] = *reinterpret_cast<const uint4*>(sP + hp * PP + ASR_SLOT(hp, ms * 4 + g));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int row = tt * NCO + wn * (NCO / WN) + j * 16 + lr;
          bfr[j] = *reinterpret_cast<const uint4*>(sW + row * PP + ASR_SLOT(row, ms * 4 + g));
        }
        if (!ASR_ABL(p, 8)) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) mma16<T>(acc[i][j], bfr[j], a[i]);   // D = (co rows) x (pixel columns)
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(a[i].x));
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" :: "v"(bfr[j].x));
        }
      }
    }
    if (has_next) {
      unsigned char* dst = (step & 1) ? sW0 : sW1;
      if (WBUF == 1) __syncthreads();       // single buffer: everybody must be done reading this step's weights first
      ASR_WWRITE(rw, dst)
      // the next step either re-stages the patch (last step of a slice -> barrier pair above) or needs this barrier
      if (sstep != SPS - 1) __syncthreads();
    }
  }

#undef ASR_WLOAD
#undef ASR_WWRITE
#undef ASR_SLOT

  if constexpr (PT) {
    // ---- pooled epilogue (conv.7 + ReLU + MaxPool2d + the view / transpose of transformer.py:50-52,74-76): bias + ReLU on the
    // accumulators, 2 x 2 maximum over the wave's row pairs (registers) and the lane pairs (lr, lr ^ 1: DPP), one selection byte per
    // pooled element (packed 16-bit arithmetic on the bf16 bit patterns, as in conv_c64.hip), the 8 x 8 x 128 pooled tile and its
    // bytes staged in LDS as [pooled column][channel][pooled row] so that the (B, W/2, C, H/2) output gets 16-byte runs along H/2.
    // The un-pooled output is never stored.  Launcher: bf16, 128 outputs, H and W multiples of 16.
    static_assert(!PT || (sizeof(T) == 2 && NCO == 128 && TH == 16 && WM == 4), "pooled epilogue: bf16, 128 channels, 16-row tile");
    __syncthreads();                       // every wave is done with the operand tiles
    bf16_t* sPool = reinterpret_cast<bf16_t*>(smem);
    uint8_t* sCode = smem + 8 * 128 * 8 * 2;
    const bool odd = (lr & 1) != 0;
    const uint32_t one = 0x00010001u;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x4_t bvj = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + j * 16 + 4 * g) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        uint32_t mx[2], cd[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const f32x4_t x0 = acc[2 * pr][j] + bvj, x1 = acc[2 * pr + 1][j] + bvj;
          const uint32_t mine0 = (uint32_t)f32_to_bf16(fmaxf(x0[2 * d], 0.f)) | ((uint32_t)f32_to_bf16(fmaxf(x0[2 * d + 1], 0.f)) << 16);
          const uint32_t mine1 = (uint32_t)f32_to_bf16(fmaxf(x1[2 * d], 0.f)) | ((uint32_t)f32_to_bf16(fmaxf(x1[2 * d + 1], 0.f)) << 16);
          const uint32_t oth0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine0, 0xB1, 0xf, 0xf, true);      // lane ^ 1
          const uint32_t oth1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine1, 0xB1, 0xf, 0xf, true);
          const uint32_t v0 = odd ? oth0 : mine0, v1 = odd ? mine0 : oth0, v2 = odd ? oth1 : mine1, v3 = odd ? mine1 : oth1;
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
          uint32_t c = one + n0 + n01 + n012;
          asm("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(c) : "v"(nz));
          mx[d] = m; cd[d] = c;
        }
        if (!odd) {
          const int base = (((lr >> 1) * 128 + j * 16 + 4 * g) * 8) + wave * 2 + pr;       // [pooled column][channel][pooled row]
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            sPool[base + (2 * d) * 8] = (bf16_t)(mx[d] & 0xffffu);
            sPool[base + (2 * d + 1) * 8] = (bf16_t)(mx[d] >> 16);
            sCode[base + (2 * d) * 8] = (uint8_t)(cd[d] & 0xffu);
            sCode[base + (2 * d + 1) * 8] = (uint8_t)((cd[d] >> 16) & 0xffu);
          }
        }
      }
    __syncthreads();
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    bf16_t* out = static_cast<bf16_t*>(p.pool);
    for (int c = tid; c < 8 * 128; c += 256) {
      const int owl = c >> 7, ch = c & 127;
      const int64_t gi = (((int64_t)b * W2 + (w0 >> 1) + owl) * 128 + ch) * H2 + (h0 >> 1);
      *reinterpret_cast<uint4*>(out + gi) = *reinterpret_cast<const uint4*>(sPool + c * 8);
      *reinterpret_cast<uint2*>(p.code + gi) = *reinterpret_cast<const uint2*>(sCode + c * 8);
    }
    return;
  }
  // ---- epilogue straight from the accumulators (operands were swapped: a fragment is (16 co rows) x (16 pixel columns), so a
  // lane holds 4 consecutive output channels of ONE pixel).  bias / ReLU in fp32, then the storage dtype; bf16 lanes exchange
  // halves with the neighbouring lane group (v_permlane16_swap) so that every lane owns one aligned 16-byte chunk of a pixel's
  // NHWC row: no LDS staging, no barrier, 64 contiguous bytes per pixel per store instruction.
  T* Y = static_cast<T*>(p.y);
  const T* Msk = static_cast<const T*>(p.mask_src);
  const int gx = w0 + lr;
  constexpr int NJ = ESZ == 2 ? FN / 2 : FN;        // chunks per pixel fragment and lane
  // channel of the lane's chunk q: bf16 pairs fragments (2q, 2q+1) -- even lane groups end up with 8 channels of 2q, odd ones of 2q+1
  int cho[NJ];
#pragma unroll
  for (int q = 0; q < NJ; ++q)
    cho[q] = wn * (NCO / WN) + (ESZ == 2 ? (2 * q + (g & 1)) * 16 + 4 * (g & 2) : q * 16 + 4 * g);
  u32x4_t mk[4][NJ];
  if (Msk) {                                        // all mask chunks first (clamped addresses): one memory round trip
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gy = min(h0 + wm * 4 + i, p.H - 1);
      const T* mrow = Msk + (((int64_t)b * p.H + gy) * p.W + min(gx, p.W - 1)) * p.Cout;
#pragma unroll
      for (int q = 0; q < NJ; ++q) mk[i][q] = *reinterpret_cast<const u32x4_t*>(mrow + cho[q]);
    }
  }
  f32x4_t bv[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
    bv[j] = p.bias ? *reinterpret_cast<const f32x4_t*>(p.bias + wn * (NCO / WN) + j * 16 + 4 * g) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gy = h0 + wm * 4 + i;
    const bool ok = gy < p.H && gx < p.W;
    T* yrow = Y + (((int64_t)b * p.H + (ok ? gy : 0)) * p.W + (ok ? gx : 0)) * p.Cout;
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
      Chunk<T> o;
      if constexpr (ESZ == 2) {
        uint32_t lo[2], hi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          f32x4_t xa = acc[i][2 * q] + bv[2 * q], xb = acc[i][2 * q + 1] + bv[2 * q + 1];
          if (p.relu) {
            xa[2 * d] = fmaxf(xa[2 * d], 0.f); xa[2 * d + 1] = fmaxf(xa[2 * d + 1], 0.f);
            xb[2 * d] = fmaxf(xb[2 * d], 0.f); xb[2 * d + 1] = fmaxf(xb[2 * d + 1], 0.f);
          }
          const uint32_t pa = (uint32_t)DT<T>::to(xa[2 * d]) | ((uint32_t)DT<T>::to(xa[2 * d + 1]) << 16);
          const uint32_t pb = (uint32_t)DT<T>::to(xb[2 * d]) | ((uint32_t)DT<T>::to(xb[2 * d + 1]) << 16);
          // (a, b) -> a' = {a.row0, b.row0, a.row2, b.row2}, b' = {a.row1, b.row1, a.row3, b.row3}
          auto sw = __builtin_amdgcn_permlane16_swap(pa, pb, false, false);
          lo[d] = sw[0]; hi[d] = sw[1];
        }
        o.v = make_uint4(lo[0], lo[1], hi[0], hi[1]);
      } else {
        f32x4_t x = acc[i][q] + bv[q];
        if (p.relu) { x[0] = fmaxf(x[0], 0.f); x[1] = fmaxf(x[1], 0.f); x[2] = fmaxf(x[2], 0.f); x[3] = fmaxf(x[3], 0.f); }
        o.v = __builtin_bit_cast(uint4, x);
      }
      if (Msk) {
        Chunk<T> m;
        m.v = __builtin_bit_cast(uint4, mk[i][q]);
#pragma unroll
        for (int e = 0; e < EPC; ++e)
          if (!(DT<T>::from(m.e[e]) > 0.f)) o.e[e] = DT<T>::to(0.f);
      }
      if (ok && !ASR_ABL(p, 4)) *reinterpret_cast<uint4*>(yrow + cho[q]) = o.v;
    }
  }
}

// ================================================================================================ max pooling
// grid.y = pooled rows (b, oh), grid.x * 256 threads = (ow, 16-byte channel group) items of a row
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W2 * groups) return;
  const int cg = t % groups, ow = t / groups;
  const int oh = blockIdx.y % H2;
  const int64_t b = blockIdx.y / H2;
  const T* base = x + (((b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
  Chunk<T> a, bq, c, d, o;
  a.v = *reinterpret_cast<const uint4*>(base);
  bq.v = *reinterpret_cast<const uint4*>(base + C);
  c.v = *reinterpret_cast<const uint4*>(base + (int64_t)W * C);
  d.v = *reinterpret_cast<const uint4*>(base + (int64_t)W * C + C);
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    const float m = fmaxf(fmaxf(DT<T>::from(a.e[j]), DT<T>::from(bq.e[j])), fmaxf(DT<T>::from(c.e[j]), DT<T>::from(d.e[j])));
    o.e[j] = DT<T>::to(m);
  }
  *reinterpret_cast<uint4*>(y + (((b * H2 + oh) * (int64_t)W2 + ow) * C) + cg * EPC) = o.v;
}
// block per (b, ow): pooled (H2, C) slab -> LDS -> written as (C, H2) i.e. feature index c*H2 + oh (transformer.py:74-76)
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_tcf_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  extern __shared__ float sp[];     // [H2][C+1]
  const int H2 = H / 2, W2 = W / 2;
  const int ow = blockIdx.x % W2, b = blockIdx.x / W2;
  for (int i = threadIdx.x; i < H2 * C; i += 256) {
    const int c = i % C, oh = i / C;
    const T* base = x + ((((int64_t)b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + c;
    const float m = fmaxf(fmaxf(DT<T>::ld(base), DT<T>::ld(base + C)),
                          fmaxf(DT<T>::ld(base + (int64_t)W * C), DT<T>::ld(base + (int64_t)W * C + C)));
    sp[oh * (C + 1) + c] = m;
  }
  __syncthreads();
  T* out = y + ((int64_t)b * W2 + ow) * (int64_t)C * H2;
  for (int i = threadIdx.x; i < H2 * C; i += 256) {
    const int oh = i % H2, c = i / H2;
    DT<T>::st(out + i, sp[oh * (C + 1) + c]);
  }
}
// The same with 16-byte accesses on both sides (C and H2 multiples of the chunk): a thread pools EPC channels of one window row
// from four 16-byte loads, and writes EPC consecutive features c*H2 + oh .. oh + EPC - 1 (gathered from LDS) as one 16-byte store.
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_tcf_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float sp[];     // [H2][C+1]
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC;
  const int ow = blockIdx.x % W2, b = blockIdx.x / W2;
  for (int i = threadIdx.x; i < H2 * groups; i += 256) {
    const int cg = i % groups, oh = i / groups;
    const T* base = x + ((((int64_t)b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
    Chunk<T> a, bb, c, d;
    a.v = *reinterpret_cast<const uint4*>(base);
    bb.v = *reinterpret_cast<const uint4*>(base + C);
    c.v = *reinterpret_cast<const uint4*>(base + (int64_t)W * C);
    d.v = *reinterpret_cast<const uint4*>(base + (int64_t)W * C + C);
#pragma unroll
    for (int j = 0; j < EPC; ++j)
      sp[oh * (C + 1) + cg * EPC + j] = fmaxf(fmaxf(DT<T>::from(a.e[j]), DT<T>::from(bb.e[j])), fmaxf(DT<T>::from(c.e[j]), DT<T>::from(d.e[j])));
  }
  __syncthreads();
  T* out = y + ((int64_t)b * W2 + ow) * (int64_t)C * H2;
  const int hg = H2 / EPC;
  for (int i = threadIdx.x; i < C * hg; i += 256) {
    const int c = i / hg, oh0 = (i % hg) * EPC;
    Chunk<T> o;
#pragma unroll
    for (int j = 0; j < EPC; ++j) o.e[j] = DT<T>::to(sp[(oh0 + j) * (C + 1) + c]);
    *reinterpret_cast<uint4*>(out + c * H2 + oh0) = o.v;
  }
}
// dx for one 2x2 window: gradient goes to the FIRST maximum in scan order (PyTorch max_pool2d), times ReLU'(x)
template <typename T>
__device__ __forceinline__ void pool_bwd_window(const T* __restrict__ x, T* __restrict__ dx, int64_t base, int C, int W, float gy) {
  const float v0 = DT<T>::ld(x + base), v1 = DT<T>::ld(x + base + C);
  const float v2 = DT<T>::ld(x + base + (int64_t)W * C), v3 = DT<T>::ld(x + base + (int64_t)W * C + C);
  int arg = 0; float m = v0;
  if (v1 > m) { m = v1; arg = 1; }
  if (v2 > m) { m = v2; arg = 2; }
  if (v3 > m) { m = v3; arg = 3; }
  const float gr = m > 0.f ? gy : 0.f;
  DT<T>::st(dx + base, arg == 0 ? gr : 0.f);
  DT<T>::st(dx + base + C, arg == 1 ? gr : 0.f);
  DT<T>::st(dx + base + (int64_t)W * C, arg == 2 ? gr : 0.f);
  DT<T>::st(dx + base + (int64_t)W * C + C, arg == 3 ? gr : 0.f);
}
// NHWC backward, one thread = EPC channels (16 bytes) of one 2x2 window: 5 vector loads, 4 vector stores.
// grid.y = pooled rows (b, oh), grid.x * 256 threads = (ow, channel group) items of a row.
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_nhwc_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                            int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W2 * groups) return;
  const int cg = t % groups, ow = t / groups;
  const int oh = blockIdx.y % H2;
  const int64_t b = blockIdx.y / H2;
  const int64_t rowp = (int64_t)W * C;
  const int64_t base = (((b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
  Chunk<T> v0, v1, v2, v3, g, o0, o1, o2, o3;
  v0.v = *reinterpret_cast<const uint4*>(x + base);
  v1.v = *reinterpret_cast<const uint4*>(x + base + C);
  v2.v = *reinterpret_cast<const uint4*>(x + base + rowp);
  v3.v = *reinterpret_cast<const uint4*>(x + base + rowp + C);
  g.v = *reinterpret_cast<const uint4*>(dy + (((b * H2 + oh) * (int64_t)W2 + ow) * C) + cg * EPC);
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    const float a0 = DT<T>::from(v0.e[j]), a1 = DT<T>::from(v1.e[j]), a2 = DT<T>::from(v2.e[j]), a3 = DT<T>::from(v3.e[j]);
    int arg = 0; float m = a0;                      // the FIRST maximum in scan order takes the gradient (PyTorch max_pool2d)
    if (a1 > m) { m = a1; arg = 1; }
    if (a2 > m) { m = a2; arg = 2; }
    if (a3 > m) { m = a3; arg = 3; }
    const T gr = m > 0.f ? g.e[j] : DT<T>::to(0.f); // times ReLU'(x)
    const T zero = DT<T>::to(0.f);
    o0.e[j] = arg == 0 ? gr : zero;
    o1.e[j] = arg == 1 ? gr : zero;
    o2.e[j] = arg == 2 ? gr : zero;
    o3.e[j] = arg == 3 ? gr : zero;
  }
  *reinterpret_cast<uint4*>(dx + base) = o0.v;
  *reinterpret_cast<uint4*>(dx + base + C) = o1.v;
  *reinterpret_cast<uint4*>(dx + base + rowp) = o2.v;
  *reinterpret_cast<uint4*>(dx + base + rowp + C) = o3.v;
}
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_tcf_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                           int B, int H, int W, int C) {
  extern __shared__ float sp[];     // [H2][C+1]
  const int H2 = H / 2, W2 = W / 2;
  const int ow = blockIdx.x % W2, b = blockIdx.x / W2;
  const T* in = dy + ((int64_t)b * W2 + ow) * (int64_t)C * H2;
  for (int i = threadIdx.x; i < H2 * C; i += 256) {
    const int oh = i % H2, c = i / H2;
    sp[oh * (C + 1) + c] = DT<T>::ld(in + i);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < H2 * C; i += 256) {
    const int c = i % C, oh = i / C;
    pool_bwd_window<T>(x, dx, ((((int64_t)b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + c, C, W, sp[oh * (C + 1) + c]);
  }
}
// The same with 16-byte accesses: dy (C, H2) comes in as chunks of EPC consecutive oh of one channel, the 2x2 windows go out as in
// pool_bwd_nhwc_kernel (EPC channels per thread: four 16-byte loads, four 16-byte stores).
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_tcf_vec_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                               int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float sp[];     // [H2][C+1]
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC, hg = H2 / EPC;
  const int ow = blockIdx.x % W2, b = blockIdx.x / W2;
  const T* in = dy + ((int64_t)b * W2 + ow) * (int64_t)C * H2;
  for (int i = threadIdx.x; i < C * hg; i += 256) {
    const int c = i / hg, oh0 = (i % hg) * EPC;
    Chunk<T> g;
    g.v = *reinterpret_cast<const uint4*>(in + c * H2 + oh0);
#pragma unroll
    for (int j = 0; j < EPC; ++j) sp[(oh0 + j) * (C + 1) + c] = DT<T>::from(g.e[j]);
  }
  __syncthreads();
  const int64_t rowp = (int64_t)W * C;
  for (int i = threadIdx.x; i < H2 * groups; i += 256) {
    const int cg = i % groups, oh = i / groups;
    const int64_t base = ((((int64_t)b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
    Chunk<T> v0, v1, v2, v3, o0, o1, o2, o3;
    v0.v = *reinterpret_cast<const uint4*>(x + base);
    v1.v = *reinterpret_cast<const uint4*>(x + base + C);
    v2.v = *reinterpret_cast<const uint4*>(x + base + rowp);
    v3.v = *reinterpret_cast<const uint4*>(x + base + rowp + C);
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      const float a0 = DT<T>::from(v0.e[j]), a1 = DT<T>::from(v1.e[j]), a2 = DT<T>::from(v2.e[j]), a3 = DT<T>::from(v3.e[j]);
      int arg = 0; float m = a0;                      // the FIRST maximum in scan order takes the gradient (PyTorch max_pool2d)
      if (a1 > m) { m = a1; arg = 1; }
      if (a2 > m) { m = a2; arg = 2; }
      if (a3 > m) { m = a3; arg = 3; }
      const T gr = DT<T>::to(m > 0.f ? sp[oh * (C + 1) + cg * EPC + j] : 0.f);       // times ReLU'(x)
      const T zero = DT<T>::to(0.f);
      o0.e[j] = arg == 0 ? gr : zero;
      o1.e[j] = arg == 1 ? gr : zero;
      o2.e[j] = arg == 2 ? gr : zero;
      o3.e[j] = arg == 3 ? gr : zero;
    }
    *reinterpret_cast<uint4*>(dx + base) = o0.v;
    *reinterpret_cast<uint4*>(dx + base + C) = o1.v;
    *reinterpret_cast<uint4*>(dx + base + rowp) = o2.v;
    *reinterpret_cast<uint4*>(dx + base + rowp + C) = o3.v;
  }
}
// ---- pooling with a selection code.  The backward kernels above find the arg max again from the pre-pool activations: for the
// first pool of the VGG front end that is 527 MB read back per step (and the only reason conv.2's full-resolution output is
// stored at all).  The *_code forms write one byte per POOLED element next to it -- 0: the maximum is <= 0 (ReLU'(x) = 0, no
// gradient), 1 + k: gradient to window position k in scan order (the FIRST maximum, PyTorch max_pool2d) -- and the backward reads
// dy and the codes only.  Code layout = layout of the pooled tensor.
__device__ __forceinline__ uint32_t pool_code(float a0, float a1, float a2, float a3, float* mx) {
  int arg = 0; float m = a0;
  if (a1 > m) { m = a1; arg = 1; }
  if (a2 > m) { m = a2; arg = 2; }
  if (a3 > m) { m = a3; arg = 3; }
  *mx = m;
  return m > 0.f ? (uint32_t)(1 + arg) : 0u;
}
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_tcf_code_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ code,
                                                                int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float sp[];     // [H2][C+1] maxima, then [H2][C+4] code bytes
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC;
  uint8_t* sc = reinterpret_cast<uint8_t*>(sp + H2 * (C + 1));
  const int ow = blockIdx.x % W2, b = blockIdx.x / W2;
  for (int i = threadIdx.x; i < H2 * groups; i += 256) {
    const int cg = i % groups, oh = i / groups;
    const T* base = x + ((((int64_t)b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
    Chunk<T> a, bb, c, d;
    a.v = *reinterpret_cast<const uint4*>(base);
    bb.v = *reinterpret_cast<const uint4*>(base + C);
    c.v = *reinterpret_cast<const uint4*>(base + (int64_t)W * C);
    d.v = *reinterpret_cast<const uint4*>(base + (int64_t)W * C + C);
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      float m;
      const uint32_t k = pool_code(DT<T>::from(a.e[j]), DT<T>::from(bb.e[j]), DT<T>::from(c.e[j]), DT<T>::from(d.e[j]), &m);
      sp[oh * (C + 1) + cg * EPC + j] = m;
      sc[oh * (C + 4) + cg * EPC + j] = (uint8_t)k;
    }
  }
  __syncthreads();
  const int64_t o0 = ((int64_t)b * W2 + ow) * (int64_t)C * H2;
  const int hg = H2 / EPC;
  for (int i = threadIdx.x; i < C * hg; i += 256) {
    const int c = i / hg, oh0 = (i % hg) * EPC;
    Chunk<T> o;
    union { uint8_t b[EPC]; uint32_t w[EPC / 4]; } k;
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      o.e[j] = DT<T>::to(sp[(oh0 + j) * (C + 1) + c]);
      k.b[j] = sc[(oh0 + j) * (C + 4) + c];
    }
    *reinterpret_cast<uint4*>(y + o0 + c * H2 + oh0) = o.v;
#pragma unroll
    for (int w = 0; w < EPC / 4; ++w) *reinterpret_cast<uint32_t*>(code + o0 + c * H2 + oh0 + 4 * w) = k.w[w];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_tcf_code_kernel(const uint8_t* __restrict__ code, const T* __restrict__ dy, T* __restrict__ dx,
                                                                int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  extern __shared__ float sp[];     // [H2][C+1] gradients, then [H2][C+4] code bytes
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC, hg = H2 / EPC;
  uint8_t* sc = reinterpret_cast<uint8_t*>(sp + H2 * (C + 1));
  const int ow = blockIdx.x % W2, b = blockIdx.x / W2;
  const int64_t o0 = ((int64_t)b * W2 + ow) * (int64_t)C * H2;
  for (int i = threadIdx.x; i < C * hg; i += 256) {
    const int c = i / hg, oh0 = (i % hg) * EPC;
    Chunk<T> g;
    g.v = *reinterpret_cast<const uint4*>(dy + o0 + c * H2 + oh0);
    union { uint8_t b[EPC]; uint32_t w[EPC / 4]; } k;
#pragma unroll
    for (int w = 0; w < EPC / 4; ++w) k.w[w] = *reinterpret_cast<const uint32_t*>(code + o0 + c * H2 + oh0 + 4 * w);
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      sp[(oh0 + j) * (C + 1) + c] = DT<T>::from(g.e[j]);
      sc[(oh0 + j) * (C + 4) + c] = k.b[j];
    }
  }
  __syncthreads();
  const int64_t rowp = (int64_t)W * C;
  for (int i = threadIdx.x; i < H2 * groups; i += 256) {
    const int cg = i % groups, oh = i / groups;
    const int64_t base = ((((int64_t)b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
    Chunk<T> o0v, o1v, o2v, o3v;
    const T zero = DT<T>::to(0.f);
#pragma unroll
    for (int j = 0; j < EPC; ++j) {
      const T gr = DT<T>::to(sp[oh * (C + 1) + cg * EPC + j]);
      const uint32_t k = sc[oh * (C + 4) + cg * EPC + j];
      o0v.e[j] = k == 1u ? gr : zero;
      o1v.e[j] = k == 2u ? gr : zero;
      o2v.e[j] = k == 3u ? gr : zero;
      o3v.e[j] = k == 4u ? gr : zero;
    }
    *reinterpret_cast<uint4*>(dx + base) = o0v.v;
    *reinterpret_cast<uint4*>(dx + base + C) = o1v.v;
    *reinterpret_cast<uint4*>(dx + base + rowp) = o2v.v;
    *reinterpret_cast<uint4*>(dx + base + rowp + C) = o3v.v;
  }
}
// NHWC forward with codes (models without the fused conv epilogue) and backward from codes: one thread = EPC channels of a window
template <typename T>
__global__ __launch_bounds__(256) void pool_fwd_nhwc_code_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ code,
                                                                 int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W2 * groups) return;
  const int cg = t % groups, ow = t / groups;
  const int oh = blockIdx.y % H2;
  const int64_t b = blockIdx.y / H2;
  const int64_t rowp = (int64_t)W * C;
  const int64_t base = (((b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
  Chunk<T> v0, v1, v2, v3, o;
  v0.v = *reinterpret_cast<const uint4*>(x + base);
  v1.v = *reinterpret_cast<const uint4*>(x + base + C);
  v2.v = *reinterpret_cast<const uint4*>(x + base + rowp);
  v3.v = *reinterpret_cast<const uint4*>(x + base + rowp + C);
  union { uint8_t b[EPC]; uint32_t w[EPC / 4]; } k;
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    float m;
    k.b[j] = (uint8_t)pool_code(DT<T>::from(v0.e[j]), DT<T>::from(v1.e[j]), DT<T>::from(v2.e[j]), DT<T>::from(v3.e[j]), &m);
    o.e[j] = DT<T>::to(m);
  }
  const int64_t po = (((b * H2 + oh) * (int64_t)W2 + ow) * C) + cg * EPC;
  *reinterpret_cast<uint4*>(y + po) = o.v;
#pragma unroll
  for (int w = 0; w < EPC / 4; ++w) *reinterpret_cast<uint32_t*>(code + po + 4 * w) = k.w[w];
}
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_nhwc_code_kernel(const uint8_t* __restrict__ code, const T* __restrict__ dy, T* __restrict__ dx,
                                                                 int B, int H, int W, int C) {
  constexpr int EPC = DT<T>::EPC;
  const int H2 = H / 2, W2 = W / 2, groups = C / EPC;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= W2 * groups) return;
  const int cg = t % groups, ow = t / groups;
  const int oh = blockIdx.y % H2;
  const int64_t b = blockIdx.y / H2;
  const int64_t rowp = (int64_t)W * C;
  const int64_t base = (((b * H + 2 * oh) * W + 2 * ow) * (int64_t)C) + cg * EPC;
  const int64_t po = (((b * H2 + oh) * (int64_t)W2 + ow) * C) + cg * EPC;
  Chunk<T> g, o0, o1, o2, o3;
  g.v = *reinterpret_cast<const uint4*>(dy + po);
  union { uint8_t b[EPC]; uint32_t w[EPC / 4]; } k;
#pragma unroll
  for (int w = 0; w < EPC / 4; ++w) k.w[w] = *reinterpret_cast<const uint32_t*>(code + po + 4 * w);
  const T zero = DT<T>::to(0.f);
#pragma unroll
  for (int j = 0; j < EPC; ++j) {
    o0.e[j] = k.b[j] == 1 ? g.e[j] : zero;
    o1.e[j] = k.b[j] == 2 ? g.e[j] : zero;
    o2.e[j] = k.b[j] == 3 ? g.e[j] : zero;
    o3.e[j] = k.b[j] == 4 ? g.e[j] : zero;
  }
  *reinterpret_cast<uint4*>(dx + base) = o0.v;
  *reinterpret_cast<uint4*>(dx + base + C) = o1.v;
  *reinterpret_cast<uint4*>(dx + base + rowp) = o2.v;
  *reinterpret_cast<uint4*>(dx + base + rowp + C) = o3.v;
}
// rows/cols that floor-mode pooling drops (odd H or W) get zero gradient: touch only those pixels
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_edges_kernel(T* __restrict__ dx, int B, int H, int W, int C) {
  const int er = H & 1, ec = W & 1;
  const int64_t per_img = (int64_t)er * W + (int64_t)ec * (H - er);       // dropped pixels per image
  const int64_t total = (int64_t)B * per_img * C;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int64_t e = (i / C) % per_img;
    const int64_t b = i / (C * per_img);
    int yh, xw;
    if (e < (int64_t)er * W) { yh = H - 1; xw = (int)e; }
    else { yh = (int)(e - (int64_t)er * W); xw = W - 1; }
    DT<T>::st(dx + (((b * H + yh) * W + xw) * (int64_t)C) + c, 0.f);
  }
}

// ================================================================================================ wgrad, NHWC native
// dW[co][ci][tap] += sum_px dY[px][co] * X[px + tap][ci]   straight from the NHWC tensors (no planar copies):
// a workgroup owns a 64(co) x 64(ci) x 9(tap) block of dW and walks 8x16-pixel patches; per patch it stages the halo
// patch of X (180 px x 64 ci) and the dY tile (128 px x 64 co) in LDS in their natural pixel-major layout and builds the
// MFMA operands (which need 8 CONSECUTIVE PIXELS per lane) with the transposing LDS read ds_read_b64_tr_b16:
//   in each 16-lane group, lane i slot j receives element (i&3) of the 8-byte row supplied by lane 4j+(i>>2)
//   (measured: tools/probes/tr_read_probe.hip), so lane i supplying &T[p0 + (i>>2)][c0 + 4*(i&3)] gets T[p0..p0+3][c0+i].
// A tap is a row offset into the halo patch, so all 9 taps reuse one staged patch: 2*64*576*128 flop per 41 KB staged.
// fp32 mode uses one 4-byte read per MFMA operand element instead (k <-> lane group, conflict free).
struct WgradNArgs {
  const void* x; const void* dy; float* dw; float* db;
  float* ws;    // optional: per-workgroup partial dW blocks [gridDim.y][gridDim.x][9][64 co][64 ci] (two-stage reduction)
  int B, H, W, Cin, Cout, tiles_h, tiles_w, npatch, patches_per_wg, nci;
  int ablate;   // tuning only (ASR_WGRAD_ABLATE): 1 = no global loads, 2 = no MFMA loop, 4 = no final atomics
};

template <typename T> struct WgPack;
template <> struct WgPack<bf16_t> {
  // pack = 8 consecutive pixels (k = 8g .. 8g+7 of a 32-pixel macro step) of channel c0 + lr
  // pixel (macro step ms, k) -> patch row 2*ms + (k >> 4), col k & 15
  template <int PITCH>
  static __device__ __forceinline__ uint4 load(const unsigned char* tile, int ms, int lr, int g, int c0, int row_pitch_px,
                                               int dy, int dx) {
    const int y = 2 * ms + (g >> 1), x = 8 * (g & 1) + (lr >> 2);
    const unsigned char* p = tile + ((y + dy) * row_pitch_px + x + dx) * PITCH + (c0 + 4 * (lr & 3)) * 2;
    const uint2 lo = asr_lds_read_tr16(p), hi = asr_lds_read_tr16(p + 4 * PITCH);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
  static constexpr int NMS = 4;      // 128 pixels / 32
};
template <> struct WgPack<float> {
  // pack element s (the s-th 16x16x4 MFMA of the macro step) <-> pixel 4*s + g of patch row ms
  template <int PITCH>
  static __device__ __forceinline__ uint4 load(const unsigned char* tile, int ms, int lr, int g, int c0, int row_pitch_px,
                                               int dy, int dx) {
    const unsigned char* p = tile + ((ms + dy) * row_pitch_px + g + dx) * PITCH + (c0 + lr) * 4;
    uint4 r;
    r.x = *reinterpret_cast<const uint32_t*>(p);
    r.y = *reinterpret_cast<const uint32_t*>(p + 4 * PITCH);
    r.z = *reinterpret_cast<const uint32_t*>(p + 8 * PITCH);
    r.w = *reinterpret_cast<const uint32_t*>(p + 12 * PITCH);
    return r;
  }
  static constexpr int NMS = 8;      // 128 pixels / 16
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_nhwc_kernel(WgradNArgs p) {
  constexpr int EPC = DT<T>::EPC, ESZ = (int)sizeof(T);
  constexpr int CPP = 64 / EPC;
  constexpr int PP = 64 * ESZ + 16;          // LDS pitch of one pixel's 64-channel slice
  constexpr int NX = 180 * CPP, NDY = 128 * CPP;
  constexpr int RX = (NX + 255) / 256, RDY = NDY / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sX = smem;                  // halo patch 10 x 18 pixels
  unsigned char* sD = smem + 180 * PP;       // dY tile 8 x 16 pixels
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, g = lane >> 4;
  const int co0 = (blockIdx.y / p.nci) * 64, ci0 = (blockIdx.y % p.nci) * 64;
  const T* X = static_cast<const T*>(p.x);
  const T* DY = static_cast<const T*>(p.dy);
  const int p_beg = blockIdx.x * p.patches_per_wg, p_end = min(p.npatch, p_beg + p.patches_per_wg);

  f32x4_t acc[9][4];                         // [tap][co fragment]; this wave's ci fragment is `wave`
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.db != nullptr && ci0 == 0 && wave == 0;

  u32x4_t rx[RX], rd[RDY];
  auto gload = [&](int patch) __attribute__((always_inline)) {
    int t = patch;
    const int tw = t % p.tiles_w; t /= p.tiles_w;
    const int th = t % p.tiles_h;
    const int b = t / p.tiles_h;
    const int h0 = th * 8, w0 = tw * 16;
#pragma unroll
    for (int i = 0; i < RX; ++i) {
      const int c = tid + i * 256;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (c < NX) {
        const int hp = c / CPP, ch = c % CPP;
        const int gy = h0 + hp / 18 - 1, gx = w0 + hp % 18 - 1;
        if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
          v = *reinterpret_cast<const u32x4_t*>(X + (((int64_t)b * p.H + gy) * p.W + gx) * p.Cin + ci0 + ch * EPC);
      }
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < RDY; ++i) {
      const int c = tid + i * 256, px = c / CPP, ch = c % CPP;
      const int gy = h0 + (px >> 4), gx = w0 + (px & 15);
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (gy < p.H && gx < p.W)
        v = *reinterpret_cast<const u32x4_t*>(DY + (((int64_t)b * p.H + gy) * p.W + gx) * p.Cout + co0 + ch * EPC);
      rd[i] = v;
    }
  };
  auto swrite = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < RX; ++i) {
      const int c = tid + i * 256;
      if (c < NX) *reinterpret_cast<u32x4_t*>(sX + (c / CPP) * PP + (c % CPP) * 16) = rx[i];
    }
#pragma unroll
    for (int i = 0; i < RDY; ++i) {
      const int c = tid + i * 256;
      *reinterpret_cast<u32x4_t*>(sD + (c / CPP) * PP + (c % CPP) * 16) = rd[i];
    }
  };

  if (p_beg < p_end) gload(p_beg);
  for (int patch = p_beg; patch < p_end; ++patch) {
    swrite();
    __syncthreads();
    if (patch + 1 < p_end && !ASR_ABL(p, 1)) gload(patch + 1);          // next patch's HBM latency hides under this patch's MFMAs
#pragma unroll 1
    for (int ms = 0; ms < (ASR_ABL(p, 2) ? 0 : WgPack<T>::NMS); ++ms) {
      uint4 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = WgPack<T>::template load<PP>(sD, ms, lr, g, i * 16, 16, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Chunk<T> c; c.v = a[i];
#pragma unroll
          for (int e = 0; e < EPC; ++e) bsum[i] += DT<T>::from(c.e[e]);
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const uint4 bfr = WgPack<T>::template load<PP>(sX, ms, lr, g, wave * 16, 18, t / 3, t % 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) mma16<T>(acc[t][i], a[i], bfr);
      }
    }
    __syncthreads();
  }

  if (p.ws) {
    // two-stage reduction: 36,864 plain stores per workgroup instead of as many fp32 atomics on the same 147 KB of dW
    // (measured: the atomics were > 50 % of this kernel's time); wgrad_reduce_kernel folds the partial blocks into dW
    float* part = p.ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (9 * 64 * 64);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[(t * 64 + i * 16 + g * 4 + r) * 64 + wave * 16 + lr] = acc[t][i][r];
  } else {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + i * 16 + g * 4 + r, ci = ci0 + wave * 16 + lr;
          if (!ASR_ABL(p, 4) || acc[t][i][r] == 12345.f) atomicAdd(p.dw + ((int64_t)co * p.Cin + ci) * 9 + t, acc[t][i][r]);
        }
  }
  if (do_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) atomicAdd(p.db + co0 + i * 16 + lr, v);
    }
  }
}

// dW[co0+co][ci0+ci][t] += sum over this slice of the workgroup partials ws[by][wg][t][co][ci]
// No atomics (1.2 M same-line fp32 atomics were most of this kernel's time: 8 slices x 147 K elements x 4 blocks at the ~40 / ns the
// chip sustains) and a fixed summation order: a workgroup owns 256 consecutive elements, its four waves each add a quarter of the
// partial blocks with 16-byte loads (eight in flight), the quarters meet in LDS and wave 0 does the plain dw += .
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* dw, int wgx, int nci, int Cin) {
  __shared__ float4 red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int e = (blockIdx.x * 64 + col) * 4;             // 4 elements of the 9 x 64 x 64 block: (t, co, ci .. ci+3), ci fastest
  const int by = blockIdx.y;
  const int per = (wgx + 3) / 4;
  const int w0 = grp * per, w1 = min(wgx, w0 + per);
  const float* src = ws + ((int64_t)by * wgx) * (9 * 64 * 64) + e;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int w = w0; w < w1; ++w) {
    const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)w * (9 * 64 * 64));
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  red[grp][col] = acc;
  __syncthreads();
  if (grp != 0) return;
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float4 v = red[k][col];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const int ci = e & 63, co = (e >> 6) & 63, t = e >> 12;
  const int co0 = (by / nci) * 64, ci0 = (by % nci) * 64;
  float* dst = dw + ((int64_t)(co0 + co) * Cin + ci0 + ci) * 9 + t;
  dst[0] += acc.x; dst[9] += acc.y; dst[18] += acc.z; dst[27] += acc.w;
}

// once per kernel instantiation (never during a stream capture: the first eager/warm-up launch does it)
template <typename K> void allow_big_lds(K kernel, size_t lds) {
  static size_t granted = 0;      // one static per template instantiation = per kernel
  if (lds > 48 * 1024 && lds > granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    granted = lds;
  }
}

template <typename T, int NCO, int TH, int TPS, int WBUF, bool PT = false>
int launch_igemm_t(const ConvArgs& a, hipStream_t s) {
  ConvArgs p = a;
  p.tiles_h = (p.H + TH - 1) / TH;
  p.tiles_w = (p.W + 15) / 16;
  p.xcd_order = asr_tuning("IGEMM_XCD", 1) != 0;
  size_t lds = (size_t)((TH + 2) * 18 + WBUF * TPS * NCO) * (64 * sizeof(T));      // (>= the 24 KB the pooled epilogue stages)
  allow_big_lds(conv3x3_igemm_kernel<T, NCO, TH, TPS, WBUF, PT>, lds);
  hipLaunchKernelGGL((conv3x3_igemm_kernel<T, NCO, TH, TPS, WBUF, PT>), dim3((unsigned)(p.B * p.tiles_h * p.tiles_w)), dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
// Tile height: 16 rows in bf16 (wave tile 4 x 4 / 4 x 8 fragments: 2 / 2.7 MFMAs per LDS operand read), 8 rows in fp32 (LDS).
// History of this choice: with register-staged patches and double-buffered weights the 16-row tile LOST to the 8-row one (a
// workgroup per CU less); once the patch came in by LDS-DMA, the weights were single buffered and the mask loads of the epilogue
// hoisted, it wins on every layer (profiles/r01_microbench_v7.txt; ASR_IGEMM_TH=8 restores the small tile).
// ASR_IGEMM_TPS=2 stages TWO taps per step at Cout 64 with the 8-row tile -- measured slower.
// Weights are SINGLE buffered in LDS (prefetched in registers): one more barrier per step, but one more workgroup per CU --
// +9 % (Cout 64) to +23 % (Cout 128) measured; ASR_IGEMM_WBUF=2 restores the double buffer.
template <typename T, int NCO>
int launch_igemm(const ConvArgs& a, hipStream_t s) {
  const int th = (int)asr_tuning("IGEMM_TH", 16);
  const int tps = (int)asr_tuning("IGEMM_TPS", 1);
  const int wbuf = (int)asr_tuning("IGEMM_WBUF", 1);
  if constexpr (sizeof(T) == 2) {
    if (th == 16) {
      // (A/B, round 3: profiles/r03_igemm_variants_ab.txt) at the 16-row tile two weight buffers or two taps per step still leave two
      // workgroups per CU (74 KB each) -- and change nothing: 307 - 313 us on the 128 -> 128 layer whichever way
      if (wbuf == 2 && tps == 1) return launch_igemm_t<T, NCO, 16, 1, 2>(a, s);
      if (tps == 2) return launch_igemm_t<T, NCO, 16, 2, 1>(a, s);
      return launch_igemm_t<T, NCO, 16, 1, 1>(a, s);
    }
  }
  if (sizeof(T) == 2 && NCO == 64 && tps == 2) return launch_igemm_t<T, NCO, 8, 2, 2>(a, s);
  if (wbuf == 1) return launch_igemm_t<T, NCO, 8, 1, 1>(a, s);
  return launch_igemm_t<T, NCO, 8, 1, 2>(a, s);
}

inline unsigned stream_grid(int64_t total_threads) {
  int64_t blocks = ceil_div64(total_threads, 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  return (unsigned)blocks;
}

}  // namespace

extern "C" int asr_conv1_fwd(const float* x, const float* w, const float* bias, void* y, int B, int H, int W, int C0, int dtype,
                             hipStream_t s) {
  ASR_CHECK_ARG(x && w && bias && y && B >= 0 && H > 0 && W > 0 && C0 > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  if (C0 % epc != 0 || !aligned16(y)) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  if (256 % (C0 / epc) != 0) return ASR_EUNSUPPORTED;
  int64_t rows = (int64_t)B * H;                                          // blocks walk image rows
  ASR_CHECK_ARG(rows < ((int64_t)1 << 31));
  unsigned grid1 = (unsigned)(rows < 8192 ? rows : 8192);
  AsrProfScope prof(ASR_OP_CONV1, s);
  if (W % C1_PW == 0) {
    if (dtype == ASR_F32) hipLaunchKernelGGL((conv1_fwd_kernel<float, true>), dim3(grid1), dim3(256), 0, s, x, w, bias, (float*)y, B, H, W, C0);
    else hipLaunchKernelGGL((conv1_fwd_kernel<bf16_t, true>), dim3(grid1), dim3(256), 0, s, x, w, bias, (bf16_t*)y, B, H, W, C0);
  } else
  if (dtype == ASR_F32) hipLaunchKernelGGL((conv1_fwd_kernel<float, false>), dim3(grid1), dim3(256), 0, s, x, w, bias, (float*)y, B, H, W, C0);
  else hipLaunchKernelGGL((conv1_fwd_kernel<bf16_t, false>), dim3(grid1), dim3(256), 0, s, x, w, bias, (bf16_t*)y, B, H, W, C0);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_conv1_wgrad(const float* x, const void* dy, float* dw, float* db, int B, int H, int W, int C0, int dtype,
                               hipStream_t s) {
  ASR_CHECK_ARG(x && dy && dw && db && B >= 0 && H > 0 && W > 0 && C0 > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  if (C0 % epc != 0 || 256 % (C0 / epc) != 0 || !aligned16(dy)) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  int64_t blocks = (int64_t)B * H;        // blocks walk image rows
  ASR_CHECK_ARG(blocks < ((int64_t)1 << 31));
  if (blocks > 1024) blocks = 1024;       // every block ends with C0*10 same-address global atomics
  if (blocks < 1) blocks = 1;
  const size_t lds = (size_t)C0 * 10 * sizeof(float);
  AsrProfScope prof(ASR_OP_CONV1, s);
  // bf16 storage, 64 channels: matrix-core kernel with the pixel as contraction index (conv1_wgrad_mfma.hip)
  const bool mfma = asr_tuning("CONV1_WGRAD_MFMA", 1) != 0;
  if (mfma && dtype == ASR_BF16 && C0 == 64) return asr_conv1_wgrad_mfma_launch(x, (const bf16_t*)dy, dw, db, B, H, W, s);
  if (dtype == ASR_F32) hipLaunchKernelGGL((conv1_wgrad_kernel<float>), dim3((unsigned)blocks), dim3(256), lds, s, x, (const float*)dy, dw, db, B, H, W, C0);
  else hipLaunchKernelGGL((conv1_wgrad_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), lds, s, x, (const bf16_t*)dy, dw, db, B, H, W, C0);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_conv_pack_weight(const float* w, void* wk, void* wd, int Cout, int Cin, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(w && (wk || wd) && Cout > 0 && Cin > 0);
  const int64_t total = (int64_t)Cout * Cin * 9;
  if (dtype == ASR_F32) hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(stream_grid(total)), dim3(256), 0, s, w, (float*)wk, (float*)wd, Cout, Cin);
  else if (dtype == ASR_BF16) hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(stream_grid(total)), dim3(256), 0, s, w, (bf16_t*)wk, (bf16_t*)wd, Cout, Cin);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_conv_pack_weight_multi(int n, const float* const* w, void* const* wk, void* const* wd, const int* Cout, const int* Cin,
                                          int dtype, hipStream_t s) {
  ASR_CHECK_ARG(n >= 0 && n <= 8 && (n == 0 || (w && wk && wd && Cout && Cin)));
  if (n == 0) return ASR_OK;
  PackMulti a{};
  a.n = n;
  int64_t total = 0;
  for (int k = 0; k < n; ++k) {
    ASR_CHECK_ARG(w[k] && (wk[k] || wd[k]) && Cout[k] > 0 && Cin[k] > 0);
    a.w[k] = w[k]; a.wk[k] = wk[k]; a.wd[k] = wd[k]; a.cout[k] = Cout[k]; a.cin[k] = Cin[k];
    a.start[k] = total;
    total += (int64_t)Cout[k] * Cin[k] * 9;
  }
  for (int k = n; k <= 8; ++k) a.start[k] = total;
  if (dtype == ASR_F32) hipLaunchKernelGGL((pack_weight_multi_kernel<float>), dim3(stream_grid(total)), dim3(256), 0, s, a);
  else if (dtype == ASR_BF16) hipLaunchKernelGGL((pack_weight_multi_kernel<bf16_t>), dim3(stream_grid(total)), dim3(256), 0, s, a);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_conv3x3_igemm(const void* x, const void* wk, const float* bias, const void* mask_src, void* y, int B, int H,
                                 int W, int Cin, int Cout, int relu, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && wk && y && B >= 0 && H > 0 && W > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (Cin % 64 != 0 || (Cout != 64 && Cout != 128) || !aligned16(x) || !aligned16(wk) || !aligned16(y) ||
      (mask_src && !aligned16(mask_src))) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  ConvArgs p{};
  p.x = x; p.wk = wk; p.bias = bias; p.mask_src = mask_src; p.y = y;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
  p.ablate = (int)asr_tuning("IGEMM_ABLATE", 0);
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  // the 64 -> 64 channel bf16 layer (full-resolution conv2 and its dgrad) has a persistent kernel with register-resident weights
  const bool c64 = asr_tuning("C64", 1) != 0;
  if (c64 && dtype == ASR_BF16 && Cin == 64 && Cout == 64 && (int64_t)B * H * W * 128 < ((int64_t)1 << 32) && !p.ablate) {
    C64Args a{};
    a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias;
    a.mask = static_cast<const bf16_t*>(mask_src); a.y = static_cast<bf16_t*>(y);
    a.B = B; a.H = H; a.W = W; a.relu = relu;
    return asr_conv3x3_c64_launch(a, s);
  }
  // 64 -> 128 channels without a mask (conv.5 forward) in ONE pass: the weight-stationary kernel of conv_ws.hip with 64 input channels
  // (a wave keeps 32 of the 128 output channels x 9 x 64 in 144 registers, two workgroups per CU, 4-row tiles; round 5).  WS64 = 0
  // (tuning): the two-pass form below
  if (dtype == ASR_BF16 && Cin == 64 && Cout == 128 && !mask_src && !p.ablate && asr_tuning("WS64", 1) != 0) {
    WsArgs a{};
    a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias; a.y = static_cast<bf16_t*>(y);
    a.B = B; a.H = H; a.W = W; a.Cin = 64; a.Cout = Cout; a.relu = relu;
    const int rc = asr_conv3x3_ws128_launch(a, s);
    if (rc != ASR_EUNSUPPORTED) return rc;
  }
  // 64 -> 128 channels without a mask (conv.5 forward): the same kernel once per half of the output channels -- each half's 72 KB of
  // weights sits in registers, the 64-channel input is read twice (the second time from L2 / MALL); 0 = the generic implicit GEMM
  if (c64 && dtype == ASR_BF16 && Cin == 64 && Cout == 128 && !mask_src && (int64_t)B * H * W * 256 < ((int64_t)1 << 32) && !p.ablate &&
      asr_tuning("C64_SPLIT", 1) != 0) {
    for (int half = 0; half < 2; ++half) {
      C64Args a{};
      a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk) + (size_t)half * 64 * 9 * 64;
      a.bias = bias ? bias + half * 64 : nullptr;
      a.y = static_cast<bf16_t*>(y) + half * 64; a.ypix = 256;
      a.B = B; a.H = H; a.W = W; a.relu = relu;
      const int rc = asr_conv3x3_c64_launch(a, s);
      if (rc != ASR_OK) return rc;
    }
    return ASR_OK;
  }
  // 128 input channels in bf16 (conv.7's data gradient with conv.5's ReLU mask, conv.5's data gradient): the persistent
  // weight-stationary kernel of conv_ws.hip; WS128 = 0 (tuning) or a shape outside its domain -> the generic implicit GEMM
  if (dtype == ASR_BF16 && Cin == 128 && !p.ablate && asr_tuning("WS128", 1) != 0) {
    WsArgs a{};
    a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias;
    a.mask = static_cast<const bf16_t*>(mask_src); a.y = static_cast<bf16_t*>(y);
    a.B = B; a.H = H; a.W = W; a.Cin = 128; a.Cout = Cout; a.relu = relu;
    const int rc = asr_conv3x3_ws128_launch(a, s);
    if (rc != ASR_EUNSUPPORTED) return rc;
  }
  if (dtype == ASR_F32) return Cout == 64 ? launch_igemm<float, 64>(p, s) : launch_igemm<float, 128>(p, s);
  return Cout == 64 ? launch_igemm<bf16_t, 64>(p, s) : launch_igemm<bf16_t, 128>(p, s);
}

extern "C" int64_t asr_relu_bits_bytes(int B, int H, int W, int C) {
  if (B < 0 || H <= 0 || W <= 0 || C != 128) return -1;
  return (int64_t)B * (2 * ((H + 7) / 8)) * ((W + 15) / 16) * (C / 32) * 256;       // one dword per (4 x 16-pixel tile, 32 channels, lane)
}

// The convolution with a ReLU mask of ONE BIT per element on either side (conv_ws.hip): bits_out -- written for this launch's ReLU output
// (conv.5 forward, bf16 64 -> 128); bits_in -- the output is zeroed where the bit is 0 (conv.7's data gradient, bf16 128 -> 128).
extern "C" int asr_conv3x3_igemm_bits(const void* x, const void* wk, const float* bias, const uint8_t* bits_in, void* y, uint8_t* bits_out,
                                      int B, int H, int W, int Cin, int Cout, int relu, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && wk && y && B >= 0 && H > 0 && W > 0);
  ASR_CHECK_ARG((bits_in != nullptr) != (bits_out != nullptr));
  if (dtype != ASR_BF16 || !aligned16(x) || !aligned16(wk) || !aligned16(y) || ((uintptr_t)bits_in & 3) || ((uintptr_t)bits_out & 3))
    return ASR_EUNSUPPORTED;
  if (bits_in ? (Cin != 128 || Cout != 128) : (Cin != 64 || Cout != 128 || !relu)) return ASR_EUNSUPPORTED;
  if (asr_tuning("WS_BITS", 1) == 0) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  WsArgs a{};
  a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias; a.y = static_cast<bf16_t*>(y);
  a.bits_in = bits_in; a.bits_out = bits_out;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.relu = relu;
  return asr_conv3x3_ws128_launch(a, s);
}

extern "C" int asr_conv3x3_relu_pool(const void* x, const void* wk, const float* bias, void* y, void* pool, int B, int H, int W,
                                     int Cin, int Cout, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && wk && y && pool && B >= 0 && H > 0 && W > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  // only the layer that has it in the model: bf16, 64 -> 64 channels (conv_c64.hip); callers fall back to conv + asr_maxpool_fwd
  if (dtype != ASR_BF16 || Cin != 64 || Cout != 64 || !aligned16(x) || !aligned16(wk) || !aligned16(y) || !aligned16(pool) ||
      (int64_t)B * H * W * 128 >= ((int64_t)1 << 32))
    return ASR_EUNSUPPORTED;
  const bool fused = asr_tuning("CONV_POOL", 1) != 0;
  if (!fused) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  C64Args a{};
  a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias;
  a.y = static_cast<bf16_t*>(y); a.pool = static_cast<bf16_t*>(pool);
  a.B = B; a.H = H; a.W = W; a.relu = 1;
  return asr_conv3x3_c64_launch(a, s);
}

/* The same without the full-resolution output: pool (B, H/2, W/2, Cout) and one selection byte per pooled element (`code`, layout of
 * pool; asr_maxpool_bwd_code consumes it) -- what the training step needs of conv.2: its un-pooled output is only ever read to find
 * the arg max again.  y_or_null != null additionally stores y. */
extern "C" int asr_conv3x3_relu_pool_code(const void* x, const void* wk, const float* bias, void* y_or_null, void* pool, uint8_t* code,
                                          int B, int H, int W, int Cin, int Cout, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && wk && pool && code && B >= 0 && H > 0 && W > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (dtype != ASR_BF16 || Cin != 64 || Cout != 64 || !aligned16(x) || !aligned16(wk) || (y_or_null && !aligned16(y_or_null)) || !aligned16(pool) ||
      (((uintptr_t)code) & 7) != 0 || (int64_t)B * H * W * 128 >= ((int64_t)1 << 32) || asr_tuning("CONV_POOL", 1) == 0)
    return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  C64Args a{};
  a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias;
  a.y = static_cast<bf16_t*>(y_or_null); a.pool = static_cast<bf16_t*>(pool); a.code = code;
  a.B = B; a.H = H; a.W = W; a.relu = 1;
  return asr_conv3x3_c64_launch(a, s);
}

/* y never stored: pool (B, W/2, Cout, H/2) = the encoder layout (B, T', C F') of the 2x2/2 max-pool of ReLU(conv3x3_pad1(x; wk) + bias), and
 * one selection byte per pooled element in the same layout (conv.7 + ReLU + MaxPool2d + view / transpose, transformer.py:50-52,74-76).
 * ASR_EUNSUPPORTED unless bf16, Cout = 128, Cin a multiple of 64, H and W multiples of 16 (callers use asr_conv3x3_igemm +
 * asr_maxpool_fwd_code). */
namespace {
int conv3x3_relu_pool_tcf_code_impl(const void* x, const void* wk, const float* bias, void* pool, uint8_t* code, int code_cl, int B, int H,
                                    int W, int Cin, int Cout, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && wk && pool && code && B >= 0 && H > 0 && W > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (dtype != ASR_BF16 || Cout != 128 || Cin % 64 != 0 || H % 8 != 0 || W % 16 != 0 || !aligned16(x) || !aligned16(wk) || !aligned16(pool) ||
      (((uintptr_t)code) & 7) != 0 || asr_tuning("CONV_POOL", 1) == 0)
    return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  if (Cin == 128 && H % 8 == 0 && asr_tuning("WS128", 1) != 0) {        // conv.7 forward: persistent weight-stationary kernel (conv_ws.hip)
    WsArgs a{};
    a.x = static_cast<const bf16_t*>(x); a.wk = static_cast<const bf16_t*>(wk); a.bias = bias;
    a.pool = static_cast<bf16_t*>(pool); a.code = code; a.code_cl = code_cl;
    a.B = B; a.H = H; a.W = W; a.Cin = 128; a.Cout = Cout; a.relu = 1;
    AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
    const int rc = asr_conv3x3_ws128_launch(a, s);
    if (rc != ASR_EUNSUPPORTED) return rc;
  }
  if (code_cl) return ASR_EUNSUPPORTED;        // channel-last selection bytes: the weight-stationary kernel only
  if (H % 16 != 0 || asr_tuning("IGEMM_TH", 16) != 16) return ASR_EUNSUPPORTED;
  ConvArgs p{};
  p.x = x; p.wk = wk; p.bias = bias; p.pool = pool; p.code = code;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = 1;
  AsrProfScope prof(ASR_OP_CONV_IGEMM, s);
  return launch_igemm_t<bf16_t, 128, 16, 1, 1, true>(p, s);
}
}  // namespace
extern "C" int asr_conv3x3_relu_pool_tcf_code(const void* x, const void* wk, const float* bias, void* pool, uint8_t* code, int B, int H,
                                              int W, int Cin, int Cout, int dtype, hipStream_t s) {
  return conv3x3_relu_pool_tcf_code_impl(x, wk, bias, pool, code, 0, B, H, W, Cin, Cout, dtype, s);
}
extern "C" int asr_conv3x3_relu_pool_tcf_codecl(const void* x, const void* wk, const float* bias, void* pool, uint8_t* code_cl, int B, int H,
                                                int W, int Cin, int Cout, int dtype, hipStream_t s) {
  return conv3x3_relu_pool_tcf_code_impl(x, wk, bias, pool, code_cl, 1, B, H, W, Cin, Cout, dtype, s);
}

extern "C" int asr_maxpool_fwd(const void* x, void* y, int B, int H, int W, int C, int out_tcf, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && y && B >= 0 && H >= 2 && W >= 2 && C > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (B == 0) return ASR_OK;
  const int epc = dtype == ASR_F32 ? 4 : 8;
  const int H2 = H / 2, W2 = W / 2;
  AsrProfScope prof(ASR_OP_POOL, s);
  if (out_tcf) {
    const size_t lds = (size_t)H2 * (C + 1) * sizeof(float);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    if (C % epc == 0 && H2 % epc == 0 && aligned16(x) && aligned16(y)) {
      if (dtype == ASR_F32) { allow_big_lds(pool_fwd_tcf_vec_kernel<float>, lds); hipLaunchKernelGGL((pool_fwd_tcf_vec_kernel<float>), dim3(B * W2), dim3(256), lds, s, (const float*)x, (float*)y, B, H, W, C); }
      else { allow_big_lds(pool_fwd_tcf_vec_kernel<bf16_t>, lds); hipLaunchKernelGGL((pool_fwd_tcf_vec_kernel<bf16_t>), dim3(B * W2), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C); }
    } else
    if (dtype == ASR_F32) { allow_big_lds(pool_fwd_tcf_kernel<float>, lds); hipLaunchKernelGGL((pool_fwd_tcf_kernel<float>), dim3(B * W2), dim3(256), lds, s, (const float*)x, (float*)y, B, H, W, C); }
    else { allow_big_lds(pool_fwd_tcf_kernel<bf16_t>, lds); hipLaunchKernelGGL((pool_fwd_tcf_kernel<bf16_t>), dim3(B * W2), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C); }
  } else {
    if (C % epc != 0 || !aligned16(x) || !aligned16(y)) return ASR_EUNSUPPORTED;
    if ((int64_t)B * H2 > 65535) return ASR_EUNSUPPORTED;
    const dim3 grid((unsigned)ceil_div64((int64_t)W2 * (C / epc), 256), (unsigned)(B * H2));
    if (dtype == ASR_F32) hipLaunchKernelGGL((pool_fwd_nhwc_kernel<float>), grid, dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C);
    else hipLaunchKernelGGL((pool_fwd_nhwc_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, B, H, W, C);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_maxpool_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int in_tcf, int dtype,
                               hipStream_t s) {
  ASR_CHECK_ARG(x && dy && dx && B >= 0 && H >= 2 && W >= 2 && C > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (B == 0) return ASR_OK;
  const int H2 = H / 2, W2 = W / 2;
  AsrProfScope prof(ASR_OP_POOL, s);
  if ((H & 1) || (W & 1)) {
    const int64_t total = (int64_t)B * ((int64_t)(H & 1) * W + (int64_t)(W & 1) * (H - (H & 1))) * C;
    if (dtype == ASR_F32) hipLaunchKernelGGL((pool_bwd_edges_kernel<float>), dim3(stream_grid(total)), dim3(256), 0, s, (float*)dx, B, H, W, C);
    else hipLaunchKernelGGL((pool_bwd_edges_kernel<bf16_t>), dim3(stream_grid(total)), dim3(256), 0, s, (bf16_t*)dx, B, H, W, C);
    ASR_LAUNCH_CHECK();
  }
  if (in_tcf) {
    const size_t lds = (size_t)H2 * (C + 1) * sizeof(float);
    if (lds > 150 * 1024) return ASR_EUNSUPPORTED;
    const int epc = dtype == ASR_F32 ? 4 : 8;
    if (C % epc == 0 && H2 % epc == 0 && aligned16(x) && aligned16(dy) && aligned16(dx)) {
      if (dtype == ASR_F32) { allow_big_lds(pool_bwd_tcf_vec_kernel<float>, lds); hipLaunchKernelGGL((pool_bwd_tcf_vec_kernel<float>), dim3(B * W2), dim3(256), lds, s, (const float*)x, (const float*)dy, (float*)dx, B, H, W, C); }
      else { allow_big_lds(pool_bwd_tcf_vec_kernel<bf16_t>, lds); hipLaunchKernelGGL((pool_bwd_tcf_vec_kernel<bf16_t>), dim3(B * W2), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C); }
    } else
    if (dtype == ASR_F32) { allow_big_lds(pool_bwd_tcf_kernel<float>, lds); hipLaunchKernelGGL((pool_bwd_tcf_kernel<float>), dim3(B * W2), dim3(256), lds, s, (const float*)x, (const float*)dy, (float*)dx, B, H, W, C); }
    else { allow_big_lds(pool_bwd_tcf_kernel<bf16_t>, lds); hipLaunchKernelGGL((pool_bwd_tcf_kernel<bf16_t>), dim3(B * W2), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C); }
  } else {
    const int epc = dtype == ASR_F32 ? 4 : 8;
    if (C % epc != 0 || !aligned16(x) || !aligned16(dy) || !aligned16(dx)) return ASR_EUNSUPPORTED;
    if ((int64_t)B * H2 > 65535) return ASR_EUNSUPPORTED;
    const dim3 grid((unsigned)ceil_div64((int64_t)W2 * (C / epc), 256), (unsigned)(B * H2));
    if (dtype == ASR_F32) hipLaunchKernelGGL((pool_bwd_nhwc_kernel<float>), grid, dim3(256), 0, s, (const float*)x, (const float*)dy, (float*)dx, B, H, W, C);
    else hipLaunchKernelGGL((pool_bwd_nhwc_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ---- pooling with selection codes (see pool_fwd_tcf_code_kernel): 16-byte layouts only, ASR_EUNSUPPORTED otherwise (callers then
// use asr_maxpool_fwd / asr_maxpool_bwd, which need the pre-pool activations)
extern "C" int asr_maxpool_fwd_code(const void* x, void* y, uint8_t* code, int B, int H, int W, int C, int out_tcf, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(x && y && code && B >= 0 && H >= 2 && W >= 2 && C > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  const int H2 = H / 2, W2 = W / 2;
  if (C % epc != 0 || !aligned16(x) || !aligned16(y) || (((uintptr_t)code) & 3) != 0) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_POOL, s);
  if (out_tcf) {
    const size_t lds = (size_t)H2 * (C + 1) * sizeof(float) + (size_t)H2 * (C + 4);
    if (lds > 150 * 1024 || H2 % epc != 0) return ASR_EUNSUPPORTED;
    if (dtype == ASR_F32) { allow_big_lds(pool_fwd_tcf_code_kernel<float>, lds); hipLaunchKernelGGL((pool_fwd_tcf_code_kernel<float>), dim3(B * W2), dim3(256), lds, s, (const float*)x, (float*)y, code, B, H, W, C); }
    else { allow_big_lds(pool_fwd_tcf_code_kernel<bf16_t>, lds); hipLaunchKernelGGL((pool_fwd_tcf_code_kernel<bf16_t>), dim3(B * W2), dim3(256), lds, s, (const bf16_t*)x, (bf16_t*)y, code, B, H, W, C); }
  } else {
    if ((int64_t)B * H2 > 65535) return ASR_EUNSUPPORTED;
    const dim3 grid((unsigned)ceil_div64((int64_t)W2 * (C / epc), 256), (unsigned)(B * H2));
    if (dtype == ASR_F32) hipLaunchKernelGGL((pool_fwd_nhwc_code_kernel<float>), grid, dim3(256), 0, s, (const float*)x, (float*)y, code, B, H, W, C);
    else hipLaunchKernelGGL((pool_fwd_nhwc_code_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)y, code, B, H, W, C);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_maxpool_bwd_code(const uint8_t* code, const void* dy, void* dx, int B, int H, int W, int C, int in_tcf, int dtype,
                                    hipStream_t s) {
  ASR_CHECK_ARG(code && dy && dx && B >= 0 && H >= 2 && W >= 2 && C > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  const int H2 = H / 2, W2 = W / 2;
  if (C % epc != 0 || !aligned16(dy) || !aligned16(dx) || (((uintptr_t)code) & 3) != 0) return ASR_EUNSUPPORTED;
  if (in_tcf && (H2 % epc != 0 || (size_t)H2 * (C + 1) * sizeof(float) + (size_t)H2 * (C + 4) > 150 * 1024)) return ASR_EUNSUPPORTED;
  if (!in_tcf && (int64_t)B * H2 > 65535) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_POOL, s);
  if ((H & 1) || (W & 1)) {
    const int64_t total = (int64_t)B * ((int64_t)(H & 1) * W + (int64_t)(W & 1) * (H - (H & 1))) * C;
    if (dtype == ASR_F32) hipLaunchKernelGGL((pool_bwd_edges_kernel<float>), dim3(stream_grid(total)), dim3(256), 0, s, (float*)dx, B, H, W, C);
    else hipLaunchKernelGGL((pool_bwd_edges_kernel<bf16_t>), dim3(stream_grid(total)), dim3(256), 0, s, (bf16_t*)dx, B, H, W, C);
    ASR_LAUNCH_CHECK();
  }
  if (in_tcf) {
    const size_t lds = (size_t)H2 * (C + 1) * sizeof(float) + (size_t)H2 * (C + 4);
    if (dtype == ASR_F32) { allow_big_lds(pool_bwd_tcf_code_kernel<float>, lds); hipLaunchKernelGGL((pool_bwd_tcf_code_kernel<float>), dim3(B * W2), dim3(256), lds, s, code, (const float*)dy, (float*)dx, B, H, W, C); }
    else { allow_big_lds(pool_bwd_tcf_code_kernel<bf16_t>, lds); hipLaunchKernelGGL((pool_bwd_tcf_code_kernel<bf16_t>), dim3(B * W2), dim3(256), lds, s, code, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C); }
  } else {
    const dim3 grid((unsigned)ceil_div64((int64_t)W2 * (C / epc), 256), (unsigned)(B * H2));
    if (dtype == ASR_F32) hipLaunchKernelGGL((pool_bwd_nhwc_code_kernel<float>), grid, dim3(256), 0, s, code, (const float*)dy, (float*)dx, B, H, W, C);
    else hipLaunchKernelGGL((pool_bwd_nhwc_code_kernel<bf16_t>), grid, dim3(256), 0, s, code, (const bf16_t*)dy, (bf16_t*)dx, B, H, W, C);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// workgroups along the pixel axis (x) and dW blocks (y) of the NHWC weight-gradient launch.  ONE definition (declared in
// conv_wgrad_dma.h): conv_level0.hip writes partial blocks on this grid and asr_conv3x3_wgrad_reduce folds them on it.
void asr_conv3x3_wgrad_grid(int B, int H, int W, int Cin, int Cout, int* wgx, int* blocks_y, int* patches_per_wg) {
  const int npatch = B * ((H + 7) / 8) * ((W + 15) / 16);
  *blocks_y = (Cout / 64) * (Cin / 64);
  int gx = 512 / *blocks_y;                         // ~2 workgroups per CU in flight
  if (gx < 1) gx = 1;
  int ppw = (npatch + gx - 1) / gx;
  if (ppw < 4) ppw = 4;
  *patches_per_wg = ppw;
  *wgx = (npatch + ppw - 1) / ppw;
}

extern "C" int64_t asr_conv3x3_wgrad_workspace(int B, int H, int W, int Cin, int Cout) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin % 64 != 0 || Cout % 64 != 0) return 0;
  int wgx, by, ppw;
  asr_conv3x3_wgrad_grid(B, H, W, Cin, Cout, &wgx, &by, &ppw);
  return (int64_t)wgx * by * 9 * 64 * 64;
}

namespace {
int conv3x3_wgrad_impl(const void* x, const void* dy, float* dw, float* db, float* workspace, int64_t workspace_floats, int B, int H,
                       int W, int Cin, int Cout, int dtype, bool reduce, hipStream_t s);
}
extern "C" int asr_conv3x3_wgrad_nhwc(const void* x, const void* dy, float* dw, float* db, float* workspace,
                                      int64_t workspace_floats, int B, int H, int W, int Cin, int Cout, int dtype, hipStream_t s) {
  return conv3x3_wgrad_impl(x, dy, dw, db, workspace, workspace_floats, B, H, W, Cin, Cout, dtype, true, s);
}
// first stage only: the per-workgroup partial dW blocks stay in `workspace` (required) until asr_conv3x3_wgrad_reduce
extern "C" int asr_conv3x3_wgrad_partials(const void* x, const void* dy, float* db, float* workspace, int64_t workspace_floats, int B,
                                          int H, int W, int Cin, int Cout, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(workspace && workspace_floats >= asr_conv3x3_wgrad_workspace(B, H, W, Cin, Cout));
  if (asr_conv3x3_wgrad_workspace(B, H, W, Cin, Cout) == 0) return ASR_EUNSUPPORTED;
  return conv3x3_wgrad_impl(x, dy, workspace /* dw is not touched without the reduction */, db, workspace, workspace_floats, B, H, W,
                            Cin, Cout, dtype, false, s);
}
// second stage: dw += the partial blocks of asr_conv3x3_wgrad_partials (same geometry arguments)
extern "C" int asr_conv3x3_wgrad_reduce(const float* workspace, float* dw, int B, int H, int W, int Cin, int Cout, hipStream_t s) {
  ASR_CHECK_ARG(workspace && dw && B >= 0 && H > 0 && W > 0);
  if (Cin % 64 != 0 || Cout % 64 != 0) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  int wgx, blocks_y, ppw;
  asr_conv3x3_wgrad_grid(B, H, W, Cin, Cout, &wgx, &blocks_y, &ppw);
  AsrProfScope prof(ASR_OP_CONV_WGRAD, s);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(9 * 64 * 64 / 256, (unsigned)blocks_y), dim3(256), 0, s, workspace, dw, wgx,
                     Cin / 64, Cin);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
namespace {
int conv3x3_wgrad_impl(const void* x, const void* dy, float* dw, float* db, float* workspace, int64_t workspace_floats, int B, int H,
                       int W, int Cin, int Cout, int dtype, bool reduce, hipStream_t s) {
  ASR_CHECK_ARG(x && dy && dw && B >= 0 && H > 0 && W > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (Cin % 64 != 0 || Cout % 64 != 0 || !aligned16(x) || !aligned16(dy)) return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  WgradNArgs p{};
  p.x = x; p.dy = dy; p.dw = dw; p.db = db;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.tiles_h = (H + 7) / 8; p.tiles_w = (W + 15) / 16;
  p.npatch = B * p.tiles_h * p.tiles_w;
  p.nci = Cin / 64;
  const int ablate = (int)asr_tuning("WGRAD_ABLATE", 0);
  p.ablate = ablate;
  int wgx, blocks_y;
  asr_conv3x3_wgrad_grid(B, H, W, Cin, Cout, &wgx, &blocks_y, &p.patches_per_wg);
  p.ws = (workspace && workspace_floats >= (int64_t)wgx * blocks_y * 9 * 64 * 64) ? workspace : nullptr;
  const int esz = dtype == ASR_F32 ? 4 : 2;
  const size_t lds = (size_t)(180 + 128) * (64 * esz + 16);
  AsrProfScope prof(ASR_OP_CONV_WGRAD, s);
  // bf16 with a workspace: the LDS-DMA pipelined kernel (conv_wgrad_dma.hip); same grid, same partial-block layout
  const bool dma = asr_tuning("WGRAD_DMA", 1) != 0;
  const int64_t cmax = Cin > Cout ? Cin : Cout;
  if (dma && dtype == ASR_BF16 && p.ws && (int64_t)B * H * W * cmax * 2 < ((int64_t)1 << 32)) {
    WgdArgs q{};
    q.x = static_cast<const bf16_t*>(x); q.dy = static_cast<const bf16_t*>(dy); q.db = db; q.ws = p.ws;
    q.B = B; q.H = H; q.W = W; q.Cin = Cin; q.Cout = Cout; q.tiles_h = p.tiles_h; q.tiles_w = p.tiles_w;
    q.npatch = p.npatch; q.patches_per_wg = p.patches_per_wg; q.nci = p.nci;
    const int rc = asr_conv3x3_wgrad_dma_launch(q, (unsigned)wgx, (unsigned)blocks_y, s);
    if (rc != ASR_OK) return rc;
  } else
  if (dtype == ASR_F32) { allow_big_lds(conv3x3_wgrad_nhwc_kernel<float>, lds); hipLaunchKernelGGL((conv3x3_wgrad_nhwc_kernel<float>), dim3((unsigned)wgx, (unsigned)blocks_y), dim3(256), lds, s, p); }
  else { allow_big_lds(conv3x3_wgrad_nhwc_kernel<bf16_t>, lds); hipLaunchKernelGGL((conv3x3_wgrad_nhwc_kernel<bf16_t>), dim3((unsigned)wgx, (unsigned)blocks_y), dim3(256), lds, s, p); }
  ASR_LAUNCH_CHECK();
  if (p.ws && reduce) {
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(9 * 64 * 64 / 256, (unsigned)blocks_y), dim3(256), 0, s, p.ws, dw, wgx,
                       p.nci, Cin);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}
}  // namespace
