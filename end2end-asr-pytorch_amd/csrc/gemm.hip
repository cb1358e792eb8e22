// NT GEMM on MFMA:  C[M,N] (op)= alpha * sum_k A[m,k] * B[n,k]  (+ bias[n]) (ReLU)
//
// Replaces every nn.Linear / Conv1d(k=1) call on the hot path (reference: models/common_layers.py:136-142,
// :181-187, :197; models/asr/transformer.py:172, :302) and, with explicitly transposed operands, their dgrad
// and wgrad.  Both operands are K-contiguous ("NT"), which is how nn.Linear stores its weight (N,K).
//
// Structure: 256 threads = 4 waves (2x2), tile BMxBN, LDS row = 128 data bytes (+16 pad) per tile row,
// register-staged global->LDS with the next tile's loads issued before the current tile's MFMAs.
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <queue>
#include <utility>
#include <vector>
#include "common.h"
#include "gemm_big.h"

namespace {

struct GemmArgs {
  const void* A; const void* B; void* C; const float* bias; const void* mask;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int k_per_split;     // multiple of BK
  float alpha;
  int relu, accumulate, atomic, vecA, vecB, vecC;
  int tiles_n, ntiles;
  int ablate;          // -DASR_TUNE_ABLATE builds only (tuning "GEMM_ABLATE"): 1 = stage A once, 2 = stage B once (stale operands: timing only)
  // NN, bf16 out (asr_gemm_nn_rowdot; gemm_big.h has the definition): the attention backward's delta from the block that is dO
  const void* dot_o; const float* dot_o32; float* dot_out; int dot_T, dot_H;
};

constexpr int kPitch = 144;   // bytes per LDS tile row: 128 data + 16 pad (keeps 16-B alignment, breaks the 128-B stride)

template <typename T>
__device__ __forceinline__ uint4 load_chunk(const T* row, int64_t k, int64_t kend, bool row_ok, bool vec) {
  Chunk<T> c;
  c.v = make_uint4(0u, 0u, 0u, 0u);
  if (row_ok) {
    if (vec) {
      if (k < kend) c.v = *reinterpret_cast<const uint4*>(row + k);
    } else {
#pragma unroll
      for (int j = 0; j < DT<T>::EPC; ++j)
        if (k + j < kend) c.e[j] = row[k + j];
    }
  }
  return c.v;
}

template <typename TO> __device__ __forceinline__ void store_out(TO* p, float v, int accumulate, int atomic);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v, int accumulate, int atomic) {
  if (atomic) atomicAdd(p, v);
  else if (accumulate) *p += v;
  else *p = v;
}
template <> __device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v, int accumulate, int) {
  if (accumulate) v += bf16_to_f32(*p);
  *p = f32_to_bf16(v);
}

template <typename T, typename TO, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
  constexpr int EPC = DT<T>::EPC;
  constexpr int BK = 128 / (int)sizeof(T);
  constexpr int CA = BM * 8 / 256, CB = BN * 8 / 256;
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;
  unsigned char* sB = smem + BM * kPitch;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x;
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  const int64_t kbeg = (int64_t)blockIdx.z * p.k_per_split;
  const int64_t kend = min((int64_t)p.K, kbeg + p.k_per_split);
  const T* A = static_cast<const T*>(p.A);
  const T* B = static_cast<const T*>(p.B);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 ra[CA], rb[CB];
  auto gload = [&](int64_t k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      int gm = m0 + row;
      ra[i] = load_chunk<T>(A + (int64_t)gm * p.lda, k0 + kc * EPC, kend, gm < p.M, p.vecA);
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      int gn = n0 + row;
      rb[i] = load_chunk<T>(B + (int64_t)gn * p.ldb, k0 + kc * EPC, kend, gn < p.N, p.vecB);
    }
  };
  auto swrite = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(sA + row * kPitch + kc * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      int c = tid + i * 256, row = c >> 3, kc = c & 7;
      *reinterpret_cast<uint4*>(sB + row * kPitch + kc * 16) = rb[i];
    }
  };

  if (kbeg < kend) gload(kbeg);
  for (int64_t k0 = kbeg; k0 < kend; k0 += BK) {
    swrite();
    __syncthreads();
    if (k0 + BK < kend) gload(k0 + BK);
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      uint4 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        a[i] = *reinterpret_cast<const uint4*>(sA + (wm * WM + i * 16 + lr) * kPitch + (ms * 4 + g) * 16);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        b[j] = *reinterpret_cast<const uint4*>(sB + (wn * WN + j * 16 + lr) * kPitch + (ms * 4 + g) * 16);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma16<T>(acc[i][j], a[i], b[j]);
    }
    __syncthreads();
  }

  TO* C = static_cast<TO*>(p.C);
  const T* Msk = static_cast<const T*>(p.mask);
  const bool add_bias = p.bias != nullptr && blockIdx.z == 0;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = n0 + wn * WN + j * 16 + lr;
    if (col >= p.N) continue;
    const float bv = add_bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WM + i * 16 + g * 4 + r;
        if (row < p.M) {
          float v = acc[i][j][r] * p.alpha + bv;
          if (p.relu) v = fmaxf(v, 0.f);
          if (Msk && !(DT<T>::ld(Msk + (int64_t)row * p.ldc + col) > 0.f)) v = 0.f;
          store_out<TO>(C + (int64_t)row * p.ldc + col, v, p.accumulate, p.atomic);
        }
      }
    }
  }
}


// ================================================================================================ fast path
// Direct-to-LDS staging (global_load_lds_dwordx4: no VGPR round trip, one wave instruction = 8 tile rows = 1 KiB),
// LDS image is lane-linear [row][8 x 16 B] with the 16-B slot XOR-swizzled by (row & 7) -- applied on the per-lane
// SOURCE address and again on the fragment read (the destination of an LDS-DMA cannot be permuted), two LDS stages,
// one barrier per K step, XCD-aware tile order, and an epilogue that goes through LDS so that bias / ReLU / mask /
// accumulate / atomics and the global stores are 16-byte row-contiguous.
// Requirements: 16-B aligned operands, lda/ldb multiples of a 16-B chunk, every K range a multiple of BK (128 bytes).
template <int ROWS>
__device__ __forceinline__ void stage_glds(unsigned char* lds_stage, const unsigned char* gbase, int64_t ld_bytes,
                                           int row0, int row_limit, int64_t kbyte0, int tid,
                                           int wave) {
#pragma unroll
  for (int i = 0; i < ROWS * 8 / 256; ++i) {
    const int c = i * 256 + tid, row = c >> 3, slot = (c & 7) ^ (row & 7);
    int gr = row0 + row;
    gr = gr < row_limit ? gr : row_limit - 1;                       // clamp: rows past the edge are never stored
    const unsigned char* src = gbase + (int64_t)gr * ld_bytes + kbyte0 + slot * 16;
    unsigned char* dst = lds_stage + (i * 256 + wave * 64) * 16;    // wave-uniform; the DMA adds lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NS LDS stages (default ONE; 2 / 3 / 4 via ASR_GEMM_NS), NS-1 K steps of LDS-DMA in flight.  A deeper pipeline needs counted
// s_waitcnt vmcnt(N), a RAW s_barrier (__syncthreads() carries a fence that drains every pending LDS-DMA write) and operand
// reads the compiler cannot see (see the asm block below).  Measured (profiles/r01_microbench_v4.txt, MI355X): with all of that
// in place, MORE stages are SLOWER on every shape of this model -- 6400x2048x512: 33.0 / 39.6 / 44 / 55 us for 1 / 2 / 3 / 4
// stages; 3200x4364x512: 37.9 / 47.9 / 53 / 66 us.  The 64x64 tile moves 32 flop per byte through L2, the kernel lives on
// workgroups per CU (8 at one stage), and the other workgroups hide the load latency better than a private prefetch queue.
template <typename T, typename TO, int BM, int BN, int NS>
__global__ __launch_bounds__(256) void gemm_glds_kernel(GemmArgs p) {
  constexpr int ESZ = (int)sizeof(T);
  constexpr int BKB = 128;                       // bytes of K per stage row
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int STAGE = (BM + BN) * BKB;
  constexpr int LPS = (BM + BN) * 8 / 256;       // LDS-DMA instructions per thread per stage
  constexpr int CPITCH = BN * 4 + 16;            // fp32 C tile staged in LDS for the epilogue
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, g = lane >> 4;
  // XCD-aware order: blocks b, b+8, b+16 ... run on the same XCD (private L2) -> give them consecutive tiles
  // The linear work id is (split, tile) with the tile index fastest, so all tiles of one K slice (which share their
  // A and B panels) sit next to each other on one XCD's L2.
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  const int split = wid / p.ntiles, tile = wid % p.ntiles;
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  const int64_t kbeg = (int64_t)split * p.k_per_split;
  const int64_t kend = min((int64_t)p.K, kbeg + p.k_per_split);
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  int nk = (int)((kend - kbeg) * ESZ / BKB);
#ifdef ASR_TUNE_ABLATE
  if (p.ablate & 16) nk = 0;
#endif

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    unsigned char* s = smem + buf * STAGE;
    const int64_t kb = (kbeg * ESZ) + (int64_t)kt * BKB;
#ifdef ASR_TUNE_ABLATE
    if (!(p.ablate & 1) || kt == 0) stage_glds<BM>(s, A, p.lda * ESZ, m0, p.M, kb, tid, wave);
    if (!(p.ablate & 2) || kt == 0) stage_glds<BN>(s + BM * BKB, B, p.ldb * ESZ, n0, p.N, kb, tid, wave);
#else
    stage_glds<BM>(s, A, p.lda * ESZ, m0, p.M, kb, tid, wave);
    stage_glds<BN>(s + BM * BKB, B, p.ldb * ESZ, n0, p.N, kb, tid, wave);
#endif
  };

#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) stage(st, st);
  int buf = 0;                                   // LDS stage of K step kt
  for (int kt = 0; kt < nk; ++kt) {
    if (NS == 1) {                               // one stage: load, wait, compute; the overlap comes from the other workgroups
      if (kt > 0) __builtin_amdgcn_s_barrier(); // everybody is done reading step kt-1
      stage(kt, 0);
    }
    // this thread's DMA of step kt has landed once at most the later steps' loads are outstanding
    const int ahead = min(NS - 2, nk - 1 - kt);
    if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * LPS>();
    else if (NS >= 3 && ahead >= 1) wait_vmcnt<LPS>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();                // step kt visible to all waves; everybody is done reading step kt-1
    if (NS > 1 && kt + NS - 1 < nk) stage(kt + NS - 1, buf == 0 ? NS - 1 : buf - 1);      // refill the stage step kt-1 used
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + BM * BKB;
    if constexpr (FM == 2 && FN == 2 && NS > 2) {
      // Operand fragments by inline asm: for a compiler-visible LDS read the waitcnt pass cannot tell the read apart from the
      // LDS-DMA writes still in flight for later stages and inserts s_waitcnt vmcnt(0) -- which serialises the pipeline again.
      // One block = the 8 reads of this K step + the wait for them; rows i*16 apart share their swizzle slot (offset:2048).
      const int ra = wm * WM + lr, rb = wn * WN + lr;
      const uint32_t aa0 = (uint32_t)(uintptr_t)(sA + ra * BKB + ((g ^ (ra & 7)) << 4));
      const uint32_t aa1 = (uint32_t)(uintptr_t)(sA + ra * BKB + (((4 + g) ^ (ra & 7)) << 4));
      const uint32_t ab0 = (uint32_t)(uintptr_t)(sB + rb * BKB + ((g ^ (rb & 7)) << 4));
      const uint32_t ab1 = (uint32_t)(uintptr_t)(sB + rb * BKB + (((4 + g) ^ (rb & 7)) << 4));
      u32x4_t fa[2][2], fb[2][2];                  // [ms][fragment]
      asm volatile(
          "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:2048\n\t"
          "ds_read_b128 %2, %10\n\tds_read_b128 %3, %10 offset:2048\n\t"
          "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:2048\n\t"
          "ds_read_b128 %6, %11\n\tds_read_b128 %7, %11 offset:2048\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(fa[0][0]), "=&v"(fa[0][1]), "=&v"(fb[0][0]), "=&v"(fb[0][1]), "=&v"(fa[1][0]), "=&v"(fa[1][1]), "=&v"(fb[1][0]),
            "=&v"(fb[1][1])
          : "v"(aa0), "v"(aa1), "v"(ab0), "v"(ab1)
          : "memory");
#pragma unroll
      for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            mma16<T>(acc[i][j], __builtin_bit_cast(uint4, fa[ms][i]), __builtin_bit_cast(uint4, fb[ms][j]));
    } else {
#pragma unroll
      for (int ms = 0; ms < 2; ++ms) {
        uint4 a[FM], b[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int r = wm * WM + i * 16 + lr;
          a[i] = *reinterpret_cast<const uint4*>(sA + r * BKB + (((ms * 4 + g) ^ (r & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int r = wn * WN + j * 16 + lr;
          b[j] = *reinterpret_cast<const uint4*>(sB + r * BKB + (((ms * 4 + g) ^ (r & 7)) << 4));
        }
#ifdef ASR_TUNE_ABLATE
        if (p.ablate & 8) {
#pragma unroll
          for (int i = 0; i < FM; ++i) asm volatile("" :: "v"(a[i].x));
#pragma unroll
          for (int j = 0; j < FN; ++j) asm volatile("" :: "v"(b[j].x));
          continue;
        }
#endif
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma16<T>(acc[i][j], a[i], b[j]);
      }
    }
    buf = buf + 1 == NS ? 0 : buf + 1;
  }
  __syncthreads();                               // all waves are done with the operand stages before the epilogue reuses them

  // ---- epilogue, storage-dtype output with whole 16-byte chunks (every bf16 / fp32-parity activation GEMM of the model):
  // alpha / bias / ReLU on the accumulators, the tile staged in the OUTPUT dtype (half the LDS bytes of an fp32 tile in
  // bf16), then one 16-byte LDS read, mask read, optional read-modify-write and store per EPC columns
  if constexpr (sizeof(TO) == sizeof(T)) {
    constexpr int EPCO = 16 / (int)sizeof(TO);
    if (p.vecC && !p.atomic && !p.accumulate && p.N % EPCO == 0 && p.ldc % EPCO == 0) {     // (+= keeps the single rounding of the fp32 path)
      constexpr int OP = BN * (int)sizeof(TO) + 16;
      const bool add_bias = p.bias != nullptr && split == 0;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = wn * WN + j * 16 + lr;
        const float bv = (add_bias && n0 + col < p.N) ? p.bias[n0 + col] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = acc[i][j][r] * p.alpha + bv;
            if (p.relu) v = fmaxf(v, 0.f);
            *reinterpret_cast<TO*>(smem + (wm * WM + i * 16 + g * 4 + r) * OP + col * sizeof(TO)) = DT<TO>::to(v);
          }
      }
      __syncthreads();
      TO* C = static_cast<TO*>(p.C);
      const T* Msk = static_cast<const T*>(p.mask);
      constexpr int CPRO = BN / EPCO;
      for (int c = tid; c < BM * CPRO; c += 256) {
        const int row = c / CPRO, col = (c % CPRO) * EPCO;
        const int gr = m0 + row, gc = n0 + col;
        if (gr >= p.M || gc >= p.N) continue;
        Chunk<TO> o;
        o.v = *reinterpret_cast<const uint4*>(smem + row * OP + col * sizeof(TO));
        TO* dst = C + (int64_t)gr * p.ldc + gc;
        if (Msk) {
          Chunk<T> m;
          m.v = *reinterpret_cast<const uint4*>(Msk + (int64_t)gr * p.ldc + gc);
#pragma unroll
          for (int e = 0; e < EPCO; ++e)
            if (!(DT<T>::from(m.e[e]) > 0.f)) o.e[e] = DT<TO>::to(0.f);
        }
#ifdef ASR_TUNE_ABLATE
        if (p.ablate & 4) continue;
#endif
        *reinterpret_cast<uint4*>(dst) = o.v;
      }
      return;
    }
  }
  // ---- general epilogue: accumulators -> LDS (fp32, padded rows) -> row-contiguous 16-byte global accesses
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float*>(smem + (wm * WM + i * 16 + g * 4 + r) * CPITCH + (wn * WN + j * 16 + lr) * 4) = acc[i][j][r] * p.alpha;
  __syncthreads();
  TO* C = static_cast<TO*>(p.C);
  const T* Msk = static_cast<const T*>(p.mask);
  const bool add_bias = p.bias != nullptr && split == 0;
  constexpr int CPR = BN / 4;                    // 4-column chunks per tile row
  for (int c = tid; c < BM * CPR; c += 256) {
    const int row = c / CPR, col = (c % CPR) * 4;
    const int gr = m0 + row, gc = n0 + col;
    if (gr >= p.M || gc >= p.N) continue;
    const float4 v4 = *reinterpret_cast<const float4*>(smem + row * CPITCH + col * 4);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    TO* dst = C + (int64_t)gr * p.ldc + gc;
    const int nvalid = min(4, p.N - gc);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e < nvalid) {
        if (add_bias) v[e] += p.bias[gc + e];
        if (p.relu) v[e] = fmaxf(v[e], 0.f);
        if (Msk && !(DT<T>::ld(Msk + (int64_t)gr * p.ldc + gc + e) > 0.f)) v[e] = 0.f;
      }
    }
    if (p.vecC && nvalid == 4 && !p.atomic) {
      if constexpr (sizeof(TO) == 4) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst) = o;
      } else {
        if (p.accumulate) {
          const uint2 old = *reinterpret_cast<const uint2*>(dst);
          v[0] += bf16_to_f32((bf16_t)(old.x & 0xffff)); v[1] += bf16_to_f32((bf16_t)(old.x >> 16));
          v[2] += bf16_to_f32((bf16_t)(old.y & 0xffff)); v[3] += bf16_to_f32((bf16_t)(old.y >> 16));
        }
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(dst) = o;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nvalid) store_out<TO>(dst + e, v[e], p.accumulate, p.atomic);
    }
  }
}

template <typename T, typename TO, int BM, int BN, int NS>
int launch_fast_ns(const GemmArgs& a, int splits, hipStream_t s) {
  GemmArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.ntiles = tiles_m * p.tiles_n;
  dim3 grid((unsigned)(p.ntiles * splits), 1, 1);
  size_t lds = (size_t)NS * (BM + BN) * 128;
  // epilogue staging: the tile in the OUTPUT dtype when the 16-byte-chunk epilogue applies (same test as in the kernel), else fp32.
  // (Sizing it as fp32 always cost the 128 x 128 tile two of its four workgroups per CU.)
  constexpr int EPCO = 16 / (int)sizeof(TO);
  const bool chunked = sizeof(TO) == sizeof(T) && p.vecC && !p.atomic && !p.accumulate && p.N % EPCO == 0 && p.ldc % EPCO == 0;
  const size_t cl = chunked ? (size_t)BM * (BN * sizeof(TO) + 16) : (size_t)BM * (BN * 4 + 16);
  if (cl > lds) lds = cl;
  static size_t granted = 0;
  if (lds > 48 * 1024 && lds > granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_glds_kernel<T, TO, BM, BN, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    granted = lds;
  }
  hipLaunchKernelGGL((gemm_glds_kernel<T, TO, BM, BN, NS>), grid, dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
template <typename T, typename TO, int BM, int BN>
int launch_fast(const GemmArgs& a, int splits, hipStream_t s) {
  // LDS stages: ONE wherever several workgroups share a CU (they cover each other's load latency: the measurement in front of
  // gemm_glds_kernel); a launch of at most NT_RING 64 x 64 blocks leaves a CU with one workgroup or two, and there the private
  // three-stage ring wins (the decoder's 3200 x 512 projections over K = 2048: 32 -> ~14 us; profiles/r03_gemm_nn_ring_ab.txt)
  const int64_t blocks = ceil_div64(a.M, BM) * ceil_div64(a.N, BN) * splits;
  const bool ring = BM == 64 && BN == 64 && sizeof(T) == 2 && a.K >= 256 && blocks <= asr_tuning("NT_RING", 512);
  const int ns = (int)asr_tuning("GEMM_NS", ring ? 3 : 1);
  if (ns == 2) return launch_fast_ns<T, TO, BM, BN, 2>(a, splits, s);
  if (ns == 3) return launch_fast_ns<T, TO, BM, BN, 3>(a, splits, s);
  if (ns == 4) return launch_fast_ns<T, TO, BM, BN, 4>(a, splits, s);
  return launch_fast_ns<T, TO, BM, BN, 1>(a, splits, s);
}

template <typename T, typename TO>
int dispatch_fast(const GemmArgs& a, int splits, hipStream_t s) {
  {                                                            // tuning hook (tools/microbench.py): -1 = automatic
    const int force = (int)asr_tuning("GEMM_TILE", -1);
    if (force == 0) return launch_fast<T, TO, 128, 128>(a, splits, s);
    if (force == 1) return launch_fast<T, TO, 128, 64>(a, splits, s);
    if (force == 2) return launch_fast<T, TO, 64, 64>(a, splits, s);
  }
  // Measured on MI355X (tools/microbench.py, profiles/r01_microbench_v5.txt): 64x64 tiles (8 workgroups per CU) win while the
  // grid is small; once 128x64 tiles still give >= ~1200 workgroups they tie or win (half the B-operand traffic through L2):
  // 6400x2048x512 26.4 vs 27.5 us, 6400x5120x512 57.8 vs 71.6 us.  The fp32-output epilogue (vocabulary logits) prefers 64x64.
  const int64_t t64 = ceil_div64(a.M, 64) * ceil_div64(a.N, 64) * splits;
  if (t64 >= 2400 && a.M > 64 && sizeof(TO) == sizeof(T)) return launch_fast<T, TO, 128, 64>(a, splits, s);
  return launch_fast<T, TO, 64, 64>(a, splits, s);
}


// ================================================================================================ TN (weight gradient)
// C[N,K] += sum_m A[m,n] * B[m,k]  (+ colsum[n] += sum_m A[m,n])  with A = dY (M,N) and B = X (M,K) in their NATURAL
// row-major layouts: the contraction index m is the slow axis of both, so the MFMA operands (8 consecutive m per lane)
// are built with the transposing LDS read ds_read_b64_tr_b16 (bf16) / one 4-byte read per element (fp32) -- no
// transposed copies of the activations are ever written (reference: autograd of every nn.Linear / Conv1d(k=1) weight).
// Workgroup: 64x64 tile of C; a stage holds RM = 4 macro steps of m; wave w contracts macro step w of every stage against
// the full 64x64 tile (16 operand reads feed 16 MFMAs), the four partial tiles are summed through LDS at the end.
struct TnArgs {
  const void* A; const void* B; float* C; float* colsum;
  float* ws;     // split over m with a workspace: partial 64x64 tiles [split][tile][64][64], folded by tn_reduce_kernel
  int64_t lda, ldb, ldc;
  int M, N, K, m_per_split, tiles_k, ntiles;
};

template <typename T> struct TnPack;
template <> struct TnPack<bf16_t> {
  static constexpr int RM = 128, ROWB = 128, CPR = 8;
  // 8 consecutive rows m = m0 + 8g .. +7 of column c0 + lr  (m0 = first row of this wave's macro step)
  template <bool GMAJOR = false>
  static __device__ __forceinline__ uint4 load(const unsigned char* tile, int m0, int lr, int g, int c0) {
    const int row = m0 + 8 * g + (lr >> 2), col = c0 + 4 * (lr & 3);          // this lane SUPPLIES 4 columns of one row
    const int chunk = col >> 3, half = (col >> 2) & 1;
    const uint2 lo = asr_lds_read_tr16(tile + row * ROWB + ((chunk ^ (row & 7)) << 4) + half * 8);
    const uint2 hi = asr_lds_read_tr16(tile + (row + 4) * ROWB + ((chunk ^ ((row + 4) & 7)) << 4) + half * 8);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
  }
};
template <> struct TnPack<float> {
  static constexpr int RM = 64, ROWB = 256, CPR = 16;
  // pack element s feeds the s-th 16x16x4 MFMA, lane group g is its k index.  GMAJOR = false: row = m0 + 4s + g (both
  // operands come from this loader); GMAJOR = true: row = m0 + 4g + s, the k order of an operand read as one aligned 16-byte
  // chunk of 4 consecutive k (the A side of the NN kernel) -- the two operands of an MFMA must agree on k.
  template <bool GMAJOR = false>
  static __device__ __forceinline__ uint4 load(const unsigned char* tile, int m0, int lr, int g, int c0) {
    const int col = c0 + lr, chunk = col >> 2, sub = (col & 3) * 4;
    uint4 r;
    uint32_t* rr = &r.x;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int row = m0 + (GMAJOR ? 4 * g + s : 4 * s + g);
      rr[s] = *reinterpret_cast<const uint32_t*>(tile + row * ROWB + ((chunk ^ (row & 7)) << 4) + sub);
    }
    return r;
  }
};

template <typename T, int NBUF>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnArgs p) {
  using P = TnPack<T>;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int STAGE = 2 * P::RM * P::ROWB;            // A tile + B tile
  constexpr int MS_ROWS = P::RM / 4;                    // rows of one macro step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  const int split = wid / p.ntiles, tile = wid % p.ntiles;
  const int n0 = (tile / p.tiles_k) * 64, k0 = (tile % p.tiles_k) * 64;
  const int m_beg = split * p.m_per_split, m_end = min(p.M, m_beg + p.m_per_split);
  const int nstage = (m_end - m_beg + P::RM - 1) / P::RM;   // the last stage of the last split may be partial
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  // column chunk (16 B) this tile may read: clamp to the operand's last whole chunk (columns past N / K are never stored)
  // (ldb < K: the rows of B are overlapping windows over one longer buffer -- the clamp is then the window's own width)
  const int a_chunks = (int)(p.lda * ESZ / 16), b_chunks = (int)((p.ldb >= p.K ? p.ldb : (int64_t)((p.K + 7) / 8 * 8)) * ESZ / 16);

  // per-thread byte offsets of its DMA chunks relative to the first row of a stage (the stage base is workgroup-uniform: the
  // loads of a whole stage then cost one scalar base update instead of ~12 vector ops of address arithmetic per chunk)
  constexpr int NIT = P::RM * P::CPR / 256;
  unsigned offA[NIT], offB[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = i * 256 + tid, row = c / P::CPR, slot = (c % P::CPR) ^ (row & 7);
    int ca = n0 * ESZ / 16 + slot; ca = ca < a_chunks ? ca : a_chunks - 1;
    int cb = k0 * ESZ / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
    offA[i] = (unsigned)(row * (int)p.lda * ESZ + ca * 16);
    offB[i] = (unsigned)(row * (int)p.ldb * ESZ + cb * 16);
  }
  auto stage = [&](int st, int buf) __attribute__((always_inline)) {
    unsigned char* s = smem + buf * STAGE;
    const int64_t mrow = m_beg + (int64_t)st * P::RM;
    if (mrow + P::RM <= p.M) {            // every row of the stage exists
      const unsigned char* ba = A + mrow * p.lda * ESZ;
      const unsigned char* bb = B + mrow * p.ldb * ESZ;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        unsigned char* d = s + (i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ba + offA[i]),
                                         (__attribute__((address_space(3))) void*)d, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bb + offB[i]),
                                         (__attribute__((address_space(3))) void*)(d + P::RM * P::ROWB), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < P::RM * P::CPR / 256; ++i) {
      const int c = i * 256 + tid, row = c / P::CPR, slot = (c % P::CPR) ^ (row & 7);
      int ca = n0 * ESZ / 16 + slot; ca = ca < a_chunks ? ca : a_chunks - 1;
      int cb = k0 * ESZ / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
      // rows past M: re-read the last valid row (finite data); the A side of those rows is zeroed in LDS before use
      const int64_t gr = mrow + row < p.M ? mrow + row : (int64_t)p.M - 1;
      const unsigned char* sa = A + gr * p.lda * ESZ + (int64_t)ca * 16;
      const unsigned char* sb = B + gr * p.ldb * ESZ + (int64_t)cb * 16;
      unsigned char* d = s + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(d + P::RM * P::ROWB), 16, 0, 0);
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = p.colsum != nullptr && k0 == 0;

  // NBUF == 2: the next stage's LDS-DMA overlaps this stage's MFMAs.  NBUF == 1: load, wait, compute -- half the LDS, twice the
  // workgroups per CU, and the overlap comes from the other workgroups (measured faster, like every occupancy trade here).
  if (NBUF == 2) {
    if (nstage > 0) stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int st = 0; st < nstage; ++st) {
    if (NBUF == 2) {
      if (st + 1 < nstage) stage(st + 1, (st + 1) & 1);
    } else {
      stage(st, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    const int cur = NBUF == 2 ? (st & 1) : 0;
    const unsigned char* sA = smem + cur * STAGE;
    const unsigned char* sB = sA + P::RM * P::ROWB;
    const int valid = m_end - (m_beg + st * P::RM);          // rows of this stage that exist (uniform over the workgroup)
    if (valid < P::RM) {
      unsigned char* zA = smem + cur * STAGE;
      for (int c = valid * (P::ROWB / 16) + tid; c < P::RM * (P::ROWB / 16); c += 256)
        *reinterpret_cast<uint4*>(zA + c * 16) = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = P::load(sA, wave * MS_ROWS, lr, g, i * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = P::load(sB, wave * MS_ROWS, lr, g, j * 16);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Chunk<T> c; c.v = a[i];
#pragma unroll
        for (int e = 0; e < DT<T>::EPC; ++e) bsum[i] += DT<T>::from(c.e[e]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma16<T>(acc[i][j], a[i], b[j]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- sum the four waves' partial tiles (tree, in fragment layout: lane-private slots, no index math), then wave 0 lays the
  // total out row-major for 16-byte row-contiguous accumulation into C.  32 KB of LDS instead of four 17 KB tiles.
  constexpr int CP = 64 * 4 + 16;
  float* slot = reinterpret_cast<float*>(smem);                 // [2][64 values][64 lanes]
  float (*s_col)[64] = reinterpret_cast<float (*)[64]>(smem + 2 * 64 * 64 * 4);
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) s_col[wave][i * 16 + lr] = v;
    }
  }
  auto put = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) slot[(sl * 64 + (i * 4 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
  };
  auto add = [&](int sl) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] += slot[(sl * 64 + (i * 4 + j) * 4 + r) * 64 + lane];
  };
  if (wave & 1) put(wave >> 1);
  __syncthreads();
  if (!(wave & 1)) add(wave >> 1);
  __syncthreads();
  if (wave == 2) put(0);
  __syncthreads();
  if (wave == 0) add(0);
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          *reinterpret_cast<float*>(smem + (i * 16 + g * 4 + r) * CP + (j * 16 + lr) * 4) = acc[i][j][r];
  }
  __syncthreads();
  const bool single = gridDim.x == (unsigned)p.ntiles;       // no split over m: plain read-modify-write
  if (!single && p.ws) {
    // two-stage reduction: the partial tile goes to the workspace with plain coalesced stores (fp32 atomics on C cost more
    // than the MFMAs of a split), tn_reduce_kernel adds the slices into C
    float* part = p.ws + ((int64_t)split * p.ntiles + tile) * 4096;
    for (int c = tid; c < 64 * 16; c += 256) {
      const int row = c >> 4, col = (c & 15) * 4;
      *reinterpret_cast<float4*>(part + row * 64 + col) = *reinterpret_cast<const float4*>(smem + row * CP + col * 4);
    }
  } else
  for (int c = tid; c < 64 * 16; c += 256) {
    const int row = c >> 4, col = (c & 15) * 4;
    const int gn = n0 + row, gk = k0 + col;
    if (gn >= p.N || gk >= p.K) continue;
    const float4 v = *reinterpret_cast<const float4*>(smem + row * CP + col * 4);
    float* dst = p.C + (int64_t)gn * p.ldc + gk;
    const int nv = min(4, p.K - gk);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    if (single && nv == 4 && ((((uintptr_t)dst) & 15) == 0)) {
      float4 o = *reinterpret_cast<float4*>(dst);
      o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
      *reinterpret_cast<float4*>(dst) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nv) { if (single) dst[e] += vv[e]; else atomicAdd(dst + e, vv[e]); }
    }
  }
  if (do_colsum && tid < 64 && n0 + tid < p.N)
    atomicAdd(p.colsum + n0 + tid, s_col[0][tid] + s_col[1][tid] + s_col[2][tid] + s_col[3][tid]);
}


// C tile += sum over the m-slices of the partial tiles written by gemm_tn_kernel
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ ws, float* C, int64_t ldc, int N, int K,
                                                        int ntiles, int tiles_k, int splits) {
  const int tile = blockIdx.x >> 2;
  const int e = ((blockIdx.x & 3) * 256 + threadIdx.x) * 4;        // 4 consecutive columns of one row of the tile
  const int row = e >> 6, col = e & 63;
  const int gn = (tile / tiles_k) * 64 + row, gk = (tile % tiles_k) * 64 + col;
  if (gn >= N || gk >= K) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < splits; ++sp) {
    const float4 t = *reinterpret_cast<const float4*>(ws + ((int64_t)sp * ntiles + tile) * 4096 + e);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  float* dst = C + (int64_t)gn * ldc + gk;
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (gk + i < K) dst[i] += vv[i];
}

// ------------------------------------------------------------------------------------------------ TN, 128 x 128 tiles (bf16)
// Same contraction for the larger weight gradients.  The 64x64-tile kernel above moves 32 flop per byte through L2 (12 TB/s
// on the 2048x512 FFN gradient): this one owns a 128 x 128 block of dW per workgroup (64 flop per byte), waves in a 2 x 2 grid
// of 64 x 64 quadrants (each wave contracts EVERY row of a stage: no cross-wave reduction, the partial block goes from the
// accumulators straight to the workspace / C), 64 rows of m per stage (one LDS stage of 32 KB).
struct Tn128Args {
  const void* A; const void* B; float* C; float* colsum; float* ws;
  int64_t lda, ldb, ldc;
  int M, N, K, m_per_split, tiles_k, ntiles;
};

__device__ __forceinline__ uint4 tn128_pack(const unsigned char* tile, int m0, int lr, int g, int c0) {
  // 8 consecutive rows m0 + 8g .. +7 of column c0 + lr of a [64][128] bf16 tile (256-byte rows, chunk c of row r in slot
  // c ^ (r & 7): the XOR permutes the low 3 bits of the 4-bit chunk index)
  const int row = m0 + 8 * g + (lr >> 2), col = c0 + 4 * (lr & 3);
  const int chunk = col >> 3, half = (col >> 2) & 1;
  const uint2 lo = asr_lds_read_tr16(tile + row * 256 + ((chunk ^ (row & 7)) << 4) + half * 8);
  const uint2 hi = asr_lds_read_tr16(tile + (row + 4) * 256 + ((chunk ^ ((row + 4) & 7)) << 4) + half * 8);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <int RM>
__global__ __launch_bounds__(256) void gemm_tn128_kernel(Tn128Args p) {
  constexpr int ROWB = 256, TILEB = RM * ROWB;               // one operand tile of a stage: 16 KB (RM = 64) / 32 KB (128)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  const int split = wid / p.ntiles, tile = wid % p.ntiles;
  const int n0 = (tile / p.tiles_k) * 128, k0 = (tile % p.tiles_k) * 128;
  const int m_beg = split * p.m_per_split, m_end = min(p.M, m_beg + p.m_per_split);
  const int nstage = (m_end - m_beg + RM - 1) / RM;
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  const int a_chunks = (int)(p.lda * 2 / 16), b_chunks = (int)((p.ldb >= p.K ? p.ldb : (int64_t)((p.K + 7) / 8 * 8)) * 2 / 16);

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = p.colsum != nullptr && k0 == 0 && wk == 0;

  for (int st = 0; st < nstage; ++st) {
    if (st > 0) __syncthreads();                 // everybody is done reading the previous stage
    const int64_t mrow = m_beg + (int64_t)st * RM;
#pragma unroll
    for (int i = 0; i < RM * 16 / 256; ++i) {
      const int c = i * 256 + tid, row = c >> 4, slot = (c & 15) ^ (row & 7);
      int ca = n0 * 2 / 16 + slot; ca = ca < a_chunks ? ca : a_chunks - 1;       // columns past N / K are never stored
      int cb = k0 * 2 / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
      const int64_t gr = mrow + row < p.M ? mrow + row : (int64_t)p.M - 1;       // rows past M: see the zero fill below
      unsigned char* d = smem + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + gr * p.lda * 2 + (int64_t)ca * 16),
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(B + gr * p.ldb * 2 + (int64_t)cb * 16),
                                       (__attribute__((address_space(3))) void*)(d + TILEB), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int valid = m_end - (m_beg + st * RM);
    if (valid < RM) {                            // partial last stage: zero the A rows that do not exist (B holds finite data)
      for (int c = valid * 16 + tid; c < RM * 16; c += 256) *reinterpret_cast<uint4*>(smem + c * 16) = make_uint4(0u, 0u, 0u, 0u);
      __syncthreads();
    }
    const unsigned char* sA = smem;
    const unsigned char* sB = smem + TILEB;
#pragma unroll
    for (int ms = 0; ms < RM / 32; ++ms) {
      uint4 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tn128_pack(sA, ms * 32, lr, g, wn * 64 + i * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = tn128_pack(sB, ms * 32, lr, g, wk * 64 + j * 16);
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          Chunk<bf16_t> c; c.v = a[i];
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[i] += bf16_to_f32(c.e[e]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) mma16<bf16_t>(acc[i][j], a[i], b[j]);
    }
  }

  // ---- the wave's 64 x 64 quadrant: lane (lr, g) holds rows 4g..4g+3 of column lr of every fragment
  const bool single = gridDim.x == (unsigned)p.ntiles;
  float* part = p.ws ? p.ws + ((int64_t)split * p.ntiles + tile) * 16384 : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wn * 64 + i * 16 + g * 4 + r, gn = n0 + row;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wk * 64 + j * 16 + lr, gk = k0 + col;
        const float v = acc[i][j][r];
        if (!single && part) part[row * 128 + col] = v;
        else if (gn < p.N && gk < p.K) {
          float* dst = p.C + (int64_t)gn * p.ldc + gk;
          if (single) *dst += v; else atomicAdd(dst, v);
        }
      }
    }
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int gn = n0 + wn * 64 + i * 16 + lr;
      if (g == 0 && gn < p.N) atomicAdd(p.colsum + gn, v);
    }
  }
}

// ---- pipelined variant: the same 128 x 128 block and wave layout, but RM = 32 rows of m per stage in an NST-deep LDS ring filled
// by HAND-ISSUED LDS-DMA (inline asm: the compiler must not know a DMA is in flight, or it drains the VMEM counter in front of
// every transposing read it can see -- the reason the single-stage kernel above cannot overlap its loads), counted
// s_waitcnt vmcnt, raw s_barrier.  One barrier per stage; the DMA of stage st + NST - 1 is issued right behind the barrier of
// stage st (the ring slot it overwrites was read at stage st - 1, which every wave has left).  A partial last stage reads the
// missing rows of A from a 16-byte zero page (the DMA cannot zero fill).  64 KB of LDS at NST = 4: two workgroups per CU.
__device__ const uint4 tn_zero_page = {0u, 0u, 0u, 0u};

__device__ __forceinline__ void tn_dma(unsigned lds_wave_base, const unsigned char* src) {
  unsigned keep;      // M0 saved and restored: neutral for whatever the compiler keeps there
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}
__device__ __forceinline__ uint4 tn32_pack(const unsigned char* tile, int lr, int g, int c0) {
  // 8 consecutive rows 8g .. 8g+7 of column c0 + lr of a [32][128] bf16 tile (256-byte rows, chunk c of row r in slot c ^ (r & 7))
  const int row = 8 * g + (lr >> 2), col = c0 + 4 * (lr & 3);
  const int chunk = col >> 3, half = (col >> 2) & 1;
  const uint2 lo = asr_lds_read_tr16(tile + row * 256 + ((chunk ^ (row & 7)) << 4) + half * 8);
  const uint2 hi = asr_lds_read_tr16(tile + (row + 4) * 256 + ((chunk ^ ((row + 4) & 7)) << 4) + half * 8);
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

template <int NST>
__device__ __forceinline__ void tn128p_body(const Tn128Args& p, unsigned char* smem, int wid, bool single) {
  constexpr int RM = 32, ROWB = 256, TILEB = RM * ROWB, STAGEB = 2 * TILEB;       // 8 KB per operand, 16 KB per stage
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int split = wid / p.ntiles, tile = wid % p.ntiles;
  const int n0 = (tile / p.tiles_k) * 128, k0 = (tile % p.tiles_k) * 128;
  const int m_beg = split * p.m_per_split, m_end = min(p.M, m_beg + p.m_per_split);
  const int nstage = (m_end - m_beg + RM - 1) / RM;
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  const int a_chunks = (int)(p.lda * 2 / 16), b_chunks = (int)((p.ldb >= p.K ? p.ldb : (int64_t)((p.K + 7) / 8 * 8)) * 2 / 16);

  // per-thread DMA pieces: 512 chunks per operand tile = 2 per thread; chunk c = (row, slot), source chunk slot ^ (row & 7)
  int64_t offA[2], offB[2];
  int rowi[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 256 + tid, row = c >> 4, slot = (c & 15) ^ (row & 7);
    int ca = n0 * 2 / 16 + slot; ca = ca < a_chunks ? ca : a_chunks - 1;       // columns past N / K are never stored
    int cb = k0 * 2 / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
    offA[i] = (int64_t)row * p.lda * 2 + (int64_t)ca * 16;
    offB[i] = (int64_t)row * p.ldb * 2 + (int64_t)cb * 16;
    rowi[i] = row;
  }
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wave_lds = smem_base + (unsigned)wave * 1024u;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(&tn_zero_page);
  auto stage = [&](int st) __attribute__((always_inline)) {
    const unsigned sl = wave_lds + (unsigned)((st % NST) * STAGEB);
    const int64_t mrow = m_beg + (int64_t)st * RM;
    const unsigned char* ba = A + mrow * p.lda * 2;
    const unsigned char* bb = B + mrow * p.ldb * 2;
    const int valid = m_end - (int)mrow;                  // rows of this stage that exist (uniform)
    if (valid >= RM) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        tn_dma(sl + i * 4096, ba + offA[i]);
        tn_dma(sl + TILEB + i * 4096, bb + offB[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool in = rowi[i] < valid;
        tn_dma(sl + i * 4096, in ? ba + offA[i] : zero);
        tn_dma(sl + TILEB + i * 4096, in ? bb + offB[i] : zero);
      }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = p.colsum != nullptr && k0 == 0 && wk == 0;

#pragma unroll
  for (int st = 0; st < NST - 1; ++st)
    if (st < nstage) stage(st);
  for (int st = 0; st < nstage; ++st) {
    // stage st has landed once at most the DMA pieces of the later stages are outstanding (4 pieces per stage and thread, in order)
    const int ahead = min(NST - 2, nstage - 1 - st);
    if (ahead >= 2) wait_vmcnt<8>();
    else if (ahead == 1) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();              // stage st visible to every wave; every wave is done reading stage st - 1
    asm volatile("" ::: "memory");
    if (st + NST - 1 < nstage) stage(st + NST - 1);
    const unsigned char* sA = smem + (st % NST) * STAGEB;
    const unsigned char* sB = sA + TILEB;
    uint4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = tn32_pack(sA, lr, g, wn * 64 + i * 16);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = tn32_pack(sB, lr, g, wk * 64 + j * 16);
    if (do_colsum) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        Chunk<bf16_t> c; c.v = a[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i] += bf16_to_f32(c.e[e]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mma16<bf16_t>(acc[i][j], a[i], b[j]);
  }

  // ---- the wave's 64 x 64 quadrant: lane (lr, g) holds rows 4g..4g+3 of column lr of every fragment
  float* part = p.ws ? p.ws + ((int64_t)split * p.ntiles + tile) * 16384 : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wn * 64 + i * 16 + g * 4 + r, gn = n0 + row;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = wk * 64 + j * 16 + lr, gk = k0 + col;
        const float v = acc[i][j][r];
        if (!single && part) part[row * 128 + col] = v;
        else if (gn < p.N && gk < p.K) {
          float* dst = p.C + (int64_t)gn * p.ldc + gk;
          if (single) *dst += v; else atomicAdd(dst, v);
        }
      }
    }
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int gn = n0 + wn * 64 + i * 16 + lr;
      if (g == 0 && gn < p.N) atomicAdd(p.colsum + gn, v);
    }
  }
}

template <int NST>
__global__ __launch_bounds__(256, 2) void gemm_tn128p_kernel(Tn128Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  tn128p_body<NST>(p, smem, wid, gridDim.x == (unsigned)p.ntiles);
}

// ---- 256 x 256 blocks of dW, eight waves (4 x 2 grid of 64 x 128 quadrants), 32 rows of m per stage in an NST-deep ring.
// Twice the flop per staged byte of the 128 x 128 block (128 instead of 64): the grouped launch below has ~500 of them in flight
// with a contraction 1600 - 6400 rows long, so the block lives in its steady state, and what bounds the 128 x 128 form there is the
// operand traffic per MFMA (4 LDS-DMA pieces and 16 transposing reads per 16 MFMAs; here 4 pieces and 24 reads per 32).
// LDS image of an operand tile: [32 rows][32 chunks of 16 B]; chunk c of row r sits in slot c ^ key(r), key(r) = ((r & 3) | ((r >> 3)
// & 1) << 2) << 1: the 8 rows one transposing read touches (r0 .. r0+3 and r0+8 .. r0+11, 32 bytes each) land on 8 different
// 32-byte bank groups.
__device__ __forceinline__ int tn256_key(int r) { return ((r & 3) | (((r >> 3) & 1) << 2)) << 1; }
// (Round 3's loop over the same LDS image, tn256_body, was removed in round 6 with its switches TN_ROT / TN_GROUP_STAGES / TN_ROT_NST:
// this loop gives the same bits -- same DMA, same MFMA order per accumulator -- and won every measurement, profiles/r05_tn_grouped.txt.)
typedef __attribute__((ext_vector_type(2))) unsigned int tn_u32x2_t;
__device__ __forceinline__ tn_u32x2_t tn_tr_read(unsigned addr, const int off2048) {
  tn_u32x2_t v;
  if (off2048) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(v) : "v"(addr));
  else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
#define TN_FRAG_READ(F, ADDR) { const tn_u32x2_t lo_ = tn_tr_read(ADDR, 0), hi_ = tn_tr_read(ADDR, 1); F = u32x4_t{lo_.x, lo_.y, hi_.x, hi_.y}; }
template <int N> __device__ __forceinline__ void tn_wait_lgkm(u32x4_t& x, u32x4_t& y) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "n"(N));
}
// SWAP: the fragment comes out transposed (k rows, n columns) -- a lane then holds four CONSECUTIVE k of one row of dW
template <bool SWAP> __device__ __forceinline__ void tn_mma2(f32x4_t& acc, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (SWAP) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, b), __builtin_bit_cast(bf16x8_t, a), acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

#ifdef TN_TIMING        // tuning builds only (ASR_HIPCC_EXTRA=-DTN_TIMING): s_memtime stamps at the section boundaries of a stage, workgroup 0
__device__ long long tn_dbg[64];
#define TN_STAMP(K) { const long long now_ = clock64(); tsec[K] += now_ - tlast; tlast = now_; }
#else
#define TN_STAMP(K)
#endif
template <int NST, bool SWAP>
__device__ __forceinline__ void tn256r_body(const Tn128Args& p, unsigned char* smem, int tile, int m_beg, int m_end, bool single) {
  static_assert(NST == 3 || NST == 4, "ring of three or four stages");
  constexpr int RM = 32, TILEB = RM * 512, STAGEB = 2 * TILEB;
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  // every argument the loop touches, in registers before it
  const int64_t lda2 = p.lda * 2, ldb2 = p.ldb * 2;
  const int pM = p.M, pN = p.N, pK = p.K;
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  float* const C = p.C;
  float* const colsum = p.colsum;
  const int64_t ldc = p.ldc;
  const int n0 = (tile / p.tiles_k) * 256, k0 = (tile % p.tiles_k) * 256;
  const int nstage = (m_end - m_beg + RM - 1) / RM;
  const int a_chunks = (int)(lda2 / 16), b_chunks = (int)((p.ldb >= pK ? p.ldb : (int64_t)((pK + 7) / 8 * 8)) * 2 / 16);
  asm volatile("" ::"s"(lda2), "s"(ldb2), "s"(pM), "s"(pN), "s"(pK), "s"(A), "s"(B), "s"(C), "s"(colsum), "s"(ldc), "s"(n0), "s"(k0), "s"(nstage));

  int64_t offA[2], offB[2];
  int rowi[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 512 + tid, row = c >> 5, slot = (c & 31) ^ tn256_key(row);
    int ca = n0 * 2 / 16 + slot; ca = ca < a_chunks ? ca : a_chunks - 1;
    int cb = k0 * 2 / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
    offA[i] = (int64_t)row * lda2 + (int64_t)ca * 16;
    offB[i] = (int64_t)row * ldb2 + (int64_t)cb * 16;
    rowi[i] = row;
  }
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wave_lds = smem_base + (unsigned)wave * 1024u;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(&tn_zero_page);
  const unsigned char* ba = A + (int64_t)m_beg * lda2;        // rows of the stage being staged next: advanced by 32 rows per call
  const unsigned char* bb = B + (int64_t)m_beg * ldb2;
  int staged = 0;                                              // stages handed to the DMA so far
  unsigned ring_w = 0;                                         // ring position (bytes) of the next stage to stage
  auto stage = [&]() __attribute__((always_inline)) {
    const unsigned sl = wave_lds + ring_w;
    const int valid = m_end - m_beg - staged * RM;
    if (valid >= RM) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        tn_dma(sl + i * 8192, ba + offA[i]);
        tn_dma(sl + TILEB + i * 8192, bb + offB[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool in = rowi[i] < valid;
        tn_dma(sl + i * 8192, in ? ba + offA[i] : zero);
        tn_dma(sl + TILEB + i * 8192, in ? bb + offB[i] : zero);
      }
    }
    ba += RM * lda2; bb += RM * ldb2;
    ++staged;
    ring_w = ring_w == (unsigned)(NST - 1) * STAGEB ? 0u : ring_w + STAGEB;
  };

  // fragment read offsets inside a stage: row 8 g + (lr >> 2) (+ 4: offset 2048), column c0 + 4 (lr & 3)
  const int frow = 8 * g + (lr >> 2), fkey = tn256_key(frow);
  unsigned fa[4], fb[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = wn * 64 + i * 16 + 4 * (lr & 3);
    fa[i] = smem_base + (unsigned)(frow * 512 + (((col >> 3) ^ fkey) << 4) + ((col >> 2) & 1) * 8);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int col = wk * 128 + j * 16 + 4 * (lr & 3);
    fb[j] = smem_base + (unsigned)(TILEB + frow * 512 + (((col >> 3) ^ fkey) << 4) + ((col >> 2) & 1) * 8);
  }

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // column sums of dY (the bias gradient) from the A fragments of the k0 = 0 blocks.  Round 6: four v_dot2c per fragment (asr_sum8_bf16)
  // instead of a chain of 16 dependent shift / mask / add, and the four fragments of a row band split between the two waves that read them
  // (wk = 0: fragments 0, 1; wk = 1: 2, 3).  In-kernel section timing (-DTN_TIMING) had the summing wave as the straggler of every stage of
  // such a block: 1 745 cycles between the barriers against 1 050 for a wave that only multiplies.
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = colsum != nullptr && k0 == 0;
  auto addsum = [&](int i, const u32x4_t& v) __attribute__((always_inline)) {
    if ((i >> 1) == wk) asr_sum8_bf16(bsum[i], __builtin_bit_cast(bf16x8_t, v));
  };

  // prologue: stages 0 .. 2 on their way, stage 0 landed and published, its fragments requested
  for (int st = 0; st < NST && st < nstage; ++st) stage();
  if (NST == 4 && nstage >= 4) wait_vmcnt<12>(); else if (nstage >= 3) wait_vmcnt<8>(); else if (nstage == 2) wait_vmcnt<4>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  u32x4_t a[4], b[8];
  unsigned ring_r = 0;                                         // ring position of the stage whose fragments are requested next
#pragma unroll
  for (int i = 0; i < 3; ++i) TN_FRAG_READ(a[i], fa[i] + ring_r)
#pragma unroll
  for (int j = 0; j < 8; ++j) TN_FRAG_READ(b[j], fb[j] + ring_r)
  TN_FRAG_READ(a[3], fa[3] + ring_r)
  ring_r = STAGEB;

#ifdef TN_TIMING
  long long tsec[6] = {0, 0, 0, 0, 0, 0}, tlast = clock64();
#endif
  for (int st = 0; st < nstage; ++st) {
    TN_STAMP(5)
    // ---- row 0: the only waits of the stage.  Requests in flight, oldest first: a0 a1 a2 b0 .. b7 a3 (two reads each)
    tn_wait_lgkm<15>(a[0], b[0]);
    if (do_colsum) addsum(0, a[0]);
    tn_mma2<SWAP>(acc[0][0], a[0], b[0]);
    tn_wait_lgkm<14>(a[0], b[1]); tn_mma2<SWAP>(acc[0][1], a[0], b[1]);
    tn_wait_lgkm<12>(a[0], b[2]); tn_mma2<SWAP>(acc[0][2], a[0], b[2]);
    tn_wait_lgkm<10>(a[0], b[3]); tn_mma2<SWAP>(acc[0][3], a[0], b[3]);
    tn_wait_lgkm<8>(a[0], b[4]);  tn_mma2<SWAP>(acc[0][4], a[0], b[4]);
    tn_wait_lgkm<6>(a[0], b[5]);  tn_mma2<SWAP>(acc[0][5], a[0], b[5]);
    tn_wait_lgkm<4>(a[0], b[6]);  tn_mma2<SWAP>(acc[0][6], a[0], b[6]);
    tn_wait_lgkm<2>(a[0], b[7]);  tn_mma2<SWAP>(acc[0][7], a[0], b[7]);
    // ---- stage st + 1 published (every wave's pieces landed), stage st's buffer free (every wave's reads of it returned)
    TN_STAMP(0)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    TN_STAMP(1)
    if (NST == 4 && st + 3 < nstage) wait_vmcnt<8>(); else if (st + 2 < nstage) wait_vmcnt<4>(); else wait_vmcnt<0>();
    TN_STAMP(2)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    TN_STAMP(3)
    if (staged < nstage) stage();                              // stage st + NST into the buffer of stage st
    // (past the last stage the requests below fetch stale bytes of the ring that nobody uses)
    TN_FRAG_READ(a[0], fa[0] + ring_r)
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      if (do_colsum) addsum(i, a[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        tn_mma2<SWAP>(acc[i][j], a[i], b[j]);
        if (i == 3) TN_FRAG_READ(b[j], fb[j] + ring_r)
      }
      TN_FRAG_READ(a[i], fa[i] + ring_r)
    }
    ring_r = ring_r == (unsigned)(NST - 1) * STAGEB ? 0u : ring_r + STAGEB;
    TN_STAMP(4)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the stale requests of the last stage: the next piece rewrites the ring
#ifdef TN_TIMING
  if (blockIdx.x == 0 && lane == 0) {
    for (int k = 0; k < 6; ++k) tn_dbg[wave * 8 + k] = tsec[k];
    tn_dbg[wave * 8 + 7] = nstage;
  }
#endif

  if constexpr (!SWAP) {
    // ---- epilogue of the shared-block forms (equal pieces, slices): lane (lr, g) holds rows 4g .. 4g+3 of column lr of a fragment --
    // an instruction's 64 lanes touch 4 runs of 64 contiguous bytes, which is what the fp32 atomics want
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gn = n0 + wn * 64 + i * 16 + g * 4 + r;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int gk = k0 + wk * 128 + j * 16 + lr;
          if (gn < pN && gk < pK) {
            float* dst = C + (int64_t)gn * ldc + gk;
            if (single) *dst += acc[i][j][r]; else atomicAdd(dst, acc[i][j][r]);
          }
        }
      }
  } else {
  // ---- epilogue.  The MFMAs ran with the operands swapped (k rows, n columns): lane (lr, g) holds, of fragment (i, j), the FOUR
  // consecutive k = k0 + wk 128 + 16 j + 4 g .. + 3 of row n = n0 + wn 64 + 16 i + lr -- 16 contiguous bytes of dW
  // A block with one owner: plain read-modify-write, the 8 loads of a fragment row issued TOGETHER (the compiler cannot prove that
  // dst(i, j) and dst(i', j') differ -- ldc is a run-time value -- and would otherwise wait for every load behind the previous store:
  // 32 dependent round trips to L2 / HBM per block).  A sliced block: fp32 atomics without a return value (nothing to wait for).
  const bool vec = (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && (pK & 3) == 0;
  const bool full = n0 + 256 <= pN && k0 + 256 <= pK;          // no edge inside the block: no per-element tests (wave-uniform)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gn = n0 + wn * 64 + i * 16 + lr;
    const int gk0 = k0 + wk * 128 + g * 4;
    float* const row = C + (int64_t)gn * ldc + gk0;
    if (full && vec && single) {
      f32x4_t old[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) old[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(row + j * 16));
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4_t*>(row + j * 16) = old[j] + acc[i][j];
    } else if (full && !single) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(row + j * 16 + r, acc[i][j][r]);
    } else if (gn < pN) {
      if (single && vec) {
        f32x4_t old[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (gk0 + j * 16 < pK) old[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(row + j * 16));
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (gk0 + j * 16 < pK) *reinterpret_cast<f32x4_t*>(row + j * 16) = old[j] + acc[i][j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (gk0 + j * 16 + r < pK) {
              if (single) row[j * 16 + r] += acc[i][j][r];
              else atomicAdd(row + j * 16 + r, acc[i][j][r]);
            }
      }
    }
  }
  }
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((i >> 1) != wk) continue;
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int gn = n0 + wn * 64 + i * 16 + lr;
      if (g == 0 && gn < pN) atomicAdd(colsum + gn, v);
    }
  }
}
#undef TN_FRAG_READ

// ---- grouped form: the weight gradients of up to TN_GROUP_MAX linear layers in ONE launch.  A weight gradient is off the
// critical path of backward (only the data gradient feeds the next layer), and alone it is latency bound: 16 - 64 blocks of dW, each
// a serial chain over all M rows.  Queued and launched together at the end of backward, the layers' blocks fill the chip
// (~1900 blocks of 128 x 128 for the 4-layer model), every block contracts ALL rows of its layer (no m-split: no partial-sum
// workspace, no fold pass, plain += into the fp32 gradient), longest layers first.
constexpr int TN_GROUP_MAX = 48;
struct TnGroupProb {                  // Tn128Args in 72 bytes: 48 problems + their prefix sums stay inside the 4 KB of kernel arguments
  const void* A; const void* B; float* C; float* colsum;
  int lda, ldb, ldc, M, N, K, m_per_split, tiles_k, ntiles, pad;
};
struct TnGroupArgs {
  TnGroupProb p[TN_GROUP_MAX];
  int first[TN_GROUP_MAX + 1];       // first[i] = number of blocks of the problems before i
  int n;
};
static_assert(sizeof(TnGroupArgs) + 16 <= 4096, "kernel argument segment");
__device__ __forceinline__ Tn128Args tn_group_prob(const TnGroupArgs& ga, int i) {
  const TnGroupProb& q = ga.p[i];
  Tn128Args p;
  p.A = q.A; p.B = q.B; p.C = q.C; p.colsum = q.colsum; p.ws = nullptr;
  p.lda = q.lda; p.ldb = q.ldb; p.ldc = q.ldc;
  p.M = q.M; p.N = q.N; p.K = q.K; p.m_per_split = q.m_per_split; p.tiles_k = q.tiles_k; p.ntiles = q.ntiles;
  return p;
}
template <int NST, bool ROT, bool SWAP = false>        // ROT: the launch is a list of whole-contraction blocks, longest first (walk order below)
__global__ __launch_bounds__(512, 2) void gemm_tn256g_kernel(TnGroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  // ROT (whole-contraction blocks, longest first): workgroups are dispatched in blockIdx order as CUs free up, so the list must be
  // walked in that order -- in rounds of 256 (one per CU), and inside a round the 32 blocks of an XCD are consecutive list entries
  // (blocks of one problem share operand columns in that XCD's L2)
  int wid;
  if constexpr (ROT) {
    const int base = bid & ~255, left = min(256, nwg - base), q = left >> 3, r = left & 7, b = bid - base;
    wid = base + ((b & 7) < r ? (b & 7) * (q + 1) : r * (q + 1) + ((b & 7) - r) * q) + (b >> 3);
  } else {
    wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  }
  int i = 0;
  while (i + 1 < ga.n && wid >= ga.first[i + 1]) ++i;
  const Tn128Args p = tn_group_prob(ga, i);
  const int w = wid - ga.first[i], split = w / p.ntiles, m_beg = split * p.m_per_split;
  tn256r_body<NST, SWAP>(p, smem, w % p.ntiles, m_beg, min(p.M, m_beg + p.m_per_split), p.m_per_split >= p.M);
}

// ---- the same blocks, scheduled by the host: the launch is ONE workgroup per CU and every workgroup gets the same number of 32-row
// stages.  All (problem, block of dW, stage) triples of the launch form one line -- problems in launch order, blocks within a
// problem, stages within a block -- which is cut into gridDim.x equal pieces; a workgroup walks its piece: the tail of one block's
// rows, whole blocks, the head of the next (a block of dW whose rows are shared between workgroups is summed with fp32 atomics).
// Measured before (profiles/r03_bench_timeline.txt, launch-by-launch listing): 184 / 300 / 304 equal-length blocks on 256 CUs
// took 185 / 340 / 266 us -- the second round of 44 blocks costs as much as the first of 256.
// first[] holds the prefix sums of STAGES per problem here; m_per_split the stages of one block of that problem.
template <int NST>
__global__ __launch_bounds__(512, 2) void gemm_tn256s_kernel(TnGroupArgs ga, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  const int total = ga.first[ga.n];
  int x = wid * per_wg;
  const int x1 = min(total, x + per_wg);
  int i = 0;
  bool again = false;
  while (x < x1) {
    while (i + 1 < ga.n && x >= ga.first[i + 1]) ++i;
    const Tn128Args p = tn_group_prob(ga, i);
    const int spb = p.m_per_split;                         // stages of one block of this problem
    const int w = x - ga.first[i], tile = w / spb, st0 = w % spb;
    const int st1 = min(spb, st0 + (x1 - x));
    if (again) __syncthreads();                            // every wave is done reading the previous piece's last stages
    tn256r_body<NST, false>(p, smem, tile, st0 * 32, min(p.M, st1 * 32), st0 == 0 && st1 == spb);
    x += st1 - st0;
    again = true;
  }
}
template <int NST>
__global__ __launch_bounds__(256, 2) void gemm_tn128g_kernel(TnGroupArgs ga) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  int i = 0;
  while (i + 1 < ga.n && wid >= ga.first[i + 1]) ++i;
  tn128p_body<NST>(tn_group_prob(ga, i), smem, wid - ga.first[i], true);
}

__global__ __launch_bounds__(256) void tn128_reduce_kernel(const float* __restrict__ ws, float* C, int64_t ldc, int N, int K,
                                                           int ntiles, int tiles_k, int splits) {
  const int tile = blockIdx.x >> 4;
  const int e = ((blockIdx.x & 15) * 256 + threadIdx.x) * 4;       // 4 consecutive columns of one row of the 128 x 128 block
  const int row = e >> 7, col = e & 127;
  const int gn = (tile / tiles_k) * 128 + row, gk = (tile % tiles_k) * 128 + col;
  if (gn >= N || gk >= K) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sp = 0; sp < splits; ++sp) {
    const float4 t = *reinterpret_cast<const float4*>(ws + ((int64_t)sp * ntiles + tile) * 16384 + e);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  float* dst = C + (int64_t)gn * ldc + gk;
  const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (gk + i < K) dst[i] += vv[i];
}

// ================================================================================================ NN (data gradient)
// C[M,N] (op)= alpha * sum_k A[m,k] * B[k,n]  with B = W (K_red, N_out) in its NATURAL master layout: dX = dY . W needs the
// contraction index as W's slow axis, so the B operand is built with the transposing LDS read (bf16) / 4-byte reads (fp32)
// exactly like gemm_tn -- no transposed weight shadows.  A staging, pipeline and epilogue are those of gemm_glds (BN = 64).
// NST = 1: one LDS stage, "load, wait, compute" -- the form for launches of several workgroups per CU, which hide each other's load
// latency.  NST = 3 (bf16): a private ring of hand-issued LDS-DMA stages with counted waits and ONE barrier per K step, for launches
// that leave a CU with a single workgroup (the decoder's data gradients: 1600 rows x 512 columns = 200 blocks over K = 1536 - 4416):
// there nothing else covers the ~0.9 us a K step spends waiting for its operands.
template <typename T, typename TO, int BM, int NST = 1>
__device__ __forceinline__ void gemm_nn_body(const GemmArgs& p, const int bid, const int nwg, unsigned char* smem) {
  using P = TnPack<T>;
  constexpr int ESZ = (int)sizeof(T);
  constexpr int BN = 64, BKB = 128;
  constexpr int BKR = BKB / ESZ;                 // reduction rows per stage: 64 (bf16) / 32 (fp32)
  constexpr int WM = BM / 2, WN = BN / 2, FM = WM / 16, FN = WN / 16;
  constexpr int STAGE = BM * BKB + BKR * P::ROWB;
  constexpr int CPITCH = BN * 4 + 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, g = lane >> 4;
  const int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  const int nk = (p.K + BKR - 1) / BKR;          // a partial last stage: A's columns past K are zero (caller), B's rows are clamped
  const int b_chunks = (int)(p.ldb * ESZ / 16);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int kt, int buf) __attribute__((always_inline)) {
    unsigned char* s = smem + buf * STAGE;
    stage_glds<BM>(s, A, p.lda * ESZ, m0, p.M, (int64_t)kt * BKB, tid, wave);
    unsigned char* sb = s + BM * BKB;
#pragma unroll
    for (int i = 0; i < BKR * P::CPR / 256; ++i) {
      const int c = i * 256 + tid, row = c / P::CPR, slot = (c % P::CPR) ^ (row & 7);
      int cb = n0 * ESZ / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
      const int br = min(kt * BKR + row, p.K - 1);
      const unsigned char* src = B + (int64_t)br * p.ldb * ESZ + (int64_t)cb * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(sb + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
  };

  auto compute = [&](const unsigned char* sA) __attribute__((always_inline)) {
    const unsigned char* sB = sA + BM * BKB;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      uint4 a[FM], b[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int r = wm * WM + i * 16 + lr;
        a[i] = *reinterpret_cast<const uint4*>(sA + r * BKB + (((ms * 4 + g) ^ (r & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = P::template load<true>(sB, ms * (BKR / 2), lr, g, wn * WN + j * 16);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma16<T>(acc[i][j], a[i], b[j]);
    }
  };
  if constexpr (NST == 1) {
    // one LDS stage: load, wait, compute (see gemm_glds_kernel: workgroups per CU beat a private prefetch queue here)
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) __syncthreads();                 // everybody is done reading step kt-1
      stage(kt, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      compute(smem);
    }
  } else {
    static_assert(ESZ == 2, "the ring form is bf16 only");
    constexpr int PA = BM * 8 / 256, PB = BKR * P::CPR / 256;      // DMA pieces per thread and stage: A rows, B rows
    const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned wave_lds = smem_base + (unsigned)wave * 1024u;
    // the DMA is hand issued (the compiler does not count it): its waits are the counted ones below, and the operand reads of
    // compute() carry no s_waitcnt vmcnt(0) of the compiler's own
    auto stage_ring = [&](int kt) __attribute__((always_inline)) {
      const unsigned sl = wave_lds + (unsigned)((kt % NST) * STAGE);
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int c = i * 256 + tid, row = c >> 3, slot = (c & 7) ^ (row & 7);
        int gr = m0 + row;
        gr = gr < p.M ? gr : p.M - 1;
        tn_dma(sl + (unsigned)(i * 4096), A + (int64_t)gr * p.lda * ESZ + (int64_t)kt * BKB + slot * 16);
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int c = i * 256 + tid, row = c / P::CPR, slot = (c % P::CPR) ^ (row & 7);
        int cb = n0 * ESZ / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
        const int br = min(kt * BKR + row, p.K - 1);
        tn_dma(sl + (unsigned)(BM * BKB + i * 4096), B + (int64_t)br * p.ldb * ESZ + (int64_t)cb * 16);
      }
    };
#pragma unroll
    for (int kt = 0; kt < NST - 1; ++kt)
      if (kt < nk) stage_ring(kt);
    for (int kt = 0; kt < nk; ++kt) {
      // step kt has landed once at most the pieces of the later steps are outstanding (PA + PB per step and thread, retired in order)
      const int ahead = min(NST - 2, nk - 1 - kt);
      if (ahead >= 1) wait_vmcnt<PA + PB>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();                // step kt visible to every wave; every wave is done reading step kt - 1
      asm volatile("" ::: "memory");
      if (kt + NST - 1 < nk) stage_ring(kt + NST - 1);
      compute(smem + (kt % NST) * STAGE);
    }
  }
  __syncthreads();                               // operand stage free for the epilogue

  // ---- storage-dtype output in whole 16-byte chunks (every data-gradient GEMM of the model).  Plain store: the tile is staged
  // in the output dtype.  Accumulate (dX added into the residual gradient): staged in fp32 so that C + acc is rounded ONCE.
  if constexpr (sizeof(TO) == sizeof(T)) {
    constexpr int EPCO = 16 / (int)sizeof(TO);
    if (p.vecC && p.N % EPCO == 0 && p.ldc % EPCO == 0) {
      TO* C = static_cast<TO*>(p.C);
      const T* Msk = static_cast<const T*>(p.mask);
      constexpr int CPRO = BN / EPCO;
      if (!p.accumulate) {
        constexpr int OP = BN * (int)sizeof(TO) + 16;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<TO*>(smem + (wm * WM + i * 16 + g * 4 + r) * OP + (wn * WN + j * 16 + lr) * sizeof(TO)) =
                  DT<TO>::to(acc[i][j][r] * p.alpha);
        __syncthreads();
        for (int c = tid; c < BM * CPRO; c += 256) {
          const int row = c / CPRO, col = (c % CPRO) * EPCO;
          const int gr = m0 + row, gc = n0 + col;
          if (gr >= p.M || gc >= p.N) continue;
          Chunk<TO> o;
          o.v = *reinterpret_cast<const uint4*>(smem + row * OP + col * sizeof(TO));
          if (Msk) {
            Chunk<T> m;
            m.v = *reinterpret_cast<const uint4*>(Msk + (int64_t)gr * p.ldc + gc);
#pragma unroll
            for (int e = 0; e < EPCO; ++e)
              if (!(DT<T>::from(m.e[e]) > 0.f)) o.e[e] = DT<TO>::to(0.f);
          }
          *reinterpret_cast<uint4*>(C + (int64_t)gr * p.ldc + gc) = o.v;
          if constexpr (sizeof(TO) == 2) {
            if (p.dot_out) {
              // the attention backward's delta = rowsum(dO * O) per head from the block that IS dO (asr_gemm_nn_rowdot; the lanes, the
              // order of additions and therefore the bits of csrc/attention_fast.hip attn_delta_bf16_d64_kernel): 8 lanes = one head
              const int64_t ooff = (int64_t)gr * p.N + gc;
              float acc = 0.f;
              if (p.dot_o32) {
                const f32x4_t o0 = *reinterpret_cast<const f32x4_t*>(p.dot_o32 + ooff), o1 = *reinterpret_cast<const f32x4_t*>(p.dot_o32 + ooff + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc += o0[e] * bf16_to_f32(o.e[e]) + o1[e] * bf16_to_f32(o.e[4 + e]);
              } else {
                Chunk<bf16_t> f;
                f.v = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.dot_o) + ooff);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += bf16_to_f32(f.e[e]) * bf16_to_f32(o.e[e]);
              }
              acc += __shfl_xor(acc, 4, 64);
              acc += __shfl_xor(acc, 2, 64);
              acc += __shfl_xor(acc, 1, 64);
              if ((c % CPRO) == 0) p.dot_out[((int64_t)(gr / p.dot_T) * p.dot_H + (gc >> 6)) * p.dot_T + gr % p.dot_T] = acc;
            }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              *reinterpret_cast<float*>(smem + (wm * WM + i * 16 + g * 4 + r) * CPITCH + (wn * WN + j * 16 + lr) * 4) = acc[i][j][r] * p.alpha;
        __syncthreads();
        for (int c = tid; c < BM * CPRO; c += 256) {
          const int row = c / CPRO, col = (c % CPRO) * EPCO;
          const int gr = m0 + row, gc = n0 + col;
          if (gr >= p.M || gc >= p.N) continue;
          TO* dst = C + (int64_t)gr * p.ldc + gc;
          Chunk<TO> o, old;
          old.v = *reinterpret_cast<const uint4*>(dst);
          Chunk<T> m;
          if (Msk) m.v = *reinterpret_cast<const uint4*>(Msk + (int64_t)gr * p.ldc + gc);
#pragma unroll
          for (int e = 0; e < EPCO; ++e) {
            float v = *reinterpret_cast<const float*>(smem + row * CPITCH + (col + e) * 4);
            if (Msk && !(DT<T>::from(m.e[e]) > 0.f)) v = 0.f;
            o.e[e] = DT<TO>::to(v + DT<TO>::from(old.e[e]));
          }
          *reinterpret_cast<uint4*>(dst) = o.v;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<float*>(smem + (wm * WM + i * 16 + g * 4 + r) * CPITCH + (wn * WN + j * 16 + lr) * 4) = acc[i][j][r] * p.alpha;
  __syncthreads();
  TO* C = static_cast<TO*>(p.C);
  const T* Msk = static_cast<const T*>(p.mask);
  constexpr int CPR = BN / 4;
  for (int c = tid; c < BM * CPR; c += 256) {
    const int row = c / CPR, col = (c % CPR) * 4;
    const int gr = m0 + row, gc = n0 + col;
    if (gr >= p.M || gc >= p.N) continue;
    const float4 v4 = *reinterpret_cast<const float4*>(smem + row * CPITCH + col * 4);
    float v[4] = {v4.x, v4.y, v4.z, v4.w};
    TO* dst = C + (int64_t)gr * p.ldc + gc;
    const int nvalid = min(4, p.N - gc);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < nvalid && Msk && !(DT<T>::ld(Msk + (int64_t)gr * p.ldc + gc + e) > 0.f)) v[e] = 0.f;
    if (p.vecC && nvalid == 4) {
      if constexpr (sizeof(TO) == 4) {
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (p.accumulate) { const float4 old = *reinterpret_cast<const float4*>(dst); o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
        *reinterpret_cast<float4*>(dst) = o;
      } else {
        if (p.accumulate) {
          const uint2 old = *reinterpret_cast<const uint2*>(dst);
          v[0] += bf16_to_f32((bf16_t)(old.x & 0xffff)); v[1] += bf16_to_f32((bf16_t)(old.x >> 16));
          v[2] += bf16_to_f32((bf16_t)(old.y & 0xffff)); v[3] += bf16_to_f32((bf16_t)(old.y >> 16));
        }
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(dst) = o;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (e < nvalid) store_out<TO>(dst + e, v[e], p.accumulate, 0);
    }
  }
}

template <typename T, typename TO, int BM, int NST = 1>
__global__ __launch_bounds__(256) void gemm_nn_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemm_nn_body<T, TO, BM, NST>(p, (int)blockIdx.x, (int)gridDim.x, smem);
}

// ------------------------------------------------------------------------------------------------ TN on quadrant waves (bf16)
// The weight-gradient contraction again, with the FOOTPRINT of the data-gradient kernel (64 x 64 tile of dW, waves in a 2 x 2
// grid of 32 x 32 quadrants, every wave contracts every row of a 64-row stage: 16 accumulator registers, one 16 KB LDS stage,
// no cross-wave reduction) so that both can be workgroups of ONE launch (gemm_nn_tnq_kernel below): a layer's dX and dW used to
// be two launches on two streams, and every fork / join of a replayed graph is a 5-10 us hole on the main stream (88 of them
// per step).  Partial tiles go to the workspace [split][tile][64][64]; asr_tn_reduce_multi folds all layers at once.
__device__ __forceinline__ void gemm_tnq_body(const TnArgs& p, const int bid, const int nwg, unsigned char* smem) {
  using P = TnPack<bf16_t>;
  constexpr int RM = 64, TILEB = RM * P::ROWB;                    // 8 KB per operand
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int wid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  const int split = wid / p.ntiles, tile = wid % p.ntiles;
  const int n0 = (tile / p.tiles_k) * 64, k0 = (tile % p.tiles_k) * 64;
  const int m_beg = split * p.m_per_split, m_end = min(p.M, m_beg + p.m_per_split);
  const int nstage = (m_end - m_beg + RM - 1) / RM;
  const unsigned char* A = static_cast<const unsigned char*>(p.A);
  const unsigned char* B = static_cast<const unsigned char*>(p.B);
  const int a_chunks = (int)(p.lda * 2 / 16), b_chunks = (int)((p.ldb >= p.K ? p.ldb : (int64_t)((p.K + 7) / 8 * 8)) * 2 / 16);
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(&tn_zero_page);

  unsigned offA[2], offB[2];
  int rowi[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = i * 256 + tid, row = c >> 3, slot = (c & 7) ^ (row & 7);
    int ca = n0 * 2 / 16 + slot; ca = ca < a_chunks ? ca : a_chunks - 1;       // columns past N / K are never stored
    int cb = k0 * 2 / 16 + slot; cb = cb < b_chunks ? cb : b_chunks - 1;
    offA[i] = (unsigned)(row * (int)p.lda * 2 + ca * 16);
    offB[i] = (unsigned)(row * (int)p.ldb * 2 + cb * 16);
    rowi[i] = row;
  }

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float bsum[2] = {0.f, 0.f};
  const bool do_colsum = p.colsum != nullptr && k0 == 0 && wk == 0;

  for (int st = 0; st < nstage; ++st) {
    if (st > 0) __syncthreads();                 // everybody is done reading stage st-1
    const int64_t mrow = m_beg + (int64_t)st * RM;
    const unsigned char* ba = A + mrow * p.lda * 2;
    const unsigned char* bb = B + mrow * p.ldb * 2;
    const int valid = m_end - (int)mrow;         // rows of this stage that exist (uniform); the others come from a page of zeros
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool in = rowi[i] < valid;
      unsigned char* d = smem + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in ? ba + offA[i] : zero),
                                       (__attribute__((address_space(3))) void*)d, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in ? bb + offB[i] : zero),
                                       (__attribute__((address_space(3))) void*)(d + TILEB), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned char* sA = smem;
    const unsigned char* sB = smem + TILEB;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = P::load(sA, ms * 32, lr, g, wn * 32 + i * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = P::load(sB, ms * 32, lr, g, wk * 32 + j * 16);
      if (do_colsum) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          Chunk<bf16_t> c; c.v = a[i];
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[i] += bf16_to_f32(c.e[e]);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma16<bf16_t>(acc[i][j], a[i], b[j]);
    }
  }

  // ---- the wave's 32 x 32 quadrant: lane (lr, g) holds rows 4g..4g+3 of column lr of every fragment
  const bool single = nwg == p.ntiles;
  float* part = p.ws ? p.ws + ((int64_t)split * p.ntiles + tile) * 4096 : nullptr;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wn * 32 + i * 16 + g * 4 + r, gn = n0 + row;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wk * 32 + j * 16 + lr, gk = k0 + col;
        const float v = acc[i][j][r];
        if (part) part[row * 64 + col] = v;
        else if (gn < p.N && gk < p.K) {
          float* dst = p.C + (int64_t)gn * p.ldc + gk;
          if (single) *dst += v; else atomicAdd(dst, v);
        }
      }
    }
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float v = bsum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int gn = n0 + wn * 32 + i * 16 + lr;
      if (g == 0 && gn < p.N) atomicAdd(p.colsum + gn, v);
    }
  }
}

// C tile += the m-slices of its partial tiles [split][tile][64][64]: one workgroup folds a quarter tile (fixed order).
__device__ __forceinline__ void tn_fold_block(const float* __restrict__ ws, float* C, const int64_t ldc, const int N, const int K,
                                              const int splits, const int bid) {
  const int tiles_k = (K + 63) / 64, ntiles = ((N + 63) / 64) * tiles_k;
  const int tile = bid >> 2;
  if (tile >= ntiles) return;
  const int e = ((bid & 3) * 256 + threadIdx.x) * 4;               // 4 consecutive columns of one row of the tile
  const int row = e >> 6, col = e & 63;
  const int gn = (tile / tiles_k) * 64 + row, gk = (tile % tiles_k) * 64 + col;
  if (gn >= N || gk >= K) return;
  const float* src = ws + (int64_t)tile * 4096 + e;
  const int64_t step = (int64_t)ntiles * 4096;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int sp = 0; sp < splits; ++sp) {
    const float4 t = *reinterpret_cast<const float4*>(src + sp * step);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  float* dst = C + (int64_t)gn * ldc + gk;
  if (gk + 3 < K && ((((uintptr_t)dst) & 15) == 0)) {
    float4 o = *reinterpret_cast<float4*>(dst);
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *reinterpret_cast<float4*>(dst) = o;
  } else {
    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (gk + i < K) dst[i] += vv[i];
  }
}

// dX workgroups and dW workgroups of one linear layer in ONE launch: the first n_tn workgroups are the (longer) weight-gradient
// tiles, then the n_nn data-gradient tiles, then -- software pipelining across launches -- the workgroups that fold the partial
// tiles the PREVIOUS layer's launch left behind (HBM-bound, they ride under the MFMA-bound tiles of this one).
struct TnFoldArgs {
  const float* ws; float* C; int64_t ldc; int N, K, splits;
};
template <int BM>
__global__ __launch_bounds__(256) void gemm_nn_tnq_kernel(GemmArgs pn, TnArgs pt, TnFoldArgs pf, int n_tn, int n_nn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int bid = (int)blockIdx.x;
  if (bid < n_tn) gemm_tnq_body(pt, bid, n_tn, smem);
  else if (bid < n_tn + n_nn) gemm_nn_body<bf16_t, bf16_t, BM>(pn, bid - n_tn, n_nn, smem);
  else tn_fold_block(pf.ws, pf.C, pf.ldc, pf.N, pf.K, pf.splits, bid - n_tn - n_nn);
}

// Second stage for up to TN_MULTI layers in one launch (blockIdx.y = layer).
constexpr int TN_MULTI = 48;
struct TnMultiArgs {
  const float* ws[TN_MULTI];
  float* C[TN_MULTI];
  int ldc[TN_MULTI], N[TN_MULTI], K[TN_MULTI], splits[TN_MULTI];
};
__global__ __launch_bounds__(256) void tn_reduce_multi_kernel(TnMultiArgs q) {
  const int l = blockIdx.y;
  tn_fold_block(q.ws[l], q.C[l], q.ldc[l], q.N[l], q.K[l], q.splits[l], (int)blockIdx.x);
}

template <typename T, typename TO, int BM, int NST = 1>
int launch_nn(const GemmArgs& a, hipStream_t s) {
  GemmArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + 63) / 64;
  p.ntiles = tiles_m * p.tiles_n;
  const int esz = (int)sizeof(T);
  size_t lds = (size_t)NST * (size_t)(BM * 128 + (128 / esz) * (64 * esz));      // NST operand stages
  const size_t cl = (size_t)BM * (64 * 4 + 16);
  if (cl > lds) lds = cl;
  static bool granted = false;
  if (lds > 48 * 1024 && !granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nn_kernel<T, TO, BM, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    granted = true;
  }
  hipLaunchKernelGGL((gemm_nn_kernel<T, TO, BM, NST>), dim3((unsigned)p.ntiles), dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

template <typename T, typename TO, int BM, int BN>
int launch(const GemmArgs& a, int splits, hipStream_t s) {
  GemmArgs p = a;
  const int tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  dim3 grid((unsigned)(tiles_m * p.tiles_n), 1, (unsigned)splits);
  const size_t lds = (size_t)(BM + BN) * kPitch;
  hipLaunchKernelGGL((gemm_nt_kernel<T, TO, BM, BN>), grid, dim3(256), lds, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

template <typename T, typename TO>
int dispatch_tile(const GemmArgs& a, int splits, hipStream_t s) {
  const int64_t t128 = ceil_div64(a.M, 128) * ceil_div64(a.N, 128) * splits;
  const int64_t t12864 = ceil_div64(a.M, 128) * ceil_div64(a.N, 64) * splits;
  if (t128 >= 384 || (a.M > 64 && a.N > 64 && t12864 < 8)) return launch<T, TO, 128, 128>(a, splits, s);
  if (t12864 >= 384 && a.M > 64) return launch<T, TO, 128, 64>(a, splits, s);
  return launch<T, TO, 64, 64>(a, splits, s);
}

}  // namespace

extern "C" int asr_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                           const float* bias, const void* relu_mask, int M, int N, int K,
                           float alpha, int flags, int splits, int in_dtype, int out_dtype, hipStream_t stream) {
  ASR_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0);
  if (M == 0 || N == 0) return ASR_OK;
  ASR_CHECK_ARG(in_dtype == ASR_F32 || in_dtype == ASR_BF16);
  ASR_CHECK_ARG(out_dtype == ASR_F32 || out_dtype == ASR_BF16);
  ASR_CHECK_ARG(!(in_dtype == ASR_F32 && out_dtype == ASR_BF16));
  const int esz = in_dtype == ASR_F32 ? 4 : 2, epc = 16 / esz, bk = 128 / esz;
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.mask = relu_mask;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.alpha = alpha;
  p.relu = (flags & ASR_GEMM_RELU) != 0;
  p.accumulate = (flags & ASR_GEMM_ACCUMULATE) != 0;
  if (splits == 0) {
    // auto: fp32 atomics are expensive (~10 ns each), so split only while the 64x64 grid cannot fill the chip
    splits = 1;
    if (p.accumulate && out_dtype == ASR_F32 && !p.relu && !relu_mask) {
      const int64_t t64 = ceil_div64(M, 64) * ceil_div64(N, 64);
      int64_t sp = (256 + t64 / 2) / (t64 > 0 ? t64 : 1);
      if (sp > 4) sp = 4;
      if (sp > K / (4 * bk)) sp = K / (4 * bk);
      if (sp >= 2) splits = (int)sp;
    }
  }
  if (splits < 1) splits = 1;
  int kps = (int)(ceil_div64(ceil_div64(K > 0 ? K : 1, splits), bk) * bk);
  splits = (int)ceil_div64(K > 0 ? K : 1, kps);
  p.k_per_split = kps;
  p.atomic = 0;
  if (splits > 1) {
    // split-K partial sums are combined with fp32 atomics: the destination must already hold the value to add to
    ASR_CHECK_ARG(out_dtype == ASR_F32 && p.accumulate && !p.relu && !relu_mask);
    p.atomic = 1;
  }
  p.vecA = aligned16(A) && (lda % epc == 0) && (K % epc == 0);
  p.vecB = aligned16(B) && (ldb % epc == 0) && (K % epc == 0);
  const int oesz = out_dtype == ASR_F32 ? 4 : 2;
  p.vecC = ((((uintptr_t)C) & 15) == 0) && (ldc % 4 == 0) && (oesz == 4 || ldc % 4 == 0);
  AsrProfScope prof(ASR_OP_GEMM, stream);
#ifdef ASR_TUNE_ABLATE
  p.ablate = (int)asr_tuning("GEMM_ABLATE", 0);
#endif
  // eight-wave 256 x 256 / 128 x 128 blocks (csrc/gemm_big.hip) for the bf16 linear layers whose shape fills the chip with them
  if (in_dtype == ASR_BF16 && splits == 1 && !relu_mask) {
    BigGemmArgs q{};
    q.A = A; q.B = B; q.C = C; q.bias = bias; q.mask = nullptr;
    q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.M = M; q.N = N; q.K = K; q.alpha = alpha;
    q.relu = p.relu; q.accumulate = p.accumulate; q.out_f32 = out_dtype == ASR_F32;
    if (asr_gemm_big_nt(q, stream)) return ASR_OK;
  }
  // fast path: LDS-DMA staging needs whole 16-B chunks everywhere and whole 128-byte K steps
  const bool fast = p.vecA && p.vecB && K > 0 && (K % bk == 0) && (kps % bk == 0) &&
                    asr_tuning("GEMM_GENERIC", 0) == 0;
  if (fast) {
    if (in_dtype == ASR_F32) return dispatch_fast<float, float>(p, splits, stream);
    if (out_dtype == ASR_BF16) return dispatch_fast<bf16_t, bf16_t>(p, splits, stream);
    return dispatch_fast<bf16_t, float>(p, splits, stream);
  }
  if (in_dtype == ASR_F32) return dispatch_tile<float, float>(p, splits, stream);
  if (out_dtype == ASR_BF16) return dispatch_tile<bf16_t, bf16_t>(p, splits, stream);
  return dispatch_tile<bf16_t, float>(p, splits, stream);
}

namespace {
// slices over m: explicit, or automatic.  With a workspace the split costs one extra pass over splits*N*K floats, so the grid is
// filled to ~2 workgroups per CU; without one the slices meet in fp32 atomics and are kept to the measured optimum (<= 4).
int tn_splits(int M, int N, int K, int splits, int dtype, bool have_ws) {
  const int ASR_TN_TARGET_WGS = (int)asr_tuning("TN_WGS", 512);
  const int rm = dtype == ASR_F32 ? 64 : 128;
  const int ntiles = ((N + 63) / 64) * ((K + 63) / 64);
  const int stages = (M + rm - 1) / rm;
  if (splits <= 0) {
    if (have_ws) {
      splits = (ASR_TN_TARGET_WGS + ntiles / 2) / ntiles;
      if (splits > 8) splits = 8;
      while (splits > 1 && stages / splits < 4) --splits;      // at least 4 stages per slice
    } else {
      splits = (160 + ntiles - 1) / ntiles;
      if (splits > 4) splits = 4;
    }
  }
  if (splits > stages) splits = stages;
  if (splits < 1) splits = 1;
  const int sps = (stages + splits - 1) / splits;
  return (stages + sps - 1) / sps;
}
}  // namespace

namespace {
// 128 x 128-tile kernel: bf16, automatic split, a workspace, and enough 128-blocks that ~512 workgroups of >= 4 stages exist.
// Returns the number of m-slices (0 = use the 64 x 64-tile kernel).
// LDS stages of the pipelined 128 x 128 kernel for an output of `nt` blocks, 0 = use the single-stage kernels.  Measured
// (tools/microbench.py tn, profiles/r02_microbench_tn.txt): from 64 blocks on (2048 x 512 and larger) the pipelined kernel wins
// (512 x 2048 over 6400 rows: 38.9 -> 33.8 us; 2048 x 512 over 12720 rows: 58.8 -> 47.6 us), below that its m-slices are too
// short to fill the ring and the 64 x 64 kernel's 8 workgroups per CU win (512 x 512: 19 vs 30 us).  3 stages (48 KB, 3
// workgroups per CU) tie or beat 4.
int tn_pipe_stages(int nt) {
  const int pipe = (int)asr_tuning("TN_PIPE", 3);
  return (pipe > 0 && nt >= (int)asr_tuning("TN_PIPE_MIN", 64)) ? pipe : 0;
}

int tn128_splits(int M, int N, int K, int splits, int dtype) {
  const int enabled = (int)asr_tuning("TN_128", 1);
  const int min_tiles = (int)asr_tuning("TN_128_MIN", 128);      // measured: wins for 512x5120 (160 blocks), loses for 64-block outputs
  if (!enabled || dtype != ASR_BF16 || splits > 0 || N < 128 || K < 128) return 0;
  const int nt = ((N + 127) / 128) * ((K + 127) / 128);
  const int pipe = tn_pipe_stages(nt);
  if (!pipe && nt < min_tiles) return 0;
  const int stages = pipe ? (M + 31) / 32 : (M + 63) / 64;
  int sp = (512 + nt / 2) / nt;
  if (sp > (pipe ? 32 : 16)) sp = pipe ? 32 : 16;
  while (sp > 1 && stages / sp < 4) --sp;
  if (sp < 1) sp = 1;
  const int sps = (stages + sp - 1) / sp;
  return (stages + sps - 1) / sps;
}
}  // namespace

extern "C" int64_t asr_gemm_tn_workspace(int M, int N, int K, int splits, int dtype) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s128 = tn128_splits(M, N, K, splits, dtype);
  if (s128 > 1) return (int64_t)s128 * ((N + 127) / 128) * ((K + 127) / 128) * 16384;
  const int sp = tn_splits(M, N, K, splits, dtype, true);
  return sp > 1 ? (int64_t)sp * ((N + 63) / 64) * ((K + 63) / 64) * 4096 : 0;
}

extern "C" int asr_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* C, int64_t ldc, float* colsum_acc,
                           float* workspace, int64_t workspace_floats, int M, int N, int K, int splits, int dtype,
                           hipStream_t stream) {
  ASR_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (N == 0 || K == 0 || M == 0) return ASR_OK;
  const int esz = dtype == ASR_F32 ? 4 : 2, epc = 16 / esz;
  const int rm = dtype == ASR_F32 ? 64 : 128;
  // 16-byte aligned rows; a partial last stage of m is zero-filled in LDS by the kernel
  if (lda % epc != 0 || ldb % epc != 0 || !aligned16(A) || !aligned16(B) || lda < N ||
      lda >= ((int64_t)1 << 22) || ldb >= ((int64_t)1 << 22))      // (32-bit byte offsets inside one stage of rows)
    return ASR_EUNSUPPORTED;
  {
    const int s128 = tn128_splits(M, N, K, splits, dtype);
    const int64_t need = (int64_t)s128 * ((N + 127) / 128) * ((K + 127) / 128) * 16384;
    if (s128 >= 1 && (s128 == 1 || (workspace && workspace_floats >= need))) {
      Tn128Args q{};
      q.A = A; q.B = B; q.C = C; q.colsum = colsum_acc; q.ws = s128 > 1 ? workspace : nullptr;
      q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.M = M; q.N = N; q.K = K;
      q.tiles_k = (K + 127) / 128;
      q.ntiles = ((N + 127) / 128) * q.tiles_k;
      const int pipe = tn_pipe_stages(q.ntiles);
      const int stages = pipe ? (M + 31) / 32 : (M + 63) / 64;
      q.m_per_split = ((stages + s128 - 1) / s128) * (pipe ? 32 : 64);
      AsrProfScope prof(ASR_OP_GEMM, stream);
      const int rm128 = (int)asr_tuning("TN_128_RM", 64);
      if (pipe) {
        static bool granted = false;
        if (!granted) {
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128p_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
          (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128p_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
          granted = true;
        }
        if (pipe == 3) hipLaunchKernelGGL(gemm_tn128p_kernel<3>, dim3((unsigned)(q.ntiles * s128)), dim3(256), 3 * 16384, stream, q);
        else hipLaunchKernelGGL(gemm_tn128p_kernel<4>, dim3((unsigned)(q.ntiles * s128)), dim3(256), 4 * 16384, stream, q);
      } else if (rm128 == 128) {
        static bool granted = false;
        if (!granted) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 256); granted = true; }
        hipLaunchKernelGGL(gemm_tn128_kernel<128>, dim3((unsigned)(q.ntiles * s128)), dim3(256), 2 * 128 * 256, stream, q);
      } else
      hipLaunchKernelGGL(gemm_tn128_kernel<64>, dim3((unsigned)(q.ntiles * s128)), dim3(256), 2 * 64 * 256, stream, q);
      ASR_LAUNCH_CHECK();
      if (q.ws) {
        hipLaunchKernelGGL(tn128_reduce_kernel, dim3((unsigned)(q.ntiles * 16)), dim3(256), 0, stream, q.ws, C, ldc, N, K, q.ntiles,
                           q.tiles_k, s128);
        ASR_LAUNCH_CHECK();
      }
      return ASR_OK;
    }
  }
  TnArgs p{};
  p.A = A; p.B = B; p.C = C; p.colsum = colsum_acc;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  const int tiles_n = (N + 63) / 64;
  p.tiles_k = (K + 63) / 64;
  p.ntiles = tiles_n * p.tiles_k;
  const int stages = (M + rm - 1) / rm;
  const int64_t tile_floats = (int64_t)p.ntiles * 4096;
  bool have_ws = workspace != nullptr && workspace_floats >= 2 * tile_floats;
  const int want = tn_splits(M, N, K, splits, dtype, have_ws);
  if (have_ws && workspace_floats < (int64_t)want * tile_floats) have_ws = false;
  splits = have_ws ? want : tn_splits(M, N, K, splits, dtype, false);
  const int sps = (stages + splits - 1) / splits;
  p.m_per_split = sps * rm;
  p.ws = (have_ws && splits > 1) ? workspace : nullptr;
  const int nbuf = (int)asr_tuning("TN_NBUF", 1);       // LDS stages (tuning hook)
  const size_t lds_stage = (size_t)(nbuf == 2 ? 2 : 1) * 2 * rm * (64 * esz);
  const size_t lds_epi = (size_t)2 * 64 * 64 * 4 + 4 * 64 * sizeof(float);
  const size_t lds = lds_stage > lds_epi ? lds_stage : lds_epi;
  AsrProfScope prof(ASR_OP_GEMM, stream);
  const dim3 grid((unsigned)(p.ntiles * splits));
#define ASR_TN_LAUNCH(T_, NB_)                                                                                                  \
  {                                                                                                                              \
    static bool granted = false;                                                                                                 \
    if (!granted) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_kernel<T_, NB_>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); granted = true; } \
    hipLaunchKernelGGL((gemm_tn_kernel<T_, NB_>), grid, dim3(256), lds, stream, p);                                              \
  }
  if (dtype == ASR_F32) { if (nbuf == 2) ASR_TN_LAUNCH(float, 2) else ASR_TN_LAUNCH(float, 1) }
  else { if (nbuf == 2) ASR_TN_LAUNCH(bf16_t, 2) else ASR_TN_LAUNCH(bf16_t, 1) }
#undef ASR_TN_LAUNCH
  ASR_LAUNCH_CHECK();
  if (p.ws) {
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)(p.ntiles * 4)), dim3(256), 0, stream, p.ws, C, ldc, N, K, p.ntiles, p.tiles_k,
                       splits);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}

// Slices per problem for the whole-block form of asr_gemm_tn_grouped: the launch's workgroups (one per block of dW and slice of its
// rows, longest first) are played through on 256 CUs -- each goes to the CU that is free first, as the dispatcher does -- for a few
// upper bounds on the slice length; a workgroup costs its stages + a fixed part (pipeline fill, epilogue; more with atomics).
// The plan of the last few distinct problem lists is kept (a training step asks for the same lists again and again).
static bool tn_rot_plan(int cnt, const int* order, const int* M, const int* N, const int* K, int* splits_out) {
  struct Memo { uint64_t key; int cnt; bool whole_wins; int splits[TN_GROUP_MAX]; };
  static Memo memo[8];
  static int memo_next = 0;
  static std::mutex memo_lock;                  // (the library is called from one thread per process; a second caller must not tear an entry)
  std::lock_guard<std::mutex> hold(memo_lock);
  uint64_t key = 1469598103934665603ull;
  for (int j = 0; j < cnt; ++j) {
    const int i = order[j];
    for (const int v : {M[i], N[i], K[i]}) key = (key ^ (uint64_t)(uint32_t)v) * 1099511628211ull;
  }
  for (const Memo& m : memo)
    if (m.key == key && m.cnt == cnt) {
      for (int j = 0; j < cnt; ++j) splits_out[order[j]] = m.splits[j];
      return m.whole_wins;
    }
  // a visit that ends in 65 536 fp32 atomics costs about as much as 100 stages (profiles/r05_tn_grouped.txt: 123 us of a 480 us
  // pass for ~2.3 visits per workgroup)
  constexpr int CUS = 256, FIXED = 10, FIXED_ATOMIC = 100;
  int64_t whole = 0;
  int longest = 0;
  for (int j = 0; j < cnt; ++j) {
    const int i = order[j];
    const int st = (M[i] + 31) / 32;
    whole += (int64_t)((N[i] + 255) / 256) * ((K[i] + 255) / 256) * st;
    longest = st > longest ? st : longest;
  }
  const int64_t share = whole / CUS > 32 ? whole / CUS : 32;
  const double cuts[] = {1e30, 1.0, 1.0 / 1.5, 0.5, 1.0 / 3, 0.25};
  int best_splits[TN_GROUP_MAX];
  int64_t best = -1;
  std::vector<std::pair<int, int>> items;          // (cost, count) runs, longest first
  for (const double cut : cuts) {
    const double lim = cut > 1e20 ? 1e30 : (double)share * cut;
    if (cut < 1e20 && lim >= longest) continue;      // the same plan as "whole"
    int sp[TN_GROUP_MAX];
    items.clear();
    for (int j = 0; j < cnt; ++j) {
      const int i = order[j];
      const int st = (M[i] + 31) / 32;
      int s = st > lim ? (int)((st + lim - 1) / lim) : 1;
      if (s > st) s = st;
      sp[j] = s;
      const int len = (st + s - 1) / s;
      items.push_back({len + (s > 1 ? FIXED_ATOMIC : FIXED), ((N[i] + 255) / 256) * ((K[i] + 255) / 256) * s});
    }
    std::sort(items.begin(), items.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first > b.first; });
    std::priority_queue<int64_t, std::vector<int64_t>, std::greater<int64_t>> free_at;
    for (int c = 0; c < CUS; ++c) free_at.push(0);
    int64_t end = 0;
    for (const auto& it : items)
      for (int c = 0; c < it.second; ++c) {
        const int64_t t = free_at.top() + it.first;
        free_at.pop();
        free_at.push(t);
        end = t > end ? t : end;
      }
    if (best < 0 || end < best) {
      best = end;
      for (int j = 0; j < cnt; ++j) best_splits[j] = sp[j];
    }
  }
  // the alternative: equal pieces of the launch's stages on one workgroup per CU -- perfectly balanced, but nearly every block is
  // shared between two workgroups (two atomic visits each)
  const int64_t pieces = share + 2 * FIXED_ATOMIC + FIXED;
  Memo& m = memo[memo_next];
  memo_next = (memo_next + 1) % 8;
  m.key = key; m.cnt = cnt; m.whole_wins = best <= pieces;
  for (int j = 0; j < cnt; ++j) { m.splits[j] = best_splits[j]; splits_out[order[j]] = best_splits[j]; }
  return m.whole_wins;
}

// The decision asr_gemm_tn_grouped takes for a list of problems, without launching anything (host only; tests, tuning): 1 = one
// workgroup per whole block of dW (splits[i] > 1: the slices of an over-long block), 0 = round 3's shared forms.
extern "C" int asr_gemm_tn_grouped_plan(int n, const int* M, const int* N, const int* K, int* splits) {
  ASR_CHECK_ARG(n >= 0 && n <= TN_GROUP_MAX && (n == 0 || (M && N && K && splits)));
  int order[TN_GROUP_MAX];
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    ASR_CHECK_ARG(M[i] >= 0 && N[i] >= 0 && K[i] >= 0);
    splits[i] = 1;
    if (M[i] > 0 && N[i] > 0 && K[i] > 0) order[cnt++] = i;
  }
  if (cnt == 0) return 1;
  for (int a = 1; a < cnt; ++a)
    for (int b = a; b > 0 && M[order[b]] > M[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  return tn_rot_plan(cnt, order, M, N, K, splits) ? 1 : 0;
}

extern "C" int asr_gemm_tn_grouped(int n, const void* const* dy, const int64_t* ld_dy, const void* const* x, const int64_t* ld_x,
                                   float* const* dw, const int64_t* ld_dw, float* const* db, const int* M, const int* N, const int* K,
                                   int dtype, hipStream_t stream) {
  ASR_CHECK_ARG(n >= 0 && n <= TN_GROUP_MAX && dtype == ASR_BF16);
  if (n == 0) return ASR_OK;
  ASR_CHECK_ARG(dy && ld_dy && x && ld_x && dw && ld_dw && db && M && N && K);
  int order[TN_GROUP_MAX];
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    ASR_CHECK_ARG(M[i] >= 0 && N[i] >= 0 && K[i] >= 0);
    if (M[i] == 0 || N[i] == 0 || K[i] == 0) continue;   // an empty problem adds nothing (its pointers may be null)
    ASR_CHECK_ARG(dy[i] && x[i] && dw[i]);
    if (ld_dy[i] % 8 != 0 || ld_x[i] % 8 != 0 || !aligned16(dy[i]) || !aligned16(x[i]) || ld_dy[i] < N[i] ||
        ld_dy[i] >= ((int64_t)1 << 22) || ld_x[i] >= ((int64_t)1 << 22) || ld_dw[i] >= ((int64_t)1 << 31))
      return ASR_EUNSUPPORTED;
    order[cnt++] = i;
  }
  if (cnt == 0) return ASR_OK;
  // longest row count first: a block's run time is proportional to M, and the late blocks decide when the launch ends
  for (int a = 1; a < cnt; ++a)
    for (int b = a; b > 0 && M[order[b]] > M[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  TnGroupArgs ga{};
  int total = 0;
  // Forms: 256 x 256 blocks (eight waves), one workgroup per CU, the launch's stages dealt out in equal pieces (gemm_tn256s_kernel);
  // one workgroup per (block of dW, slice of <= TN_GROUP_MROWS rows), fp32 atomics where a block has more than one slice (A/B of the
  // slice length: profiles/r03_grouped_wgrad_ab.txt); the 128 x 128 / four-wave form, one workgroup per whole contraction.  TN_GROUP_TILE:
  // 0 (default): by the longest contraction of the group -- equal pieces up to TN_GROUP_SLICE_MIN rows, per-slice blocks from there on: with
  // >= 3 slices per block of dW the launch is several rounds of workgroups whatever the form, and the slices of one block run side by
  // side on one L2 (47 % hits against 7 %, profiles/r03_tn_group_l2.txt): configs[3] (12 720 rows) 13.16 -> 12.92 ms/step, while the
  // headline's 6 400 rows (2 slices: 1.2 rounds) keep the equal pieces.  1: equal pieces always; 256: per-slice always; 128.
  const int tmode = (int)asr_tuning("TN_GROUP_TILE", 0);
  const int mrows = (int)asr_tuning("TN_GROUP_MROWS", 3200);
  int max_m = 0;
  for (int j = 0; j < cnt; ++j) max_m = M[order[j]] > max_m ? M[order[j]] : max_m;
  // Round 5: tn256r_body, and ONE workgroup per block of dW over the WHOLE contraction, dispatched longest
  // first -- no block is shared between workgroups, so no fp32 atomics and a vector epilogue: round 3's equal pieces shared almost
  // every block (pieces of 130 - 160 stages against blocks of 100 / 200), and the 65 536 atomics per visit were a quarter of the
  // launch (profiles/r05_tn_grouped.txt).
  // (every 256 x 256 form runs tn256r_body since round 6; tmode != 0 forces a form for the tests)
  const bool rot = tmode == 0;
  const bool big = tmode != 128;
  // ... cut into slices of its rows (summed with fp32 atomics, as before) where that shortens the launch: one block longer than a CU's
  // share (emb_cnn's window contractions: one or two blocks over several hundred thousand rows), or equal blocks whose count is an
  // awkward multiple of the CUs (configs[3]: 576 blocks of 398 stages = 2.25 rounds).  tn_rot_plan() decides by playing the dispatch
  // through for a few slice lengths; the list is then ordered by SLICE length.
  int rot_splits[TN_GROUP_MAX];
  // where the whole blocks would leave CUs idle (tn_rot_plan), round 3's shared forms (equal pieces / slices) with the new loop
  const bool whole = rot && tn_rot_plan(cnt, order, M, N, K, rot_splits);
  const bool sched = !whole && (tmode == 1 || (tmode == 0 && max_m < asr_tuning("TN_GROUP_SLICE_MIN", 9600)));
  if (whole) {
    int slice[TN_GROUP_MAX];
    for (int j = 0; j < cnt; ++j) {
      const int i = order[j];
      slice[i] = ((M[i] + 31) / 32 + rot_splits[i] - 1) / rot_splits[i];
    }
    for (int a = 1; a < cnt; ++a)
      for (int b = a; b > 0 && slice[order[b]] > slice[order[b - 1]]; --b) { const int t = order[b]; order[b] = order[b - 1]; order[b - 1] = t; }
  }
  for (int j = 0; j < cnt; ++j) {
    const int i = order[j];
    TnGroupProb& q = ga.p[j];
    q.A = dy[i]; q.B = x[i]; q.C = dw[i]; q.colsum = db[i];
    q.lda = (int)ld_dy[i]; q.ldb = (int)ld_x[i]; q.ldc = (int)ld_dw[i]; q.M = M[i]; q.N = N[i]; q.K = K[i];
    const int T = big ? 256 : 128;
    q.tiles_k = (K[i] + T - 1) / T;
    q.ntiles = ((N[i] + T - 1) / T) * q.tiles_k;
    ga.first[j] = total;
    if (sched) {
      q.m_per_split = (M[i] + 31) / 32;                      // stages per block
      total += q.ntiles * q.m_per_split;
      continue;
    }
    int splits = whole ? rot_splits[i] : 1;
    if (big && mrows > 0 && !whole) splits = (M[i] + mrows - 1) / mrows;
    if (splits < 1) splits = 1;
    q.m_per_split = ((M[i] + splits - 1) / splits + 31) / 32 * 32;
    splits = (M[i] + q.m_per_split - 1) / q.m_per_split;
    total += q.ntiles * splits;
  }
  ga.first[cnt] = total;
  ga.n = cnt;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128g_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 16384);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256g_kernel<3, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256g_kernel<3, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn256s_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32768);
    granted = true;
  }
  AsrProfScope prof(ASR_OP_GEMM, stream);
  if (sched) {
    // one workgroup per CU (TN_GROUP_WGS), at least 16 stages each; pieces of equal length
    int nwg = (int)asr_tuning("TN_GROUP_WGS", 256);
    if (nwg > total / 16) nwg = total / 16;
    if (nwg < 1) nwg = 1;
    const int per_wg = (total + nwg - 1) / nwg;
    nwg = (total + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(gemm_tn256s_kernel<3>, dim3((unsigned)nwg), dim3(512), 3 * 32768, stream, ga, per_wg);
  } else if (big) {
    if (whole) hipLaunchKernelGGL((gemm_tn256g_kernel<3, true, true>), dim3((unsigned)total), dim3(512), 3 * 32768, stream, ga);
    else hipLaunchKernelGGL((gemm_tn256g_kernel<3, true, false>), dim3((unsigned)total), dim3(512), 3 * 32768, stream, ga);
  } else hipLaunchKernelGGL(gemm_tn128g_kernel<3>, dim3((unsigned)total), dim3(256), 3 * 16384, stream, ga);
  ASR_LAUNCH_CHECK();
#ifdef TN_TIMING
  if (big && !sched) {
    long long h[64];
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(tn_dbg), sizeof(h));
    static int shown = 0;
    if (shown++ % 8 == 3)
      for (int w = 0; w < 8; w += 3)
        fprintf(stderr, "tn256g timing wave %d stages %lld: row0 %lld lgkm %lld vmcnt %lld barrier %lld rest %lld looptop %lld (s_memtime ticks)\n", w, h[w * 8 + 7],
                h[w * 8 + 0], h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5]);
  }
#endif
  return ASR_OK;
}

extern "C" int asr_gemm_nn(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, const void* relu_mask,
                           int M, int N, int K, float alpha, int flags, int in_dtype, int out_dtype, hipStream_t stream) {
  ASR_CHECK_ARG(A && B && C && M >= 0 && N >= 0 && K >= 0);
  ASR_CHECK_ARG(in_dtype == ASR_F32 || in_dtype == ASR_BF16);
  ASR_CHECK_ARG(out_dtype == in_dtype || (in_dtype == ASR_BF16 && out_dtype == ASR_F32));
  if (M == 0 || N == 0) return ASR_OK;
  const int esz = in_dtype == ASR_F32 ? 4 : 2, epc = 16 / esz, bkr = 128 / esz;
  if (K <= 0 || lda % epc != 0 || ldb % epc != 0 || !aligned16(A) || !aligned16(B) || ldb < N || lda < (K + bkr - 1) / bkr * bkr)
    return ASR_EUNSUPPORTED;
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.mask = relu_mask;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha;
  p.accumulate = (flags & ASR_GEMM_ACCUMULATE) != 0;
  p.vecC = ((((uintptr_t)C) & 15) == 0) && (ldc % 4 == 0);
  AsrProfScope prof(ASR_OP_GEMM, stream);
  if (in_dtype == ASR_BF16 && out_dtype == ASR_BF16) {    // eight-wave 128 x 128 blocks (csrc/gemm_big.hip) where the shape fills the chip with them
    BigGemmArgs q{};
    q.A = A; q.B = B; q.C = C; q.bias = nullptr; q.mask = relu_mask;
    q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.M = M; q.N = N; q.K = K; q.alpha = alpha;
    q.relu = 0; q.accumulate = p.accumulate; q.out_f32 = 0;
    if (asr_gemm_big_nn(q, stream)) return ASR_OK;
  }
  const int64_t t64 = ceil_div64(M, 64) * ceil_div64(N, 64);
  const int64_t nn_big = asr_tuning("NN_BIG", 1700);      // 128x64 tiles from this many 64x64 tiles on
  const bool big = t64 >= nn_big && M > 64;
  if (in_dtype == ASR_F32) return big ? launch_nn<float, float, 128>(p, stream) : launch_nn<float, float, 64>(p, stream);
  // a launch that leaves a CU with one workgroup or two and walks at least four K steps: the private three-stage ring (NN_RING: the
  // largest number of 64 x 64 blocks that takes it; 0 = never).  profiles/r03_gemm_nn_ring_ab.txt
  if (out_dtype == ASR_BF16 && !big && t64 <= asr_tuning("NN_RING", 512) && K >= 256)
    return launch_nn<bf16_t, bf16_t, 64, 3>(p, stream);
  if (out_dtype == ASR_BF16) return big ? launch_nn<bf16_t, bf16_t, 128>(p, stream) : launch_nn<bf16_t, bf16_t, 64>(p, stream);
  return big ? launch_nn<bf16_t, float, 128>(p, stream) : launch_nn<bf16_t, float, 64>(p, stream);
}

// asr_gemm_nn (bf16, alpha = 1, no mask, no +=) whose epilogue also writes the attention backward's delta (include/asr_hip.h)
extern "C" int asr_gemm_nn_rowdot(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, const void* O, const float* O32,
                                  float* rowdot, int M, int N, int K, int T, int dtype, hipStream_t stream) {
  ASR_CHECK_ARG(A && B && C && (O || O32) && rowdot && M >= 0 && N >= 0 && K >= 0 && T > 0);
  if (dtype != ASR_BF16 || N % 64 != 0 || M % T != 0 || asr_tuning("NN_ROWDOT", 1) == 0) return ASR_EUNSUPPORTED;
  if (M == 0 || N == 0) return ASR_OK;
  if (K <= 0 || lda % 8 != 0 || ldb % 8 != 0 || !aligned16(A) || !aligned16(B) || !aligned16(C) || (O && !aligned16(O)) ||
      (O32 && !aligned16(O32)) || ldb < N || lda < (K + 63) / 64 * 64)
    return ASR_EUNSUPPORTED;
  AsrProfScope prof(ASR_OP_GEMM, stream);
  BigGemmArgs q{};
  q.A = A; q.B = B; q.C = C; q.lda = lda; q.ldb = ldb; q.ldc = N; q.M = M; q.N = N; q.K = K; q.alpha = 1.f;
  q.dot_o = O; q.dot_o32 = O32; q.dot_out = rowdot; q.dot_T = T; q.dot_H = N / 64;
  if (asr_gemm_big_nn(q, stream)) return ASR_OK;
  GemmArgs p{};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = N; p.M = M; p.N = N; p.K = K; p.alpha = 1.f;
  p.vecC = 1;
  p.dot_o = O; p.dot_o32 = O32; p.dot_out = rowdot; p.dot_T = T; p.dot_H = N / 64;
  const int64_t t64 = ceil_div64(M, 64) * ceil_div64(N, 64);
  if (t64 <= asr_tuning("NN_RING", 512) && K >= 256) return launch_nn<bf16_t, bf16_t, 64, 3>(p, stream);
  return launch_nn<bf16_t, bf16_t, 64>(p, stream);
}

// asr_gemm_nn whose epilogue is the second max-pool's backward (include/asr_hip.h): the encoder input projection's data gradient lands
// directly in the un-pooled NHWC gradient of conv.7's output.  Eight-wave 128 x 128 blocks only (csrc/gemm_big.hip).
extern "C" int asr_gemm_nn_poolbwd(const void* A, int64_t lda, const void* Bp, int64_t ldb, const uint8_t* code_cl, void* dy, int M, int K,
                                   int H2, int W2, int C, int dtype, hipStream_t stream) {
  ASR_CHECK_ARG(A && Bp && code_cl && dy && M >= 0 && K >= 0 && H2 > 0 && W2 > 0 && C > 0);
  const int64_t N = (int64_t)H2 * C;
  if (dtype != ASR_BF16 || C % 8 != 0 || M % W2 != 0 || N >= ((int64_t)1 << 30)) return ASR_EUNSUPPORTED;
  if (M == 0) return ASR_OK;
  if (K <= 0 || K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || !aligned16(A) || !aligned16(Bp) || !aligned16(dy) || (((uintptr_t)code_cl) & 7) != 0 ||
      ldb < N || lda < K)
    return ASR_EUNSUPPORTED;
  AsrProfScope prof(ASR_OP_GEMM, stream);
  BigGemmArgs q{};
  q.A = A; q.B = Bp; q.C = dy; q.lda = lda; q.ldb = ldb; q.ldc = N; q.M = M; q.N = (int)N; q.K = K; q.alpha = 1.f;
  q.pool_code = code_cl; q.pool_H2 = H2; q.pool_W2 = W2; q.pool_C = C;
  return asr_gemm_big_nn(q, stream) ? ASR_OK : ASR_EUNSUPPORTED;
}

// ---- one launch for a linear layer's backward: dx (M,K) (+)= dy (M,N) . w (N,K) [ReLU mask]  AND  the partial sums of
// dw (N,K) += dy^T . x (M,K), db (N) += column sums of dy.  bf16 operands.  splits = 0: chosen here.
static int nn_tn_splits(int M, int splits, int* m_per_split) {
  const int stages = (M + 63) / 64;
  if (splits <= 0) {
    const int per = (int)asr_tuning("NNTN_STAGES", 16);           // 64-row stages of one weight-gradient workgroup
    splits = (stages + per - 1) / (per > 0 ? per : 16);
  }
  splits = splits < 1 ? 1 : (splits > stages ? stages : splits);
  const int mps = ((stages + splits - 1) / splits) * 64;
  if (m_per_split) *m_per_split = mps;
  return (M + mps - 1) / mps;                                      // no empty slice
}

extern "C" int asr_gemm_nn_tn_splits(int M, int splits) { return M > 0 ? nn_tn_splits(M, splits, nullptr) : 0; }

extern "C" int64_t asr_gemm_nn_tn_workspace(int M, int N, int K, int splits) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)nn_tn_splits(M, splits, nullptr) * ((N + 63) / 64) * ((K + 63) / 64) * 4096;
}

extern "C" int asr_gemm_nn_tn(const void* dy, int64_t ld_dy, const void* w, int64_t ldw, const void* x, int64_t ldx, void* dx,
                              int64_t ld_dx, const void* relu_mask, float* db, float* workspace, int64_t workspace_floats, int M,
                              int N, int K, int flags, int splits, int dtype, const float* fold_ws, float* fold_dw,
                              int64_t fold_ld, int fold_N, int fold_K, int fold_splits, hipStream_t stream) {
  ASR_CHECK_ARG(dy && w && x && dx && workspace && M > 0 && N > 0 && K > 0);
  if (dtype != ASR_BF16) return ASR_EUNSUPPORTED;
  if (ld_dy % 8 != 0 || ldw % 8 != 0 || ldx % 8 != 0 || !aligned16(dy) || !aligned16(w) || !aligned16(x) || ldw < K ||
      ld_dy < (N + 63) / 64 * 64 || ld_dy >= ((int64_t)1 << 22) || ldx >= ((int64_t)1 << 22))
    return ASR_EUNSUPPORTED;
  int m_per_split = 0;
  splits = nn_tn_splits(M, splits, &m_per_split);
  TnArgs t{};
  t.A = dy; t.B = x; t.C = nullptr; t.colsum = db; t.ws = workspace;
  t.lda = ld_dy; t.ldb = ldx; t.ldc = 0;
  t.M = M; t.N = N; t.K = K;
  t.tiles_k = (K + 63) / 64;
  t.ntiles = ((N + 63) / 64) * t.tiles_k;
  t.m_per_split = m_per_split;
  if (workspace_floats < (int64_t)splits * t.ntiles * 4096) return ASR_EINVAL;
  GemmArgs p{};
  p.A = dy; p.B = w; p.C = dx; p.mask = relu_mask;
  p.lda = ld_dy; p.ldb = ldw; p.ldc = ld_dx;
  p.M = M; p.N = K; p.K = N; p.alpha = 1.f;
  p.accumulate = (flags & ASR_GEMM_ACCUMULATE) != 0;
  p.vecC = ((((uintptr_t)dx) & 15) == 0) && (ld_dx % 4 == 0);
  p.tiles_n = (K + 63) / 64;
  const int64_t t64 = ceil_div64(M, 64) * p.tiles_n;
  const bool big = t64 >= asr_tuning("NN_BIG", 1700) && M > 64;
  const int bm = big ? 128 : 64;
  p.ntiles = ((M + bm - 1) / bm) * p.tiles_n;
  const int n_tn = t.ntiles * splits;
  size_t lds = (size_t)(bm * 128 + 64 * 128);
  const size_t cl = (size_t)bm * (64 * 4 + 16);
  if (cl > lds) lds = cl;
  AsrProfScope prof(ASR_OP_GEMM, stream);
  TnFoldArgs f{};
  int n_fold = 0;
  if (fold_ws) {
    ASR_CHECK_ARG(fold_dw && fold_N > 0 && fold_K > 0 && fold_splits > 0);
    f.ws = fold_ws; f.C = fold_dw; f.ldc = fold_ld; f.N = fold_N; f.K = fold_K; f.splits = fold_splits;
    n_fold = ((fold_N + 63) / 64) * ((fold_K + 63) / 64) * 4;
  }
  const dim3 grid((unsigned)(n_tn + p.ntiles + n_fold));
  if (big) hipLaunchKernelGGL(gemm_nn_tnq_kernel<128>, grid, dim3(256), lds, stream, p, t, f, n_tn, p.ntiles);
  else hipLaunchKernelGGL(gemm_nn_tnq_kernel<64>, grid, dim3(256), lds, stream, p, t, f, n_tn, p.ntiles);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_tn_reduce_multi(const float* const* workspaces, float* const* dw, const int64_t* ld_dw, const int* N, const int* K,
                                   const int* splits, int count, hipStream_t stream) {
  ASR_CHECK_ARG(count >= 0 && (count == 0 || (workspaces && dw && ld_dw && N && K && splits)));
  for (int base = 0; base < count; base += TN_MULTI) {
    const int n = count - base < TN_MULTI ? count - base : TN_MULTI;
    TnMultiArgs q{};
    int max_tiles = 0;
    for (int i = 0; i < n; ++i) {
      ASR_CHECK_ARG(workspaces[base + i] && dw[base + i] && N[base + i] > 0 && K[base + i] > 0 && splits[base + i] > 0);
      q.ws[i] = workspaces[base + i]; q.C[i] = dw[base + i]; q.ldc[i] = (int)ld_dw[base + i];
      q.N[i] = N[base + i]; q.K[i] = K[base + i]; q.splits[i] = splits[base + i];
      const int nt = ((q.N[i] + 63) / 64) * ((q.K[i] + 63) / 64);
      if (nt > max_tiles) max_tiles = nt;
    }
    AsrProfScope prof(ASR_OP_GEMM, stream);
    hipLaunchKernelGGL(tn_reduce_multi_kernel, dim3((unsigned)(max_tiles * 4), (unsigned)n), dim3(256), 0, stream, q);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}
