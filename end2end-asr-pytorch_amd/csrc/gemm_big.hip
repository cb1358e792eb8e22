// Eight-wave GEMMs for the linear layers of the Transformer (reference: models/common_layers.py:136-142,181-198 nn.Linear /
// Conv1d(k=1) through autograd): C = act(alpha A B^T + bias) ("NT", forward) and C (+)= (A B) [masked] ("NN", data gradient).
//
// Why a second family next to csrc/gemm.hip's four-wave kernels.  The model's contractions are SHORT (K = 512: 8 steps of 64) over
// 3200 - 6400 rows: a 64 x 64 or 128 x 64 block issues 64 - 128 MFMAs per wave in its whole life, so every launch is mostly
// fill / drain (measured in the replayed step: 6400 x 2048 x 512 in 25 us = 530 TF/s, 3200 x 512 x 2048 in 31 us = 216 TF/s).
// Here one 512-thread workgroup owns a 256 x 256 (or 128 x 128) block of C:
//   * 8 waves as 2 (m) x 4 (n): a wave's quadrant is 128 x 64 (64 x 32), i.e. 64 (16) MFMAs per 64-deep step against 24 (12)
//     16-byte operand reads -- two waves per SIMD, one reading while the other multiplies;
//   * both operands come in by LDS-DMA (global_load_lds_dwordx4, hand-issued: csrc/gemm.hip tn_dma) in 128-byte rows, the
//     16-byte slot XOR-ed with (row & 7) on the SOURCE side and again on the fragment read -- conflict-free ds_read_b128;
//     NS stages in a ring with counted vmcnt, ONE s_barrier per step (the barrier that publishes step t also frees the
//     buffer step t + NS - 1 is loaded into);
//   * the epilogue stages the block in the OUTPUT type through the same LDS and stores 16-byte row pieces.
// The NN form reads its B operand (the weight, (K, N) row-major = contraction index along rows) with ds_read_b64_tr_b16.
#include "common.h"
#include "gemm_big.h"

namespace {

__device__ __forceinline__ void big_dma(unsigned lds_wave_base, const unsigned char* src) {
  unsigned keep;      // M0 saved and restored: neutral for whatever the compiler keeps there
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_wave_base), "v"(src)
               : "memory");
}
template <int N> __device__ __forceinline__ void big_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// One operand tile of R rows x 128 bytes: piece i of a thread = chunk c = i * 512 + tid -> (row c >> 3, slot c & 7); the source is
// chunk slot ^ (row & 7) of global row (row0 + row), clamped to the last valid row (rows past the edge are never stored).
template <int R>
__device__ __forceinline__ void big_src(const unsigned char* base, int64_t ld_bytes, int row0, int row_limit, int tid,
                                        const unsigned char* (&src)[R / 64]) {
#pragma unroll
  for (int i = 0; i < R / 64; ++i) {
    const int c = i * 512 + tid, row = c >> 3, slot = (c & 7) ^ (row & 7);
    int gr = row0 + row;
    gr = gr < row_limit ? gr : row_limit - 1;
    src[i] = base + (int64_t)gr * ld_bytes + slot * 16;
  }
}

// ------------------------------------------------------------------------------------------------ NT
// TO: bf16_t or float.  Requirements (checked by the launcher): K % 64 == 0, 16-byte aligned A / B rows, C rows in whole 16-byte pieces.
template <int BM, int BN, int NS, typename TO>
__global__ __launch_bounds__(512, 2) void gemm_big_nt_kernel(BigGemmArgs p, int tiles_n, int ntiles) {
  constexpr int WM = BM / 2, WN = BN / 4, FM = WM / 16, FN = WN / 16;
  constexpr int TA = BM * 128, TB = BN * 128, STAGE = TA + TB;
  constexpr int LA = BM / 64, LB = BN / 64, LPS = LA + LB;          // LDS-DMA pieces per thread and stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware order: blocks b, b + 8, ... share an XCD (private L2) -> consecutive tiles (n fastest: they share their A rows)
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  if (tile >= ntiles) return;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = p.K / 64;

  const unsigned char* srcA[LA];
  const unsigned char* srcB[LB];
  big_src<BM>(static_cast<const unsigned char*>(p.A), p.lda * 2, m0, p.M, tid, srcA);
  big_src<BN>(static_cast<const unsigned char*>(p.B), p.ldb * 2, n0, p.N, tid, srcB);
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wave_lds = smem_base + (unsigned)wave * 1024u;
  auto stage = [&](int kt) __attribute__((always_inline)) {
    const unsigned sl = wave_lds + (unsigned)((kt % NS) * STAGE);
    const int64_t kb = (int64_t)kt * 128;
#pragma unroll
    for (int i = 0; i < LA; ++i) big_dma(sl + i * 8192, srcA[i] + kb);
#pragma unroll
    for (int i = 0; i < LB; ++i) big_dma(sl + TA + i * 8192, srcB[i] + kb);
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row (w * W + f * 16 + lr), 16-byte slot (ms * 4 + g) ^ (lr & 7): two per lane, the rest immediates
  const int offA = (wm * WM + lr) * 128, offB = TA + (wn * WN + lr) * 128;
  const int sw0 = ((g) ^ (lr & 7)) << 4, sw1 = ((4 + g) ^ (lr & 7)) << 4;

#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) stage(st);
  for (int kt = 0; kt < nk; ++kt) {
    // step kt has landed once at most the pieces of the later steps are outstanding (LPS per step and thread, retired in order)
    const int ahead = min(NS - 2, nk - 1 - kt);
    if (NS >= 4 && ahead >= 2) big_wait_vmcnt<2 * LPS>();
    else if (NS >= 3 && ahead >= 1) big_wait_vmcnt<LPS>();
    else big_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();              // step kt visible to every wave; every wave is done reading step kt - 1
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) stage(kt + NS - 1);   // into the buffer step kt - 1 used
    const unsigned char* s = smem + (kt % NS) * STAGE;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int sw = ms ? sw1 : sw0;
      uint4 a[FM], b[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const uint4*>(s + offB + j * 2048 + sw);
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const uint4*>(s + offA + i * 2048 + sw);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma16<bf16_t>(acc[i][j], b[j], a[i]);     // operands swapped: the fragment is C^T, see the epilogue
    }
  }
  __syncthreads();                               // every wave is done with the operand stages: the epilogue reuses them

  // ---- epilogue: alpha / bias / ReLU on the accumulators, the block staged in the output type, 16-byte row pieces out.
  // fp32 output of a 256-row block: two passes of 128 rows (the block does not fit the LDS in fp32).
  constexpr int OESZ = (int)sizeof(TO), EPC = 16 / OESZ;
  constexpr int OP = BN * OESZ + 16;             // row pitch: + 16 bytes keeps 16-byte alignment and staggers the banks
  constexpr int PASSES = (BM * OP > 140 * 1024) ? 2 : 1;
  constexpr int PR = BM / PASSES;                // rows per pass
  // The MFMAs ran with the operands swapped (first = B rows, second = A rows), so lane (lr, g) holds C[m = lr][n = 4 g + r]: four
  // CONSECUTIVE columns of one row -- one 8-byte (bf16) or 16-byte (fp32) LDS write per fragment instead of four 2-byte ones.
  TO* C = static_cast<TO*>(p.C);
  float bv[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int col = n0 + wn * WN + j * 16 + g * 4 + r;
      bv[j][r] = (p.bias != nullptr && col < p.N) ? p.bias[col] : 0.f;
    }
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if (PASSES == 1 || wm == pass) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r] * p.alpha + bv[j][r];
            if (p.relu) v[r] = fmaxf(v[r], 0.f);
          }
          const int row = (PASSES == 1 ? wm * WM : 0) + i * 16 + lr;
          unsigned char* dst = smem + row * OP + (wn * WN + j * 16 + g * 4) * OESZ;
          if constexpr (OESZ == 2) {
            uint2 w;
            w.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            w.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            *reinterpret_cast<uint2*>(dst) = w;
          } else {
            *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
    }
    __syncthreads();
    constexpr int CPR = BN / EPC;                // 16-byte pieces per row
    for (int c = tid; c < PR * CPR; c += 512) {
      const int row = c / CPR, pc = c % CPR;
      const int grow = m0 + pass * PR + row, gcol = n0 + pc * EPC;
      if (grow < p.M && gcol < p.N) {            // N is a whole number of pieces (launcher)
        uint4 v = *reinterpret_cast<const uint4*>(smem + row * OP + pc * 16);
        TO* dst = C + (int64_t)grow * p.ldc + gcol;
        if (p.accumulate) {
          Chunk<TO> o, n;
          o.v = *reinterpret_cast<const uint4*>(dst);
          n.v = v;
#pragma unroll
          for (int e = 0; e < EPC; ++e) n.e[e] = DT<TO>::to(DT<TO>::from(o.e[e]) + DT<TO>::from(n.e[e]));
          v = n.v;
        }
        *reinterpret_cast<uint4*>(dst) = v;
      }
    }
    if (PASSES > 1) __syncthreads();
  }
}

template <int BM, int BN, int NS, typename TO>
void launch_nt(const BigGemmArgs& p, hipStream_t s) {
  constexpr int OP = BN * (int)sizeof(TO) + 16;
  constexpr int PASSES = (BM * OP > 140 * 1024) ? 2 : 1;
  constexpr size_t lds_stage = (size_t)NS * (BM + BN) * 128, lds_out = (size_t)(BM / PASSES) * OP;
  constexpr size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_nt_kernel<BM, BN, NS, TO>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    granted = true;
  }
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
  hipLaunchKernelGGL((gemm_big_nt_kernel<BM, BN, NS, TO>), dim3((unsigned)ntiles), dim3(512), lds, s, p, tiles_n, ntiles);
}

// ------------------------------------------------------------------------------------------------ NN
// C (M, N) bf16 (+)= A (M, K) . B (K, N), optionally zeroed where mask (laid out like C) <= 0: a linear layer's data gradient
// dX = dY . W with W in its natural (out_features, in_features) layout (reference: nn.Linear backward through autograd).
// 128 x 128 blocks.  The A side is the NT kernel's; the B tile is 64 contraction rows x 128 columns (256-byte rows, 16-byte slot
// XOR (row & 7)), read with ds_read_b64_tr_b16 so that a lane gets 8 consecutive contraction indices of ONE column -- the same
// (column, k-chunk) operand layout the NT kernel reads with plain 16-byte loads.  The block is staged in fp32 for the epilogue, so
// that `+=` into a bf16 destination and the ReLU mask act on unrounded sums (one rounding, like the four-wave kernel).
template <int NS>
__global__ __launch_bounds__(512, 2) void gemm_big_nn_kernel(BigGemmArgs p, int tiles_n, int ntiles) {
  constexpr int BM = 128, BN = 128;
  constexpr int WM = BM / 2, WN = BN / 4, FM = WM / 16, FN = WN / 16;
  constexpr int TA = BM * 128, TB = 64 * BN * 2, STAGE = TA + TB;
  constexpr int LA = BM / 64, LB = TB / 8192, LPS = LA + LB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qn = nwg >> 3, rn = nwg & 7;
  const int tile = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
  if (tile >= ntiles) return;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk = p.K / 64;

  const unsigned char* srcA[LA];
  const unsigned char* srcB[LB];
  big_src<BM>(static_cast<const unsigned char*>(p.A), p.lda * 2, m0, p.M, tid, srcA);
  const int nchunks = p.N / 8;                     // N is a whole number of 16-byte chunks (launcher)
#pragma unroll
  for (int i = 0; i < LB; ++i) {
    const int c = i * 512 + tid, row = c >> 4, slot = (c & 15) ^ (row & 7);
    int gc = n0 / 8 + slot;
    gc = gc < nchunks ? gc : nchunks - 1;          // columns past N are never stored
    srcB[i] = static_cast<const unsigned char*>(p.B) + (int64_t)row * p.ldb * 2 + (int64_t)gc * 16;
  }
  const int64_t stepB = (int64_t)64 * p.ldb * 2;
  const unsigned smem_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const unsigned wave_lds = smem_base + (unsigned)wave * 1024u;
  auto stage = [&](int kt) __attribute__((always_inline)) {
    const unsigned sl = wave_lds + (unsigned)((kt % NS) * STAGE);
#pragma unroll
    for (int i = 0; i < LA; ++i) big_dma(sl + i * 8192, srcA[i] + (int64_t)kt * 128);
#pragma unroll
    for (int i = 0; i < LB; ++i) big_dma(sl + TA + i * 8192, srcB[i] + kt * stepB);
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int offA = (wm * WM + lr) * 128;
  const int sw0 = ((g) ^ (lr & 7)) << 4, sw1 = ((4 + g) ^ (lr & 7)) << 4;
  // B fragment j, macro step ms: rows 32 ms + 8 g + (lr >> 2) and + 4, columns wn * WN + 16 j + 4 (lr & 3) ..
  const int brow = 8 * g + (lr >> 2), bcol = wn * WN + 4 * (lr & 3);
  const int bhalf = ((bcol >> 2) & 1) * 8;

#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < nk) stage(st);
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = min(NS - 2, nk - 1 - kt);
    if (NS >= 4 && ahead >= 2) big_wait_vmcnt<2 * LPS>();
    else if (NS >= 3 && ahead >= 1) big_wait_vmcnt<LPS>();
    else big_wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) stage(kt + NS - 1);
    const unsigned char* s = smem + (kt % NS) * STAGE;
    const unsigned char* sB = s + TA;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      const int sw = ms ? sw1 : sw0;
      uint4 a[FM], b[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int r0 = 32 * ms + brow, chunk = (bcol + j * 16) >> 3;
        const uint2 lo = asr_lds_read_tr16(sB + r0 * 256 + ((chunk ^ (r0 & 7)) << 4) + bhalf);
        const uint2 hi = asr_lds_read_tr16(sB + (r0 + 4) * 256 + ((chunk ^ ((r0 + 4) & 7)) << 4) + bhalf);
        b[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const uint4*>(s + offA + i * 2048 + sw);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma16<bf16_t>(acc[i][j], b[j], a[i]);
    }
  }
  __syncthreads();

  // ---- epilogue: the block in fp32 through LDS, then 16-byte pieces of 8 bf16: mask, optional +=, one rounding
  constexpr int OP = BN * 4 + 16;
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int row = wm * WM + i * 16 + lr;
      *reinterpret_cast<float4*>(smem + row * OP + (wn * WN + j * 16 + g * 4) * 4) =
          make_float4(acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha);
    }
  __syncthreads();
  bf16_t* C = static_cast<bf16_t*>(p.C);
  const bf16_t* Msk = static_cast<const bf16_t*>(p.mask);
  constexpr int CPR = BN / 8;
  if (p.pool_code) {
    // The second max-pool's backward (asr_gemm_nn_poolbwd).  A piece = 8 channels of pooled pixel (h2, w2) of image b (row m = (b, w2),
    // column n = (h2, c)); it is routed by its 8 selection bytes to the 2 x 2 window, and all four positions are written.  The loop
    // runs over OUTPUT pixels, not pooled pieces: piece fastest, then the window column, then the pooled row -- a wave's 64 stores are
    // the 4 consecutive pixels (2 w2, 2 w2 + 1, 2 w2 + 2, 2 w2 + 3) of one image row = 1 KB of consecutive bytes (a first version
    // stored the four positions of a piece from one thread: 256-byte segments with 256-byte holes per instruction, 2.6 TB/s).
    // thread <-> (piece column pc, window column dxk, pooled rows r0 + 16 j): its 8 pieces' selection bytes are loaded up front (8
    // independent loads in flight: one round trip, not one per store -- a first version loaded inside the store loop and the launch
    // became a chain of 16 L2 latencies per workgroup, 114 us), then the two image rows of the window are written
    static_assert(CPR == 16, "the pooled-gradient epilogue maps 16 pieces per row");
    const int pc = tid & 15, dxk = (tid >> 4) & 1, r0 = tid >> 5;
    const int gcol = n0 + pc * 8;
    const int h2 = gcol / p.pool_C, cc = gcol - h2 * p.pool_C;
    uint2 kk[8];
    bf16_t* dst[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int grow = m0 + r0 + 16 * j;
      const bool ok = grow < p.M && gcol < p.N;
      const int gr = ok ? grow : 0;
      const int bb = gr / p.pool_W2, w2 = gr - bb * p.pool_W2;
      kk[j] = ok ? *reinterpret_cast<const uint2*>(p.pool_code + ((int64_t)gr * p.pool_H2 + h2) * p.pool_C + cc) : make_uint2(0xffffffffu, 0xffffffffu);
      dst[j] = ok ? C + (((int64_t)bb * 2 * p.pool_H2 + 2 * h2) * (2 * p.pool_W2) + 2 * w2 + dxk) * p.pool_C + cc : nullptr;
    }
    const int64_t rowp = (int64_t)2 * p.pool_W2 * p.pool_C;
#pragma unroll
    for (int dyk = 0; dyk < 2; ++dyk) {
      const uint32_t want = (uint32_t)(dyk * 2 + dxk + 1);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (dst[j] == nullptr) continue;
        const int row = r0 + 16 * j;
        const float4 v0 = *reinterpret_cast<const float4*>(smem + row * OP + pc * 32);
        const float4 v1 = *reinterpret_cast<const float4*>(smem + row * OP + pc * 32 + 16);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        Chunk<bf16_t> q;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const uint32_t code = ((e < 4 ? kk[j].x : kk[j].y) >> (8 * (e & 3))) & 0xffu;
          q.e[e] = code == want ? f32_to_bf16(v[e]) : (bf16_t)0;
        }
        *reinterpret_cast<uint4*>(dst[j] + dyk * rowp) = q.v;
      }
    }
    return;
  }
  for (int c = tid; c < BM * CPR; c += 512) {
    const int row = c / CPR, pc = c % CPR;
    const int grow = m0 + row, gcol = n0 + pc * 8;
    if (grow < p.M && gcol < p.N) {
      const float4 v0 = *reinterpret_cast<const float4*>(smem + row * OP + pc * 32);
      const float4 v1 = *reinterpret_cast<const float4*>(smem + row * OP + pc * 32 + 16);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const int64_t off = (int64_t)grow * p.ldc + gcol;
      if (Msk) {
        Chunk<bf16_t> m;
        m.v = *reinterpret_cast<const uint4*>(Msk + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32(m.e[e]) > 0.f ? v[e] : 0.f;
      }
      Chunk<bf16_t> o;
      if (p.accumulate) {
        o.v = *reinterpret_cast<const uint4*>(C + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bf16_to_f32(o.e[e]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o.e[e] = f32_to_bf16(v[e]);
      *reinterpret_cast<uint4*>(C + off) = o.v;
      if (p.dot_out) {
        // the attention backward's delta = rowsum(dO * O) per head from the block that IS dO (csrc/attention_fast.hip
        // attn_delta_bf16_d64_kernel, same lanes per row, same order of additions: the same bits): 8 lanes = the 64 columns of a head
        const int64_t ooff = (int64_t)grow * p.N + gcol;
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
        if ((pc & 7) == 0) p.dot_out[((int64_t)(grow / p.dot_T) * p.dot_H + (gcol >> 6)) * p.dot_T + grow % p.dot_T] = acc;
      }
    }
  }
}

template <int NS>
void launch_nn(const BigGemmArgs& p, hipStream_t s) {
  constexpr size_t lds_stage = (size_t)NS * (128 * 128 + 64 * 128 * 2), lds_out = (size_t)128 * (128 * 4 + 16);
  constexpr size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
  static bool granted = false;
  if (!granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_nn_kernel<NS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    granted = true;
  }
  const int tiles_m = (p.M + 127) / 128, tiles_n = (p.N + 127) / 128, ntiles = tiles_m * tiles_n;
  hipLaunchKernelGGL((gemm_big_nn_kernel<NS>), dim3((unsigned)ntiles), dim3(512), lds, s, p, tiles_n, ntiles);
}

}  // namespace

bool asr_gemm_big_nt(const BigGemmArgs& p, hipStream_t stream) {
  const int mode = (int)asr_tuning("GEMM_BIG", 1);            // 0: off; 1: automatic; 256 / 128: force the block size
  if (mode == 0 || p.mask != nullptr || (p.accumulate && !p.out_f32)) return false;   // (+= in bf16 keeps the four-wave kernel's single rounding)
  const int oesz = p.out_f32 ? 4 : 2, epc = 16 / oesz;
  if (p.K <= 0 || p.K % 64 != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0 || !aligned16(p.A) || !aligned16(p.B) || !aligned16(p.C) ||
      p.ldc % epc != 0 || p.N % epc != 0 || p.lda < p.K || p.ldb < p.K)
    return false;
  // block size (profiles/r03_gemm_big_ab.txt): the 128 x 128 form wins wherever the four-wave kernels were latency-bound -- long
  // contractions (K >= 2048: 4 stages) and tall operands; 256 x 256 pays for fp32 output (the vocabulary projection) only
  const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
  int tile = mode == 256 || mode == 128 ? mode : (p.out_f32 && p.N >= 2048 ? 256 : 128);
  if (tile == 128 && mode == 1 && t128 < (p.K >= 2048 ? 150 : asr_tuning("GEMM_BIG_MIN", 128))) return false;      // too few blocks: 64 x 64 four-wave blocks (with their ring, gemm.hip launch_fast) fill the chip better -- 48 -> 128: headline 6.02 -> 5.98 ms
  if (tile == 256) {
    if (p.out_f32) launch_nt<256, 256, 2, float>(p, stream); else launch_nt<256, 256, 2, bf16_t>(p, stream);
  } else {
    // stages of the 128 x 128 form: 2 (64 KB: two workgroups per CU), 3 (default) or 4 (one workgroup per CU, deeper prefetch)
    // two stages = 64 KB = two workgroups per CU: best whenever there are more blocks than CUs; else deeper prefetch for long K
    const int ns = (int)asr_tuning("GEMM_BIG_NS", (t128 > 256 || p.K < 2048) ? 2 : 4);
    if (p.out_f32) launch_nt<128, 128, 3, float>(p, stream);
    else if (ns == 2) launch_nt<128, 128, 2, bf16_t>(p, stream);
    else if (ns == 4) launch_nt<128, 128, 4, bf16_t>(p, stream);
    else launch_nt<128, 128, 3, bf16_t>(p, stream);
  }
  return hipGetLastError() == hipSuccess;
}

bool asr_gemm_big_nn(const BigGemmArgs& p, hipStream_t stream) {
  const int mode = (int)asr_tuning("GEMM_BIG_NN", 1);         // 0: off; 1: automatic; 2: always (tests)
  if (mode == 0 || p.out_f32 || p.bias != nullptr || p.relu) return false;
  if (p.pool_code && (p.accumulate || p.mask || p.dot_out || p.pool_C % 8 != 0 || p.N != p.pool_H2 * p.pool_C || p.M % p.pool_W2 != 0)) return false;
  if (p.dot_out && (p.N % 64 != 0 || p.accumulate || p.mask || p.N != p.dot_H * 64)) return false;
  if (p.K <= 0 || p.K % 64 != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0 || p.ldc % 8 != 0 || p.N % 8 != 0 || !aligned16(p.A) || !aligned16(p.B) ||
      !aligned16(p.C) || (p.mask && !aligned16(p.mask)) || p.lda < p.K || p.ldb < p.N)
    return false;
  const int64_t t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
  // (profiles/r03_gemm_big_nn_shapes.txt: below ~150 blocks the four-wave 64 x 64 kernel fills the chip better -- 3200 x 512 over K = 1536:
  // 13.8 vs 19.0 us)
  if (mode == 1 && t128 < 150 && !p.pool_code) return false;      // (the pooled-gradient epilogue exists in this kernel only)
  const int ns = (int)asr_tuning("GEMM_BIG_NS", (t128 > 256 || p.K < 2048) ? 2 : 4);
  if (ns == 2) launch_nn<2>(p, stream); else if (ns == 4) launch_nn<4>(p, stream); else launch_nn<3>(p, stream);
  return hipGetLastError() == hipSuccess;
}
