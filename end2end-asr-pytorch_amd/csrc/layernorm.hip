// Fused sub-layer epilogue:  z = dropout(y) + residual ; out = (LN(z)*gamma + beta + post_add) * row_keep
// (reference: models/common_layers.py:140-141 and :197-198; models/asr/transformer.py:172-173, :198, :201,
//  :536-543).  One wave per row, row kept in registers, fp32 statistics (biased variance, eps inside sqrt).
// HBM-bound: algorithmic bytes per row = (2 reads + 2 writes) * D * sizeof(T).
#include "common.h"

namespace {


// fp32 parameter vector: EPC consecutive values starting at c0 (16-byte aligned: c0 is a multiple of EPC >= 4)
template <int EPC> __device__ __forceinline__ void load_params(const float* __restrict__ p, int c0, float* out) {
#pragma unroll
  for (int j = 0; j < EPC; j += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + c0 + j);
    out[j] = v.x; out[j + 1] = v.y; out[j + 2] = v.z; out[j + 3] = v.w;
  }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(T* __restrict__ y_z, const T* __restrict__ res,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ post, int post_period,
                                                         const uint8_t* __restrict__ keep, T* __restrict__ out,
                                                         float* __restrict__ mean, float* __restrict__ rstd, int M, int D,
                                                         float eps, uint32_t thr, float inv_keep, uint64_t seed0,
                                                         const uint64_t* __restrict__ seed_dev) {
  constexpr int EPC = DT<T>::EPC;
  const uint64_t seed = asr_mix_seed(seed0, seed_dev);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  T* yr = y_z + (int64_t)row * D;
  // every load of the row is issued up front (optional operands read a valid dummy address and are masked by a flag): the
  // row costs ONE memory round trip, not one per operand
  const T* rr = res ? res + (int64_t)row * D : yr;
  const float* pr = post ? post + (int64_t)(row % post_period) * D : beta;
  const float fres = res ? 1.f : 0.f, fpost = post ? 1.f : 0.f;
  const float kp = keep ? (keep[row] ? 1.f : 0.f) : 1.f;
  Chunk<T> cy[NCH], cr[NCH];
  float gm[NCH][EPC], bt[NCH][EPC], ps[NCH][EPC];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = (ch * 64 + lane) * EPC;
    const int cc = c0 < D ? c0 : 0;
    cy[ch].v = *reinterpret_cast<const uint4*>(yr + cc);
    cr[ch].v = *reinterpret_cast<const uint4*>(rr + cc);
    load_params<EPC>(gamma, cc, gm[ch]);
    load_params<EPC>(beta, cc, bt[ch]);
    load_params<EPC>(pr, cc, ps[ch]);
  }
  float z[NCH * EPC];
  float s = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = (ch * 64 + lane) * EPC;
    if (c0 < D) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        float v = DT<T>::from(cy[ch].e[j]);
        if (thr) v = asr_keep(seed, (uint64_t)row * D + c0 + j, thr) ? v * inv_keep : 0.f;
        v += fres * DT<T>::from(cr[ch].e[j]);
        // z is what backward sees: round it to the storage type first so fwd and bwd agree bit for bit
        cy[ch].e[j] = DT<T>::to(v);
        v = DT<T>::from(cy[ch].e[j]);
        z[ch * EPC + j] = v;
        s += v;
      }
      *reinterpret_cast<uint4*>(yr + c0) = cy[ch].v;
    } else {
#pragma unroll
      for (int j = 0; j < EPC; ++j) z[ch * EPC + j] = 0.f;
    }
  }
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = (ch * 64 + lane) * EPC;
    if (c0 < D) {
#pragma unroll
      for (int j = 0; j < EPC; ++j) { const float d = z[ch * EPC + j] - mu; q += d * d; }
    }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = (ch * 64 + lane) * EPC;
    if (c0 < D) {
      Chunk<T> co;
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        const float v = (z[ch * EPC + j] - mu) * rs * gm[ch][j] + bt[ch][j] + fpost * ps[ch][j];
        co.e[j] = DT<T>::to(v * kp);
      }
      *reinterpret_cast<uint4*>(out + (int64_t)row * D + c0) = co.v;
    }
  }
}

// Backward.  Each block owns `rows_per_block` consecutive rows; lane l of every wave owns the same columns, so
// dgamma/dbeta partials live in registers across rows and are reduced across the block's 4 waves through LDS.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ z,
                                                         const float* __restrict__ mean, const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, const uint8_t* __restrict__ keep,
                                                         T* __restrict__ d_res, T* __restrict__ d_y, float* dgamma,
                                                         float* dbeta, float* __restrict__ partial, int M, int D,
                                                         int rows_per_block, uint32_t thr,
                                                         float inv_keep, uint64_t seed0, const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = asr_mix_seed(seed0, seed_dev);
  constexpr int EPC = DT<T>::EPC;
  constexpr int NPL = NCH * EPC;   // elements per lane
  extern __shared__ float red[];   // [4][2*D] -> only waves 1..3 write
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[NPL], ab[NPL];
#pragma unroll
  for (int i = 0; i < NPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  const int r_beg = blockIdx.x * rows_per_block, r_end = min(M, r_beg + rows_per_block);
  // gamma does not depend on the row: loaded once per thread
  float gm[NCH][EPC];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = (ch * 64 + lane) * EPC;
    load_params<EPC>(gamma, c0 < D ? c0 : 0, gm[ch]);
  }
  const uint8_t* kptr = keep ? keep : reinterpret_cast<const uint8_t*>(gamma);     // dummy byte source when there is no row mask
  for (int row = r_beg + wave; row < r_end; row += 4) {
    // all loads of the row up front and unconditional (clamped column, dummy mask byte): one memory round trip per row
    Chunk<T> cdv[NCH], czv[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = (ch * 64 + lane) * EPC;
      const int cc = c0 < D ? c0 : 0;
      cdv[ch].v = *reinterpret_cast<const uint4*>(dout + (int64_t)row * D + cc);
      czv[ch].v = *reinterpret_cast<const uint4*>(z + (int64_t)row * D + cc);
    }
    const uint8_t kb = kptr[keep ? row : 0];
    const float mu = mean[row], rs = rstd[row];
    const float kp = keep ? (kb ? 1.f : 0.f) : 1.f;
    float xh[NPL], dyh[NPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = (ch * 64 + lane) * EPC;
      if (c0 < D) {
        const Chunk<T>& cd = cdv[ch];
        const Chunk<T>& cz = czv[ch];
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const float go = DT<T>::from(cd.e[j]) * kp;
          const float x = (DT<T>::from(cz.e[j]) - mu) * rs;
          const float gy = go * gm[ch][j];
          xh[ch * EPC + j] = x; dyh[ch * EPC + j] = gy;
          s1 += gy; s2 += gy * x;
          ag[ch * EPC + j] += go * x; ab[ch * EPC + j] += go;
        }
      }
    }
    s1 = wave_sum(s1) / (float)D;
    s2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = (ch * 64 + lane) * EPC;
      if (c0 < D) {
        Chunk<T> cr, cy;
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          const float dz = rs * (dyh[ch * EPC + j] - s1 - xh[ch * EPC + j] * s2);
          cr.e[j] = DT<T>::to(dz);
          float dy = dz;
          if (thr) dy = asr_keep(seed, (uint64_t)row * D + c0 + j, thr) ? dz * inv_keep : 0.f;
          cy.e[j] = DT<T>::to(dy);
        }
        *reinterpret_cast<uint4*>(d_res + (int64_t)row * D + c0) = cr.v;
        if (d_y && d_y != d_res) *reinterpret_cast<uint4*>(d_y + (int64_t)row * D + c0) = cy.v;
      }
    }
  }
  // block reduction of the column partials
  if (wave > 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = (ch * 64 + lane) * EPC;
      if (c0 < D) {
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          red[(wave - 1) * 2 * D + c0 + j] = ag[ch * EPC + j];
          red[(wave - 1) * 2 * D + D + c0 + j] = ab[ch * EPC + j];
        }
      }
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = (ch * 64 + lane) * EPC;
      if (c0 < D) {
#pragma unroll
        for (int j = 0; j < EPC; ++j) {
          float g = ag[ch * EPC + j], b = ab[ch * EPC + j];
#pragma unroll
          for (int w = 0; w < 3; ++w) { g += red[w * 2 * D + c0 + j]; b += red[w * 2 * D + D + c0 + j]; }
          if (partial) {                 // two-stage reduction: no atomics on the 2*D hot addresses
            partial[(int64_t)blockIdx.x * 2 * D + c0 + j] = g;
            partial[(int64_t)blockIdx.x * 2 * D + D + c0 + j] = b;
          } else {
            atomicAdd(dgamma + c0 + j, g);
            atomicAdd(dbeta + c0 + j, b);
          }
        }
      }
    }
  }
}

// second stage: column c of [dgamma | dbeta] += sum over a slice of the per-block partial rows (grid.y slices)
__global__ __launch_bounds__(256) void ln_partial_reduce_kernel(const float* __restrict__ partial, int nblk, int D2,
                                                                float* dgamma, float* dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= D2) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = min(nblk, b0 + per);
  float acc = 0.f;
  for (int b = b0; b < b1; ++b) acc += partial[(int64_t)b * D2 + c];
  const int D = D2 / 2;
  atomicAdd(c < D ? dgamma + c : dbeta + (c - D), acc);
}

// the same for up to LN_MULTI layers in ONE launch (blockIdx.z = layer): the parameter-gradient sums are not on the critical
// path of backward, so the per-layer second stages are collected and run together (asr_ln_reduce_multi)
constexpr int LN_MULTI = 24;
struct LnMultiArgs {
  const float* ws[LN_MULTI]; float* dgamma[LN_MULTI]; float* dbeta[LN_MULTI]; int nblk[LN_MULTI];
  int D2;
};
// No atomics (64 slices x 1024 columns x 20 layers of same-address fp32 atomics were most of this launch) and a fixed summation
// order: a 1024-thread workgroup owns 256 columns of one layer, its 16 waves each add a sixteenth of the partial rows with 16-byte
// loads, the sixteen meet in LDS and wave 0 does the plain += .
__global__ __launch_bounds__(1024) void ln_partial_reduce_multi_kernel(LnMultiArgs a) {
  __shared__ float4 red[16][64];
  const int L = blockIdx.y, cg = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cg) * 4;                   // 4 consecutive columns of [dgamma | dbeta]
  const int nblk = a.nblk[L];
  const int per = (nblk + 15) / 16;
  const int b0 = sl * per, b1 = min(nblk, b0 + per);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < a.D2) {
    const float* partial = a.ws[L] + c;
#pragma unroll 8
    for (int b = b0; b < b1; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (int64_t)b * a.D2);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[sl][cg] = acc;
  __syncthreads();
  if (sl != 0 || c >= a.D2) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) {
    const float4 v = red[k][cg];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const int D = a.D2 / 2;
  float* dst = c < D ? a.dgamma[L] + c : a.dbeta[L] + (c - D);
  dst[0] += acc.x; dst[1] += acc.y; dst[2] += acc.z; dst[3] += acc.w;
}

}  // namespace

namespace {
template <typename T, int NCH>
void launch_fwd(T* y_z, const T* res, const float* gamma, const float* beta, const float* post, int period, const uint8_t* keep,
                T* out, float* mean, float* rstd, int M, int D, float eps, uint32_t thr, float inv, uint64_t seed,
                const uint64_t* seed_dev, hipStream_t s) {
  hipLaunchKernelGGL((add_ln_fwd_kernel<T, NCH>), dim3((M + 3) / 4), dim3(256), 0, s, y_z, res, gamma, beta, post, period, keep, out,
                     mean, rstd, M, D, eps, thr, inv, seed, seed_dev);
}
template <typename T, int NCH>
void launch_bwd(const T* dout, const T* z, const float* mean, const float* rstd, const float* gamma, const uint8_t* keep, T* d_res,
                T* d_y, float* dgamma, float* dbeta, float* ws, int64_t ws_floats, int M, int D, uint32_t thr, float inv, uint64_t seed,
                const uint64_t* seed_dev, hipStream_t s, bool reduce = true) {
  // with a workspace: 8 rows per block (fills the chip) and a two-stage column reduction; without: 32 rows + atomics
  int rpb = 8;
  int nblk = (M + rpb - 1) / rpb;
  if (!ws || ws_floats < (int64_t)nblk * 2 * D) { rpb = 32; nblk = (M + rpb - 1) / rpb; ws = nullptr; }
  hipLaunchKernelGGL((add_ln_bwd_kernel<T, NCH>), dim3(nblk), dim3(256), (size_t)3 * 2 * D * sizeof(float), s, dout, z,
                     mean, rstd, gamma, keep, d_res, d_y, dgamma, dbeta, ws, M, D, rpb, thr, inv, seed, seed_dev);
  if (ws && reduce) {
    const int slices = nblk >= 512 ? 64 : (nblk >= 64 ? 16 : 1);      // ~13 partial rows per thread
    hipLaunchKernelGGL(ln_partial_reduce_kernel, dim3((2 * D + 255) / 256, slices), dim3(256), 0, s, ws, nblk, 2 * D, dgamma, dbeta);
  }
}
// chunks of 16 bytes per lane needed to cover a row of D elements with one wave
template <typename T> int chunks_for(int D) { return (D + 64 * DT<T>::EPC - 1) / (64 * DT<T>::EPC); }
}  // namespace

#define ASR_LN_DISPATCH(T_, CALL)                                      \
  switch (chunks_for<T_>(D)) {                                         \
    case 1: CALL(T_, 1); break;                                        \
    case 2: CALL(T_, 2); break;                                        \
    case 3: case 4: CALL(T_, 4); break;                                \
    default: CALL(T_, 8); break;                                       \
  }

extern "C" int asr_add_ln_fwd(void* y_z, const void* residual, const float* gamma, const float* beta, const float* post_add,
                              int post_period, const uint8_t* row_keep, void* out, float* mean, float* rstd, int M, int D,
                              float eps, float p, uint64_t seed, const uint64_t* seed_dev, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(y_z && gamma && beta && out && mean && rstd && M >= 0 && D > 0 && p >= 0.f && p < 1.f);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  if (D % epc != 0 || D > 64 * 8 * epc) return ASR_EUNSUPPORTED;
  ASR_CHECK_ARG(aligned16(y_z) && aligned16(out) && (!residual || aligned16(residual)));
  ASR_CHECK_ARG(!post_add || post_period > 0);
  if (M == 0) return ASR_OK;
  const uint32_t thr = asr_drop_threshold(p);
  const float inv = 1.f / (1.f - p);
  AsrProfScope prof(ASR_OP_ADD_LN, s);
#define ASR_CALL_F(T_, N_) launch_fwd<T_, N_>((T_*)y_z, (const T_*)residual, gamma, beta, post_add, post_period, row_keep, (T_*)out, mean, rstd, M, D, eps, thr, inv, seed, seed_dev, s)
  if (dtype == ASR_F32) { ASR_LN_DISPATCH(float, ASR_CALL_F) } else { ASR_LN_DISPATCH(bf16_t, ASR_CALL_F) }
#undef ASR_CALL_F
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int64_t asr_add_ln_bwd_workspace(int M, int D) { return (int64_t)((M + 7) / 8) * 2 * D; }

extern "C" int asr_add_ln_bwd(const void* dout, const void* z, const float* mean, const float* rstd, const float* gamma,
                              const uint8_t* row_keep, void* d_res, void* d_y, float* dgamma, float* dbeta, float* workspace,
                              int64_t workspace_floats, int M, int D, float p, uint64_t seed, const uint64_t* seed_dev, int dtype,
                              hipStream_t s) {
  ASR_CHECK_ARG(dout && z && mean && rstd && gamma && d_res && dgamma && dbeta && M >= 0 && D > 0 && p >= 0.f && p < 1.f);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  const int epc = dtype == ASR_F32 ? 4 : 8;
  if (D % epc != 0 || D > 64 * 8 * epc) return ASR_EUNSUPPORTED;
  ASR_CHECK_ARG(aligned16(dout) && aligned16(z) && aligned16(d_res) && (!d_y || aligned16(d_y)));
  if (p > 0.f) ASR_CHECK_ARG(d_y && d_y != d_res);
  if (M == 0) return ASR_OK;
  const uint32_t thr = asr_drop_threshold(p);
  const float inv = 1.f / (1.f - p);
  AsrProfScope prof(ASR_OP_ADD_LN, s);
#define ASR_CALL_B(T_, N_) launch_bwd<T_, N_>((const T_*)dout, (const T_*)z, mean, rstd, gamma, row_keep, (T_*)d_res, (T_*)d_y, dgamma, dbeta, workspace, workspace_floats, M, D, thr, inv, seed, seed_dev, s)
  if (dtype == ASR_F32) { ASR_LN_DISPATCH(float, ASR_CALL_B) } else { ASR_LN_DISPATCH(bf16_t, ASR_CALL_B) }
#undef ASR_CALL_B
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

/* asr_add_ln_bwd without its second stage: the per-block column partials stay in `workspace` (required) until
 * asr_ln_reduce_multi adds them into dgamma / dbeta. */
extern "C" int asr_add_ln_bwd_partials(const void* dout, const void* z, const float* mean, const float* rstd, const float* gamma,
                                       const uint8_t* row_keep, void* d_res, void* d_y, float* workspace, int64_t workspace_floats,
                                       int M, int D, float p, uint64_t seed, const uint64_t* seed_dev, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(dout && z && mean && rstd && gamma && d_res && workspace && M >= 0 && D > 0 && p >= 0.f && p < 1.f);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  ASR_CHECK_ARG(workspace_floats >= asr_add_ln_bwd_workspace(M, D));
  const int epc = dtype == ASR_F32 ? 4 : 8;
  if (D % epc != 0 || D > 64 * 8 * epc) return ASR_EUNSUPPORTED;
  ASR_CHECK_ARG(aligned16(dout) && aligned16(z) && aligned16(d_res) && (!d_y || aligned16(d_y)));
  if (p > 0.f) ASR_CHECK_ARG(d_y && d_y != d_res);
  if (M == 0) return ASR_OK;
  const uint32_t thr = asr_drop_threshold(p);
  const float inv = 1.f / (1.f - p);
  float* dgamma = nullptr;
  float* dbeta = nullptr;
  AsrProfScope prof(ASR_OP_ADD_LN, s);
#define ASR_CALL_P(T_, N_) launch_bwd<T_, N_>((const T_*)dout, (const T_*)z, mean, rstd, gamma, row_keep, (T_*)d_res, (T_*)d_y, dgamma, dbeta, workspace, workspace_floats, M, D, thr, inv, seed, seed_dev, s, false)
  if (dtype == ASR_F32) { ASR_LN_DISPATCH(float, ASR_CALL_P) } else { ASR_LN_DISPATCH(bf16_t, ASR_CALL_P) }
#undef ASR_CALL_P
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_ln_reduce_multi(const float* const* workspaces, const int* rows, float* const* dgamma, float* const* dbeta, int n,
                                   int D, hipStream_t s) {
  ASR_CHECK_ARG(n >= 0 && D > 0 && D % 2 == 0 && (n == 0 || (workspaces && rows && dgamma && dbeta)));     // (16-byte partial rows)
  for (int i0 = 0; i0 < n; i0 += LN_MULTI) {
    LnMultiArgs a{};
    const int cnt = n - i0 < LN_MULTI ? n - i0 : LN_MULTI;
    for (int i = 0; i < cnt; ++i) {
      ASR_CHECK_ARG(workspaces[i0 + i] && dgamma[i0 + i] && dbeta[i0 + i] && rows[i0 + i] >= 0);
      a.ws[i] = workspaces[i0 + i]; a.dgamma[i] = dgamma[i0 + i]; a.dbeta[i] = dbeta[i0 + i];
      a.nblk[i] = (rows[i0 + i] + 7) / 8;                   // the first stage's 8 rows per block
    }
    a.D2 = 2 * D;
    hipLaunchKernelGGL(ln_partial_reduce_multi_kernel, dim3((2 * D + 255) / 256, cnt), dim3(1024), 0, s, a);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
