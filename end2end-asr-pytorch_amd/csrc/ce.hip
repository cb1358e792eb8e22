// Label-smoothed cross entropy + argmax + num_correct (reference: utils/metrics.py:78-132; topk at
// models/asr/transformer.py:80).  One 256-thread block per token row; wave-first reductions.
// HBM-bound: forward reads M*V*4 bytes once (second pass is L2-resident), backward reads M*V*4 and writes
// M*ldd*sizeof(T).
#include "common.h"

namespace {

struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {     // greater value, lowest index on ties
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}

// One wave per row, CE_ROWS rows per block: the three running sums (loss, #non-PAD rows, #correct) are reduced over the block's
// rows in LDS first -- one set of same-address atomics per block instead of per row (they were most of this kernel's time).
constexpr int CE_ROWS = 8;

__global__ __launch_bounds__(64 * CE_ROWS) void ce_fwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                              const int64_t* __restrict__ gold, int M, int V, float eps, int pad_id,
                                                              float* __restrict__ row_lse, int64_t* __restrict__ argmax, float* sums,
                                                              float* __restrict__ partials) {
  __shared__ float s_part[CE_ROWS][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * CE_ROWS + wave;
  float part[3] = {0.f, 0.f, 0.f};
  if (row < M) {
    const float* l = logits + (int64_t)row * ld;
    const bool vec = ((((uintptr_t)l) & 15) == 0);
    const int V4 = vec ? (V & ~3) : 0;
    MaxIdx mi{-INFINITY, 0x7fffffff};
    float sum = 0.f;
#pragma unroll 4
    for (int v = lane * 4; v < V4; v += 256) {         // four 16-byte loads in flight per lane: the row is one memory round trip deep, not 17
      const float4 x = *reinterpret_cast<const float4*>(l + v);
      sum += (x.x + x.y) + (x.z + x.w);
      if (x.x > mi.v) { mi.v = x.x; mi.i = v; }
      if (x.y > mi.v) { mi.v = x.y; mi.i = v + 1; }
      if (x.z > mi.v) { mi.v = x.z; mi.i = v + 2; }
      if (x.w > mi.v) { mi.v = x.w; mi.i = v + 3; }
    }
    for (int v = V4 + lane; v < V; v += 64) {
      const float x = l[v];
      sum += x;
      if (x > mi.v) { mi.v = x; mi.i = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      MaxIdx t{__shfl_xor(mi.v, o, 64), __shfl_xor(mi.i, o, 64)};
      mi = better(mi, t);
    }
    sum = wave_sum(sum);
    float e = 0.f;
#pragma unroll 4
    for (int v = lane * 4; v < V4; v += 256) {
      const float4 x = *reinterpret_cast<const float4*>(l + v);
      e += (expf(x.x - mi.v) + expf(x.y - mi.v)) + (expf(x.z - mi.v) + expf(x.w - mi.v));
    }
    for (int v = V4 + lane; v < V; v += 64) e += expf(l[v] - mi.v);
    e = wave_sum(e);
    if (lane == 0) {
      const float lse = mi.v + logf(e);
      row_lse[row] = lse;
      argmax[row] = mi.i == 0x7fffffff ? 0 : mi.i;        // (a row without a finite value: index 0, like argmax_rows_kernel)
      const int64_t g = gold[row];
      if (g != pad_id) {
        const float lpg = l[g] - lse;
        float loss;
        if (eps > 0.f) {
          const float sum_lp = sum - (float)V * lse;
          loss = -((1.f - eps) * lpg + (eps / (float)V) * (sum_lp - lpg));
        } else {
          loss = -lpg;
        }
        part[0] = loss;
        part[1] = 1.f;
        part[2] = mi.i == g ? 1.f : 0.f;
      }
    }
  }
  if (lane == 0) { s_part[wave][0] = part[0]; s_part[wave][1] = part[1]; s_part[wave][2] = part[2]; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < CE_ROWS; ++w) t += s_part[w][threadIdx.x];
    // partials (asr_ce_fwd_partials): the block's three sums go to their own slot and asr_ce_finish adds the slots in a fixed order -- the
    // statistics (and the loss) are then the same bits run to run, and nobody has to zero a destination first
    if (partials) partials[(int64_t)blockIdx.x * 3 + threadIdx.x] = t;
    else if (t != 0.f) atomicAdd(sums + threadIdx.x, t);
  }
}

// sums[k] = the blocks' partial sums added in a fixed order (thread t takes blocks t, t + 256, ...; then a tree over the threads);
// loss = sums[0] / (den ? den[0] : sums[1])
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ partials, int nblocks, const float* __restrict__ den,
                                                        float* __restrict__ sums, float* __restrict__ loss) {
  __shared__ float red[3][256];
  const int tid = threadIdx.x;
  float a[3] = {0.f, 0.f, 0.f};
  for (int b = tid; b < nblocks; b += 256)
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] += partials[(int64_t)b * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) red[k][tid] = a[k];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o)
#pragma unroll
      for (int k = 0; k < 3; ++k) red[k][tid] += red[k][tid + o];
    __syncthreads();
  }
  if (tid < 3) sums[tid] = red[tid][0];
  if (tid == 0) loss[0] = red[0][0] / (den ? den[0] : red[1][0]);
}

__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                          int64_t* __restrict__ out) {
  __shared__ float s_v[4]; __shared__ int s_i[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* l = logits + (int64_t)row * ld;
  MaxIdx mi{-INFINITY, 0x7fffffff};
  const int V4 = ((((uintptr_t)l) & 15) == 0) ? (V & ~3) : 0;
#pragma unroll 4
  for (int v = tid * 4; v < V4; v += 1024) {           // 16-byte loads, four in flight (strictly-greater keeps the lowest index per lane)
    const float4 x = *reinterpret_cast<const float4*>(l + v);
    if (x.x > mi.v) { mi.v = x.x; mi.i = v; }
    if (x.y > mi.v) { mi.v = x.y; mi.i = v + 1; }
    if (x.z > mi.v) { mi.v = x.z; mi.i = v + 2; }
    if (x.w > mi.v) { mi.v = x.w; mi.i = v + 3; }
  }
  for (int v = V4 + tid; v < V; v += 256) {
    const float x = l[v];
    if (x > mi.v) { mi.v = x; mi.i = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxIdx t{__shfl_xor(mi.v, o, 64), __shfl_xor(mi.i, o, 64)};
    mi = better(mi, t);
  }
  if (lane == 0) { s_v[wave] = mi.v; s_i[wave] = mi.i; }
  __syncthreads();
  if (tid == 0) {
    MaxIdx m{s_v[0], s_i[0]};
#pragma unroll
    for (int w = 1; w < 4; ++w) m = better(m, MaxIdx{s_v[w], s_i[w]});
    out[row] = m.i == 0x7fffffff ? 0 : m.i;
  }
}

// Beam-search scoring (reference: transformer.py:446-449, F.log_softmax + torch.topk per hypothesis): the k best log-probabilities
// of a logits row and their indices, best first (equal values: lowest index first).  One workgroup per row: row max and
// sum of exponentials, then k block-wide arg-max passes over the row (L2 resident), each excluding the indices already taken.
constexpr int TOPK_MAX = 16;
__global__ __launch_bounds__(256) void logsoftmax_topk_kernel(const float* __restrict__ logits, int64_t ld, int V, int k,
                                                              float* __restrict__ vals, int64_t* __restrict__ idx) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  __shared__ float s_red[2][4];
  __shared__ int taken[TOPK_MAX];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* l = logits + (int64_t)row * ld;
  float mx = -INFINITY;
  for (int v = tid; v < V; v += 256) mx = fmaxf(mx, l[v]);
  mx = wave_max(mx);
  if (lane == 0) s_red[0][wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3]));
  float se = 0.f;
  for (int v = tid; v < V; v += 256) se += expf(l[v] - mx);
  se = wave_sum(se);
  if (lane == 0) s_red[1][wave] = se;
  __syncthreads();
  const float lse = mx + logf(s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]);
  for (int j = 0; j < k; ++j) {
    MaxIdx mi{-INFINITY, 0x7fffffff};
    for (int v = tid; v < V; v += 256) {
      bool used = false;
      for (int t = 0; t < j; ++t) used |= taken[t] == v;
      const float x = l[v];
      if (!used && (x > mi.v || (x == mi.v && v < mi.i))) { mi.v = x; mi.i = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      MaxIdx t{__shfl_xor(mi.v, o, 64), __shfl_xor(mi.i, o, 64)};
      mi = better(mi, t);
    }
    if (lane == 0) { s_v[wave] = mi.v; s_i[wave] = mi.i; }
    __syncthreads();
    if (tid == 0) {
      MaxIdx m{s_v[0], s_i[0]};
#pragma unroll
      for (int w = 1; w < 4; ++w) m = better(m, MaxIdx{s_v[w], s_i[w]});
      taken[j] = m.i;
      vals[(int64_t)row * k + j] = m.v - lse;
      idx[(int64_t)row * k + j] = m.i == 0x7fffffff ? 0 : m.i;
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int64_t ld,
                                                     const int64_t* __restrict__ gold, const float* __restrict__ row_lse,
                                                     int V, float eps, int pad_id, const float* __restrict__ grad_out,
                                                     const float* __restrict__ count, T* __restrict__ dl, int64_t ldd) {
  const int row = blockIdx.x;
  const int64_t g = gold[row];
  T* d = dl + (int64_t)row * ldd;
  if (g == pad_id) {
    for (int v = threadIdx.x; v < ldd; v += 256) DT<T>::st(d + v, 0.f);
    return;
  }
  const float coef = (*grad_out) / (*count);
  const float* l = logits + (int64_t)row * ld;
  const float lse = row_lse[row];
  const float q_other = eps > 0.f ? eps / (float)V : 0.f;
  const float q_gold = eps > 0.f ? 1.f - eps : 1.f;
  const float sum_q = q_gold + (float)(V - 1) * q_other;
  for (int v = threadIdx.x; v < ldd; v += 256) {
    float o = 0.f;
    if (v < V) o = coef * (expf(l[v] - lse) * sum_q - (v == g ? q_gold : q_other));
    DT<T>::st(d + v, o);
  }
}

}  // namespace

extern "C" int asr_ce_fwd(const float* logits, int64_t ld, const int64_t* gold, int M, int V, float smoothing, int pad_id,
                          float* row_lse, int64_t* argmax, float* sums, hipStream_t s) {
  ASR_CHECK_ARG(logits && gold && row_lse && argmax && sums && M >= 0 && V > 0 && ld >= V && smoothing >= 0.f);
  if (M == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_CE, s);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((M + CE_ROWS - 1) / CE_ROWS), dim3(64 * CE_ROWS), 0, s, logits, ld, gold, M, V, smoothing, pad_id,
                     row_lse, argmax, sums, (float*)nullptr);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_ce_partial_blocks(int M) { return M > 0 ? (M + CE_ROWS - 1) / CE_ROWS : 0; }

extern "C" int asr_ce_fwd_partials(const float* logits, int64_t ld, const int64_t* gold, int M, int V, float smoothing, int pad_id,
                                   float* row_lse, int64_t* argmax, float* partials, hipStream_t s) {
  ASR_CHECK_ARG(logits && gold && row_lse && argmax && partials && M >= 0 && V > 0 && ld >= V && smoothing >= 0.f);
  if (M == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_CE, s);
  hipLaunchKernelGGL(ce_fwd_kernel, dim3((M + CE_ROWS - 1) / CE_ROWS), dim3(64 * CE_ROWS), 0, s, logits, ld, gold, M, V, smoothing, pad_id,
                     row_lse, argmax, (float*)nullptr, partials);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_ce_finish(const float* partials, int nblocks, const float* den, float* sums, float* loss, hipStream_t s) {
  ASR_CHECK_ARG(sums && loss && nblocks >= 0 && (partials || nblocks == 0));
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, s, partials, nblocks, den, sums, loss);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_ce_bwd(const float* logits, int64_t ld, const int64_t* gold, const float* row_lse, int M, int V,
                          float smoothing, int pad_id, const float* grad_out, const float* count, void* dlogits, int64_t ldd,
                          int out_dtype, hipStream_t s) {
  ASR_CHECK_ARG(logits && gold && row_lse && grad_out && count && dlogits && M >= 0 && V > 0 && ld >= V && ldd >= V);
  if (M == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_CE, s);
  if (out_dtype == ASR_F32)
    hipLaunchKernelGGL((ce_bwd_kernel<float>), dim3(M), dim3(256), 0, s, logits, ld, gold, row_lse, V, smoothing, pad_id,
                       grad_out, count, (float*)dlogits, ldd);
  else if (out_dtype == ASR_BF16)
    hipLaunchKernelGGL((ce_bwd_kernel<bf16_t>), dim3(M), dim3(256), 0, s, logits, ld, gold, row_lse, V, smoothing, pad_id,
                       grad_out, count, (bf16_t*)dlogits, ldd);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_argmax_rows(const float* logits, int64_t ld, int M, int V, int64_t* out, hipStream_t s) {
  ASR_CHECK_ARG(logits && out && M >= 0 && V > 0 && ld >= V);
  if (M == 0) return ASR_OK;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(M), dim3(256), 0, s, logits, ld, V, out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_logsoftmax_topk(const float* logits, int64_t ld, int M, int V, int k, float* vals, int64_t* idx, hipStream_t s) {
  ASR_CHECK_ARG(logits && vals && idx && M >= 0 && V > 0 && ld >= V && k > 0 && k <= V);
  if (k > TOPK_MAX) return ASR_EUNSUPPORTED;
  if (M == 0) return ASR_OK;
  hipLaunchKernelGGL(logsoftmax_topk_kernel, dim3(M), dim3(256), 0, s, logits, ld, V, k, vals, idx);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
