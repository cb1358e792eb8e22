// HBM-bound helpers: transposes, weight shadows, bias-gradient column sums, embedding, decoder framing,
// fused Adam, gradient-norm.  All are coalesced along the contiguous axis; reductions are wave-first.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ transpose
// 64x64 tile through LDS (+1 pad).  in (rows, cols) -> out (cols, rows).  TI -> TO conversion on the way.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, int64_t ld_in, TO* __restrict__ out,
                                                        int64_t ld_out, TO* __restrict__ out_same, int64_t ld_same, int rows, int cols,
                                                        float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  __shared__ float csum[4][64];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
  for (int i = ty; i < 64; i += 4) {
    int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = DT<TI>::ld(in + (int64_t)r * ld_in + c);
      if (out_same) DT<TO>::st(out_same + (int64_t)r * ld_same + c, v);
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (colsum) {      // bias gradient for free: column sums of the tile we already hold (rows beyond `rows` are zero)
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += tile[ty * 16 + i][tx];
    csum[ty][tx] = a;
    __syncthreads();
    if (ty == 0 && c0 + tx < cols) atomicAdd(colsum + c0 + tx, csum[0][tx] + csum[1][tx] + csum[2][tx] + csum[3][tx]);
  }
  if (out) {
    for (int i = ty; i < 64; i += 4) {
      int c = c0 + i, r = r0 + tx;
      if (r < rows && c < cols) DT<TO>::st(out + (int64_t)c * ld_out + r, tile[tx][i]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ colsum
// out[n] += sum_m X[m, n].  Block = 256 threads = 64 columns x 4 row-lanes; grid.y strides over row slabs.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int64_t ld, int M, int N, float* out,
                                                     int rows_per_block) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int m_beg = blockIdx.y * rows_per_block, m_end = min(M, m_beg + rows_per_block);
  float s = 0.f;
  if (n < N)
    for (int m = m_beg + ty; m < m_end; m += 4) s += DT<T>::ld(X + (int64_t)m * ld + n);
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) atomicAdd(out + n, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// ------------------------------------------------------------------------------------------------ embedding
template <typename T>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ tok, const float* __restrict__ table,
                                                        const float* __restrict__ pe, T* __restrict__ out, int T_len,
                                                        int D, float scale, uint32_t thr, float inv_keep, uint64_t seed0,
                                                        const uint64_t* __restrict__ seed_dev) {
  const uint64_t seed = asr_mix_seed(seed0, seed_dev);
  const int row = blockIdx.x;                 // b*T + t
  const int t = row % T_len;
  const int64_t id = tok[row];
  const float* e = table + id * (int64_t)D;
  const float* pr = pe + (int64_t)t * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float v = e[c] * scale + pr[c];
    if (thr) v = asr_keep(seed, (uint64_t)row * D + c, thr) ? v * inv_keep : 0.f;
    DT<T>::st(out + (int64_t)row * D + c, v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ tok, const T* __restrict__ dout,
                                                        float* dtable, int D, float scale, uint32_t thr, float inv_keep,
                                                        uint64_t seed0, const uint64_t* __restrict__ seed_dev, int pad_id) {
  const uint64_t seed = asr_mix_seed(seed0, seed_dev);
  const int row = blockIdx.x;
  const int64_t id = tok[row];
  if (id == pad_id) return;                   // nn.Embedding(padding_idx=PAD): no gradient for the PAD row
  float* d = dtable + id * (int64_t)D;
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    float g = DT<T>::ld(dout + (int64_t)row * D + c) * scale;
    if (thr) g = asr_keep(seed, (uint64_t)row * D + c, thr) ? g * inv_keep : 0.f;
    atomicAdd(d + c, g);
  }
}

// ------------------------------------------------------------------------------------------------ preprocess
// one wave per target row; stable compaction of non-PAD tokens with ballot prefix counts.
__global__ __launch_bounds__(64) void preprocess_kernel(const int64_t* __restrict__ tgt, int L, int Td, int64_t* seq_in,
                                                        int64_t* seq_out, uint8_t* key_pad, uint8_t* row_keep,
                                                        int32_t* overflow) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int64_t* y = tgt + (int64_t)b * L;
  int64_t* si = seq_in + (int64_t)b * Td;
  int64_t* so = seq_out + (int64_t)b * Td;
  for (int t = lane; t < Td; t += 64) { si[t] = 2; so[t] = 0; }          // EOS / PAD fill (transformer.py:263-264)
  __syncthreads();
  int n = 0;
  for (int base = 0; base < L; base += 64) {
    const int i = base + lane;
    const int64_t v = i < L ? y[i] : 0;
    const bool keep = v != 0;
    const unsigned long long m = __ballot(keep);
    const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
    if (keep) {
      if (pos + 1 < Td) si[pos + 1] = v;
      if (pos < Td) so[pos] = v;
    }
    n += __popcll(m);
  }
  if (lane == 0) {
    si[0] = 1;                                                             // SOS
    if (n < Td) so[n] = 2;                                                 // EOS
    if (n + 1 > Td) *overflow = 1;
  }
  __syncthreads();
  for (int t = lane; t < Td; t += 64) {
    const bool is_eos = si[t] == 2;
    key_pad[(int64_t)b * Td + t] = is_eos ? 1 : 0;
    row_keep[(int64_t)b * Td + t] = is_eos ? 0 : 1;
  }
}

// row_keep[b, t] = t < len[b]   (get_non_pad_mask with input_lengths, common_layers.py:33-38)
__global__ __launch_bounds__(256) void length_mask_kernel(const int32_t* __restrict__ len, int T, uint8_t* __restrict__ out) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t < T) out[(int64_t)b * T + t] = t < len[b] ? 1 : 0;
}
__global__ void ratio_kernel(const float* num, const float* den, float* out) { *out = *num / *den; }

// ------------------------------------------------------------------------------------------------ Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n4, int64_t n, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2_sqrt,
                                                   const float* __restrict__ gscale) {
  const float gs = gscale ? *gscale : 1.f;
  const float step = lr / bc1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* Mo = &mm.x; float* Vo = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = G[j] * gs;
      Mo[j] = b1 * Mo[j] + (1.f - b1) * gj;
      Vo[j] = b2 * Vo[j] + (1.f - b2) * gj * gj;
      const float denom = sqrtf(Vo[j]) / bc2_sqrt + eps;
      P[j] -= step * (Mo[j] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // scalar tail
  const int64_t tail = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tail < n) {
    const float gj = g[tail] * gs;
    const float mo = b1 * m[tail] + (1.f - b1) * gj;
    const float vo = b2 * v[tail] + (1.f - b2) * gj * gj;
    m[tail] = mo; v[tail] = vo;
    p[tail] -= step * (mo / (sqrtf(vo) / bc2_sqrt + eps));
  }
}

// Same update with the step count (and therefore Noam's learning rate and Adam's bias corrections) read from DEVICE
// memory, so that a captured hipGraph replays a correct optimiser step (reference: utils/optimizer.py:15-32).
__global__ __launch_bounds__(256) void adam_noam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n4, int64_t n,
                                                        uint64_t* __restrict__ state, float b1, float b2, float eps,
                                                        float factor_ms, float warmup, float min_lr,
                                                        const float* __restrict__ gscale, float* __restrict__ lr_out,
                                                        const float* __restrict__ guard, bf16_t* __restrict__ shadow) {
  const float gs = gscale ? *gscale : 1.f;
  const double t = (double)state[1];
  const double w15 = pow((double)warmup, -1.5);
  const double sched = fmin(pow(t, -0.5), t * w15);
  const float lr = fmaxf(min_lr, (float)((double)factor_ms * sched));
  // a non-finite loss (or clipping coefficient) skips the whole update: the reference's `if loss == inf: continue`
  // (trainer/asr/trainer.py:102-104 -- the batch is dropped BEFORE opt.step(), so neither the weights nor the step count move).
  // state[2] tells the next asr_step_advance that this step number was not consumed; the rate is reported either way.
  const bool skip = !isfinite(gs) || (guard && !isfinite(*guard));
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (lr_out) *lr_out = lr;
    state[2] = skip ? 1u : 0u;
  }
  if (skip) return;
  const float bc1 = (float)(1.0 - pow((double)b1, t));
  const float bc2_sqrt = sqrtf((float)(1.0 - pow((double)b2, t)));
  const float step = lr / bc1;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* P = &pp.x; float* G = &gg.x; float* Mo = &mm.x; float* Vo = &vv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = G[j] * gs;
      Mo[j] = b1 * Mo[j] + (1.f - b1) * gj;
      Vo[j] = b2 * Vo[j] + (1.f - b2) * gj * gj;
      P[j] -= step * (Mo[j] / (sqrtf(Vo[j]) / bc2_sqrt + eps));
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    // the bf16 shadow every GEMM of the next step reads, written here instead of by a cast pass over the masters
    if (shadow) reinterpret_cast<uint2*>(shadow)[i] = make_uint2((uint32_t)f32_to_bf16(pp.x) | ((uint32_t)f32_to_bf16(pp.y) << 16),
                                                                 (uint32_t)f32_to_bf16(pp.z) | ((uint32_t)f32_to_bf16(pp.w) << 16));
  }
  const int64_t tail = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tail < n) {
    const float gj = g[tail] * gs;
    const float mo = b1 * m[tail] + (1.f - b1) * gj;
    const float vo = b2 * v[tail] + (1.f - b2) * gj * gj;
    m[tail] = mo; v[tail] = vo;
    const float pn = p[tail] - step * (mo / (sqrtf(vo) / bc2_sqrt + eps));
    p[tail] = pn;
    if (shadow) shadow[tail] = f32_to_bf16(pn);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void cast_flat_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i * 4 < n; i += stride) {
    if (i * 4 + 4 <= n) {
      const float4 v = reinterpret_cast<const float4*>(src)[i];
      DT<T>::st(dst + i * 4, v.x); DT<T>::st(dst + i * 4 + 1, v.y); DT<T>::st(dst + i * 4 + 2, v.z); DT<T>::st(dst + i * 4 + 3, v.w);
    } else {
      for (int64_t e = i * 4; e < n; ++e) DT<T>::st(dst + e, src[e]);
    }
  }
}
// state = {dropout seed counter, optimiser step, 1 when the previous optimiser launch skipped its update, unused}
__global__ void step_advance_kernel(uint64_t* state) { state[0] += 1; state[1] += 1 - (state[2] ? 1 : 0); state[2] = 0; }

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* acc) {
  __shared__ float red[4];
  float s = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}
__global__ void clip_coef_kernel(const float* sumsq, float max_norm, const float* denom, float* coef) {
  // gradients in the buffer are `denom` times the true ones (un-normalised loss sum, see asr_grad_coef)
  const float s = denom ? 1.f / fmaxf(*denom, 1.f) : 1.f;
  float c = 1.f;
  if (sumsq) {
    const float nrm = sqrtf(*sumsq) * s;
    c = max_norm / (nrm + 1e-6f);               // torch.nn.utils.clip_grad_norm_
    c = c < 1.f ? c : 1.f;
  }
  *coef = s * c;
}

}  // namespace

extern "C" int asr_transpose(const void* in, int64_t ld_in, void* out, int64_t ld_out, int rows, int cols, float* colsum,
                             int dtype, hipStream_t s) {
  ASR_CHECK_ARG(in && out && rows >= 0 && cols >= 0);
  if (rows == 0 || cols == 0) return ASR_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  AsrProfScope prof(ASR_OP_LAYOUT, s);
  if (dtype == ASR_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float*)in, ld_in, (float*)out, ld_out,
                       (float*)nullptr, (int64_t)0, rows, cols, colsum);
  else if (dtype == ASR_BF16)
    hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)in, ld_in, (bf16_t*)out,
                       ld_out, (bf16_t*)nullptr, (int64_t)0, rows, cols, colsum);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_cast_weight(const float* src, int64_t ld_src, void* dst, int64_t ld_dst, void* dst_t, int64_t ld_dst_t,
                               int rows, int cols, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(src && rows >= 0 && cols >= 0 && (dst || dst_t) && ld_src >= cols);
  ASR_CHECK_ARG((!dst || ld_dst >= cols) && (!dst_t || ld_dst_t >= rows));
  if (rows == 0 || cols == 0) return ASR_OK;
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  AsrProfScope prof(ASR_OP_LAYOUT, s);
  if (dtype == ASR_F32)
    hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, s, src, ld_src, (float*)dst_t, ld_dst_t,
                       (float*)dst, ld_dst, rows, cols, (float*)nullptr);
  else if (dtype == ASR_BF16)
    hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, s, src, ld_src, (bf16_t*)dst_t, ld_dst_t,
                       (bf16_t*)dst, ld_dst, rows, cols, (float*)nullptr);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_colsum_acc(const void* X, int64_t ld, int M, int N, float* out, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(X && out && M >= 0 && N >= 0);
  if (M == 0 || N == 0) return ASR_OK;
  const int rpb = 256;
  dim3 grid((N + 63) / 64, (M + rpb - 1) / rpb);
  if (dtype == ASR_F32) hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, s, (const float*)X, ld, M, N, out, rpb);
  else if (dtype == ASR_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)X, ld, M, N, out, rpb);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_embed_fwd(const int64_t* tok, const float* table, const float* pe, void* out, int B, int T, int D,
                             float scale, float p, uint64_t seed, const uint64_t* seed_dev, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(tok && table && pe && out && B >= 0 && T > 0 && D > 0 && p >= 0.f && p < 1.f);
  if (B == 0) return ASR_OK;
  const uint32_t thr = asr_drop_threshold(p);
  const float inv = 1.f / (1.f - p);
  if (dtype == ASR_F32) hipLaunchKernelGGL((embed_fwd_kernel<float>), dim3(B * T), dim3(256), 0, s, tok, table, pe, (float*)out, T, D, scale, thr, inv, seed, seed_dev);
  else if (dtype == ASR_BF16) hipLaunchKernelGGL((embed_fwd_kernel<bf16_t>), dim3(B * T), dim3(256), 0, s, tok, table, pe, (bf16_t*)out, T, D, scale, thr, inv, seed, seed_dev);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
// ---- incremental decoding with a device-side position (one captured hipGraph serves every step) -----------------------
// state[0] = position t of the token being fed.  pe_cur = pe[t]; key_len[b] = t + 1 (the self-attention of this step sees the
// cache rows 0..t); `advance` != 0: afterwards state[0] = t + 1 (issued as the LAST launch of a step).
__global__ __launch_bounds__(256) void decode_prepare_kernel(const float* __restrict__ pe, int D, float* __restrict__ pe_cur,
                                                             int32_t* __restrict__ key_len, int B, int64_t* state, int advance) {
  const int64_t t = state[0];
  if (advance) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[0] = t + 1;
    return;
  }
  for (int i = blockIdx.x * 256 + threadIdx.x; i < D; i += gridDim.x * 256) pe_cur[i] = pe[t * D + i];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B; i += gridDim.x * 256) key_len[i] = (int32_t)(t + 1);
}
// rows (B, ncols) with row stride `src_ld` -> cache[b][t][0..ncols) for two caches (this position's key and value rows)
template <typename T>
__global__ __launch_bounds__(256) void kv_append_kernel(const T* __restrict__ k_src, const T* __restrict__ v_src, int64_t src_ld,
                                                        T* __restrict__ k_cache, T* __restrict__ v_cache, int ncols, int max_len,
                                                        const int64_t* __restrict__ state) {
  const int64_t t = state[0];
  if (t >= max_len) return;
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < ncols; c += 256) {
    k_cache[((int64_t)b * max_len + t) * ncols + c] = k_src[(int64_t)b * src_ld + c];
    v_cache[((int64_t)b * max_len + t) * ncols + c] = v_src[(int64_t)b * src_ld + c];
  }
}

extern "C" int asr_decode_prepare(const float* pe, int D, float* pe_cur, int32_t* key_len, int B, int64_t* state, int advance,
                                  hipStream_t s) {
  ASR_CHECK_ARG(state && (advance || (pe && pe_cur && key_len && D > 0 && B >= 0)));
  hipLaunchKernelGGL(decode_prepare_kernel, dim3(advance ? 1 : 4), dim3(256), 0, s, pe, D, pe_cur, key_len, B, state, advance);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
extern "C" int asr_kv_append(const void* k_src, const void* v_src, int64_t src_ld, void* k_cache, void* v_cache, int B, int ncols,
                             int max_len, const int64_t* state, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(k_src && v_src && k_cache && v_cache && state && B >= 0 && ncols > 0 && max_len > 0);
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  if (B == 0) return ASR_OK;
  if (dtype == ASR_F32) hipLaunchKernelGGL((kv_append_kernel<float>), dim3(B), dim3(256), 0, s, (const float*)k_src, (const float*)v_src, src_ld, (float*)k_cache, (float*)v_cache, ncols, max_len, state);
  else hipLaunchKernelGGL((kv_append_kernel<bf16_t>), dim3(B), dim3(256), 0, s, (const bf16_t*)k_src, (const bf16_t*)v_src, src_ld, (bf16_t*)k_cache, (bf16_t*)v_cache, ncols, max_len, state);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_embed_bwd(const int64_t* tok, const void* dout, float* dtable, int B, int T, int D, float scale,
                             float p, uint64_t seed, const uint64_t* seed_dev, int pad_id, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(tok && dout && dtable && B >= 0 && T > 0 && D > 0 && p >= 0.f && p < 1.f);
  if (B == 0) return ASR_OK;
  const uint32_t thr = asr_drop_threshold(p);
  const float inv = 1.f / (1.f - p);
  if (dtype == ASR_F32) hipLaunchKernelGGL((embed_bwd_kernel<float>), dim3(B * T), dim3(256), 0, s, tok, (const float*)dout, dtable, D, scale, thr, inv, seed, seed_dev, pad_id);
  else if (dtype == ASR_BF16) hipLaunchKernelGGL((embed_bwd_kernel<bf16_t>), dim3(B * T), dim3(256), 0, s, tok, (const bf16_t*)dout, dtable, D, scale, thr, inv, seed, seed_dev, pad_id);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_decoder_preprocess(const int64_t* tgt, int B, int L, int Td, int64_t* seq_in, int64_t* seq_out,
                                      uint8_t* key_pad, uint8_t* row_keep, int32_t* overflow, hipStream_t s) {
  ASR_CHECK_ARG(tgt && seq_in && seq_out && key_pad && row_keep && overflow && B >= 0 && L >= 0 && Td >= 1);
  if (B == 0) return ASR_OK;
  hipLaunchKernelGGL(preprocess_kernel, dim3(B), dim3(64), 0, s, tgt, L, Td, seq_in, seq_out, key_pad, row_keep, overflow);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_length_mask(const int32_t* lengths, int B, int T, uint8_t* row_keep, hipStream_t s) {
  ASR_CHECK_ARG(lengths && row_keep && B >= 0 && T >= 0);
  if (B == 0 || T == 0) return ASR_OK;
  hipLaunchKernelGGL(length_mask_kernel, dim3((T + 255) / 256, B), dim3(256), 0, s, lengths, T, row_keep);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
extern "C" int asr_ratio(const float* num, const float* den, float* out, hipStream_t s) {
  ASR_CHECK_ARG(num && den && out);
  hipLaunchKernelGGL(ratio_kernel, dim3(1), dim3(1), 0, s, num, den, out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float bc1, float bc2, const float* gscale, hipStream_t s) {
  ASR_CHECK_ARG(p && g && m && v && n >= 0 && bc1 > 0.f && bc2 > 0.f);
  if (n == 0) return ASR_OK;
  ASR_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v));
  const int64_t n4 = n / 4;
  int64_t blocks = ceil_div64(n4 > 0 ? n4 : 1, 256);
  if (blocks > 4096) blocks = 4096;
  AsrProfScope prof(ASR_OP_ADAM, s);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n4, n, lr, beta1, beta2, eps, bc1,
                     sqrtf(bc2), gscale);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_sumsq_acc(const float* g, int64_t n, float* acc, hipStream_t s) {
  ASR_CHECK_ARG(g && acc && n >= 0);
  if (n == 0) return ASR_OK;
  int64_t blocks = ceil_div64(n, 256 * 8);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, n, acc);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
extern "C" int asr_clip_coef(const float* sumsq, float max_norm, float* coef, hipStream_t s) {
  ASR_CHECK_ARG(sumsq && coef);
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, sumsq, max_norm, (const float*)nullptr, coef);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
extern "C" int asr_grad_coef(const float* sumsq, float max_norm, const float* denom, float* coef, hipStream_t s) {
  ASR_CHECK_ARG(coef && (sumsq || denom));
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, s, sumsq, max_norm, denom, coef);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_step_advance(uint64_t* state, hipStream_t s) {
  ASR_CHECK_ARG(state);
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, state);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_adam_noam_step(float* p, const float* g, float* m, float* v, int64_t n, uint64_t* state, float beta1,
                                  float beta2, float eps, float factor_ms, float warmup, float min_lr, const float* gscale,
                                  float* lr_out, const float* guard, void* shadow_bf16, hipStream_t s) {
  ASR_CHECK_ARG(p && g && m && v && state && n >= 0 && warmup > 0.f);
  if (n == 0) return ASR_OK;
  ASR_CHECK_ARG(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && (!shadow_bf16 || (((uintptr_t)shadow_bf16) & 7) == 0));
  const int64_t n4 = n / 4;
  int64_t blocks = ceil_div64(n4 > 0 ? n4 : 1, 256);
  if (blocks > 4096) blocks = 4096;
  AsrProfScope prof(ASR_OP_ADAM, s);
  hipLaunchKernelGGL(adam_noam_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n4, n, state, beta1, beta2, eps,
                     factor_ms, warmup, min_lr, gscale, lr_out, guard, static_cast<bf16_t*>(shadow_bf16));
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// dst[i] = (float) src[i] for a bf16 buffer: the way back from the bf16 wire format of the data-parallel gradient exchange
__global__ __launch_bounds__(256) void widen_flat_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i * 8 < n; i += stride) {
    if (i * 8 + 8 <= n) {
      const uint4 v = reinterpret_cast<const uint4*>(src)[i];
      float4 lo, hi;
      lo.x = __uint_as_float(v.x << 16); lo.y = __uint_as_float(v.x & 0xffff0000u);
      lo.z = __uint_as_float(v.y << 16); lo.w = __uint_as_float(v.y & 0xffff0000u);
      hi.x = __uint_as_float(v.z << 16); hi.y = __uint_as_float(v.z & 0xffff0000u);
      hi.z = __uint_as_float(v.w << 16); hi.w = __uint_as_float(v.w & 0xffff0000u);
      reinterpret_cast<float4*>(dst)[2 * i] = lo;
      reinterpret_cast<float4*>(dst)[2 * i + 1] = hi;
    } else {
      for (int64_t e = i * 8; e < n; ++e) dst[e] = bf16_to_f32(src[e]);
    }
  }
}
extern "C" int asr_widen_flat(const void* src_bf16, float* dst, int64_t n, hipStream_t s) {
  ASR_CHECK_ARG(src_bf16 && dst && n >= 0 && aligned16(src_bf16) && aligned16(dst));
  if (n == 0) return ASR_OK;
  int64_t blocks = ceil_div64(n, 2048 * 2);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(widen_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const bf16_t*>(src_bf16), dst, n);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// dst[r][h2 * C + c] = src[r][c * H2 + h2]: the model's feature order (c * F' + f, transformer.py:74-76) -> channel last.  One thread =
// one 16-byte piece of dst (EPC consecutive channels of one h2), gathered from EPC elements H2 apart (5 MB at the benchmark's
// 512 x 5120 weight: L2 resident, one launch per step).
template <typename T>
__global__ __launch_bounds__(256) void permute_cols_tcf_kernel(const T* __restrict__ src, int64_t ld_src, T* __restrict__ dst, int64_t ld_dst,
                                                               int rows, int C, int H2) {
  constexpr int EPC = DT<T>::EPC;
  const int pieces = H2 * (C / EPC);
  const int64_t total = (int64_t)rows * pieces, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const int r = (int)(i / pieces), q = (int)(i - (int64_t)r * pieces), h2 = q / (C / EPC), c0 = (q - h2 * (C / EPC)) * EPC;
    Chunk<T> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.e[e] = src[(int64_t)r * ld_src + (int64_t)(c0 + e) * H2 + h2];
    *reinterpret_cast<uint4*>(dst + (int64_t)r * ld_dst + (int64_t)h2 * C + c0) = o.v;
  }
}
extern "C" int asr_permute_cols_tcf(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int rows, int C, int H2, int dtype,
                                    hipStream_t s) {
  ASR_CHECK_ARG(src && dst && rows >= 0 && C > 0 && H2 > 0 && (dtype == ASR_F32 || dtype == ASR_BF16));
  const int epc = dtype == ASR_F32 ? 4 : 8;
  if (C % epc != 0 || ld_dst % epc != 0 || !aligned16(dst) || ld_src < (int64_t)C * H2 || ld_dst < (int64_t)C * H2) return ASR_EUNSUPPORTED;
  if (rows == 0) return ASR_OK;
  int64_t blocks = ceil_div64((int64_t)rows * H2 * (C / epc), 256);
  if (blocks > 4096) blocks = 4096;
  if (dtype == ASR_F32) hipLaunchKernelGGL((permute_cols_tcf_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, ld_src, (float*)dst, ld_dst, rows, C, H2);
  else hipLaunchKernelGGL((permute_cols_tcf_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)src, ld_src, (bf16_t*)dst, ld_dst, rows, C, H2);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_cast_flat(const float* src, void* dst, int64_t n, int dtype, hipStream_t s) {
  ASR_CHECK_ARG(src && dst && n >= 0 && aligned16(src));
  if (n == 0) return ASR_OK;
  int64_t blocks = ceil_div64(n, 1024 * 2);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (dtype == ASR_F32) hipLaunchKernelGGL((cast_flat_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, src, (float*)dst, n);
  else if (dtype == ASR_BF16) hipLaunchKernelGGL((cast_flat_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, src, (bf16_t*)dst, n);
  else return ASR_EINVAL;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
