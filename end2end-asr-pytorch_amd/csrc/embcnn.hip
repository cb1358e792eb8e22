// emb_cnn front end (reference: models/asr/transformer.py:33-40): Conv2d(1,32,(41,11),s=(2,2),p=(0,10)) + BatchNorm2d +
// Hardtanh(0,20) + Conv2d(32,32,(21,11),s=(2,1)) + BatchNorm2d + Hardtanh(0,20).
//
// The two big-window strided convolutions are lowered to GEMMs on the existing MFMA kernels (gemm.hip):
//   forward   y  = col  . W^T (+bias)          asr_gemm_nt     col = im2col(x), rows m = (b, oh, ow), k = (ky, kx, c)
//   wgrad     dW += dy^T . col  (db fused)     asr_gemm_tn
//   dgrad     dcol = dy . W ;  dx = col2im(dcol)   asr_gemm_nn + asr_col2im (gather form, no atomics)
// This file holds the data-movement kernels either side of those GEMMs and the BatchNorm(+Hardtanh) kernels.
// Activations are NHWC; the last BN/Hardtanh writes the encoder's (B, T', C*F') layout directly (feature = c*F' + f).
#include "common.h"

namespace {

struct ColArgs {
  int B, H, W, C, KH, KW, SH, SW, PH, PW, OH, OW;
  int64_t ld, M, rows_alloc;
  int K;
};

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return DT<T>::ld(p); }

// ---------------------------------------------------------------------------------------------- im2col
// one thread = 8 consecutive k of one row.  C % 8 == 0: the 8 elements are 8 channels of one tap (one vector load).
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) im2col_kernel(const TI* __restrict__ x, TO* __restrict__ col, ColArgs a) {
  const int64_t chunks = a.ld / 8;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= a.rows_alloc * chunks) return;
  const int64_t m = gid / chunks;
  const int k0 = (int)(gid - m * chunks) * 8;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = 0.f;
  if (m < a.M && k0 < a.K) {
    const int ow = (int)(m % a.OW);
    const int64_t t = m / a.OW;
    const int oh = (int)(t % a.OH);
    const int b = (int)(t / a.OH);
    const int ih0 = oh * a.SH - a.PH, iw0 = ow * a.SW - a.PW;
    if (a.C % 8 == 0) {
      const int tap = k0 / a.C, c0 = k0 - tap * a.C;
      const int ky = tap / a.KW, kx = tap - ky * a.KW;
      const int ih = ih0 + ky, iw = iw0 + kx;
      if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {
        const TI* p = x + (((int64_t)b * a.H + ih) * a.W + iw) * a.C + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ldf(p + j);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        if (k < a.K) {
          const int tap = k / a.C, c = k - tap * a.C;
          const int ky = tap / a.KW, kx = tap - ky * a.KW;
          const int ih = ih0 + ky, iw = iw0 + kx;
          if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) v[j] = ldf(x + (((int64_t)b * a.H + ih) * a.W + iw) * a.C + c);
        }
      }
    }
  }
  TO* o = col + m * a.ld + k0;
  if constexpr (sizeof(TO) == 2) {
    Chunk<bf16_t> c;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.e[j] = f32_to_bf16(v[j]);
    *reinterpret_cast<uint4*>(o) = c.v;
  } else {
    *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// ---------------------------------------------------------------------------------------------- col2im (gather)
// dx[b,h,w,c] = sum over the taps (ky,kx) whose window covers (h,w) of dcol[(b,oh,ow)][(ky,kx,c)].  One thread = 8 channels
// of one input pixel; every dcol element is read exactly once.
template <typename T>
__global__ void __launch_bounds__(256) col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, ColArgs a) {
  const int cch = a.C / 8;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t npx = (int64_t)a.B * a.H * a.W;
  if (gid >= npx * cch) return;
  const int64_t px = gid / cch;
  const int c0 = (int)(gid - px * cch) * 8;
  const int w = (int)(px % a.W);
  const int64_t t = px / a.W;
  const int h = (int)(t % a.H);
  const int b = (int)(t / a.H);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < a.KH; ++ky) {
    const int th = h + a.PH - ky;
    if (th < 0 || th % a.SH != 0) continue;
    const int oh = th / a.SH;
    if (oh >= a.OH) continue;
    for (int kx = 0; kx < a.KW; ++kx) {
      const int tw = w + a.PW - kx;
      if (tw < 0 || tw % a.SW != 0) continue;
      const int ow = tw / a.SW;
      if (ow >= a.OW) continue;
      const T* p = dcol + (((int64_t)b * a.OH + oh) * a.OW + ow) * a.ld + (ky * a.KW + kx) * a.C + c0;
      if constexpr (sizeof(T) == 2) {
        Chunk<bf16_t> c;
        c.v = *reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += bf16_to_f32(c.e[j]);
      } else {
        const float4 u0 = *reinterpret_cast<const float4*>(p), u1 = *reinterpret_cast<const float4*>(p + 4);
        acc[0] += u0.x; acc[1] += u0.y; acc[2] += u0.z; acc[3] += u0.w;
        acc[4] += u1.x; acc[5] += u1.y; acc[6] += u1.z; acc[7] += u1.w;
      }
    }
  }
  T* o = dx + px * a.C + c0;
#pragma unroll
  for (int j = 0; j < 8; ++j) DT<T>::st(o + j, acc[j]);
}

// ---------------------------------------------------------------------------------------------- BatchNorm (+Hardtanh)
// y is the fp32 GEMM output (M, ldy), channel = column.  256 threads = (256 / C) row lanes x C channels.
struct BnArgs {
  const float* y;
  int64_t ldy, M;
  int C;
  const float *mean, *rstd, *gamma, *beta;
  float lo, hi;
  int tH, tW;          // > 0: rows are (b, h, w) over (B, tH, tW) and the activation side uses (B, tW, C*tH), feature c*tH + h
  int64_t ldo;
  int yW, yOW;         // yOW > 0: y lives on the row grid of a window GEMM -- groups of yW rows of which the first yOW are outputs
                       // (row m of the convolution = y row (m / yOW) * yW + m % yOW); 0: y is compact
  // Length-masked statistics (round 6; the *_v entry points): rows are (group, t) with t = m % vW the time step; only t < *valid take
  // part in the batch statistics and receive a gradient -- the rest is padding that a shape BUCKET added behind the batch as collated
  // (trainer --graph-buckets), which the reference's BatchNorm never sees.  *valid lives in device memory: one captured graph serves
  // every batch of its bucket.  valid == nullptr: every row counts.
  const int* valid;
  int vW;
};
__device__ __forceinline__ bool bn_row_ok(const BnArgs& a, int64_t m, int nv) { return a.valid == nullptr || (int)((unsigned)m % (unsigned)a.vW) < nv; }

// row m of the convolution output -> row of a buffer laid out in groups of gw rows with gow outputs each (gow == 0: identity)
__device__ __forceinline__ int64_t grid_row(int64_t m, int gw, int gow) {
  if (gow <= 0) return m;
  const unsigned q = (unsigned)m / (unsigned)gow;          // bn_args_ok: M < 2^31
  return (int64_t)q * gw + ((unsigned)m - q * (unsigned)gow);
}
__device__ __forceinline__ float bn_y(const BnArgs& a, int64_t m, int c) { return a.y[grid_row(m, a.yW, a.yOW) * a.ldy + c]; }

__device__ __forceinline__ int64_t act_index(const BnArgs& a, int64_t m, int c) {
  if (a.tH > 0) {
    const int w = (int)(m % a.tW);
    const int64_t t = m / a.tW;
    const int h = (int)(t % a.tH);
    const int64_t b = t / a.tH;
    return ((b * a.tW + w) * a.C + c) * a.tH + h;
  }
  return m * a.ldo + c;
}

constexpr int BN_ROWS = 512;       // rows per workgroup (the streaming reductions keep 4 independent row loads per thread in flight:
                                   // one load per iteration ran at 1.5 TB/s on the 200 MB fp32 conv output)

// out[0:C] += sum (y - center), out[C:2C] += sum (y - center)^2   (center == nullptr: 0)
// (partial != nullptr: the workgroup's sums go to partial[blockIdx.x][2C] instead -- summed in a fixed order by the caller, so
//  the batch statistics, and with them the forward pass, are bit-reproducible from run to run)
__global__ void __launch_bounds__(256) bn_stats_kernel(BnArgs a, const float* __restrict__ center, float* __restrict__ out,
                                                       float* __restrict__ partial) {
  __shared__ float red[2][256];
  const int C = a.C, c = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
  const float ctr = center ? center[c] : 0.f;
  const int nv = a.valid ? *a.valid : 0;
  const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS;
  const int64_t r1 = r0 + BN_ROWS < a.M ? r0 + BN_ROWS : a.M;
  float s = 0.f, q = 0.f;
  int64_t r = r0 + rl;
  for (; r + 3 * nrl < r1; r += 4 * nrl) {
    float v0 = bn_y(a, r, c) - ctr, v1 = bn_y(a, r + nrl, c) - ctr;
    float v2 = bn_y(a, r + 2 * nrl, c) - ctr, v3 = bn_y(a, r + 3 * nrl, c) - ctr;
    if (a.valid) {          // (the loads stay unconditional: four in flight; padding rows hold finite numbers)
      v0 = bn_row_ok(a, r, nv) ? v0 : 0.f; v1 = bn_row_ok(a, r + nrl, nv) ? v1 : 0.f;
      v2 = bn_row_ok(a, r + 2 * nrl, nv) ? v2 : 0.f; v3 = bn_row_ok(a, r + 3 * nrl, nv) ? v3 : 0.f;
    }
    s += (v0 + v1) + (v2 + v3);
    q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
  }
  for (; r < r1; r += nrl) {
    const float v = bn_row_ok(a, r, nv) ? bn_y(a, r, c) - ctr : 0.f;
    s += v;
    q += v * v;
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0) {
    for (int i = 1; i < nrl; ++i) {
      s += red[0][i * C + c];
      q += red[1][i * C + c];
    }
    if (partial) {
      partial[(int64_t)blockIdx.x * 2 * C + c] = s;
      partial[(int64_t)blockIdx.x * 2 * C + C + c] = q;
    } else {
      atomicAdd(out + c, s);
      atomicAdd(out + C + c, q);
    }
  }
}

// out[c] = sum over the workgroups' partial rows in a FIXED order: 1024 / C2 row groups (rows g, g + G, ...) per column, each a
// serial chain, then the groups in index order (C2 <= 512)
__global__ void __launch_bounds__(1024) bn_partial_sum_kernel(const float* __restrict__ partial, int64_t nblk, int C2, float* __restrict__ out) {
  __shared__ float red[1024];
  const int G = 1024 / C2, c = threadIdx.x % C2, g = threadIdx.x / C2;
  float s = 0.f;
  if (g < G)
    for (int64_t b = g; b < nblk; b += G) s += partial[b * C2 + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0) {
    for (int i = 1; i < G; ++i) s += red[i * C2 + c];
    out[c] = s;
  }
}

// The two finishing steps of nn.BatchNorm2d's training statistics, each as the tail of the fixed-order sum above (C <= 256):
// mode 1: mean[c] = sum(y) / M;   mode 2 (partial holds the CENTRED sums): var = sum((y - mean)^2) / M, rstd = rsqrt(var + eps) and, with
// momentum >= 0, running_mean / running_var (unbiased: var * M / (M - 1)) / num_batches_tracked -- what cost ten small torch launches per layer.
__global__ void __launch_bounds__(1024) bn_finish_kernel(const float* __restrict__ partial, int64_t nblk, int C, int mode, float inv_m,
                                                         float unbias, float eps, float momentum, float* __restrict__ mean,
                                                         float* __restrict__ rstd, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, int64_t* __restrict__ num_batches,
                                                         const int* __restrict__ valid, int64_t groups) {
  __shared__ float red[1024];
  if (valid) {              // length-masked statistics: the count is groups x *valid, known on the device only
    const float m = (float)(groups * (int64_t)valid[0]);
    inv_m = 1.f / fmaxf(m, 1.f);
    unbias = m / fmaxf(m - 1.f, 1.f);
  }
  const int C2 = 2 * C, G = 1024 / C2, c = threadIdx.x % C2, g = threadIdx.x / C2;
  float s = 0.f;
  if (g < G)
    for (int64_t b = g; b < nblk; b += G) s += partial[b * C2 + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (g != 0) return;
  for (int i = 1; i < G; ++i) s += red[i * C2 + c];
  if (mode == 1) {
    if (c < C) mean[c] = s * inv_m;
    return;
  }
  if (c < C) return;
  const int ch = c - C;
  const float var = s * inv_m;
  rstd[ch] = rsqrtf(var + eps);
  if (momentum >= 0.f && running_mean) {
    running_mean[ch] = running_mean[ch] * (1.f - momentum) + momentum * mean[ch];
    running_var[ch] = running_var[ch] * (1.f - momentum) + momentum * unbias * var;
    if (ch == 0 && num_batches) num_batches[0] += 1;
  }
}

template <typename T>
__global__ void __launch_bounds__(256) bn_act_fwd_kernel(BnArgs a, T* __restrict__ out) {
  const int C = a.C, c = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
  const float mu = a.mean[c], rs = a.rstd[c], g = a.gamma[c], be = a.beta[c];
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int64_t r = r0 + rl; r < r0 + 64 && r < a.M; r += nrl) {
    float z = (bn_y(a, r, c) - mu) * rs * g + be;
    z = fminf(fmaxf(z, a.lo), a.hi);
    DT<T>::st(out + act_index(a, r, c), z);
  }
}

// sums[0:C] += sum dz, sums[C:2C] += sum dz * xhat   with dz = dout masked by lo < z < hi (Hardtanh backward)
template <typename T>
__global__ void __launch_bounds__(256) bn_act_bwd_reduce_kernel(BnArgs a, const T* __restrict__ dout, float* __restrict__ sums) {
  __shared__ float red[2][256];
  const int C = a.C, c = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
  const float mu = a.mean[c], rs = a.rstd[c], g = a.gamma[c], be = a.beta[c];
  const int64_t r0 = (int64_t)blockIdx.x * BN_ROWS;
  const int64_t r1 = r0 + BN_ROWS < a.M ? r0 + BN_ROWS : a.M;
  float s = 0.f, q = 0.f;
  const int nv = a.valid ? *a.valid : 0;
  int64_t r = r0 + rl;
  for (; r + 3 * nrl < r1; r += 4 * nrl) {          // all 8 loads of four rows first; rows outside the Hardtanh window contribute 0
    float yv[4], dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      yv[u] = bn_y(a, r + u * nrl, c);
      dv[u] = DT<T>::ld(dout + act_index(a, r + u * nrl, c));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float xh = (yv[u] - mu) * rs;
      const float z = xh * g + be;
      const float d = (z > a.lo && z < a.hi && bn_row_ok(a, r + u * nrl, nv)) ? dv[u] : 0.f;
      s += d;
      q += d * xh;
    }
  }
  for (; r < r1; r += nrl) {
    const float xh = (bn_y(a, r, c) - mu) * rs;
    const float z = xh * g + be;
    if (z > a.lo && z < a.hi && bn_row_ok(a, r, nv)) {
      const float d = DT<T>::ld(dout + act_index(a, r, c));
      s += d;
      q += d * xh;
    }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = q;
  __syncthreads();
  if (rl == 0) {
    for (int i = 1; i < nrl; ++i) {
      s += red[0][i * C + c];
      q += red[1][i * C + c];
    }
    atomicAdd(sums + c, s);
    atomicAdd(sums + C + c, q);
  }
}

// dy = gamma * rstd * (dz - sum(dz)/M - xhat * sum(dz*xhat)/M)     (training-mode BatchNorm backward)
template <typename T>
__global__ void __launch_bounds__(256) bn_act_bwd_kernel(BnArgs a, const T* __restrict__ dout, const float* __restrict__ sums,
                                                         T* __restrict__ dy, int64_t lddy, int dW, int dOW) {
  const int C = a.C, c = threadIdx.x % C, rl = threadIdx.x / C, nrl = 256 / C;
  const float mu = a.mean[c], rs = a.rstd[c], g = a.gamma[c], be = a.beta[c];
  const int nv = a.valid ? *a.valid : 0;
  const float inv = a.valid ? 1.f / fmaxf((float)((a.M / a.vW) * (int64_t)nv), 1.f) : 1.f / (float)a.M;
  const float m1 = sums[c] * inv, m2 = sums[C + c] * inv;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  for (int64_t r = r0 + rl; r < r0 + 64 && r < a.M; r += nrl) {
    const float xh = (bn_y(a, r, c) - mu) * rs;
    const float z = xh * g + be;
    const float d = (z > a.lo && z < a.hi) ? DT<T>::ld(dout + act_index(a, r, c)) : 0.f;
    // a padding row took no part in the statistics: it gets no gradient (its dout is zero already: the encoder masks it)
    DT<T>::st(dy + grid_row(r, dW, dOW) * lddy + c, bn_row_ok(a, r, nv) ? g * rs * (d - m1 - xh * m2) : 0.f);
  }
}

bool col_args_ok(const ColArgs& a) {
  return a.B > 0 && a.H > 0 && a.W > 0 && a.C > 0 && a.KH > 0 && a.KW > 0 && a.SH > 0 && a.SW > 0 && a.PH >= 0 && a.PW >= 0 &&
         a.OH == (a.H + 2 * a.PH - a.KH) / a.SH + 1 && a.OW == (a.W + 2 * a.PW - a.KW) / a.SW + 1 && a.OH > 0 && a.OW > 0;
}

bool grid_ok(int64_t M, int gw, int gow) { return gow == 0 || (gow > 0 && gw >= gow && M < ((int64_t)1 << 31) && M % gow == 0); }

bool bn_args_ok(const BnArgs& a) {
  return a.y && a.M > 0 && a.C > 0 && a.C <= 256 && 256 % a.C == 0 && a.ldy >= a.C && a.mean && a.rstd && a.gamma && a.beta &&
         (a.tH == 0 || (a.tW > 0 && a.M % ((int64_t)a.tH * a.tW) == 0)) && grid_ok(a.M, a.yW, a.yOW);
}

}  // namespace

extern "C" int asr_im2col(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW,
                          int OH, int OW, int64_t ld_col, int64_t rows_alloc, int in_dtype, int out_dtype, hipStream_t stream) {
  ASR_CHECK_ARG(x && col);
  ColArgs a{B, H, W, C, KH, KW, SH, SW, PH, PW, OH, OW, ld_col, (int64_t)B * OH * OW, rows_alloc, KH * KW * C};
  ASR_CHECK_ARG(col_args_ok(a) && ld_col % 8 == 0 && ld_col >= a.K && rows_alloc >= a.M && aligned16(col));
  ASR_CHECK_ARG((in_dtype == ASR_F32 || in_dtype == ASR_BF16) && (out_dtype == ASR_F32 || out_dtype == ASR_BF16));
  AsrProfScope prof(ASR_OP_LAYOUT, stream);
  const int64_t n = rows_alloc * (ld_col / 8);
  const unsigned grid = (unsigned)ceil_div64(n, 256);
  if (in_dtype == ASR_F32 && out_dtype == ASR_F32)
    im2col_kernel<float, float><<<grid, 256, 0, stream>>>((const float*)x, (float*)col, a);
  else if (in_dtype == ASR_F32)
    im2col_kernel<float, bf16_t><<<grid, 256, 0, stream>>>((const float*)x, (bf16_t*)col, a);
  else if (out_dtype == ASR_BF16)
    im2col_kernel<bf16_t, bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)x, (bf16_t*)col, a);
  else
    return ASR_EUNSUPPORTED;
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_col2im(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int SH, int SW, int PH, int PW,
                          int OH, int OW, int64_t ld_col, int dtype, hipStream_t stream) {
  ASR_CHECK_ARG(dcol && dx);
  ColArgs a{B, H, W, C, KH, KW, SH, SW, PH, PW, OH, OW, ld_col, (int64_t)B * OH * OW, 0, KH * KW * C};
  ASR_CHECK_ARG(col_args_ok(a) && C % 8 == 0 && ld_col % 8 == 0 && ld_col >= a.K && aligned16(dcol) && aligned16(dx));
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  AsrProfScope prof(ASR_OP_LAYOUT, stream);
  const int64_t n = (int64_t)B * H * W * (C / 8);
  const unsigned grid = (unsigned)ceil_div64(n, 256);
  if (dtype == ASR_F32)
    col2im_kernel<float><<<grid, 256, 0, stream>>>((const float*)dcol, (float*)dx, a);
  else
    col2im_kernel<bf16_t><<<grid, 256, 0, stream>>>((const bf16_t*)dcol, (bf16_t*)dx, a);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

// ---------------------------------------------------------------------------------------------- window sum
// The unit-time-stride convolution as ONE dense GEMM over the KW = 1 patches: Z[r', (kx, co)] = X2[r', :] . W[co, kx, :] for every
// (padded) time step r' and tap kx, then y[(g, j), co] = bias[co] + sum_kx Z[g Wg + j + kx, kx Cout + co] for the OW valid steps of
// each (b, oh) group g.  Every element of Z is read exactly once.  One thread = 4 columns of one output row; columns >= Cout of y
// (the row pitch of the BatchNorm kernels) are written as 0.
__global__ void __launch_bounds__(256) window_sum_kernel(const float* __restrict__ Z, int64_t ldz, float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ bias, int64_t M, int Wg, int OW, int KW, int Cout) {
  const int cq = (int)(ldy / 4);
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (gid >= M * cq) return;
  const int64_t m = gid / cq;
  const int c0 = (int)(gid - m * cq) * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c0 < Cout) {
    const int64_t g = m / OW;
    const int j = (int)(m - g * OW);
    const float* z = Z + (g * Wg + j) * ldz + c0;
    if (bias) acc = *reinterpret_cast<const float4*>(bias + c0);
    for (int kx = 0; kx < KW; ++kx) {
      const float4 v = *reinterpret_cast<const float4*>(z + (int64_t)kx * (ldz + Cout));
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(y + m * ldy + c0) = acc;
}

extern "C" int asr_window_sum(const float* Z, int64_t ldz, float* y, int64_t ldy, const float* bias, int64_t groups, int Wg, int OW,
                              int KW, int Cout, hipStream_t stream) {
  ASR_CHECK_ARG(Z && y && groups >= 0 && Wg > 0 && OW > 0 && KW > 0 && Cout > 0 && OW + KW - 1 <= Wg + KW - 1);
  ASR_CHECK_ARG(Cout % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0 && ldy >= Cout && ldz >= (int64_t)KW * Cout && aligned16(Z) && aligned16(y) &&
                (!bias || aligned16(bias)));
  if (groups == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_LAYOUT, stream);
  const int64_t n = groups * OW * (ldy / 4);
  window_sum_kernel<<<(unsigned)ceil_div64(n, 256), 256, 0, stream>>>(Z, ldz, y, ldy, bias, groups * OW, Wg, OW, KW, Cout);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_bn_stats(const float* y, int64_t ldy, int64_t M, int C, const float* center, float* sums, int y_grid_w,
                            int y_grid_ow, hipStream_t stream) {
  ASR_CHECK_ARG(y && sums && M > 0 && C > 0 && C <= 256 && 256 % C == 0 && ldy >= C && grid_ok(M, y_grid_w, y_grid_ow));
  BnArgs a{};
  a.y = y; a.ldy = ldy; a.M = M; a.C = C; a.yW = y_grid_w; a.yOW = y_grid_ow;
  AsrProfScope prof(ASR_OP_ADD_LN, stream);
  bn_stats_kernel<<<(unsigned)ceil_div64(M, BN_ROWS), 256, 0, stream>>>(a, center, sums, nullptr);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int64_t asr_bn_stats_blocks(int64_t M) { return M > 0 ? ceil_div64(M, BN_ROWS) : 0; }

extern "C" int asr_bn_stats_partial(const float* y, int64_t ldy, int64_t M, int C, const float* center, float* partial, float* sums,
                                    int y_grid_w, int y_grid_ow, hipStream_t stream) {
  ASR_CHECK_ARG(y && partial && sums && M > 0 && C > 0 && C <= 256 && 256 % C == 0 && ldy >= C && grid_ok(M, y_grid_w, y_grid_ow));
  BnArgs a{};
  a.y = y; a.ldy = ldy; a.M = M; a.C = C; a.yW = y_grid_w; a.yOW = y_grid_ow;
  AsrProfScope prof(ASR_OP_ADD_LN, stream);
  const int64_t nblk = ceil_div64(M, BN_ROWS);
  bn_stats_kernel<<<(unsigned)nblk, 256, 0, stream>>>(a, center, nullptr, partial);
  bn_partial_sum_kernel<<<1, 1024, 0, stream>>>(partial, nblk, 2 * C, sums);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_bn_batch_stats(const float* y, int64_t ldy, int64_t M, int C, float* partial, float* mean, float* rstd, float eps,
                                  float momentum, float* running_mean, float* running_var, int64_t* num_batches, int y_grid_w,
                                  int y_grid_ow, hipStream_t stream) {
  return asr_bn_batch_stats_v(y, ldy, M, C, partial, mean, rstd, eps, momentum, running_mean, running_var, num_batches, y_grid_w, y_grid_ow,
                              nullptr, 0, stream);
}
extern "C" int asr_bn_batch_stats_v(const float* y, int64_t ldy, int64_t M, int C, float* partial, float* mean, float* rstd, float eps,
                                    float momentum, float* running_mean, float* running_var, int64_t* num_batches, int y_grid_w,
                                    int y_grid_ow, const int* valid_w, int row_w, hipStream_t stream) {
  ASR_CHECK_ARG(y && partial && mean && rstd && M > 0 && C > 0 && C <= 256 && 256 % C == 0 && ldy >= C && grid_ok(M, y_grid_w, y_grid_ow));
  ASR_CHECK_ARG(momentum < 0.f || (running_mean && running_var));
  ASR_CHECK_ARG(valid_w == nullptr || (row_w > 0 && M % row_w == 0 && M < ((int64_t)1 << 31)));
  BnArgs a{};
  a.y = y; a.ldy = ldy; a.M = M; a.C = C; a.yW = y_grid_w; a.yOW = y_grid_ow; a.valid = valid_w; a.vW = row_w;
  const int64_t groups = valid_w ? M / row_w : 0;
  AsrProfScope prof(ASR_OP_ADD_LN, stream);
  const int64_t nblk = ceil_div64(M, BN_ROWS);
  const float inv_m = 1.f / (float)M, unbias = (float)M / (float)(M > 1 ? M - 1 : 1);
  bn_stats_kernel<<<(unsigned)nblk, 256, 0, stream>>>(a, nullptr, nullptr, partial);
  bn_finish_kernel<<<1, 1024, 0, stream>>>(partial, nblk, C, 1, inv_m, unbias, eps, momentum, mean, rstd, nullptr, nullptr, nullptr, valid_w,
                                           groups);
  bn_stats_kernel<<<(unsigned)nblk, 256, 0, stream>>>(a, mean, nullptr, partial);
  bn_finish_kernel<<<1, 1024, 0, stream>>>(partial, nblk, C, 2, inv_m, unbias, eps, momentum, mean, rstd, running_mean, running_var,
                                           num_batches, valid_w, groups);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_bn_act_fwd(const float* y, int64_t ldy, void* out, int64_t ldo, int64_t M, int C, const float* mean,
                              const float* rstd, const float* gamma, const float* beta, float lo, float hi, int tH, int tW,
                              int y_grid_w, int y_grid_ow, int dtype, hipStream_t stream) {
  BnArgs a{y, ldy, M, C, mean, rstd, gamma, beta, lo, hi, tH, tW, ldo, y_grid_w, y_grid_ow, nullptr, 0};
  ASR_CHECK_ARG(out && bn_args_ok(a) && (tH > 0 || ldo >= C));
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  AsrProfScope prof(ASR_OP_ADD_LN, stream);
  const unsigned grid = (unsigned)ceil_div64(M, 64);
  if (dtype == ASR_F32)
    bn_act_fwd_kernel<float><<<grid, 256, 0, stream>>>(a, (float*)out);
  else
    bn_act_fwd_kernel<bf16_t><<<grid, 256, 0, stream>>>(a, (bf16_t*)out);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_bn_act_bwd_reduce(const void* dout, int64_t ldo, const float* y, int64_t ldy, int64_t M, int C,
                                     const float* mean, const float* rstd, const float* gamma, const float* beta, float lo,
                                     float hi, int tH, int tW, int y_grid_w, int y_grid_ow, float* sums, int dtype, hipStream_t stream) {
  return asr_bn_act_bwd_reduce_v(dout, ldo, y, ldy, M, C, mean, rstd, gamma, beta, lo, hi, tH, tW, y_grid_w, y_grid_ow, sums, nullptr, 0,
                                 dtype, stream);
}
extern "C" int asr_bn_act_bwd_reduce_v(const void* dout, int64_t ldo, const float* y, int64_t ldy, int64_t M, int C,
                                       const float* mean, const float* rstd, const float* gamma, const float* beta, float lo,
                                       float hi, int tH, int tW, int y_grid_w, int y_grid_ow, float* sums, const int* valid_w, int row_w,
                                       int dtype, hipStream_t stream) {
  BnArgs a{y, ldy, M, C, mean, rstd, gamma, beta, lo, hi, tH, tW, ldo, y_grid_w, y_grid_ow, valid_w, row_w};
  ASR_CHECK_ARG(valid_w == nullptr || (row_w > 0 && M % row_w == 0));
  ASR_CHECK_ARG(dout && sums && bn_args_ok(a) && (tH > 0 || ldo >= C));
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  AsrProfScope prof(ASR_OP_ADD_LN, stream);
  const unsigned grid = (unsigned)ceil_div64(M, BN_ROWS);
  if (dtype == ASR_F32)
    bn_act_bwd_reduce_kernel<float><<<grid, 256, 0, stream>>>(a, (const float*)dout, sums);
  else
    bn_act_bwd_reduce_kernel<bf16_t><<<grid, 256, 0, stream>>>(a, (const bf16_t*)dout, sums);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_bn_act_bwd(const void* dout, int64_t ldo, const float* y, int64_t ldy, void* dy, int64_t lddy, int64_t M, int C,
                              const float* mean, const float* rstd, const float* gamma, const float* beta, float lo, float hi,
                              int tH, int tW, int y_grid_w, int y_grid_ow, int dy_grid_w, int dy_grid_ow, const float* sums, int dtype,
                              hipStream_t stream) {
  return asr_bn_act_bwd_v(dout, ldo, y, ldy, dy, lddy, M, C, mean, rstd, gamma, beta, lo, hi, tH, tW, y_grid_w, y_grid_ow, dy_grid_w,
                          dy_grid_ow, sums, nullptr, 0, dtype, stream);
}
extern "C" int asr_bn_act_bwd_v(const void* dout, int64_t ldo, const float* y, int64_t ldy, void* dy, int64_t lddy, int64_t M, int C,
                                const float* mean, const float* rstd, const float* gamma, const float* beta, float lo, float hi,
                                int tH, int tW, int y_grid_w, int y_grid_ow, int dy_grid_w, int dy_grid_ow, const float* sums,
                                const int* valid_w, int row_w, int dtype, hipStream_t stream) {
  BnArgs a{y, ldy, M, C, mean, rstd, gamma, beta, lo, hi, tH, tW, ldo, y_grid_w, y_grid_ow, valid_w, row_w};
  ASR_CHECK_ARG(valid_w == nullptr || (row_w > 0 && M % row_w == 0));
  ASR_CHECK_ARG(dout && dy && sums && bn_args_ok(a) && (tH > 0 || ldo >= C) && lddy >= C && grid_ok(M, dy_grid_w, dy_grid_ow));
  ASR_CHECK_ARG(dtype == ASR_F32 || dtype == ASR_BF16);
  AsrProfScope prof(ASR_OP_ADD_LN, stream);
  const unsigned grid = (unsigned)ceil_div64(M, 64);
  if (dtype == ASR_F32)
    bn_act_bwd_kernel<float><<<grid, 256, 0, stream>>>(a, (const float*)dout, sums, (float*)dy, lddy, dy_grid_w, dy_grid_ow);
  else
    bn_act_bwd_kernel<bf16_t><<<grid, 256, 0, stream>>>(a, (const bf16_t*)dout, sums, (bf16_t*)dy, lddy, dy_grid_w, dy_grid_ow);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
