// Greedy decoding, one token per sequence and step (reference: models/asr/transformer.py:316-393 re-runs the whole decoder over
// the growing prefix 300 times; asr_hip/decode.py keeps per-layer key / value caches and pushes ONE row per sequence through the
// layers).  At B <= 32 rows every launch of that step sits on the launch-latency floor (4.5 us per dependent kernel in a replayed
// hipGraph, measured: profiles/r02_decode_trace.txt), so what matters is the NUMBER of launches and one memory round trip per launch:
//   asr_dec_gemm    out(B, N) = act(x W^T + b) for B <= 32 rows: a workgroup owns 32 output columns, its 4 waves split K, every
//                   operand fragment of v_mfma_f32_32x32x16_bf16 is ONE 16-byte load straight from global memory (weights: no LDS,
//                   no barrier in the K loop, all loads of a wave in flight at once), partial sums meet in LDS.  The input row can be
//                   produced by a prologue instead of read: x = LayerNorm(y + residual) (the previous sub-layer's epilogue: every
//                   workgroup recomputes the 32 rows -- 64 KB from L2 -- and workgroup 0 stores them for the next residual) or
//                   x = embedding[token] * scale + pe[t].  That folds 13 LayerNorm / embedding launches per token into the GEMMs.
//   asr_dec_attn    one query row per (sequence, head), a wave each: appends this position's key / value row to the cache, scores
//                   by lane-per-key dot products, softmax in registers, P V by 16-byte value chunks.  Position t is read from device
//                   memory (state[0]), so one captured graph serves every step.
//   asr_dec_finish  arg max of the logits row (lowest index on ties, like torch.argmax / the reference's topk(1)) -> next token,
//                   done flag, output column t; the LAST workgroup to finish advances state[0].
// 62 launches per token -> 34.  bf16 storage, fp32 accumulation (the fp32 parity mode keeps the kernel-per-op path).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct DecGemmArgs {
  const bf16_t* W; const float* bias; void* out;
  int64_t ldw, ldo;
  int B, N, K, relu;
  const bf16_t* X; int64_t ldx;                                   // prologue 0: the input rows
  const bf16_t* Y; const bf16_t* R; const float* gamma; const float* beta; float eps;   // prologue 1: x = LN(Y + R) gamma + beta
  bf16_t* x_out;                                                   // prologue 1 / 2: workgroup 0 stores x (B, K)
  const int64_t* tok; const float* table; const float* pe; float scale; const int64_t* state;   // prologue 2
};

// PRO 0: x read from memory; 1: LayerNorm(Y + R); 2: embedding row * scale + pe[t].   GS = K steps whose loads are issued together.
template <int PRO, typename TO, int GS>
__global__ __launch_bounds__(256) void dec_gemm_kernel(DecGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int K = p.K;
  const int XP = K * 2 + 16;                                       // LDS row pitch of the x tile: rows 4 banks apart
  float* red = reinterpret_cast<float*>(smem);                     // [3][16][64] partial accumulators of waves 1..3
  unsigned char* xs = smem + 3 * 16 * 64 * 4;                      // PRO != 0: x tile [32][XP]

  if (PRO != 0) {
    const int c0 = lane * 8;                                       // K <= 512: one 16-byte chunk per lane and row
    const bool live = c0 < K;
    float gm[8], bt[8];
    if (PRO == 1) {
      const int cc = live ? c0 : 0;
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + cc + j);
        const float4 b4 = *reinterpret_cast<const float4*>(p.beta + cc + j);
        gm[j] = g4.x; gm[j + 1] = g4.y; gm[j + 2] = g4.z; gm[j + 3] = g4.w;
        bt[j] = b4.x; bt[j + 1] = b4.y; bt[j + 2] = b4.z; bt[j + 3] = b4.w;
      }
    }
    const int64_t t = PRO == 2 ? p.state[0] : 0;
    // all loads of the wave's 8 rows first (one memory round trip), then the arithmetic
    Chunk<bf16_t> cy[8], cr[8];
    float ev[PRO == 2 ? 8 : 1][8];
    float pv[8];
    if (PRO == 2) {
      const int cc = live ? c0 : 0;
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        const float4 q4 = *reinterpret_cast<const float4*>(p.pe + t * K + cc + j);
        pv[j] = q4.x; pv[j + 1] = q4.y; pv[j + 2] = q4.z; pv[j + 3] = q4.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = wave * 8 + i;
      const int mm = m < p.B ? m : p.B - 1, cc = live ? c0 : 0;
      if (PRO == 1) {
        cy[i].v = *reinterpret_cast<const uint4*>(p.Y + (int64_t)mm * K + cc);
        cr[i].v = *reinterpret_cast<const uint4*>(p.R + (int64_t)mm * K + cc);
      } else {
        const float* e = p.table + p.tok[mm] * (int64_t)K + cc;
#pragma unroll
        for (int j = 0; j < 8; j += 4) {
          const float4 e4 = *reinterpret_cast<const float4*>(e + j);
          ev[PRO == 2 ? i : 0][j] = e4.x; ev[PRO == 2 ? i : 0][j + 1] = e4.y; ev[PRO == 2 ? i : 0][j + 2] = e4.z; ev[PRO == 2 ? i : 0][j + 3] = e4.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = wave * 8 + i;
      Chunk<bf16_t> o;
      o.v = make_uint4(0u, 0u, 0u, 0u);
      if (PRO == 1) {
        // z = y + residual rounded to the storage type first (what asr_add_ln_fwd stores and normalises), fp32 statistics
        float z[8], s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          z[j] = live ? bf16_to_f32(f32_to_bf16(bf16_to_f32(cy[i].e[j]) + bf16_to_f32(cr[i].e[j]))) : 0.f;
          s += z[j];
        }
        const float mu = wave_sum(s) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = live ? z[j] - mu : 0.f; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / (float)K + p.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = f32_to_bf16((z[j] - mu) * rs * gm[j] + bt[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = f32_to_bf16(ev[PRO == 2 ? i : 0][j] * p.scale + pv[j]);
      }
      if (m >= p.B) o.v = make_uint4(0u, 0u, 0u, 0u);
      if (live) {
        *reinterpret_cast<uint4*>(xs + m * XP + c0 * 2) = o.v;
        if (blockIdx.x == 0 && p.x_out && m < p.B) *reinterpret_cast<uint4*>(p.x_out + (int64_t)m * K + c0) = o.v;
      }
    }
    __syncthreads();
  }

  // ---- K loop: wave w contracts k in [w K/4, (w+1) K/4); A = 32 weight rows (output columns), B = the 32 input rows
  const int kw = wave * (K >> 2);
  const int ksteps = K >> 6;
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nrow = n0 + lr < p.N ? n0 + lr : p.N - 1;
  const bf16_t* wp = p.W + (int64_t)nrow * p.ldw + kw + 8 * half;
  const bf16_t* xp = PRO == 0 ? p.X + (int64_t)(lr < p.B ? lr : p.B - 1) * p.ldx + kw + 8 * half : nullptr;
  const unsigned char* xl = xs + lr * XP + (kw + 8 * half) * 2;
  for (int s0 = 0; s0 < ksteps; s0 += GS) {
    uint4 a[GS], b[GS];
#pragma unroll
    for (int i = 0; i < GS; ++i)
      if (s0 + i < ksteps) {
        a[i] = *reinterpret_cast<const uint4*>(wp + (s0 + i) * 16);
        if (PRO == 0) b[i] = *reinterpret_cast<const uint4*>(xp + (s0 + i) * 16);
        else b[i] = *reinterpret_cast<const uint4*>(xl + (s0 + i) * 32);
      }
#pragma unroll
    for (int i = 0; i < GS; ++i)
      if (s0 + i < ksteps)
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[i]), acc, 0, 0, 0);
  }

  // ---- the four waves' partial tiles meet in LDS (fragment layout: lane-private slots)
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
  // D fragment: column (input row m) = lane & 31, row (output column) = 8 (r / 4) + 4 (lane >> 5) + (r & 3)
  const int m = lr;
  if (m >= p.B) return;
  TO* orow = static_cast<TO*>(p.out) + (int64_t)m * p.ldo;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = n0 + 8 * q + 4 * half;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = acc[4 * q + e] + ((p.bias && n + e < p.N) ? p.bias[n + e] : 0.f);
      if (p.relu) v[e] = fmaxf(v[e], 0.f);
    }
    if (n + 3 < p.N) {
      if constexpr (sizeof(TO) == 4) {
        *reinterpret_cast<float4*>(orow + n) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<uint2*>(orow + n) = o;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < p.N) DT<TO>::st(orow + n + e, v[e]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ single-query attention
struct DecAttnArgs {
  const bf16_t* q; int64_t ldq;
  const bf16_t* kn; const bf16_t* vn; int64_t ldn;       // this position's key / value rows (self attention), or null
  bf16_t* kc; bf16_t* vc; int64_t cbs, cld;              // caches / encoder keys and values: (B, rows, H * 64), batch and row strides
  int rows;                                              // cache length (self attention) or number of keys (cross attention)
  bf16_t* out; int64_t ldo;
  int B, H; float scale;
  const int64_t* state;                                  // self attention: state[0] = position t (keys 0..t); null: all `rows` keys
};

constexpr int DEC_NI = 8;                                // keys per lane: up to 512 keys

__global__ __launch_bounds__(256) void dec_attn_kernel(DecAttnArgs p) {
  __shared__ float pbuf[4][DEC_NI * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * 4 + wave;
  if (bh >= p.B * p.H) return;                           // waves are independent: no workgroup barrier below
  const int b = bh / p.H, h = bh % p.H;
  int t = -1, L = p.rows;
  if (p.state) {
    const int64_t ts = p.state[0];
    if (ts < p.rows) { t = (int)ts; L = t + 1; }
  }
  const bf16_t* kc = p.kc + b * p.cbs + h * 64;
  const bf16_t* vc = p.vc + b * p.cbs + h * 64;
  const bf16_t* kn = p.kn ? p.kn + (int64_t)b * p.ldn + h * 64 : kc;
  const bf16_t* vn = p.vn ? p.vn + (int64_t)b * p.ldn + h * 64 : vc;
  const int tsel = p.kn ? t : -1;                        // row t comes from the source rows when it is appended by this launch
  if (tsel >= 0 && lane < 16) {                          // append: row t of both caches (read below from the source rows)
    const int c = (lane & 7) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>((lane < 8 ? kn : vn) + c);
    *reinterpret_cast<uint4*>((lane < 8 ? p.kc : p.vc) + b * p.cbs + (int64_t)t * p.cld + h * 64 + c) = v;
  }
  float qf[64];
  {
    const bf16_t* q = p.q + (int64_t)b * p.ldq + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      Chunk<bf16_t> ch;
      ch.v = *reinterpret_cast<const uint4*>(q + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[c * 8 + e] = bf16_to_f32(ch.e[e]) * p.scale;
    }
  }
  float s[DEC_NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < DEC_NI; ++i) {
    s[i] = -INFINITY;
    if (i * 64 < L) {
      const int j = lane + 64 * i;
      const int jj = j < L ? j : L - 1;
      const bf16_t* kr = jj == tsel ? kn : kc + (int64_t)jj * p.cld;
      Chunk<bf16_t> ch[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) ch[c].v = *reinterpret_cast<const uint4*>(kr + c * 8);
      float d0 = 0.f, d1 = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          d0 += qf[c * 8 + e] * bf16_to_f32(ch[c].e[e]);
          d1 += qf[c * 8 + e + 1] * bf16_to_f32(ch[c].e[e + 1]);
        }
      if (j < L) s[i] = d0 + d1;
      mx = fmaxf(mx, s[i]);
    }
  }
  mx = wave_max(mx);
  float l = 0.f;
#pragma unroll
  for (int i = 0; i < DEC_NI; ++i)
    if (i * 64 < L) {
      const float pr = s[i] == -INFINITY ? 0.f : __expf(s[i] - mx);
      l += pr;
      pbuf[wave][lane + 64 * i] = bf16_to_f32(f32_to_bf16(pr));      // the probabilities enter P V in the storage type
    }
  l = wave_sum(l);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // P V: lane = (8-channel chunk c, key group jg): keys jg, jg + 8, ...; the 8 groups meet by three xor shuffles
  const int c = (lane & 7) * 8, jg = lane >> 3;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll 4
  for (int j = jg; j < L; j += 8) {
    const float pj = pbuf[wave][j];
    const bf16_t* vr = j == tsel ? vn : vc + (int64_t)j * p.cld;
    Chunk<bf16_t> ch;
    ch.v = *reinterpret_cast<const uint4*>(vr + c);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += pj * bf16_to_f32(ch.e[e]);
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (jg == 0) {
    const float inv = l > 0.f ? 1.f / l : 0.f;
    Chunk<bf16_t> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = f32_to_bf16(acc[e] * inv);
    *reinterpret_cast<uint4*>(p.out + (int64_t)b * p.ldo + h * 64 + c) = o.v;
  }
}

// ------------------------------------------------------------------------------------------------ next token
__global__ __launch_bounds__(256) void dec_finish_kernel(const float* __restrict__ logits, int64_t ld, int V, int64_t* __restrict__ tok,
                                                         uint8_t* __restrict__ done, int64_t* __restrict__ out, int B, int max_len,
                                                         int eos, int64_t* state, int32_t* ticket) {
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t t = state[0];
  const float* l = logits + (int64_t)row * ld;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = tid; v < V; v += 256) {
    const float x = l[v];
    if (x > bv) { bv = x; bi = v; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
  __syncthreads();                                       // also: every thread of this workgroup has read state[0]
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
    const int64_t id = bi == 0x7fffffff ? 0 : bi;
    tok[row] = id;
    if (id == eos) done[row] = 1;
    if (out && t < max_len) out[t * B + row] = id;
    // the last workgroup to arrive advances the position: by then every workgroup has read state[0]
    if (atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
      *ticket = 0;
      state[0] = t + 1;
    }
  }
}

}  // namespace

extern "C" int asr_dec_gemm(const void* W, int64_t ldw, const float* bias, void* out, int64_t ldo, int B, int N, int K, int relu,
                            int out_dtype, int prologue, const void* X, int64_t ldx, const void* Y, const void* R, const float* gamma,
                            const float* beta, float eps, void* x_out, const int64_t* tok, const float* table, const float* pe,
                            float scale, const int64_t* state, hipStream_t s) {
  ASR_CHECK_ARG(W && out && B >= 0 && N > 0 && K > 0 && prologue >= 0 && prologue <= 2);
  ASR_CHECK_ARG(out_dtype == ASR_F32 || out_dtype == ASR_BF16);
  if (B > 32 || K % 64 != 0 || ldw % 8 != 0 || !aligned16(W) || ldo % 4 != 0 || !aligned16(out)) return ASR_EUNSUPPORTED;
  if (prologue != 0 && K > 512) return ASR_EUNSUPPORTED;
  if (prologue == 0) ASR_CHECK_ARG(X && ldx >= K);
  if (prologue == 0 && (ldx % 8 != 0 || !aligned16(X))) return ASR_EUNSUPPORTED;
  if (prologue == 1) ASR_CHECK_ARG(Y && R && gamma && beta && aligned16(Y) && aligned16(R) && aligned16(gamma) && aligned16(beta));
  if (prologue == 2) ASR_CHECK_ARG(tok && table && pe && state && aligned16(table) && aligned16(pe));
  if (x_out) ASR_CHECK_ARG(aligned16(x_out));
  if (B == 0) return ASR_OK;
  DecGemmArgs p{};
  p.W = static_cast<const bf16_t*>(W); p.bias = bias; p.out = out; p.ldw = ldw; p.ldo = ldo;
  p.B = B; p.N = N; p.K = K; p.relu = relu;
  p.X = static_cast<const bf16_t*>(X); p.ldx = ldx;
  p.Y = static_cast<const bf16_t*>(Y); p.R = static_cast<const bf16_t*>(R); p.gamma = gamma; p.beta = beta; p.eps = eps;
  p.x_out = static_cast<bf16_t*>(x_out);
  p.tok = tok; p.table = table; p.pe = pe; p.scale = scale; p.state = state;
  const unsigned grid = (unsigned)((N + 31) / 32);
  const size_t lds = 3 * 16 * 64 * 4 + (prologue ? (size_t)32 * (K * 2 + 16) : 0);
  const bool f32 = out_dtype == ASR_F32;
  const bool deep = (K >> 6) > 8;                        // more than 8 K steps per wave: 16 loads per operand in flight
  AsrProfScope prof(ASR_OP_GEMM, s);
#define ASR_DEC_LAUNCH(PRO_, TO_, GS_) hipLaunchKernelGGL((dec_gemm_kernel<PRO_, TO_, GS_>), dim3(grid), dim3(256), lds, s, p)
  if (prologue == 0) {
    if (f32) { if (deep) ASR_DEC_LAUNCH(0, float, 16); else ASR_DEC_LAUNCH(0, float, 8); }
    else { if (deep) ASR_DEC_LAUNCH(0, bf16_t, 16); else ASR_DEC_LAUNCH(0, bf16_t, 8); }
  } else if (prologue == 1) {
    if (f32) ASR_DEC_LAUNCH(1, float, 8); else ASR_DEC_LAUNCH(1, bf16_t, 8);
  } else {
    if (f32) ASR_DEC_LAUNCH(2, float, 8); else ASR_DEC_LAUNCH(2, bf16_t, 8);
  }
#undef ASR_DEC_LAUNCH
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_dec_attn(const void* q, int64_t ldq, const void* k_new, const void* v_new, int64_t ld_new, void* k_cache,
                            void* v_cache, int64_t cache_batch_stride, int64_t cache_row_stride, int rows, void* out, int64_t ldo,
                            int B, int H, int dk, float scale, const int64_t* state, hipStream_t s) {
  ASR_CHECK_ARG(q && k_cache && v_cache && out && B >= 0 && H > 0 && rows > 0);
  ASR_CHECK_ARG((k_new == nullptr) == (v_new == nullptr));
  ASR_CHECK_ARG(!k_new || state);                        // an appended row needs its position
  if (dk != 64 || rows > DEC_NI * 64) return ASR_EUNSUPPORTED;
  if (ldq % 8 != 0 || ldo % 8 != 0 || cache_row_stride % 8 != 0 || cache_batch_stride % 8 != 0 || (k_new && ld_new % 8 != 0) ||
      !aligned16(q) || !aligned16(out) || !aligned16(k_cache) || !aligned16(v_cache) || (k_new && (!aligned16(k_new) || !aligned16(v_new))))
    return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  DecAttnArgs p{};
  p.q = static_cast<const bf16_t*>(q); p.ldq = ldq;
  p.kn = static_cast<const bf16_t*>(k_new); p.vn = static_cast<const bf16_t*>(v_new); p.ldn = ld_new;
  p.kc = static_cast<bf16_t*>(k_cache); p.vc = static_cast<bf16_t*>(v_cache); p.cbs = cache_batch_stride; p.cld = cache_row_stride;
  p.rows = rows; p.out = static_cast<bf16_t*>(out); p.ldo = ldo; p.B = B; p.H = H; p.scale = scale; p.state = state;
  AsrProfScope prof(ASR_OP_ATTN_FWD, s);
  hipLaunchKernelGGL(dec_attn_kernel, dim3((unsigned)((B * H + 3) / 4)), dim3(256), 0, s, p);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_dec_finish(const float* logits, int64_t ld, int V, int64_t* tok, uint8_t* done, int64_t* out, int B, int max_len,
                              int eos, int64_t* state, int32_t* ticket, hipStream_t s) {
  ASR_CHECK_ARG(logits && tok && done && state && ticket && B >= 0 && V > 0 && ld >= V && max_len > 0);
  if (B == 0) return ASR_OK;
  hipLaunchKernelGGL(dec_finish_kernel, dim3((unsigned)B), dim3(256), 0, s, logits, ld, V, tok, done, out, B, max_len, eos, state, ticket);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
