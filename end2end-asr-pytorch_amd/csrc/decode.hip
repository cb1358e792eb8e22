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
  int w_frag, x_frag, out_frag;                                   // fragment-major operands (see asr_hip.h)
  const bf16_t* X; int64_t ldx;                                   // prologue 0: the input rows
  const bf16_t* Y; const bf16_t* R; const float* gamma; const float* beta; float eps;   // prologue 1: x = LN(Y + R) gamma + beta
  bf16_t* x_out;                                                   // prologue 1 / 2: workgroup 0 stores x (B, K)
  const int64_t* tok; const float* table; const float* pe; float scale; const int64_t* state;   // prologue 2
};

// PRO 0: x read from memory; 1: LayerNorm(Y + R); 2: embedding row * scale + pe[t].  NW waves split K (4, or 8 when K > 512);
// GS = K steps per wave whose loads are issued together (all of them).
template <int PRO, typename TO, int NW, int GS>
__global__ __launch_bounds__(NW * 64) void dec_gemm_kernel(DecGemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, half = lane >> 5;
  const int n0 = blockIdx.x * 32;
  const int K = p.K;
  const int XP = K * 2 + 16;                                       // LDS row pitch of the x tile: rows 4 banks apart
  float* red = reinterpret_cast<float*>(smem);                     // [NW - 1][16][64] partial accumulators of waves 1..
  unsigned char* xs = smem + (NW - 1) * 16 * 64 * 4;               // PRO != 0: x tile [32][XP]
  float* gb = red;                                                 // PRO 1: gamma | beta [2][K] fp32, dead before `red` is written

  // ---- the weight fragments do not depend on the prologue: their loads go out first and fly while x is produced
  const int kper = K / NW;                                         // k range of a wave (a multiple of 16)
  const int kw = wave * kper;
  const int ksteps = kper >> 4;
  const int nrow = n0 + lr < p.N ? n0 + lr : p.N - 1;
  // row-major: lane (row lr, k half) reads 16 bytes of its row per step; fragment-major: the 64 lanes of a step read 1 KB in a row
  const bf16_t* wp = p.w_frag ? p.W + (((int64_t)blockIdx.x * (K >> 4) + (kw >> 4)) * 64 + lane) * 8
                              : p.W + (int64_t)nrow * p.ldw + kw + 8 * half;
  const int wstep = p.w_frag ? 512 : 16;
  uint4 a[GS], b[GS];
#pragma unroll
  for (int i = 0; i < GS; ++i)
    if (i < ksteps) a[i] = *reinterpret_cast<const uint4*>(wp + i * wstep);

  if (PRO != 0) {
    // thread = (row m = tid / 8, sub = tid % 8): the row's 16-byte chunks sub, sub + 8, ... (8 lanes read 128 contiguous bytes);
    // statistics over the 8 lanes of a row by three xor shuffles -- all 32 rows at once, one memory round trip
    static_assert(PRO == 0 || NW == 4, "prologues: 256 threads = 32 rows x 8 lanes");
    const int m = tid >> 3, sub = tid & 7;
    const int nch = K >> 3;                                        // chunks per row (<= 64)
    const int mm = m < p.B ? m : p.B - 1;
    Chunk<bf16_t> z[8];
    if (PRO == 1) {
      Chunk<bf16_t> cy[8], cr[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = j * 8 + sub, cc = c < nch ? c : 0;
        cy[j].v = *reinterpret_cast<const uint4*>(p.Y + (int64_t)mm * K + cc * 8);
        cr[j].v = *reinterpret_cast<const uint4*>(p.R + (int64_t)mm * K + cc * 8);
      }
      for (int c = tid; c < K / 4; c += 256) {                     // gamma | beta -> LDS (read after the barrier below)
        *reinterpret_cast<float4*>(gb + c * 4) = *reinterpret_cast<const float4*>(p.gamma + c * 4);
        *reinterpret_cast<float4*>(gb + K + c * 4) = *reinterpret_cast<const float4*>(p.beta + c * 4);
      }
      // z = y + residual rounded to the storage type first (what asr_add_ln_fwd stores and normalises), fp32 statistics
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool live = j * 8 + sub < nch;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          z[j].e[e] = f32_to_bf16(bf16_to_f32(cy[j].e[e]) + bf16_to_f32(cr[j].e[e]));
          s += live ? bf16_to_f32(z[j].e[e]) : 0.f;
        }
      }
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      const float mu = s / (float)K;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool live = j * 8 + sub < nch;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = live ? bf16_to_f32(z[j].e[e]) - mu : 0.f; q += d * d; }
      }
      q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
      const float rs = rsqrtf(q / (float)K + p.eps);
      __syncthreads();                                             // gamma / beta are in LDS
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = j * 8 + sub;
        if (c < nch) {
          Chunk<bf16_t> o;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            o.e[e] = f32_to_bf16((bf16_to_f32(z[j].e[e]) - mu) * rs * gb[c * 8 + e] + gb[K + c * 8 + e]);
          if (m >= p.B) o.v = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(xs + m * XP + c * 16) = o.v;
          if (blockIdx.x == 0 && p.x_out && m < p.B) *reinterpret_cast<uint4*>(p.x_out + (int64_t)m * K + c * 8) = o.v;
        }
      }
    } else {
      const int64_t t = p.state[0];
      const float* erow = p.table + p.tok[mm] * (int64_t)K;
      const float* prow = p.pe + t * K;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = j * 8 + sub;
        if (c < nch) {
          const float4 e0 = *reinterpret_cast<const float4*>(erow + c * 8), e1 = *reinterpret_cast<const float4*>(erow + c * 8 + 4);
          const float4 q0 = *reinterpret_cast<const float4*>(prow + c * 8), q1 = *reinterpret_cast<const float4*>(prow + c * 8 + 4);
          Chunk<bf16_t> o;
          o.e[0] = f32_to_bf16(e0.x * p.scale + q0.x); o.e[1] = f32_to_bf16(e0.y * p.scale + q0.y);
          o.e[2] = f32_to_bf16(e0.z * p.scale + q0.z); o.e[3] = f32_to_bf16(e0.w * p.scale + q0.w);
          o.e[4] = f32_to_bf16(e1.x * p.scale + q1.x); o.e[5] = f32_to_bf16(e1.y * p.scale + q1.y);
          o.e[6] = f32_to_bf16(e1.z * p.scale + q1.z); o.e[7] = f32_to_bf16(e1.w * p.scale + q1.w);
          if (m >= p.B) o.v = make_uint4(0u, 0u, 0u, 0u);
          *reinterpret_cast<uint4*>(xs + m * XP + c * 16) = o.v;
          if (blockIdx.x == 0 && p.x_out && m < p.B) *reinterpret_cast<uint4*>(p.x_out + (int64_t)m * K + c * 8) = o.v;
        }
      }
    }
    __syncthreads();
  }

  // ---- K loop: wave w contracts its k range; A = 32 weight rows (output columns), B = the 32 input rows
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bf16_t* xp = PRO != 0 ? nullptr
                     : (p.x_frag ? p.X + ((int64_t)(kw >> 4) * 64 + lane) * 8 : p.X + (int64_t)(lr < p.B ? lr : p.B - 1) * p.ldx + kw + 8 * half);
  const int xstep = p.x_frag ? 512 : 16;
  const unsigned char* xl = xs + lr * XP + (kw + 8 * half) * 2;
#pragma unroll
  for (int i = 0; i < GS; ++i)
    if (i < ksteps) {
      if (PRO == 0) b[i] = *reinterpret_cast<const uint4*>(xp + i * xstep);
      else b[i] = *reinterpret_cast<const uint4*>(xl + i * 32);
    }
#pragma unroll
  for (int i = 0; i < GS; ++i)
    if (i < ksteps)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]), __builtin_bit_cast(bf16x8_t, b[i]), acc, 0, 0, 0);

  // ---- the waves' partial tiles meet in LDS (fragment layout: lane-private slots)
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < NW - 1; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[(w * 16 + r) * 64 + lane];
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
    if constexpr (sizeof(TO) == 2) {
      if (p.out_frag) {                 // chunk n / 8 of row m in the layout the next GEMM reads: ((chunk/2) 64 + (chunk%2) 32 + m) 8
        const int ch = n >> 3;
        uint2 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        if (n + 3 < p.N)
          *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.out) + ((int64_t)(ch >> 1) * 64 + (ch & 1) * 32 + m) * 8 + (n & 7)) = o;
        continue;
      }
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
  int B, H; float scale; int out_frag;
  const int64_t* state;                                  // self attention: state[0] = position t (keys 0..t); null: all `rows` keys
};

constexpr int DEC_MAX_KEYS = 512;                        // keys per pass of dec_attn_kernel; the fused kernel's limit

// One workgroup per (sequence, head).  thread = (16-byte channel chunk c = tid % 8, key group jg = tid / 8): keys jg, jg + 32, ...
// -- the 8 lanes of a key read its 128-byte row as ONE contiguous access, for the scores (partial dot products over 8 channels,
// three xor shuffles) and for P V alike; all <= 16 + 16 loads of a thread are issued before the first use (the values do not
// depend on the softmax), the softmax statistics go through LDS.  More than 512 keys (encoder outputs of 795 frames at the
// Librispeech shape): passes of 512 keys with a running maximum -- sum and accumulators are rescaled when it moves.
__global__ __launch_bounds__(256) void dec_attn_kernel(DecAttnArgs p) {
  constexpr int NI = DEC_MAX_KEYS / 32;
  __shared__ float wmax[2][4];
  __shared__ float wsum[4];
  __shared__ float ored[4][8][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
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
  if (tsel >= 0 && tid < 16) {                           // append: row t of both caches (read below from the source rows)
    const int c = (tid & 7) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>((tid < 8 ? kn : vn) + c);
    *reinterpret_cast<uint4*>((tid < 8 ? p.kc : p.vc) + b * p.cbs + (int64_t)t * p.cld + h * 64 + c) = v;
  }
  const int c = (tid & 7) * 8, jg = tid >> 3;
  Chunk<bf16_t> kq;
  kq.v = *reinterpret_cast<const uint4*>(p.q + (int64_t)b * p.ldq + h * 64 + c);
  float qf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = bf16_to_f32(kq.e[e]);
  float run = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  int par = 0;
#pragma unroll 1
  for (int j0 = 0; j0 < L; j0 += DEC_MAX_KEYS, par ^= 1) {
    const int n = L - j0;                                // keys of this pass (uniform), >= 1
    Chunk<bf16_t> kk[NI], vv[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * 32 < n) {                                  // uniform
        const int j = j0 + jg + 32 * i, jj = j < L ? j : L - 1;
        kk[i].v = *reinterpret_cast<const uint4*>((jj == tsel ? kn : kc + (int64_t)jj * p.cld) + c);
      }
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * 32 < n) {
        const int j = j0 + jg + 32 * i, jj = j < L ? j : L - 1;
        vv[i].v = *reinterpret_cast<const uint4*>((jj == tsel ? vn : vc + (int64_t)jj * p.cld) + c);
      }
    // ---- scores
    float sc[NI];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      sc[i] = -INFINITY;
      if (i * 32 < n) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qf[e] * bf16_to_f32(kk[i].e[e]);
        d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
        if (j0 + jg + 32 * i < L) sc[i] = d * p.scale;
        mx = fmaxf(mx, sc[i]);
      }
    }
    mx = wave_max(mx);
    if (lane == 0) wmax[par][wave] = mx;
    __syncthreads();                                     // (the other parity is written next pass: no second barrier needed)
    mx = fmaxf(fmaxf(fmaxf(wmax[par][0], wmax[par][1]), fmaxf(wmax[par][2], wmax[par][3])), run);
    if (run != -INFINITY && mx != run) {                 // uniform: the running maximum moved
      const float corr = __expf(run - mx);
      l *= corr;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= corr;
    }
    run = mx;
    // ---- probabilities (every lane of a key group holds the same value) and P V
#pragma unroll
    for (int i = 0; i < NI; ++i)
      if (i * 32 < n) {
        const float pr = sc[i] == -INFINITY ? 0.f : __expf(sc[i] - mx);
        l += pr;
        const float pj = bf16_to_f32(f32_to_bf16(pr));   // the probabilities enter P V in the storage type
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pj * bf16_to_f32(vv[i].e[e]);
      }
  }
  // sums over the 8 key groups of a wave (lane bits 3..5), then over the 4 waves through LDS; l is replicated 8 x per key
  l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int o = 8; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ored[wave][lane][e] = acc[e];
    if (lane == 0) wsum[wave] = l;
  }
  __syncthreads();
  if (tid < 8) {
    const float lt = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    Chunk<bf16_t> o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o.e[e] = f32_to_bf16((ored[0][tid][e] + ored[1][tid][e] + ored[2][tid][e] + ored[3][tid][e]) * inv);
    const int ch = h * 8 + tid;
    if (p.out_frag) *reinterpret_cast<uint4*>(p.out + ((int64_t)(ch >> 1) * 64 + (ch & 1) * 32 + b) * 8) = o.v;
    else *reinterpret_cast<uint4*>(p.out + (int64_t)b * p.ldo + h * 64 + tid * 8) = o.v;
  }
}

// ------------------------------------------------------------------------------------------------ projections + attention
// The sub-layer input row AND this head's projections inside the attention launch: workgroup (sequence b, head h), 512 threads,
//   1. every load is issued first: the head's 64 (cross: query) or 192 (self: query, key, value) weight rows -- thread (output o =
//      tid / 8, k eighth tid % 8) reads 128 consecutive bytes per row, 8 lanes a whole 1 KB row --, the cached keys / values (thread =
//      (16-byte chunk, key group of 64)), and the two rows of the prologue;
//   2. wave 0: x = LayerNorm(Y[b] + R[b]) (or embedding[tok[b]] scale + pe[t]) -> LDS, head 0 stores it for the next residual;
//   3. 64 / 192 dot products of length D from registers x LDS, 8-lane xor reduction, bias, rounded to bf16 like the GEMM's output;
//   4. self attention: key / value row t -> cache, and taken from LDS for the scores; then softmax and P V as in dec_attn_kernel.
// Replaces two launches (the LayerNorm-prologue GEMM and the attention) by one: 34 -> 26 launches per token.
struct DecFusedArgs {
  const bf16_t* Y; const bf16_t* R; const float* gamma; const float* beta; float eps;
  const int64_t* tok; const float* table; const float* pe; float emb_scale;
  bf16_t* x_out;
  const bf16_t* W; const float* bias; int D;
  bf16_t* kc; bf16_t* vc; int64_t cbs, cld; int rows;
  bf16_t* out; int64_t ldo; int B, H; float scale; int out_frag;
  const int64_t* state;
};

template <bool SELF, bool EMBED>
__global__ __launch_bounds__(512) void dec_attn_fused_kernel(DecFusedArgs p) {
  constexpr int NP = SELF ? 2 : 1;                       // projections of the first pass (the value rows' weights are loaded
  constexpr int NI = DEC_MAX_KEYS / 64;                  // once the query / key dot products have freed their registers)
  __shared__ float xs[512];                              // the sub-layer input row (bf16 values as fp32)
  __shared__ float pr[SELF ? 3 : 1][64];                 // this head's query (, key, value) row
  __shared__ float wred[2][8];
  __shared__ float ored[8][8][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
  const int D = p.D, HD = p.H * 64;
  int t = -1, L = p.rows;
  if (SELF || EMBED) {
    const int64_t ts = p.state[0];
    if (SELF && ts < p.rows) { t = (int)ts; L = t + 1; }
  }
  // ---- 1. loads
  const int o = tid >> 3, ke = tid & 7;
  const int nck = D >> 6;                                // 16-byte chunks of a weight row per thread (<= 8)
  Chunk<bf16_t> wq[NP][8];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const bf16_t* wr = p.W + ((int64_t)j * HD + h * 64 + o) * D + ke * 8;      // chunks ke, ke + 8, ...: 8 lanes read 128 consecutive bytes
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nck) wq[j][c].v = *reinterpret_cast<const uint4*>(wr + c * 64);
  }
  const bf16_t* kc = p.kc + b * p.cbs + h * 64;
  const bf16_t* vc = p.vc + b * p.cbs + h * 64;
  const int c8 = (tid & 7) * 8, jg = tid >> 3;
  Chunk<bf16_t> kk[NI], vv[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (i * 64 < L) {                                    // uniform
      const int j = jg + 64 * i, jj = j < L ? j : L - 1;
      kk[i].v = *reinterpret_cast<const uint4*>(kc + (int64_t)jj * p.cld + c8);
    }
  // ---- 2. the input row (wave 0; D <= 512: one 16-byte chunk per lane)
  if (wave == 0) {
    const int c0 = lane * 8;
    const bool live = c0 < D;
    const int cc = live ? c0 : 0;
    Chunk<bf16_t> xo;
    if (EMBED) {
      const int64_t tt = p.state[0];
      const float* e = p.table + p.tok[b] * (int64_t)D + cc;
      const float* q = p.pe + tt * D + cc;
      const float4 e0 = *reinterpret_cast<const float4*>(e), e1 = *reinterpret_cast<const float4*>(e + 4);
      const float4 q0 = *reinterpret_cast<const float4*>(q), q1 = *reinterpret_cast<const float4*>(q + 4);
      xo.e[0] = f32_to_bf16(e0.x * p.emb_scale + q0.x); xo.e[1] = f32_to_bf16(e0.y * p.emb_scale + q0.y);
      xo.e[2] = f32_to_bf16(e0.z * p.emb_scale + q0.z); xo.e[3] = f32_to_bf16(e0.w * p.emb_scale + q0.w);
      xo.e[4] = f32_to_bf16(e1.x * p.emb_scale + q1.x); xo.e[5] = f32_to_bf16(e1.y * p.emb_scale + q1.y);
      xo.e[6] = f32_to_bf16(e1.z * p.emb_scale + q1.z); xo.e[7] = f32_to_bf16(e1.w * p.emb_scale + q1.w);
    } else {
      Chunk<bf16_t> cy, cr;
      cy.v = *reinterpret_cast<const uint4*>(p.Y + (int64_t)b * D + cc);
      cr.v = *reinterpret_cast<const uint4*>(p.R + (int64_t)b * D + cc);
      float gm[8], bt[8];
#pragma unroll
      for (int j = 0; j < 8; j += 4) {
        const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + cc + j), b4 = *reinterpret_cast<const float4*>(p.beta + cc + j);
        gm[j] = g4.x; gm[j + 1] = g4.y; gm[j + 2] = g4.z; gm[j + 3] = g4.w;
        bt[j] = b4.x; bt[j + 1] = b4.y; bt[j + 2] = b4.z; bt[j + 3] = b4.w;
      }
      float z[8], sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        z[e] = live ? bf16_to_f32(f32_to_bf16(bf16_to_f32(cy.e[e]) + bf16_to_f32(cr.e[e]))) : 0.f;
        sum += z[e];
      }
      const float mu = wave_sum(sum) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = live ? z[e] - mu : 0.f; q += d * d; }
      const float rs = rsqrtf(wave_sum(q) / (float)D + p.eps);
#pragma unroll
      for (int e = 0; e < 8; ++e) xo.e[e] = f32_to_bf16((z[e] - mu) * rs * gm[e] + bt[e]);
    }
    if (live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) xs[c0 + e] = bf16_to_f32(xo.e[e]);
      if (h == 0 && p.x_out) *reinterpret_cast<uint4*>(p.x_out + (int64_t)b * D + c0) = xo.v;
    }
  }
  __syncthreads();
  // ---- 3. projections
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nck) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d += bf16_to_f32(wq[j][c].e[e]) * xs[c * 64 + ke * 8 + e];
      }
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
    if (ke == 0) pr[j][o] = bf16_to_f32(f32_to_bf16(d + (p.bias ? p.bias[j * HD + h * 64 + o] : 0.f)));
  }
  // the cached values (and the value rows' weights) go out now, into the registers the first-pass weights have freed, and land
  // under the scores and the softmax
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (i * 64 < L) {
      const int j = jg + 64 * i, jj = j < L ? j : L - 1;
      vv[i].v = *reinterpret_cast<const uint4*>(vc + (int64_t)jj * p.cld + c8);
    }
  Chunk<bf16_t> wv[8];
  if (SELF) {
    const bf16_t* wr = p.W + ((int64_t)2 * HD + h * 64 + o) * D + ke * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nck) wv[c].v = *reinterpret_cast<const uint4*>(wr + c * 64);
  }
  __syncthreads();
  // ---- 4. attention
  if (SELF && t >= 0 && tid < 8) {                       // this position's key row -> cache
    const int c = tid * 8;
    Chunk<bf16_t> ch;
#pragma unroll
    for (int e = 0; e < 8; ++e) ch.e[e] = f32_to_bf16(pr[1][c + e]);
    *reinterpret_cast<uint4*>(p.kc + b * p.cbs + (int64_t)t * p.cld + h * 64 + c) = ch.v;
  }
  float qf[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] = pr[0][c8 + e];
  float sc[NI];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    sc[i] = -INFINITY;
    if (i * 64 < L) {
      const int j = jg + 64 * i;
      float d = 0.f;
      if (SELF && j == t) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qf[e] * pr[1][c8 + e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) d += qf[e] * bf16_to_f32(kk[i].e[e]);
      }
      d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
      if (j < L) sc[i] = d * p.scale;
      mx = fmaxf(mx, sc[i]);
    }
  }
  mx = wave_max(mx);
  if (lane == 0) wred[0][wave] = mx;
  if (SELF) {                                            // value projection, then its row -> LDS and cache
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c)
      if (c < nck) {
#pragma unroll
        for (int e = 0; e < 8; ++e) d += bf16_to_f32(wv[c].e[e]) * xs[c * 64 + ke * 8 + e];
      }
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64);
    if (ke == 0) pr[2][o] = bf16_to_f32(f32_to_bf16(d + (p.bias ? p.bias[2 * HD + h * 64 + o] : 0.f)));
  }
  __syncthreads();
  if (SELF && t >= 0 && tid < 8) {
    const int c = tid * 8;
    Chunk<bf16_t> ch;
#pragma unroll
    for (int e = 0; e < 8; ++e) ch.e[e] = f32_to_bf16(pr[2][c + e]);
    *reinterpret_cast<uint4*>(p.vc + b * p.cbs + (int64_t)t * p.cld + h * 64 + c) = ch.v;
  }
  mx = wred[0][0];
#pragma unroll
  for (int w = 1; w < 8; ++w) mx = fmaxf(mx, wred[0][w]);
  float l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i)
    if (i * 64 < L) {
      const int j = jg + 64 * i;
      const float prb = sc[i] == -INFINITY ? 0.f : __expf(sc[i] - mx);
      l += prb;
      const float pj = bf16_to_f32(f32_to_bf16(prb));
      if (SELF && j == t) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pj * pr[2][c8 + e];
      } else if (j < L) {                                // (the clamped row of a lane past L may be the not yet written row t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += pj * bf16_to_f32(vv[i].e[e]);
      }
    }
  l += __shfl_xor(l, 8, 64); l += __shfl_xor(l, 16, 64); l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int of = 8; of < 64; of <<= 1)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += __shfl_xor(acc[e], of, 64);
  if (lane < 8) {
#pragma unroll
    for (int e = 0; e < 8; ++e) ored[wave][lane][e] = acc[e];
    if (lane == 0) wred[1][wave] = l;
  }
  __syncthreads();
  if (tid < 8) {
    float lt = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) lt += wred[1][w];
    const float inv = lt > 0.f ? 1.f / lt : 0.f;
    Chunk<bf16_t> oc;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += ored[w][tid][e];
      oc.e[e] = f32_to_bf16(a * inv);
    }
    const int ch = h * 8 + tid;
    if (p.out_frag) *reinterpret_cast<uint4*>(p.out + ((int64_t)(ch >> 1) * 64 + (ch & 1) * 32 + b) * 8) = oc.v;
    else *reinterpret_cast<uint4*>(p.out + (int64_t)b * p.ldo + h * 64 + tid * 8) = oc.v;
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
                            int out_dtype, int layout, int prologue, const void* X, int64_t ldx, const void* Y, const void* R, const float* gamma,
                            const float* beta, float eps, void* x_out, const int64_t* tok, const float* table, const float* pe,
                            float scale, const int64_t* state, hipStream_t s) {
  ASR_CHECK_ARG(W && out && B >= 0 && N > 0 && K > 0 && prologue >= 0 && prologue <= 2);
  ASR_CHECK_ARG(out_dtype == ASR_F32 || out_dtype == ASR_BF16);
  ASR_CHECK_ARG(layout >= 0 && layout < 8 && !((layout & ASR_DEC_X_FRAG) && prologue != 0) && !((layout & ASR_DEC_OUT_FRAG) && out_dtype != ASR_BF16));
  if (B > 32 || K % 64 != 0 || ldw % 8 != 0 || !aligned16(W) || ldo % 4 != 0 || !aligned16(out)) return ASR_EUNSUPPORTED;
  if ((layout & ASR_DEC_OUT_FRAG) && N % 16 != 0) return ASR_EUNSUPPORTED;
  if (prologue != 0 && K > 512) return ASR_EUNSUPPORTED;
  if (prologue == 0) ASR_CHECK_ARG(X && ldx >= K);
  if (prologue == 0 && !(layout & ASR_DEC_X_FRAG) && ldx % 8 != 0) return ASR_EUNSUPPORTED;
  if (prologue == 0 && !aligned16(X)) return ASR_EUNSUPPORTED;
  if (prologue == 1) ASR_CHECK_ARG(Y && R && gamma && beta && aligned16(Y) && aligned16(R) && aligned16(gamma) && aligned16(beta));
  if (prologue == 2) ASR_CHECK_ARG(tok && table && pe && state && aligned16(table) && aligned16(pe));
  if (x_out) ASR_CHECK_ARG(aligned16(x_out));
  if (B == 0) return ASR_OK;
  DecGemmArgs p{};
  p.W = static_cast<const bf16_t*>(W); p.bias = bias; p.out = out; p.ldw = ldw; p.ldo = ldo;
  p.B = B; p.N = N; p.K = K; p.relu = relu;
  p.w_frag = (layout & ASR_DEC_W_FRAG) != 0; p.x_frag = (layout & ASR_DEC_X_FRAG) != 0; p.out_frag = (layout & ASR_DEC_OUT_FRAG) != 0;
  p.X = static_cast<const bf16_t*>(X); p.ldx = ldx;
  p.Y = static_cast<const bf16_t*>(Y); p.R = static_cast<const bf16_t*>(R); p.gamma = gamma; p.beta = beta; p.eps = eps;
  p.x_out = static_cast<bf16_t*>(x_out);
  p.tok = tok; p.table = table; p.pe = pe; p.scale = scale; p.state = state;
  const unsigned grid = (unsigned)((N + 31) / 32);
  const bool f32 = out_dtype == ASR_F32;
  const int nw = (prologue == 0 && K > 512 && K % 128 == 0) ? 8 : 4;            // waves that split K
  const int ksteps = K / nw / 16;
  if (ksteps > 16) return ASR_EUNSUPPORTED;                                  // K <= 1024 (4 waves) / 2048 (8 waves)
  const size_t lds = (size_t)(nw - 1) * 16 * 64 * 4 + (prologue ? (size_t)32 * (K * 2 + 16) : 0);
  AsrProfScope prof(ASR_OP_GEMM, s);
#define ASR_DEC_LAUNCH(PRO_, TO_, NW_, GS_) hipLaunchKernelGGL((dec_gemm_kernel<PRO_, TO_, NW_, GS_>), dim3(grid), dim3(NW_ * 64), lds, s, p)
  if (prologue == 0) {
    if (nw == 8) { if (f32) ASR_DEC_LAUNCH(0, float, 8, 16); else ASR_DEC_LAUNCH(0, bf16_t, 8, 16); }
    else if (ksteps > 8) { if (f32) ASR_DEC_LAUNCH(0, float, 4, 16); else ASR_DEC_LAUNCH(0, bf16_t, 4, 16); }
    else { if (f32) ASR_DEC_LAUNCH(0, float, 4, 8); else ASR_DEC_LAUNCH(0, bf16_t, 4, 8); }
  } else if (prologue == 1) {
    if (f32) ASR_DEC_LAUNCH(1, float, 4, 8); else ASR_DEC_LAUNCH(1, bf16_t, 4, 8);
  } else {
    if (f32) ASR_DEC_LAUNCH(2, float, 4, 8); else ASR_DEC_LAUNCH(2, bf16_t, 4, 8);
  }
#undef ASR_DEC_LAUNCH
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_dec_attn(const void* q, int64_t ldq, const void* k_new, const void* v_new, int64_t ld_new, void* k_cache,
                            void* v_cache, int64_t cache_batch_stride, int64_t cache_row_stride, int rows, void* out, int64_t ldo,
                            int B, int H, int dk, float scale, int out_frag, const int64_t* state, hipStream_t s) {
  ASR_CHECK_ARG(q && k_cache && v_cache && out && B >= 0 && H > 0 && rows > 0);
  ASR_CHECK_ARG((k_new == nullptr) == (v_new == nullptr));
  ASR_CHECK_ARG(!k_new || state);                        // an appended row needs its position
  if (dk != 64) return ASR_EUNSUPPORTED;
  if (ldq % 8 != 0 || ldo % 8 != 0 || cache_row_stride % 8 != 0 || cache_batch_stride % 8 != 0 || (k_new && ld_new % 8 != 0) ||
      !aligned16(q) || !aligned16(out) || !aligned16(k_cache) || !aligned16(v_cache) || (k_new && (!aligned16(k_new) || !aligned16(v_new))))
    return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  DecAttnArgs p{};
  p.q = static_cast<const bf16_t*>(q); p.ldq = ldq;
  p.kn = static_cast<const bf16_t*>(k_new); p.vn = static_cast<const bf16_t*>(v_new); p.ldn = ld_new;
  p.kc = static_cast<bf16_t*>(k_cache); p.vc = static_cast<bf16_t*>(v_cache); p.cbs = cache_batch_stride; p.cld = cache_row_stride;
  p.rows = rows; p.out = static_cast<bf16_t*>(out); p.ldo = ldo; p.B = B; p.H = H; p.scale = scale; p.state = state;
  p.out_frag = out_frag;
  if (out_frag && B > 32) return ASR_EUNSUPPORTED;
  AsrProfScope prof(ASR_OP_ATTN_FWD, s);
  hipLaunchKernelGGL(dec_attn_kernel, dim3((unsigned)(B * H)), dim3(256), 0, s, p);
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

extern "C" int asr_dec_attn_fused(const void* W, const float* bias, int D, int self_attention, const void* Y, const void* R,
                                  const float* gamma, const float* beta, float eps, const int64_t* tok, const float* table,
                                  const float* pe, float emb_scale, void* x_out, void* k_cache, void* v_cache,
                                  int64_t cache_batch_stride, int64_t cache_row_stride, int rows, void* out, int64_t ldo, int B, int H,
                                  int dk, float scale, int out_frag, const int64_t* state, hipStream_t s) {
  ASR_CHECK_ARG(W && k_cache && v_cache && out && B >= 0 && H > 0 && rows > 0 && D > 0);
  const bool embed = tok != nullptr;
  ASR_CHECK_ARG(embed ? (table && pe && state && self_attention) : (Y && R && gamma && beta));
  ASR_CHECK_ARG(!self_attention || state);
  if (dk != 64 || rows > DEC_MAX_KEYS || D % 64 != 0 || D > 512 || (out_frag && B > 32)) return ASR_EUNSUPPORTED;
  if (ldo % 8 != 0 || cache_row_stride % 8 != 0 || cache_batch_stride % 8 != 0 || !aligned16(W) || !aligned16(out) || !aligned16(k_cache) ||
      !aligned16(v_cache) || (x_out && !aligned16(x_out)) || (!embed && (!aligned16(Y) || !aligned16(R) || !aligned16(gamma) || !aligned16(beta))) ||
      (embed && (!aligned16(table) || !aligned16(pe))))
    return ASR_EUNSUPPORTED;
  if (B == 0) return ASR_OK;
  DecFusedArgs p{};
  p.Y = static_cast<const bf16_t*>(Y); p.R = static_cast<const bf16_t*>(R); p.gamma = gamma; p.beta = beta; p.eps = eps;
  p.tok = tok; p.table = table; p.pe = pe; p.emb_scale = emb_scale; p.x_out = static_cast<bf16_t*>(x_out);
  p.W = static_cast<const bf16_t*>(W); p.bias = bias; p.D = D;
  p.kc = static_cast<bf16_t*>(k_cache); p.vc = static_cast<bf16_t*>(v_cache); p.cbs = cache_batch_stride; p.cld = cache_row_stride;
  p.rows = rows; p.out = static_cast<bf16_t*>(out); p.ldo = ldo; p.B = B; p.H = H; p.scale = scale; p.out_frag = out_frag; p.state = state;
  AsrProfScope prof(ASR_OP_ATTN_FWD, s);
  const dim3 grid((unsigned)(B * H)), block(512);
  if (self_attention) {
    if (embed) hipLaunchKernelGGL((dec_attn_fused_kernel<true, true>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((dec_attn_fused_kernel<true, false>), grid, block, 0, s, p);
  } else {
    hipLaunchKernelGGL((dec_attn_fused_kernel<false, false>), grid, block, 0, s, p);
  }
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
