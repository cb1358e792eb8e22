// CTC loss (reference: utils/metrics.py:133-154 -- F.log_softmax over the vocabulary followed by F.ctc_loss(reduction="mean"),
// blank = 0, zero_infinity = False) and its gradient with respect to the LOGITS, log-softmax included.
//
//   lp_t(v)      = logits[b,t,v] - lse[b,t]
//   alpha_t(s)   = lp_t(l'_s) + logsumexp(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if l'_s != blank and l'_s != l'_{s-2}])
//   beta_t(s)    = lp_t(l'_s) + logsumexp(beta_{t+1}(s), beta_{t+1}(s+1), [beta_{t+1}(s+2) if l'_s != blank and l'_s != l'_{s+2}])
//   nll_b        = -logsumexp(alpha_{T_b-1}(S_b-1), alpha_{T_b-1}(S_b-2)),   l' = blank-interleaved target, S_b = 2 len_b + 1
//   loss         = mean_b(nll_b / max(len_b, 1))
//   d nll / d logits[b,t,v] = softmax_t(v) - sum_{s : l'_s = v} exp(alpha_t(s) + beta_t(s) - lp_t(v) + nll_b)      (t < T_b, else 0)
//
// The recursions are sequential in t: one workgroup per utterance walks time with the previous row in LDS (the whole lattice
// goes to a caller-owned workspace for the backward pass); everything else is row parallel.  Path probabilities through (t, s)
// never exceed the total p(l|x), so exp(alpha + beta - lp + nll) lies in [0, 1]: the per-label sums of the gradient are
// accumulated in linear space with LDS atomics, no second max pass.
#include "common.h"

namespace {

constexpr float NEG_INF = -INFINITY;

__device__ __forceinline__ float lae2(float a, float b) {      // log(exp(a) + exp(b)), -inf safe
  const float m = fmaxf(a, b);
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m));
}
__device__ __forceinline__ float lae3(float a, float b, float c) {
  const float m = fmaxf(fmaxf(a, b), c);
  if (m == NEG_INF) return NEG_INF;
  return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}

// one wave per row: lse[row] = log sum_v exp(logits[row, v])
__global__ __launch_bounds__(256) void ctc_lse_kernel(const float* __restrict__ logits, int64_t ld, int64_t rows, int V,
                                                      float* __restrict__ lse) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* x = logits + row * ld;
  float m = NEG_INF;
  for (int v = lane; v < V; v += 64) m = fmaxf(m, x[v]);
  m = wave_max(m);
  float s = 0.f;
  for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
  s = wave_sum(s);
  if (lane == 0) lse[row] = m + logf(s);
}

// grid (B, 2): y = 0 the alpha recursion, y = 1 the beta recursion.  lat (B, T, S) fp32.
__global__ __launch_bounds__(256) void ctc_lattice_kernel(const float* __restrict__ logits, int64_t ld, const float* __restrict__ lse,
                                                          const int64_t* __restrict__ targets, int Lmax,
                                                          const int32_t* __restrict__ in_len, const int32_t* __restrict__ tg_len,
                                                          int T, int S, int blank, float* __restrict__ alpha,
                                                          float* __restrict__ beta, float* __restrict__ nll) {
  extern __shared__ float sh[];       // [2][S] previous / current row, then [S] labels as int
  float* row0 = sh;
  float* row1 = sh + S;
  int* lab = reinterpret_cast<int*>(sh + 2 * S);
  const int b = blockIdx.x;
  const bool backward = blockIdx.y == 1;
  int Tb = in_len[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
  int Lb = tg_len[b];
  Lb = Lb < 0 ? 0 : (Lb > Lmax ? Lmax : Lb);
  const int Sb = 2 * Lb + 1;
  for (int s = threadIdx.x; s < S; s += 256) lab[s] = (s & 1) ? (int)targets[(int64_t)b * Lmax + (s >> 1 < Lmax ? s >> 1 : Lmax - 1)] : blank;
  __syncthreads();
  float* lat = (backward ? beta : alpha) + (int64_t)b * T * S;
  if (Tb == 0) {
    if (!backward && threadIdx.x == 0) nll[b] = Lb == 0 ? 0.f : INFINITY;      // empty input: only the empty target is reachable
    return;
  }
  const float* lg = logits + (int64_t)b * T * ld;
  const float* ls = lse + (int64_t)b * T;
  float* prev = row0;
  float* cur = row1;
  for (int step = 0; step < Tb; ++step) {
    const int t = backward ? Tb - 1 - step : step;
    const float l_t = ls[t];
    for (int s = threadIdx.x; s < S; s += 256) {
      float v = NEG_INF;
      if (s < Sb) {
        const float lp = lg[(int64_t)t * ld + lab[s]] - l_t;
        if (step == 0) {
          const bool start = backward ? (s == Sb - 1 || s == Sb - 2) : (s == 0 || s == 1);
          v = start ? lp : NEG_INF;
        } else if (!backward) {
          const float a0 = prev[s], a1 = s >= 1 ? prev[s - 1] : NEG_INF;
          const float a2 = (s >= 2 && lab[s] != blank && lab[s] != lab[s - 2]) ? prev[s - 2] : NEG_INF;
          v = lae3(a0, a1, a2) + lp;
        } else {
          const float a0 = prev[s], a1 = s + 1 < Sb ? prev[s + 1] : NEG_INF;
          const float a2 = (s + 2 < Sb && lab[s] != blank && lab[s] != lab[s + 2]) ? prev[s + 2] : NEG_INF;
          v = lae3(a0, a1, a2) + lp;
        }
      }
      cur[s] = v;
      lat[(int64_t)t * S + s] = v;
    }
    __syncthreads();
    float* tmp = prev; prev = cur; cur = tmp;
  }
  if (!backward && threadIdx.x == 0) {
    const float a = prev[Sb - 1], c = Sb >= 2 ? prev[Sb - 2] : NEG_INF;
    nll[b] = -lae2(a, c);
  }
}

// loss = mean_b nll_b / max(len_b, 1)
__global__ __launch_bounds__(64) void ctc_loss_kernel(const float* __restrict__ nll, const int32_t* __restrict__ tg_len, int B,
                                                      float* __restrict__ loss) {
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 64) s += nll[b] / (float)(tg_len[b] > 1 ? tg_len[b] : 1);
  s = wave_sum(s);
  if (threadIdx.x == 0) *loss = s / (float)B;
}

// block per (b, t): dlogits row
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, int64_t ld, const float* __restrict__ lse,
                                                       const int64_t* __restrict__ targets, int Lmax,
                                                       const int32_t* __restrict__ in_len, const int32_t* __restrict__ tg_len,
                                                       int B, int T, int V, int S, int blank, const float* __restrict__ alpha,
                                                       const float* __restrict__ beta, const float* __restrict__ nll,
                                                       const float* __restrict__ grad_out, float* __restrict__ dl, int64_t ldo) {
  extern __shared__ float acc[];       // [V]
  const int64_t row = blockIdx.x;
  const int b = (int)(row / T), t = (int)(row % T);
  float* out = dl + row * ldo;
  int Tb = in_len[b];
  Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
  const float nl = nll[b];
  if (t >= Tb) {                       // padded frame (ctc_loss gives it no gradient)
    for (int v = threadIdx.x; v < ldo; v += 256) out[v] = 0.f;
    return;
  }
  int Lb = tg_len[b];
  Lb = Lb < 0 ? 0 : (Lb > Lmax ? Lmax : Lb);
  const int Sb = 2 * Lb + 1;
  for (int v = threadIdx.x; v < V; v += 256) acc[v] = 0.f;
  __syncthreads();
  const float* lg = logits + row * ld;
  const float l_t = lse[row];
  const float* a = alpha + row * S;
  const float* be = beta + row * S;
  for (int s = threadIdx.x; s < Sb; s += 256) {
    const int v = (s & 1) ? (int)targets[(int64_t)b * Lmax + (s >> 1)] : blank;
    const float e = a[s] + be[s] - (lg[v] - l_t) + nl;
    if (e > NEG_INF) atomicAdd(&acc[v], expf(e));
  }
  __syncthreads();
  const float scale = grad_out[0] / ((float)(Lb > 1 ? Lb : 1) * (float)B);
  const bool finite = nl < INFINITY;   // an unreachable target has loss = inf; its gradient is NaN in torch (zero_infinity=False)
  for (int v = threadIdx.x; v < ldo; v += 256) {
    float g = 0.f;
    if (v < V) g = finite ? (expf(lg[v] - l_t) - acc[v]) * scale : NAN;
    out[v] = g;
  }
}

}  // namespace

extern "C" int64_t asr_ctc_workspace(int B, int T, int Lmax) {
  if (B <= 0 || T <= 0 || Lmax < 0) return 0;
  return (int64_t)B * T + 2 * (int64_t)B * T * (2 * Lmax + 1) + B;       // row lse | alpha | beta | nll
}

extern "C" int asr_ctc_fwd(const float* logits, int64_t ld, const int64_t* targets, const int32_t* input_lengths,
                           const int32_t* target_lengths, int B, int T, int V, int Lmax, int blank, float* workspace,
                           int64_t workspace_floats, float* loss, hipStream_t s) {
  ASR_CHECK_ARG(logits && targets && input_lengths && target_lengths && workspace && loss);
  ASR_CHECK_ARG(B > 0 && T > 0 && V > 0 && Lmax >= 1 && ld >= V && blank >= 0 && blank < V);
  ASR_CHECK_ARG(workspace_floats >= asr_ctc_workspace(B, T, Lmax));
  const int S = 2 * Lmax + 1;
  if ((size_t)(3 * S) * 4 > 64 * 1024) return ASR_EUNSUPPORTED;
  float* lse = workspace;
  float* alpha = lse + (int64_t)B * T;
  float* beta = alpha + (int64_t)B * T * S;
  float* nll = beta + (int64_t)B * T * S;
  const int64_t rows = (int64_t)B * T;
  AsrProfScope prof(ASR_OP_CE, s);
  hipLaunchKernelGGL(ctc_lse_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, logits, ld, rows, V, lse);
  ASR_LAUNCH_CHECK();
  const size_t lds = (size_t)3 * S * sizeof(float);
  static bool granted = false;
  if (lds > 48 * 1024 && !granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_lattice_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    granted = true;
  }
  // both recursions in one launch (alpha and beta are independent): the backward pass only runs the gradient kernel
  hipLaunchKernelGGL(ctc_lattice_kernel, dim3(B, 2), dim3(256), lds, s, logits, ld, lse, targets, Lmax, input_lengths, target_lengths,
                     T, S, blank, alpha, beta, nll);
  ASR_LAUNCH_CHECK();
  hipLaunchKernelGGL(ctc_loss_kernel, dim3(1), dim3(64), 0, s, nll, target_lengths, B, loss);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_ctc_bwd(const float* logits, int64_t ld, const int64_t* targets, const int32_t* input_lengths,
                           const int32_t* target_lengths, int B, int T, int V, int Lmax, int blank, const float* workspace,
                           const float* grad_out, float* dlogits, int64_t ldo, hipStream_t s) {
  ASR_CHECK_ARG(logits && targets && input_lengths && target_lengths && workspace && grad_out && dlogits);
  ASR_CHECK_ARG(B > 0 && T > 0 && V > 0 && Lmax >= 1 && ld >= V && ldo >= V && blank >= 0 && blank < V);
  const int S = 2 * Lmax + 1;
  const size_t lds = (size_t)V * sizeof(float);
  if (lds > 64 * 1024) return ASR_EUNSUPPORTED;
  const float* lse = workspace;
  const float* alpha = lse + (int64_t)B * T;
  const float* beta = alpha + (int64_t)B * T * S;
  const float* nll = beta + (int64_t)B * T * S;
  static bool granted = false;
  if (lds > 48 * 1024 && !granted) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ctc_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    granted = true;
  }
  AsrProfScope prof(ASR_OP_CE, s);
  hipLaunchKernelGGL(ctc_grad_kernel, dim3((unsigned)((int64_t)B * T)), dim3(256), lds, s, logits, ld, lse, targets, Lmax, input_lengths,
                     target_lengths, B, T, V, S, blank, alpha, beta, nll, grad_out, dlogits, ldo);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}
