// Error strings, ABI version and the built-in HIP-event profiler used by bench.py's roofline leg:
// every launch of ONE selected op id is bracketed by hipEvents on the stream it is launched on.
#include <mutex>
#include <vector>

#include "common.h"
#include <atomic>
#include <string.h>

namespace {
struct Pair { hipEvent_t a, b; };
struct OpProf {
  bool enabled = false;
  std::vector<Pair> pool;   // created lazily, reused across captures
  size_t used = 0;
};
OpProf g_prof[ASR_OP_COUNT];
std::mutex g_mu;
constexpr size_t kMaxPairs = 1 << 16;
}  // namespace

AsrProfScope::AsrProfScope(int op_, hipStream_t s_) : op(op_), s(s_), slot(nullptr) {
  if (op < 0 || op >= ASR_OP_COUNT || !g_prof[op].enabled) return;
  std::lock_guard<std::mutex> lk(g_mu);
  OpProf& P = g_prof[op];
  if (P.used >= kMaxPairs) return;
  if (P.used == P.pool.size()) {
    Pair pr;
    if (hipEventCreate(&pr.a) != hipSuccess || hipEventCreate(&pr.b) != hipSuccess) return;
    P.pool.push_back(pr);
  }
  Pair* pr = &P.pool[P.used++];
  (void)hipEventRecord(pr->a, s);
  slot = reinterpret_cast<void*>(P.used);   // index + 1
}
AsrProfScope::~AsrProfScope() {
  if (!slot) return;
  std::lock_guard<std::mutex> lk(g_mu);
  OpProf& P = g_prof[op];
  const size_t idx = reinterpret_cast<size_t>(slot) - 1;
  if (idx < P.pool.size()) (void)hipEventRecord(P.pool[idx].b, s);
}

extern "C" const char* asr_strerror(int code) {
  switch (code) {
    case ASR_OK: return "ok";
    case ASR_EINVAL: return "invalid argument";
    case ASR_ELAUNCH: return "kernel launch failed";
    case ASR_EUNSUPPORTED: return "unsupported shape/alignment for this kernel";
    case ASR_ERUNTIME: return "HIP runtime error";
    default: return "unknown error";
  }
}
extern "C" int asr_abi_version(void) { return 4; }

extern "C" int asr_prof_enable(int op, int enable) {
  if (op < 0 || op >= ASR_OP_COUNT) return ASR_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof[op].enabled = enable != 0;
  if (enable) g_prof[op].used = 0;
  return ASR_OK;
}
extern "C" int asr_prof_collect(int op, double* total_ms, int64_t* launches) {
  if (op < 0 || op >= ASR_OP_COUNT || !total_ms || !launches) return ASR_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  OpProf& P = g_prof[op];
  double tot = 0.0;
  for (size_t i = 0; i < P.used; ++i) {
    if (hipEventSynchronize(P.pool[i].b) != hipSuccess) return ASR_ERUNTIME;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, P.pool[i].a, P.pool[i].b) != hipSuccess) return ASR_ERUNTIME;
    tot += ms;
  }
  *total_ms = tot;
  *launches = (int64_t)P.used;
  return ASR_OK;
}

// ------------------------------------------------------------------------------------------------ tuning switches
// The library reads NO environment variables: A/B switches are set through the ABI (asr_set_tuning) by the host layer
// (asr_hip/lib.py forwards ASR_<NAME> variables once, at load).  Unknown names are refused.
namespace {
const char* const kTuningNames[] = {
    "ATTN_GENERIC", "IGEMM_TH", "IGEMM_TPS", "IGEMM_WBUF", "CONV1_WGRAD_MFMA", "IGEMM_ABLATE", "C64", "CONV_POOL", "WGRAD_ABLATE",
    "WGRAD_DMA", "CONV1_WGRAD_WGS", "C64_PER_CU", "C64_ABLATE", "C64_SHAPE", "GEMM_NS", "GEMM_TILE", "GEMM_GENERIC", "TN_WGS",
    "TN_128", "TN_128_MIN", "TN_128_RM", "TN_NBUF", "NN_BIG", "NN_RING", "TN_GROUP_SLICE_MIN", "GEMM_BIG_MIN", "NT_RING", "TN_PIPE", "TN_PIPE_MIN", "GEMM_ABLATE", "ATTN_SHORT", "ATTN_SHORT_BWD", "ATTN_BOTH", "NNTN_STAGES",
    "ATTN_PP", "ATTN_PP_MIN", "ATTN_PP_TAIL", "ATTN_PP_PRIO", "ATTN_PP_STAGGER", "ATTN_PP_STAGGER_SEL", "TN_GROUP_TILE", "TN_GROUP_MROWS",
    "TN_GROUP_WGS", "WGRAD_XCD", "IGEMM_XCD", "C64_SPLIT", "GEMM_BIG", "GEMM_BIG_NS", "GEMM_BIG_NN", "L0_WSPLIT", "WS128", "WS64", "WS64_PER_CU", "WS_PAIR", "WS_BITS", "NN_ROWDOT", "ATTN_BWD_FUSED",
#ifdef ASR_TUNE_ABLATE
    "WS_DBG",          // development builds only: a device ADDRESS the timing instantiations of conv_ws.hip write through
#endif
};
constexpr int kNumTuning = (int)(sizeof(kTuningNames) / sizeof(kTuningNames[0]));
std::atomic<int64_t> g_tuning_value[kNumTuning];
std::atomic<bool> g_tuning_set[kNumTuning];
int tuning_index(const char* name) {
  for (int i = 0; i < kNumTuning; ++i)
    if (strcmp(name, kTuningNames[i]) == 0) return i;
  return -1;
}
}  // namespace

int64_t asr_tuning(const char* name, int64_t dflt) {
  const int i = tuning_index(name);
  return (i >= 0 && g_tuning_set[i].load(std::memory_order_acquire)) ? g_tuning_value[i].load(std::memory_order_relaxed) : dflt;
}

extern "C" int asr_set_tuning(const char* name, int64_t value) {
  if (!name) return ASR_EINVAL;
  const int i = tuning_index(name);
  if (i < 0) return ASR_EINVAL;
  g_tuning_value[i].store(value, std::memory_order_relaxed);
  g_tuning_set[i].store(true, std::memory_order_release);
  return ASR_OK;
}
extern "C" int asr_clear_tuning(const char* name) {
  if (!name) { for (int i = 0; i < kNumTuning; ++i) g_tuning_set[i].store(false, std::memory_order_release); return ASR_OK; }
  const int i = tuning_index(name);
  if (i < 0) return ASR_EINVAL;
  g_tuning_set[i].store(false, std::memory_order_release);
  return ASR_OK;
}
