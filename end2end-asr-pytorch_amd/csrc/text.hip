// Host-side Levenshtein distance for CER / WER (reference: utils/metrics.py:48-76 calls the python-Levenshtein C extension once per
// utterance and step, trainer.py:62-75; a pure-Python distance costs 67 ms per batch of 32 x 100 characters -- nine GPU steps).
// Plain C++ on the host, no device work: sequences are int32 symbols (UTF-32 code points, or word ids), a batch per call.
#include <stdint.h>

#include <vector>

#include "../../include/asr_hip.h"

extern "C" int asr_edit_distance_batch(const int32_t* a, const int64_t* a_off, const int32_t* b, const int64_t* b_off, int n,
                                       int32_t* out) {
  if (n < 0 || (n > 0 && (!a_off || !b_off || !out))) return ASR_EINVAL;
  std::vector<int32_t> prev, cur;
  for (int p = 0; p < n; ++p) {
    const int64_t la = a_off[p + 1] - a_off[p], lb = b_off[p + 1] - b_off[p];
    if (la < 0 || lb < 0 || ((la > 0 && !a) || (lb > 0 && !b))) return ASR_EINVAL;
    const int32_t* x = a + a_off[p];
    const int32_t* y = b + b_off[p];
    prev.resize((size_t)lb + 1);
    cur.resize((size_t)lb + 1);
    for (int64_t j = 0; j <= lb; ++j) prev[(size_t)j] = (int32_t)j;
    for (int64_t i = 1; i <= la; ++i) {
      cur[0] = (int32_t)i;
      const int32_t xi = x[i - 1];
      for (int64_t j = 1; j <= lb; ++j) {
        const int32_t sub = prev[(size_t)j - 1] + (xi != y[j - 1]);
        const int32_t del = prev[(size_t)j] + 1, ins = cur[(size_t)j - 1] + 1;
        cur[(size_t)j] = sub < del ? (sub < ins ? sub : ins) : (del < ins ? del : ins);
      }
      prev.swap(cur);
    }
    out[p] = prev[(size_t)lb];
  }
  return ASR_OK;
}
