// Spectrogram front end on the GPU (reference: SpectrogramParser.parse_audio, utils/data_loader.py:72-89):
//   D = |STFT(y)| with n_fft = win_length = 320, hop 160, symmetric Hamming window, centred frames with reflect padding,
//   spect = log1p(D), then (spect - mean) / std over the whole utterance with the unbiased std.
// The DFT is a dense contraction: windowed frames (B*T, 320) x [cos | -sin] basis (322, 320)^T on asr_gemm_nt in fp32
// (MFMA 16x16x4 f32, fp32 accumulate); this file holds the kernels either side of it:
//   asr_stft_frames   : padded waveforms -> windowed frames, reflect padding resolved per sample, rows of frames past an
//                       utterance's end written as zeros
//   asr_spect_logmag  : (re, im) rows -> log1p(sqrt(re^2 + im^2)) stored as (B, F, T) (T contiguous, the loader's layout,
//                       data_loader.py:196-209: zero padded along T) + per-utterance sum
//   asr_spect_sqdev   : per-utterance sum of squared deviations from the mean (two-pass variance)
//   asr_spect_normalize: in place (x - mean) * rstd on the valid frames
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void stft_frames_kernel(const float* __restrict__ wav, int64_t wav_stride,
                                                          const int32_t* __restrict__ lengths, const float* __restrict__ window,
                                                          float* __restrict__ frames, int B, int Tmax, int n_fft, int hop) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)B * Tmax * n_fft;
  if (i >= total) return;
  const int n = (int)(i % n_fft);
  const int64_t ft = i / n_fft;
  const int t = (int)(ft % Tmax), b = (int)(ft / Tmax);
  const int len = max(lengths[b], 2);                 // the host path pads utterances shorter than 2 samples (audio.py)
  const int nfr = 1 + len / hop;                      // frames of this utterance: 1 + (len + n_fft - n_fft) / hop
  float v = 0.f;
  if (t < nfr) {
    const int pad = n_fft / 2;
    int j = t * hop + n - pad;                        // index into the unpadded signal
    bool ok = true;
    if (len > pad) {                                  // numpy 'reflect' (no edge repeat)
      if (j < 0) j = -j;
      if (j >= len) j = 2 * (len - 1) - j;
      ok = j >= 0 && j < len;
    } else {
      ok = j >= 0 && j < len;                         // very short signals: zero padding (audio.py)
    }
    if (ok) {
      const float s = j < lengths[b] ? wav[b * wav_stride + j] : 0.f;
      v = s * window[n];
    }
  }
  frames[i] = v;
}

// block = 256 threads over consecutive t of one (b, f) row segment: coalesced writes along T; reads re/im with stride ld
__global__ __launch_bounds__(256) void spect_logmag_kernel(const float* __restrict__ reim, int64_t ld, const int32_t* __restrict__ lengths,
                                                           float* __restrict__ spect, float* __restrict__ sums, int B, int F,
                                                           int Tmax, int hop) {
  __shared__ float red[4];
  const int tb = (Tmax + 255) / 256;
  const int tblk = blockIdx.x % tb, f = (blockIdx.x / tb) % F, b = blockIdx.x / (tb * F);
  const int t = tblk * 256 + threadIdx.x;
  const int nfr = 1 + max(lengths[b], 2) / hop;
  float v = 0.f;
  if (t < Tmax) {
    if (t < nfr) {
      const float* r = reim + ((int64_t)b * Tmax + t) * ld;
      const float re = r[f], im = r[F + f];
      v = log1pf(sqrtf(re * re + im * im));
    }
    spect[((int64_t)b * F + f) * Tmax + t] = v;
  }
  const float s = block_sum(v, red);
  if (threadIdx.x == 0) atomicAdd(sums + b, s);
}

__global__ __launch_bounds__(256) void spect_sqdev_kernel(const float* __restrict__ spect, const int32_t* __restrict__ lengths,
                                                          const float* __restrict__ sums, float* __restrict__ sq, int B, int F, int Tmax,
                                                          int hop) {
  __shared__ float red[4];
  const int tb = (Tmax + 255) / 256;
  const int tblk = blockIdx.x % tb, f = (blockIdx.x / tb) % F, b = blockIdx.x / (tb * F);
  const int t = tblk * 256 + threadIdx.x;
  const int nfr = 1 + max(lengths[b], 2) / hop;
  const float mean = sums[b] / ((float)nfr * (float)F);
  float v = 0.f;
  if (t < nfr && t < Tmax) {
    const float d = spect[((int64_t)b * F + f) * Tmax + t] - mean;
    v = d * d;
  }
  const float s = block_sum(v, red);
  if (threadIdx.x == 0) atomicAdd(sq + b, s);
}

__global__ __launch_bounds__(256) void spect_normalize_kernel(float* __restrict__ spect, const int32_t* __restrict__ lengths,
                                                              const float* __restrict__ sums, const float* __restrict__ sq, int B, int F,
                                                              int Tmax, int hop) {
  const int tb = (Tmax + 255) / 256;
  const int tblk = blockIdx.x % tb, f = (blockIdx.x / tb) % F, b = blockIdx.x / (tb * F);
  const int t = tblk * 256 + threadIdx.x;
  const int nfr = 1 + max(lengths[b], 2) / hop;
  if (t >= nfr || t >= Tmax) return;
  const float n = (float)nfr * (float)F;
  const float mean = sums[b] / n;
  const float rstd = rsqrtf(sq[b] / (n - 1.f));       // unbiased, as torch.Tensor.std() (data_loader.py:87-88)
  float* p = spect + ((int64_t)b * F + f) * Tmax + t;
  *p = (*p - mean) * rstd;
}

}  // namespace

extern "C" int asr_stft_frames(const float* wav, int64_t wav_stride, const int32_t* lengths, const float* window, float* frames, int B,
                               int Tmax, int n_fft, int hop, hipStream_t stream) {
  ASR_CHECK_ARG(wav && lengths && window && frames && B >= 0 && Tmax >= 0 && n_fft > 0 && hop > 0);
  const int64_t total = (int64_t)B * Tmax * n_fft;
  if (total == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_LAYOUT, stream);
  stft_frames_kernel<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>(wav, wav_stride, lengths, window, frames, B, Tmax, n_fft, hop);
  ASR_LAUNCH_CHECK();
  return ASR_OK;
}

extern "C" int asr_spect_finish(const float* reim, int64_t ld, const int32_t* lengths, float* spect, float* sums, float* sqdev, int B,
                                int F, int Tmax, int hop, int normalize, hipStream_t stream) {
  ASR_CHECK_ARG(reim && lengths && spect && sums && sqdev && B >= 0 && F > 0 && Tmax >= 0 && hop > 0 && ld >= 2 * F);
  if (B == 0 || Tmax == 0) return ASR_OK;
  AsrProfScope prof(ASR_OP_LAYOUT, stream);
  const unsigned grid = (unsigned)((int64_t)B * F * ((Tmax + 255) / 256));
  spect_logmag_kernel<<<grid, 256, 0, stream>>>(reim, ld, lengths, spect, sums, B, F, Tmax, hop);
  ASR_LAUNCH_CHECK();
  if (normalize) {
    spect_sqdev_kernel<<<grid, 256, 0, stream>>>(spect, lengths, sums, sqdev, B, F, Tmax, hop);
    ASR_LAUNCH_CHECK();
    spect_normalize_kernel<<<grid, 256, 0, stream>>>(spect, lengths, sums, sqdev, B, F, Tmax, hop);
    ASR_LAUNCH_CHECK();
  }
  return ASR_OK;
}
