// Internal interface of conv_level0.hip: the full-resolution level of vgg_cnn (reference: models/asr/transformer.py:42-47 --
// Conv2d(1, 64, 3, padding 1) + ReLU, Conv2d(64, 64, 3, padding 1) + ReLU, MaxPool2d(2, 2)) with the 64-channel full-resolution
// activations never stored: forward and both backward kernels re-create conv.0's output from the log-mel frames inside their own
// loaders, and the backward kernels expand the pooled gradient through the pooling's selection codes.
#pragma once
#include "common.h"

struct L0Args {
  const float* src;     // (B, H, W) fp32: the single input channel (B, 1, F, T) of the front end
  const float* w0;      // (64, 1, 3, 3) conv.0 weight (fp32 master)
  const float* b0;      // (64) conv.0 bias
  const bf16_t* wk;     // (64, 9, 64) conv.2 weights packed by asr_conv_pack_weight: forward = wk (co, tap, ci), data gradient = wd (ci, flipped tap, co)
  const float* b2;      // (64) conv.2 bias (forward)
  bf16_t* pool;         // forward out: (B, H/2, W/2, 64) = MaxPool(ReLU(conv.2(ReLU(conv.0(src)))))
  uint8_t* code;        // forward out / backward in: one selection byte per pooled element (0 = maximum is 0, 1 + k = first maximum at window position k)
  const bf16_t* dpool;  // backward in: gradient of `pool`
  float* ws;            // backward: per-workgroup partial sums (layout per kernel)
  float* db;            // weight-gradient kernel: (64) bias gradient of conv.2, accumulated with atomics (as conv3x3_wgrad_dma does)
  int B, H, W;
  int tiles_h, tiles_w, ntiles, patches_per_wg;   // filled by the launchers
  int wsplit;           // 1: conv.0's weights as whi + wlo (default); 0: bf16-rounded weights alone (tuning L0_WSPLIT)
  long long* dbg;       // tuning only (-DL0_TIMING builds): per-section clock totals of workgroup 0 of the forward kernel
};
