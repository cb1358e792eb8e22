// Internal interface between conv.hip (asr_conv3x3_igemm dispatch) and conv_c64.hip (persistent 64 -> 64 channel bf16 kernel).
#pragma once
#include "common.h"

struct C64Args {
  const bf16_t* x;      // (B, H, W, 64) NHWC
  const bf16_t* wk;     // (64 co, 9 taps, 64 ci) packed weights (asr_conv_pack_weight)
  const float* bias;    // (64) or null
  const bf16_t* mask;   // (B, H, W, 64) or null: output zeroed where mask <= 0 (ReLU mask of the consumer's input, dgrad)
  bf16_t* y;            // (B, H, W, 64)
  bf16_t* pool;         // optional (forward, ReLU, no mask): (B, H/2, W/2, 64) = 2x2/2 floor max-pool of y, written by the same epilogue
  uint8_t* code;        // optional, with pool: one selection byte per pooled element (csrc/conv.hip pool_code); y may then be null (not stored)
  int B, H, W, relu;
  int ypix;             // bytes per pixel of y (0 = 128: 64 channels); 256 = y is a 128-channel tensor and this launch fills 64 of them (no mask then)
  int tiles_h, tiles_w, ntiles;   // filled by the launcher
  long long* dbg;                 // tuning only (-DC64_TIMING builds): per-section cycle totals of workgroup 0
  int ablate;                     // tuning only (ASR_C64_ABLATE in -DASR_TUNE_ABLATE builds)
};

int asr_conv3x3_c64_launch(const C64Args& a, hipStream_t s);
