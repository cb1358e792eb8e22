// Internal interface between conv.hip (asr_conv3x3_igemm / asr_conv3x3_relu_pool_tcf_code dispatch) and conv_ws.hip (persistent,
// weight-stationary bf16 kernel for vgg_cnn's second level: the three launches with 128 input channels and conv.5's forward, 64 -> 128).
#pragma once
#include "common.h"

struct WsArgs {
  const bf16_t* x;      // (B, H, W, Cin) NHWC
  const bf16_t* wk;     // (Cout, 9 taps, Cin) packed weights (asr_conv_pack_weight)
  const float* bias;    // (Cout) or null
  const bf16_t* mask;   // (B, H, W, Cout) or null: output zeroed where mask <= 0 (ReLU mask of the consumer's input, dgrad)
  bf16_t* y;            // (B, H, W, Cout); unused by the pooled form
  bf16_t* pool;         // pooled form: (B, W/2, Cout, H/2) = the encoder layout (B, T', C F') of max-pool(ReLU(conv))
  uint8_t* code;        // pooled form: one selection byte per pooled element, same layout -- or, with code_cl != 0, CHANNEL LAST
                        // (B, W/2, H/2, Cout): the 128 bytes of a pooled pixel contiguous, what asr_gemm_nn_poolbwd's epilogue reads
  int code_cl;
  const uint8_t* bits_in;   // or null: ReLU mask of the output as ONE BIT per element (layout: asr_relu_bits_bytes), Cin = Cout = 128
  uint8_t* bits_out;        // or null: write such bits for this launch's ReLU output (Cin = 64, Cout = 128)
  int B, H, W, Cin, Cout, relu;      // Cin = 128 (any form) or 64 (Cout = 128, no mask, not pooled: conv.5 forward)
  int tiles_h, tiles_w, ntiles;   // filled by the launcher
  long long* dbg;                 // development only (tuning WS_DBG): per-section clock totals of workgroup 0
};

// ASR_EUNSUPPORTED when the shape is outside the kernel's domain (the caller falls back to the generic implicit GEMM)
int asr_conv3x3_ws128_launch(const WsArgs& a, hipStream_t s);
