// Internal interface between conv.hip (asr_conv1_wgrad dispatch) and conv1_wgrad_mfma.hip (bf16 storage mode, 64 channels).
#pragma once
#include "common.h"

int asr_conv1_wgrad_mfma_launch(const float* x, const bf16_t* dy, float* dw, float* db, int B, int H, int W, hipStream_t s);
