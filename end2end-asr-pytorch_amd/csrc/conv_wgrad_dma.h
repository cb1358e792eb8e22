// Internal interface between conv.hip (asr_conv3x3_wgrad_nhwc dispatch) and conv_wgrad_dma.hip (LDS-DMA pipelined bf16 kernel).
#pragma once
#include "common.h"

struct WgdArgs {
  const bf16_t* x;      // (B, H, W, Cin) NHWC
  const bf16_t* dy;     // (B, H, W, Cout) NHWC
  float* db;            // (Cout) accumulated with atomics, or null
  float* ws;            // per-workgroup partial dW blocks [blocks_y][wgx][9][64 co][64 ci]
  int B, H, W, Cin, Cout, tiles_h, tiles_w, npatch, patches_per_wg, nci;
  int wgx, blocks_y, xcd_order;   // filled by the launcher: the grid is ONE dimension of wgx * blocks_y workgroups
};

int asr_conv3x3_wgrad_dma_launch(const WgdArgs& p, unsigned wgx, unsigned blocks_y, hipStream_t s);

// the grid of the weight-gradient launch and of its partial-block workspace (conv.hip); also used by conv_level0.hip
void asr_conv3x3_wgrad_grid(int B, int H, int W, int Cin, int Cout, int* wgx, int* blocks_y, int* patches_per_wg);
