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

// Swizzle key of patch / tile column x in the weight-gradient stages (conv_wgrad_dma.hip, conv_level0.hip): the 16-byte chunk c of the pixel
// in column x sits in slot c ^ key(x) of the pixel's 128 bytes.  ds_read_b64_tr_b16 serves lanes 0 - 31 in one LDS cycle: two lane groups,
// 4 consecutive pixels x 32 bytes each, EIGHT columns apart -- with the plain key x & 7 (rounds 2 - 5) columns x and x + 8 land on the same
// banks and every transposing read takes two cycles (SQ_LDS_BANK_CONFLICT 42 M of 91 M LDS cycles, profiles/r04_level0_pmc.txt); bit 3 of
// the column flips chunk bit 2 and the two groups use different quarters of the 256-byte bank row.
__host__ __device__ __forceinline__ int wgd_key(int x) { return (x & 7) ^ (((x >> 3) & 1) << 2); }

int asr_conv3x3_wgrad_dma_launch(const WgdArgs& p, unsigned wgx, unsigned blocks_y, hipStream_t s);

// the grid of the weight-gradient launch and of its partial-block workspace (conv.hip); also used by conv_level0.hip
void asr_conv3x3_wgrad_grid(int B, int H, int W, int Cin, int Cout, int* wgx, int* blocks_y, int* patches_per_wg);
