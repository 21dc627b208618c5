// 3x3 / stride 1 / pad 1 INT8 convolution with the input HALO resident in LDS (csrc/iconv.hip).
#pragma once
#include "common.h"

namespace mixdq {

struct HaloConvArgs {
  const int8_t* X;        // [NI, H, W, C] int8 (NHWC)
  const int8_t* Wt;       // [K, 3, 3, C] int8
  const float* scale;     // [K]
  const __half* bias;     // [K] or null
  const float* table;     // [81][K] tap-rectangle sums (mixdq_conv_border_table)
  const float* zp;        // device scalar: the activation zero point
  __half* D;              // [NI, H, W, K]
  const __half* res;      // residual or null: [NI, H, W, K] (res_div == 1) or [NI, K] (res_div == H * W)
  int64_t res_div;
  int NI, H, W, C, K;     // H, W: the conv's input (= output) size
  int unfused;
  int ups;                // 1: X is [NI, H / 2, W / 2, C] and the conv reads its nearest-neighbour 2x
                          //    upsampling (Upsample2D: conv(interpolate(x))) without materialising it
};

// Tile id the halo kernel would run this problem on (90: 8x16 pixels x 80 channels, 91: 8x8 x 80, 92: 16x16 x 80, 93: 16x16 x 160), or 0
// when the problem is outside its range (then the implicit-GEMM family of csrc/igemm.hip runs it).
int halo_conv_select(int NI, int H, int W, int C, int K, int R, int S, int stride, int pad);

// Launch; `tile` from halo_conv_select (or forced).  Returns a mixdq_status.
int halo_conv_launch(const HaloConvArgs& a, int tile, hipStream_t stream);

}  // namespace mixdq
