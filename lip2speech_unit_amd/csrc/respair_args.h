// Arguments shared by the fused conv-pair kernels (respair.hip: C = 64 / 128; respair256.hip: C = 256).
#pragma once
#include <stdint.h>

namespace l2s_rp {

struct RpArgs {
  const uint16_t* X; const uint16_t* W1; const uint16_t* W2; const float* b1; const float* b2;
  uint16_t* Y; float* XS; const int32_t* lens;
  int len_mul, T, k, dil, h1, h2, S, ntiles, tiles_per_clip, accumulate, xcd_order, write_xs;
  float slope;
};

// The last pairs of a stage's ResBlocks in ONE launch (respair_phase.hip: respair_final_kernel): Y = leaky_relu(sum_j x'_j)
struct RpFinalArgs {
  const uint16_t* X[3]; const uint16_t* W1[3]; const uint16_t* W2[3]; const float* b1[3]; const float* b2[3];
  int k[3], dil[3];
  uint16_t* Y; const int32_t* lens;
  int nj, len_mul, T, h2max, S, ntiles, tiles_per_clip, xcd_order;
  float slope;
};

}  // namespace l2s_rp
