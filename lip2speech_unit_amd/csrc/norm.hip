// LayerNorm with fp32 statistics, one wavefront per row (64-lane shuffle reduction, row cached in registers).
//   fairseq LayerNorm (eps 1e-5): avhubert/hubert.py:400,720 and the 24 transformer layers + final layer_norm
//   espnet LayerNorm (eps 1e-12): espnet/nets/pytorch_backend/transformer/layer_norm.py:12-33
// zero_prefix > 0 implements the video-only modality fuse of hubert.py:706-720: the normalised vector is
// [zeros(zero_prefix) || x]; the zero half contributes -mean*rstd*gamma+beta and is materialised for post_extract_proj.
#include "l2s_common.h"
#include <cstdlib>

namespace {

constexpr int MAXV4 = 8;  // float4 per lane -> C <= 2048

template <typename ET, bool XF32, bool YF32>
__global__ __launch_bounds__(256) void layernorm_kernel(const void* __restrict__ x, int ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps,
                                                        void* __restrict__ y, int ldy, uint16_t* __restrict__ y2,
                                                        int ldy2, int M, int C, int zp,
                                                        const int32_t* __restrict__ lens, int len_mul, int mask_T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  bool keep = true;
  if (lens) {
    const int clip = row / mask_T;
    keep = (row - clip * mask_T) < lens[clip] * len_mul;
  }
  const int nv = C >> 2;  // float4 groups
  float4 v[MAXV4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV4; ++i) {
    const int gi = i * 64 + lane;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gi < nv) {
      if (XF32) {
        v[i] = *reinterpret_cast<const float4*>((const float*)x + (int64_t)row * ldx + gi * 4);
      } else {
        const uint2 q = *reinterpret_cast<const uint2*>((const uint16_t*)x + (int64_t)row * ldx + gi * 4);
        v[i] = make_float4(ET::to_f32((uint16_t)(q.x & 0xffff)), ET::to_f32((uint16_t)(q.x >> 16)),
                           ET::to_f32((uint16_t)(q.y & 0xffff)), ET::to_f32((uint16_t)(q.y >> 16)));
      }
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float ctot = (float)(C + zp);
  const float mean = wave_sum(sum) / ctot;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV4; ++i) {
    const int gi = i * 64 + lane;
    if (gi < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (c * c + d * d);
    }
  }
  sq = wave_sum(sq) + (float)zp * mean * mean;
  const float rstd = rsqrtf(sq / ctot + eps);

  auto store4 = [&](int col, float4 o) {
    if (!keep) o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (YF32) {
      *reinterpret_cast<float4*>((float*)y + (int64_t)row * ldy + col) = o;
    } else {
      uint2 q;
      q.x = (uint32_t)ET::from_f32(o.x) | ((uint32_t)ET::from_f32(o.y) << 16);
      q.y = (uint32_t)ET::from_f32(o.z) | ((uint32_t)ET::from_f32(o.w) << 16);
      *reinterpret_cast<uint2*>((uint16_t*)y + (int64_t)row * ldy + col) = q;
    }
    if (y2) {
      uint2 q;
      q.x = (uint32_t)ET::from_f32(o.x) | ((uint32_t)ET::from_f32(o.y) << 16);
      q.y = (uint32_t)ET::from_f32(o.z) | ((uint32_t)ET::from_f32(o.w) << 16);
      *reinterpret_cast<uint2*>(y2 + (int64_t)row * ldy2 + col) = q;
    }
  };
  // zero prefix: (0 - mean) * rstd * gamma + beta
  const float z = -mean * rstd;
  for (int gi = lane; gi < (zp >> 2); gi += 64) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + gi * 4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + gi * 4);
    store4(gi * 4, make_float4(z * g.x + bt.x, z * g.y + bt.y, z * g.z + bt.z, z * g.w + bt.w));
  }
#pragma unroll
  for (int i = 0; i < MAXV4; ++i) {
    const int gi = i * 64 + lane;
    if (gi < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + zp + gi * 4);
      const float4 bt = *reinterpret_cast<const float4*>(beta + zp + gi * 4);
      store4(zp + gi * 4, make_float4((v[i].x - mean) * rstd * g.x + bt.x, (v[i].y - mean) * rstd * g.y + bt.y,
                                      (v[i].z - mean) * rstd * g.z + bt.z, (v[i].w - mean) * rstd * g.w + bt.w));
    }
  }
}

// The two widths the path is made of (C = 1024: 49 encoder LayerNorms, C = 512: 61 conformer ones), no zero prefix: a wave
// takes ROWS rows at once with every load issued before the first reduction, so the two dependent shuffle reductions of
// one row overlap the memory latency of the others (one row per short-lived wave left the kernel at 4.0 TB/s of its
// 4 B in + 2 B out per element), and the 16-bit row leaves as 16-byte stores after a lane-pair exchange.
template <typename ET, int VPL, int ROWS, bool YF32>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps,
                                                             void* __restrict__ y, int ldy, int M,
                                                             const int32_t* __restrict__ lens, int len_mul, int mask_T) {
  constexpr int C = VPL * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * ROWS;
  if (row0 >= M) return;
  float4 v[ROWS][VPL];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      // lane pairs (2p, 2p+1) own 8 consecutive channels of each 512-channel group: float4 index i*128... + lane
      v[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row0 + r < M) v[r][i] = *reinterpret_cast<const float4*>(x + (int64_t)(row0 + r) * ldx + (i * 64 + lane) * 4);
    }
  float4 g[VPL], bt[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    g[i] = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    bt[i] = *reinterpret_cast<const float4*>(beta + (i * 64 + lane) * 4);
  }
  float mean[ROWS], rstd[ROWS];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);
    mean[r] = s;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    mean[r] *= (1.0f / (float)C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float a = v[r][i].x - mean[r], b = v[r][i].y - mean[r], c = v[r][i].z - mean[r], d = v[r][i].w - mean[r];
      q += (a * a + b * b) + (c * c + d * d);
    }
    rstd[r] = q;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int r = 0; r < ROWS; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = row0 + r;
    if (row >= M) break;
    const float rs = rsqrtf(rstd[r] * (1.0f / (float)C) + eps);
    bool keep = true;
    if (lens) {
      const int clip = row / mask_T;
      keep = (row - clip * mask_T) < lens[clip] * len_mul;
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float4 o = make_float4((v[r][i].x - mean[r]) * rs * g[i].x + bt[i].x, (v[r][i].y - mean[r]) * rs * g[i].y + bt[i].y,
                             (v[r][i].z - mean[r]) * rs * g[i].z + bt[i].z, (v[r][i].w - mean[r]) * rs * g[i].w + bt[i].w);
      if (!keep) o = make_float4(0.f, 0.f, 0.f, 0.f);
      if (YF32) {
        *reinterpret_cast<float4*>((float*)y + (int64_t)row * ldy + (i * 64 + lane) * 4) = o;
      } else {
        // 8 bytes per lane -> 16 bytes per even lane: the odd neighbour hands over its packed half
        const uint32_t lo = ET::pack2(o.x, o.y), hi = ET::pack2(o.z, o.w);
        const uint32_t nlo = __shfl_xor(lo, 1, 64), nhi = __shfl_xor(hi, 1, 64);
        if (!(lane & 1))
          *reinterpret_cast<uint4*>((uint16_t*)y + (int64_t)row * ldy + (i * 64 + lane) * 4) = make_uint4(lo, hi, nlo, nhi);
      }
    }
  }
}

// Split-K tail fused with the LayerNorm that follows a residual-stream Linear at one-clip M (l2s_splitk_reduce, then
// layernorm_rows_kernel, in one launch: at M = 100-500 rows both are latency-bound launches of a few microseconds).
// One wave per row: v = x[row] + sum_{s < S} P[row, s*C ..] (s ascending), x[row] = v unless the LayerNorm writes fp32 over x
// itself, y[row] = LayerNorm(v).  The additions are l2s_splitk_reduce's in the same order; the statistics are written as in
// layernorm_rows_kernel (results equal to fp32 rounding: the compiler contracts the multiply-adds of the two kernels differently).
template <typename ET, int VPL, bool YF32>
__global__ __launch_bounds__(256) void splitk_reduce_ln_kernel(const float* __restrict__ P, int ldp, int S, float* x, int ldx,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, void* y, int ldy, int M,
                                                               const int32_t* __restrict__ lens, int len_mul, int mask_T, int write_x) {
  constexpr int C = VPL * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  float4 v[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) v[i] = *reinterpret_cast<const float4*>(x + (int64_t)row * ldx + (i * 64 + lane) * 4);
  for (int s = 0; s < S; ++s) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float4 q = *reinterpret_cast<const float4*>(P + (int64_t)row * ldp + (int64_t)s * C + (i * 64 + lane) * 4);
      v[i].x += q.x; v[i].y += q.y; v[i].z += q.z; v[i].w += q.w;
    }
  }
  if (write_x) {
#pragma unroll
    for (int i = 0; i < VPL; ++i) *reinterpret_cast<float4*>(x + (int64_t)row * ldx + (i * 64 + lane) * 4) = v[i];
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum * (1.0f / (float)C);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rs = rsqrtf(q * (1.0f / (float)C) + eps);
  bool keep = true;
  if (lens) {
    const int clip = row / mask_T;
    keep = (row - clip * mask_T) < lens[clip] * len_mul;
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float4 g = *reinterpret_cast<const float4*>(gamma + (i * 64 + lane) * 4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + (i * 64 + lane) * 4);
    float4 o = make_float4((v[i].x - mean) * rs * g.x + bt.x, (v[i].y - mean) * rs * g.y + bt.y,
                           (v[i].z - mean) * rs * g.z + bt.z, (v[i].w - mean) * rs * g.w + bt.w);
    if (!keep) o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (YF32) {
      *reinterpret_cast<float4*>((float*)y + (int64_t)row * ldy + (i * 64 + lane) * 4) = o;
    } else {
      const uint32_t lo = ET::pack2(o.x, o.y), hi = ET::pack2(o.z, o.w);
      *reinterpret_cast<uint2*>((uint16_t*)y + (int64_t)row * ldy + (i * 64 + lane) * 4) = make_uint2(lo, hi);
    }
  }
}

template <typename ET>
int launch_splitk_ln(const float* P, int ldp, int S, float* x, int ldx, const float* g, const float* b, float eps, void* y, int yf,
                     int ldy, int M, int C, const int32_t* lens, int len_mul, int mask_T, hipStream_t st) {
  dim3 grid((M + 3) / 4), block(256);
  const int write_x = !(yf && y == (void*)x);
#define L2S_SKLN(VPL, YF) hipLaunchKernelGGL((splitk_reduce_ln_kernel<ET, VPL, YF>), grid, block, 0, st, P, ldp, S, x, ldx, g, b, eps, y, ldy, M, lens, len_mul, mask_T, write_x)
  if (C == 1024) { if (yf) L2S_SKLN(4, true); else L2S_SKLN(4, false); }
  else { if (yf) L2S_SKLN(2, true); else L2S_SKLN(2, false); }
#undef L2S_SKLN
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET>
int launch_ln(const void* x, int xf, int ldx, const float* g, const float* b, float eps, void* y, int yf, int ldy,
              uint16_t* y2, int ldy2, int M, int C, int zp, const int32_t* lens, int len_mul, int mask_T, hipStream_t st) {
  dim3 grid((M + 3) / 4), block(256);
  static const int rows_on = [] { const char* e = getenv("L2S_LN_ROWS"); return e ? atoi(e) : 1; }();   // A/B switch
  if (rows_on && xf && !y2 && zp == 0 && (C == 1024 || C == 512) && (yf || (((uintptr_t)y & 15) == 0 && (ldy & 7) == 0))) {
    const float* xf32 = (const float*)x;
    if (C == 1024) {
      dim3 g2((M + 7) / 8);
      if (yf) hipLaunchKernelGGL((layernorm_rows_kernel<ET, 4, 2, true>), g2, block, 0, st, xf32, ldx, g, b, eps, y, ldy, M, lens, len_mul, mask_T);
      else hipLaunchKernelGGL((layernorm_rows_kernel<ET, 4, 2, false>), g2, block, 0, st, xf32, ldx, g, b, eps, y, ldy, M, lens, len_mul, mask_T);
    } else {
      dim3 g4((M + 15) / 16);
      if (yf) hipLaunchKernelGGL((layernorm_rows_kernel<ET, 2, 4, true>), g4, block, 0, st, xf32, ldx, g, b, eps, y, ldy, M, lens, len_mul, mask_T);
      else hipLaunchKernelGGL((layernorm_rows_kernel<ET, 2, 4, false>), g4, block, 0, st, xf32, ldx, g, b, eps, y, ldy, M, lens, len_mul, mask_T);
    }
    L2S_CHECK_LAUNCH();
    return L2S_OK;
  }
  if (xf && yf) hipLaunchKernelGGL((layernorm_kernel<ET, true, true>), grid, block, 0, st, x, ldx, g, b, eps, y, ldy, y2, ldy2, M, C, zp, lens, len_mul, mask_T);
  else if (xf) hipLaunchKernelGGL((layernorm_kernel<ET, true, false>), grid, block, 0, st, x, ldx, g, b, eps, y, ldy, y2, ldy2, M, C, zp, lens, len_mul, mask_T);
  else if (yf) hipLaunchKernelGGL((layernorm_kernel<ET, false, true>), grid, block, 0, st, x, ldx, g, b, eps, y, ldy, y2, ldy2, M, C, zp, lens, len_mul, mask_T);
  else hipLaunchKernelGGL((layernorm_kernel<ET, false, false>), grid, block, 0, st, x, ldx, g, b, eps, y, ldy, y2, ldy2, M, C, zp, lens, len_mul, mask_T);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

}  // namespace

extern "C" int l2s_layernorm(const void* x, int x_is_f32, int ldx, const float* gamma, const float* beta, float eps,
                             void* y, int y_is_f32, int ldy, void* y2, int ldy2, int M, int C, int zero_prefix,
                             const int32_t* lens, int len_mul, int mask_T, int dtype, void* stream) {
  if (!x || !gamma || !beta || !y) return L2S_EINVAL;
  if (M <= 0 || C <= 0 || zero_prefix < 0) return L2S_ESHAPE;
  if (C > MAXV4 * 64 * 4) return L2S_EUNSUPPORTED;
  if (lens && (len_mul <= 0 || mask_T <= 0)) return L2S_EINVAL;
  if ((C & 3) || (zero_prefix & 3) || (ldx & 3) || (ldy & 3) || (y2 && (ldy2 & 3))) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == L2S_F16)
    return launch_ln<ElemF16>(x, x_is_f32, ldx, gamma, beta, eps, y, y_is_f32, ldy, (uint16_t*)y2, ldy2, M, C, zero_prefix, lens, len_mul, mask_T, st);
  if (dtype == L2S_BF16)
    return launch_ln<ElemBF16>(x, x_is_f32, ldx, gamma, beta, eps, y, y_is_f32, ldy, (uint16_t*)y2, ldy2, M, C, zero_prefix, lens, len_mul, mask_T, st);
  return L2S_EINVAL;
}

extern "C" int l2s_splitk_reduce_layernorm(const float* P, int ldp, int S, float* x, int ldx, const float* gamma, const float* beta,
                                           float eps, void* y, int y_is_f32, int ldy, int M, int C, const int32_t* lens,
                                           int len_mul, int mask_T, int dtype, void* stream) {
  if (!P || !x || !gamma || !beta || !y) return L2S_EINVAL;
  if (M <= 0 || S <= 0) return L2S_ESHAPE;
  if (C != 1024 && C != 512) return L2S_EUNSUPPORTED;     // the path's two model widths (layernorm_rows_kernel)
  if (ldp < S * C || ldx < C || ldy < C) return L2S_ESHAPE;
  if (lens && (len_mul <= 0 || mask_T <= 0)) return L2S_EINVAL;
  if ((ldp & 3) || (ldx & 3) || (ldy & 3) || ((uintptr_t)P & 15) || ((uintptr_t)x & 15) || ((uintptr_t)y & (y_is_f32 ? 15 : 7)))
    return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == L2S_F16) return launch_splitk_ln<ElemF16>(P, ldp, S, x, ldx, gamma, beta, eps, y, y_is_f32, ldy, M, C, lens, len_mul, mask_T, st);
  if (dtype == L2S_BF16) return launch_splitk_ln<ElemBF16>(P, ldp, S, x, ldx, gamma, beta, eps, y, y_is_f32, ldy, M, C, lens, len_mul, mask_T, st);
  return L2S_EINVAL;
}
