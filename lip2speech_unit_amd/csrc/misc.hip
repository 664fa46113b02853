// Bandwidth-bound layout / glue kernels of the path (casts, time tiling, embedding lookup, vocoder tail).
#include "l2s_common.h"

namespace {

inline int grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

// sequence_generator.py:130-131: encoder_out.repeat_interleave(2, dim=0) on [T,B,C]; here rows are (b,t).
template <typename ET>
__global__ void repeat2_cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t rows, int C) {
  const int c4 = C >> 2;
  const int64_t total = rows * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t r = i / c4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * C + cc * 4);
    uint2 q;
    q.x = (uint32_t)ET::from_f32(v.x) | ((uint32_t)ET::from_f32(v.y) << 16);
    q.y = (uint32_t)ET::from_f32(v.z) | ((uint32_t)ET::from_f32(v.w) << 16);
    *reinterpret_cast<uint2*>(y + (2 * r) * C + cc * 4) = q;
    *reinterpret_cast<uint2*>(y + (2 * r + 1) * C + cc * 4) = q;
  }
}

template <typename ET>
__global__ void cast_to16_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ y, int ldy, int64_t M,
                                 int C) {
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    y[r * ldy + c] = ET::from_f32(x[r * ldx + c]);
  }
}

template <typename ET>
__global__ void cast_to32_kernel(const uint16_t* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int64_t M,
                                 int C) {
  const int64_t total = M * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    y[r * ldy + c] = ET::to_f32(x[r * ldx + c]);
  }
}

template <typename ET, bool VF32>
__global__ void broadcast_rows_kernel(const void* __restrict__ v, int ldv, uint16_t* __restrict__ y, int ldy, int col0,
                                      const int32_t* __restrict__ lens, int len_mul, int B, int T, int C) {
  const int64_t total = (int64_t)B * T * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    float f = VF32 ? ((const float*)v)[(int64_t)b * ldv + c] : ET::to_f32(((const uint16_t*)v)[(int64_t)b * ldv + c]);
    if (lens && t >= lens[b] * len_mul) f = 0.f;
    y[r * ldy + col0 + c] = ET::from_f32(f);
  }
}

// [B,C,T] fp32 -> rows (b,t), cols col0..col0+C; LDS transpose tile so both sides are coalesced
template <typename ET>
__global__ void transpose_ct_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int ldy, int col0,
                                    const int32_t* __restrict__ lens, int len_mul, int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, t = t0 + tx;
    tile[k][tx] = (c < C && t < T) ? x[((int64_t)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  const int lim = lens ? lens[b] * len_mul : T;
  for (int k = ty; k < 32; k += 8) {
    const int t = t0 + k, c = c0 + tx;
    if (t < T && c < C) y[((int64_t)b * T + t) * ldy + col0 + c] = ET::from_f32(t < lim ? tile[tx][k] : 0.f);
  }
}

template <typename ET>
__global__ void embedding_kernel(const int32_t* __restrict__ code, const uint16_t* __restrict__ table,
                                 uint16_t* __restrict__ y, int ldy, const int32_t* __restrict__ lens, int B, int L,
                                 int C) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)B * L * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t r = i / c8;
    const int b = (int)(r / L), l = (int)(r - (int64_t)b * L);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!lens || l < lens[b]) v = *reinterpret_cast<const uint4*>(table + (int64_t)code[r] * C + cc * 8);
    *reinterpret_cast<uint4*>(y + r * ldy + cc * 8) = v;
  }
}

// ---- in-memory stage 1 -> stage 2 hand-off: what the reference does through files -----------------------
// predicted tokens -> unit ids -> nn.Embedding rows: inference.py:267-274 writes the units as text, create_dataset.py:366-428
// copies them, dataset_multi_input.py:41-110 parses them back; token t is unit t - 4 (fairseq's 4 specials precede the units)
template <typename ET>
__global__ void embedding_tokens_kernel(const int32_t* __restrict__ tok, int ldt, int token_offset, int n_rows,
                                        const uint16_t* __restrict__ table, uint16_t* __restrict__ y, int ldy,
                                        const int32_t* __restrict__ lens, int len_mul, int B, int L, int C) {
  const int c8 = C >> 3;
  const int64_t total = (int64_t)B * L * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t r = i / c8;
    const int b = (int)(r / L), l = (int)(r - (int64_t)b * L);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (!lens || l < lens[b] * len_mul) {
      int u = tok[(int64_t)b * ldt + l] - token_offset;
      u = u < 0 ? 0 : (u >= n_rows ? n_rows - 1 : u);        // a special symbol inside the valid range cannot index the table
      v = *reinterpret_cast<const uint4*>(table + (int64_t)u * C + cc * 8);
    }
    *reinterpret_cast<uint4*>(y + r * ldy + cc * 8) = v;
  }
}

// time-major fp32 rows (the mel head's [B, 4T, 80] output) -> 16-bit columns col0.. of the vocoder's concat buffer
template <typename ET>
__global__ void rows_to16_masked_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ y, int ldy, int col0,
                                        const int32_t* __restrict__ lens, int len_mul, int B, int T, int C) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * T * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t r = i / c4;
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    uint2 q = make_uint2(0, 0);
    if (!lens || t < lens[b] * len_mul) {
      const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + cc * 4);
      q.x = ET::pack2(v.x, v.y);
      q.y = ET::pack2(v.z, v.w);
    }
    *reinterpret_cast<uint2*>(y + r * ldy + col0 + cc * 4) = q;
  }
}

// sequence_generator.py:64-65: src_lengths = T - padding_mask.sum(-1); no mask = every clip is T frames long
__global__ void lens_from_mask_kernel(const uint8_t* __restrict__ mask, int32_t* __restrict__ lens, int B, int T) {
  const int b = blockIdx.x;
  int n = 0;
  if (mask)
    for (int t = threadIdx.x; t < T; t += 64) n += mask[(int64_t)b * T + t] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
  if (threadIdx.x == 0) lens[b] = T - n;
}

// Reference-precision vocoder switch (the reference's vocoder runs fp32, multi_input_vocoder/inference.py:73-82): an fp32
// activation, after its elementwise activation and row mask, as the sum of two 16-bit arrays hi + lo (lo = the rounding
// residue of hi).  Three 16-bit MFMA GEMMs per layer (a_hi w_hi + a_lo w_hi + a_hi w_lo, fp32 accumulate) then carry 2 x the
// 16-bit mantissa through every product: 22 bits with fp16 operands, 16 with bf16.  ACT: 0 none, 1 leaky_relu(slope), 2 GELU (erf).
template <typename ET, int ACT>
__global__ void split_hi_lo_kernel(const float* __restrict__ x, int ldx, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                   int ld16, const int32_t* __restrict__ lens, int len_mul, int B, int T, int C, float slope) {
  const int c4 = C >> 2;
  const int64_t total = (int64_t)B * T * c4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c4);
    const int64_t r = i / c4;
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (!lens || t < lens[b] * len_mul) {
      const float4 q = *reinterpret_cast<const float4*>(x + r * ldx + cc * 4);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ACT == 1) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope;
        if (ACT == 2) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
      }
    }
    uint16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = ET::from_f32(v[e]);
      l[e] = ET::from_f32(v[e] - ET::to_f32(h[e]));
    }
    *reinterpret_cast<uint2*>(hi + r * ld16 + cc * 4) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    *reinterpret_cast<uint2*>(lo + r * ld16 + cc * 4) = make_uint2((uint32_t)l[0] | ((uint32_t)l[1] << 16), (uint32_t)l[2] | ((uint32_t)l[3] << 16));
  }
}

// speech-resynthesis/models.py:110-112 + multi_input_vocoder/inference.py:79-81
constexpr int CP_TILE = 256;
__global__ __launch_bounds__(256) void conv_post_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float bias, float* __restrict__ wav, int16_t* __restrict__ pcm,
                                                        const int32_t* __restrict__ lens, int len_mul, int T, int C,
                                                        int k) {
  extern __shared__ float sm[];  // [(CP_TILE + k - 1)][C + 1] then w[k*C]
  const int b = blockIdx.y, t0 = blockIdx.x * CP_TILE, half = (k - 1) / 2;
  const int rows = CP_TILE + k - 1, ldc = C + 1;
  float* sw = sm + rows * ldc;
  const int lim = lens ? min(lens[b] * len_mul, T) : T;
  for (int i = threadIdx.x; i < rows * C; i += 256) {
    const int r = i / C, c = i - r * C;
    const int t = t0 + r - half;
    float v = 0.f;
    if (t >= 0 && t < lim) {
      v = x[((int64_t)b * T + t) * C + c];
      v = v >= 0.f ? v : v * 0.01f;  // F.leaky_relu default slope (models.py:110)
    }
    sm[r * ldc + c] = v;
  }
  for (int i = threadIdx.x; i < k * C; i += 256) sw[i] = w[i];
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t >= T) return;
  float acc = bias;
  for (int tap = 0; tap < k; ++tap) {
    const float* xr = sm + (threadIdx.x + tap) * ldc;
    const float* wr = sw + tap * C;
    for (int c = 0; c < C; ++c) acc += xr[c] * wr[c];
  }
  float o = tanhf(acc);
  if (t >= lim) o = 0.f;
  wav[(int64_t)b * T + t] = o;
  if (pcm) pcm[(int64_t)b * T + t] = (int16_t)(o * 32768.0f);  // astype('int16'): truncation toward zero, |o| < 1
}

// The generator's own shape (upsample_initial_channel / 2^5 = 16 channels, conv_post kernel 7): the 1 KB-per-16-samples input
// is loaded and kept as float4, rows padded to 80 B so that the 7 x 4 ds_read_b128 of consecutive samples are conflict-free
// (the generic kernel above: 16 scalar global loads with an integer division each and 112 ds_read_b32 per sample).
__global__ __launch_bounds__(256) void conv_post16_kernel(const float* __restrict__ x, const float* __restrict__ w, float bias,
                                                          float* __restrict__ wav, int16_t* __restrict__ pcm,
                                                          const int32_t* __restrict__ lens, int len_mul, int T) {
  constexpr int C = 16, K = 7, HALF = 3, ROWS = CP_TILE + K - 1, LD = 20;
  __shared__ __attribute__((aligned(16))) float sm[ROWS * LD];
  __shared__ __attribute__((aligned(16))) float sw[K * C];
  const int b = blockIdx.y, t0 = blockIdx.x * CP_TILE;
  const int lim = lens ? min(lens[b] * len_mul, T) : T;
  for (int i = threadIdx.x; i < ROWS * 4; i += 256) {
    const int r = i >> 2, q = i & 3;
    const int t = t0 + r - HALF;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= 0 && t < lim) {
      v = *reinterpret_cast<const float4*>(x + ((int64_t)b * T + t) * C + q * 4);
      v.x = v.x >= 0.f ? v.x : v.x * 0.01f;  // F.leaky_relu default slope (models.py:110)
      v.y = v.y >= 0.f ? v.y : v.y * 0.01f;
      v.z = v.z >= 0.f ? v.z : v.z * 0.01f;
      v.w = v.w >= 0.f ? v.w : v.w * 0.01f;
    }
    *reinterpret_cast<float4*>(sm + r * LD + q * 4) = v;
  }
  if (threadIdx.x < K * C) sw[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const int t = t0 + threadIdx.x;
  if (t >= T) return;
  float acc = bias;                              // same summation order as the generic kernel: tap-major, channel-minor
#pragma unroll
  for (int tap = 0; tap < K; ++tap) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 xv = *reinterpret_cast<const float4*>(sm + (threadIdx.x + tap) * LD + q * 4);
      const float4 wv = *reinterpret_cast<const float4*>(sw + tap * C + q * 4);
      acc += xv.x * wv.x; acc += xv.y * wv.y; acc += xv.z * wv.z; acc += xv.w * wv.w;
    }
  }
  float o = tanhf(acc);
  if (t >= lim) o = 0.f;
  wav[(int64_t)b * T + t] = o;
  if (pcm) pcm[(int64_t)b * T + t] = (int16_t)(o * 32768.0f);  // astype('int16'): truncation toward zero, |o| < 1
}

}  // namespace

#define DISPATCH_ET(dtype, CALL_F16, CALL_BF16) \
  if ((dtype) == L2S_F16) { CALL_F16; } else if ((dtype) == L2S_BF16) { CALL_BF16; } else return L2S_EINVAL;

extern "C" int l2s_repeat2_cast(const float* x, void* y, int B, int T, int C, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0) return L2S_ESHAPE;
  if (C & 3) return L2S_EALIGN;
  const int64_t rows = (int64_t)B * T;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for(rows * (C >> 2), 256)), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((repeat2_cast_kernel<ElemF16>), g, blk, 0, st, x, (uint16_t*)y, rows, C),
              hipLaunchKernelGGL((repeat2_cast_kernel<ElemBF16>), g, blk, 0, st, x, (uint16_t*)y, rows, C));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

// Split-K tail: x[m, n] += sum over s < S of P[m, s*N + n], s in ascending order (deterministic), float4 per thread.
__global__ void splitk_reduce_kernel(const float* __restrict__ P, int ldp, int S, float* __restrict__ x, int ldx, int M, int N4) {
  const int64_t total = (int64_t)M * N4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % N4) * 4;
    const int64_t r = i / N4;
    float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float* pr = P + r * ldp + c;
    for (int s = 0; s < S; ++s) {
      const float4 q = *reinterpret_cast<const float4*>(pr + (int64_t)s * N4 * 4);
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    *reinterpret_cast<float4*>(x + r * ldx + c) = v;
  }
}

extern "C" int l2s_splitk_reduce(const float* P, int ldp, int S, float* x, int ldx, int M, int N, void* stream) {
  if (!P || !x) return L2S_EINVAL;
  if (M <= 0 || N <= 0 || S <= 0) return L2S_ESHAPE;
  if ((N & 3) || (ldp & 3) || (ldx & 3) || ((uintptr_t)P & 15) || ((uintptr_t)x & 15)) return L2S_EALIGN;
  if (ldp < S * N || ldx < N) return L2S_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)M * (N / 4), 256)), blk(256);
  hipLaunchKernelGGL(splitk_reduce_kernel, g, blk, 0, st, P, ldp, S, x, ldx, M, N / 4);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_cast_f32_to_16(const float* x, int ldx, void* y, int ldy, int M, int C, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (M <= 0 || C <= 0) return L2S_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)M * C, 256)), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((cast_to16_kernel<ElemF16>), g, blk, 0, st, x, ldx, (uint16_t*)y, ldy, (int64_t)M, C),
              hipLaunchKernelGGL((cast_to16_kernel<ElemBF16>), g, blk, 0, st, x, ldx, (uint16_t*)y, ldy, (int64_t)M, C));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_cast_16_to_f32(const void* x, int ldx, float* y, int ldy, int M, int C, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (M <= 0 || C <= 0) return L2S_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)M * C, 256)), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((cast_to32_kernel<ElemF16>), g, blk, 0, st, (const uint16_t*)x, ldx, y, ldy, (int64_t)M, C),
              hipLaunchKernelGGL((cast_to32_kernel<ElemBF16>), g, blk, 0, st, (const uint16_t*)x, ldx, y, ldy, (int64_t)M, C));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_broadcast_rows(const void* v, int ldv, void* y, int ldy, int col0, const int32_t* lens, int len_mul,
                                  int B, int T, int C, int v_is_f32, int dtype, void* stream) {
  if (!v || !y) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0 || col0 < 0) return L2S_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)B * T * C, 256)), blk(256);
  uint16_t* yp = (uint16_t*)y;
  if (v_is_f32) {
    DISPATCH_ET(dtype,
                hipLaunchKernelGGL((broadcast_rows_kernel<ElemF16, true>), g, blk, 0, st, v, ldv, yp, ldy, col0, lens, len_mul, B, T, C),
                hipLaunchKernelGGL((broadcast_rows_kernel<ElemBF16, true>), g, blk, 0, st, v, ldv, yp, ldy, col0, lens, len_mul, B, T, C));
  } else {
    DISPATCH_ET(dtype,
                hipLaunchKernelGGL((broadcast_rows_kernel<ElemF16, false>), g, blk, 0, st, v, ldv, yp, ldy, col0, lens, len_mul, B, T, C),
                hipLaunchKernelGGL((broadcast_rows_kernel<ElemBF16, false>), g, blk, 0, st, v, ldv, yp, ldy, col0, lens, len_mul, B, T, C));
  }
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_transpose_ct_to_tc(const float* x, void* y, int ldy, int col0, const int32_t* lens, int len_mul,
                                      int B, int C, int T, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0 || col0 < 0) return L2S_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  dim3 g((T + 31) / 32, (C + 31) / 32, B), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((transpose_ct_kernel<ElemF16>), g, blk, 0, st, x, (uint16_t*)y, ldy, col0, lens, len_mul, C, T),
              hipLaunchKernelGGL((transpose_ct_kernel<ElemBF16>), g, blk, 0, st, x, (uint16_t*)y, ldy, col0, lens, len_mul, C, T));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_embedding(const int32_t* code, const void* table, void* y, int ldy, const int32_t* lens, int B,
                             int L, int C, int dtype, void* stream) {
  if (!code || !table || !y) return L2S_EINVAL;
  if (B <= 0 || L <= 0 || C <= 0) return L2S_ESHAPE;
  if ((C & 7) || (ldy & 7)) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)B * L * (C >> 3), 256)), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((embedding_kernel<ElemF16>), g, blk, 0, st, code, (const uint16_t*)table, (uint16_t*)y, ldy, lens, B, L, C),
              hipLaunchKernelGGL((embedding_kernel<ElemBF16>), g, blk, 0, st, code, (const uint16_t*)table, (uint16_t*)y, ldy, lens, B, L, C));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_embedding_tokens(const int32_t* tok, int ldt, int token_offset, const void* table, int n_rows, void* y,
                                    int ldy, const int32_t* lens, int len_mul, int B, int L, int C, int dtype,
                                    void* stream) {
  if (!tok || !table || !y) return L2S_EINVAL;
  if (B <= 0 || L <= 0 || C <= 0 || n_rows <= 0 || ldt < L || len_mul <= 0) return L2S_ESHAPE;
  if ((C & 7) || (ldy & 7)) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)B * L * (C >> 3), 256)), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((embedding_tokens_kernel<ElemF16>), g, blk, 0, st, tok, ldt, token_offset, n_rows, (const uint16_t*)table, (uint16_t*)y, ldy, lens, len_mul, B, L, C),
              hipLaunchKernelGGL((embedding_tokens_kernel<ElemBF16>), g, blk, 0, st, tok, ldt, token_offset, n_rows, (const uint16_t*)table, (uint16_t*)y, ldy, lens, len_mul, B, L, C));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_rows_f32_to_16_masked(const float* x, int ldx, void* y, int ldy, int col0, const int32_t* lens,
                                         int len_mul, int B, int T, int C, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0 || col0 < 0 || len_mul <= 0) return L2S_ESHAPE;
  if ((C & 3) || (ldx & 3) || (ldy & 3) || (col0 & 3)) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)B * T * (C >> 2), 256)), blk(256);
  DISPATCH_ET(dtype,
              hipLaunchKernelGGL((rows_to16_masked_kernel<ElemF16>), g, blk, 0, st, x, ldx, (uint16_t*)y, ldy, col0, lens, len_mul, B, T, C),
              hipLaunchKernelGGL((rows_to16_masked_kernel<ElemBF16>), g, blk, 0, st, x, ldx, (uint16_t*)y, ldy, col0, lens, len_mul, B, T, C));
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_lens_from_mask(const uint8_t* mask, int32_t* lens, int B, int T, void* stream) {
  if (!lens) return L2S_EINVAL;
  if (B <= 0 || T <= 0) return L2S_ESHAPE;
  hipLaunchKernelGGL(lens_from_mask_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, mask, lens, B, T);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_split_hi_lo(const float* x, int ldx, void* hi, void* lo, int ld16, int act, float slope,
                               const int32_t* lens, int len_mul, int B, int T, int C, int dtype, void* stream) {
  if (!x || !hi || !lo) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0 || len_mul <= 0 || act < 0 || act > 2) return L2S_ESHAPE;
  if ((C & 3) || (ldx & 3) || (ld16 & 3)) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  dim3 g(grid_for((int64_t)B * T * (C >> 2), 256)), blk(256);
  uint16_t *h = (uint16_t*)hi, *l = (uint16_t*)lo;
#define L2S_SPLIT_CASE(A)                                                                                                   \
  DISPATCH_ET(dtype, hipLaunchKernelGGL((split_hi_lo_kernel<ElemF16, A>), g, blk, 0, st, x, ldx, h, l, ld16, lens, len_mul, B, T, C, slope), \
              hipLaunchKernelGGL((split_hi_lo_kernel<ElemBF16, A>), g, blk, 0, st, x, ldx, h, l, ld16, lens, len_mul, B, T, C, slope))
  if (act == 0) { L2S_SPLIT_CASE(0); } else if (act == 1) { L2S_SPLIT_CASE(1); } else { L2S_SPLIT_CASE(2); }
#undef L2S_SPLIT_CASE
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_conv_post_tanh(const float* x, const float* w, float bias, float* wav, int16_t* pcm,
                                  const int32_t* lens, int len_mul, int B, int T, int C, int k, void* stream) {
  if (!x || !w || !wav) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0 || k <= 0 || !(k & 1)) return L2S_ESHAPE;
  const size_t smem = ((size_t)(CP_TILE + k - 1) * (C + 1) + (size_t)k * C) * sizeof(float);
  if (smem > 64 * 1024) return L2S_EUNSUPPORTED;
  dim3 g((T + CP_TILE - 1) / CP_TILE, B), blk(256);
  if (C == 16 && k == 7 && !((uintptr_t)x & 15))
    hipLaunchKernelGGL(conv_post16_kernel, g, blk, 0, (hipStream_t)stream, x, w, bias, wav, pcm, lens, len_mul, T);
  else
    hipLaunchKernelGGL(conv_post_kernel, g, blk, smem, (hipStream_t)stream, x, w, bias, wav, pcm, lens, len_mul, T, C, k);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}
