// Tap-GEMM: the one dense-contraction kernel family of the path (Linear / Conv1d / Conv2d / ConvTranspose1d phases).
//
//   C[o(m), n] = epi( sum_tap sum_c A[src(m,tap), c] * W[n, tap*Cin + c] )
//
// gfx950 design: 256 threads = 4 waves, block tile BM x BN, K-step 32 (one v_mfma_f32_16x16x32 per 16x16 sub-tile),
// global -> VGPR -> LDS staging with two LDS buffers (one barrier per K-step, next tile's global loads in flight
// under the MFMAs), 64-byte LDS rows with a 2-bit XOR chunk swizzle that makes every ds_read_b128 fragment read
// bank-conflict free for the gfx950 b128 lane groups.  The MFMA is issued "swapped" (W rows as the A operand) so each
// lane ends with 4 consecutive output channels of one output row: bias / residual / store are 8- or 16-byte vectors.
#include "l2s_common.h"

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <typename ET, int BM, int BN, int WM_, int WN_, int MODE>
__global__ __launch_bounds__(256) void tapgemm_kernel(const l2s_gemm_desc p) {
  constexpr int WAVE_M = BM / WM_, WAVE_N = BN / WN_;
  constexpr int MI = WAVE_M / 16, NI = WAVE_N / 16;
  constexpr int A_PER_T = (BM * 4 + 255) / 256;
  constexpr int W_PER_T = (BN * 4 + 255) / 256;
  static_assert(WM_ * WN_ == 4, "4 waves");

  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * (BM + BN) * BK];
  constexpr int BUF = (BM + BN) * BK;  // elements per LDS buffer: A tile then W tile

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN_, wn = wave % WN_;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int grp = blockIdx.z;
  const uint16_t* __restrict__ A = (const uint16_t*)p.A + grp * p.a_gstride;
  const uint16_t* __restrict__ W = (const uint16_t*)p.W + (int64_t)grp * p.w_gstride;
  const int Cin = p.Cin;
  const int Ktot = Cin * p.ntaps;
  const int nk = (Ktot + BK - 1) / BK;
  const float inv_cin = 1.0f / (float)Cin;

  // ---- per-thread staging assignment -------------------------------------------------------------------------
  const int sc = tid & 3;   // 16-byte chunk within the 64-byte K row
  const int sr = tid >> 2;  // row 0..63 (+64 for the second chunk)
  bool a_ok[A_PER_T];
  int64_t a_base[A_PER_T];  // LINEAR: src row ; CONV: clip/img base row
  int a_t[A_PER_T], a_x[A_PER_T];
#pragma unroll
  for (int j = 0; j < A_PER_T; ++j) {
    const int r = sr + j * 64;
    const int m = m0 + r;
    a_ok[j] = (r < BM) && (m < p.M);
    a_base[j] = 0; a_t[j] = 0; a_x[j] = 0;
    if (a_ok[j]) {
      if (MODE == L2S_MODE_LINEAR) {
        a_base[j] = m;
      } else if (MODE == L2S_MODE_CONV1D) {
        const int b = m / p.T_out, t = m - b * p.T_out;
        a_base[j] = (int64_t)b * p.T_in;
        a_t[j] = t * p.stride + p.off;
      } else {
        const int hw = p.Ho * p.Wo;
        const int img = m / hw, rem = m - img * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_base[j] = (int64_t)img * p.Hi * p.Wi;
        a_t[j] = oy * p.stride - p.pad;
        a_x[j] = ox * p.stride - p.pad;
      }
    }
  }
  bool w_ok[W_PER_T];
#pragma unroll
  for (int j = 0; j < W_PER_T; ++j) {
    const int r = sr + j * 64;
    w_ok[j] = (r < BN) && (n0 + r < p.N);
  }

  uint4 ra[A_PER_T], rw[W_PER_T];
  auto g_load = [&](int kt) {
    const int kk = kt * BK + sc * 8;
    const bool kok = kk < Ktot;
    int tap = 0, cc = kk;
    if (MODE != L2S_MODE_LINEAR) {
      tap = (int)(((float)kk + 0.5f) * inv_cin);
      cc = kk - tap * Cin;
    }
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      bool ok = a_ok[j] && kok;
      int64_t src = a_base[j];
      if (MODE == L2S_MODE_CONV1D) {
        const int st = a_t[j] + tap * p.dil;
        ok = ok && (st >= 0) && (st < p.T_in);
        src += st;
      } else if (MODE == L2S_MODE_CONV2D) {
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        const int iy = a_t[j] + ky, ix = a_x[j] + kx;
        ok = ok && (iy >= 0) && (iy < p.Hi) && (ix >= 0) && (ix < p.Wi);
        src += (int64_t)iy * p.Wi + ix;
      }
      ra[j] = ok ? *reinterpret_cast<const uint4*>(A + src * p.lda + cc) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < W_PER_T; ++j) {
      const int r = sr + j * 64;
      rw[j] = (w_ok[j] && kok) ? *reinterpret_cast<const uint4*>(W + (int64_t)(n0 + r) * Ktot + kk)
                               : make_uint4(0, 0, 0, 0);
    }
  };
  auto s_store = [&](int buf) {
#pragma unroll
    for (int j = 0; j < A_PER_T; ++j) {
      const int r = sr + j * 64;
      if (r < BM) *reinterpret_cast<uint4*>(lds + buf * BUF + r * BK + ((sc ^ swz(r)) << 3)) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < W_PER_T; ++j) {
      const int r = sr + j * 64;
      if (r < BN) *reinterpret_cast<uint4*>(lds + buf * BUF + BM * BK + r * BK + ((sc ^ swz(r)) << 3)) = rw[j];
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int lm = lane & 15, lg = lane >> 4;
  const int frag_off = lm * BK + ((lg ^ swz(lm)) << 3);

  g_load(0);
  s_store(0);
  __syncthreads();
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) g_load(kt + 1);
    frag16 fa[MI], fw[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
      fa[i].u = *reinterpret_cast<const uint4*>(lds + cur * BUF + (wm * WAVE_M + i * 16) * BK + frag_off);
#pragma unroll
    for (int j = 0; j < NI; ++j)
      fw[j].u = *reinterpret_cast<const uint4*>(lds + cur * BUF + BM * BK + (wn * WAVE_N + j * 16) * BK + frag_off);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw[j], fa[i], acc[i][j]);
    if (kt + 1 < nk) s_store(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: lane holds rows n = 4*lg..4*lg+3 of column m = lm of every 16x16 sub-tile -------------------
  const int flags = p.flags;
  const int ncol_g = grp * p.c_gstride;
  const float* bias = p.bias ? p.bias + grp * p.N : nullptr;
  const float* slope = p.slope ? p.slope + grp * p.N : nullptr;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = m0 + wm * WAVE_M + i * 16 + lm;
    if (m >= p.M) continue;
    const int64_t o = (int64_t)m * p.out_row_mul + p.out_row_add;
    bool keep = true;
    if (flags & L2S_F_MASK) {
      const int clip = (int)(o / p.mask_T);
      const int t = (int)(o - (int64_t)clip * p.mask_T);
      keep = t < p.lens[clip] * p.mask_mul;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int n = n0 + wn * WAVE_N + j * 16 + lg * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
      const int col = ncol_g + n;
      float rr[4] = {0.f, 0.f, 0.f, 0.f};
      if (flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) {
        if (flags & L2S_F_RES_F32) {
          const float4 q = *reinterpret_cast<const float4*>((const float*)p.R + o * p.ldr + col);
          rr[0] = q.x; rr[1] = q.y; rr[2] = q.z; rr[3] = q.w;
        } else {
          const uint2 q = *reinterpret_cast<const uint2*>((const uint16_t*)p.R + o * p.ldr + col);
          rr[0] = ET::to_f32((uint16_t)(q.x & 0xffff)); rr[1] = ET::to_f32((uint16_t)(q.x >> 16));
          rr[2] = ET::to_f32((uint16_t)(q.y & 0xffff)); rr[3] = ET::to_f32((uint16_t)(q.y >> 16));
        }
      }
      if (flags & L2S_F_RES_PRE) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rr[r];
      }
      switch (p.act) {
        case L2S_ACT_RELU:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
          break;
        case L2S_ACT_GELU:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = l2s_gelu(v[r]);
          break;
        case L2S_ACT_SWISH:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = l2s_swish(v[r]);
          break;
        case L2S_ACT_PRELU: {
          const float4 s = *reinterpret_cast<const float4*>(slope + n);
          v[0] = v[0] >= 0.f ? v[0] : v[0] * s.x; v[1] = v[1] >= 0.f ? v[1] : v[1] * s.y;
          v[2] = v[2] >= 0.f ? v[2] : v[2] * s.z; v[3] = v[3] >= 0.f ? v[3] : v[3] * s.w;
        } break;
        case L2S_ACT_LRELU:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = v[r] >= 0.f ? v[r] : v[r] * p.act_slope;
          break;
        case L2S_ACT_TANH:
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
          break;
        default: break;
      }
      if (flags & L2S_F_RES_POST) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rr[r];
      }
      if (flags & L2S_F_ACCUM) {
        if (flags & L2S_F_OUT_F32) {
          const float4 q = *reinterpret_cast<const float4*>((const float*)p.C + o * p.ldc + col);
          v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w;
        } else {
          const uint2 q = *reinterpret_cast<const uint2*>((const uint16_t*)p.C + o * p.ldc + col);
          v[0] += ET::to_f32((uint16_t)(q.x & 0xffff)); v[1] += ET::to_f32((uint16_t)(q.x >> 16));
          v[2] += ET::to_f32((uint16_t)(q.y & 0xffff)); v[3] += ET::to_f32((uint16_t)(q.y >> 16));
        }
      }
      if (!keep) { v[0] = v[1] = v[2] = v[3] = 0.f; }
      if (flags & L2S_F_OUT_F32) {
        *reinterpret_cast<float4*>((float*)p.C + o * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        uint2 q;
        q.x = (uint32_t)ET::from_f32(v[0]) | ((uint32_t)ET::from_f32(v[1]) << 16);
        q.y = (uint32_t)ET::from_f32(v[2]) | ((uint32_t)ET::from_f32(v[3]) << 16);
        *reinterpret_cast<uint2*>((uint16_t*)p.C + o * p.ldc + col) = q;
      }
      if (flags & L2S_F_DUAL) {
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = v[r] >= 0.f ? v[r] : v[r] * p.slope2;
        uint2 q;
        q.x = (uint32_t)ET::from_f32(w[0]) | ((uint32_t)ET::from_f32(w[1]) << 16);
        q.y = (uint32_t)ET::from_f32(w[2]) | ((uint32_t)ET::from_f32(w[3]) << 16);
        *reinterpret_cast<uint2*>((uint16_t*)p.C2 + o * p.ldc2 + col) = q;
      }
    }
  }
}

// tile selection shared by the launcher and l2s_tapgemm_variant(): returns BM*1000 + BN
inline int pick_tile(int M, int N, int G) {
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  if (N <= 16) return 128016;
  if (N <= 32) return 128032;
  const long b128 = (long)cdiv(M, 128) * cdiv(N, 128) * G;
  const long b64 = (long)cdiv(M, 128) * cdiv(N, 64) * G;
  if (N >= 128 && b128 >= 512) return 128128;
  if (b64 >= 256) return 128064;
  return 64064;
}

template <typename ET, int MODE>
int launch_mode(const l2s_gemm_desc& d, hipStream_t st) {
  const int M = d.M, N = d.N, G = d.groups > 0 ? d.groups : 1;
  auto cdiv = [](int a, int b) { return (a + b - 1) / b; };
  switch (pick_tile(M, N, G)) {
    case 128016:
      hipLaunchKernelGGL((tapgemm_kernel<ET, 128, 16, 4, 1, MODE>), dim3(cdiv(M, 128), 1, G), dim3(256), 0, st, d);
      break;
    case 128032:
      hipLaunchKernelGGL((tapgemm_kernel<ET, 128, 32, 4, 1, MODE>), dim3(cdiv(M, 128), 1, G), dim3(256), 0, st, d);
      break;
    case 128128:
      hipLaunchKernelGGL((tapgemm_kernel<ET, 128, 128, 2, 2, MODE>), dim3(cdiv(M, 128), cdiv(N, 128), G), dim3(256), 0, st, d);
      break;
    case 128064:
      hipLaunchKernelGGL((tapgemm_kernel<ET, 128, 64, 2, 2, MODE>), dim3(cdiv(M, 128), cdiv(N, 64), G), dim3(256), 0, st, d);
      break;
    default:
      hipLaunchKernelGGL((tapgemm_kernel<ET, 64, 64, 2, 2, MODE>), dim3(cdiv(M, 64), cdiv(N, 64), G), dim3(256), 0, st, d);
      break;
  }
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET>
int launch_dtype(const l2s_gemm_desc& d, hipStream_t st) {
  switch (d.mode) {
    case L2S_MODE_LINEAR: return launch_mode<ET, L2S_MODE_LINEAR>(d, st);
    case L2S_MODE_CONV1D: return launch_mode<ET, L2S_MODE_CONV1D>(d, st);
    case L2S_MODE_CONV2D: return launch_mode<ET, L2S_MODE_CONV2D>(d, st);
    default: return L2S_EINVAL;
  }
}

}  // namespace

extern "C" int l2s_tapgemm(const l2s_gemm_desc* hd, void* stream) {
  if (!hd) return L2S_EINVAL;
  l2s_gemm_desc d = *hd;
  if (!d.A || !d.W || !d.C) return L2S_EINVAL;
  if (d.M <= 0 || d.N <= 0 || d.Cin <= 0 || d.ntaps <= 0) return L2S_ESHAPE;
  if (d.groups <= 0) d.groups = 1;
  if (d.out_row_mul <= 0) d.out_row_mul = 1;
  // 16-byte vector loads of A/W chunks, 8/16-byte vector epilogue
  if ((d.Cin & 7) || (d.lda & 7) || (d.N & 3) || (d.ldc & 3) || (d.a_gstride & 7) || (d.c_gstride & 3) ||
      (d.w_gstride & 7))
    return L2S_EALIGN;
  if (((uintptr_t)d.A & 15) || ((uintptr_t)d.W & 15) || ((uintptr_t)d.C & 15)) return L2S_EALIGN;
  if ((d.flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) && (!d.R || (d.ldr & 3))) return L2S_EINVAL;
  if ((d.flags & L2S_F_DUAL) && (!d.C2 || (d.ldc2 & 3))) return L2S_EINVAL;
  if ((d.flags & L2S_F_MASK) && (!d.lens || d.mask_T <= 0 || d.mask_mul <= 0)) return L2S_EINVAL;
  if (d.act == L2S_ACT_PRELU && !d.slope) return L2S_EINVAL;
  if (d.mode == L2S_MODE_CONV1D && (d.T_out <= 0 || d.T_in <= 0)) return L2S_ESHAPE;
  if (d.mode == L2S_MODE_CONV2D && (d.Ho <= 0 || d.Wo <= 0 || d.Hi <= 0 || d.Wi <= 0 || d.KW <= 0)) return L2S_ESHAPE;
  if (d.mode != L2S_MODE_LINEAR && (int64_t)d.Cin * d.ntaps > (1 << 15)) return L2S_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (d.dtype == L2S_F16) return launch_dtype<ElemF16>(d, st);
  if (d.dtype == L2S_BF16) return launch_dtype<ElemBF16>(d, st);
  return L2S_EINVAL;
}

extern "C" int l2s_tapgemm_variant(const l2s_gemm_desc* hd) {
  if (!hd || hd->M <= 0 || hd->N <= 0) return L2S_EINVAL;
  return pick_tile(hd->M, hd->N, hd->groups > 0 ? hd->groups : 1);
}
