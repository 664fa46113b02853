// Fused HiFi-GAN ResBlock1 for the narrow vocoder stages (C = 16 / 32 channels at 32 000 / 64 000 samples per clip).
//   speech-resynthesis/models.py:34-41: for (c1,c2,d) in zip(convs1, convs2, (1,3,5)):
//       xt = c2(leaky_relu(c1(leaky_relu(x)), 0.1)) ; x = xt + x
//   and :103-109: xs = sum over the three ResBlocks (k = 3, 7, 11) of one stage.
// As six separate tap-GEMMs these stages are HBM-bound: every conv re-reads and re-writes a 65 MB activation (~19 passes
// per stage).  Here one block keeps a time tile (+ halo) of the activation in LDS, runs all six convolutions of a
// ResBlock on MFMA straight out of LDS and touches HBM twice: one read of leaky_relu(x), one accumulate into the fp32 xs.
//
// LDS holds XL = leaky_relu(x) (the conv input) and T1 = leaky_relu(c1(.)); the residual x is recovered from XL by the
// exact inverse of leaky_relu (x = xl >= 0 ? xl : xl / slope), so no second copy of x is kept.  Rows outside the clip
// are zero in both buffers (the reference's zero padding); rows whose receptive field leaves the tile hold finite
// garbage that never reaches the tile's own output rows (halo = sum of the six receptive half-widths).
#include "l2s_common.h"

namespace {

constexpr int RB_GUARD = 32;  // zero guard rows on each side: the widest single tap offset is 5*5 = 25 rows
// Row stride (elements) and time tile: dense 32-byte rows make the C=16 fragment reads conflict-free and let three blocks
// share a CU; C=32, k = 3 / 7 use dense 64-byte rows (2-way conflicts, measured irrelevant) and a 384 / 368-sample tile so that two
// blocks fit the 160 KB of LDS - the second block hides the first one's LDS/VALU phases; for k = 11 the halo (2 x 60
// rows) makes the small tile a loss, it keeps 512 samples, padded 96-byte rows (conflict-free) and one block per CU.
template <int C, int K> constexpr int RB_RS() { return C == 16 ? 16 : (K == 11 ? 48 : 32); }
template <int C, int K> constexpr int RB_TT() { return (C == 32 && K != 11) ? (K == 7 ? 368 : 384) : 512; }

// waves per SIMD the register allocation must allow: three blocks per CU at C = 16, two at C = 32 (k <= 7), one at k = 11
template <int C, int K> constexpr int RB_WPE() { return C == 16 ? 6 : (K == 11 ? 2 : 4); }

template <typename ET, int C, int K>
__global__ __launch_bounds__(512, (RB_WPE<C, K>())) void resblock_kernel(const uint16_t* __restrict__ xl_in,
                                                       const uint16_t* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ xs, uint16_t* __restrict__ xl_out,
                                                       const int32_t* __restrict__ lens, int len_mul, int T, int TT,
                                                       int d0, int d1, int d2, int accumulate, float slope) {
  constexpr int HALF = (K - 1) / 2;
  constexpr int RS = RB_RS<C, K>();                   // row stride in elements
  constexpr int NI = C / 16;
  constexpr int KPAD = ((K * C + 31) / 32) * 32;
  constexpr int STEPS = KPAD / 32;
  constexpr int NW = 8;
  extern __shared__ __attribute__((aligned(16))) uint16_t sm[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, lg = lane >> 4;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * TT;
  const int H = HALF * (d0 + d1 + d2 + 3);
  const int R = ((TT + 2 * H + 15) / 16) * 16;     // rows computed by every conv
  const int RB = R + 2 * RB_GUARD;
  uint16_t* XL = sm;
  uint16_t* T1 = sm + RB * RS;
  // Conv weights reach the waves' registers through LDS: read straight from global, the eight waves of a block pull
  // 8 x C*KPAD*2 bytes through the CU's 64 B/clk vector-memory path after every conv's barrier (13-21 % of the kernel);
  // here the block fetches the NEXT conv's C*KPAD*2 bytes once, into registers while the current conv computes, and
  // the waves fill their fragments from LDS.  16-byte chunk c of weight row n sits at chunk (c & ~3) | ((c & 3) ^
  // ((n >> 2) & 3)): the row stride is 48 banks (mod 64) for C = 32, so rows n, n+4, n+8, n+12 would collide.
  uint16_t* WB = sm + 2 * RB * RS;
  constexpr int CPR = KPAD / 8;                        // 16-byte chunks per weight row
  constexpr int WCH = C * CPR;                         // chunks per conv
  constexpr int WPT = (WCH + 511) / 512;               // chunks per thread
  uint4 wpre[WPT];
  auto w_fetch = [&](int cv) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int l = tid + i * 512;
      wpre[i] = l < WCH ? *reinterpret_cast<const uint4*>(w + (int64_t)cv * C * KPAD + l * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto w_commit = [&]() {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int l = tid + i * 512;
      const int n = l / CPR, c = l - n * CPR;
      if (l < WCH) *reinterpret_cast<uint4*>(WB + (n * CPR + ((c & ~3) | ((c & 3) ^ ((n >> 2) & 3)))) * 8) = wpre[i];
    }
  };
  w_fetch(0);
  int lim = lens ? lens[b] * len_mul : T;
  lim = lim < T ? lim : T;
  const int g0 = t0 - H - RB_GUARD;               // global time of buffer row 0
  const float inv_slope = 1.0f / slope;

  // ---- load XL tile (zero outside the clip), clear T1 guards ----
  for (int i = tid; i < RB * (C / 8); i += 512) {
    const int rb = i / (C / 8), ch = i - rb * (C / 8);
    const int t = g0 + rb;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t >= 0 && t < lim) v = *reinterpret_cast<const uint4*>(xl_in + ((int64_t)b * T + t) * C + ch * 8);
    *reinterpret_cast<uint4*>(XL + rb * RS + ch * 8) = v;
    if (rb < RB_GUARD || rb >= RB - RB_GUARD) *reinterpret_cast<uint4*>(T1 + rb * RS + ch * 8) = make_uint4(0, 0, 0, 0);
  }
  w_commit();
  __syncthreads();

#pragma unroll 1
  for (int cv = 0; cv < 6; ++cv) {
    const uint16_t* src = (cv & 1) ? T1 : XL;
    const int d = (cv & 1) ? 1 : (cv == 0 ? d0 : (cv == 2 ? d1 : d2));
    // weights of this conv -> registers (MFMA A operand: lane = output channel lm of N-tile ni, k-chunk lg)
    frag16 wf[NI][STEPS];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int s = 0; s < STEPS; ++s)
        wf[ni][s].u = *reinterpret_cast<const uint4*>(WB + ((ni * 16 + lm) * CPR + s * 4 + (lg ^ ((lm >> 2) & 3))) * 8);
    float4 bs[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bs[ni] = *reinterpret_cast<const float4*>(bias + cv * C + ni * 16 + lg * 4);
    __syncthreads();                                   // every wave holds its fragments: the weight buffer is free
    if (cv < 5) w_fetch(cv + 1);                       // in flight during this conv
    // per-lane source offset of each k-step: K index = tap*C + c
    int aoff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      int tap, chunk;
      if (C == 32) { tap = s; chunk = lg; } else { tap = 2 * s + (lg >> 1); chunk = lg & 1; }
      tap = tap < K ? tap : K - 1;  // K padding: weights are zero there, keep the address in range
      aoff[s] = ((tap - HALF) * d) * RS + chunk * 8;
    }
    for (int grp = wave; grp < R / 16; grp += NW) {
      const int r = grp * 16 + lm;                 // tile row (0..R) of this lane's output
      const uint16_t* ap = src + (RB_GUARD + r) * RS;
      f32x4_t acc[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        frag16 fa;
        fa.u = *reinterpret_cast<const uint4*>(ap + aoff[s]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[ni] = ET::mfma(wf[ni][s], fa, acc[ni]);
      }
      const int t = t0 - H + r;                    // global time of the row
      const bool inclip = (t >= 0) && (t < lim);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = ni * 16 + lg * 4;
        float v[4] = {acc[ni][0] + bs[ni].x, acc[ni][1] + bs[ni].y, acc[ni][2] + bs[ni].z, acc[ni][3] + bs[ni].w};
        if (!(cv & 1)) {
          // c1: T1 = leaky_relu(conv + b)
          uint2 q = make_uint2(0, 0);
          if (inclip) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * slope;
            q.x = ET::pack2(v[0], v[1]);
            q.y = ET::pack2(v[2], v[3]);
          }
          *reinterpret_cast<uint2*>(T1 + (RB_GUARD + r) * RS + n) = q;
        } else {
          // c2: x = conv + b + x, residual recovered from XL = leaky_relu(x)
          uint16_t* xp = XL + (RB_GUARD + r) * RS + n;
          const uint2 q0 = *reinterpret_cast<const uint2*>(xp);
          const float xr[4] = {ET::to_f32((uint16_t)(q0.x & 0xffff)), ET::to_f32((uint16_t)(q0.x >> 16)),
                               ET::to_f32((uint16_t)(q0.y & 0xffff)), ET::to_f32((uint16_t)(q0.y >> 16))};
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = inclip ? v[e] + (xr[e] >= 0.f ? xr[e] : xr[e] * inv_slope) : 0.f;
          if (cv < 5) {
            uint2 q;
            float l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) l[e] = v[e] >= 0.f ? v[e] : v[e] * slope;
            q.x = ET::pack2(l[0], l[1]);
            q.y = ET::pack2(l[2], l[3]);
            *reinterpret_cast<uint2*>(xp) = q;
          } else if (r >= H && r < H + TT && t < T) {
            // ResBlock output for the tile's own rows: accumulate into xs (models.py:105-108)
            float* op = xs + ((int64_t)b * T + t) * C + n;
            if (accumulate) {
              const float4 o = *reinterpret_cast<const float4*>(op);
              v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
            }
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            if (xl_out) {
              float l[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) l[e] = v[e] >= 0.f ? v[e] : v[e] * slope;
              uint2 q;
              q.x = ET::pack2(l[0], l[1]);
              q.y = ET::pack2(l[2], l[3]);
              *reinterpret_cast<uint2*>(xl_out + ((int64_t)b * T + t) * C + n) = q;
            }
          }
        }
      }
    }
    if (cv < 5) w_commit();
    __syncthreads();
  }
}

template <typename ET, int C, int K>
int launch_rb(const void* xl, const void* w, const float* bias, float* xs, void* xl_out, const int32_t* lens,
              int len_mul, int B, int T, int d0, int d1, int d2, int accumulate, float slope, hipStream_t st) {
  constexpr int HALF = (K - 1) / 2;
  constexpr int RS = RB_RS<C, K>();
  const int H = HALF * (d0 + d1 + d2 + 3);
  const int TT = RB_TT<C, K>();
  const int R = ((TT + 2 * H + 15) / 16) * 16;
  const int RB = R + 2 * RB_GUARD;
  constexpr int KPAD = ((K * C + 31) / 32) * 32;
  const int smem = 2 * RB * RS * 2 + C * KPAD * 2;
  auto kern = resblock_kernel<ET, C, K>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  if (smem > 160 * 1024) return L2S_EUNSUPPORTED;
  dim3 grid((T + TT - 1) / TT, B);
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, (const uint16_t*)xl, (const uint16_t*)w, bias, xs,
                     (uint16_t*)xl_out, lens, len_mul, T, TT, d0, d1, d2, accumulate, slope);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET>
int dispatch_rb(int C, int K, const void* xl, const void* w, const float* bias, float* xs, void* xl_out,
                const int32_t* lens, int len_mul, int B, int T, int d0, int d1, int d2, int accumulate, float slope,
                hipStream_t st) {
#define RB_CASE(CC, KK) \
  if (C == CC && K == KK) return launch_rb<ET, CC, KK>(xl, w, bias, xs, xl_out, lens, len_mul, B, T, d0, d1, d2, accumulate, slope, st);
  RB_CASE(16, 3) RB_CASE(16, 7) RB_CASE(16, 11) RB_CASE(32, 3) RB_CASE(32, 7) RB_CASE(32, 11)
#undef RB_CASE
  return L2S_EUNSUPPORTED;
}

}  // namespace

extern "C" int l2s_resblock_fused(const void* xl, const void* w, const float* bias, float* xs, void* xl_out,
                                  const int32_t* lens, int len_mul, int B, int T, int C, int k, int d0, int d1, int d2,
                                  int accumulate, float slope, int dtype, void* stream) {
  if (!xl || !w || !bias || !xs) return L2S_EINVAL;
  if (B <= 0 || T <= 0) return L2S_ESHAPE;
  if (d0 < 1 || d1 < 1 || d2 < 1 || d0 > 5 || d1 > 5 || d2 > 5 || slope <= 0.f) return L2S_EUNSUPPORTED;
  if (lens && len_mul <= 0) return L2S_EINVAL;
  if (((uintptr_t)xl & 15) || ((uintptr_t)w & 15) || ((uintptr_t)xs & 15) || ((uintptr_t)xl_out & 7)) return L2S_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == L2S_F16)
    return dispatch_rb<ElemF16>(C, k, xl, w, bias, xs, xl_out, lens, len_mul, B, T, d0, d1, d2, accumulate, slope, st);
  if (dtype == L2S_BF16)
    return dispatch_rb<ElemBF16>(C, k, xl, w, bias, xs, xl_out, lens, len_mul, B, T, d0, d1, d2, accumulate, slope, st);
  return L2S_EINVAL;
}
