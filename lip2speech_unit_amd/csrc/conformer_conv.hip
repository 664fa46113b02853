// Conformer convolution-module core between the two pointwise GEMMs (espnet convolution.py:57-62):
//   GLU over channels -> depthwise Conv1d(k=31, pad 15) -> BatchNorm1d (eval, folded into w/bias) -> Swish.
// One block = TT (128 or 100) time steps x 64 channels of one clip; the GLU'd halo tile lives in LDS, each thread slides
// a (TT/4 + 30)-sample register window over TT/4 outputs of one channel (lane = channel: conflict-free LDS reads).
// VALU-side bound (31 FMAs per output + the GLU's exp / reciprocal), so the sigmoid uses v_rcp_f32 instead of a full
// division.  The tile length is picked per launch to waste the fewest computed rows: the path's 4-s clips have T = 200,
// which two 100-step tiles cover exactly (two 128-step tiles compute 56 rows for nothing: 21 % of the kernel).
#include "l2s_common.h"

namespace {

constexpr int CT = 64, KMAX = 31;   // 128 steps per block: the 30-row halo costs 1.23x instead of 1.47x at 64

template <typename ET, int TT>
__global__ __launch_bounds__(256) void glu_dwconv_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, uint16_t* __restrict__ y,
                                                         const int32_t* __restrict__ lens, int len_mul, int T, int C,
                                                         int k) {
  constexpr int OPT = TT / 4;                   // outputs per thread
  __shared__ float g[(TT + KMAX - 1) * CT];
  const int b = blockIdx.z, t0 = blockIdx.x * TT, c0 = blockIdx.y * CT;
  const int half = (k - 1) / 2, rows = TT + k - 1;
  int lim = lens ? lens[b] * len_mul : T;
  lim = lim < T ? lim : T;
  // stage GLU(a, gate) = a * sigmoid(gate); rows outside [0, lim) are the conv's zero padding
  for (int i = threadIdx.x; i < rows * (CT / 8); i += 256) {
    const int r = i / (CT / 8), ch = i - r * (CT / 8);
    const int t = t0 + r - half;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t >= 0 && t < lim) {
      const uint16_t* rp = x + ((int64_t)b * T + t) * (2 * C) + c0 + ch * 8;
      frag16 a, gt;
      a.u = *reinterpret_cast<const uint4*>(rp);
      gt.u = *reinterpret_cast<const uint4*>(rp + C);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = ET::to_f32(a.s[e]) * __builtin_amdgcn_rcpf(1.0f + __expf(-ET::to_f32(gt.s[e])));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) g[r * CT + ch * 8 + e] = o[e];
  }
  __syncthreads();
  const int c = threadIdx.x & 63, tq = threadIdx.x >> 6;  // 4 groups x OPT outputs
  float wr[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) wr[j] = j < k ? w[j * C + c0 + c] : 0.f;
  const float bs = bias[c0 + c];
  float win[OPT + KMAX - 1];
#pragma unroll
  for (int j = 0; j < OPT + KMAX - 1; ++j) win[j] = (tq * OPT + j < rows) ? g[(tq * OPT + j) * CT + c] : 0.f;
#pragma unroll
  for (int o = 0; o < OPT; ++o) {
    const int t = t0 + tq * OPT + o;
    float acc = bs;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) acc += win[o + j] * wr[j];
    acc = l2s_swish(acc);
    if (t < T) y[((int64_t)b * T + t) * C + c0 + c] = ET::from_f32(t < lim ? acc : 0.f);
  }
}

}  // namespace

extern "C" int l2s_glu_dwconv_swish(const void* x, const float* w, const float* bias, void* y, const int32_t* lens,
                                    int len_mul, int B, int T, int C, int k, int dtype, void* stream) {
  if (!x || !w || !bias || !y) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || C <= 0) return L2S_ESHAPE;
  if (k <= 0 || k > KMAX || !(k & 1)) return L2S_EUNSUPPORTED;
  if (C % CT) return L2S_EALIGN;
  // rows computed = tiles * TT: take the tile length that wastes fewer of them (ties: the longer tile, less halo)
  const int r128 = ((T + 127) / 128) * 128, r100 = ((T + 99) / 100) * 100;
  const bool use100 = r100 + ((T + 99) / 100) * 10 < r128 + ((T + 127) / 128) * 10;
  const int TTr = use100 ? 100 : 128;
  dim3 grid((T + TTr - 1) / TTr, C / CT, B), blk(256);
  hipStream_t st = (hipStream_t)stream;
#define GLU_LAUNCH(ET_, TT_) hipLaunchKernelGGL((glu_dwconv_kernel<ET_, TT_>), grid, blk, 0, st, (const uint16_t*)x, w, bias, (uint16_t*)y, lens, len_mul, T, C, k)
  if (dtype == L2S_F16) { if (use100) GLU_LAUNCH(ElemF16, 100); else GLU_LAUNCH(ElemF16, 128); }
  else if (dtype == L2S_BF16) { if (use100) GLU_LAUNCH(ElemBF16, 100); else GLU_LAUNCH(ElemBF16, 128); }
  else return L2S_EINVAL;
#undef GLU_LAUNCH
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}
