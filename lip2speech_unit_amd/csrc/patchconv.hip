// Patch convolution: stride-1 Conv1d / 3x3 Conv2d with Cin = 64 and N = 64 (vocoder stage C=64: speech-resynthesis/
// models.py:19-31 at 16 000 samples per clip; ResNet layer1: avhubert/resnet.py:61-74 at 22x22) - the wide-M, narrow-K
// convolutions where the generic tap-GEMM re-fetches every input row once per tap through L2.
//
// One block keeps the input PATCH of its 256 output positions (+ halo) in LDS, loaded once by LDS-DMA, and forms every
// tap's MFMA operand by reading the patch at a shifted row: HBM/L2 traffic per tile drops from k x (A + W) tiles to
// one patch + k small weight tiles.  Conv2d works in a padded-flattened position space (image (H+2) x (W+2), all images
// back to back): a tap is then a constant row shift, the zero border comes for free from the DMA's per-lane source
// address (border positions read the zero page) and border outputs are simply not stored.
// Weight tiles (64 x 64 per tap, two taps per stage) stream through a 3-deep ring behind a counted vmcnt; the patch is double buffered and
// the next tile's patch is fetched under the current tile's taps (persistent blocks).  Epilogue: tapgemm_common.h.
#include "tapgemm_common.h"

using namespace l2s;

namespace {

constexpr int PBM = 256;          // output positions per tile
constexpr int PROWS = 320;        // patch rows: 256 + halo (<= 64), multiple of 64
constexpr int PATCH_B = PROWS * 128;
constexpr int WST_B = 2 * 64 * 128;   // one ring stage: the weight tiles of two consecutive taps
constexpr int PNW = 8;            // waves: each owns 32 positions x 64 channels

template <typename ET, int MODE, int EPI>
__global__ __launch_bounds__(512) void patchconv64_kernel(const l2s_gemm_desc p, const int ntiles,
                                                          const int tiles_per_clip, const int lo_shift) {
  constexpr int MI = 2, NI = 4;
  constexpr int P_PER_W = PROWS / 8 / PNW;  // patch DMA instructions per wave (5)
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];  // [2 patches][3 weight stages]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, lg = lane >> 4;
  const int srow = lane >> 3, schunk = (lane & 7) ^ (srow & 7);
  const int ntaps = p.ntaps;
  const int Ktot = ntaps * 64;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);
  const uint16_t* A = (const uint16_t*)p.A;
  const uint16_t* W = (const uint16_t*)p.W;
  const int PW = p.Wi + 2, PH = p.Hi + 2;       // CONV2D padded image
  const int my_n = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_n <= 0) return;
  const int nsteps = (ntaps + 1) / 2;          // two taps per ring stage / barrier
  const int total = my_n * nsteps;

  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t wring = lds_base + 2 * PATCH_B;

  // first position of a tile; CONV1D: (clip, t0) ; CONV2D: flattened padded position Q0
  auto tile_origin = [&](int i, int& unit, int& q0) {
    const int L = blockIdx.x + i * gridDim.x;
    if (MODE == L2S_MODE_CONV1D) { unit = L / tiles_per_clip; q0 = (L - unit * tiles_per_clip) * PBM; }
    else { unit = 0; q0 = L * PBM; }
  };
  // global source pointer of patch row pr of a tile (or the zero page)
  auto patch_src = [&](int unit, int q0, int pr) -> const uint16_t* {
    if (MODE == L2S_MODE_CONV1D) {
      const int ts = q0 + lo_shift + pr;
      return ((unsigned)ts < (unsigned)p.T_in) ? A + ((int64_t)unit * p.T_in + ts) * p.lda + schunk * 8 : zero;
    } else {
      const int Q = q0 - (PW + 1) + pr;        // padded-flattened source position
      if (Q < 0) return zero;
      const int img = Q / (PH * PW), rem = Q - img * (PH * PW);
      const int py = rem / PW, px = rem - py * PW;
      const bool in = (py >= 1) && (py <= p.Hi) && (px >= 1) && (px <= p.Wi) && ((int64_t)img * p.Hi * p.Wi < (int64_t)p.M);
      return in ? A + (((int64_t)img * p.Hi + (py - 1)) * p.Wi + (px - 1)) * p.lda + schunk * 8 : zero;
    }
  };
  auto issue_patch = [&](int i, int buf) {
    int unit, q0;
    tile_origin(i, unit, q0);
#pragma unroll
    for (int j = 0; j < P_PER_W; ++j) {
      const int instr = wave * P_PER_W + j;
      const uint16_t* g = patch_src(unit, q0, instr * 8 + srow);
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + buf * (PATCH_B / 2) + instr * 512), 16, 0, 0);
    }
  };
  const uint16_t* w_ptr = W + (int64_t)(wave * 8 + srow) * Ktot + schunk * 8;
  auto issue_w = [&](int step, int ws) {     // always two DMAs per wave (a phantom second tap reads the zero page)
    const int t0 = 2 * step, t1 = 2 * step + 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(w_ptr + t0 * 64), (lptr_t)(lds + PATCH_B + ws * (WST_B / 2) + wave * 512),
                                     16, 0, 0);
    const uint16_t* g1 = t1 < ntaps ? w_ptr + t1 * 64 : zero;
    __builtin_amdgcn_global_load_lds((gptr_t)g1, (lptr_t)(lds + PATCH_B + ws * (WST_B / 2) + 4096 + wave * 512), 16, 0, 0);
  };
  // row shift of a tap inside the patch
  auto tap_shift = [&](int tap) -> int {
    if (MODE == L2S_MODE_CONV1D) return tap * p.dil + p.off - lo_shift;
    const int ky = tap / 3, kx = tap - ky * 3;
    return ky * PW + kx;                       // (ky-1)*PW + (kx-1) + (PW+1)
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  int pbuf_cur = 0;

  // weight fragment offsets inside a stage (row n = ni*16 + lm, chunk (ks*4+lg) ^ (lm & 7))
  const uint32_t wk0 = (uint32_t)(lm * 8 + ((0 + lg) ^ (lm & 7))) * 16;
  const uint32_t wk1 = (uint32_t)(lm * 8 + ((4 + lg) ^ (lm & 7))) * 16;

  // ---- pipeline: stages g, g+1 in flight; the patch of tile i+1 is issued at step 0 of tile i ----
  int s_step = 0, s_ws = 0, issued = 0;   // weight-issue cursor (step cycles over every tile)
  auto issue_next_w = [&]() {
    issue_w(s_step, s_ws);
    ++issued;
    s_ws = s_ws == 2 ? 0 : s_ws + 1;
    s_step = s_step + 1 == nsteps ? 0 : s_step + 1;
  };
  issue_patch(0, 0);
  issue_next_w();
  if (total > 1) issue_next_w();

  auto tap_mfma = [&](int tap, uint32_t wb) {  // one tap: 2 k-steps x (2 x 4) MFMAs out of the patch and a weight tile
    const int pr = wave * 32 + lm + tap_shift(tap);  // patch row of this lane's first output row (i = 0)
    const int x = pr & 7;
    const uint32_t pa = lds_base + (uint32_t)pbuf_cur * PATCH_B + (uint32_t)pr * 128;
    const uint32_t a0 = pa + (uint32_t)(((0 + lg) ^ x) << 4), a1 = pa + (uint32_t)(((4 + lg) ^ x) << 4);
    frag16 fa0[MI], fa1[MI], fw0[NI], fw1[NI];
    lds_read_b128<0>(fa0[0], a0); lds_read_b128<2048>(fa0[1], a0);
    lds_read_b128<0>(fw0[0], wb + wk0); lds_read_b128<2048>(fw0[1], wb + wk0);
    lds_read_b128<4096>(fw0[2], wb + wk0); lds_read_b128<6144>(fw0[3], wb + wk0);
    lds_wait();
    lds_read_b128<0>(fa1[0], a1); lds_read_b128<2048>(fa1[1], a1);
    lds_read_b128<0>(fw1[0], wb + wk1); lds_read_b128<2048>(fw1[1], wb + wk1);
    lds_read_b128<4096>(fw1[2], wb + wk1); lds_read_b128<6144>(fw1[3], wb + wk1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw0[j], fa0[i], acc[i][j]);
    lds_wait();
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw1[j], fa1[i], acc[i][j]);
  };

  int c_i = 0, c_step = 0, ws = 0, pbuf = 0;
  bool patch_just_issued = false;
  for (int g = 0; g < total; ++g) {
    // in flight behind stage g: stage g+1 (2 DMAs) and, right after a patch issue, that patch's 5 DMAs + the stage issued with it
    if (issued - g - 1 > 0) {
      if (patch_just_issued) wait_vmcnt<P_PER_W + 2>(); else wait_vmcnt<2>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    patch_just_issued = false;
    if (c_step == 0 && c_i + 1 < my_n) {       // next tile's patch first, then the weight stage (order fixes the vmcnt)
      issue_patch(c_i + 1, pbuf ^ 1);
      patch_just_issued = true;
    }
    if (issued < total) issue_next_w();
    pbuf_cur = pbuf;
    const uint32_t wb = wring + (uint32_t)ws * WST_B;
    tap_mfma(2 * c_step, wb);
    if (2 * c_step + 1 < ntaps) tap_mfma(2 * c_step + 1, wb + 8192);
    ws = ws == 2 ? 0 : ws + 1;
    if (++c_step < nsteps) continue;
    c_step = 0;

    // ---- tile done: epilogue through the (now dead) patch buffer of this tile ----
    int unit, q0;
    tile_origin(c_i, unit, q0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const uint32_t scr = lds_base + (uint32_t)pbuf * PATCH_B + (uint32_t)wave * epilogue_scratch_bytes<MI, NI>();
    epilogue<ET, MI, NI, EPI>(p, acc, scr, lane, wave * 32, 0, 0, [&](int r) -> int64_t {
      if (MODE == L2S_MODE_CONV1D) {
        const int t = q0 + r;
        return t < p.T_out ? ((int64_t)unit * p.T_out + t) * p.out_row_mul + p.out_row_add : (int64_t)-1;
      } else {
        const int Q = q0 + r;
        const int img = Q / (PH * PW), rem = Q - img * (PH * PW);
        const int py = rem / PW, px = rem - py * PW;
        const bool in = (py >= 1) && (py <= p.Hi) && (px >= 1) && (px <= p.Wi);
        const int64_t m = ((int64_t)img * p.Hi + (py - 1)) * p.Wi + (px - 1);
        return (in && m < p.M) ? m * p.out_row_mul + p.out_row_add : (int64_t)-1;
      }
    });
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    ++c_i;
    pbuf ^= 1;
  }
}

template <typename ET, int MODE, int EPI>
int launch_patch(const l2s_gemm_desc& d, hipStream_t st) {
  constexpr int SMEM = 2 * PATCH_B + 3 * WST_B;
  auto kern = patchconv64_kernel<ET, MODE, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  int ntiles, tiles_per_clip = 1, lo = 0;
  if (MODE == L2S_MODE_CONV1D) {
    const int clips = d.M / d.T_out;
    tiles_per_clip = (d.T_out + PBM - 1) / PBM;
    ntiles = clips * tiles_per_clip;
    const int a = d.off, b = (d.ntaps - 1) * d.dil + d.off;
    lo = a < b ? a : b;
  } else {
    const int64_t imgs = (int64_t)d.M / ((int64_t)d.Hi * d.Wi);
    const int64_t npos = imgs * (d.Hi + 2) * (d.Wi + 2);
    ntiles = (int)((npos + PBM - 1) / PBM);
  }
  const int grid = ntiles < 256 ? ntiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, st, d, ntiles, tiles_per_clip, lo);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

}  // namespace

// Does this descriptor fit the patch kernel?  (called by l2s_tapgemm before the generic path)
bool l2s_patchconv_eligible(const l2s_gemm_desc& d) {
  if (d.Cin != 64 || d.N != 64 || (d.groups > 1)) return false;
  if (d.mode == L2S_MODE_CONV1D) {
    if (d.stride != 1 || d.T_out != d.T_in || d.ntaps < 2 || d.M % d.T_out) return false;
    const int a = d.off, b = (d.ntaps - 1) * d.dil + d.off;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return (hi - lo) <= PROWS - PBM && lo <= 0 && hi >= 0 && (int64_t)d.M >= 64 * 1024;
  }
  if (d.mode == L2S_MODE_CONV2D) {
    if (d.stride != 1 || d.KW != 3 || d.ntaps != 9 || d.pad != 1 || d.Ho != d.Hi || d.Wo != d.Wi) return false;
    if (d.M % (d.Hi * d.Wi)) return false;
    return 2 * (d.Wi + 3) <= PROWS - PBM && (int64_t)d.M >= 64 * 1024;
  }
  return false;
}

namespace {
template <typename ET, int MODE>
int launch_patch_epi(const l2s_gemm_desc& d, hipStream_t st) {
  switch (pick_epilogue(d.flags, d.act)) {   // one epilogue family per kernel (tapgemm_tiles.h); rare ones share a superset
    case L2S_EPI_F16 + 0: case L2S_EPI_F16 + 2: return launch_patch<ET, MODE, L2S_EPI_F16 + 2>(d, st);
    case L2S_EPI_F16 + 1: case L2S_EPI_F16 + 3: return launch_patch<ET, MODE, L2S_EPI_F16 + 3>(d, st);
    case L2S_EPI_G16A: return launch_patch<ET, MODE, L2S_EPI_G16A>(d, st);
    case L2S_EPI_G16B: return launch_patch<ET, MODE, L2S_EPI_G16B>(d, st);
    default: return launch_patch<ET, MODE, L2S_EPI_ALL>(d, st);
  }
}
}  // namespace

int l2s_patchconv_launch(const l2s_gemm_desc& d, hipStream_t st) {
  if (d.dtype == L2S_F16)
    return d.mode == L2S_MODE_CONV1D ? launch_patch_epi<ElemF16, L2S_MODE_CONV1D>(d, st) : launch_patch_epi<ElemF16, L2S_MODE_CONV2D>(d, st);
  if (d.dtype == L2S_BF16)
    return d.mode == L2S_MODE_CONV1D ? launch_patch_epi<ElemBF16, L2S_MODE_CONV1D>(d, st) : launch_patch_epi<ElemBF16, L2S_MODE_CONV2D>(d, st);
  return L2S_EINVAL;
}
