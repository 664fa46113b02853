// Patch convolution: stride-1 Conv1d / 3x3 Conv2d with Cin = 64 and N = 64 (vocoder stage C=64: speech-resynthesis/
// models.py:19-31 at 16 000 samples per clip; ResNet layer1: avhubert/resnet.py:61-74 at 22x22) - the wide-M, narrow-K
// convolutions where the generic tap-GEMM re-fetches every input row once per tap through L2.
//
// One block keeps the input PATCH of its 256 output positions (+ halo) in LDS, loaded once by LDS-DMA, and forms every
// tap's MFMA operand by reading the patch at a shifted row: HBM/L2 traffic per tile drops from k x (A + W) tiles to
// one patch + k small weight tiles.  Conv2d works in a padded-flattened position space (image (H+2) x (W+2), all images
// back to back): a tap is then a constant row shift, the zero border comes for free from the DMA's per-lane source
// address (border positions read the zero page) and border outputs are simply not stored.
//
// Schedule: 4 waves of 64 positions x 64 channels (0.5 fragment reads per MFMA), ONE patch buffer and a ring of four
// single-tap 64 x 64 weight tiles = 72 KB, so TWO blocks share a CU: while one block runs its epilogue, waits at the
// tile barriers and loads its next patch, the other one has the MFMA pipes.  (The first version - one 8-wave block per
// CU, 32x64 wave tiles, double-buffered patch - had ~3.5 us of tile-end work per 4.8-8.3 us tile with nothing to
// overlap it: this one measured 6-15 % faster on every layer but the k = 3 residual conv, +6 %.)
// Inside a tile the taps are software-pipelined with COUNTED lgkmcnt waits: each group of 8 fragment reads is issued
// one k-step (16 MFMAs) before it is needed; the barrier that publishes tap t+1's weights sits under tap t's first 16
// MFMAs, and the weight DMA issued behind it refills the slot of tap t-1.  Persistent blocks; the weight stream runs
// across a block's tiles.  Epilogue: tapgemm_common.h, scratch in the (then dead) patch buffer.
#include "tapgemm_common.h"
#include <cstdlib>

using namespace l2s;

namespace {

constexpr int PBM = 256;          // output positions per tile
constexpr int PROWS = 320;        // patch rows: 256 + halo (<= 64), multiple of 64
constexpr int PATCH_B = PROWS * 128;
constexpr int QRING = 4;
// CH = Cin = N is 64 or 128.  The 128-channel kernel is the same schedule over (tap, 64-channel K half) stream elements:
// two half-patches of 64 channels, weight elements of 128 x 64, 8 waves = 4 position blocks x 2 channel blocks of
// 64 x 64, 144 KB of LDS and one block per CU (its two waves per SIMD overlap each other instead of a second block).
constexpr int q_el_b(int ch) { return ch * 128; }                                  // one stream element: CH x 64 weights
constexpr int q_smem(int ch) { return (ch / 64) * PATCH_B + QRING * q_el_b(ch); }  // 72 KB / 144 KB

#ifdef L2S_PATCH_STAMPS
__device__ unsigned long long* g_patch_stamps = nullptr;
#define PSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ps_acc[i] += now_ - ps_last; ps_last = now_; }
#else
#define PSTAMP(i)
#endif
template <typename ET, int MODE, int EPI, int CH>
__global__ __launch_bounds__(CH * 4, (CH == 64 ? 2 : 1)) void patchconv64_kernel(const l2s_gemm_desc p, const int ntiles,
                                                                                const int tiles_per_clip, const int lo_shift) {
  constexpr int MI = 4, NI = 4;
  // 16-bit families store straight from the MFMA layout (epilogue_direct16): their weight fragments are read in the paired row
  // order and the weight tiles use the paired swizzle key; the catch-all family keeps the transposing epilogue
#ifdef L2S_NO_PAIRED     // (A/B switch of the diagnostic builds)
  constexpr bool PAIRED = false;
#else
  constexpr bool PAIRED = EPI <= L2S_EPI_G16B;
#endif
  constexpr int HALVES = CH / 64;              // 64-channel K halves = half-patches
  constexpr int NWC = CH / 64, NW = 4 * NWC;   // wave grid: 4 position blocks x NWC channel blocks
  constexpr int P_PER_W = HALVES * (PROWS / 8) / NW;   // patch DMA instructions per wave (10)
  constexpr int PATCH_TOT = HALVES * PATCH_B;
  constexpr int QEL_B = q_el_b(CH);
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];  // [patch][QRING weight tiles]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, lg = lane >> 4;
  const int srow = lane >> 3, schunk = (lane & 7) ^ (srow & 7);
  const int ntaps = p.ntaps;
  const int Ktot = ntaps * CH;
  const int wm = wave / NWC, wc = wave - wm * NWC;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);
  const uint16_t* A = (const uint16_t*)p.A;
  const uint16_t* W = (const uint16_t*)p.W;
  const int PW = p.Wi + 2, PH = p.Hi + 2;       // CONV2D padded image
  const int my_n = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_n <= 0) return;
  const int nel = ntaps * HALVES;              // stream elements per tile: (tap, K half)
  const int total = my_n * nel;                // weight stream across the block's tiles

  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t wring = lds_base + PATCH_TOT;

  auto tile_origin = [&](int i, int& unit, int& q0) {
    const int L = blockIdx.x + i * gridDim.x;
    if (MODE == L2S_MODE_CONV1D) { unit = L / tiles_per_clip; q0 = (L - unit * tiles_per_clip) * PBM; }
    else { unit = 0; q0 = L * PBM; }
  };
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
  auto issue_patch = [&](int i) {
    int unit, q0;
    tile_origin(i, unit, q0);
    if (MODE == L2S_MODE_CONV1D) {
#pragma unroll
      for (int j = 0; j < P_PER_W; ++j) {
        const int instr = wave * P_PER_W + j;      // [0, 40 * HALVES): half-patch instr / 40, row block instr % 40
        const int hp = instr / (PROWS / 8), blk = instr - hp * (PROWS / 8);
        const uint16_t* g = patch_src(unit, q0, blk * 8 + srow);
        g = (g == zero) ? zero : g + hp * 64;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + hp * (PATCH_B / 2) + blk * 512), 16, 0, 0);
      }
    } else {
      // padded-flattened position -> (image, row, column) by ONE division pair for the wave's first row; the other
      // nine instructions advance it by 8 positions each (two 32-bit divisions per row cost ~70 VALU ops, and ten of
      // them per wave were a fifth of the tile time)
      const int hp = (wave * P_PER_W) / (PROWS / 8), blk0 = wave * P_PER_W - hp * (PROWS / 8);   // a wave stays in one half
      int Q = q0 - (PW + 1) + blk0 * 8 + srow;
      int img = 0, py = 0, px = 0;
      if (Q >= 0) {
        img = Q / (PH * PW);
        const int rem = Q - img * (PH * PW);
        py = rem / PW;
        px = rem - py * PW;
      }
#pragma unroll
      for (int j = 0; j < P_PER_W; ++j) {
        const bool in = (Q >= 0) && (py >= 1) && (py <= p.Hi) && (px >= 1) && (px <= p.Wi) && ((int64_t)img * p.Hi * p.Wi < (int64_t)p.M);
        const uint16_t* g = in ? A + (((int64_t)img * p.Hi + (py - 1)) * p.Wi + (px - 1)) * p.lda + hp * 64 + schunk * 8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + hp * (PATCH_B / 2) + (blk0 + j) * 512), 16, 0, 0);
        Q += 8;
        if (Q >= 0) {
          if (Q >= 8) px += 8;                   // the coordinates were valid: advance them by 8 positions
          else { img = 0; py = 0; px = Q; }      // the position just became non-negative
          while (px >= PW) { px -= PW; ++py; }
          while (py >= PH) { py -= PH; ++img; }
        }
      }
    }
  };
  // weight element (tap, K half): CH rows x 128 B = CH / 8 DMA instructions, two per wave
  // (tile rows 16 wave + srow and + 8; 16-byte chunk c of tile row r is stored at chunk c ^ key(r): key = r & 7, or the paired key)
  const int wrow0 = wave * 16 + srow;
  const int wch0 = (lane & 7) ^ (PAIRED ? paired_w_key(wrow0) : (srow & 7));
  const int wch1 = (lane & 7) ^ (PAIRED ? paired_w_key(wrow0 + 8) : (srow & 7));
  const uint16_t* w_ptr0 = W + (int64_t)wrow0 * Ktot + wch0 * 8;
  const uint16_t* w_ptr1 = W + (int64_t)(wrow0 + 8) * Ktot + wch1 * 8;
  int s_el = 0, s_slot = 0, issued = 0;        // weight-stream cursor: element inside the tile = tap * HALVES + half
  auto issue_next_w = [&]() {
    uint16_t* dst = lds + PATCH_TOT / 2 + s_slot * (QEL_B / 2) + wave * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)(w_ptr0 + s_el * 64), (lptr_t)dst, 16, 0, 0);   // K offset tap*CH + half*64 = el*64
    __builtin_amdgcn_global_load_lds((gptr_t)(w_ptr1 + s_el * 64), (lptr_t)(dst + 512), 16, 0, 0);
    ++issued;
    s_slot = s_slot == QRING - 1 ? 0 : s_slot + 1;
    s_el = s_el + 1 == nel ? 0 : s_el + 1;
  };
  auto tap_shift = [&](int tap) -> int {
    if (MODE == L2S_MODE_CONV1D) return tap * p.dil + p.off - lo_shift;
    const int ky = tap / 3, kx = tap - ky * 3;
    return ky * PW + kx;
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  // weight fragment of block j at k-step ks: plain order = tile row 16 j + lm; paired order = row 32 (j >> 1) + 8 (lm >> 2) + 4 (j & 1)
  // + (lm & 3) (tapgemm_common.h: paired_w_off).  wk[ks][s]: byte offset for blocks with j & 1 == s, block pair 0
  uint32_t wk[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      wk[ks][s2] = PAIRED ? paired_w_off(lm, s2, 4 * ks + lg)
                          : (uint32_t)((lm + 16 * s2) * 8 + ((4 * ks + lg) ^ (lm & 7))) * 16;
  constexpr int WJ2 = PAIRED ? 4096 : 4096;   // blocks 2, 3 = blocks 0, 1 + 32 tile rows in both orders

  frag16 fa0[MI], fw0[NI], fa1[MI], fw1[NI];
  uint32_t a1_next = 0;
  auto read_k0 = [&](int el, int slot) {
    const int tap = el / HALVES, hp = el - tap * HALVES;
    const int pr = wm * 64 + lm + tap_shift(tap);    // patch row of this lane's first output row (i = 0)
    const int x = pr & 7;
    const uint32_t pa = lds_base + (uint32_t)hp * PATCH_B + (uint32_t)pr * 128;
    const uint32_t a0 = pa + (uint32_t)(((0 + lg) ^ x) << 4);
    a1_next = pa + (uint32_t)(((4 + lg) ^ x) << 4);
    const uint32_t wb = wring + (uint32_t)slot * QEL_B + (uint32_t)wc * 8192;
    lds_read_b128<0>(fa0[0], a0); lds_read_b128<2048>(fa0[1], a0); lds_read_b128<4096>(fa0[2], a0); lds_read_b128<6144>(fa0[3], a0);
    lds_read_b128<0>(fw0[0], wb + wk[0][0]); lds_read_b128<0>(fw0[1], wb + wk[0][1]);
    lds_read_b128<WJ2>(fw0[2], wb + wk[0][0]); lds_read_b128<WJ2>(fw0[3], wb + wk[0][1]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_k1 = [&](int slot) {               // k1 of the tap whose k0 was read last
    const uint32_t wb = wring + (uint32_t)slot * QEL_B + (uint32_t)wc * 8192;
    lds_read_b128<0>(fa1[0], a1_next); lds_read_b128<2048>(fa1[1], a1_next); lds_read_b128<4096>(fa1[2], a1_next); lds_read_b128<6144>(fa1[3], a1_next);
    lds_read_b128<0>(fw1[0], wb + wk[1][0]); lds_read_b128<0>(fw1[1], wb + wk[1][1]);
    lds_read_b128<WJ2>(fw1[2], wb + wk[1][0]); lds_read_b128<WJ2>(fw1[3], wb + wk[1][1]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_k = [&](frag16(&fw)[NI], frag16(&fa)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw[j], fa[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  };

  // kernel start: first patch and the first two weight tiles
  issue_patch(0);
  issue_next_w();
  if (total > 1) issue_next_w();
  int g = 0;                                   // weight-stream element being consumed
  int slot = 0;
#ifdef L2S_PATCH_STAMPS
  unsigned long long ps_acc[5] = {0, 0, 0, 0, 0};
  unsigned long long ps_last = __builtin_amdgcn_s_memtime();
#endif
  for (int c_i = 0; c_i < my_n; ++c_i) {
    // ---- tile start: the patch and the first tap's weights are visible to every wave ----
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PSTAMP(0)
    if (issued < total) issue_next_w();        // element g+2
    read_k0(0, slot);
    read_k1(slot);
    for (int t = 0; t < nel; ++t) {
      const bool more = t + 1 < nel;
      lds_wait_n<8>();                         // k0(t) landed, k1(t) may still be in flight
      mfma_k(fw0, fa0);
      const int nslot = slot == QRING - 1 ? 0 : slot + 1;
      if (more) {
        // publish tap t+1's weights: in flight behind them is only element g+2 (two DMAs)
        if (issued - g - 2 > 0) wait_vmcnt<2>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < total) issue_next_w();    // element g+3 -> the slot of element g-1, read by nobody any more
        read_k0(t + 1, nslot);
        lds_wait_n<8>();                       // k1(t)
      } else {
        lds_wait_n<0>();
      }
      mfma_k(fw1, fa1);
      if (more) read_k1(nslot);
      slot = nslot;
      ++g;
    }
    PSTAMP(1)
    // ---- tile done: every wave has finished reading the patch; epilogue through the patch buffer ----
    int unit, q0;
    tile_origin(c_i, unit, q0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PSTAMP(2)
    const uint32_t scr = lds_base + (uint32_t)wave * epilogue_scratch_bytes<MI, NI>();
    int rm_r = 0x7fffffff, rm_img = 0, rm_py = 0, rm_px = 0;   // CONV2D row -> pixel cache of the lambda below
    auto rowmap = [&](int r) -> int64_t {
      if (MODE == L2S_MODE_CONV1D) {
        const int t = q0 + r;
        return t < p.T_out ? ((int64_t)unit * p.T_out + t) * p.out_row_mul + p.out_row_add : (int64_t)-1;
      } else {
        // the epilogue asks for ascending rows: divide once per lane and tile, then step the coordinates
        if (r < rm_r) {
          const int Q = q0 + r;
          rm_img = Q / (PH * PW);
          const int rem = Q - rm_img * (PH * PW);
          rm_py = rem / PW;
          rm_px = rem - rm_py * PW;
        } else {
          rm_px += r - rm_r;
          while (rm_px >= PW) { rm_px -= PW; ++rm_py; }
          while (rm_py >= PH) { rm_py -= PH; ++rm_img; }
        }
        rm_r = r;
        const bool in = (rm_py >= 1) && (rm_py <= p.Hi) && (rm_px >= 1) && (rm_px <= p.Wi);
        const int64_t m = ((int64_t)rm_img * p.Hi + (rm_py - 1)) * p.Wi + (rm_px - 1);
        return (in && m < p.M) ? m * p.out_row_mul + p.out_row_add : (int64_t)-1;
      }
    };
    // PAIRED: no scratch, so the patch buffer is free from the barrier above: the next tile's patch is requested from inside
    // the epilogue (behind its own loads) and travels under the rest of it
    if constexpr (PAIRED)
      epilogue_direct16<ET, MI, NI, EPI>(p, acc, lane, wm * 64, wc * 64, 0, rowmap, -1, 0, [&] { if (c_i + 1 < my_n) issue_patch(c_i + 1); });
    else epilogue<ET, MI, NI, EPI>(p, acc, scr, lane, wm * 64, wc * 64, 0, rowmap);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    PSTAMP(3)
    if (!PAIRED && c_i + 1 < my_n) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();            // every wave is done with its epilogue scratch: the patch buffer is free
      asm volatile("" ::: "memory");
      issue_patch(c_i + 1);
    }
    PSTAMP(4)
  }
  wait_vmcnt<0>();                             // no LDS-DMA may outlive the block
#ifdef L2S_PATCH_STAMPS
  if (lane == 0 && wave == 0 && g_patch_stamps) {
    unsigned long long* o = g_patch_stamps + (int64_t)blockIdx.x * 8;
    for (int i = 0; i < 5; ++i) o[i] = ps_acc[i];
    o[5] = (unsigned long long)my_n;
  }
#endif
}

template <typename ET, int MODE, int EPI, int CH>
int launch_patch(const l2s_gemm_desc& d, hipStream_t st) {
  constexpr int QSMEM = q_smem(CH);
  auto kern = patchconv64_kernel<ET, MODE, EPI, CH>;
  static L2sSmemOptIn opt_in;  // > 64 KB of dynamic LDS: opt-in per instantiation and device
  if (int e = l2s_smem_opt_in(kern, QSMEM, opt_in)) return e;
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
  constexpr int slots = CH == 64 ? 512 : 256;   // resident blocks: two per CU at 64 channels, one at 128
  const int grid = ntiles < slots ? ntiles : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(CH * 4), QSMEM, st, d, ntiles, tiles_per_clip, lo);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

}  // namespace

#ifdef L2S_PATCH_STAMPS
extern "C" int l2s_debug_patch_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_patch_stamps), &buf, sizeof(buf)); }
#endif
// Does this descriptor fit the patch kernel?  (called by l2s_tapgemm before the generic path)
bool l2s_patchconv_eligible(const l2s_gemm_desc& d) {
  static const int ch128 = [] { const char* e = getenv("L2S_PATCH128"); return e ? atoi(e) : 1; }();   // A/B switch
  if (!((d.Cin == 64 && d.N == 64) || (ch128 && d.Cin == 128 && d.N == 128)) || (d.groups > 1)) return false;
  if (d.mode == L2S_MODE_CONV1D) {
    if (d.stride != 1 || d.T_out != d.T_in || d.ntaps < 2 || d.M % d.T_out) return false;
    const int a = d.off, b = (d.ntaps - 1) * d.dil + d.off;
    const int lo = a < b ? a : b, hi = a < b ? b : a;
    return (hi - lo) <= PROWS - PBM && lo <= 0 && hi >= 0 && (int64_t)d.M >= 64 * 1024;
  }
  if (d.mode == L2S_MODE_CONV2D) {
    if (d.stride != 1 || d.KW != 3 || d.ntaps != 9 || d.pad != 1 || d.Ho != d.Hi || d.Wo != d.Wi) return false;
    if (d.M % (d.Hi * d.Wi)) return false;
    // the padded-flattened position space computes (H+2)(W+2) positions per image: 1.19x at 22x22, 1.40x at 11x11,
    // where the generic kernel measured 17 % faster
    if (4 * (d.Hi + 2) * (d.Wi + 2) > 5 * d.Hi * d.Wi) return false;
    return 2 * (d.Wi + 3) <= PROWS - PBM && (int64_t)d.M >= 64 * 1024;
  }
  return false;
}

namespace {
template <typename ET, int MODE, int CH>
int launch_patch_epi(const l2s_gemm_desc& d, hipStream_t st) {
  switch (pick_epilogue(d.flags, d.act)) {   // one epilogue family per kernel (tapgemm_tiles.h); rare ones share a superset
    case L2S_EPI_F16 + 0: case L2S_EPI_F16 + 2: return launch_patch<ET, MODE, L2S_EPI_F16 + 2, CH>(d, st);
    case L2S_EPI_F16 + 1: case L2S_EPI_F16 + 3: return launch_patch<ET, MODE, L2S_EPI_F16 + 3, CH>(d, st);
    case L2S_EPI_G16A: return launch_patch<ET, MODE, L2S_EPI_G16A, CH>(d, st);
    case L2S_EPI_G16B: return launch_patch<ET, MODE, L2S_EPI_G16B, CH>(d, st);
    default: return launch_patch<ET, MODE, L2S_EPI_ALL, CH>(d, st);
  }
}
template <typename ET, int MODE>
int launch_patch_ch(const l2s_gemm_desc& d, hipStream_t st) {
  return d.N == 128 ? launch_patch_epi<ET, MODE, 128>(d, st) : launch_patch_epi<ET, MODE, 64>(d, st);
}
}  // namespace

int l2s_patchconv_launch(const l2s_gemm_desc& d, hipStream_t st) {
  if (d.dtype == L2S_F16)
    return d.mode == L2S_MODE_CONV1D ? launch_patch_ch<ElemF16, L2S_MODE_CONV1D>(d, st) : launch_patch_ch<ElemF16, L2S_MODE_CONV2D>(d, st);
  if (d.dtype == L2S_BF16)
    return d.mode == L2S_MODE_CONV1D ? launch_patch_ch<ElemBF16, L2S_MODE_CONV1D>(d, st) : launch_patch_ch<ElemBF16, L2S_MODE_CONV2D>(d, st);
  return L2S_EINVAL;
}
