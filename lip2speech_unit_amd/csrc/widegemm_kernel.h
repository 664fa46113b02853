// Wide-wave-tile GEMM for the Linear layers of the encoder and the conformer (fairseq q,k,v,out / fc1,fc2, espnet
// positionwise_feed_forward.py:28-30, attention.py:50-53, convolution.py:26-45): 256x256 output tile, FOUR waves (2 x 2) of
// 128x128 - one wave per SIMD, its 64 accumulator tiles in the AGPR half of the 512-register file.
//
// Why a second big-tile kernel next to phasegemm_kernel.h (8 waves of 128x64): under load these GEMMs are power-limited
// ("where a GEMM tile's time goes": the K loop runs at 1.45-1.8 GHz against 2.05 GHz MFMA-only), so
// what the K loop costs per MFMA besides the MFMA decides its speed.  A 128x128 wave tile reads 16 fragments per 64 MFMAs
// (0.25 per MFMA; 128 KB of LDS reads per 256x256x64 K-tile) where 128x64 reads 12 per 32 (0.375; 192 KB), and four waves
// issue half the LDS-DMA / barrier / address instructions of eight.
//
// K-tile = 64.  LDS: two K-tile stages of 64 KB (A rows 0-255 then W rows 0-255, 128-byte rows, 16-byte chunks XOR-swizzled
// through the DMA source address).  With ONE wave per SIMD nothing else hides a wave's own latencies, so the loop is a
// software pipeline inside the wave:
//   step 0 of K-tile kt (k 0-31):  64 MFMAs | the 16 fragment reads of step 1
//   wait lgkmcnt(0), vmcnt(0) - this wave's DMA of K-tile kt+1, issued a whole K-tile ago - and the block barrier:
//        every wave's part of stage kt+1 has landed, and every wave has finished READING stage kt (both steps' fragments)
//   step 1 (k 32-63):               64 MFMAs | the 16 fragment reads of step 0 of K-tile kt+1 | the 16 LDS-DMA instructions
//                                    of K-tile kt+2 into the stage just freed
// One barrier per 128 MFMAs, no fragment read ever waits at a K-tile boundary, every DMA has a K-tile (2 048 MFMA cycles) to
// land.  The stream of K-tiles runs across the block's output tiles (persistent blocks, XCD-aware banded order as
// tapgemm_kernel.h), so the next tile's first two K-tiles travel under the epilogue.
#pragma once
#include "tapgemm_common.h"
#include <cstdlib>

using namespace l2s;

namespace {

constexpr int WBM = 256, WBN = 256, WBK = 64;
constexpr int W_STAGE = (WBM + WBN) * WBK * 2;           // 64 KB
constexpr int w_smem() { return 2 * W_STAGE; }

template <typename ET, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void widegemm_kernel(const l2s_gemm_desc p, const int tilesM, const int tilesN, const int chunk, const int band) {
  constexpr int MI = 8, NI = 8;
  constexpr bool PAIRED = EPI <= L2S_EPI_G16A;           // 16-bit families store straight from the MFMA layout (epilogue_direct16)
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int K = p.Cin;
  const int nk = K / WBK;

  // ---- persistent tile schedule (as tapgemm_kernel.h) --------------------------------------------------------------
  const int ntiles = tilesM * tilesN;
  const int slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int lo = (blockIdx.x & 7) * chunk;
  const int hi = lo + chunk < ntiles ? lo + chunk : ntiles;
  const int my_n = (lo + slot < hi) ? (hi - lo - slot + slots - 1) / slots : 0;
  if (my_n == 0) return;
  const int total_kt = my_n * nk;                        // K-tiles this block consumes
  auto tile_coords = [&](int i, int& m0, int& n0) {
    const int l = lo + slot + i * slots;
    const int bsz = band * tilesN;
    const int bi = l / bsz, idx = l - bi * bsz;
    const int rows = tilesM - bi * band < band ? tilesM - bi * band : band;
    const int tn = idx / rows;
    m0 = (bi * band + idx - tn * rows) * WBM;
    n0 = tn * WBN;
  };

  // ---- staging: wave w brings A rows [64w, 64w+64) and W rows [64w, 64w+64) of the tile, 8 + 8 instructions of 8 rows ----
  // lane -> (row lane>>3, slot lane&7) fetches chunk slot ^ key(row): the swizzle lives in the source address, the LDS image
  // is lane-linear.  The row group of an instruction is uniform, so its address is an SGPR base (tile, row group, K-tile)
  // plus ONE per-lane 32-bit offset (A) or one of four (W, whose paired swizzle key depends on the row group): no per-row
  // pointer registers.  M % 8 == 0 and N % 8 == 0 (checked by the launcher): a row group is inside the matrix or clamped whole.
  const int srow = lane >> 3;
  const uint32_t a_lane = (uint32_t)(srow * p.lda * 2 + (((lane & 7) ^ srow) << 4));
  uint32_t w_lane[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int key = PAIRED ? paired_w_key(8 * jj + srow) : srow;
    w_lane[jj] = (uint32_t)(srow * K * 2 + (((lane & 7) ^ key) << 4));
  }
  int s_i = 0, s_kt = 0, staged = 0;                      // stream cursor: (tile, K-tile); K-tiles staged so far
  int s_m0 = 0, s_n0 = 0;
  tile_coords(0, s_m0, s_n0);
  // instruction g (0..15) of the K-tile at the cursor: g < 8 A rows, else W rows
  auto stage_instr = [&](int g) {
    uint16_t* dst = lds + (staged & 1) * (W_STAGE / 2) + (g >> 3) * (WBM * WBK) + (64 * wave + 8 * (g & 7)) * WBK;
    const char* src;
    if (g < 8) {
      int mg = s_m0 + 64 * wave + 8 * (g & 7);
      mg = mg < p.M ? mg : p.M - 8;                        // row groups past M / N are clamped (their outputs are never stored)
      src = (const char*)p.A + ((int64_t)mg * p.lda + s_kt * WBK) * 2 + a_lane;
    } else {
      int ng = s_n0 + 64 * wave + 8 * (g & 7);
      ng = ng < p.N ? ng : p.N - 8;
      src = (const char*)p.W + ((int64_t)ng * K + s_kt * WBK) * 2 + w_lane[g & 3];
    }
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  };
  // past the block's last K-tile the cursor stays where it is: the two trailing stagings re-fetch that K-tile into a free
  // stage (no dummy source, no select in front of every DMA instruction)
  auto stage_advance = [&]() {
    ++staged;
    if (staged < total_kt && ++s_kt == nk) {
      s_kt = 0;
      ++s_i;
      tile_coords(s_i, s_m0, s_n0);
    }
  };

  // ---- fragments ------------------------------------------------------------------------------------------------------
  const int lm = lane & 15, lg = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  uint32_t a_off[2], b_off[2][2];                          // [ks] ; [s][ks]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_off[ks] = (uint32_t)((wr * 128 + lm) * 128 + (((4 * ks + lg) ^ (lm & 7)) << 4));
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      b_off[s2][ks] = (uint32_t)(WBM * WBK * 2 + wc * 128 * 128) +
                      (PAIRED ? paired_w_off(lm, s2, 4 * ks + lg)
                              : (uint32_t)((16 * s2 + lm) * 128 + (((4 * ks + lg) ^ (lm & 7)) << 4)));
  }
  frag16 fa[2][MI], fb[2][NI];                             // [buffer][block]
  // read #g (0..15) of the fragments of k-step ks from the stage at byte address sb into buffer bf
  auto read_frag = [&](auto g_tag, const int bf, const uint32_t sb, const int ks) {
    constexpr int g = decltype(g_tag)::value;
    if constexpr (g < 8) {
      lds_read_b128<g * 2048>(fa[bf][g], sb + a_off[ks]);
    } else {
      constexpr int j = g - 8;                             // block j = 2b + s: plain rows 16 j + lm; paired b * 32 + ...
      lds_read_b128<(j >> 1) * 4096>(fb[bf][j], sb + b_off[j & 1][ks]);
    }
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // one k-step: 64 MFMAs out of buffer BF in 16 groups of 4, with fragment read #g of the next k-step (stage byte address
  // nsb, k-step nks, into buffer BF ^ 1) and, when DMA, LDS-DMA instruction #g in front of every group
  auto kstep = [&](auto bf_tag, auto dma_tag, const uint32_t nsb, const int nks) {
    constexpr int BF = decltype(bf_tag)::value;
    constexpr bool DMA = decltype(dma_tag)::value;
    // One wave per SIMD: whatever this wave issues between two MFMAs must fit under the 16 cycles of the first, or the pipe
    // idles - so the group's fragment read and its (up to two) DMA instructions each sit BETWEEN two MFMAs, pinned there.
    auto group = [&](auto g_tag) {
      constexpr int g = decltype(g_tag)::value;
      constexpr int i = g >> 1, j0 = (g & 1) * 4;
      acc[i][j0 + 0] = ET::mfma(fb[BF][j0 + 0], fa[BF][i], acc[i][j0 + 0]);
      __builtin_amdgcn_sched_barrier(0);
      read_frag(g_tag, BF ^ 1, nsb, nks);
      __builtin_amdgcn_sched_barrier(0);
      acc[i][j0 + 1] = ET::mfma(fb[BF][j0 + 1], fa[BF][i], acc[i][j0 + 1]);
      __builtin_amdgcn_sched_barrier(0);
      // all sixteen in the first eight groups: the last one then has 1.5 k-steps to land instead of one
      if constexpr (DMA && g < 8) { stage_instr(2 * g); __builtin_amdgcn_sched_barrier(0); }
      acc[i][j0 + 2] = ET::mfma(fb[BF][j0 + 2], fa[BF][i], acc[i][j0 + 2]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (DMA && g < 8) { stage_instr(2 * g + 1); __builtin_amdgcn_sched_barrier(0); }
      acc[i][j0 + 3] = ET::mfma(fb[BF][j0 + 3], fa[BF][i], acc[i][j0 + 3]);
      __builtin_amdgcn_sched_barrier(0);
    };
    __builtin_amdgcn_s_setprio(1);
    group(std::integral_constant<int, 0>{});  group(std::integral_constant<int, 1>{});
    group(std::integral_constant<int, 2>{});  group(std::integral_constant<int, 3>{});
    group(std::integral_constant<int, 4>{});  group(std::integral_constant<int, 5>{});
    group(std::integral_constant<int, 6>{});  group(std::integral_constant<int, 7>{});
    group(std::integral_constant<int, 8>{});  group(std::integral_constant<int, 9>{});
    group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
    group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{});
    group(std::integral_constant<int, 14>{}); group(std::integral_constant<int, 15>{});
    __builtin_amdgcn_s_setprio(0);
  };
  auto read_all = [&](const int bf, const uint32_t sb, const int ks) {
    read_frag(std::integral_constant<int, 0>{}, bf, sb, ks);  read_frag(std::integral_constant<int, 1>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 2>{}, bf, sb, ks);  read_frag(std::integral_constant<int, 3>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 4>{}, bf, sb, ks);  read_frag(std::integral_constant<int, 5>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 6>{}, bf, sb, ks);  read_frag(std::integral_constant<int, 7>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 8>{}, bf, sb, ks);  read_frag(std::integral_constant<int, 9>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 10>{}, bf, sb, ks); read_frag(std::integral_constant<int, 11>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 12>{}, bf, sb, ks); read_frag(std::integral_constant<int, 13>{}, bf, sb, ks);
    read_frag(std::integral_constant<int, 14>{}, bf, sb, ks); read_frag(std::integral_constant<int, 15>{}, bf, sb, ks);
  };

  // ---- prologue: K-tiles 0 and 1 of the stream staged, K-tile 0 landed, its step-0 fragments requested --------------------
#pragma unroll
  for (int g = 0; g < 16; ++g) stage_instr(g);
  stage_advance();
#pragma unroll
  for (int g = 0; g < 16; ++g) stage_instr(g);
  stage_advance();
  wait_vmcnt<16>();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_all(0, lds_base, 0);

  int q = 0;                                               // K-tile of the stream being computed
  for (int ti = 0; ti < my_n; ++ti) {
    for (int kt = 0; kt < nk; ++kt, ++q) {
      const uint32_t sb = lds_base + (uint32_t)(q & 1) * W_STAGE, sn = lds_base + (uint32_t)((q + 1) & 1) * W_STAGE;
      lds_wait();                                          // step-0 fragments (buffer 0)
      kstep(std::integral_constant<int, 0>{}, std::false_type{}, sb, 1);
      lds_wait();                                          // step-1 fragments (buffer 1): stage q is read out
      wait_vmcnt<0>();                                     // this wave's share of stage q + 1
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      kstep(std::integral_constant<int, 1>{}, std::true_type{}, sn, 0);   // + K-tile q + 2 into stage q's slot
      stage_advance();
    }
    int m0, n0;
    tile_coords(ti, m0, n0);
    auto rowmap = [&](int m) -> int64_t { return m < p.M ? (int64_t)m * p.out_row_mul + p.out_row_add : (int64_t)-1; };
    lds_wait();      // fragment reads issued under the last k-step still target registers: retire them before the epilogue's code
    if constexpr (EPI == L2S_EPI_S32) {
      epilogue_direct32<ET, MI, NI>(p, acc, lane, m0 + wr * 128, n0 + wc * 128, 0, rowmap);
    } else {
      epilogue_direct16<ET, MI, NI, EPI, decltype(rowmap), NoHook, true>(p, acc, lane, m0 + wr * 128, n0 + wc * 128, 0, rowmap);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // the step-0 fragments of the next tile's first K-tile are requested again here: the 64 registers of the copy requested
    // under the tile's last k-step would otherwise stay live through the whole epilogue
    read_all(0, lds_base + (uint32_t)(q & 1) * W_STAGE, 0);
  }
  lds_wait();        // the trailing fragment reads (of a dummy stage) ...
  wait_vmcnt<0>();   // ... and no LDS-DMA (the trailing dummies) may outlive the block's LDS allocation
}

template <typename ET, int EPI>
int launch_wide(const l2s_gemm_desc& d, hipStream_t st) {
  auto kern = widegemm_kernel<ET, EPI>;
  static L2sSmemOptIn opt_in;  // > 64 KB of dynamic LDS: opt-in per instantiation and device
  if (int e = l2s_smem_opt_in(kern, w_smem(), opt_in)) return e;
  const int tilesM = (d.M + WBM - 1) / WBM, tilesN = (d.N + WBN - 1) / WBN;
  const int ntiles = tilesM * tilesN;
  const int chunk = (ntiles + 7) / 8;
  const int slots = chunk < 32 ? chunk : 32;
  const double ap = (double)WBM * d.Cin * 2.0, wp = (double)WBN * d.Cin * 2.0;
  auto cdivi = [](int a, int b) { return (a + b - 1) / b; };
  int band = 1;
  double best = 1e300;
  for (int b = 1; b <= tilesM; ++b) {
    const int wn = cdivi(chunk, b) < tilesN ? cdivi(chunk, b) : tilesN;
    const int an = b * cdivi(chunk, b * tilesN);
    const double fp = ap * (an < tilesM ? an : tilesM) + wp * wn;
    if (fp < best) { best = fp; band = b; }
  }
  hipLaunchKernelGGL(kern, dim3(8 * slots), dim3(256), w_smem(), st, d, tilesM, tilesN, chunk, band);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET>
int launch_wide_epi(const l2s_gemm_desc& d, hipStream_t st) {
  switch (pick_epilogue(d.flags, d.act)) {
    case 0: return launch_wide<ET, 0>(d, st);
    case 1: return launch_wide<ET, 1>(d, st);
    case 2: return launch_wide<ET, 2>(d, st);
    case 3: return launch_wide<ET, 3>(d, st);
    case 4: return launch_wide<ET, 4>(d, st);
    case 5: return launch_wide<ET, 5>(d, st);
    case L2S_EPI_G16A: return launch_wide<ET, L2S_EPI_G16A>(d, st);
    case L2S_EPI_S32: return launch_wide<ET, L2S_EPI_S32>(d, st);
    default: return L2S_EUNSUPPORTED;
  }
}

}  // namespace
