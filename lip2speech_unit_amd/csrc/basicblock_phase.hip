// Fused ResNet BasicBlocks of the lip frontend on the phase-staggered schedule (avhubert/resnet.py:43-74, stride 1, no downsample,
// eval BatchNorm folded into weight + bias):
//     out = prelu2(conv2(prelu1(conv1(x) + b1)) + b2 + x)
// built for layer2's second block (11 x 11 x 128) and - CH = 64, see the end of this comment - for layer1 (22 x 22 x 64, its blocks
// back to back).  The text describes the 128-channel geometry.
// basicblock.hip (C = 64) keeps one 22 x 22 image and whole-tap weight tiles in LDS; at C = 128 a tap of weights is 32 KB and an
// 11 x 11 image fills less than half a 256-row tile, so this kernel is the LDS-resident-image data flow on the phase-staggered
// schedule of respair_phase.hip (8 waves = 4 wave rows x 2 wave columns of 64 x 64, weights streamed as 8 KB quarters through a
// 4-slot ring by LDS-DMA, one quarter = one phase of 16 MFMAs per wave between two raw barriers, the upper wave rows one barrier
// behind the lower ones):
//   * a tile = IPT = 2 consecutive images.  Their PADDED maps ((H+2) x (W+2) positions each, zero border from the DMA source
//     address, image pitch 176 rows) live in LDS as two 64-channel blocks of 352 rows x 128 B; a 3x3 tap is a constant row
//     shift (ky-1)(W+2) + (kx-1) in that space;
//   * the MFMA rows are the 242 INTERIOR positions only (row m of the tile = position m % 121 of image m / 121): every lane
//     keeps the padded row of its four row groups and forms a tap's fragment addresses from them (4 VALU per group and tap),
//     so no border position is computed (the padded-flattened tile of basicblock.hip would compute 169 per 121) and no halo:
//     242 of the tile's 256 rows are real;
//   * conv1's bias + PReLU happen in the MFMA register layout, t1 goes back to the SAME region rows (the residual rows x were
//     taken to registers first; the border rows keep the zeros of the patch = conv2's padding), conv2 runs out of t1,
//     bias + residual + PReLU in registers, 16 bytes per lane straight from the paired MFMA layout to HBM;
//   * the weight stream runs across conv1 -> conv2 -> the next tile without draining; bias and slope vectors wait in LDS behind
//     the ring; the next tile's patch DMA is issued when conv2 has finished reading t1 and travels under the epilogue.
// HBM sees x once and out once; per image 2 x 9 x 128 x 128 x 121 MACs.
//
// TAIL variant (l2s_basicstage128_tail_fused): the rest of the strided stage behind its first convolution in ONE launch
// (avhubert/resnet.py:61-74 with downsample, :101-118): out0 = prelu(conv3x3(t0) + b + Wd x[::2, ::2] + bd) - the second conv of
// the stage's first block with the 1x1 stride-2 downsample of the stage input as its residual - followed by the BasicBlock above on
// out0.  The downsample is not a launch and not a residual pass: the 64-channel stage input at the even positions is one more
// 64-wide K-tile of the SAME accumulation (its rows sit in a third LDS block, Wd rides behind the nine taps in the packed weight
// matrix [128][9*128 + 64 + 64 zeros] so that conv A is a whole number of ring turns), bd is folded into the bias on the host.
// out0 stays in registers as the BasicBlock's residual and goes into the region as its input; HBM sees t0 and x once, out once.
//
// CH = 64 (layer1, l2s_basicblock_fused / l2s_basiclayer_fused at 22 x 22): 8 wave rows x 1 wave column, a tile = ONE image = 484
// interior rows of 512 (basicblock.hip computes all 576 padded positions), region 576 rows x 128 B, 4 KB quarters staged by the even
// waves, a tap = two phases = half a ring turn (nine taps: a convolution starts on slot 0 or 2 alternately, respair_phase.hip's
// CH = 64 scheme).  Up to four blocks run back to back: a block's output stays in registers as the next block's residual and goes
// back into the region as its input - the 16-bit values a launch of its own would read back, so the result is bit-identical to
// one launch per block.
#include "tapgemm_common.h"
#include <cstdlib>

using namespace l2s;

namespace {

constexpr int BP_MAXCV = 8;                          // convolutions per launch: 2 per block, CH = 64: up to 4 blocks

template <int CH_, int H_, int W_, bool TAIL>
struct PGeo {
  static constexpr int CH = CH_, NWC = CH_ / 64, NWR = 8 / NWC, NBLK = CH_ / 64;
  static constexpr int H = H_, W = W_, PW = W_ + 2, PH = H_ + 2, HW = H_ * W_, PP = PH * PW;
  static constexpr int RM = NWR * 64;                 // MFMA rows per tile
  static constexpr int IPT = RM / HW;                 // images per tile
  static constexpr int IMGP = (PP + 7) / 8 * 8;       // LDS rows per image (padded map, rounded to the DMA's 8-row granule)
  static constexpr int RPR = IPT * IMGP;              // rows of a region block
  static constexpr int BLK_B = RPR * 128;
  static constexpr int REGION = NBLK * BLK_B;
  static constexpr int Q_B = NWC * 32 * 128;          // one weight quarter: 32 rows per wave column x 64 K values (8 / 4 KB)
  static constexpr int XRQ = 4;
  static constexpr int XS_B = TAIL ? RM * 128 : 0;    // TAIL: the stage input at the even positions, RM rows x 64 channels
  static constexpr int RING_OFF = REGION + XS_B;
  static constexpr int TAB_OFF = RING_OFF + XRQ * Q_B;   // per convolution bias | slope as fp32 (TAIL: conv A's pair last)
  static constexpr int MAXNB = CH_ == 64 ? BP_MAXCV / 2 : 1;
  static constexpr int NTAB = 4 * MAXNB + (TAIL ? 2 : 0);
  static constexpr int SMEM = TAB_OFF + NTAB * CH * 4;
  static constexpr int PI = NBLK * (RPR / 8) + (TAIL ? RM / 8 : 0);   // patch DMA instructions per tile
  static constexpr int PPW = (PI + 7) / 8;
  static constexpr int KA = 9 * CH + 128;             // TAIL: K of conv A's packed weights (9 taps | Wd | 64 zero columns)
  static_assert(IPT >= 1 && SMEM <= 160 * 1024 && (NBLK == 1 || BLK_B + 2048 < 65536) && (!TAIL || CH_ == 128), "geometry");
};

struct BpArgs {
  const uint16_t* X; uint16_t* Y;
  // per convolution (conv1, conv2 of block 0, conv1, conv2 of block 1, ...): weights [CH][9*CH], bias, PReLU slope
  const uint16_t* Wt[BP_MAXCV]; const float* bias[BP_MAXCV]; const float* slope[BP_MAXCV];
  int nimg, ntiles, nb;
  // TAIL: the stage input [nimg][(2H)(2W)][64] and conv A (packed weights [128][PGeo::KA], bias incl. the downsample's, slope)
  const uint16_t* X0; const uint16_t* WA; const float* bA; const float* sA;
};
// arr[i] for a wave-uniform i without indexing the kernel-argument array dynamically (that would copy it to scratch)
// (N: the entries this instantiation can use - the others are never read, so their SGPRs are free)
template <int N, typename P>
__device__ __forceinline__ P bp_pick(const P (&arr)[BP_MAXCV], int i) {
  P r = arr[0];
#pragma unroll
  for (int j = 1; j < N; ++j) r = i == j ? arr[j] : r;
  return r;
}

__device__ __forceinline__ void bp_write_u4(uint32_t addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// (reads that stay live across a convolution carry their own wait: see respair_phase.hip)
__device__ __forceinline__ void bp_read8_u4_sync(u32x4_t (&v)[4][2], const uint32_t (&ad)[4][2]) {
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
               "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[3][0]), "=&v"(v[3][1])
               : "v"(ad[0][0]), "v"(ad[0][1]), "v"(ad[1][0]), "v"(ad[1][1]), "v"(ad[2][0]), "v"(ad[2][1]), "v"(ad[3][0]), "v"(ad[3][1]));
}
__device__ __forceinline__ void bp_read4_f4_sync(f32x4_t (&v)[4], const uint32_t (&ad)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]));
}

template <typename ET, int CH_, int H_, int W_, bool TAIL>
__global__ __launch_bounds__(512) void basicblock_phase_kernel(const BpArgs a) {
  using G = PGeo<CH_, H_, W_, TAIL>;
  constexpr int MI = 4, NI = 4, CH = G::CH;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / G::NWC, wc = wave % G::NWC;
  const bool upper = wave >= 4;                // the second wave of every SIMD: runs one barrier behind
  const int lm = lane & 15, lg = lane >> 4;
  const int srow = lane >> 3;
  constexpr int Ktot = 9 * CH;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);
  const int my_n = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_n <= 0) return;
  constexpr int nkt = 9 * G::NBLK;             // K-tiles per convolution: (tap, 64-channel block)
  const int ncv = 2 * a.nb;                    // convolutions of the blocks (TAIL: conv A comes before them)

  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t wring = lds_base + G::RING_OFF;

  // ---- patch: PI instructions of 8 region rows x 128 B; region row R of a block = padded position R % IMGP of image R / IMGP ----
  auto issue_patch = [&](int i) {
    asm volatile("" : "+s"(i));               // (opaque: the next tile's per-lane source addresses are formed HERE, not at the tile start and spilled)
    const int img0 = ((int)blockIdx.x + i * (int)gridDim.x) * G::IPT;
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
      const int instr = wave * G::PPW + j;     // wave-uniform
      int ln = lane;
      asm volatile("" : "+v"(ln));            // (opaque input: keeps the PPW row decompositions inside the tile loop, see basicblock.hip;
      const int sr = ln >> 3;                  //  the chunk too: base + chunk is otherwise hoisted as per-lane 64-bit pointers, lane & 7 alone
      const int sc8 = ((ln & 7) ^ (sr & 7)) * 8;   //  as a register of its own - and spilled)
      if (instr < G::NBLK * (G::RPR / 8)) {
        const int cq = instr / (G::RPR / 8), blk = instr - cq * (G::RPR / 8);
        const int R = blk * 8 + sr;
        const int im = R / G::IMGP, p = R - im * G::IMGP;
        const int py = p / G::PW, px = p - py * G::PW;
        const bool in = py >= 1 && py <= G::H && px >= 1 && px <= G::W && img0 + im < a.nimg;
        const uint16_t* g = in ? a.X + ((int64_t)(img0 + im) * G::HW + (py - 1) * G::W + (px - 1)) * CH + cq * 64 + sc8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + cq * (G::BLK_B / 2) + blk * 512), 16, 0, 0);
      } else if (TAIL && instr < G::PI) {
        // row m of the third block = the stage input (64 channels, (2H) x (2W) map) at position (2y, 2x) of MFMA row m's pixel
        const int blk = instr - G::NBLK * (G::RPR / 8);
        const int m = blk * 8 + sr;
        const int im = m / G::HW, rem = m - im * G::HW;
        const int y = rem / G::W, x = rem - y * G::W;
        const bool in = im < G::IPT && img0 + im < a.nimg;
        const uint16_t* g = in ? a.X0 + ((int64_t)(img0 + im) * (4 * G::HW) + (2 * y) * (2 * G::W) + 2 * x) * 64 + sc8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + G::REGION / 2 + blk * 512), 16, 0, 0);
      }
    }
  };

  // ---- weight stream (respair_phase.hip): quarter (conv, K-tile, half h) = rows {wc*64 + 32 h + 0..31} of every wave column; a
  // staging wave moves 8 quarter rows with ONE instruction (CH = 128: every wave; CH = 64: the even waves); paired row order
  // (a lane ends with 8 consecutive channels) ----
  const bool stager = CH >= 128 || !(wave & 1);
  const int sw = CH >= 128 ? wave : wave >> 1;
  uint32_t w_lane, w_laneA = 0;
  {
    const int qrow = sw * 8 + srow;            // row of the quarter: wave column qrow / 32, row qrow % 32 of its share
    const int within = qrow & 31;
    const int n = (qrow >> 5) * 64 + within;
    w_lane = (uint32_t)(n * Ktot + (((lane & 7) ^ paired_w_key(within)) * 8)) * 2u;
    if (TAIL) w_laneA = (uint32_t)(n * G::KA + (((lane & 7) ^ paired_w_key(within)) * 8)) * 2u;
  }
  constexpr uint32_t h_bytes = (uint32_t)(32 * Ktot) * 2u, h_bytesA = (uint32_t)(32 * G::KA) * 2u;
  // cursor: cycles (WA ->) conv 0 -> conv 1 -> ...; the stagings past the block's end re-fetch valid memory.  s_cv: the convolution
  // being staged, -1 = conv A (TAIL: the tile starts with it)
  const char* kt_ptr = (const char*)(TAIL ? a.WA : a.Wt[0]);
  int s_kt = 0, s_cv = TAIL ? -1 : 0;
  auto stage_one = [&](auto dslot_tag, auto sh_tag) {
    constexpr int DSLOT = decltype(dslot_tag)::value, SH = decltype(sh_tag)::value;
    const bool in_a = TAIL && s_cv < 0;
    const char* wb = kt_ptr + (size_t)(SH ? (in_a ? h_bytesA : h_bytes) : 0u);
    uint16_t* dst = lds + G::RING_OFF / 2 + DSLOT * (G::Q_B / 2) + sw * 512;
    if (stager) __builtin_amdgcn_global_load_lds((gptr_t)(wb + (size_t)(in_a ? w_laneA : w_lane)), (lptr_t)dst, 16, 0, 0);
    if constexpr (SH == 1) {
      kt_ptr += 128;
      if (++s_kt == (in_a ? G::KA / 64 : nkt)) {
        s_kt = 0;
        s_cv = s_cv + 1 == ncv ? (TAIL ? -1 : 0) : s_cv + 1;
        kt_ptr = (const char*)(s_cv < 0 ? a.WA : bp_pick<2 * G::MAXNB>(a.Wt, s_cv));
      }
    }
  };

  // ---- fragments ----
  uint32_t wP, wQ;
  {
    const int row0 = 8 * (lm >> 2) + (lm & 3);
    const int c0 = lg ^ paired_w_key(row0);
    wP = wring + (uint32_t)(wc * 4096 + row0 * 128 + (c0 << 4));
    wQ = wring + (uint32_t)(wc * 4096 + row0 * 128 + ((c0 ^ 4) << 4));
  }
  frag16 fa[MI][2], fb[2][2];
  auto read_b = [&](auto slot_tag) {
    constexpr int SO = decltype(slot_tag)::value * G::Q_B;
    lds_read_b128<SO>(fb[0][0], wP); lds_read_b128<SO + 512>(fb[1][0], wQ);
    lds_read_b128<SO>(fb[0][1], wQ); lds_read_b128<SO + 512>(fb[1][1], wP);
  };
  // per row group: the k-step 0 address of the tap's row in channel block 0; k-step 1 is the same row at chunk ^ 4
  auto read_a = [&](auto off_tag, const uint32_t (&a0)[MI]) {
    constexpr int AO = decltype(off_tag)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i) lds_read_b128<AO>(fa[i][0], a0[i]);
#pragma unroll
    for (int i = 0; i < MI; ++i) lds_read_b128<AO>(fa[i][1], a0[i] ^ 64u);
  };

  // ---- per-lane geometry of the four row groups (the same for every tile) ----
  int rowpad[MI];                              // region row of MFMA row m = wr*64 + 16 i + lm (its interior position in the padded map)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int m = wr * 64 + i * 16 + lm;
    const int im = m / G::HW, rem = m - im * G::HW;
    const int y = rem / G::W, x = rem - y * G::W;
    const bool real = im < G::IPT;
    rowpad[i] = real ? im * G::IMGP + (y + 1) * G::PW + (x + 1) : G::PW + 1;   // the tile's RM - IPT*HW spare rows read a valid row, write nothing
  }

  f32x4_t acc[MI][NI];
  // bias | slope of every convolution wait in LDS behind the ring (table 2 cv = bias, 2 cv + 1 = slope; TAIL: conv A's pair behind
  // them); the accumulators start from the convolution's bias
  uint32_t tab_ad[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) tab_ad[j] = lds_base + G::TAB_OFF + (uint32_t)((wc * 64 + 32 * (j >> 1) + 8 * lg + 4 * (j & 1)) * 4);
  constexpr int TAB_A = 4 * G::MAXNB;          // TAIL: conv A's bias, + 1 its slope
  for (int t = tid; t < (2 * ncv + (TAIL ? 2 : 0)) * CH; t += 512) {
    const int q = t / CH, c = t - q * CH;
    const float* src = q < 2 * ncv ? ((q & 1) ? bp_pick<2 * G::MAXNB>(a.slope, q >> 1) : bp_pick<2 * G::MAXNB>(a.bias, q >> 1)) : (q == 2 * ncv ? a.bA : a.sA);
    const int slot = q < 2 * ncv ? q : TAB_A + (q - 2 * ncv);
    reinterpret_cast<float*>(lds)[G::TAB_OFF / 4 + slot * CH + c] = src[c];
  }
  __syncthreads();
  auto read_tab = [&](f32x4_t (&v)[NI], const int which) {
    uint32_t ad[4];
#pragma unroll
    for (int j = 0; j < NI; ++j) ad[j] = tab_ad[j] + (uint32_t)(which * CH * 4);
    bp_read4_f4_sync(v, ad);
  };
  auto init_acc = [&](const int which) {
    f32x4_t bj[NI];
    read_tab(bj, which);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = bj[j];
  };

  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using IB = std::integral_constant<int, G::BLK_B>;
  uint32_t a0[MI];
  // one phase = one quarter (ring slot SLOT): half H of the wave's 64 columns x all 64 rows x K = 64
  auto phase = [&](auto h_tag, auto slot_tag, auto aoff_tag) {
    constexpr int H = decltype(h_tag)::value, SLOT = decltype(slot_tag)::value;
    read_b(slot_tag);
    if (H == 0) { __builtin_amdgcn_sched_barrier(0); read_a(aoff_tag, a0); }
    stage_one(std::integral_constant<int, (SLOT + 2) & 3>{}, h_tag);   // quarter g+2 (the same half) -> the slot of quarter g-2
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<1>();                           // quarter g+1 (staged one phase ago) has landed: read one barrier from now
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    lds_wait();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        acc[i][2 * H + s2] = ET::mfma(fb[s2][0], fa[i][0], acc[i][2 * H + s2]);
        acc[i][2 * H + s2] = ET::mfma(fb[s2][1], fa[i][1], acc[i][2 * H + s2]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);         // (nothing between the last MFMA and the barrier, nothing hoisted above it: respair_phase.hip)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tap_addr = [&](int shift) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int pr = rowpad[i] + shift;
      a0[i] = lds_base + (uint32_t)(pr * 128 + ((lg ^ (pr & 7)) << 4));
    }
  };
  // one convolution out of the region (every conv reads position p + (ky - 1) PW + (kx - 1) of the rows its input sits in).
  // CH = 128: a tap is one ring turn.  CH = 64: a tap is two phases; taps in pairs, the ninth alone - a convolution that starts on
  // slot 0 leaves the ring half a turn on, the next one starts on slot 2 (`odd_tag`) and brings it back
  auto run_conv = [&](auto odd_tag) {
    constexpr int S0 = (CH == 64 && decltype(odd_tag)::value) ? 2 : 0;
    using SA = std::integral_constant<int, S0>; using SB = std::integral_constant<int, S0 + 1>;
    using SC = std::integral_constant<int, S0 ^ 2>; using SD = std::integral_constant<int, (S0 ^ 2) + 1>;
    if constexpr (CH == 64) {
      constexpr int sh[9] = {-(G::PW + 1), -G::PW, -(G::PW - 1), -1, 0, 1, G::PW - 1, G::PW, G::PW + 1};
      for (int tp = 0; tp < 4; ++tp) {
        const int s_a = tp == 0 ? sh[0] : tp == 1 ? sh[2] : tp == 2 ? sh[4] : sh[6];
        const int s_b = tp == 0 ? sh[1] : tp == 1 ? sh[3] : tp == 2 ? sh[5] : sh[7];
        tap_addr(s_a);
        phase(I0{}, SA{}, I0{}); phase(I1{}, SB{}, I0{});
        tap_addr(s_b);
        phase(I0{}, SC{}, I0{}); phase(I1{}, SD{}, I0{});
      }
      tap_addr(sh[8]);
      phase(I0{}, SA{}, I0{}); phase(I1{}, SB{}, I0{});
    } else {
      int shift = -(G::PW + 1);
      for (int ky = 0; ky < 3; ++ky, shift += G::PW - 3)
        for (int kx = 0; kx < 3; ++kx, ++shift) {
          tap_addr(shift);
          phase(I0{}, I0{}, I0{}); phase(I1{}, I1{}, I0{});      // channel block 0
          phase(I0{}, I2{}, IB{}); phase(I1{}, I3{}, I0{});      // 1
        }
    }
  };
  // TAIL: the downsample's K-tile (Wd against the stage input at the even positions, the third LDS block) and the zero K-tile that
  // completes the ring turn (run against the same finite rows)
  auto run_xs = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int m = wr * 64 + i * 16 + lm;
      a0[i] = lds_base + (uint32_t)(G::REGION + m * 128 + ((lg ^ (m & 7)) << 4));
    }
    phase(I0{}, I0{}, I0{}); phase(I1{}, I1{}, I0{});
    phase(I0{}, I2{}, I0{}); phase(I1{}, I3{}, I0{});
  };

  // this lane's (row group, column half 0) chunk in the region: its residual rows, later its result rows (half 1 = chunk ^ 4)
  // (formed from rowpad where it is used - three hand-overs and a read per tile: four registers less across the tap loops)
  auto chunk_ad = [&](int i) -> uint32_t {
    int rp = rowpad[i];
    asm volatile("" : "+v"(rp));               // (opaque: hipcc otherwise hoists the four sums again - and spills three of them)
    return lds_base + (uint32_t)wc * G::BLK_B + (uint32_t)rp * 128 + (uint32_t)((lg ^ (rp & 7)) << 4);
  };
  // output: row m of the tile is row img0*HW + m of Y; this lane's bytes inside a row: (wc*64 + 32 h + 8 lg) * 2
  const uint32_t y_lane = (uint32_t)(((wr * 64 + lm) * CH + wc * 64 + 8 * lg) * 2);

  init_acc(TAIL ? TAB_A : 0);
  issue_patch(0);
  stage_one(I0{}, I0{});
  stage_one(I1{}, I1{});
  for (int c_i = 0; c_i < my_n; ++c_i) {
    const int img0 = ((int)blockIdx.x + c_i * (int)gridDim.x) * G::IPT;
    // ---- tile start (the wave rows are level here): the patch and the first two quarters are visible to every wave ----
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (upper) __builtin_amdgcn_s_barrier();               // the upper wave rows run one barrier behind from here on
    u32x4_t resp[MI][2];                                   // the current block's residual rows, 8 channels per (row group, column half)
    // prelu(acc (+ residual rows), slope table `which`) as packed 16-bit rows in the paired layout
    auto activate = [&](u32x4_t (&o16)[MI][2], const int which, auto res_tag) {
      constexpr bool RES = decltype(res_tag)::value;
      f32x4_t sj[NI];
      read_tab(sj, which);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint32_t rw[4] = {0u, 0u, 0u, 0u};
          if constexpr (RES) { const u32x4_t r = resp[i][h]; rw[0] = r.x; rw[1] = r.y; rw[2] = r.z; rw[3] = r.w; }
          uint32_t w[4];
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const f32x4_t v = acc[i][2 * h + s2], s = sj[2 * h + s2];      // bias already inside
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = v[e];
              if (RES) x += ET::to_f32((uint16_t)((rw[2 * s2 + (e >> 1)] >> ((e & 1) * 16)) & 0xffff));
              o[e] = fmaxf(x, 0.f) + fminf(x, 0.f) * s[e];
            }
            w[2 * s2] = ET::pack2(o[0], o[1]);
            w[2 * s2 + 1] = ET::pack2(o[2], o[3]);
          }
          o16[i][h] = u32x4_t{w[0], w[1], w[2], w[3]};
        }
    };
    // a convolution's 16-bit result into the region rows it was computed from (the spare rows point at a real row: they write nothing),
    // the next convolution's bias into the accumulators, then the stagger again
    auto hand_over = [&](const u32x4_t (&o16)[MI][2], const int next_bias) {
      __builtin_amdgcn_s_barrier();                        // every wave is done reading the rows
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < MI; ++i)
        if (wr * 64 + i * 16 + lm < G::IPT * G::HW) {
          const uint32_t ca = chunk_ad(i);
          bp_write_u4(ca, o16[i][0]);
          bp_write_u4(ca ^ 64u, o16[i][1]);
        }
      init_acc(next_bias);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // the rows are visible
      asm volatile("" ::: "memory");
      if (upper) __builtin_amdgcn_s_barrier();             // stagger again
    };
    using NoRes = std::false_type; using Res = std::true_type;
    if constexpr (TAIL) {
      // ---- conv A: nine taps out of t0 + the downsample's K-tile; out0 = prelu(.) stays in registers as the block's residual and
      // replaces t0 in the region ----
      run_conv(I0{});
      run_xs();
      if (!upper) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      activate(resp, TAB_A + 1, NoRes{});
      hand_over(resp, 0);
    }
    for (int bi = 0; bi < a.nb; ++bi) {
      const bool last = bi + 1 == a.nb;
      run_conv(I0{});
      // ---- conv1 done.  Level the rows, (first block) take the residual rows x to registers, then t1 = prelu1(conv1 + b1) into the
      // same rows ----
      if (!upper) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!TAIL && bi == 0) {
        uint32_t ad[MI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) { ad[i][0] = chunk_ad(i); ad[i][1] = ad[i][0] ^ 64u; }
        bp_read8_u4_sync(resp, ad);
      }
      {
        u32x4_t t1v[MI][2];
        activate(t1v, 4 * bi + 1, NoRes{});
        hand_over(t1v, 4 * bi + 2);
      }
      run_conv(I1{});
      // ---- conv2 done: level the rows; the region is free once every wave has finished reading t1 ----
      if (!upper) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!last) {
        // the block's output: the next block's residual (registers) and input (region; border rows keep their zeros)
        activate(resp, 4 * bi + 3, Res{});                 // (in place: an element's residual is read before it is overwritten)
        hand_over(resp, 4 * bi + 4);
        continue;
      }
      if (c_i + 1 < my_n) issue_patch(c_i + 1);
      // out = prelu2(conv2 + b2 + x): 16 bytes per lane and (row group, column half)
      activate(resp, 4 * bi + 3, Res{});
      int img_e = img0;
      asm volatile("" : "+s"(img_e));                      // (opaque: the store addresses are formed here, not at the tile start and spilled)
      int left = a.nimg - img_e;
      left = left < G::IPT ? left : G::IPT;
      const int mlim = left * G::HW;                       // rows of the tile that exist
      char* yb = reinterpret_cast<char*>(a.Y) + (int64_t)img_e * G::HW * CH * 2;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // (opaque inputs: the per-store rows and zero-extended lane offsets are otherwise formed once per kernel, spilled, and
          // reloaded here behind an s_waitcnt vmcnt(0) that also drains the patch DMA issued just above)
          int l_o = lm;
          uint32_t yl = y_lane;
          asm volatile("" : "+v"(l_o), "+v"(yl));
          const int mr = wr * 64 + i * 16 + l_o;
          if (mr < mlim)
            *reinterpret_cast<uint4*>(yb + (yl + (uint32_t)((i * 16 * CH + 32 * h) * 2))) =
                make_uint4(resp[i][h].x, resp[i][h].y, resp[i][h].z, resp[i][h].w);
        }
    }
    init_acc(TAIL ? TAB_A : 0);
  }
  wait_vmcnt<0>();                             // no LDS-DMA (the trailing dummies) may outlive the block
}

template <typename ET, int CH_, int H_, int W_, bool TAIL>
int launch_bp(const BpArgs& a, hipStream_t st) {
  using G = PGeo<CH_, H_, W_, TAIL>;
  if (a.nb < 1 || a.nb > G::MAXNB) return L2S_EUNSUPPORTED;
  auto kern = basicblock_phase_kernel<ET, CH_, H_, W_, TAIL>;
  static L2sSmemOptIn opt_in;
  if (int e = l2s_smem_opt_in(kern, G::SMEM, opt_in)) return e;
  BpArgs b = a;
  b.ntiles = (a.nimg + G::IPT - 1) / G::IPT;
  const int grid = b.ntiles < 256 ? b.ntiles : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::SMEM, st, b);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

}  // namespace

// l2s_basicblock_fused / l2s_basiclayer_fused families built on this kernel (basicblock.hip dispatches here): the map sizes per C
bool l2s_basicblock_phase_supports(int C, int H, int W) { return (C == 128 && H == 11 && W == 11) || (C == 64 && H == 22 && W == 22); }

// w / bias / slope: 2 * n_blocks host arrays of device pointers (conv1, conv2 of block 0, ...)
int l2s_basicblock_phase_launch(const void* x, const void* const* w, const float* const* bias, const float* const* slope, int n_blocks,
                                void* y, int n_images, int H, int W, int C, int dtype, hipStream_t st) {
  if (!l2s_basicblock_phase_supports(C, H, W)) return L2S_EUNSUPPORTED;
  if (n_blocks < 1 || 2 * n_blocks > BP_MAXCV) return L2S_EUNSUPPORTED;
  BpArgs a = {};
  a.X = (const uint16_t*)x; a.Y = (uint16_t*)y; a.nimg = n_images; a.nb = n_blocks;
  for (int i = 0; i < BP_MAXCV; ++i) {
    const int j = i < 2 * n_blocks ? i : 0;
    a.Wt[i] = (const uint16_t*)w[j]; a.bias[i] = bias[j]; a.slope[i] = slope[j];
  }
  if (C == 128) {
    if (dtype == L2S_F16) return launch_bp<ElemF16, 128, 11, 11, false>(a, st);
    if (dtype == L2S_BF16) return launch_bp<ElemBF16, 128, 11, 11, false>(a, st);
  } else {
    if (dtype == L2S_F16) return launch_bp<ElemF16, 64, 22, 22, false>(a, st);
    if (dtype == L2S_BF16) return launch_bp<ElemBF16, 64, 22, 22, false>(a, st);
  }
  return L2S_EINVAL;
}

/*
 * include/lip2speech_hip.h: l2s_basicstage128_tail_fused
 */
extern "C" int l2s_basicstage128_tail_fused(const void* x0, const void* t0, const void* wa, const float* ba, const float* sa,
                                            const void* w1, const float* b1, const float* s1, const void* w2, const float* b2,
                                            const float* s2, void* y, int n_images, int H, int W, int dtype, void* stream) {
  if (!x0 || !t0 || !wa || !ba || !sa || !w1 || !b1 || !s1 || !w2 || !b2 || !s2 || !y) return L2S_EINVAL;
  if (n_images <= 0 || H <= 0 || W <= 0) return L2S_ESHAPE;
  if (!l2s_basicblock_phase_supports(128, H, W)) return L2S_EUNSUPPORTED;
  const void* al[] = {x0, t0, wa, ba, sa, w1, b1, s1, w2, b2, s2, y};
  for (const void* q : al)
    if ((uintptr_t)q & 15) return L2S_EALIGN;
  if ((int64_t)n_images * H * W * 4 * 64 >= ((int64_t)1 << 31)) return L2S_EUNSUPPORTED;
  BpArgs a = {};
  a.X = (const uint16_t*)t0; a.Y = (uint16_t*)y; a.nimg = n_images; a.nb = 1;
  for (int i = 0; i < BP_MAXCV; ++i) {
    a.Wt[i] = (const uint16_t*)((i & 1) ? w2 : w1); a.bias[i] = (i & 1) ? b2 : b1; a.slope[i] = (i & 1) ? s2 : s1;
  }
  a.X0 = (const uint16_t*)x0; a.WA = (const uint16_t*)wa; a.bA = ba; a.sA = sa;
  if (dtype == L2S_F16) return launch_bp<ElemF16, 128, 11, 11, true>(a, (hipStream_t)stream);
  if (dtype == L2S_BF16) return launch_bp<ElemBF16, 128, 11, 11, true>(a, (hipStream_t)stream);
  return L2S_EINVAL;
}
