// Fused convolution PAIR of a HiFi-GAN ResBlock1 for the C = 256 vocoder stage (2 000 samples per 4-s clip):
//   x' = c2(leaky_relu(c1(leaky_relu(x)))) + x        speech-resynthesis/models.py:34-41, one (c1, c2, d) step; :103-109 the sum.
//
// respair.hip keeps a 192-row patch of C <= 128 channels plus a ring of whole (tap, 64-channel) weight tiles in LDS; at
// C = 256 neither fits (a 256-channel patch of 192 + 50 rows is 121 KB, one weight tile 32 KB), so the stage ran as 18 tap-GEMM
// launches that moved t1, x, x' and leaky_relu(x') through HBM and re-read every input row once per tap (PMC: 2.6 x the
// algorithmic bytes on the first convolutions).  This kernel is respair's data flow on the phase-staggered schedule of
// phasegemm_kernel.h:
//   * one block owns 128 time steps of one clip x all 256 channels: the patch (128 + 2*h1 <= 178 rows of the LeakyReLU'd
//     input, four 64-channel blocks of 184 x 128 B = 92 KB) arrives by LDS-DMA; conv1 runs out of it (tap = row shift), t1 goes
//     back into the same region as 16-bit rows, conv2 runs out of t1; HBM sees the pair's input once (+ the halo) and its output
//     once, the raw x of the residual is recovered from the LeakyReLU'd copy by the inverse of leaky_relu;
//   * 8 waves = 2 wave rows x 4 wave columns of 64 x 64 (MI = NI = 4).  Only the WEIGHTS are streamed: a K-tile (tap, 64 input
//     channels) = 256 x 64 weights = two 16 KB QUARTERS (the first / second 32 output channels of every wave column), four
//     quarter slots = 64 KB.  A PHASE is one quarter: 16 MFMAs per wave (4 row groups x 2 column blocks x K = 64) between two
//     raw barriers; the first phase of a K-tile reads 8 A + 4 W fragments, the second 4 W fragments.  The upper wave row
//     (waves 4-7, the second wave of every SIMD) runs ONE BARRIER BEHIND the lower one, so one wave of a SIMD issues MFMAs
//     while the other issues fragment reads and LDS-DMA.  Quarter e+2 is staged in phase e (into the slot of quarter e-2, whose
//     last read - by the upper row - retired one barrier earlier), `s_waitcnt vmcnt(2)` in phase e retires quarter e+1 one
//     barrier before anybody reads it;
//   * the stream runs across conv1 -> conv2 -> the block's next tile without draining; at the two points where the region
//     changes hands (patch -> t1, t1 -> next patch) the wave rows are brought level for the hand-over and staggered again;
//   * mid pairs read the weight fragments in the PAIRED row order of tapgemm_common.h, so a lane ends with 8 consecutive
//     channels: t1 rows are written back by ds_write_b128 and leaky_relu(x') leaves through epilogue_direct16 as whole 128-byte
//     lines, no LDS scratch (there is none left: 92 + 64 KB); last pairs keep the plain order, in which a lane owns 4
//     consecutive fp32 channels of the ResBlock sum (16-byte accesses).
#include "tapgemm_common.h"
#include "respair_args.h"
#include <cstdlib>

using namespace l2s;
using l2s_rp::RpArgs;

namespace {

constexpr int XT1 = 8;                       // t1 row p lives at region row p + 8: conv2's taps reach rows p - h2 >= -5
constexpr int XRQ = 4;                       // quarter slots
// Geometry per channel count.  8 waves of 64 x 64: CH / 64 wave columns, the rest wave rows; the tile is as tall as the wave rows.
//   CH = 256: 2 x 4 waves, 128-row tiles, region 4 x 184 rows (92 KB), 16 KB quarters (2 DMA instructions per wave and phase)
//   CH = 128: 4 x 2 waves, 256-row tiles, region 2 x 312 rows (78 KB),  8 KB quarters (1 DMA instruction per wave and phase)
//   CH =  64: 8 x 1 waves, 512-row tiles, region 1 x 568 rows (71 KB),  4 KB quarters (1 DMA instruction per EVEN wave and phase);
//             a tap is two phases = half a turn of the ring, so taps go in pairs and conv2 (k is odd) starts on slot 2
template <int CH>
struct XGeo {
  static constexpr int NWC = CH / 64, NWR = 8 / NWC, NBLK = CH / 64;
  static constexpr int RM = NWR * 64;              // rows both convolutions compute per tile
  static constexpr int RPR = RM + 56;              // rows of a region block: the patch (RM + 2*h1 <= RM + 50), later t1 (rows 8 .. 8+RM)
  static constexpr int BLK_B = RPR * 128;          // one 64-channel block of the region
  static constexpr int REGION = NBLK * BLK_B;
  static constexpr int QROWS = NWC * 32;           // weight rows of a quarter: 32 per wave column
  static constexpr int Q_B = QROWS * 128;          // one weight quarter: QROWS rows x 64 K values
  static constexpr int DPW = QROWS >= 64 ? QROWS / 64 : 1;   // LDS-DMA instructions per staging wave and quarter (8 rows each)
  static constexpr int RPS = DPW * 8;              // quarter rows per staging wave (all 8 waves stage; at CH = 64 the even ones)
  static constexpr int BIAS_OFF = REGION + XRQ * Q_B;   // b1 | b2 as fp32 behind the ring
  static constexpr int SMEM = BIAS_OFF + 2 * CH * 4;
  static constexpr int PI = NBLK * (RPR / 8);      // patch DMA instructions per tile (8 rows x 128 B each): 92 / 78
  static constexpr int PPW = (PI + 7) / 8;         // ... per wave: 12 / 10
  static_assert(RPR % 8 == 0 && (CH == 256 || (NBLK - 1) * BLK_B + 6144 < 65536), "geometry");
};

__device__ __forceinline__ void lds_write_u4(uint32_t addr, u32x4_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// Reads whose results must survive a long stretch of code (the residual rows: held across the whole second convolution) carry
// their own wait: after an asm LDS read without one the compiler believes the destination is valid at once, and under
// register pressure it may spill the register BEFORE the separate s_waitcnt - the spill slot then holds the stale contents
// (seen with the last-pair epilogue: garbage in exactly the spilled (row group, block) entries).
__device__ __forceinline__ void lds_read4_u2_sync(u32x2_t (&v)[4], const uint32_t (&ad)[4]) {
  asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]));
}
__device__ __forceinline__ void lds_read8_u4_sync(u32x4_t (&v)[4][2], const uint32_t (&ad)[4][2]) {
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %10\n\tds_read_b128 %3, %11\n\t"
               "ds_read_b128 %4, %12\n\tds_read_b128 %5, %13\n\tds_read_b128 %6, %14\n\tds_read_b128 %7, %15\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]), "=&v"(v[2][0]), "=&v"(v[2][1]), "=&v"(v[3][0]), "=&v"(v[3][1])
               : "v"(ad[0][0]), "v"(ad[0][1]), "v"(ad[1][0]), "v"(ad[1][1]), "v"(ad[2][0]), "v"(ad[2][1]), "v"(ad[3][0]), "v"(ad[3][1]));
}
__device__ __forceinline__ void lds_read4_f4_sync(f32x4_t (&v)[4], const uint32_t (&ad)[4]) {
  asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]));
}
__device__ __forceinline__ void lds_write_u2(uint32_t addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// Diagnostic build (-DL2S_PAIR_STAMPS, tools/pair_stamps.py): waves 0 and 7 of every block accumulate s_memtime deltas of
// [0] the wait at the tile start (patch + first quarters), [1] conv1, [2] the patch -> t1 hand-over (after [7], the wait that levels
// the wave rows), [3] conv2, [5] the levelling wait behind conv2, [6] the next patch's DMA issue, [4] the epilogue proper.
#ifdef L2S_PAIR_STAMPS
__device__ unsigned long long* g_pair_stamps = nullptr;
#define PRSTAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pr_acc[i] += now_ - pr_last; pr_last = now_; }
#else
#define PRSTAMP(i)
#endif

// KIND 0: mid pair (Y = leaky_relu(x'));  KIND 1: last pair of a ResBlock (XS (+)= x', optional Y = leaky_relu(XS))
template <typename ET, int CH, int KIND>
__global__ __launch_bounds__(512) void respair_phase_kernel(const RpArgs a) {
  using G = XGeo<CH>;
  constexpr int MI = 4, NI = 4;
  constexpr bool PAIRED = KIND == 0;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / G::NWC, wc = wave % G::NWC;
  const bool upper = wave >= 4;                // the second wave of every SIMD: runs one barrier behind
  const int lm = lane & 15, lg = lane >> 4;
  const int srow = lane >> 3, schunk = (lane & 7) ^ (srow & 7);
  const int k = a.k, dil = a.dil, h1 = a.h1, h2 = a.h2, T = a.T;
  const int Ktot = k * CH;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);

  // ---- tiles of this block (respair.hip's XCD-aware order: blocks b and b+8 share an L2 and walk neighbouring tiles) ----
  int my_n = 0;
  if (!a.xcd_order) {
    my_n = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  } else {
    const int per = (a.ntiles + 7) >> 3, lo = ((int)blockIdx.x & 7) * per;
    int hi = lo + per;
    hi = hi < a.ntiles ? hi : a.ntiles;
    const int gxx = ((int)gridDim.x + 7) >> 3, b8 = (int)blockIdx.x >> 3;
    if (lo + b8 < hi) my_n = (hi - lo - b8 + gxx - 1) / gxx;
  }
  if (my_n <= 0) return;
  const int nkt = k * G::NBLK;               // K-tiles per convolution: (tap, 64-channel block)
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = (gridDim.x + 7) >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  auto tile_origin = [&](int i, int& unit, int& g0) {
    const int L = a.xcd_order ? xcd * per_xcd + bx + i * gx : (int)blockIdx.x + i * (int)gridDim.x;
    unit = L / a.tiles_per_clip;
    g0 = (L - unit * a.tiles_per_clip) * a.S - h2;      // global time of conv row 0 (t1 row 0 / output row 0)
  };

  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t wring = lds_base + G::REGION;

  // ---- patch: PI instructions of 8 rows x 128 B, PPW per wave (the last wave has fewer); rows past the patch are not fetched ----
  const int patch_rows = G::RM + 2 * h1;
  auto issue_patch = [&](int i) {
    int unit, g0;
    tile_origin(i, unit, g0);
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
      const int instr = wave * G::PPW + j;
      const int cq = instr / (G::RPR / 8), blk = instr - cq * (G::RPR / 8);
      if (instr < G::PI && blk * 8 < patch_rows) {
        const int ts = g0 - h1 + blk * 8 + srow;
        const uint16_t* g = ((unsigned)ts < (unsigned)T) ? a.X + ((int64_t)unit * T + ts) * CH + cq * 64 + schunk * 8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + cq * (G::BLK_B / 2) + blk * 512), 16, 0, 0);
      }
    }
  };

  // ---- weight stream: quarter (conv, K-tile, half h) = rows {wc*64 + 32 h + 0..31}; wave w stages quarter rows
  // w * QROWS/8 .. + QROWS/8 - 1 (16 or 8: two or one instruction of 8 rows) ----
  // source = uniform cursor (SGPR pair) + one per-lane 32-bit byte offset per instruction: no 64-bit vector arithmetic in the loop
  const bool stager = CH >= 128 || !(wave & 1);
  const int sw = CH >= 128 ? wave : wave >> 1;                   // index among the staging waves
  uint32_t w_lane[G::DPW];
#pragma unroll
  for (int j = 0; j < G::DPW; ++j) {
    const int qrow = sw * G::RPS + 8 * j + srow;                 // row of the quarter: wave column qrow / 32, row qrow % 32 of its share
    const int within = qrow & 31;
    const int n = (qrow >> 5) * 64 + within;
    const int chunk = PAIRED ? ((lane & 7) ^ paired_w_key(within)) : schunk;
    w_lane[j] = (uint32_t)(n * Ktot + chunk * 8) * 2u;
  }
  const uint32_t h_bytes = (uint32_t)(32 * Ktot) * 2u;
  // cursor: the K-tile being staged (uniform pointer to its first half) - it simply keeps cycling W1 -> W2 -> W1 ..., so the
  // stagings past the block's last quarter re-fetch the first quarters of W1 into slots nobody reads (valid memory, exact counts)
  const char* kt_ptr = (const char*)a.W1;
  int s_kt = 0, s_conv = 0;
  // exactly DPW LDS-DMA instructions per wave into slot DSLOT (a compile-time LDS address: M0 is one s_mov) for half SH of the K-tile
  auto stage_one = [&](auto dslot_tag, auto sh_tag) {
    constexpr int DSLOT = decltype(dslot_tag)::value, SH = decltype(sh_tag)::value;
    const char* wb = kt_ptr + (size_t)(SH ? h_bytes : 0u);
    uint16_t* dst = lds + G::REGION / 2 + DSLOT * (G::Q_B / 2) + sw * G::RPS * 64;
#pragma unroll
    for (int j = 0; j < G::DPW; ++j) {
#ifndef L2S_PAIR_ABL_NODMA     // (diagnostic, timing only) no weight staging
      if (stager) __builtin_amdgcn_global_load_lds((gptr_t)(wb + (size_t)w_lane[j]), (lptr_t)(dst + j * 512), 16, 0, 0);
#else
      asm volatile("" ::"v"(wb + (size_t)w_lane[j]), "v"(dst));
#endif
    }
    if constexpr (SH == 1) {
      kt_ptr += 128;
      if (++s_kt == nkt) { s_kt = 0; s_conv ^= 1; kt_ptr = (const char*)(s_conv ? a.W2 : a.W1); }
    }
  };

  // ---- fragments: two per-lane bases each for W and A, everything else is an immediate offset ----
  // W quarter, this wave column's 32 rows (4 KB).  plain order: block s, k-step ks at  s*2048 + lm*128 + ((4ks + lg) ^ (lm & 7))*16;
  // paired order (paired_w_off): row 8 (lm >> 2) + 4 s + (lm & 3), chunk (lg ^ key0) ^ 4 (ks ^ s): with c0 = lg ^ key0 the four
  // fragments sit at P, Q (ks = 1), Q + 512 (s = 1), P + 512 (s = 1, ks = 1) for P = row0 + c0*16, Q = row0 + (c0 ^ 4)*16
  uint32_t wP, wQ;
  if constexpr (PAIRED) {
    const int row0 = 8 * (lm >> 2) + (lm & 3);
    const int c0 = lg ^ paired_w_key(row0);
    wP = wring + (uint32_t)(wc * 4096 + row0 * 128 + (c0 << 4));
    wQ = wring + (uint32_t)(wc * 4096 + row0 * 128 + ((c0 ^ 4) << 4));
  } else {
    wP = wring + (uint32_t)(wc * 4096 + lm * 128 + (((0 + lg) ^ (lm & 7)) << 4));
    wQ = wring + (uint32_t)(wc * 4096 + lm * 128 + (((4 + lg) ^ (lm & 7)) << 4));
  }
  frag16 fa[MI][2], fb[2][2];
  auto read_b = [&](auto slot_tag) {
    constexpr int SO = decltype(slot_tag)::value * G::Q_B;
    if constexpr (PAIRED) {
      lds_read_b128<SO>(fb[0][0], wP); lds_read_b128<SO + 512>(fb[1][0], wQ);
      lds_read_b128<SO>(fb[0][1], wQ); lds_read_b128<SO + 512>(fb[1][1], wP);
    } else {
      lds_read_b128<SO>(fb[0][0], wP); lds_read_b128<SO + 2048>(fb[1][0], wP);
      lds_read_b128<SO>(fb[0][1], wQ); lds_read_b128<SO + 2048>(fb[1][1], wQ);
    }
  };
  auto read_a = [&](auto off_tag, uint32_t a0, uint32_t a1) {
    constexpr int AO = decltype(off_tag)::value;
    lds_read_b128<AO>(fa[0][0], a0); lds_read_b128<AO + 2048>(fa[1][0], a0); lds_read_b128<AO + 4096>(fa[2][0], a0); lds_read_b128<AO + 6144>(fa[3][0], a0);
    lds_read_b128<AO>(fa[0][1], a1); lds_read_b128<AO + 2048>(fa[1][1], a1); lds_read_b128<AO + 4096>(fa[2][1], a1); lds_read_b128<AO + 6144>(fa[3][1], a1);
  };

  f32x4_t acc[MI][NI];
  // the accumulators start from the convolution's bias (b1 before conv1, b2 before conv2): no bias pass in either epilogue.
  // The two bias vectors wait in LDS behind the ring (2 KB) instead of 32 registers held across both tap loops.
  uint32_t bias_ad[NI];
  auto init_acc = [&](const int which) {
    uint32_t ad[4];
#pragma unroll
    for (int j = 0; j < NI; ++j) ad[j] = bias_ad[j] + (uint32_t)(which * CH * 4);
    f32x4_t bj[NI];
    lds_read4_f4_sync(bj, ad);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = bj[j];
  };

  // one phase = one quarter (ring slot SLOT, a compile-time constant: a tap is 8 phases = two turns of the ring, and every
  // convolution starts on slot 0): half H of the wave's 64 columns x all 64 rows x K = 64
  auto phase = [&](auto h_tag, auto slot_tag, auto aoff_tag, uint32_t a0, uint32_t a1) {
    constexpr int H = decltype(h_tag)::value, SLOT = decltype(slot_tag)::value;
#ifndef L2S_PAIR_ABL_NOREAD    // (diagnostic, timing only) no fragment reads
    read_b(slot_tag);
    if (H == 0) { __builtin_amdgcn_sched_barrier(0); read_a(aoff_tag, a0, a1); }
#endif
    stage_one(std::integral_constant<int, (SLOT + 2) & (XRQ - 1)>{}, h_tag);   // quarter g+2 (the same half) -> the slot of quarter g-2
    __builtin_amdgcn_sched_barrier(0);         // (the staging cursor's bookkeeping stays in front of the wait, off the MFMA path)
    wait_vmcnt<G::DPW>();                      // quarter g+1 (staged one phase ago) has landed: read one barrier from now
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    lds_wait();
    __builtin_amdgcn_s_setprio(1);
#ifndef L2S_PAIR_ABL_NOMFMA    // (diagnostic, timing only) no matrix instructions
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        acc[i][2 * H + s2] = ET::mfma(fb[s2][0], fa[i][0], acc[i][2 * H + s2]);
        acc[i][2 * H + s2] = ET::mfma(fb[s2][1], fa[i][1], acc[i][2 * H + s2]);
      }
#endif
    __builtin_amdgcn_s_setprio(0);
    // nothing may sit between the last MFMA and the barrier: the partner wave of this SIMD starts its MFMAs behind it.  Without the
    // second fence hipcc hoists the next phase's address arithmetic (16 SALU / VALU instructions, ~80 cycles) above the barrier
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // one convolution out of the region: conv 0 reads the patch (row shift tap*dil), conv 1 reads t1 (shift tap - h2 + 8).
  // Per tap two per-lane row addresses (k-steps 0 / 1) for channel blocks 0-1 and two for blocks 2-3 (the 16-bit offset field)
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using IB = std::integral_constant<int, G::BLK_B>;
  auto run_conv = [&](auto conv_tag) {
    constexpr int conv = decltype(conv_tag)::value;
    auto row_addr = [&](int tap, uint32_t& a0, uint32_t& a1) {
      const int pr = wr * 64 + lm + (conv == 0 ? tap * dil : tap - h2 + XT1);
      const int x = pr & 7;
      const uint32_t pa = lds_base + (uint32_t)pr * 128;
      a0 = pa + (uint32_t)(((0 + lg) ^ x) << 4);
      a1 = pa + (uint32_t)(((4 + lg) ^ x) << 4);
    };
    if constexpr (CH == 64) {
      // one channel block: a tap is two phases; taps in pairs (one turn of the ring), the odd last tap alone.  conv1 starts on
      // slot 0 and leaves the ring half a turn on (k is odd), conv2 starts on slot 2 and brings it back to 0
      using SA = std::integral_constant<int, conv ? 2 : 0>; using SB = std::integral_constant<int, conv ? 3 : 1>;
      using SC = std::integral_constant<int, conv ? 0 : 2>; using SD = std::integral_constant<int, conv ? 1 : 3>;
      int tap = 0;
      for (; tap + 1 < k; tap += 2) {
        uint32_t a0, a1, c0, c1;
        row_addr(tap, a0, a1);
        row_addr(tap + 1, c0, c1);
        phase(I0{}, SA{}, I0{}, a0, a1); phase(I1{}, SB{}, I0{}, 0u, 0u);
        phase(I0{}, SC{}, I0{}, c0, c1); phase(I1{}, SD{}, I0{}, 0u, 0u);
      }
      uint32_t a0, a1;
      row_addr(tap, a0, a1);
      phase(I0{}, SA{}, I0{}, a0, a1); phase(I1{}, SB{}, I0{}, 0u, 0u);
    } else {
      for (int tap = 0; tap < k; ++tap) {
        uint32_t a0, a1;
        row_addr(tap, a0, a1);
        phase(I0{}, I0{}, I0{}, a0, a1); phase(I1{}, I1{}, I0{}, 0u, 0u);      // channel block 0
        phase(I0{}, I2{}, IB{}, a0, a1); phase(I1{}, I3{}, I0{}, 0u, 0u);      // 1
        if constexpr (CH == 256) {                                              // (a tap is two turns of the ring; at 128: one)
          const uint32_t c0 = a0 + 2 * G::BLK_B, c1 = a1 + 2 * G::BLK_B;
          phase(I0{}, I0{}, I0{}, c0, c1); phase(I1{}, I1{}, I0{}, 0u, 0u);    // 2
          phase(I0{}, I2{}, IB{}, c0, c1); phase(I1{}, I3{}, I0{}, 0u, 0u);    // 3
        }
      }
    }
  };

  // channel of acc[i][j][e]: paired: wc*64 + 32 (j >> 1) + 8 lg + 4 (j & 1) + e;  plain: wc*64 + 16 j + 4 lg + e
  auto chan = [&](int j) { return PAIRED ? wc * 64 + 32 * (j >> 1) + 8 * lg + 4 * (j & 1) : wc * 64 + 16 * j + 4 * lg; };
#pragma unroll
  for (int j = 0; j < NI; ++j) bias_ad[j] = lds_base + G::BIAS_OFF + (uint32_t)(chan(j) * 4);
  if (tid < 2 * CH) reinterpret_cast<float*>(lds)[G::BIAS_OFF / 4 + tid] = tid < CH ? a.b1[tid] : a.b2[tid - CH];
  __syncthreads();
  const float slope = a.slope, inv_slope = 1.0f / a.slope;
  init_acc(0);

  issue_patch(0);
  stage_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  stage_one(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
#ifdef L2S_PAIR_STAMPS
  unsigned long long pr_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long pr_last = __builtin_amdgcn_s_memtime();
  const unsigned long long pr_t0 = pr_last;
#endif
  for (int c_i = 0; c_i < my_n; ++c_i) {
    int unit, g0;
    tile_origin(c_i, unit, g0);
    int len = a.lens ? a.lens[unit] * a.len_mul : T;
    len = len < T ? len : T;
    // ---- tile start (the wave rows are level here): the patch and the first two quarters are visible to every wave ----
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (upper) __builtin_amdgcn_s_barrier();               // the upper wave rows run one barrier behind from here on
    PRSTAMP(0)
    run_conv(I0{});
    PRSTAMP(1)

    // ---- conv1 done.  Level the rows (the lower one waits for the upper one's last phase), save the residual rows, then
    // t1 = mask(leaky_relu(conv1 + b1)) into the region ----
    if (!upper) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PRSTAMP(7)
    u32x4_t resp[MI][2];       // paired: 8 consecutive channels (one 16-byte chunk) per (row group, column half)
    u32x2_t resq[MI][NI];      // plain: 4 consecutive channels per (row group, block)
    (void)resp; (void)resq;
    if constexpr (PAIRED) {
      uint32_t ad[MI][2];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int R = wr * 64 + i * 16 + lm + h1;          // patch row of conv row p: the pair's input at the same time step
        const uint32_t ra = lds_base + (uint32_t)wc * G::BLK_B + (uint32_t)R * 128;
        ad[i][0] = ra + (uint32_t)(((0 + lg) ^ (R & 7)) << 4);
        ad[i][1] = ra + (uint32_t)(((4 + lg) ^ (R & 7)) << 4);
      }
      lds_read8_u4_sync(resp, ad);                         // one LDS round trip for the whole wave tile
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int R = wr * 64 + i * 16 + lm + h1;
        const uint32_t ra = lds_base + (uint32_t)wc * G::BLK_B + (uint32_t)R * 128;
        uint32_t ad[4];
#pragma unroll
        for (int j = 0; j < NI; ++j) ad[j] = ra + (uint32_t)(((2 * j + (lg >> 1)) ^ (R & 7)) << 4) + (uint32_t)((lg & 1) * 8);
        lds_read4_u2_sync(resq[i], ad);
      }
    }
    u32x2_t t1v[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int t = g0 + wr * 64 + i * 16 + lm;
      const uint32_t km = (unsigned)t < (unsigned)len ? 0xffffffffu : 0u;   // outside [0, len) the reference sees zero padding
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f32x4_t v = acc[i][j];                               // bias already inside
        const f32x4_t sc = v * slope;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], sc[e]);
        t1v[i][j].x = ET::pack2(v[0], v[1]) & km;
        t1v[i][j].y = ET::pack2(v[2], v[3]) & km;
      }
    }
    __builtin_amdgcn_s_barrier();                          // every wave is done reading the patch
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int R = wr * 64 + i * 16 + lm + XT1;
      const uint32_t ta = lds_base + (uint32_t)wc * G::BLK_B + (uint32_t)R * 128;
      if constexpr (PAIRED) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
          lds_write_u4(ta + (uint32_t)(((4 * h + lg) ^ (R & 7)) << 4),
                       u32x4_t{t1v[i][2 * h].x, t1v[i][2 * h].y, t1v[i][2 * h + 1].x, t1v[i][2 * h + 1].y});
      } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) lds_write_u2(ta + (uint32_t)(((2 * j + (lg >> 1)) ^ (R & 7)) << 4) + (uint32_t)((lg & 1) * 8), t1v[i][j]);
      }
    }
    init_acc(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // t1 is visible
    asm volatile("" ::: "memory");
    if (upper) __builtin_amdgcn_s_barrier();               // stagger again
    PRSTAMP(2)
    run_conv(I1{});
    PRSTAMP(3)

    // ---- conv2 done: level the rows; the region is free once every wave has finished reading t1 ----
    if (!upper) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PRSTAMP(5)
    const bool late_patch = (KIND == 1) && a.accumulate;   // that epilogue loads XS: a patch DMA in flight would be drained by it
    if (!late_patch && c_i + 1 < my_n) issue_patch(c_i + 1);
    PRSTAMP(6)

    auto rowmap = [&](int r) -> int64_t {
      const int t = g0 + r;
      return (r >= h2 && r < G::RM - h2 && t < T) ? (int64_t)unit * T + t : (int64_t)-1;
    };
    if constexpr (KIND == 0) {
      // x' = conv2 + b2 + x, x recovered from its LeakyReLU'd copy; leaky_relu(x') as 16-bit, whole lines from the MFMA layout
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32x4_t q = resp[i][h];
          const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float r = ET::to_f32((uint16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffff));
            acc[i][2 * h + (e >> 2)][e & 3] += fminf(r, r * inv_slope);      // inverse of leaky_relu (slope in (0, 1])
          }
        }
      l2s_gemm_desc p = {};
      p.C = a.Y; p.bias = nullptr; p.N = CH; p.ldc = CH; p.act = L2S_ACT_LRELU; p.act_slope = slope; p.alpha = 1.f;
      p.mask_T = T; p.mask_mul = 1;
      epilogue_direct16<ET, MI, NI, L2S_EPI_F16 + 3, decltype(rowmap), NoHook, true>(p, acc, lane, wr * 64, wc * 64, 0, rowmap,
                                                                                    unit * T, len);
    } else {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const float r[4] = {ET::to_f32((uint16_t)(resq[i][j].x & 0xffff)), ET::to_f32((uint16_t)(resq[i][j].x >> 16)),
                              ET::to_f32((uint16_t)(resq[i][j].y & 0xffff)), ET::to_f32((uint16_t)(resq[i][j].y >> 16))};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][e] += fminf(r[e], r[e] * inv_slope);
        }
      // fp32 sum of the stage's ResBlocks: a lane owns 4 consecutive fp32 channels of its rows (16-byte accesses); the previous
      // sums of TWO row groups are requested up front through unconditional (clamped) addresses (one wait per pair of groups)
      auto do_pair = [&](auto pr_tag) {
        constexpr int pr = decltype(pr_tag)::value;
        int64_t orow[2];
        f32x4_t old[2][NI];
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          orow[g2] = rowmap(wr * 64 + (2 * pr + g2) * 16 + lm);
          const int64_t os = orow[g2] < 0 ? (int64_t)unit * T : orow[g2];
          const float* xs = a.XS + os * CH + wc * 64 + lg * 4;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            old[g2][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (a.accumulate) { const float4 q = *reinterpret_cast<const float4*>(xs + j * 16); old[g2][j] = f32x4_t{q.x, q.y, q.z, q.w}; }
          }
        }
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          const int i = 2 * pr + g2;
          const int r = wr * 64 + i * 16 + lm;
          const int64_t o = orow[g2];
          const bool keep = g0 + r < len;
          float* xs = a.XS + (o < 0 ? 0 : o) * CH + wc * 64 + lg * 4;
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            f32x4_t v = acc[i][j];                             // b2 already inside
            if (!keep) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
            v = v + old[g2][j];
            if (o >= 0) {
              if (a.write_xs) *reinterpret_cast<float4*>(xs + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
              if (a.Y) {
                const f32x4_t sc = v * slope;
                uint2 q;
                q.x = ET::pack2(fmaxf(v[0], sc[0]), fmaxf(v[1], sc[1]));
                q.y = ET::pack2(fmaxf(v[2], sc[2]), fmaxf(v[3], sc[3]));
                *reinterpret_cast<uint2*>(a.Y + o * CH + wc * 64 + j * 16 + lg * 4) = q;
              }
            }
          }
        }
      };
      do_pair(std::integral_constant<int, 0>{});
      do_pair(std::integral_constant<int, 1>{});
    }
    init_acc(0);
    if (late_patch && c_i + 1 < my_n) issue_patch(c_i + 1);
    PRSTAMP(4)
  }
  wait_vmcnt<0>();                             // no LDS-DMA (the trailing dummies) may outlive the block
#ifdef L2S_PAIR_STAMPS
  if (lane == 0 && (wave == 0 || wave == 7) && g_pair_stamps) {
    unsigned long long* o = g_pair_stamps + ((int64_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 16;
    for (int i = 0; i < 8; ++i) o[i] = pr_acc[i];
    o[8] = (unsigned long long)my_n;
    o[9] = __builtin_amdgcn_s_memtime() - pr_t0;
  }
#endif
}

// ---- the LAST pairs of a stage's ResBlocks in one launch ---------------------------------------------------------------------------
// speech-resynthesis/models.py:103-109: xs = sum_j resblocks[j](x); each ResBlock ends with a (c1, c2, d = 5) pair (:34-41).  As one
// launch per last pair the fp32 stage sum is read and rewritten by every one of them (a third of a last pair's tile time is that
// epilogue: stamps above) and the MFMA pipe idles through three epilogues.  Here one block runs, per tile, the last pair of EVERY
// ResBlock back to back - patch_j -> conv1_j -> t1_j -> conv2_j - with conv2 accumulating straight into ONE running accumulator set
// (`sum`, initialised with b2_0 + b2_1 + ...; the residual x_j is added into it at the hand-over, so no residual rows are held across
// conv2) while conv1 uses a second set: 128 accumulator + 48 fragment registers.  The stage sum never exists in HBM: the only
// output is Y = leaky_relu(sum) in 16 bits (every wide stage feeds leaky_relu + ups next, models.py:109,101; a stage whose fp32
// sum is read - the last one - keeps the per-pair launches).  Tiles are cut for the largest k (S = RM - 2 h2max); the weight
// stream runs W1_0, W2_0, W1_1, ... across the j loop and the tiles; everything else is respair_phase_kernel's mid-pair path
// (paired W order, t1 by ds_write_b128, whole-line 16-bit epilogue).
template <typename ET, int CH>
__global__ __launch_bounds__(512) void respair_final_kernel(const l2s_rp::RpFinalArgs a) {
  using G = XGeo<CH>;
  constexpr int MI = 4, NI = 4;
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / G::NWC, wc = wave % G::NWC;
  const bool upper = wave >= 4;
  const int lm = lane & 15, lg = lane >> 4;
  const int srow = lane >> 3, schunk = (lane & 7) ^ (srow & 7);
  const int T = a.T, nj = a.nj, h2max = a.h2max;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);
  auto kof = [&](int j) { return j == 0 ? a.k[0] : (j == 1 ? a.k[1] : a.k[2]); };
  auto dof = [&](int j) { return j == 0 ? a.dil[0] : (j == 1 ? a.dil[1] : a.dil[2]); };
  auto xof = [&](int j) { return j == 0 ? a.X[0] : (j == 1 ? a.X[1] : a.X[2]); };
  auto w1of = [&](int j) { return j == 0 ? a.W1[0] : (j == 1 ? a.W1[1] : a.W1[2]); };
  auto w2of = [&](int j) { return j == 0 ? a.W2[0] : (j == 1 ? a.W2[1] : a.W2[2]); };

  int my_n = 0;
  if (!a.xcd_order) {
    my_n = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  } else {
    const int per = (a.ntiles + 7) >> 3, lo = ((int)blockIdx.x & 7) * per;
    int hi = lo + per;
    hi = hi < a.ntiles ? hi : a.ntiles;
    const int gxx = ((int)gridDim.x + 7) >> 3, b8 = (int)blockIdx.x >> 3;
    if (lo + b8 < hi) my_n = (hi - lo - b8 + gxx - 1) / gxx;
  }
  if (my_n <= 0) return;
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = (gridDim.x + 7) >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  auto tile_origin = [&](int i, int& unit, int& g0) {
    const int L = a.xcd_order ? xcd * per_xcd + bx + i * gx : (int)blockIdx.x + i * (int)gridDim.x;
    unit = L / a.tiles_per_clip;
    g0 = (L - unit * a.tiles_per_clip) * a.S - h2max;   // global time of conv row 0, the same for every ResBlock of the tile
  };
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t wring = lds_base + G::REGION;

  // patch of ResBlock j for tile i
  auto issue_patch = [&](int i, int j) {
    int unit, g0;
    tile_origin(i, unit, g0);
    const int h1 = ((kof(j) - 1) / 2) * dof(j);
    const int patch_rows = G::RM + 2 * h1;
    const uint16_t* X = xof(j);
#pragma unroll
    for (int q = 0; q < G::PPW; ++q) {
      const int instr = wave * G::PPW + q;
      const int cq = instr / (G::RPR / 8), blk = instr - cq * (G::RPR / 8);
      if (instr < G::PI && blk * 8 < patch_rows) {
        const int ts = g0 - h1 + blk * 8 + srow;
        const uint16_t* g = ((unsigned)ts < (unsigned)T) ? X + ((int64_t)unit * T + ts) * CH + cq * 64 + schunk * 8 : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + cq * (G::BLK_B / 2) + blk * 512), 16, 0, 0);
      }
    }
  };

  // ---- weight stream: W1_0, W2_0, W1_1, W2_1, ... cycling; the per-lane byte offset depends on k_j (row pitch k_j * CH) ----
  const bool stager = CH >= 128 || !(wave & 1);
  const int sw = CH >= 128 ? wave : wave >> 1;
  uint32_t w_row[G::DPW], w_chunk[G::DPW], w_lane[G::DPW];
#pragma unroll
  for (int q = 0; q < G::DPW; ++q) {
    const int qrow = sw * G::RPS + 8 * q + srow;
    const int within = qrow & 31;
    w_row[q] = (uint32_t)((qrow >> 5) * 64 + within);
    w_chunk[q] = (uint32_t)(((lane & 7) ^ paired_w_key(within)) * 16);
  }
  int s_j = 0, s_conv = 0, s_kt = 0, s_nkt = kof(0) * G::NBLK;
  uint32_t h_bytes = (uint32_t)(64 * kof(0) * CH);          // 32 weight rows further
  const char* kt_ptr = (const char*)a.W1[0];
  auto set_lane_offsets = [&]() {
    const uint32_t pitch = (uint32_t)(kof(s_j) * CH * 2);
#pragma unroll
    for (int q = 0; q < G::DPW; ++q) w_lane[q] = w_row[q] * pitch + w_chunk[q];
  };
  set_lane_offsets();
  auto stage_one = [&](auto dslot_tag, auto sh_tag) {
    constexpr int DSLOT = decltype(dslot_tag)::value, SH = decltype(sh_tag)::value;
    const char* wb = kt_ptr + (size_t)(SH ? h_bytes : 0u);
    uint16_t* dst = lds + G::REGION / 2 + DSLOT * (G::Q_B / 2) + sw * G::RPS * 64;
#pragma unroll
    for (int q = 0; q < G::DPW; ++q)
      if (stager) __builtin_amdgcn_global_load_lds((gptr_t)(wb + (size_t)w_lane[q]), (lptr_t)(dst + q * 512), 16, 0, 0);
    if constexpr (SH == 1) {
      kt_ptr += 128;
      if (++s_kt == s_nkt) {
        s_kt = 0;
        if (s_conv == 0) {
          s_conv = 1;
          kt_ptr = (const char*)w2of(s_j);
        } else {
          s_conv = 0;
          s_j = s_j + 1 == nj ? 0 : s_j + 1;
          s_nkt = kof(s_j) * G::NBLK;
          h_bytes = (uint32_t)(64 * kof(s_j) * CH);
          kt_ptr = (const char*)w1of(s_j);
          set_lane_offsets();
        }
      }
    }
  };

  // ---- fragments (paired W order) ----
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
  auto read_a = [&](auto off_tag, uint32_t a0, uint32_t a1) {
    constexpr int AO = decltype(off_tag)::value;
    lds_read_b128<AO>(fa[0][0], a0); lds_read_b128<AO + 2048>(fa[1][0], a0); lds_read_b128<AO + 4096>(fa[2][0], a0); lds_read_b128<AO + 6144>(fa[3][0], a0);
    lds_read_b128<AO>(fa[0][1], a1); lds_read_b128<AO + 2048>(fa[1][1], a1); lds_read_b128<AO + 4096>(fa[2][1], a1); lds_read_b128<AO + 6144>(fa[3][1], a1);
  };

  f32x4_t acc1[MI][NI], sum[MI][NI];
  // LDS stash behind the ring: b1_0 | b1_1 | b1_2 | b2_0 + b2_1 + b2_2, CH floats each
  uint32_t bias_ad[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) bias_ad[j] = lds_base + G::BIAS_OFF + (uint32_t)((wc * 64 + 32 * (j >> 1) + 8 * lg + 4 * (j & 1)) * 4);
  for (int i = tid; i < 4 * CH; i += 512) {
    const int j = i / CH, c = i - j * CH;
    float v = 0.f;
    if (j < 3) { if (j < nj) v = (j == 0 ? a.b1[0] : (j == 1 ? a.b1[1] : a.b1[2]))[c]; }
    else { v = a.b2[0][c]; if (nj > 1) v += a.b2[1][c]; if (nj > 2) v += a.b2[2][c]; }
    reinterpret_cast<float*>(lds)[G::BIAS_OFF / 4 + i] = v;
  }
  __syncthreads();
  auto init_acc = [&](f32x4_t (&accx)[MI][NI], const int which) {
    uint32_t ad[4];
#pragma unroll
    for (int j = 0; j < NI; ++j) ad[j] = bias_ad[j] + (uint32_t)(which * CH * 4);
    f32x4_t bj[NI];
    lds_read4_f4_sync(bj, ad);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) accx[i][j] = bj[j];
  };

  auto phase = [&](f32x4_t (&accx)[MI][NI], auto h_tag, auto slot_tag, auto aoff_tag, uint32_t a0, uint32_t a1) {
    constexpr int H = decltype(h_tag)::value, SLOT = decltype(slot_tag)::value;
    read_b(slot_tag);
    if (H == 0) { __builtin_amdgcn_sched_barrier(0); read_a(aoff_tag, a0, a1); }
    stage_one(std::integral_constant<int, (SLOT + 2) & (XRQ - 1)>{}, h_tag);
    __builtin_amdgcn_sched_barrier(0);
    wait_vmcnt<G::DPW>();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    lds_wait();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        accx[i][2 * H + s2] = ET::mfma(fb[s2][0], fa[i][0], accx[i][2 * H + s2]);
        accx[i][2 * H + s2] = ET::mfma(fb[s2][1], fa[i][1], accx[i][2 * H + s2]);
      }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  using IB = std::integral_constant<int, G::BLK_B>;
  int k = 0, dil = 0, h2 = 0;      // of the ResBlock being computed
  auto run_conv = [&](auto conv_tag, f32x4_t (&accx)[MI][NI]) {
    constexpr int conv = decltype(conv_tag)::value;
    auto row_addr = [&](int tap, uint32_t& a0, uint32_t& a1) {
      const int pr = wr * 64 + lm + (conv == 0 ? tap * dil : tap - h2 + XT1);
      const int x = pr & 7;
      const uint32_t pa = lds_base + (uint32_t)pr * 128;
      a0 = pa + (uint32_t)(((0 + lg) ^ x) << 4);
      a1 = pa + (uint32_t)(((4 + lg) ^ x) << 4);
    };
    if constexpr (CH == 64) {
      using SA = std::integral_constant<int, conv ? 2 : 0>; using SB = std::integral_constant<int, conv ? 3 : 1>;
      using SC = std::integral_constant<int, conv ? 0 : 2>; using SD = std::integral_constant<int, conv ? 1 : 3>;
      int tap = 0;
      for (; tap + 1 < k; tap += 2) {
        uint32_t a0, a1, c0, c1;
        row_addr(tap, a0, a1);
        row_addr(tap + 1, c0, c1);
        phase(accx, I0{}, SA{}, I0{}, a0, a1); phase(accx, I1{}, SB{}, I0{}, 0u, 0u);
        phase(accx, I0{}, SC{}, I0{}, c0, c1); phase(accx, I1{}, SD{}, I0{}, 0u, 0u);
      }
      uint32_t a0, a1;
      row_addr(tap, a0, a1);
      phase(accx, I0{}, SA{}, I0{}, a0, a1); phase(accx, I1{}, SB{}, I0{}, 0u, 0u);
    } else {
      for (int tap = 0; tap < k; ++tap) {
        uint32_t a0, a1;
        row_addr(tap, a0, a1);
        phase(accx, I0{}, I0{}, I0{}, a0, a1); phase(accx, I1{}, I1{}, I0{}, 0u, 0u);
        phase(accx, I0{}, I2{}, IB{}, a0, a1); phase(accx, I1{}, I3{}, I0{}, 0u, 0u);
        if constexpr (CH == 256) {
          const uint32_t c0 = a0 + 2 * G::BLK_B, c1 = a1 + 2 * G::BLK_B;
          phase(accx, I0{}, I0{}, I0{}, c0, c1); phase(accx, I1{}, I1{}, I0{}, 0u, 0u);
          phase(accx, I0{}, I2{}, IB{}, c0, c1); phase(accx, I1{}, I3{}, I0{}, 0u, 0u);
        }
      }
    }
  };

  const float slope = a.slope, inv_slope = 1.0f / a.slope;
  issue_patch(0, 0);
  stage_one(I0{}, I0{});
  stage_one(I1{}, I1{});
  for (int c_i = 0; c_i < my_n; ++c_i) {
    int unit, g0;
    tile_origin(c_i, unit, g0);
    int len = a.lens ? a.lens[unit] * a.len_mul : T;
    len = len < T ? len : T;
    for (int j = 0; j < nj; ++j) {
      k = kof(j); dil = dof(j); h2 = (k - 1) / 2;
      const int h1 = h2 * dil;
      // ---- ResBlock j of the tile: its patch and the next quarters are visible to every wave (the wave rows are level here) ----
      init_acc(acc1, j);
      if (j == 0) init_acc(sum, 3);
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (upper) __builtin_amdgcn_s_barrier();
      run_conv(I0{}, acc1);

      // ---- conv1 done: level the rows; sum += x_j (recovered from the LeakyReLU'd patch rows); t1 = mask(leaky_relu(conv1 + b1)) ----
      if (!upper) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      {
        u32x4_t resp[MI][2];
        uint32_t ad[MI][2];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int R = wr * 64 + i * 16 + lm + h1;
          const uint32_t ra = lds_base + (uint32_t)wc * G::BLK_B + (uint32_t)R * 128;
          ad[i][0] = ra + (uint32_t)(((0 + lg) ^ (R & 7)) << 4);
          ad[i][1] = ra + (uint32_t)(((4 + lg) ^ (R & 7)) << 4);
        }
        lds_read8_u4_sync(resp, ad);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const u32x4_t q = resp[i][h];
            const uint32_t w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float r = ET::to_f32((uint16_t)((w4[e >> 1] >> ((e & 1) * 16)) & 0xffff));
              sum[i][2 * h + (e >> 2)][e & 3] += fminf(r, r * inv_slope);
            }
          }
      }
      u32x2_t t1v[MI][NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int t = g0 + wr * 64 + i * 16 + lm;
        const uint32_t km = (unsigned)t < (unsigned)len ? 0xffffffffu : 0u;
#pragma unroll
        for (int jj = 0; jj < NI; ++jj) {
          f32x4_t v = acc1[i][jj];
          const f32x4_t sc = v * slope;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], sc[e]);
          t1v[i][jj].x = ET::pack2(v[0], v[1]) & km;
          t1v[i][jj].y = ET::pack2(v[2], v[3]) & km;
        }
      }
      __builtin_amdgcn_s_barrier();                        // every wave is done reading the patch
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int R = wr * 64 + i * 16 + lm + XT1;
        const uint32_t ta = lds_base + (uint32_t)wc * G::BLK_B + (uint32_t)R * 128;
#pragma unroll
        for (int h = 0; h < 2; ++h)
          lds_write_u4(ta + (uint32_t)(((4 * h + lg) ^ (R & 7)) << 4),
                       u32x4_t{t1v[i][2 * h].x, t1v[i][2 * h].y, t1v[i][2 * h + 1].x, t1v[i][2 * h + 1].y});
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // t1 is visible
      asm volatile("" ::: "memory");
      if (upper) __builtin_amdgcn_s_barrier();
      run_conv(I1{}, sum);

      // ---- conv2 done: level the rows; the region is free: the next ResBlock's (or the next tile's) patch travels now ----
      if (!upper) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (j + 1 < nj) issue_patch(c_i, j + 1);
      else if (c_i + 1 < my_n) issue_patch(c_i + 1, 0);
    }
    // ---- Y = leaky_relu(sum) for the rows whose taps stayed inside the tile for EVERY ResBlock, masked by the clip length ----
    auto rowmap = [&](int r) -> int64_t {
      const int t = g0 + r;
      return (r >= h2max && r < G::RM - h2max && t < T) ? (int64_t)unit * T + t : (int64_t)-1;
    };
    l2s_gemm_desc p = {};
    p.C = a.Y; p.bias = nullptr; p.N = CH; p.ldc = CH; p.act = L2S_ACT_LRELU; p.act_slope = slope; p.alpha = 1.f;
    p.mask_T = T; p.mask_mul = 1;
    epilogue_direct16<ET, MI, NI, L2S_EPI_F16 + 3, decltype(rowmap), NoHook, true>(p, sum, lane, wr * 64, wc * 64, 0, rowmap, unit * T, len);
  }
  wait_vmcnt<0>();
}

template <typename ET, int CH>
int launch_respair_final(const l2s_rp::RpFinalArgs& a, hipStream_t st) {
  using G = XGeo<CH>;
  constexpr int SMEM = G::BIAS_OFF + 4 * CH * 4;
  static_assert(SMEM <= 160 * 1024, "LDS");
  auto kern = respair_final_kernel<ET, CH>;
  static L2sSmemOptIn opt_in;
  if (int e = l2s_smem_opt_in(kern, SMEM, opt_in)) return e;
  const int need = (a.ntiles + 7) & ~7;
  const int grid = need < 256 ? need : 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), SMEM, st, a);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET, int CH, int KIND>
int launch_respair_phase(const RpArgs& a, hipStream_t st) {
  using G = XGeo<CH>;
  auto kern = respair_phase_kernel<ET, CH, KIND>;
  static L2sSmemOptIn opt_in;  // > 64 KB of dynamic LDS: opt-in per instantiation and device
  if (int e = l2s_smem_opt_in(kern, G::SMEM, opt_in)) return e;
  const int need = (a.ntiles + 7) & ~7;         // a multiple of 8: every XCD group has the same number of blocks
  const int grid = need < 256 ? need : 256;     // one resident block per CU
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), G::SMEM, st, a);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

}  // namespace

#ifdef L2S_PAIR_STAMPS
extern "C" int l2s_debug_pair_stamps(void* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_stamps), &buf, sizeof(buf)); }
#endif

// geometry of the phase-staggered kernel for l2s_respair (respair.hip): rows per tile and the limits on the halos
int l2s_respair_phase_rows(int C) { return C == 256 ? XGeo<256>::RM : (C == 128 ? XGeo<128>::RM : XGeo<64>::RM); }
bool l2s_respair_phase_supports(int C, int h1, int h2) {
  if (C != 256 && C != 128 && C != 64) return false;
  const int rm = l2s_respair_phase_rows(C), rpr = rm + 56;
  return rm + 2 * h1 <= rpr && h2 <= XT1 && XT1 + rm + h2 <= rpr && rm - 2 * h2 >= 16;
}

int l2s_respair_phase_launch(const RpArgs& a, int C, int dtype, int kind, hipStream_t st) {
  auto go = [&](auto et) -> int {
    using ET = decltype(et);
    if (C == 256) return kind ? launch_respair_phase<ET, 256, 1>(a, st) : launch_respair_phase<ET, 256, 0>(a, st);
    if (C == 128) return kind ? launch_respair_phase<ET, 128, 1>(a, st) : launch_respair_phase<ET, 128, 0>(a, st);
    return kind ? launch_respair_phase<ET, 64, 1>(a, st) : launch_respair_phase<ET, 64, 0>(a, st);
  };
  if (dtype == L2S_F16) return go(ElemF16{});
  if (dtype == L2S_BF16) return go(ElemBF16{});
  return L2S_EINVAL;
}

/*
 * include/lip2speech_hip.h: l2s_respair_final
 */
extern "C" int l2s_respair_final(const l2s_respair_final_desc* d, void* stream) {
  if (!d || !d->Y) return L2S_EINVAL;
  if (d->n < 1 || d->n > 3 || d->B <= 0 || d->T <= 0) return L2S_ESHAPE;
  if (d->C != 256 && d->C != 128 && d->C != 64) return L2S_EUNSUPPORTED;
  if (!(d->slope > 0.f && d->slope <= 1.f) || (d->lens && d->len_mul <= 0)) return L2S_EINVAL;
  if ((int64_t)d->B * d->T >= ((int64_t)1 << 31) / 2) return L2S_EUNSUPPORTED;
  l2s_rp::RpFinalArgs a = {};
  int h2max = 0;
  for (int j = 0; j < d->n; ++j) {
    if (!d->X[j] || !d->W1[j] || !d->W2[j] || !d->b1[j] || !d->b2[j]) return L2S_EINVAL;
    if (d->k[j] < 1 || !(d->k[j] & 1) || d->dil[j] < 1) return L2S_ESHAPE;
    const int h2 = (d->k[j] - 1) / 2, h1 = h2 * d->dil[j];
    if (!l2s_respair_phase_supports(d->C, h1, h2)) return L2S_EUNSUPPORTED;
    if (((uintptr_t)d->X[j] & 15) || ((uintptr_t)d->W1[j] & 15) || ((uintptr_t)d->W2[j] & 15) || ((uintptr_t)d->b1[j] & 15) ||
        ((uintptr_t)d->b2[j] & 15))
      return L2S_EALIGN;
    a.X[j] = (const uint16_t*)d->X[j]; a.W1[j] = (const uint16_t*)d->W1[j]; a.W2[j] = (const uint16_t*)d->W2[j];
    a.b1[j] = d->b1[j]; a.b2[j] = d->b2[j]; a.k[j] = d->k[j]; a.dil[j] = d->dil[j];
    h2max = h2 > h2max ? h2 : h2max;
  }
  for (int j = d->n; j < 3; ++j) { a.X[j] = a.X[0]; a.W1[j] = a.W1[0]; a.W2[j] = a.W2[0]; a.b1[j] = a.b1[0]; a.b2[j] = a.b2[0]; a.k[j] = a.k[0]; a.dil[j] = a.dil[0]; }
  if ((uintptr_t)d->Y & 15) return L2S_EALIGN;
  a.Y = (uint16_t*)d->Y; a.lens = d->lens; a.nj = d->n; a.len_mul = d->len_mul; a.T = d->T; a.h2max = h2max;
  a.S = l2s_respair_phase_rows(d->C) - 2 * h2max;
  a.tiles_per_clip = (d->T + a.S - 1) / a.S;
  a.ntiles = d->B * a.tiles_per_clip;
  a.xcd_order = 1;
  a.slope = d->slope;
  hipStream_t st = (hipStream_t)stream;
  auto go = [&](auto et) -> int {
    using ET = decltype(et);
    if (d->C == 256) return launch_respair_final<ET, 256>(a, st);
    if (d->C == 128) return launch_respair_final<ET, 128>(a, st);
    return launch_respair_final<ET, 64>(a, st);
  };
  if (d->dtype == L2S_F16) return go(ElemF16{});
  if (d->dtype == L2S_BF16) return go(ElemBF16{});
  return L2S_EINVAL;
}
