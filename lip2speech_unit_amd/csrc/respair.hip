// Fused convolution PAIR of a HiFi-GAN ResBlock1 for the wide vocoder stages (C = 64 at 16 000, C = 128 at 8 000 samples
// per 4-s clip):   x' = c2(leaky_relu(c1(leaky_relu(x)))) + x        speech-resynthesis/models.py:34-41, one (c1, c2, d) step.
//
// As two tap-GEMM launches a pair moves six activation arrays through HBM (c1: read lrelu(x), write t1; c2: read t1, read
// x, write x', write lrelu(x')) and the C = 64 / 128 second convolutions ran at 3-4.8 TB/s of algorithmic traffic: HBM-bound
// by construction.  Here one block owns a tile of 192 time steps of one clip and HBM sees TWO arrays:
//   * the input is the LeakyReLU'd copy only: patch rows (192 + 2*h1 halo, h1 = (k-1)/2*dil <= 25) arrive by LDS-DMA; the raw
//     x of the residual is recovered from it by the inverse of leaky_relu (x = xl >= 0 ? xl : xl / slope), as resblock.hip does;
//   * conv1 runs on MFMA out of the patch (tap = row shift), its bias + LeakyReLU + length mask are applied in the MFMA
//     register layout and t1 goes back into the SAME LDS region as 16-bit rows (the patch is dead by then; the residual
//     rows were saved to registers first), never to HBM;
//   * conv2 runs out of t1; the 192 - 2*h2 rows whose taps stayed inside the tile are the tile's output (h2 = (k-1)/2):
//     mid pairs store leaky_relu(x') as 16-bit (the next pair's input), the last pair of a ResBlock adds x' into the fp32
//     sum of the stage's ResBlocks (models.py:103-108) and, for the last ResBlock, also stores leaky_relu of that sum.
// Schedule: the tap loop of patchconv.hip (weights streamed through a 4-slot LDS ring by LDS-DMA across phases and tiles,
// counted vmcnt, one raw barrier per (tap, K half) element, fragment reads one k-step ahead), 48 x 64 wave tiles
// (MI = 3, NI = 4), 4 waves and two blocks per CU at C = 64 (80 KB each), 8 waves and one block at C = 128 (160 KB).  The next
// tile's patch DMA is issued as soon as conv2 has finished reading t1 and travels under the epilogue, which touches no
// global memory besides its stores (the mid-pair epilogue: residual from registers, the clip length is tile-uniform).
#include "tapgemm_common.h"
#include "respair_args.h"
#include <cstdlib>

using namespace l2s;
using l2s_rp::RpArgs;

namespace {

constexpr int RM = 192;       // rows both convolutions compute per tile
constexpr int RPR = 256;      // rows of LDS region A: the patch (RM + 2*h1 <= 242 rows), later t1 (rows 8 .. 8+RM)
constexpr int T1_ROW0 = 8;    // t1 row p lives at region row p + 8: conv2's taps reach rows p - h2 >= -5
constexpr int RQ = 4;         // weight ring slots
constexpr int rp_smem(int ch) { return (ch / 64) * RPR * 128 + RQ * ch * 128 + (ch / 16) * 4096; }   // 80 KB / 160 KB


template <int OFF>
__device__ __forceinline__ u32x2_t lds_read_b64(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// KIND 0: mid pair (Y = leaky_relu(x'));  KIND 1: last pair of a ResBlock (XS (+)= x', optional Y = leaky_relu(XS))
template <typename ET, int CH, int KIND>
__global__ __launch_bounds__(CH * 4, (CH == 64 ? 2 : 1)) void respair_kernel(const RpArgs a) {
  constexpr int MI = 3, NI = 4;
  constexpr int HALVES = CH / 64, NWC = CH / 64, NW = 4 * NWC;
  constexpr int HALF_B = RPR * 128;               // one 64-channel half of region A
  constexpr int REGION_A = HALVES * HALF_B;
  constexpr int QEL_B = CH * 128;                 // one weight stream element: CH rows x 64 K values
  constexpr int P_PER_W = HALVES * (RPR / 8) / NW;   // patch DMA instructions per wave (8)
  extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lm = lane & 15, lg = lane >> 4;
  const int srow = lane >> 3, schunk = (lane & 7) ^ (srow & 7);
  const int wm = wave / NWC, wc = wave - wm * NWC;
  const int k = a.k, dil = a.dil, h1 = a.h1, h2 = a.h2, T = a.T;
  const int Ktot = k * CH;
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_zero16);
  // tiles of this block: its XCD's range [xcd*per, min((xcd+1)*per, ntiles)) walked with stride gridDim/8 from offset bx
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
  const int nel = k * HALVES;               // stream elements per convolution: (tap, K half)
  const int total = my_n * 2 * nel;         // weight stream across both convolutions of all of this block's tiles

  const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
  const uint32_t wring = lds_base + REGION_A;
  const uint32_t scr = wring + RQ * QEL_B + (uint32_t)wave * 4096;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so each XCD walks ONE contiguous range of the tile
  // list and the blocks of an XCD work on neighbouring tiles at the same time - the 2*(h1+h2) halo rows two neighbours
  // share are then fetched from HBM once instead of once per XCD (speed only: any order is correct)
  const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, gx = (gridDim.x + 7) >> 3;
  const int per_xcd = (a.ntiles + 7) >> 3;
  auto tile_index = [&](int i) { return a.xcd_order ? xcd * per_xcd + bx + i * gx : (int)blockIdx.x + i * (int)gridDim.x; };
  auto tile_origin = [&](int i, int& unit, int& g0) {
    const int L = tile_index(i);
    unit = L / a.tiles_per_clip;
    g0 = (L - unit * a.tiles_per_clip) * a.S - h2;      // global time of conv row 0 (t1 row 0 / output row 0)
  };
  auto issue_patch = [&](int i) {
    int unit, g0;
    tile_origin(i, unit, g0);
#pragma unroll
    for (int j = 0; j < P_PER_W; ++j) {
      const int instr = wave * P_PER_W + j;            // [0, 32 * HALVES): half instr / 32, 8-row block instr % 32
      const int hp = instr / (RPR / 8), blk = instr - hp * (RPR / 8);
      const int ts = g0 - h1 + blk * 8 + srow;
      const uint16_t* g = ((unsigned)ts < (unsigned)T) ? a.X + ((int64_t)unit * T + ts) * CH + hp * 64 + schunk * 8 : zero;
      __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)(lds + hp * (HALF_B / 2) + blk * 512), 16, 0, 0);
    }
  };
  // weight element: CH rows x 128 B = CH / 8 DMA instructions, two per wave; the stream alternates W1 / W2 per phase
  const int64_t w_lane = (int64_t)(wave * 16 + srow) * Ktot + schunk * 8;
  int s_e = 0, s_slot = 0, issued = 0;
  auto issue_next_w = [&]() {
    const uint16_t* w = (s_e < nel ? a.W1 : a.W2) + w_lane + (s_e < nel ? s_e : s_e - nel) * 64;
    uint16_t* dst = lds + REGION_A / 2 + s_slot * (QEL_B / 2) + wave * 1024;
    __builtin_amdgcn_global_load_lds((gptr_t)w, (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(w + (int64_t)8 * Ktot), (lptr_t)(dst + 512), 16, 0, 0);
    ++issued;
    s_slot = s_slot == RQ - 1 ? 0 : s_slot + 1;
    s_e = s_e + 1 == 2 * nel ? 0 : s_e + 1;
  };

  f32x4_t acc[MI][NI];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  const uint32_t wk0 = (uint32_t)(lm * 8 + ((0 + lg) ^ (lm & 7))) * 16;
  const uint32_t wk1 = (uint32_t)(lm * 8 + ((4 + lg) ^ (lm & 7))) * 16;

  frag16 fa0[MI], fw0[NI], fa1[MI], fw1[NI];
  uint32_t a1_next = 0;
  // rowshift: region-A row of this wave's first conv row for the element's tap
  auto read_k0 = [&](int rowshift, int hp, int slot) {
    const int pr = wm * 48 + lm + rowshift;
    const int x = pr & 7;
    const uint32_t pa = lds_base + (uint32_t)hp * HALF_B + (uint32_t)pr * 128;
    const uint32_t a0 = pa + (uint32_t)(((0 + lg) ^ x) << 4);
    a1_next = pa + (uint32_t)(((4 + lg) ^ x) << 4);
    const uint32_t wb = wring + (uint32_t)slot * QEL_B + (uint32_t)wc * 8192 + wk0;
    lds_read_b128<0>(fa0[0], a0); lds_read_b128<2048>(fa0[1], a0); lds_read_b128<4096>(fa0[2], a0);
    lds_read_b128<0>(fw0[0], wb); lds_read_b128<2048>(fw0[1], wb); lds_read_b128<4096>(fw0[2], wb); lds_read_b128<6144>(fw0[3], wb);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_k1 = [&](int slot) {
    const uint32_t wb = wring + (uint32_t)slot * QEL_B + (uint32_t)wc * 8192 + wk1;
    lds_read_b128<0>(fa1[0], a1_next); lds_read_b128<2048>(fa1[1], a1_next); lds_read_b128<4096>(fa1[2], a1_next);
    lds_read_b128<0>(fw1[0], wb); lds_read_b128<2048>(fw1[1], wb); lds_read_b128<4096>(fw1[2], wb); lds_read_b128<6144>(fw1[3], wb);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mfma_k = [&](frag16(&fw)[NI], frag16(&fa)[MI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = ET::mfma(fw[j], fa[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
  };

  int g = 0;      // weight-stream element being consumed
  int slot = 0;
  // one convolution: nel elements out of region A (phase 0: the patch, row shift tap*dil; phase 1: t1, shift tap - h2 + 8)
  auto run_phase = [&](const int phase) {
    auto shift = [&](int el) -> int {
      const int tap = el / HALVES;
      return phase == 0 ? tap * dil : tap - h2 + T1_ROW0;
    };
    read_k0(shift(0), 0, slot);
    read_k1(slot);
    for (int t = 0; t < nel; ++t) {
      const bool more = t + 1 < nel;
      lds_wait_n<7>();                         // k0(t) landed, k1(t) may still be in flight
      mfma_k(fw0, fa0);
      const int nslot = slot == RQ - 1 ? 0 : slot + 1;
      if (more) {
        // publish element t+1's weights: in flight behind them is only element g+2 (two DMAs)
        if (issued - g - 2 > 0) wait_vmcnt<2>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < total) issue_next_w();    // element g+3 -> the slot of element g-1, read by nobody any more
        const int e1 = t + 1;
        read_k0(shift(e1), e1 - (e1 / HALVES) * HALVES, nslot);
        lds_wait_n<7>();                       // k1(t)
      } else {
        lds_wait_n<0>();
      }
      mfma_k(fw1, fa1);
      if (more) read_k1(nslot);
      slot = nslot;
      ++g;
    }
  };

  // bias fragments in the MFMA layout: a lane owns channels wc*64 + j*16 + lg*4 .. +3 of its rows
  f32x4_t b1j[NI], b2j[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const float4 q1 = *reinterpret_cast<const float4*>(a.b1 + wc * 64 + j * 16 + lg * 4);
    const float4 q2 = *reinterpret_cast<const float4*>(a.b2 + wc * 64 + j * 16 + lg * 4);
    b1j[j] = f32x4_t{q1.x, q1.y, q1.z, q1.w};
    b2j[j] = f32x4_t{q2.x, q2.y, q2.z, q2.w};
  }
  const float slope = a.slope, inv_slope = 1.0f / a.slope;

  issue_patch(0);
  issue_next_w();
  if (total > 1) issue_next_w();
  for (int c_i = 0; c_i < my_n; ++c_i) {
    int unit, g0;
    tile_origin(c_i, unit, g0);
    int len = a.lens ? a.lens[unit] * a.len_mul : T;
    len = len < T ? len : T;
    // ---- tile start: the patch and the first element's weights are visible to every wave ----
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < total) issue_next_w();
    run_phase(0);

    // ---- conv1 done: residual rows out of the patch, t1 = mask(leaky_relu(conv1 + b1)) into the same region ----
    u32x2_t res[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int R = wm * 48 + i * 16 + lm + h1;          // patch row of conv row p: the pair's input at the same time step
      const uint32_t ra = lds_base + (uint32_t)wc * HALF_B + (uint32_t)R * 128 + (uint32_t)((lg & 1) * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) res[i][j] = lds_read_b64<0>(ra + (uint32_t)(((2 * j + (lg >> 1)) ^ (R & 7)) << 4));
    }
    u32x2_t t1v[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int t = g0 + wm * 48 + i * 16 + lm;
      const bool keep = (unsigned)t < (unsigned)len;      // outside [0, len) the reference sees zero padding
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f32x4_t v = acc[i][j] + b1j[j];
        const f32x4_t sc = v * slope;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep ? fmaxf(v[e], sc[e]) : 0.f;
        t1v[i][j].x = ET::pack2(v[0], v[1]);
        t1v[i][j].y = ET::pack2(v[2], v[3]);
      }
    }
    lds_wait();                                            // the residual reads have landed
    __builtin_amdgcn_s_barrier();                          // every wave is done reading the patch
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int R = wm * 48 + i * 16 + lm + T1_ROW0;
      const uint32_t ta = lds_base + (uint32_t)wc * HALF_B + (uint32_t)R * 128 + (uint32_t)((lg & 1) * 8);
#pragma unroll
      for (int j = 0; j < NI; ++j) lds_write_b64<0>(ta + (uint32_t)(((2 * j + (lg >> 1)) ^ (R & 7)) << 4), t1v[i][j]);
    }
    zero_acc();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // publish t1 and conv2's first weight element (behind it in flight: at most one more element)
    if (issued - g - 1 > 1) wait_vmcnt<4>(); else if (issued - g - 1 > 0) wait_vmcnt<2>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < total) issue_next_w();
    run_phase(1);

    // ---- conv2 done: region A is free once every wave has finished reading t1 ----
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool late_patch = (KIND == 1) && a.accumulate;   // that epilogue loads XS: a patch DMA in flight would be drained by it
    if (!late_patch && c_i + 1 < my_n) issue_patch(c_i + 1);

    // x' = conv2 + b2 + x, x recovered from its LeakyReLU'd copy
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        float r[4] = {ET::to_f32((uint16_t)(res[i][j].x & 0xffff)), ET::to_f32((uint16_t)(res[i][j].x >> 16)),
                      ET::to_f32((uint16_t)(res[i][j].y & 0xffff)), ET::to_f32((uint16_t)(res[i][j].y >> 16))};
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] += r[e] < 0.f ? r[e] * inv_slope : r[e];
      }
    auto rowmap = [&](int r) -> int64_t {
      const int t = g0 + r;
      return (r >= h2 && r < RM - h2 && t < T) ? (int64_t)unit * T + t : (int64_t)-1;
    };
    if constexpr (KIND == 0) {
      // leaky_relu(x') as 16-bit through the lean tap-GEMM epilogue (bias, LeakyReLU, uniform length mask, transposition)
      l2s_gemm_desc p = {};
      p.C = a.Y; p.bias = a.b2; p.N = CH; p.ldc = CH; p.act = L2S_ACT_LRELU; p.act_slope = slope; p.alpha = 1.f;
      p.mask_T = T; p.mask_mul = 1;
      epilogue_fast16<ET, MI, NI, 1, true, 1>(p, acc, scr, lane, wm * 48, wc * 64, 0, rowmap, unit * T, len);
    } else {
      // fp32 sum of the stage's ResBlocks: a lane owns 4 consecutive fp32 channels of its rows (16-byte accesses).  The previous
      // sums of ALL row groups are requested first, through unconditional (clamped) addresses: with the load inside the per-row
      // branch hipcc waited vmcnt(0) behind every row group - one dependent memory round trip per 16 rows.
      int64_t orow[MI];
      f32x4_t old[MI][NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        orow[i] = rowmap(wm * 48 + i * 16 + lm);
        const int64_t os = orow[i] < 0 ? (int64_t)unit * T : orow[i];
        const float* xs = a.XS + os * CH + wc * 64 + lg * 4;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          old[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (a.accumulate) { const float4 q = *reinterpret_cast<const float4*>(xs + j * 16); old[i][j] = f32x4_t{q.x, q.y, q.z, q.w}; }
        }
      }
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int r = wm * 48 + i * 16 + lm;
        const int64_t o = orow[i];
        const bool keep = g0 + r < len;
        float* xs = a.XS + (o < 0 ? 0 : o) * CH + wc * 64 + lg * 4;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          f32x4_t v = acc[i][j] + b2j[j];
          if (!keep) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
          v = v + old[i][j];
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
    }
    zero_acc();
    if (late_patch && c_i + 1 < my_n) issue_patch(c_i + 1);
  }
  wait_vmcnt<0>();                             // no LDS-DMA may outlive the block
}

template <typename ET, int CH, int KIND>
int launch_respair(const RpArgs& a, hipStream_t st) {
  constexpr int SMEM = rp_smem(CH);
  auto kern = respair_kernel<ET, CH, KIND>;
  static L2sSmemOptIn opt_in;  // > 64 KB of dynamic LDS: opt-in per instantiation and device
  if (int e = l2s_smem_opt_in(kern, SMEM, opt_in)) return e;
  constexpr int slots = CH == 64 ? 512 : 256;   // resident blocks: two per CU at 64 channels, one at 128
  const int need = (a.ntiles + 7) & ~7;         // a multiple of 8: every XCD group has the same number of blocks
  const int grid = need < slots ? need : slots;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(CH * 4), SMEM, st, a);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

}  // namespace

// respair_phase.hip: the C = 256 and C = 128 stages on the phase-staggered schedule
int l2s_respair_phase_rows(int C);
bool l2s_respair_phase_supports(int C, int h1, int h2);
int l2s_respair_phase_launch(const RpArgs& a, int C, int dtype, int kind, hipStream_t st);

extern "C" int l2s_respair(const l2s_respair_desc* d, void* stream) {
  if (!d || !d->X || !d->W1 || !d->W2 || !d->b1 || !d->b2) return L2S_EINVAL;
  if (d->last < 0 || d->last > 2 || (d->last ? !d->XS : !d->Y)) return L2S_EINVAL;
  if (d->last == 2 && !d->Y) return L2S_EINVAL;     // XS read, not written: Y is the only output
  if (d->B <= 0 || d->T <= 0 || d->k < 1 || !(d->k & 1) || d->dil < 1) return L2S_ESHAPE;
  if (d->C != 64 && d->C != 128 && d->C != 256) return L2S_EUNSUPPORTED;
  const int h2 = (d->k - 1) / 2, h1 = h2 * d->dil;
  // C = 256 always, C = 128 / 64 by default run on the phase-staggered kernel of respair_phase.hip; L2S_RESPAIR_PHASE is a bit
  // mask (1: C = 128, 2: C = 64; 0 = this file's 192-row kernels, the A/B reference)
  static const int phase128 = [] { const char* e = getenv("L2S_RESPAIR_PHASE"); return e ? atoi(e) : 3; }();
  const bool c256 = d->C == 256 || ((d->C == 128 || d->C == 64) && (phase128 & (d->C == 128 ? 1 : 2)) &&
                                    l2s_respair_phase_supports(d->C, h1, h2));
  const int rm = c256 ? l2s_respair_phase_rows(d->C) : RM;
  if (c256 ? !l2s_respair_phase_supports(d->C, h1, h2) : (h1 > (RPR - RM) / 2 || h2 > T1_ROW0 || RM - 2 * h2 < 16)) return L2S_EUNSUPPORTED;
  if (!(d->slope > 0.f && d->slope <= 1.f) || (d->lens && d->len_mul <= 0)) return L2S_EINVAL;
  if (((uintptr_t)d->X & 15) || ((uintptr_t)d->W1 & 15) || ((uintptr_t)d->W2 & 15) || ((uintptr_t)d->Y & 15) ||
      ((uintptr_t)d->XS & 15) || ((uintptr_t)d->b1 & 15) || ((uintptr_t)d->b2 & 15))
    return L2S_EALIGN;
  if ((int64_t)d->B * d->T >= ((int64_t)1 << 31) / 2) return L2S_EUNSUPPORTED;   // 32-bit row math in the epilogue
  RpArgs a;
  a.X = (const uint16_t*)d->X; a.W1 = (const uint16_t*)d->W1; a.W2 = (const uint16_t*)d->W2;
  a.b1 = d->b1; a.b2 = d->b2; a.Y = (uint16_t*)d->Y; a.XS = d->XS; a.lens = d->lens;
  a.len_mul = d->len_mul; a.T = d->T; a.k = d->k; a.dil = d->dil; a.h1 = h1; a.h2 = h2;
  a.S = rm - 2 * h2;
  a.tiles_per_clip = (d->T + a.S - 1) / a.S;
  a.ntiles = d->B * a.tiles_per_clip;
  a.accumulate = d->accumulate ? 1 : 0;
  a.write_xs = d->last != 2;
  static const int xcd_on = [] { const char* e = getenv("L2S_RESPAIR_XCD"); return e ? atoi(e) : 1; }();   // A/B switch
  a.xcd_order = xcd_on;
  a.slope = d->slope;
  hipStream_t st = (hipStream_t)stream;
  if (c256) return l2s_respair_phase_launch(a, d->C, d->dtype, d->last ? 1 : 0, st);
  auto go = [&](auto et) -> int {
    using ET = decltype(et);
    if (d->C == 64) return d->last ? launch_respair<ET, 64, 1>(a, st) : launch_respair<ET, 64, 0>(a, st);
    return d->last ? launch_respair<ET, 128, 1>(a, st) : launch_respair<ET, 128, 0>(a, st);
  };
  if (d->dtype == L2S_F16) return go(ElemF16{});
  if (d->dtype == L2S_BF16) return go(ElemBF16{});
  return L2S_EINVAL;
}
