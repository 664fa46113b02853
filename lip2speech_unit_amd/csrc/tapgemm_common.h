// Pieces shared by the tap-GEMM kernels (tapgemm_kernel.h, patchconv.hip): LDS asm accessors and the fused epilogue.
#pragma once
#include "l2s_common.h"
#include "tapgemm_tiles.h"
#include <type_traits>

namespace l2s {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// source of every zero-filled 16-byte chunk (conv padding, M/N/K tails): LDS-DMA cannot write an immediate
__device__ const uint4 g_zero16 = {0u, 0u, 0u, 0u};

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// Fragment reads are inline asm on purpose: for a compiler-visible LDS load hipcc (ROCm 7.2) inserts s_waitcnt vmcnt(0)
// while any LDS-DMA is outstanding, which would drain the tiles we keep in flight.  The reads are ordered against the
// DMA by the counted vmcnt + barrier of the main loop and against the MFMAs by lds_wait() below.
template <int OFF>
__device__ __forceinline__ void lds_read_b128(frag16& f, uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  f.u = make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void lds_wait() {
  // MFMAs are register-only, so the scheduler is free to move them across the asm: pin both sides.  Without the first
  // barrier hipcc sinks most of the preceding MFMA batch below the wait and the wave stalls on LDS with an empty pipe.
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// counted variant: returns when at most N of this wave's LDS operations (issued in order) are still outstanding
template <int N>
__device__ __forceinline__ void lds_wait_n() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int OFF>
__device__ __forceinline__ void lds_write_f4(uint32_t addr, f32x4_t v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x4_t lds_read_f4(uint32_t addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_write_b64(uint32_t addr, u32x2_t v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
// offset given as a value: must fold to a constant (fully unrolled loop indices do)
__device__ __forceinline__ void lds_write_b64_at(uint32_t addr, u32x2_t v, const int off) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(off) : "memory");
}
__device__ __forceinline__ u32x4_t lds_read_u4(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}


// Fused epilogue of one wave's (MI*16) x (NI*16) accumulator sub-tile.
//   scr       byte address of this wave's private LDS scratch (16 rows x (NI*16+4) floats)
//   row_base  local row of the sub-tile's first row; rowmap(local_row) -> output row, or -1 to skip
//   ncol_base first output channel of the sub-tile
// Bytes of wave-private LDS scratch the epilogue of an (MI*16) x (NI*16) sub-tile needs (fp32 and 16-bit paths).
template <int MI, int NI>
__host__ __device__ constexpr int epilogue_scratch_bytes() {
  constexpr int a = 16 * (NI * 16 + 4) * 4, b = (MI >= 2 ? 32 : 16) * (NI * 32 + 16);
  return a > b ? a : b;
}

// flag sets of the epilogue families of tapgemm_tiles.h::pick_epilogue
constexpr int F_G16A = L2S_F_RES_PRE | L2S_F_RES_POST;
constexpr int F_G16B = L2S_F_RES_PRE | L2S_F_RES_POST | L2S_F_DUAL | L2S_F_MASK;
constexpr int F_S32 = L2S_F_RES_PRE | L2S_F_RES_POST | L2S_F_OUT_F32 | L2S_F_RES_F32;

// FEAT: the flag bits this instantiation supports (all others are known to be clear, their code folds away); the
// dispatcher below picks the leanest instantiation once per tile.
// LIN: only the "linear family" of activations {none, relu, prelu, lrelu} can occur; they share one branch-free form
// y = max(x, 0) + min(x, 0) * s  (s = 1, 0, per-channel slope, scalar slope), identical in value to the branchy forms.
// SWZ: the scratch is 16 rows x 256 B without padding (4 KB per wave, NI = 4 only): 16-byte chunk c of row r lives at
// chunk c ^ r, which keeps both the MFMA-layout writes and the row-layout reads conflict-free (phasegemm_kernel.h, whose
// 128 KB quarter ring leaves exactly 32 KB for the eight waves).
template <typename ET, int MI, int NI, int FEAT, bool LIN, bool SWZ = false, typename RowMap>
__device__ __forceinline__ void epilogue_impl(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const uint32_t scr,
                                              const int lane, const int row_base, const int ncol_base, const int grp,
                                              RowMap rowmap) {
  // ---- epilogue -------------------------------------------------------------------------------------------------
  // The MFMA leaves each lane with 4 consecutive channels of 16 different rows: storing that directly costs one
  // partial cache line per lane (store-issue bound, ~0.5 us per 16x16 sub-tile).  Instead each wave transposes 16 rows
  // at a time through a private LDS scratch (in the ring stage that was just consumed; the other stages keep
  // receiving the next tile's DMA) so that a lane owns 8 consecutive channels of one row: bias / residual / accumulate
  // loads and the stores become 16-byte accesses and a wave-instruction covers whole 128-byte row segments.
  constexpr int WAVE_N = NI * 16;
  constexpr int LPR = WAVE_N / 8;               // lanes per row after the transpose
  constexpr int SROW = SWZ ? WAVE_N : WAVE_N + 4;   // scratch row stride in floats
  static_assert(!SWZ || NI == 4, "swizzled scratch: 16 chunks of 16 B per row");
  const int lm = lane & 15, lg = lane >> 4;
  const uint32_t scr_w = scr + (uint32_t)(lm * SROW + lg * 4) * 4;
  const int rr = lane / LPR, cc = lane - rr * LPR;
  const uint32_t scr_r = scr + (uint32_t)(rr * SROW + cc * 8) * 4;
  constexpr int RPP_ = 64 / LPR, PASSES_ = RPP_ < 16 ? 16 / RPP_ : 1;
  uint32_t swz_w[NI], swz_lo[PASSES_], swz_hi[PASSES_];   // SWZ only
  (void)swz_w; (void)swz_lo; (void)swz_hi; (void)scr_w; (void)scr_r;
  if constexpr (SWZ) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
      swz_w[j] = scr + (uint32_t)(lm * 256 + (((j ^ (lm >> 2)) << 6) | ((lg ^ (lm & 3)) << 4)));
#pragma unroll
    for (int h = 0; h < PASSES_; ++h) {
      const int r = rr + h * RPP_;
      swz_lo[h] = scr + (uint32_t)(r * 256 + (((2 * cc) ^ r) << 4));
      swz_hi[h] = scr + (uint32_t)(r * 256 + (((2 * cc + 1) ^ r) << 4));
    }
  }
  const bool lane_on = lane < 16 * LPR;
  const int flags = p.flags & FEAT;
  const int n = ncol_base + cc * 8;
  const bool ok_lo = lane_on && (n < p.N), ok_hi = lane_on && (n + 4 < p.N);
  const int col = grp * p.c_gstride + n;
  const float* bias = p.bias ? p.bias + grp * p.N : nullptr;
  const float* slope = (p.act == L2S_ACT_PRELU) ? p.slope + grp * p.N : nullptr;
  float bv[8], sv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bv[e] = 0.f; sv[e] = 0.f; }
  if (bias && ok_lo) { const float4 q = *reinterpret_cast<const float4*>(bias + n); bv[0] = q.x; bv[1] = q.y; bv[2] = q.z; bv[3] = q.w; }
  if (bias && ok_hi) { const float4 q = *reinterpret_cast<const float4*>(bias + n + 4); bv[4] = q.x; bv[5] = q.y; bv[6] = q.z; bv[7] = q.w; }
  if (slope && ok_lo) { const float4 q = *reinterpret_cast<const float4*>(slope + n); sv[0] = q.x; sv[1] = q.y; sv[2] = q.z; sv[3] = q.w; }
  if (slope && ok_hi) { const float4 q = *reinterpret_cast<const float4*>(slope + n + 4); sv[4] = q.x; sv[5] = q.y; sv[6] = q.z; sv[7] = q.w; }
  float sl[8];   // linear-family slope of the negative half
  {
    const float s_uni = p.act == L2S_ACT_RELU ? 0.f : (p.act == L2S_ACT_LRELU ? p.act_slope : 1.f);
#pragma unroll
    for (int e = 0; e < 8; ++e) sl[e] = p.act == L2S_ACT_PRELU ? sv[e] : s_uni;
  }
  const bool has_res = (flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) != 0;
  auto ld8 = [&](const void* base, int64_t row, int ld, bool f32, float(&out)[8]) {  // 8 values at (row, col)
    if (f32) {
      const float* q = (const float*)base + row * ld + col;
      if (ok_lo) { const float4 t = *reinterpret_cast<const float4*>(q); out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w; }
      if (ok_hi) { const float4 t = *reinterpret_cast<const float4*>(q + 4); out[4] = t.x; out[5] = t.y; out[6] = t.z; out[7] = t.w; }
    } else {
      const uint16_t* q = (const uint16_t*)base + row * ld + col;
      if (ok_lo) {
        const uint2 t = *reinterpret_cast<const uint2*>(q);
        out[0] = ET::to_f32((uint16_t)(t.x & 0xffff)); out[1] = ET::to_f32((uint16_t)(t.x >> 16));
        out[2] = ET::to_f32((uint16_t)(t.y & 0xffff)); out[3] = ET::to_f32((uint16_t)(t.y >> 16));
      }
      if (ok_hi) {
        const uint2 t = *reinterpret_cast<const uint2*>(q + 4);
        out[4] = ET::to_f32((uint16_t)(t.x & 0xffff)); out[5] = ET::to_f32((uint16_t)(t.x >> 16));
        out[6] = ET::to_f32((uint16_t)(t.y & 0xffff)); out[7] = ET::to_f32((uint16_t)(t.y >> 16));
      }
    }
  };
  // whole 8-channel groups at 16-byte aligned columns: the stores need no per-lane alignment / tail branches
  const bool st16_uniform = ((p.N & 7) == 0) && ((p.c_gstride & 7) == 0 || grp == 0) && ((ncol_base & 7) == 0);
  auto st8_16 = [&](void* base, int64_t row, int ld, const float(&v)[8]) {  // 8 values stored as 16-bit
    uint16_t* q = (uint16_t*)base + row * ld + col;
    uint4 t;
    t.x = ET::pack2(v[0], v[1]);
    t.y = ET::pack2(v[2], v[3]);
    t.z = ET::pack2(v[4], v[5]);
    t.w = ET::pack2(v[6], v[7]);
    if (st16_uniform && (((uintptr_t)base & 15) == 0) && ((ld & 7) == 0)) {   // wave-uniform: every row is 16-byte aligned
      *reinterpret_cast<uint4*>(q) = t;
    } else if (ok_hi && (((uintptr_t)q & 15) == 0)) {
      *reinterpret_cast<uint4*>(q) = t;
    } else {
      if (ok_lo) *reinterpret_cast<uint2*>(q) = make_uint2(t.x, t.y);
      if (ok_hi) *reinterpret_cast<uint2*>(q + 4) = make_uint2(t.z, t.w);
    }
  };
  // 64 lanes cover 64/LPR rows per pass: two passes when the sub-tile is 64 channels wide
  constexpr int RPP = 64 / LPR, PASSES = RPP < 16 ? 16 / RPP : 1;
  // output rows of this lane and its residual operand for EVERY (row group, pass), loaded before the transposition
  // loop so the memory round trips overlap instead of adding up per row group
  // (tall sub-tiles go in chunks of CH row groups so the operands of all of them need not be held at once)
  constexpr int CH = (MI * PASSES >= 8) ? 2 : MI;   // at most 4 (row group, pass) operand sets in registers
  static_assert(MI % CH == 0, "row-group chunking");
  auto do_chunk = [&](auto c0_tag) {
  constexpr int c0 = decltype(c0_tag)::value;
  int orow[CH][PASSES];   // output rows fit 32 bits (checked on the host): 32-bit index math, 64-bit only in the address
  uint4 rraw[CH][PASSES][2];
  int mt[CH][PASSES], mlen[CH][PASSES];   // F_MASK: the row is kept when mt < mlen * mask_mul (lens loaded up front too)
#pragma unroll
  for (int i = c0; i < c0 + CH; ++i)
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int o = (int)rowmap(row_base + i * 16 + rr + h * RPP);
      orow[i - c0][h] = (lane_on && ok_lo) ? o : -1;
      mt[i - c0][h] = 0;
      mlen[i - c0][h] = 1;
      if ((flags & L2S_F_MASK) && orow[i - c0][h] >= 0) {
        const int clip = (int)((unsigned)o / (unsigned)p.mask_T);   // o >= 0 here
        mt[i - c0][h] = o - clip * p.mask_T;
        mlen[i - c0][h] = p.lens[clip];
      }
      rraw[i - c0][h][0] = make_uint4(0, 0, 0, 0);
      rraw[i - c0][h][1] = make_uint4(0, 0, 0, 0);
      if (has_res && orow[i - c0][h] >= 0) {
        if (flags & L2S_F_RES_F32) {
          const float* q = (const float*)p.R + (int64_t)o * p.ldr + col;
          rraw[i - c0][h][0] = *reinterpret_cast<const uint4*>(q);
          if (ok_hi) rraw[i - c0][h][1] = *reinterpret_cast<const uint4*>(q + 4);
        } else {
          const uint16_t* q = (const uint16_t*)p.R + (int64_t)o * p.ldr + col;
          const uint2 a = *reinterpret_cast<const uint2*>(q);
          const uint2 b = ok_hi ? *reinterpret_cast<const uint2*>(q + 4) : make_uint2(0, 0);
          rraw[i - c0][h][0] = make_uint4(a.x, a.y, b.x, b.y);
        }
      }
    }
#pragma unroll
  for (int i = c0; i < c0 + CH; ++i) {
    // 16 rows of this wave's sub-tile -> scratch (N-tile j at floats [16j, 16j+16) of a row) -> 8 channels per lane
    f32x4_t lo_[PASSES], hi_[PASSES];
    if constexpr (SWZ) {
#pragma unroll
      for (int j = 0; j < NI; ++j) lds_write_f4<0>(swz_w[j], acc[i][j]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int h = 0; h < PASSES; ++h) { lo_[h] = lds_read_f4<0>(swz_lo[h]); hi_[h] = lds_read_f4<0>(swz_hi[h]); }
    } else {
    lds_write_f4<0>(scr_w, acc[i][0]);
    if (NI > 1) lds_write_f4<64>(scr_w, acc[i][NI > 1 ? 1 : 0]);
    if (NI > 2) lds_write_f4<128>(scr_w, acc[i][NI > 2 ? 2 : 0]);
    if (NI > 3) lds_write_f4<192>(scr_w, acc[i][NI > 3 ? 3 : 0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lo_[0] = lds_read_f4<0>(scr_r);
    hi_[0] = lds_read_f4<16>(scr_r);
    if (PASSES > 1) {
      lo_[PASSES - 1] = lds_read_f4<RPP * SROW * 4>(scr_r);
      hi_[PASSES - 1] = lds_read_f4<RPP * SROW * 4 + 16>(scr_r);
    }
    }
    lds_wait();
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
    const f32x4_t lo = lo_[h], hi = hi_[h];
    const int64_t o = orow[i - c0][h];  // output row of this lane's local row, or -1 (widened for the address math)
    if (o < 0) continue;
    const bool keep = !(flags & L2S_F_MASK) || (mt[i - c0][h] < mlen[i - c0][h] * p.mask_mul);
    float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (has_res) {
      const uint4 r0 = rraw[i - c0][h][0], r1 = rraw[i - c0][h][1];
      if (flags & L2S_F_RES_F32) {
        rv[0] = __builtin_bit_cast(float, r0.x); rv[1] = __builtin_bit_cast(float, r0.y);
        rv[2] = __builtin_bit_cast(float, r0.z); rv[3] = __builtin_bit_cast(float, r0.w);
        rv[4] = __builtin_bit_cast(float, r1.x); rv[5] = __builtin_bit_cast(float, r1.y);
        rv[6] = __builtin_bit_cast(float, r1.z); rv[7] = __builtin_bit_cast(float, r1.w);
      } else {
        rv[0] = ET::to_f32((uint16_t)(r0.x & 0xffff)); rv[1] = ET::to_f32((uint16_t)(r0.x >> 16));
        rv[2] = ET::to_f32((uint16_t)(r0.y & 0xffff)); rv[3] = ET::to_f32((uint16_t)(r0.y >> 16));
        rv[4] = ET::to_f32((uint16_t)(r0.z & 0xffff)); rv[5] = ET::to_f32((uint16_t)(r0.z >> 16));
        rv[6] = ET::to_f32((uint16_t)(r0.w & 0xffff)); rv[7] = ET::to_f32((uint16_t)(r0.w >> 16));
      }
    }
    if (flags & L2S_F_ACCUM) ld8(p.C, o, p.ldc, (flags & L2S_F_OUT_F32) != 0, cv);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] + bv[e];
    if (p.alpha != 1.f) {   // wave-uniform: the common alpha = 1 skips the multiplies
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= p.alpha;
    }
    if (flags & L2S_F_RES_PRE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (LIN) {
      if (p.act != L2S_ACT_NONE) {   // wave-uniform: the residual convs carry no activation
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f) + fminf(v[e], 0.f) * sl[e];
      }
    } else {
    switch (p.act) {
      case L2S_ACT_RELU:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        break;
      case L2S_ACT_GELU:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = l2s_gelu(v[e]);
        break;
      case L2S_ACT_SWISH:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = l2s_swish(v[e]);
        break;
      case L2S_ACT_PRELU:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * sv[e];
        break;
      case L2S_ACT_LRELU:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * p.act_slope;
        break;
      case L2S_ACT_TANH:
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
        break;
      default: break;
    }
    }
    if (flags & L2S_F_RES_POST) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (flags & L2S_F_ACCUM) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += cv[e];
    }
    if (!keep) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    if (flags & L2S_F_OUT_F32) {
      float* q = (float*)p.C + o * p.ldc + col;
      *reinterpret_cast<float4*>(q) = make_float4(v[0], v[1], v[2], v[3]);
      if (ok_hi) *reinterpret_cast<float4*>(q + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      st8_16(p.C, o, p.ldc, v);
    }
    if (flags & L2S_F_DUAL) {
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = v[e] * p.slope2;
      if (p.slope2 > 0.f && p.slope2 <= 1.f) {   // wave-uniform; max(x, s x) == leaky_relu(x) bit for bit when 0 < s <= 1
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = fmaxf(v[e], w[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = v[e] >= 0.f ? v[e] : w[e];
      }
      st8_16(p.C2, o, p.ldc2, w);
    }
    }  // passes
  }
  if (MI > CH) asm volatile("" ::: "memory");  // keep the next chunk's operand loads behind this chunk's stores
  };  // chunk
  do_chunk(std::integral_constant<int, 0>{});
  if constexpr (MI > CH) do_chunk(std::integral_constant<int, CH>{});
  if constexpr (MI > 2 * CH) do_chunk(std::integral_constant<int, 2 * CH>{});
  if constexpr (MI > 3 * CH) do_chunk(std::integral_constant<int, 3 * CH>{});
  static_assert(MI <= 4 * CH, "chunk list");
}

// Lean epilogue for the common case "bias, alpha, activation, 16-bit store (+ optional length mask)": no residual,
// accumulate, second output or fp32 store.  All arithmetic happens in the MFMA register layout (a lane holds 4
// consecutive channels, so bias / PReLU slopes are 4 registers per N sub-tile, loaded once per tile), the result is
// rounded to 16 bits there, and only then transposed through LDS - as 8-byte writes and 16-byte reads of half the
// volume of the fp32 transposition - so that each lane stores 8 consecutive channels of a row (16 bytes).  Two row
// groups share one write/read round trip.  Bit-identical to epilogue_impl (same operations in the same order).
//   scr: this wave's private scratch, (MI >= 2 ? 32 : 16) rows x (NI*32 + 16) bytes.
//   ACTK: 0 no activation, 1 linear family, 2 GELU;  MASKED: rows at or past the clip length are zeroed.
//   G2MAX: row groups per LDS round trip (2 halves the round trips, 1 halves the scratch).
//   ubase >= 0 (MASKED only): every row of the tile belongs to ONE clip whose rows start at output row ubase and whose
//   valid length is ulen rows - the mask then needs no per-lane division and no load of lens[] (respair.hip keeps an
//   LDS-DMA in flight under this epilogue, which any compiler-visible global load would drain).
template <typename ET, int MI, int NI, int ACTK, bool MASKED, int G2MAX = 2, typename RowMap>
__device__ __forceinline__ void epilogue_fast16(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const uint32_t scr,
                                                const int lane, const int row_base, const int ncol_base,
                                                const int grp, RowMap rowmap, const int ubase = -1, const int ulen = 0) {
  constexpr int ROWB = NI * 32 + 16;           // scratch row stride in bytes (16-byte aligned, breaks the bank period)
  constexpr int G2 = (MI >= 2 && G2MAX >= 2) ? 2 : 1;   // row groups per round
  constexpr int ROWS = G2 * 16;
  constexpr int LPR = NI * 2;                  // lanes per row after the transpose (8 channels = 16 bytes each)
  constexpr int RPP = 64 / LPR;                // rows per read pass
  constexpr int PASSES = (ROWS + RPP - 1) / RPP;
  static_assert(MI % G2 == 0, "row groups per round");
  const int lm = lane & 15, lg = lane >> 4;
  const int rr = lane / LPR, cc = lane - rr * LPR;
  const int n8 = ncol_base + cc * 8;           // first of this lane's 8 channels after the transpose
  const int col = grp * p.c_gstride + n8;
  const bool ok_lo = n8 < p.N, ok_hi = n8 + 4 < p.N;
  const float alpha = p.alpha;
  const int act = p.act;
  // bias / slopes of this lane's 4 channels in every N sub-tile (MFMA layout: channel = j*16 + lg*4 + e)
  f32x4_t bj[NI], sj[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = ncol_base + j * 16 + lg * 4;
    bj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    sj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (n < p.N) {
      if (p.bias) { const float4 q = *reinterpret_cast<const float4*>(p.bias + grp * p.N + n); bj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
      if (ACTK == 1 && act == L2S_ACT_PRELU) { const float4 q = *reinterpret_cast<const float4*>(p.slope + grp * p.N + n); sj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
    }
    if (ACTK == 1 && act != L2S_ACT_PRELU) {   // linear family y = max(x,0) + min(x,0)*s: none 1, relu 0, lrelu act_slope
      const float s_uni = act == L2S_ACT_RELU ? 0.f : (act == L2S_ACT_LRELU ? p.act_slope : 1.f);
      sj[j] = f32x4_t{s_uni, s_uni, s_uni, s_uni};
    }
  }
  // whole 8-channel groups at 16-byte aligned rows: the stores need no per-lane alignment / tail branches
  const bool st16_uniform = ((p.N & 7) == 0) && ((p.c_gstride & 7) == 0 || grp == 0) && ((ncol_base & 7) == 0) &&
                            ((p.ldc & 7) == 0) && (((uintptr_t)p.C & 15) == 0);
  const uint32_t scr_w = scr + (uint32_t)(lm * ROWB + lg * 8);
  const uint32_t scr_r = scr + (uint32_t)(rr * ROWB + cc * 16);
  auto do_round = [&](auto r0_tag) {   // (a plain unrolled loop is not unrolled for tall sub-tiles and acc goes to scratch)
    constexpr int r0 = decltype(r0_tag)::value;
    // output rows (and mask operands) of this lane's reads of the round: issued first, used after the LDS round trip
    int orow[PASSES];
    int mt[PASSES], mlen[PASSES];
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int lr = h * RPP + rr;             // row inside the round
      const int o = (int)rowmap(row_base + r0 * 16 + lr);
      orow[h] = (lr < ROWS && ok_lo) ? o : -1;
      mt[h] = 0;
      mlen[h] = 1;
      if (MASKED && orow[h] >= 0) {
        if (ubase >= 0) {                                            // wave-uniform: one clip per tile
          mt[h] = o - ubase;
          mlen[h] = ulen;
        } else {
          const int clip = (int)((unsigned)o / (unsigned)p.mask_T);   // o >= 0 here
          mt[h] = o - clip * p.mask_T;
          mlen[h] = p.lens[clip];
        }
      }
    }
    f32x4_t v[G2][NI];
#pragma unroll
    for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
      for (int j = 0; j < NI; ++j) v[g2][j] = acc[r0 + g2][j] + bj[j];
    if (alpha != 1.f) {   // wave-uniform: the common alpha = 1 skips the multiplies
#pragma unroll
      for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
        for (int j = 0; j < NI; ++j) v[g2][j] *= alpha;
    }
    if constexpr (ACTK == 2) {
#pragma unroll
      for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
#ifdef L2S_GELU_ERF      // (A/B switch of the diagnostic builds: the 1.5e-7 erf form)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g2][j][e] = l2s_gelu(v[g2][j][e]);
#else
          const f32x2_t lo = l2s_gelu2(f32x2_t{v[g2][j][0], v[g2][j][1]}), hi = l2s_gelu2(f32x2_t{v[g2][j][2], v[g2][j][3]});
          v[g2][j] = f32x4_t{lo.x, lo.y, hi.x, hi.y};
#endif
        }
    } else if constexpr (ACTK == 1) {
      if (p.act == L2S_ACT_LRELU && p.act_slope > 0.f && p.act_slope <= 1.f) {
        // wave-uniform; max(x, s x) == leaky_relu(x) bit for bit when 0 < s <= 1: 1.5 VALU ops per value instead of 3
        const float s_l = p.act_slope;
#pragma unroll
        for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const f32x4_t sc = v[g2][j] * s_l;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[g2][j][e] = fmaxf(v[g2][j][e], sc[e]);
          }
      } else {
#pragma unroll
      for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[g2][j][e] = fmaxf(v[g2][j][e], 0.f) + fminf(v[g2][j][e], 0.f) * sj[j][e];
      }
    }
#pragma unroll
    for (int g2 = 0; g2 < G2; ++g2)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        u32x2_t pk;
        pk.x = ET::pack2(v[g2][j][0], v[g2][j][1]);
        pk.y = ET::pack2(v[g2][j][2], v[g2][j][3]);
        // (row g2*16 + lm, channels j*16 + lg*4 ..): the lane part is in scr_w, the sub-tile part an immediate
        lds_write_b64_at(scr_w, pk, g2 * 16 * ROWB + j * 32);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4_t t[PASSES];
#pragma unroll
    for (int h = 0; h < PASSES; ++h) t[h] = lds_read_u4(scr_r + (uint32_t)(h * RPP * ROWB));
    lds_wait();
#pragma unroll
    for (int h = 0; h < PASSES; ++h) {
      const int64_t o = orow[h];   // widened for the address math
      if (o < 0) continue;
      u32x4_t d = t[h];
      if (MASKED && !(mt[h] < mlen[h] * (ubase >= 0 ? 1 : p.mask_mul))) d = u32x4_t{0u, 0u, 0u, 0u};
      uint16_t* q = (uint16_t*)p.C + o * p.ldc + col;
      if (st16_uniform) {                       // wave-uniform: every row is 16-byte aligned
        *reinterpret_cast<uint4*>(q) = make_uint4(d.x, d.y, d.z, d.w);
      } else if (ok_hi && (((uintptr_t)q & 15) == 0)) {
        *reinterpret_cast<uint4*>(q) = make_uint4(d.x, d.y, d.z, d.w);
      } else {
        *reinterpret_cast<uint2*>(q) = make_uint2(d.x, d.y);
        if (ok_hi) *reinterpret_cast<uint2*>(q + 4) = make_uint2(d.z, d.w);
      }
    }
  };
  do_round(std::integral_constant<int, 0>{});
  if constexpr (MI > G2) do_round(std::integral_constant<int, G2>{});
  if constexpr (MI > 2 * G2) do_round(std::integral_constant<int, 2 * G2>{});
  if constexpr (MI > 3 * G2) do_round(std::integral_constant<int, 3 * G2>{});
  if constexpr (MI > 4 * G2) do_round(std::integral_constant<int, 4 * G2>{});
  if constexpr (MI > 5 * G2) do_round(std::integral_constant<int, 5 * G2>{});
  if constexpr (MI > 6 * G2) do_round(std::integral_constant<int, 6 * G2>{});
  if constexpr (MI > 7 * G2) do_round(std::integral_constant<int, 7 * G2>{});
  static_assert(MI <= 8 * G2, "round list");
}

// fp32-residual-stream epilogue WITHOUT the LDS transposition (family L2S_EPI_S32: bias, alpha, linear-family
// activation, fp32 or 16-bit residual before/after it, fp32 output).  In the MFMA layout a lane already owns 4
// consecutive fp32 channels = 16 bytes, so residual loads and stores are 16-byte accesses of 64-byte row segments as
// they are: no scratch, no round trips.  Same operations in the same order as epilogue_impl<F_S32, true>.
// ---- 16-bit epilogue WITHOUT the LDS transposition: PAIRED weight-row order -------------------------------------------------
// The MFMAs are issued swapped (weights as the A operand), so a lane of the 16x16 result block j holds 4 CONSECUTIVE channels
// (weight rows 4*lg + e of the block) of one output row (lm): 8 bytes at 16 bits - a quarter of a 32-byte sector, which is why
// the epilogues above transpose through LDS until a lane owns 16 bytes (two dependent LDS round trips per 16-32 rows, the
// longest serial chain of a tile end: ~500 cycles each).  A kernel that READS ITS WEIGHT FRAGMENTS in the paired order
//     block 2b + s, fragment row 4*lg + e   <-   weight row (= output channel) 32 b + 8 lg + 4 s + e
// gets, in the two accumulators acc[i][2b] and acc[i][2b+1] of ONE lane, the 8 consecutive channels 32b + 8lg .. +7 of its row:
// a 16-byte store straight from registers (64-byte row segments per instruction, 16 rows per instruction), 16-byte residual
// loads in the same layout, no scratch, no waits.  Only the LDS address of the weight fragment reads changes (and the XOR key
// of the weight tile's swizzle, paired_w_key(), so that those reads stay conflict-free); weights in HBM keep their order.
// Arithmetic per element is that of epilogue_fast16 / epilogue_impl, in the same order.
__device__ __forceinline__ int paired_w_key(int row) { return (2 * ((row >> 3) & 3)) ^ (row & 7); }
// byte offset, inside a [rows][64 k] 16-bit weight tile with 128-byte rows, of this lane's fragment chunk for blocks 2b + s:
// row 32 b + 8 (lm >> 2) + 4 s + (lm & 3), 16-byte chunk kc (0..7) stored at chunk kc ^ paired_w_key(row); add b * 4096
__device__ __forceinline__ uint32_t paired_w_off(int lm, int s, int kc) {
  const int row = 8 * (lm >> 2) + 4 * s + (lm & 3);
  return (uint32_t)(row * 128 + ((kc ^ paired_w_key(row)) << 4));
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_loads()` runs once, after the first row group has been finished - i.e. behind the wait that retired the epilogue's
// own global loads (bias, slopes, the first chunk's residuals): the place to issue LDS-DMA for the next tile, which hipcc
// would otherwise drain with `vmcnt(0)` at the first use of any of those loads.
// lane lm <-> lm ^ 8 inside the 16-lane rows (same lg): four dwords (inline asm: see dpp_row_ror8 below for why)
__device__ __forceinline__ uint4 dpp_row_ror8_u4(uint4 v) {
  uint32_t r0, r1, r2, r3;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
               : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
  return make_uint4(r0, r1, r2, r3);
}

// FULL (NI % 4 == 0, rows mapped linearly): the two 64-byte pair segments of a row are completed to whole 128-byte lines by the
// lane exchange of epilogue_stream32: instruction A stores rows 0-7 of the group, B rows 8-15.
template <typename ET, int MI, int NI, int EPI, typename RowMap, typename Hook = NoHook, bool FULL = false>
__device__ __forceinline__ void epilogue_direct16(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const int lane,
                                                  const int row_base, const int ncol_base, const int grp, RowMap rowmap,
                                                  const int ubase = -1, const int ulen = 0, Hook after_loads = Hook()) {
  static_assert(NI % 2 == 0, "blocks are consumed in pairs");
  static_assert(!FULL || NI % 4 == 0, "whole lines need both pairs of a 64-channel group");
  static_assert(EPI <= L2S_EPI_G16B, "16-bit families only");
  constexpr bool LEAN = EPI < L2S_EPI_G16A;
  constexpr int ACTK = LEAN ? (EPI - L2S_EPI_F16) / 2 : 1;          // 0 none, 1 linear family, 2 GELU
  constexpr bool CAN_MASK = LEAN ? (((EPI - L2S_EPI_F16) & 1) != 0) : (EPI == L2S_EPI_G16B);
  constexpr bool CAN_RES = !LEAN, CAN_DUAL = EPI == L2S_EPI_G16B;
  constexpr int NB = NI / 2;
  const int lm = lane & 15, lg = lane >> 4;
  const int flags = p.flags, act = p.act;
  const float alpha = p.alpha;
  const bool has_res = CAN_RES && (flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) != 0;
  const bool masked = CAN_MASK && (LEAN || (flags & L2S_F_MASK));
  const bool dual = CAN_DUAL && (flags & L2S_F_DUAL);
  // bias / slopes of this lane's channels: block j = 2b + s holds channels n0 + 32 b + 8 lg + 4 s + e
  const int n0 = ncol_base + 8 * lg;
  f32x4_t bj[NI], sj[NI];
  bool okb[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    okb[b] = n0 + 32 * b + 8 <= p.N;        // whole 8-channel groups (the launchers admit N % 8 == 0 only)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int j = 2 * b + s2, n = n0 + 32 * b + 4 * s2;
      bj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      sj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (okb[b]) {
        if (p.bias) { const float4 q = *reinterpret_cast<const float4*>(p.bias + grp * p.N + n); bj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
        if (ACTK == 1 && act == L2S_ACT_PRELU) { const float4 q = *reinterpret_cast<const float4*>(p.slope + grp * p.N + n); sj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
      }
      if (ACTK == 1 && act != L2S_ACT_PRELU) {
        const float s_uni = act == L2S_ACT_RELU ? 0.f : (act == L2S_ACT_LRELU ? p.act_slope : 1.f);
        sj[j] = f32x4_t{s_uni, s_uni, s_uni, s_uni};
      }
    }
  }
  const int col0 = grp * p.c_gstride + n0;
  constexpr int CHK = MI < 4 ? MI : 4;      // row groups per chunk: bounds the registers of the up-front loads
  auto do_chunk = [&](auto c0_tag) {
    constexpr int c0 = decltype(c0_tag)::value;
    int orow[CHK];
    bool keep[CHK];
    uint4 rraw[CHK][NB];
    // rows, mask operands and residuals of the whole chunk first (unconditional, clamped addresses: one wait for all)
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
      orow[i] = (int)rowmap(row_base + (c0 + i) * 16 + lm);
      const int os = orow[i] < 0 ? 0 : orow[i];
      keep[i] = true;
      int mlen = 0, mt = 0;
      if (masked) {
        if (ubase >= 0) { mt = os - ubase; mlen = ulen; }
        else {
          const int clip = (int)((unsigned)os / (unsigned)p.mask_T);
          mt = os - clip * p.mask_T;
          mlen = p.lens[clip] * p.mask_mul;
        }
        keep[i] = mt < mlen;
      }
      if (has_res) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
          rraw[i][b] = *reinterpret_cast<const uint4*>((const uint16_t*)p.R + (int64_t)os * p.ldr + (okb[b] ? col0 + 32 * b : 0));
      }
    }
#pragma unroll
    for (int i = 0; i < CHK; ++i) {
      uint4 held[2], held2[2];   // FULL: the packed results of an even / odd block pair
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        float v[8], rv[8];
        if (has_res) {
          const uint4 r = rraw[i][b];
          rv[0] = ET::to_f32((uint16_t)(r.x & 0xffff)); rv[1] = ET::to_f32((uint16_t)(r.x >> 16));
          rv[2] = ET::to_f32((uint16_t)(r.y & 0xffff)); rv[3] = ET::to_f32((uint16_t)(r.y >> 16));
          rv[4] = ET::to_f32((uint16_t)(r.z & 0xffff)); rv[5] = ET::to_f32((uint16_t)(r.z >> 16));
          rv[6] = ET::to_f32((uint16_t)(r.w & 0xffff)); rv[7] = ET::to_f32((uint16_t)(r.w >> 16));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = acc[c0 + i][2 * b + (e >> 2)][e & 3] + bj[2 * b + (e >> 2)][e & 3];
        if (alpha != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= alpha;
        }
        if (CAN_RES && (flags & L2S_F_RES_PRE)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if constexpr (ACTK == 2) {
#ifdef L2S_GELU_ERF
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = l2s_gelu(v[e]);
#else
#pragma unroll
          for (int e = 0; e < 8; e += 2) { const f32x2_t g = l2s_gelu2(f32x2_t{v[e], v[e + 1]}); v[e] = g.x; v[e + 1] = g.y; }
#endif
        } else if constexpr (ACTK == 1) {
          if (LEAN && act == L2S_ACT_LRELU && p.act_slope > 0.f && p.act_slope <= 1.f) {   // as epilogue_fast16
            const float s_l = p.act_slope;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * s_l);
          } else if (LEAN && act == L2S_ACT_RELU) {     // one v_max instead of the branch-free family form (3 VALU per value):
#pragma unroll                                         // the conformer's FFN-in GEMMs (positionwise_feed_forward.py:28-30)
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (LEAN || act != L2S_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f) + fminf(v[e], 0.f) * sj[2 * b + (e >> 2)][e & 3];
          }
        }
        if (CAN_RES && (flags & L2S_F_RES_POST)) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += rv[e];
        }
        if (!keep[i]) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
        const uint4 pk = make_uint4(ET::pack2(v[0], v[1]), ET::pack2(v[2], v[3]), ET::pack2(v[4], v[5]), ET::pack2(v[6], v[7]));
        uint4 pk2 = make_uint4(0, 0, 0, 0);
        if (dual) {
          float w[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = v[e] * p.slope2;
          if (p.slope2 > 0.f && p.slope2 <= 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = fmaxf(v[e], w[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = v[e] >= 0.f ? v[e] : w[e];
          }
          pk2 = make_uint4(ET::pack2(w[0], w[1]), ET::pack2(w[2], w[3]), ET::pack2(w[4], w[5]), ET::pack2(w[6], w[7]));
        }
        if constexpr (!FULL) {
          if (orow[i] >= 0 && okb[b]) {
            *reinterpret_cast<uint4*>((uint16_t*)p.C + (int64_t)orow[i] * p.ldc + col0 + 32 * b) = pk;
            if (dual) *reinterpret_cast<uint4*>((uint16_t*)p.C2 + (int64_t)orow[i] * p.ldc2 + col0 + 32 * b) = pk2;
          }
        } else {
          held[b & 1] = pk;
          held2[b & 1] = pk2;
          if (b & 1) {      // both pairs of a 64-channel line are ready: complete lines through the lane exchange
            const bool lower = lm < 8;
            const uint4 got = dpp_row_ror8_u4(held[1]);
            const uint4 xa = lower ? held[0] : got, xb = lower ? got : held[0];
            // instruction A: row (lm & 7), half-line (lm >> 3); instruction B: row 8 + (lm & 7), the other half
            const int hA = lm >> 3, hB = 1 - hA;
            const int cA = col0 + 32 * (b - 1) + 32 * hA, cB = col0 + 32 * (b - 1) + 32 * hB;
            const int oa = (int)rowmap(row_base + (c0 + i) * 16 + (lm & 7)), ob = (int)rowmap(row_base + (c0 + i) * 16 + 8 + (lm & 7));
            const bool okA = oa >= 0 && (hA ? okb[b] : okb[b - 1]), okB = ob >= 0 && (hB ? okb[b] : okb[b - 1]);
            if (okA) *reinterpret_cast<uint4*>((uint16_t*)p.C + (int64_t)oa * p.ldc + cA) = xa;
            if (okB) *reinterpret_cast<uint4*>((uint16_t*)p.C + (int64_t)ob * p.ldc + cB) = xb;
            if (dual) {
              const uint4 got2 = dpp_row_ror8_u4(held2[1]);
              const uint4 ya = lower ? held2[0] : got2, yb = lower ? got2 : held2[0];
              if (okA) *reinterpret_cast<uint4*>((uint16_t*)p.C2 + (int64_t)oa * p.ldc2 + cA) = ya;
              if (okB) *reinterpret_cast<uint4*>((uint16_t*)p.C2 + (int64_t)ob * p.ldc2 + cB) = yb;
            }
          }
        }
      }
      if (c0 == 0 && i == 0) after_loads();
    }
  };
  do_chunk(std::integral_constant<int, 0>{});
  if constexpr (MI > CHK) do_chunk(std::integral_constant<int, CHK>{});
  if constexpr (MI > 2 * CHK) do_chunk(std::integral_constant<int, 2 * CHK>{});
  if constexpr (MI > 3 * CHK) do_chunk(std::integral_constant<int, 3 * CHK>{});
  static_assert(MI <= 4 * CHK, "chunk list");
}

// The residual-stream update x += alpha * (W h + b) (fp32 residual in, fp32 out, no activation, RES_POST: out-proj, FC2, the
// conformer's N = 512 projections) for a wave tile that lies wholly inside the matrix.  The generic code below reaches every
// residual element through its own branch (row valid? column valid? residual kind?), and hipcc then waits `vmcnt(0)` behind
// each load: 32 dependent memory round trips per wave tile, 19-22 us per 256 x 256 tile measured with in-kernel stamps
// (tools/phase_stamps.py) - a third of the whole tile for out-proj.  Here nothing is conditional: offsets are 32-bit from the
// uniform base pointers, the residual loads of two row groups (8 x 16 bytes per lane) are issued back to back, and two such
// batches stay in flight while the previous batch is finished and stored (counted vmcnt by the compiler, 64 VGPRs of landing
// space - the K loop's fragment registers are dead here).  Same arithmetic in the same order as the generic path.
// Whole 128-byte lines per instruction: in the MFMA layout the 16-byte pieces of one instruction are 64-byte row segments of 16
// rows (lanes lg 0..3 of a row); the other half of each line belongs to the SAME lanes' next block.  One DPP rotation inside the
// 16-lane rows (lane lm <-> lm ^ 8: `row_ror:8`) hands every lane its partner's second-block values, after which instruction A
// covers rows 0-7 of the group and instruction B rows 8-15, each row as one full line (lanes lm < 8 the first half, lm >= 8 the
// second).  The residual is loaded in that layout and added there: same operations per element, in the same order.
__device__ __forceinline__ f32x4_t dpp_row_ror8(f32x4_t v) {
  // (inline asm: hipcc of ROCm 7.2 merged the four __builtin_amdgcn_update_dpp calls of one vector into the first one's result;
  // the s_nop covers the two wait states a DPP read needs behind the VALU write of its source)
  float r0, r1, r2, r3;
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
               "v_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
               : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
  return f32x4_t{r0, r1, r2, r3};
}

template <typename ET, int MI, int NI, typename RowMap>
__device__ __forceinline__ void epilogue_stream32(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const int lane,
                                                  const int row_base, const int ncol_base, RowMap rowmap) {
  static_assert(MI % 2 == 0 && MI >= 4 && NI % 2 == 0, "row groups are streamed in pairs, two pairs in flight; blocks in pairs");
  constexpr int NB = NI / 2;
  const int lm = lane & 15, lg = lane >> 4;
  const bool lower = lm < 8;                 // this lane stores its own first-block values in instruction A, the partner's in B
  const float alpha = p.alpha;
  const char* __restrict__ rbase = (const char*)p.R;
  char* __restrict__ cbase = (char*)p.C;
  f32x4_t bj[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    bj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (p.bias) { const float4 q = *reinterpret_cast<const float4*>(p.bias + ncol_base + j * 16 + lg * 4); bj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
  }
  // store-layout coordinates: instruction A = row (lm & 7), half (lm >> 3); instruction B = row 8 + (lm & 7), the other half
  const uint32_t colA = (uint32_t)(ncol_base + 16 * (lm >> 3) + lg * 4) * 4u;
  const uint32_t colB = (uint32_t)(ncol_base + 16 * (1 - (lm >> 3)) + lg * 4) * 4u;
  // byte offsets of row group 0, block pair 0; group i adds i * gstep (uniform: the tile lies inside the matrix, where both
  // callers' row maps are m * out_row_mul + out_row_add), pair b adds 128.  R and C share the row pitch (stream32_ok).
  const uint32_t pitch = (uint32_t)p.ldc * 4u;
  const uint32_t gstep = 16u * (uint32_t)p.out_row_mul * pitch;
  const uint32_t offA0 = (uint32_t)rowmap(row_base + (lm & 7)) * pitch + colA;
  const uint32_t offB0 = (uint32_t)rowmap(row_base + 8 + (lm & 7)) * pitch + colB;
  auto ld = [&](uint32_t off) {
    const float4 q = *reinterpret_cast<const float4*>(rbase + (size_t)off);
    return f32x4_t{q.x, q.y, q.z, q.w};
  };
  auto st = [&](uint32_t off, f32x4_t v) { *reinterpret_cast<float4*>(cbase + (size_t)off) = make_float4(v[0], v[1], v[2], v[3]); };
  // rv[g][b][0/1]: residual lines of instruction A / B of row group 2 pr + g, block pair b
  auto load_pair = [&](f32x4_t (&rv)[2][NB][2], auto pr_tag) {
    constexpr int pr = decltype(pr_tag)::value;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        rv[g][b][0] = ld(offA0 + (uint32_t)(2 * pr + g) * gstep + (uint32_t)(b * 128));
        rv[g][b][1] = ld(offB0 + (uint32_t)(2 * pr + g) * gstep + (uint32_t)(b * 128));
      }
  };
  auto finish_pair = [&](f32x4_t (&rv)[2][NB][2], auto pr_tag) {
    constexpr int pr = decltype(pr_tag)::value;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int i = 2 * pr + g;
        const f32x4_t u0 = (acc[i][2 * b] + bj[2 * b]) * alpha;          // own row, first block
        const f32x4_t u1 = (acc[i][2 * b + 1] + bj[2 * b + 1]) * alpha;  // own row, second block -> to the partner lane
        const f32x4_t got = dpp_row_ror8(u1);
        f32x4_t xa, xb;
#pragma unroll
        for (int e = 0; e < 4; ++e) { xa[e] = lower ? u0[e] : got[e]; xb[e] = lower ? got[e] : u0[e]; }
        st(offA0 + (uint32_t)i * gstep + (uint32_t)(b * 128), xa + rv[g][b][0]);
        st(offB0 + (uint32_t)i * gstep + (uint32_t)(b * 128), xb + rv[g][b][1]);
      }
  };
  f32x4_t ra[2][NB][2], rb[2][NB][2];
  load_pair(ra, std::integral_constant<int, 0>{});
  load_pair(rb, std::integral_constant<int, 1>{});
  finish_pair(ra, std::integral_constant<int, 0>{});
  if constexpr (MI > 4) load_pair(ra, std::integral_constant<int, 2>{});
  finish_pair(rb, std::integral_constant<int, 1>{});
  if constexpr (MI > 6) load_pair(rb, std::integral_constant<int, 3>{});
  if constexpr (MI > 4) finish_pair(ra, std::integral_constant<int, 2>{});
  if constexpr (MI > 6) finish_pair(rb, std::integral_constant<int, 3>{});
  static_assert(MI <= 8, "row group pairs");
}

// The ResBlock-sum update of the un-fused vocoder stage (family L2S_EPI_X32, speech-resynthesis/models.py:103-108 on top of a
// ResBlock's last conv, :34-41):  xs = [xs +] (conv + b) * alpha + r16, rows at or past the clip length zeroed, optionally
// C2 = leaky_relu(xs) in 16 bits for the next stage.  The catch-all epilogue spills on a 128x64 wave tile (the family ran at
// 0.6x on this kernel and was kept on the 256x128 one); this one works in the MFMA layout (a lane owns 4 consecutive fp32
// channels = 16 bytes), loads the 16-bit residuals (and the previous sum) of two row groups up front through clamped,
// unconditional addresses, and predicates only its stores.  Operations per element as epilogue_impl, in the same order.
template <typename ET, int MI, int NI, typename RowMap>
__device__ __forceinline__ void epilogue_stream32x(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const int lane,
                                                   const int row_base, const int ncol_base, RowMap rowmap) {
  static_assert(MI % 2 == 0, "row groups are streamed in pairs");
  const int lm = lane & 15, lg = lane >> 4;
  const int flags = p.flags;
  const bool accum = (flags & L2S_F_ACCUM) != 0, dual = (flags & L2S_F_DUAL) != 0, masked = (flags & L2S_F_MASK) != 0;
  const float alpha = p.alpha;
  f32x4_t bj[NI];
  bool okj[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = ncol_base + j * 16 + lg * 4;
    okj[j] = n + 4 <= p.N;
    bj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (okj[j] && p.bias) { const float4 q = *reinterpret_cast<const float4*>(p.bias + n); bj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
  }
  const int col = ncol_base + lg * 4;
  auto do_pair = [&](auto pr_tag) {
    constexpr int pr = decltype(pr_tag)::value;
    int orow[2];
    bool keep[2];
    uint2 rr[2][NI];
    float4 old[2][NI];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      orow[g] = (int)rowmap(row_base + (2 * pr + g) * 16 + lm);
      const int os = orow[g] < 0 ? 0 : orow[g];
      keep[g] = true;
      if (masked) {
        const int clip = (int)((unsigned)os / (unsigned)p.mask_T);
        keep[g] = (os - clip * p.mask_T) < p.lens[clip] * p.mask_mul;
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int c = okj[j] ? col + j * 16 : 0;
        rr[g][j] = *reinterpret_cast<const uint2*>((const uint16_t*)p.R + (int64_t)os * p.ldr + c);
        old[g][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (accum) old[g][j] = *reinterpret_cast<const float4*>((const float*)p.C + (int64_t)os * p.ldc + c);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        f32x4_t v = acc[2 * pr + g][j] + bj[j];
        if (alpha != 1.f) v = v * alpha;
        const uint2 q = rr[g][j];
        v = v + f32x4_t{ET::to_f32((uint16_t)(q.x & 0xffff)), ET::to_f32((uint16_t)(q.x >> 16)),
                        ET::to_f32((uint16_t)(q.y & 0xffff)), ET::to_f32((uint16_t)(q.y >> 16))};
        if (accum) v = v + f32x4_t{old[g][j].x, old[g][j].y, old[g][j].z, old[g][j].w};
        if (!keep[g]) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (orow[g] >= 0 && okj[j]) {
          *reinterpret_cast<float4*>((float*)p.C + (int64_t)orow[g] * p.ldc + col + j * 16) = make_float4(v[0], v[1], v[2], v[3]);
          if (dual) {
            float w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = v[e] * p.slope2;
            if (p.slope2 > 0.f && p.slope2 <= 1.f) {
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] = fmaxf(v[e], w[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] = v[e] >= 0.f ? v[e] : w[e];
            }
            *reinterpret_cast<uint2*>((uint16_t*)p.C2 + (int64_t)orow[g] * p.ldc2 + col + j * 16) =
                make_uint2(ET::pack2(w[0], w[1]), ET::pack2(w[2], w[3]));
          }
        }
      }
  };
  do_pair(std::integral_constant<int, 0>{});
  if constexpr (MI > 2) do_pair(std::integral_constant<int, 1>{});
  if constexpr (MI > 4) do_pair(std::integral_constant<int, 2>{});
  if constexpr (MI > 6) do_pair(std::integral_constant<int, 3>{});
  static_assert(MI <= 8, "row group pairs");
}

// Is this wave tile the plain residual-stream case epilogue_stream32 covers?  (wave-uniform)
__device__ __forceinline__ bool stream32_ok(const l2s_gemm_desc& p, const int row_base, const int ncol_base, const int rows,
                                            const int cols) {
  const int want = L2S_F_RES_POST | L2S_F_RES_F32 | L2S_F_OUT_F32;
  // 32-bit byte offsets: the last output row's end must lie below 4 GB for both arrays
  const uint64_t last = (uint64_t)((int64_t)(p.M - 1) * p.out_row_mul + p.out_row_add + 1);
  return p.flags == want && p.act == L2S_ACT_NONE && row_base + rows <= p.M && ncol_base + cols <= p.N && p.ldr == p.ldc &&
         last * (uint64_t)p.ldr * 4u < (1ull << 32) && last * (uint64_t)p.ldc * 4u < (1ull << 32);
}

template <typename ET, int MI, int NI, typename RowMap>
__device__ __forceinline__ void epilogue_direct32(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const int lane,
                                                  const int row_base, const int ncol_base, const int grp,
                                                  RowMap rowmap) {
#ifndef L2S_NO_STREAM32   // (A/B switch of the diagnostic builds)
  if constexpr (MI % 2 == 0 && MI >= 4) {
    if (grp == 0 && stream32_ok(p, row_base, ncol_base, MI * 16, NI * 16)) {
      epilogue_stream32<ET, MI, NI>(p, acc, lane, row_base, ncol_base, rowmap);
      return;
    }
  }
#endif
  const int lm = lane & 15, lg = lane >> 4;
  const int flags = p.flags;
  const bool has_res = (flags & (L2S_F_RES_PRE | L2S_F_RES_POST)) != 0;
  const float alpha = p.alpha;
  f32x4_t bj[NI], sj[NI];
  bool okj[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int n = ncol_base + j * 16 + lg * 4;
    okj[j] = n < p.N;
    bj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    sj[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (okj[j]) {
      if (p.bias) { const float4 q = *reinterpret_cast<const float4*>(p.bias + grp * p.N + n); bj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
      if (p.act == L2S_ACT_PRELU) { const float4 q = *reinterpret_cast<const float4*>(p.slope + grp * p.N + n); sj[j] = f32x4_t{q.x, q.y, q.z, q.w}; }
    }
    if (p.act != L2S_ACT_PRELU) {
      const float s_uni = p.act == L2S_ACT_RELU ? 0.f : (p.act == L2S_ACT_LRELU ? p.act_slope : 1.f);
      sj[j] = f32x4_t{s_uni, s_uni, s_uni, s_uni};
    }
  }
  auto do_group = [&](auto i_tag) {
    constexpr int i = decltype(i_tag)::value;
    const int o32 = (int)rowmap(row_base + i * 16 + lm);
    if (o32 < 0) return;
    const int64_t o = o32;
    f32x4_t rv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {   // the residual of the whole row group first: the loads overlap
      rv[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (has_res && okj[j]) {
        const int col = grp * p.c_gstride + ncol_base + j * 16 + lg * 4;
        if (flags & L2S_F_RES_F32) {
          const float4 q = *reinterpret_cast<const float4*>((const float*)p.R + o * p.ldr + col);
          rv[j] = f32x4_t{q.x, q.y, q.z, q.w};
        } else {
          const uint2 q = *reinterpret_cast<const uint2*>((const uint16_t*)p.R + o * p.ldr + col);
          rv[j] = f32x4_t{ET::to_f32((uint16_t)(q.x & 0xffff)), ET::to_f32((uint16_t)(q.x >> 16)),
                          ET::to_f32((uint16_t)(q.y & 0xffff)), ET::to_f32((uint16_t)(q.y >> 16))};
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      if (!okj[j]) continue;
      f32x4_t v = (acc[i][j] + bj[j]) * alpha;
      if (flags & L2S_F_RES_PRE) v = v + rv[j];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f) + fminf(v[e], 0.f) * sj[j][e];
      if (flags & L2S_F_RES_POST) v = v + rv[j];
      const int col = grp * p.c_gstride + ncol_base + j * 16 + lg * 4;
      *reinterpret_cast<float4*>((float*)p.C + o * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  do_group(std::integral_constant<int, 0>{});
  if constexpr (MI > 1) do_group(std::integral_constant<int, 1>{});
  if constexpr (MI > 2) do_group(std::integral_constant<int, 2>{});
  if constexpr (MI > 3) do_group(std::integral_constant<int, 3>{});
  if constexpr (MI > 4) do_group(std::integral_constant<int, 4>{});
  if constexpr (MI > 5) do_group(std::integral_constant<int, 5>{});
  if constexpr (MI > 6) do_group(std::integral_constant<int, 6>{});
  if constexpr (MI > 7) do_group(std::integral_constant<int, 7>{});
  static_assert(MI <= 8, "row group list");
}

// One epilogue family per kernel instantiation (tapgemm_tiles.h: pick_epilogue): the host launches the kernel whose
// family covers the descriptor's flags / activation.
template <typename ET, int MI, int NI, int EPI, typename RowMap>
__device__ __forceinline__ void epilogue(const l2s_gemm_desc& p, f32x4_t (&acc)[MI][NI], const uint32_t scr,
                                         const int lane, const int row_base, const int ncol_base, const int grp,
                                         RowMap rowmap) {
  if constexpr (EPI < L2S_EPI_G16A)
    epilogue_fast16<ET, MI, NI, (EPI - L2S_EPI_F16) / 2, ((EPI - L2S_EPI_F16) & 1) != 0>(p, acc, scr, lane, row_base, ncol_base, grp, rowmap);
  else if constexpr (EPI == L2S_EPI_G16A) epilogue_impl<ET, MI, NI, F_G16A, true>(p, acc, scr, lane, row_base, ncol_base, grp, rowmap);
  else if constexpr (EPI == L2S_EPI_G16B) epilogue_impl<ET, MI, NI, F_G16B, true>(p, acc, scr, lane, row_base, ncol_base, grp, rowmap);
  else if constexpr (EPI == L2S_EPI_S32) {
#ifndef L2S_NO_STREAM32
    if constexpr (MI % 2 == 0 && MI >= 4) {      // the plain x += W h + b of a full wave tile: no transposition, deep loads
      if (grp == 0 && stream32_ok(p, row_base, ncol_base, MI * 16, NI * 16)) {
        epilogue_stream32<ET, MI, NI>(p, acc, lane, row_base, ncol_base, rowmap);
        return;
      }
    }
#endif
    epilogue_impl<ET, MI, NI, F_S32, true>(p, acc, scr, lane, row_base, ncol_base, grp, rowmap);
  }
  else epilogue_impl<ET, MI, NI, 0x7f, false>(p, acc, scr, lane, row_base, ncol_base, grp, rowmap);
}

}  // namespace l2s
