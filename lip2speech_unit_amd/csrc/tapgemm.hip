// Tap-GEMM: the one dense-contraction kernel family of the path (Linear / Conv1d / Conv2d / ConvTranspose1d phases).
//
//   C[o(m), n] = epi( sum_tap sum_c A[src(m,tap), c] * W[n, tap*Cin + c] )
//
// gfx950 design: 4 or 8 waves, block tile BM x BN, K-tile 64 (two v_mfma_f32_16x16x32 k-steps per 16x16
// sub-tile).  Both operands are staged global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
// VGPR round trip, no ds_write): the per-lane SOURCE address carries the im2col gather (Linear / Conv1d / Conv2d taps),
// the zero fill (lanes outside the tensor read a 16-byte zero page) and the XOR swizzle (chunk ^ (row & 7) on 128-byte
// rows) that makes every ds_read_b128 fragment read bank-conflict free; the LDS destination stays lane-linear.
// 2- or 3-deep LDS ring (up to 144 KiB): with 3 stages two tiles stay in flight behind a counted s_waitcnt vmcnt(N)
// and a raw s_barrier (a __syncthreads() would drain the DMA queue), one barrier per tile; 256-row tiles run 8 waves.
// Blocks are persistent: each walks its list of output tiles with the K-loop flattened across tiles, so the ring keeps
// streaming the next tile's operands under the current tile's epilogue; tiles are dealt so that the blocks of one XCD
// (blockIdx % 8) work on neighbouring tiles and share operand panels in that XCD's L2.  The MFMA is issued
// "swapped" (W rows as the A operand) so each lane ends with 4 consecutive output channels of one output row: bias /
// residual / store are 8- or 16-byte vectors.
#include "l2s_common.h"
#include "tapgemm_tiles.h"

using l2s::pick_tile;

// tapgemm_inst.hip, one object per (dtype, mode, epilogue family)
#define L2S_DECL(m, e)                                                     \
  int l2s_tapgemm_f16_m##m##_e##e(const l2s_gemm_desc& d, hipStream_t st); \
  int l2s_tapgemm_bf16_m##m##_e##e(const l2s_gemm_desc& d, hipStream_t st);
#define L2S_DECL_MODE(m) L2S_DECL(m, 0) L2S_DECL(m, 1) L2S_DECL(m, 2) L2S_DECL(m, 3) L2S_DECL(m, 4) L2S_DECL(m, 5) \
                         L2S_DECL(m, 6) L2S_DECL(m, 7) L2S_DECL(m, 8) L2S_DECL(m, 9)
L2S_DECL_MODE(0) L2S_DECL_MODE(1) L2S_DECL_MODE(2)
#undef L2S_DECL_MODE
#undef L2S_DECL

static int launch_dtype_mode(const l2s_gemm_desc& d, hipStream_t st) {
  typedef int (*fn_t)(const l2s_gemm_desc&, hipStream_t);
#define L2S_E(t, m, e) l2s_tapgemm_##t##_m##m##_e##e
#define L2S_ROW(t, m) {L2S_E(t, m, 0), L2S_E(t, m, 1), L2S_E(t, m, 2), L2S_E(t, m, 3), L2S_E(t, m, 4), \
                       L2S_E(t, m, 5), L2S_E(t, m, 6), L2S_E(t, m, 7), L2S_E(t, m, 8), L2S_E(t, m, 9)}
  static const fn_t table[2][3][l2s::L2S_EPI_COUNT] = {{L2S_ROW(f16, 0), L2S_ROW(f16, 1), L2S_ROW(f16, 2)},
                                                       {L2S_ROW(bf16, 0), L2S_ROW(bf16, 1), L2S_ROW(bf16, 2)}};
#undef L2S_ROW
#undef L2S_E
  if (d.mode < 0 || d.mode > 2) return L2S_EINVAL;
  return table[d.dtype == L2S_F16 ? 0 : 1][d.mode][l2s::pick_epilogue(d.flags, d.act)](d, st);
}

// phasegemm.hip: phase-staggered 256x256 kernel for the wide lean-epilogue Linear layers
bool l2s_phasegemm_eligible(const l2s_gemm_desc& d);
int l2s_phasegemm_launch(const l2s_gemm_desc& d, hipStream_t st);
int l2s_phasegemm_ktab_launch(const l2s_gemm_desc& d, hipStream_t st);
// patchconv.hip: LDS-resident-patch kernel for the Cin = N = 64 stride-1 convolutions
bool l2s_patchconv_eligible(const l2s_gemm_desc& d);
int l2s_patchconv_launch(const l2s_gemm_desc& d, hipStream_t st);
static bool patch_enabled() {
  static const bool on = [] { const char* e = getenv("L2S_NO_PATCHCONV"); return !(e && atoi(e)); }();  // tuning aid
  return on;
}

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
  if ((int64_t)d.M * d.out_row_mul + d.out_row_add >= ((int64_t)1 << 31)) return L2S_EUNSUPPORTED;  // 32-bit row index math
  hipStream_t st = (hipStream_t)stream;
  if (d.dtype != L2S_F16 && d.dtype != L2S_BF16) return L2S_EINVAL;
  if (d.ktab) return l2s_phasegemm_ktab_launch(d, st);
  if (l2s_phasegemm_eligible(d)) return l2s_phasegemm_launch(d, st);
  if (patch_enabled() && l2s_patchconv_eligible(d)) return l2s_patchconv_launch(d, st);
  return launch_dtype_mode(d, st);
}

extern "C" int l2s_tapgemm_variant(const l2s_gemm_desc* hd) {
  if (!hd || hd->M <= 0 || hd->N <= 0) return L2S_EINVAL;
  if (hd->ktab || l2s_phasegemm_eligible(*hd)) return 256256;           // phasegemm.hip
  if (patch_enabled() && l2s_patchconv_eligible(*hd)) return 999000 + hd->N;  // patchconv.hip: 999064 / 999128
  return pick_tile(hd->M, hd->N, hd->groups > 0 ? hd->groups : 1);
}

extern "C" int l2s_tapgemm_epilogue_family(const l2s_gemm_desc* hd) {
  if (!hd) return L2S_EINVAL;
  return l2s::pick_epilogue(hd->flags, hd->act);
}
