// ResNet-18 3D/2D frontend kernels that are not plain tap-GEMMs (avhubert/resnet.py:131-169):
//   stem Conv3d(1->64,k5x7x7,s1x2x2)+BN3d+PReLU as an implicit-GEMM on MFMA with the lip-crop frame tile staged in LDS,
//   MaxPool3d(1x3x3,s1x2x2), AdaptiveAvgPool2d(1).
#include "l2s_common.h"
#include <cstdlib>

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Stem.  One block = one output frame (b,t) x 22 conv rows (half of the 44x44 conv output).
// LDS holds the 5-frame x 49-row x 88(+8)-col input window as 16-bit; column x is stored at x+3 so that the 8
// consecutive taps dx=0..7 of output column ox start at element 2*ox (4-byte aligned ds_read_b32 x4 per fragment).
// K ordering k = (dt*7+dy)*8 + dx (dx==7 carries a zero weight), K = 280 padded to 288 = 9 MFMA k-steps.
// Each wave keeps all 64x288 weights as 36 MFMA fragments in registers (read once per block from L2) and walks
// 16-pixel tiles; the im2col operand is gathered from LDS, never materialised.
// ---------------------------------------------------------------------------------------------------------------
constexpr int SH = 88, SW = 88, SHO = 44, SWO = 44;
constexpr int SR = 22;                 // conv rows per block
constexpr int SROWS = 2 * SR + 5;      // 49 input rows
constexpr int SCOLS = 96;              // 88 + 3 left pad + 5 right pad
constexpr int SKS = 9;                 // k-steps of 32

template <typename ET, bool XF32, bool SWISH>
__global__ __launch_bounds__(256) void stem_kernel(const void* __restrict__ xin, const uint16_t* __restrict__ w,
                                                   const float* __restrict__ bias, const float* __restrict__ slope,
                                                   uint16_t* __restrict__ y, int B, int T) {
  __shared__ __attribute__((aligned(16))) uint16_t tile[5 * SROWS * SCOLS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = blockIdx.x;  // 0/1: conv rows [22*half, 22*half+22)
  const int bt = blockIdx.y;
  const int b = bt / T, t = bt - b * T;
  const int ylo = 2 * (SR * half) - 3;  // global input row of local row 0

  // ---- stage the input window (zero outside the frame / clip) ----
  for (int idx = tid; idx < 5 * SROWS * (SCOLS / 2); idx += 256) {
    const int cp = idx % (SCOLS / 2);       // column pair
    const int rr = idx / (SCOLS / 2);
    const int row = rr % SROWS, f = rr / SROWS;
    const int tt = t + f - 2, gy = ylo + row;
    const int x0 = cp * 2 - 3;              // global column of the first element of the pair
    float v0 = 0.f, v1 = 0.f;
    if (tt >= 0 && tt < T && gy >= 0 && gy < SH) {
      const int64_t base = (((int64_t)b * T + tt) * SH + gy) * SW;
      if (XF32) {
        const float* xp = (const float*)xin + base;
        if (x0 >= 0 && x0 < SW) v0 = xp[x0];
        if (x0 + 1 >= 0 && x0 + 1 < SW) v1 = xp[x0 + 1];
      } else {
        const uint16_t* xp = (const uint16_t*)xin + base;
        if (x0 >= 0 && x0 < SW) v0 = ET::to_f32(xp[x0]);
        if (x0 + 1 >= 0 && x0 + 1 < SW) v1 = ET::to_f32(xp[x0 + 1]);
      }
    }
    const uint32_t pk = ET::pack2(v0, v1);
    *reinterpret_cast<uint32_t*>(tile + (f * SROWS + row) * SCOLS + cp * 2) = pk;
  }

  // ---- weights -> registers: wf[ni][ks] is the MFMA operand for channels ni*16.. and k-step ks ----
  const int lm = lane & 15, lg = lane >> 4;
  frag16 wf[4][SKS];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks)
      wf[ni][ks].u = *reinterpret_cast<const uint4*>(w + (ni * 16 + lm) * (SKS * 32) + ks * 32 + lg * 8);

  int koff[SKS];  // LDS element offset of (dt,dy) for this lane's k-chunk
#pragma unroll
  for (int ks = 0; ks < SKS; ++ks) {
    int q = ks * 4 + lg;
    q = q > 34 ? 34 : q;  // K padding: weights there are zero, keep the address in bounds
    const int dt = q / 7, dy = q - dt * 7;
    koff[ks] = (dt * SROWS + dy) * SCOLS;
  }
  float4 bs[4], sl[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    bs[ni] = *reinterpret_cast<const float4*>(bias + ni * 16 + lg * 4);
    sl[ni] = SWISH ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(slope + ni * 16 + lg * 4);
  }
  __syncthreads();

  constexpr int NPIX = SR * SWO;                 // 968
  constexpr int NTILES = (NPIX + 15) / 16;       // 61
  for (int tl = wave; tl < NTILES; tl += 4) {
    const int p = tl * 16 + lm;
    const bool pv = p < NPIX;
    const int pp = pv ? p : NPIX - 1;
    const int oyl = pp / SWO, ox = pp - oyl * SWO;
    const int abase = (2 * oyl) * SCOLS + 2 * ox;
    f32x4_t acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[ni] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(tile + abase + koff[ks]);
      frag16 fa;
      fa.u = make_uint4(src[0], src[1], src[2], src[3]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = ET::mfma(wf[ni][ks], fa, acc[ni]);
    }
    if (pv) {
      const int oy = SR * half + oyl;
      uint16_t* yo = y + (((int64_t)bt * SHO + oy) * SWO + ox) * 64;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        float v0 = acc[ni][0] + bs[ni].x, v1 = acc[ni][1] + bs[ni].y;
        float v2 = acc[ni][2] + bs[ni].z, v3 = acc[ni][3] + bs[ni].w;
        if (SWISH) {   // espnet conv3d_extractor.py:63-64 (relu_type 'swish')
          v0 = l2s_swish(v0); v1 = l2s_swish(v1); v2 = l2s_swish(v2); v3 = l2s_swish(v3);
        } else {
          v0 = v0 >= 0.f ? v0 : v0 * sl[ni].x; v1 = v1 >= 0.f ? v1 : v1 * sl[ni].y;
          v2 = v2 >= 0.f ? v2 : v2 * sl[ni].z; v3 = v3 >= 0.f ? v3 : v3 * sl[ni].w;
        }
        uint2 q;
        q.x = ET::pack2(v0, v1);
        q.y = ET::pack2(v2, v3);
        *reinterpret_cast<uint2*>(yo + ni * 16 + lg * 4) = q;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused stem: Conv3d + BN + PReLU + MaxPool3d(1,3,3 / 1,2,2) without the 248 KB/frame conv activation ever reaching HBM.
// One block = one clip x 4 pooled rows (9 conv rows, 23 input rows) x a chunk of FT consecutive frames.  The 5-frame
// input window lives in an LDS ring (one new 23x88 slab per frame), the 64x288 weights stay in registers for the whole
// chunk, the 9x44x64 conv tile is staged in LDS and pooled from there: HBM sees the frames once (x 1.44 row-halo) and the
// pooled [22,22,64] output once.
// ---------------------------------------------------------------------------------------------------------------
constexpr int FP = 4;                    // pooled rows per block
constexpr int FCR = 2 * FP + 1;          // conv rows per block (9)
constexpr int FIR = 2 * FCR + 5;         // input rows per slab (23)
constexpr int FT = 10;                   // frames per block (25 when the grid stays large: see stem_pool_dispatch)
// conv tile in LDS: [9 rows][1 halo + 44 columns][64 channels + 8 pad]: the 144-byte pixel pitch (16-byte aligned for the b128 reads) spreads the MFMA-layout
// writes of 16 pixels over all banks (no XOR swizzle), so every pooling tap sits at a compile-time offset from the window's
// first one; the halo column (conv column -1) and the first group's row -1 hold -inf and are never written
constexpr int CW = SWO + 1;
constexpr int CPITCH = 72;

#ifdef L2S_STEM_STAMPS
__device__ unsigned long long* g_stem_stamps = nullptr;
#endif
template <int N> struct IntC { static constexpr int value = N; };
// v_pk_max_f16 as is: the compiler's fmaxnum lowering first canonicalises both operands (two more VALU ops per maximum)
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Raw decoder frames as the stem's input (XK == 2): uint8 [B,T,Hin,Win]; the centre crop and the (x/255 - mean)/std
// normalisation of hubert_dataset.py:242-245 / utils.py:56-95 happen in the slab fetch, in l2s_preprocess_frames' exact
// arithmetic, so the 16-bit normalised frames never exist in HBM.
struct StemU8 { int Hin, Win, dy, dx; float mean, inv_std; };

template <typename ET, int XK, bool SWISH>     // XK: 0 = 16-bit frames, 1 = fp32 frames, 2 = raw uint8 frames
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(const void* __restrict__ xin, const uint16_t* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ slope,
                                                        uint16_t* __restrict__ y, int B, int T, int ft, StemU8 u8) {
  __shared__ __attribute__((aligned(16))) uint16_t ring[5 * FIR * SCOLS];     // 22 KB
  __shared__ __attribute__((aligned(16))) uint16_t cbuf[FCR * CW * CPITCH];   // 57 KB
  __shared__ __attribute__((aligned(16))) float sbs[128];                     // bias, PReLU slope
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, lg = lane >> 4;
  const int grp = blockIdx.x, b = blockIdx.z;
  const int t_begin = blockIdx.y * ft;
  const int t_end = t_begin + ft < T ? t_begin + ft : T;
  const int p0 = grp * FP;                       // first pooled row
  const int cy0 = 2 * p0 - 1;                    // first conv row (may be -1)
  const int ylo = 2 * cy0 - 3;                   // input row of slab row 0

  // A slab is fetched into registers one frame ahead (global loads in flight during the MFMA loop) and committed to its
  // ring slot after that frame's conv: the fetch latency is off the per-frame critical path.  The fetch is branch-free:
  // one item = 4 consecutive columns of one slab row as ONE load of the RAW elements (a dword of bytes / 8 B of 16-bit /
  // 16 B of fp32; the byte rows of a decoder frame start anywhere, gfx950 takes the unaligned dword), 506 items = 2 per
  // thread, rows clamped into the frame.  Every conversion waits for the commit: a conditional load with its conversion
  // behind it compiles to one `s_waitcnt vmcnt(0)` per element - ten serialised global round trips per frame.  The ring's
  // padding columns and the rows outside the image are zeroed once and never written.
  constexpr int QPR = SW / 4;                    // quads per row (22)
  constexpr int SLAB_ITEMS = FIR * QPR;          // 506
  constexpr int NIT = (SLAB_ITEMS + 255) / 256;  // per thread (2)
  constexpr int RW = XK == 2 ? 1 : (XK == 0 ? 2 : 4);   // dwords per item
  typedef uint32_t RawSlab[NIT][RW];
  RawSlab raw;
  int ioff[NIT];                                 // ring element offset of the item inside a slot, -1: row outside the image
  uint32_t goff[NIT];                            // element offset of the item inside a frame
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int idx = tid + i * 256;
    const int idc = idx < SLAB_ITEMS ? idx : SLAB_ITEMS - 1;
    const int row = idc / QPR, k = idc - row * QPR;
    const int gy = ylo + row;
    const bool rin = idx < SLAB_ITEMS && gy >= 0 && gy < SH;
    const int gyc = gy < 0 ? 0 : (gy > SH - 1 ? SH - 1 : gy);
    ioff[i] = rin ? row * SCOLS + 4 * k + 3 : -1;
    goff[i] = XK == 2 ? (uint32_t)((gyc + u8.dy) * u8.Win + u8.dx + 4 * k) : (uint32_t)(gyc * SW + 4 * k);
  }
  auto fetch_into = [&](int tt, RawSlab& raw) {  // frame tt -> registers
    const int tc = tt < 0 ? 0 : (tt >= T ? T - 1 : tt);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      if (XK == 2) {
        const uint8_t* fb = (const uint8_t*)xin + ((int64_t)b * T + tc) * u8.Hin * u8.Win;
        __builtin_memcpy(&raw[i][0], fb + goff[i], 4);
      } else if (XK == 0) {
        const uint16_t* fb = (const uint16_t*)xin + ((int64_t)b * T + tc) * (SH * SW);
        const uint2 v = *reinterpret_cast<const uint2*>(fb + goff[i]);
        raw[i][0] = v.x; raw[i][RW > 1 ? 1 : 0] = v.y;
      } else {
        const float* fb = (const float*)xin + ((int64_t)b * T + tc) * (SH * SW);
        const uint4 v = *reinterpret_cast<const uint4*>(fb + goff[i]);
        raw[i][0] = v.x; raw[i][RW > 1 ? 1 : 0] = v.y; raw[i][RW > 2 ? 2 : 0] = v.z; raw[i][RW > 3 ? 3 : 0] = v.w;
      }
    }
  };
  auto commit_from = [&](int tt, const RawSlab& raw) {   // registers -> ring slot tt mod 5 (zeros outside the clip)
    const int slot = ((tt % 5) + 5) % 5;
    const bool tin = (tt >= 0) && (tt < T);
    uint16_t* dst = ring + slot * (FIR * SCOLS);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      uint32_t p01, p23;                         // columns 4k, 4k+1 | 4k+2, 4k+3 as packed 16-bit
      if (XK == 2) {
        // (x / 255.0f - mean) * inv_std in l2s_preprocess_frames' arithmetic: q = x * (1/255) refined by one fma step is the
        // correctly rounded quotient for every x in 0..255 (checked exhaustively), 3 VALU ops instead of a division's 12
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x = (float)((raw[i][0] >> (8 * j)) & 255u), rc = 1.0f / 255.0f;
          float q = x * rc;
          q = __builtin_fmaf(__builtin_fmaf(-q, 255.0f, x), rc, q);
          v[j] = (q - u8.mean) * u8.inv_std;
        }
        p01 = ET::pack2(v[0], v[1]); p23 = ET::pack2(v[2], v[3]);
      } else if (XK == 0) {
        p01 = raw[i][0]; p23 = raw[i][RW > 1 ? 1 : 0];
      } else {
        p01 = ET::pack2(__uint_as_float(raw[i][0]), __uint_as_float(raw[i][RW > 1 ? 1 : 0]));
        p23 = ET::pack2(__uint_as_float(raw[i][RW > 2 ? 2 : 0]), __uint_as_float(raw[i][RW > 3 ? 3 : 0]));
      }
      if (!tin) { p01 = 0; p23 = 0; }
      if (ioff[i] >= 0) {                        // column x lives at element x + 3: 2 + 4 + 2 bytes
        uint16_t* d = dst + ioff[i];
        d[0] = (uint16_t)p01;
        *reinterpret_cast<uint32_t*>(d + 1) = (p01 >> 16) | (p23 << 16);
        d[3] = (uint16_t)(p23 >> 16);
      }
    }
  };
  auto fetch_slab = [&](int tt) { fetch_into(tt, raw); };
  auto commit_slab = [&](int tt) { commit_from(tt, raw); };
  for (int i = tid; i < 5 * FIR * SCOLS / 8; i += 256) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  {
    const uint32_t ninf = ET::kDtype == L2S_F16 ? 0xFC00FC00u : 0xFF80FF80u;     // -inf: the pooling window's padding
    for (int i = tid; i < FCR * CW * CPITCH / 8; i += 256) reinterpret_cast<uint4*>(cbuf)[i] = make_uint4(ninf, ninf, ninf, ninf);
  }
  __syncthreads();
  {                                              // window of the first frame: five slabs in flight together
    RawSlab first[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) fetch_into(t_begin - 2 + f, first[f]);
#pragma unroll
    for (int f = 0; f < 5; ++f) commit_from(t_begin - 2 + f, first[f]);
  }

  frag16 wf[4][SKS];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks)
      wf[ni][ks].u = *reinterpret_cast<const uint4*>(w + (ni * 16 + lm) * (SKS * 32) + ks * 32 + lg * 8);
  if (tid < 64) { sbs[tid] = bias[tid]; sbs[64 + tid] = SWISH ? 0.f : slope[tid]; }   // read per tile in the epilogue: 32 VGPRs freed
  // PReLU with slopes >= 0 is non-decreasing and commutes with the max: the conv tile is then pooled raw and the activation
  // applied to the pooled quarter (a trained stem looks like that; any negative slope, and Swish, take the exact order)
  bool mono = false;
  if (!SWISH) mono = __builtin_amdgcn_ballot_w64(slope[lane] >= 0.f) == ~0ull;
  // the weights are "used" here: the compiler's vmcnt bookkeeping would otherwise wait for them at their first MFMA of
  // every frame, i.e. behind that frame's slab loads, which are meant to stay in flight during the conv
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks) { f16x8_t t = wf[ni][ks].h; asm volatile("" : "+v"(t)); wf[ni][ks].h = t; }

  // conv rows of this group inside the 44-row map (the first group's row -1 and the last group's rows 44.. are never pooled)
  const int r0 = cy0 < 0 ? -cy0 : 0;
  const int r1 = SHO - cy0 < FCR ? SHO - cy0 : FCR;
  const int NPIX = (r1 - r0) * SWO;              // 352 / 396 / 220
  const int NTILES = (NPIX + 15) >> 4;           // 22 / 25 / 14
  const int NFULL = NTILES & ~3;                 // whole rounds of the four waves; the 1-2 tiles left are split by channel
  const int wv = __builtin_amdgcn_readfirstlane(wave);
#ifdef L2S_STEM_STAMPS
  unsigned long long st_acc[5] = {0, 0, 0, 0, 0};
#define STAMP(i, expr) { const unsigned long long t0_ = __builtin_amdgcn_s_memtime(); expr; st_acc[i] += __builtin_amdgcn_s_memtime() - t0_; }
#else
#define STAMP(i, expr) { expr; }
#endif
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) fetch_slab(t + 3);        // in flight during this frame's conv, committed behind it
 STAMP(0, __syncthreads());                             // slab landed; previous frame's pooling finished reading cbuf
#ifdef L2S_STEM_STAMPS
    const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
    int koff[SKS];                               // LDS element offset of (dt,dy) for this lane's k-chunk, ring-aware
#pragma unroll
    for (int ks = 0; ks < SKS; ++ks) {
      int q = ks * 4 + lg;
      q = q > 34 ? 34 : q;
      const int dt = q / 7, dy = q - dt * 7;
      const int slot = (((t + dt - 2) % 5) + 5) % 5;
      koff[ks] = (slot * FIR + dy) * SCOLS;
    }
    // Fragments run through a 3-deep register ring, two k-steps ahead of the MFMAs and across tile boundaries (the
    // LDS latency of a k-step is ~2x its four MFMAs; with the weights in 144 VGPRs a deeper ring does not fit).
    auto tile_pix = [&](int tl, bool& pv, int& oyl, int& ox) {
      const int p = tl * 16 + lm;
      pv = p < NPIX;
      const int pp = pv ? p : NPIX - 1;
      const int row = pp / SWO;
      ox = pp - row * SWO;
      oyl = r0 + row;
      return (2 * oyl) * SCOLS + 2 * ox;
    };
    auto ld_frag = [&](frag16& f, int abase, int ks) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(ring + abase + koff[ks]);
      f.u = make_uint4(src[0], src[1], src[2], src[3]);
    };
    auto put = [&](int ni, const f32x4_t& a, int q) {    // q: pixel slot (row * CW + column + 1)
      float v[4];
      if (mono) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = a[r];
      } else {
        const f32x4_t slv = *reinterpret_cast<const f32x4_t*>(sbs + 64 + ni * 16 + lg * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = SWISH ? l2s_swish(a[r]) : (a[r] >= 0.f ? a[r] : a[r] * slv[r]);
      }
      *reinterpret_cast<uint2*>(cbuf + q * CPITCH + ni * 16 + lg * 4) = make_uint2(ET::pack2(v[0], v[1]), ET::pack2(v[2], v[3]));
    };
    bool pv; int oyl, ox;
    const int t_first = wv < NFULL ? wv : NFULL;              // NFULL >= 12: every wave has whole tiles
    int abase = tile_pix(t_first, pv, oyl, ox);
    frag16 fr[3];
    ld_frag(fr[0], abase, 0);
    ld_frag(fr[1], abase, 1);
    for (int tl = wv; tl < NFULL; tl += 4) {
      bool pv_n; int oyl_n, ox_n;
      // next: this wave's next whole tile, else the first split tile, else (past the end) a harmless re-read of this one
      const int tn = tl + 4 < NFULL ? tl + 4 : (NFULL < NTILES ? NFULL : tl);
      const int abase_n = tile_pix(tn, pv_n, oyl_n, ox_n);
      // the accumulators start from the bias (read while the first fragments are in flight)
      f32x4_t acc[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[ni] = *reinterpret_cast<const f32x4_t*>(sbs + ni * 16 + lg * 4);
#pragma unroll
      for (int ks = 0; ks < SKS; ++ks) {
        if (ks + 2 < SKS) ld_frag(fr[(ks + 2) % 3], abase, ks + 2);
        else ld_frag(fr[(ks + 2) % 3], abase_n, ks + 2 - SKS);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = ET::mfma(wf[ni][ks], fr[ks % 3], acc[ni]);
      }
#ifdef L2S_STEM_STAMPS
      const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
      if (pv) {
        const int q = oyl * CW + ox + 1;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) put(ni, acc[ni], q);
      }
      abase = abase_n; pv = pv_n; oyl = oyl_n; ox = ox_n;
#ifdef L2S_STEM_STAMPS
      st_acc[4] += __builtin_amdgcn_s_memtime() - te0;
#endif
    }
    // the tiles beyond the whole rounds: every wave takes its own 16 channels of each (a quarter of the MFMAs), so no
    // wave runs a whole tile longer than the others before the barrier
    auto split_tiles = [&](auto NIc) {
      constexpr int NI = decltype(NIc)::value;
      for (int tl = NFULL; tl < NTILES; ++tl) {
        bool pv_n; int oyl_n, ox_n;
        const int abase_n = tile_pix(tl + 1 < NTILES ? tl + 1 : tl, pv_n, oyl_n, ox_n);
        f32x4_t a = *reinterpret_cast<const f32x4_t*>(sbs + NI * 16 + lg * 4);
#pragma unroll
        for (int ks = 0; ks < SKS; ++ks) {
          if (ks + 2 < SKS) ld_frag(fr[(ks + 2) % 3], abase, ks + 2);
          else ld_frag(fr[(ks + 2) % 3], abase_n, ks + 2 - SKS);
          a = ET::mfma(wf[NI][ks], fr[ks % 3], a);
        }
        if (pv) put(NI, a, oyl * CW + ox + 1);
        abase = abase_n; pv = pv_n; oyl = oyl_n; ox = ox_n;
      }
    };
    if (NFULL < NTILES) {
      if (wv == 0) split_tiles(IntC<0>{});
      else if (wv == 1) split_tiles(IntC<1>{});
      else if (wv == 2) split_tiles(IntC<2>{});
      else split_tiles(IntC<3>{});
    }
#ifdef L2S_STEM_STAMPS
    st_acc[1] += __builtin_amdgcn_s_memtime() - tc0;
    const unsigned long long tp0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                             // conv tile complete, ring slot of frame t-2 free
#ifdef L2S_STEM_STAMPS
    st_acc[2] += __builtin_amdgcn_s_memtime() - tp0;
    const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
    // the next frame's newest slab goes in BEFORE the pooling: its loads have landed under the conv, and the wait for
    // them would otherwise also wait for the pooled rows' stores (one counter)
    if (t + 1 < t_end) commit_slab(t + 3);
    // ---- 3x3 / stride-2 max pool out of cbuf: item = (pooled pixel, 8-channel chunk) ----
    // Window of pooled pixel (pyl, px): local rows 2 pyl .. +2, slots 2 px .. +2 (slot 0 = the -inf halo): nine reads at
    // immediate offsets from one address, no clamping.  f16 maxima are taken packed (v_pk_max_f16, exact).
    for (int it = tid; it < FP * 22 * 8; it += 256) {
      const int ch = it & 7, pix = it >> 3;
      const int pyl = pix / 22, px = pix - pyl * 22;
      const int py = p0 + pyl;
      if (py >= 22) continue;
      const uint16_t* win = cbuf + ((2 * pyl) * CW + 2 * px) * CPITCH + ch * 8;
      frag16 o;
      if constexpr (ET::kDtype == L2S_F16) {
        union { uint4 u; uint32_t w[4]; } m, f[9];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) f[tap].u = *reinterpret_cast<const uint4*>(win + ((tap / 3) * CW + tap % 3) * CPITCH);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t a01 = pk_max_f16(f[0].w[j], f[1].w[j]), a23 = pk_max_f16(f[2].w[j], f[3].w[j]);
          const uint32_t a45 = pk_max_f16(f[4].w[j], f[5].w[j]), a67 = pk_max_f16(f[6].w[j], f[7].w[j]);
          m.w[j] = pk_max_f16(pk_max_f16(pk_max_f16(a01, a23), pk_max_f16(a45, a67)), f[8].w[j]);
        }
        o.u = m.u;
        if (mono) {
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sbs + 64 + ch * 8);
          const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(sbs + 64 + ch * 8 + 4);
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            v[j] = ET::to_f32(o.s[j]);
            const float sj = j < 4 ? s0[j & 3] : s1[j & 3];
            v[j] = v[j] >= 0.f ? v[j] : v[j] * sj;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) m.w[j] = ET::pack2(v[2 * j], v[2 * j + 1]);
          o.u = m.u;
        }
      } else {
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          frag16 f;
          f.u = *reinterpret_cast<const uint4*>(win + ((tap / 3) * CW + tap % 3) * CPITCH);
#pragma unroll
          for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], ET::to_f32(f.s[j]));
        }
        if (mono) {
          const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(sbs + 64 + ch * 8);
          const f32x4_t s1 = *reinterpret_cast<const f32x4_t*>(sbs + 64 + ch * 8 + 4);
#pragma unroll
          for (int j = 0; j < 8; ++j) m[j] = m[j] >= 0.f ? m[j] : m[j] * (j < 4 ? s0[j & 3] : s1[j & 3]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o.s[j] = ET::from_f32(m[j]);
      }
      *reinterpret_cast<uint4*>(y + ((((int64_t)b * T + t) * 22 + py) * 22 + px) * 64 + ch * 8) = o.u;
    }
#ifdef L2S_STEM_STAMPS
    st_acc[3] += __builtin_amdgcn_s_memtime() - tq0;
#endif
  }
#ifdef L2S_STEM_STAMPS
  if (lane == 0 && g_stem_stamps) {
    unsigned long long* o = g_stem_stamps + ((int64_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + wave * 4;
    o[0] = st_acc[0]; o[1] = st_acc[1]; o[2] = st_acc[4]; o[3] = st_acc[3];
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
template <typename ET>
__global__ void maxpool_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int N, int H, int W, int C,
                               int Ho, int Wo) {
  const int c8 = C / 8;
  const int64_t total = (int64_t)N * Ho * Wo * c8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    int64_t r = i / c8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int64_t n = r / Ho;
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox - 1 + dx;
        if (ix < 0 || ix >= W) continue;
        frag16 f;
        f.u = *reinterpret_cast<const uint4*>(x + ((n * H + iy) * W + ix) * C + cc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], ET::to_f32(f.s[j]));
      }
    }
    frag16 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.s[j] = ET::from_f32(m[j]);
    *reinterpret_cast<uint4*>(y + ((n * Ho + oy) * Wo + ox) * C + cc * 8) = o.u;
  }
}

template <typename ET>
__global__ void avgpool_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int N, int HW, int C) {
  const int c8 = C / 8;
  const int64_t total = (int64_t)N * c8;
  const float inv = 1.0f / (float)HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    const int64_t n = i / c8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < HW; ++p) {
      frag16 f;
      f.u = *reinterpret_cast<const uint4*>(x + (n * HW + p) * C + cc * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += ET::to_f32(f.s[j]);
    }
    frag16 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.s[j] = ET::from_f32(s[j] * inv);
    *reinterpret_cast<uint4*>(y + n * C + cc * 8) = o.u;
  }
}

template <typename ET>
__global__ void preprocess_kernel(const uint8_t* __restrict__ f, uint16_t* __restrict__ y, int64_t nframes, int Hin,
                                  int Win, int crop, float mean, float inv_std) {
  const int dy = (Hin - crop) / 2, dx = (Win - crop) / 2;  // utils.py:90-91: int(round(h - th) / 2.) truncates
  const int64_t total = nframes * crop * crop;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int xx = (int)(i % crop);
    const int yy = (int)((i / crop) % crop);
    const int64_t n = i / ((int64_t)crop * crop);
    const float v = (float)f[(n * Hin + yy + dy) * Win + xx + dx];
    y[i] = ET::from_f32((v / 255.0f - mean) * inv_std);
  }
}

inline int grid_for(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int l2s_stem_conv3d(const void* x, int x_is_f32, const void* w, const float* bias, const float* slope,
                               void* y, int B, int T, int H, int W, int dtype, void* stream) {
  if (!x || !w || !bias || !y) return L2S_EINVAL;   // slope == NULL selects Swish (ESPnet Conv3dResNet) instead of PReLU
  if (B <= 0 || T <= 0) return L2S_ESHAPE;
  if (H != SH || W != SW) return L2S_EUNSUPPORTED;  // image_crop_size = 88 (hubert_pretraining.py config)
  if (((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return L2S_EALIGN;
  dim3 grid(2, B * T);
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* wp = (const uint16_t*)w;
  uint16_t* yp = (uint16_t*)y;
  if (dtype == L2S_F16) {
    if (x_is_f32) { if (slope) hipLaunchKernelGGL((stem_kernel<ElemF16, true, false>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); else hipLaunchKernelGGL((stem_kernel<ElemF16, true, true>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); }
    else { if (slope) hipLaunchKernelGGL((stem_kernel<ElemF16, false, false>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); else hipLaunchKernelGGL((stem_kernel<ElemF16, false, true>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); }
  } else if (dtype == L2S_BF16) {
    if (x_is_f32) { if (slope) hipLaunchKernelGGL((stem_kernel<ElemBF16, true, false>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); else hipLaunchKernelGGL((stem_kernel<ElemBF16, true, true>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); }
    else { if (slope) hipLaunchKernelGGL((stem_kernel<ElemBF16, false, false>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); else hipLaunchKernelGGL((stem_kernel<ElemBF16, false, true>), grid, dim3(256), 0, st, x, wp, bias, slope, yp, B, T); }
  } else {
    return L2S_EINVAL;
  }
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

template <typename ET, int XK>
static void launch_stem_pool(dim3 grid, hipStream_t st, const void* x, const uint16_t* w, const float* bias,
                             const float* slope, uint16_t* y, int B, int T, int ft, StemU8 u8) {
  if (slope) hipLaunchKernelGGL((stem_pool_kernel<ET, XK, false>), grid, dim3(256), 0, st, x, w, bias, slope, y, B, T, ft, u8);
  else hipLaunchKernelGGL((stem_pool_kernel<ET, XK, true>), grid, dim3(256), 0, st, x, w, bias, slope, y, B, T, ft, u8);
}

static int stem_pool_dispatch(const void* x, int xk, const void* w, const float* bias, const float* slope, void* y, int B,
                              int T, int dtype, StemU8 u8, void* stream) {
  if (!x || !w || !bias || !y) return L2S_EINVAL;   // slope == NULL selects Swish (ESPnet Conv3dResNet) instead of PReLU
  if (B <= 0 || T <= 0) return L2S_ESHAPE;
  if (((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return L2S_EALIGN;
  if (xk != 2 && ((uintptr_t)x & 15)) return L2S_EALIGN;   // 16-bit / fp32 frames are fetched as 8- / 16-byte quads (uint8: any address)
  // frames per block: every block pays the weight fetch, the LDS initialisation and a five-slab window once; longer chunks
  // amortise that as long as the grid still holds several rounds of the 512 resident blocks
  static const int ft_env = [] { const char* e = getenv("L2S_STEM_FT"); return e ? atoi(e) : 0; }();   // A/B switch
  int ft = ft_env > 0 ? ft_env : (6L * ((T + 24) / 25) * B >= 2048 ? 25 : FT);
  dim3 grid((22 + FP - 1) / FP, (T + ft - 1) / ft, B);
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* wp = (const uint16_t*)w;
  uint16_t* yp = (uint16_t*)y;
  if (dtype == L2S_F16) {
    if (xk == 2) launch_stem_pool<ElemF16, 2>(grid, st, x, wp, bias, slope, yp, B, T, ft, u8);
    else if (xk == 1) launch_stem_pool<ElemF16, 1>(grid, st, x, wp, bias, slope, yp, B, T, ft, u8);
    else launch_stem_pool<ElemF16, 0>(grid, st, x, wp, bias, slope, yp, B, T, ft, u8);
  } else if (dtype == L2S_BF16) {
    if (xk == 2) launch_stem_pool<ElemBF16, 2>(grid, st, x, wp, bias, slope, yp, B, T, ft, u8);
    else if (xk == 1) launch_stem_pool<ElemBF16, 1>(grid, st, x, wp, bias, slope, yp, B, T, ft, u8);
    else launch_stem_pool<ElemBF16, 0>(grid, st, x, wp, bias, slope, yp, B, T, ft, u8);
  } else {
    return L2S_EINVAL;
  }
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_stem_pool_fused(const void* x, int x_is_f32, const void* w, const float* bias, const float* slope,
                                   void* y, int B, int T, int H, int W, int dtype, void* stream) {
  if (H != SH || W != SW) return L2S_EUNSUPPORTED;
  return stem_pool_dispatch(x, x_is_f32 ? 1 : 0, w, bias, slope, y, B, T, dtype, StemU8{0, 0, 0, 0, 0.f, 0.f}, stream);
}

extern "C" int l2s_stem_pool_fused_u8(const uint8_t* frames, int Hin, int Win, int crop, float mean, float std,
                                      const void* w, const float* bias, const float* slope, void* y, int B, int T,
                                      int dtype, void* stream) {
  if (crop != SH) return L2S_EUNSUPPORTED;          // image_crop_size = 88
  if (Hin < crop || Win < crop || std == 0.f) return L2S_ESHAPE;
  // utils.py:90-91: int(round(h - th) / 2.) truncates
  return stem_pool_dispatch(frames, 2, w, bias, slope, y, B, T, dtype,
                            StemU8{Hin, Win, (Hin - crop) / 2, (Win - crop) / 2, mean, 1.0f / std}, stream);
}

extern "C" int l2s_maxpool2d_3x3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return L2S_ESHAPE;
  if (C & 7) return L2S_EALIGN;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == L2S_F16)
    hipLaunchKernelGGL((maxpool_kernel<ElemF16>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const uint16_t*)x,
                       (uint16_t*)y, N, H, W, C, Ho, Wo);
  else if (dtype == L2S_BF16)
    hipLaunchKernelGGL((maxpool_kernel<ElemBF16>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const uint16_t*)x,
                       (uint16_t*)y, N, H, W, C, Ho, Wo);
  else return L2S_EINVAL;
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_avgpool_hw(const void* x, void* y, int N, int HW, int C, int dtype, void* stream) {
  if (!x || !y) return L2S_EINVAL;
  if (N <= 0 || HW <= 0 || C <= 0) return L2S_ESHAPE;
  if (C & 7) return L2S_EALIGN;
  const int64_t total = (int64_t)N * (C / 8);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == L2S_F16)
    hipLaunchKernelGGL((avgpool_kernel<ElemF16>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const uint16_t*)x,
                       (uint16_t*)y, N, HW, C);
  else if (dtype == L2S_BF16)
    hipLaunchKernelGGL((avgpool_kernel<ElemBF16>), dim3(grid_for(total, 256)), dim3(256), 0, st, (const uint16_t*)x,
                       (uint16_t*)y, N, HW, C);
  else return L2S_EINVAL;
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" int l2s_preprocess_frames(const uint8_t* frames, void* y, int B, int T, int Hin, int Win, int crop,
                                     float mean, float std, int dtype, void* stream) {
  if (!frames || !y) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || crop <= 0 || Hin < crop || Win < crop || std == 0.f) return L2S_ESHAPE;
  const int64_t nf = (int64_t)B * T;
  const int64_t total = nf * crop * crop;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == L2S_F16)
    hipLaunchKernelGGL((preprocess_kernel<ElemF16>), dim3(grid_for(total, 256)), dim3(256), 0, st, frames,
                       (uint16_t*)y, nf, Hin, Win, crop, mean, 1.0f / std);
  else if (dtype == L2S_BF16)
    hipLaunchKernelGGL((preprocess_kernel<ElemBF16>), dim3(grid_for(total, 256)), dim3(256), 0, st, frames,
                       (uint16_t*)y, nf, Hin, Win, crop, mean, 1.0f / std);
  else return L2S_EINVAL;
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

#ifdef L2S_STEM_STAMPS
extern "C" int l2s_debug_stem_stamps(void* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stem_stamps), &buf, sizeof(buf));
}
#endif
