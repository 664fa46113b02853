// Fused multi-head self-attention (head dim 64) with key-padding mask and fp32 online softmax.
//   plain   : fairseq MultiheadAttention inside TransformerSentenceEncoderLayer (avhubert/hubert.py:739-743)
//   rel-pos : espnet RelPositionMultiHeadedAttention (attention.py:240-280) incl. rel_shift (:218-238) and
//             forward_attention (:59-90).  score[i,j] = (q_i+u).k_j + (q_i+v).P(i-j); the [B,H,T,2T-1] tensor is never built:
//             per (query tile, key tile) only the 127 relative positions it can touch are projected on MFMA and the
//             rel_shift becomes an index skew through a per-wave LDS scratch.
//
// gfx950 layout: block = 4 or 8 waves = 64 or 128 query rows of one (clip, head); wave = 16 query rows (the 128-row
// block stages every K/V tile once where two 64-row blocks would each fetch it: T = 100 and T = 200 use it).  All three products are
// issued "swapped" so that the query row lives on the lane (lane&15) for S^T = K.Q^T, BD^T = P.Qv^T and O^T = V^T.P^T:
// row max / row sum are 2 shuffles (lanes l, l^16, l^32 share a row), and the probabilities feed the PV MFMA as its
// B operand straight from registers with a permuted key order that the V^T LDS image mirrors.
#include "l2s_common.h"
#include <cstdlib>
#include <type_traits>

namespace {

constexpr int D = 64;        // head dim
constexpr int KB = 64;       // keys per tile
constexpr int VLD = 68;      // V^T row stride in elements (136 B: conflict-free ds_read_b64)

// LDS carve-up (dynamic: the 128-row rel-pos variant needs 81 KB)
template <int QB, bool RELPOS> struct AttnLds {
  static constexpr int PROWS = QB + 64;  // relative-position rows staged per (query tile, key tile)
  static constexpr int K_OFF = 0;
  static constexpr int V_OFF = K_OFF + KB * D * 2;
  static constexpr int P_OFF = V_OFF + ((D * VLD * 2 + 15) & ~15);
  static constexpr int BD_OFF = P_OFF + (RELPOS ? PROWS * D * 2 : 0);
  static constexpr int BYTES = BD_OFF + (RELPOS ? (QB / 16) * 80 * 16 * 4 : 0);
};

__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ (row & 7); }  // 16-byte chunk swizzle, 128-B rows

template <typename ET, bool RELPOS, int QB>
__global__ __launch_bounds__(QB * 4) void attention_kernel(const uint16_t* __restrict__ qkv, int ldq,
                                                        uint16_t* __restrict__ out, int ldo,
                                                        const uint16_t* __restrict__ pos, int ldp,
                                                        const float* __restrict__ bias_u,
                                                        const float* __restrict__ bias_v,
                                                        const int32_t* __restrict__ lens, int len_mul, int T, int H) {
  using L = AttnLds<QB, RELPOS>;
  constexpr int NT = QB * 4;  // threads
  extern __shared__ __attribute__((aligned(16))) unsigned char attn_lds[];
  uint16_t* sK = reinterpret_cast<uint16_t*>(attn_lds + L::K_OFF);
  uint16_t* sVt = reinterpret_cast<uint16_t*>(attn_lds + L::V_OFF);
  uint16_t* sP = reinterpret_cast<uint16_t*>(attn_lds + L::P_OFF);
  float* sBD = reinterpret_cast<float*>(attn_lds + L::BD_OFF);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, lg = lane >> 4;
  const int qt = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int i0 = qt * QB;            // first query row of the block
  const int iw0 = i0 + wave * 16;    // first query row of the wave
  int klen = T;
  if (lens) { klen = lens[b] * len_mul; klen = klen < T ? klen : T; }
  const int64_t rowbase = (int64_t)b * T;
  const int qcol = h * D, kcol = H * D + h * D, vcol = 2 * H * D + h * D;

  // ---- Q fragments (B operand: lane = query row lm, k-chunk lg) ----
  frag16 qu[2], qv[2];
  {
    const int qi = iw0 + lm;
    const bool ok = qi < T;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag16 q;
      q.u = ok ? *reinterpret_cast<const uint4*>(qkv + (rowbase + qi) * ldq + qcol + ks * 32 + lg * 8)
               : make_uint4(0, 0, 0, 0);
      if (RELPOS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = ET::to_f32(q.s[j]);
          const int d = ks * 32 + lg * 8 + j;
          qu[ks].s[j] = ET::from_f32(f + bias_u[h * D + d]);
          qv[ks].s[j] = ET::from_f32(f + bias_v[h * D + d]);
        }
      } else {
        qu[ks] = q;
      }
    }
  }

  f32x4_t acc_o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc_o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  const int nkt = (klen + KB - 1) / KB;
  for (int jt = 0; jt < nkt; ++jt) {
    const int j0 = jt * KB;
    __syncthreads();  // previous tile fully consumed
    // ---- stage K (row-major, swizzled) and V^T ----
#pragma unroll
    for (int it = 0; it < 512 / NT; ++it) {
      const int idx = tid + it * NT;       // 512 chunks of 16 B
      const int key = idx >> 3, ch = idx & 7;
      const int j = j0 + key;
      uint4 kv = make_uint4(0, 0, 0, 0);
      frag16 vv; vv.u = make_uint4(0, 0, 0, 0);
      if (j < T) {
        const uint16_t* rp = qkv + (rowbase + j) * ldq;
        kv = *reinterpret_cast<const uint4*>(rp + kcol + ch * 8);
        vv.u = *reinterpret_cast<const uint4*>(rp + vcol + ch * 8);
      }
      *reinterpret_cast<uint4*>(sK + key * D + kswz(key, ch) * 8) = kv;
#pragma unroll
      for (int e = 0; e < 8; ++e) sVt[(ch * 8 + e) * VLD + key] = vv.s[e];
    }
    if (RELPOS) {
      // row cb <-> relative position rel = i0 - j0 - 63 + cb <-> pos table row k = (T-1) - rel
#pragma unroll
      for (int it = 0; it < L::PROWS * 8 / NT; ++it) {
        const int idx = tid + it * NT;     // PROWS x 8 chunks
        const int cb = idx >> 3, ch = idx & 7;
        const int k = (T - 1) - (i0 - j0 - 63 + cb);
        uint4 pv = make_uint4(0, 0, 0, 0);
        if (k >= 0 && k < 2 * T - 1) pv = *reinterpret_cast<const uint4*>(pos + (int64_t)k * ldp + h * D + ch * 8);
        *reinterpret_cast<uint4*>(sP + cb * D + kswz(cb, ch) * 8) = pv;
      }
    }
    __syncthreads();

    // ---- S^T[key][q] = K . Qu^T ----
    f32x4_t s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        frag16 kf;
        const int row = kt * 16 + lm;
        kf.u = *reinterpret_cast<const uint4*>(sK + row * D + kswz(row, ks * 4 + lg) * 8);
        s[kt] = ET::mfma(kf, qu[ks], s[kt]);
      }
    }
    if (RELPOS) {
      // BD^T[c][q] for the 80 relative positions this wave's 16 rows can reach, then skew through LDS
      float* bd = sBD + wave * (80 * 16);
#pragma unroll
      for (int rt = 0; rt < 5; ++rt) {
        f32x4_t a = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          frag16 pf;
          const int row = wave * 16 + rt * 16 + lm;
          pf.u = *reinterpret_cast<const uint4*>(sP + row * D + kswz(row, ks * 4 + lg) * 8);
          a = ET::mfma(pf, qv[ks], a);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bd[(rt * 16 + lg * 4 + r) * 16 + lm] = a[r];
      }
      // same-wave LDS round trip: writes above are visible to this wave's reads after the wait
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jj = kt * 16 + lg * 4 + r;
          s[kt][r] += bd[(lm - jj + 63) * 16 + lm];
        }
    }

    // ---- mask + online softmax (row = lm; lanes lm, lm+16, lm+32, lm+48 share it) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + kt * 16 + lg * 4 + r;
        if (j >= klen) s[kt][r] = -INFINITY;
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __expf(m_run - m_use);  // m_run = -inf on the first tile -> 0
    float rs = 0.f;
    frag16 pf[2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = __expf(s[kt][r] - m_use);
        rs += pr;
        pf[kt >> 1].s[(kt & 1) * 4 + r] = ET::from_f32(pr);
      }
    rs += __shfl_xor(rs, 16, 64);
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc_o[dt][r] *= alpha;

    // ---- O^T[d][q] += V^T . P^T ; k-position 8*lg+e of step ks2 <-> key (2*ks2 + (e>>2))*16 + 4*lg + (e&3) ----
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int ks2 = 0; ks2 < 2; ++ks2) {
        frag16 vf;
        const uint16_t* vp = sVt + (dt * 16 + lm) * VLD + ks2 * 32 + lg * 4;
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
        vf.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
        acc_o[dt] = ET::mfma(vf, pf[ks2], acc_o[dt]);
      }
    }
  }

  // ---- normalise and store: lane holds O[q = iw0+lm][d = dt*16 + 4*lg .. +3] ----
  const int qi = iw0 + lm;
  if (qi < T) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    uint16_t* op = out + (rowbase + qi) * ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 q;
      q.x = (uint32_t)ET::from_f32(acc_o[dt][0] * inv) | ((uint32_t)ET::from_f32(acc_o[dt][1] * inv) << 16);
      q.y = (uint32_t)ET::from_f32(acc_o[dt][2] * inv) | ((uint32_t)ET::from_f32(acc_o[dt][3] * inv) << 16);
      *reinterpret_cast<uint2*>(op + dt * 16 + lg * 4) = q;
    }
  }
}

}  // namespace

extern "C" int l2s_attention(const void* qkv, int ldq, void* out, int ldo, const void* pos, int ldp,
                             const float* bias_u, const float* bias_v, const int32_t* lens, int len_mul, int B, int T,
                             int H, int dtype, void* stream) {
  if (!qkv || !out) return L2S_EINVAL;
  if (B <= 0 || T <= 0 || H <= 0) return L2S_ESHAPE;
  if ((ldq & 7) || (ldo & 3) || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return L2S_EALIGN;
  if (ldq < 3 * H * D || ldo < H * D) return L2S_ESHAPE;
  if (pos && (!bias_u || !bias_v || (ldp & 7) || ldp < H * D || ((uintptr_t)pos & 15))) return L2S_EINVAL;
  if (lens && len_mul <= 0) return L2S_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* q = (const uint16_t*)qkv;
  uint16_t* o = (uint16_t*)out;
  const uint16_t* pp = (const uint16_t*)pos;
  // 128-row blocks when they do not add padded query rows over 64-row blocks (or T is long enough not to care)
  static const int force_qb = [] { const char* e = getenv("L2S_ATTN_QB"); return e ? atoi(e) : 0; }();  // 64 / 128: tools
  const bool big = force_qb ? force_qb == 128 : ((((T + 63) / 64) % 2 == 0) || T >= 512);
  auto go = [&](auto et, auto rel, auto qb) -> int {
    using ET = decltype(et);
    constexpr bool R = decltype(rel)::value;
    constexpr int Q = decltype(qb)::value;
    auto* k = attention_kernel<ET, R, Q>;
    constexpr int lds = AttnLds<Q, R>::BYTES;
    static bool attr_set = false;
    if (!attr_set) {
      hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return (int)e;
      attr_set = true;
    }
    dim3 grid((T + Q - 1) / Q, H, B);
    hipLaunchKernelGGL(k, grid, dim3(Q * 4), lds, st, q, ldq, o, ldo, pp, ldp, bias_u, bias_v, lens, len_mul, T, H);
    return L2S_OK;
  };
  auto go_q = [&](auto et, auto rel) -> int {
    return big ? go(et, rel, std::integral_constant<int, 128>{}) : go(et, rel, std::integral_constant<int, 64>{});
  };
  int rc;
  if (dtype == L2S_F16) rc = pos ? go_q(ElemF16{}, std::true_type{}) : go_q(ElemF16{}, std::false_type{});
  else if (dtype == L2S_BF16) rc = pos ? go_q(ElemBF16{}, std::true_type{}) : go_q(ElemBF16{}, std::false_type{});
  else return L2S_EINVAL;
  if (rc != L2S_OK) return rc;
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}
