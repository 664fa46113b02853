// Fused multi-head self-attention (head dim 64) with key-padding mask and fp32 online softmax.
//   plain   : fairseq MultiheadAttention inside TransformerSentenceEncoderLayer (avhubert/hubert.py:739-743)
//   rel-pos : espnet RelPositionMultiHeadedAttention (attention.py:240-280) incl. rel_shift (:218-238) and
//             forward_attention (:59-90).  score[i,j] = (q_i+u).k_j + (q_i+v).P(i-j); the [B,H,T,2T-1] tensor is never built:
//             per (query tile, key tile) only the 127 relative positions it can touch are projected on MFMA and the
//             rel_shift becomes an index skew through a per-wave LDS scratch.
//
// gfx950 layout: block = 4 waves = 64 query rows of one (clip, head); wave = 16 query rows.  All three products are
// issued "swapped" so that the query row lives on the lane (lane&15) for S^T = K.Q^T, BD^T = P.Qv^T and O^T = V^T.P^T:
// row max / row sum are 2 shuffles (lanes l, l^16, l^32 share a row), and the probabilities feed the PV MFMA as its
// B operand straight from registers with a permuted key order that the V^T LDS image mirrors.
#include "l2s_common.h"

namespace {

constexpr int D = 64;        // head dim
constexpr int QB = 64;       // query rows per block
constexpr int KB = 64;       // keys per tile
constexpr int VLD = 68;      // V^T row stride in elements (136 B: conflict-free ds_read_b64)
constexpr int PROWS = 128;   // relative-position rows staged per (query tile, key tile)

__device__ __forceinline__ int kswz(int row, int chunk) { return chunk ^ (row & 7); }  // 16-byte chunk swizzle, 128-B rows

template <typename ET, bool RELPOS>
__global__ __launch_bounds__(256) void attention_kernel(const uint16_t* __restrict__ qkv, int ldq,
                                                        uint16_t* __restrict__ out, int ldo,
                                                        const uint16_t* __restrict__ pos, int ldp,
                                                        const float* __restrict__ bias_u,
                                                        const float* __restrict__ bias_v,
                                                        const int32_t* __restrict__ lens, int len_mul, int T, int H) {
  __shared__ __attribute__((aligned(16))) uint16_t sK[KB * D];
  __shared__ __attribute__((aligned(16))) uint16_t sVt[D * VLD];
  __shared__ __attribute__((aligned(16))) uint16_t sP[RELPOS ? PROWS * D : 8];
  __shared__ __attribute__((aligned(16))) float sBD[RELPOS ? 4 * 80 * 16 : 4];

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
    for (int it = 0; it < 2; ++it) {
      const int idx = tid + it * 256;      // 512 chunks of 16 B
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
      for (int it = 0; it < 4; ++it) {
        const int idx = tid + it * 256;    // 1024 chunks
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
  dim3 grid((T + QB - 1) / QB, H, B);
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* q = (const uint16_t*)qkv;
  uint16_t* o = (uint16_t*)out;
  const uint16_t* pp = (const uint16_t*)pos;
  if (dtype == L2S_F16) {
    if (pos) hipLaunchKernelGGL((attention_kernel<ElemF16, true>), grid, dim3(256), 0, st, q, ldq, o, ldo, pp, ldp, bias_u, bias_v, lens, len_mul, T, H);
    else hipLaunchKernelGGL((attention_kernel<ElemF16, false>), grid, dim3(256), 0, st, q, ldq, o, ldo, pp, ldp, bias_u, bias_v, lens, len_mul, T, H);
  } else if (dtype == L2S_BF16) {
    if (pos) hipLaunchKernelGGL((attention_kernel<ElemBF16, true>), grid, dim3(256), 0, st, q, ldq, o, ldo, pp, ldp, bias_u, bias_v, lens, len_mul, T, H);
    else hipLaunchKernelGGL((attention_kernel<ElemBF16, false>), grid, dim3(256), 0, st, q, ldq, o, ldo, pp, ldp, bias_u, bias_v, lens, len_mul, T, H);
  } else {
    return L2S_EINVAL;
  }
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}
