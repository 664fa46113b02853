// Greedy unit decode: hypothesis 0 of the reference's beam search in one launch.
// multi_target_lip2speech/sequence_generator.py:235-494 runs 2T+1 python steps of {div_(temperature), fp32 log_softmax,
// NaN->-inf, mask pad/bos/eos/unk, forced EOS at step >= target length, BeamSearch.step, finalize}.  Step scores do not
// depend on the history (non-autoregressive logits :253-256), so the best hypothesis is the per-step argmax over ids
// [4,V) with the log-softmax taken over all V; its score is sum(lprob)/(L+1)^lenpen (avhubert/sequence_generator.py:650-651).
#include "l2s_common.h"

namespace {

// one wavefront per (clip, step): argmax over ids [4,V) + log-softmax over all V
__global__ __launch_bounds__(256) void greedy_decode_kernel(const float* __restrict__ logits, int ldl,
                                                            const int32_t* __restrict__ lens, int len_mul, int T2,
                                                            int V, float temperature, int32_t* __restrict__ tokens,
                                                            float* __restrict__ lprobs) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wave;
  if (t > T2) return;
  int L = lens ? lens[b] * len_mul : T2;
  L = L < T2 ? L : T2;
  int tok = 1;      // pad
  float lp = 0.f;
  if (t < L) {
    const float* row = logits + ((int64_t)b * T2 + t) * ldl;
    float mx = -INFINITY, best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lane; v < V; v += 64) {
      float x = row[v] / temperature;   // :256 div_(temperature)
      if (x != x) x = -INFINITY;  // NaN never wins (:274)
      mx = fmaxf(mx, x);
      if (v >= 4 && (x > best || (x == best && v < bi))) { best = x; bi = v; }
    }
    mx = wave_max(mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    float se = 0.f;
    for (int v = lane; v < V; v += 64) {
      const float x = row[v] / temperature;
      se += (x == x) ? __expf(x - mx) : 0.f;
    }
    se = wave_sum(se);
    tok = bi;
    lp = best - mx - __logf(se);
  } else if (t == L) {
    tok = 2;  // eos, forced with lprob 0 (:286-298)
  }
  if (lane == 0) {
    tokens[(int64_t)b * (T2 + 1) + t] = tok;
    lprobs[(int64_t)b * (T2 + 1) + t] = lp;
  }
}

// hypothesis score = sum of positional scores / (L+1)^lenpen, summed in a fixed order (bitwise reproducible)
__global__ __launch_bounds__(64) void decode_score_kernel(const float* __restrict__ lprobs,
                                                          const int32_t* __restrict__ lens, int len_mul, int T2,
                                                          float lenpen, float* __restrict__ score) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int L = lens ? lens[b] * len_mul : T2;
  L = L < T2 ? L : T2;
  float s = 0.f;
  for (int t = lane; t <= L; t += 64) s += lprobs[(int64_t)b * (T2 + 1) + t];
  s = wave_sum(s);
  if (lane == 0) score[b] = s / powf((float)(L + 1), lenpen);
}

// wave arg-max: larger value wins, ties go to the smaller index
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o, 64);
    const int oi = __shfl_xor(i, o, 64);
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// The reference's whole beam search for one clip on one wavefront (n-best output; hypothesis 0 == greedy_decode_kernel).
// Candidate scores are S_beam + lprob_token with lprobs independent of the history, so per step the top `beam` of the
// beam x V candidates (fairseq BeamSearch.step: add the cumulative score, top-k over the flattened candidates; only beam 0 at
// step 0) is a k-way merge of the sorted beam scores with the sorted top-`beam` token lprobs: lane i keeps beam i, the
// token list lives one entry per lane, `beam` rounds of a wave arg-max pop the best head.  Back pointers, tokens and
// cumulative scores go to a per-clip workspace; at step L (forced EOS with lprob 0, :286-298) every live beam is
// finalised in score order (avhubert/sequence_generator.py:605-721) and traced back by its own lane.
__global__ __launch_bounds__(64) void beam_decode_kernel(const float* __restrict__ logits, int ldl,
                                                         const int32_t* __restrict__ lens, int len_mul, int T2, int V,
                                                         float temperature, float lenpen, int beam,
                                                         int16_t* __restrict__ wbp, int16_t* __restrict__ wtk,
                                                         float* __restrict__ wcum, int32_t* __restrict__ tokens,
                                                         float* __restrict__ pos, float* __restrict__ score,
                                                         int32_t* __restrict__ nhyp) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int L = lens ? lens[b] * len_mul : T2;
  L = L < T2 ? L : T2;
  const int64_t wbase = (int64_t)b * (T2 + 1) * beam;
  float S = 0.f;          // cumulative score of beam `lane`
  int nb = 1;             // live beams (1 before the first step)
  for (int s = 0; s < L; ++s) {
    const float* row = logits + ((int64_t)b * T2 + s) * ldl;
    float x[4], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int v = lane + 64 * r;
      x[r] = v < V ? row[v] / temperature : -INFINITY;
      if (x[r] != x[r]) x[r] = -INFINITY;
      mx = fmaxf(mx, x[r]);
    }
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) se += (lane + 64 * r < V) ? __expf(x[r] - mx) : 0.f;
    const float lse = __logf(wave_sum(se));
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = (lane + 64 * r >= 4 && lane + 64 * r < V) ? x[r] - mx - lse : -INFINITY;
    // sorted top-`beam` tokens: entry j on lane j
    float tval = -INFINITY;
    int tidx = 0;
    for (int j = 0; j < beam; ++j) {
      float bv = x[0];
      int bi = lane;
#pragma unroll
      for (int r = 1; r < 4; ++r)
        if (x[r] > bv) { bv = x[r]; bi = lane + 64 * r; }
      wave_argmax(bv, bi);
      if (lane == j) { tval = bv; tidx = bi; }
      if ((bi & 63) == lane) x[bi >> 6] = -INFINITY;   // taken
    }
    // k-way merge of beams x tokens
    int p = 0;
    float nS = -INFINITY;
    int nbp = 0, ntk = 1;
    for (int k = 0; k < beam; ++k) {
      const float head = __shfl(tval, p, 64);   // every lane executes the shuffle: a source lane must be active
      float c = lane < nb ? S + head : -INFINITY;
      int ci = lane;
      wave_argmax(c, ci);
      const int tk = __shfl(tidx, __shfl(p, ci, 64), 64);
      if (lane == k) { nS = c; nbp = ci; ntk = tk; }
      if (lane == ci) ++p;
    }
    S = nS;
    nb = beam;
    if (lane < beam) {
      wbp[wbase + (int64_t)s * beam + lane] = (int16_t)nbp;
      wtk[wbase + (int64_t)s * beam + lane] = (int16_t)ntk;
      wcum[wbase + (int64_t)s * beam + lane] = nS;
    }
  }
  if (lane == 0) nhyp[b] = nb;
  __threadfence();   // this wave's workspace stores are visible to its own trace-back loads (other lanes' entries)
  if (lane < beam) {
    int32_t* trow = tokens + ((int64_t)b * beam + lane) * (T2 + 1);
    float* prow = pos + ((int64_t)b * beam + lane) * (T2 + 1);
    for (int t = L + 1; t <= T2; ++t) { trow[t] = 1; prow[t] = 0.f; }
    if (lane < nb) {
      trow[L] = 2;                 // forced EOS, candidate score S + 0
      int cur = lane;
      float hi = S;
      for (int s = L - 1; s >= 0; --s) {
        const int64_t o = wbase + (int64_t)s * beam + cur;
        const float c = wcum[o];
        prow[s + 1] = hi - c;      // positional scores = differences of the cumulative scores (finalize_hypos :662-664)
        hi = c;
        trow[s] = wtk[o];
        cur = wbp[o];
      }
      prow[0] = hi;
      score[(int64_t)b * beam + lane] = S / powf((float)(L + 1), lenpen);
    } else {
      for (int t = 0; t <= L; ++t) { trow[t] = 1; prow[t] = 0.f; }
      score[(int64_t)b * beam + lane] = -INFINITY;
    }
  }
}

}  // namespace

extern "C" int l2s_beam_decode(const float* logits, int ldl, const int32_t* lens, int len_mul, int B, int T2, int V,
                               float temperature, float lenpen, int beam, void* workspace, size_t workspace_bytes,
                               int32_t* tokens, float* pos_scores, float* score, int32_t* nhyp, void* stream) {
  if (!logits || !tokens || !pos_scores || !score || !nhyp || !workspace) return L2S_EINVAL;
  if (B <= 0 || T2 <= 0 || V <= 4 || ldl < V) return L2S_ESHAPE;
  if (V > 256 || beam < 1 || beam > 64 || beam > V - 4) return L2S_EUNSUPPORTED;
  if (temperature <= 0.f || (lens && len_mul <= 0)) return L2S_EINVAL;
  const size_t n = (size_t)B * (T2 + 1) * beam;
  if (workspace_bytes < l2s_beam_decode_workspace(B, T2, beam) || ((uintptr_t)workspace & 3)) return L2S_ESHAPE;
  float* wcum = (float*)workspace;
  int16_t* wbp = (int16_t*)(wcum + n);
  int16_t* wtk = wbp + n;
  hipLaunchKernelGGL(beam_decode_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, ldl, lens, len_mul, T2, V,
                     temperature, lenpen, beam, wbp, wtk, wcum, tokens, pos_scores, score, nhyp);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}

extern "C" size_t l2s_beam_decode_workspace(int B, int T2, int beam) {
  if (B <= 0 || T2 <= 0 || beam <= 0) return 0;
  return (size_t)B * (T2 + 1) * beam * (sizeof(float) + 2 * sizeof(int16_t));
}

extern "C" int l2s_greedy_decode(const float* logits, int ldl, const int32_t* lens, int len_mul, int B, int T2, int V,
                                 float temperature, float lenpen, int32_t* tokens, float* lprobs, float* score,
                                 void* stream) {
  if (!logits || !tokens || !lprobs || !score) return L2S_EINVAL;
  if (B <= 0 || T2 <= 0 || V <= 4 || ldl < V) return L2S_ESHAPE;
  if (temperature <= 0.f || (lens && len_mul <= 0)) return L2S_EINVAL;
  hipLaunchKernelGGL(greedy_decode_kernel, dim3((T2 + 1 + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, logits, ldl, lens,
                     len_mul, T2, V, temperature, tokens, lprobs);
  L2S_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_score_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, lprobs, lens, len_mul, T2, lenpen,
                     score);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}
