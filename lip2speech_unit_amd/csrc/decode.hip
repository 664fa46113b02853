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
                                                            int V, float inv_temp, int32_t* __restrict__ tokens,
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
      float x = row[v] * inv_temp;
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
      const float x = row[v] * inv_temp;
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

}  // namespace

extern "C" int l2s_greedy_decode(const float* logits, int ldl, const int32_t* lens, int len_mul, int B, int T2, int V,
                                 float temperature, float lenpen, int32_t* tokens, float* lprobs, float* score,
                                 void* stream) {
  if (!logits || !tokens || !lprobs || !score) return L2S_EINVAL;
  if (B <= 0 || T2 <= 0 || V <= 4 || ldl < V) return L2S_ESHAPE;
  if (temperature <= 0.f || (lens && len_mul <= 0)) return L2S_EINVAL;
  hipLaunchKernelGGL(greedy_decode_kernel, dim3((T2 + 1 + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, logits, ldl, lens,
                     len_mul, T2, V, 1.0f / temperature, tokens, lprobs);
  L2S_CHECK_LAUNCH();
  hipLaunchKernelGGL(decode_score_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, lprobs, lens, len_mul, T2, lenpen,
                     score);
  L2S_CHECK_LAUNCH();
  return L2S_OK;
}
