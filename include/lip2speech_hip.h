/*
 * lip2speech_hip.h — C ABI of liblip2speech_hip.so (gfx950 / MI355X only).
 *
 * Drop-in boundary for the lip->speech inference hot path of
 * DomhnallBoyle/lip2speech-unit.  The reference has no FFI of its own (every op
 * is a torch.nn call); each entry point below names the reference call site(s)
 * (path:line relative to the reference repo) whose arithmetic it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless
 *    the comment says host; `stream` is a hipStream_t passed as void*.
 *  - nothing allocates, frees or synchronises: all calls are asynchronous on
 *    `stream` and are safe to capture into a hipGraph.
 *  - activations are channels-last ("rows x channels"): row = (clip, time) or
 *    (frame, y, x); 16-bit storage (fp16 or bf16 selected by `dtype`), fp32
 *    accumulation; the fp32 residual stream of the transformers is fp32.
 *  - return value: 0 on success, negative L2S_E* on a rejected argument
 *    (nothing launched), positive = hipError_t of a failed launch.
 */
#ifndef LIP2SPEECH_HIP_H
#define LIP2SPEECH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define L2S_ABI_VERSION 14

/* element type of 16-bit operands */
enum { L2S_F16 = 0, L2S_BF16 = 1 };

/* error codes */
enum { L2S_OK = 0, L2S_EINVAL = -1, L2S_ESHAPE = -2, L2S_EALIGN = -3, L2S_EUNSUPPORTED = -4 };

/* epilogue activations */
enum {
  L2S_ACT_NONE = 0,
  L2S_ACT_RELU = 1,   /* espnet positionwise_feed_forward.py:30 */
  L2S_ACT_GELU = 2,   /* erf GELU: fairseq gelu, model_avhubert.py:234, models_multi_input.py:41 */
  L2S_ACT_SWISH = 3,  /* espnet convolution.py:68-73 */
  L2S_ACT_PRELU = 4,  /* per-channel slope, avhubert/resnet.py:47-48 */
  L2S_ACT_LRELU = 5,  /* speech-resynthesis/models.py:36-38,101,110 */
  L2S_ACT_TANH = 6    /* speech-resynthesis/models.py:112 */
};

/* epilogue flags */
enum {
  L2S_F_RES_PRE  = 1 << 0, /* v += R before the activation (BasicBlock: resnet.py:71-72) */
  L2S_F_RES_POST = 1 << 1, /* v += R after the activation (x + f(x) residuals) */
  L2S_F_ACCUM    = 1 << 2, /* v += previous contents of C (sum of ResBlocks, models.py:105-108) */
  L2S_F_DUAL     = 1 << 3, /* also store C2 = leaky_relu(v, slope2) as 16-bit */
  L2S_F_MASK     = 1 << 4, /* zero output rows whose time index >= lens[clip]*mask_mul */
  L2S_F_OUT_F32  = 1 << 5, /* C is fp32 (else 16-bit `dtype`) */
  L2S_F_RES_F32  = 1 << 6  /* R is fp32 (else 16-bit `dtype`) */
};

/* A-row addressing modes of the tap-GEMM */
enum {
  L2S_MODE_LINEAR = 0, /* src_row = m                                   (nn.Linear, 1x1 conv) */
  L2S_MODE_CONV1D = 1, /* rows = (clip, t): src_t = t*stride + tap*dil + off   (Conv1d / one phase of ConvTranspose1d) */
  L2S_MODE_CONV2D = 2  /* rows = (img, y, x): iy = y*stride + ky - pad, ix likewise (Conv2d) */
};

/*
 * Tap-GEMM: C[o(m), n] = epi( sum_{tap<ntaps} sum_{c<Cin} A[src(m,tap), c] * W[n, tap*Cin + c] )
 * with out-of-range source rows reading as zero (= the conv's zero padding).
 * One kernel family serves every dense contraction of the path:
 *   nn.Linear            avhubert/hubert.py:327,727 ; fairseq q/k/v/out_proj, fc1, fc2 (hubert.py:739) ;
 *                        espnet attention.py:50-53,257 ; positionwise_feed_forward.py:28-30 ;
 *                        model_avhubert.py:258,273,285 ; models_multi_input.py:70,80
 *   nn.Conv2d 3x3 / 1x1  avhubert/resnet.py:15-24,61-74 (BatchNorm folded into W and bias)
 *   nn.Conv1d            fairseq pos_conv (hubert.py:399, groups=16 via `groups`) ; model_avhubert.py:231-241 ;
 *                        espnet convolution.py:26-45 (pointwise) ; speech-resynthesis/models.py:19-31,78-79
 *   nn.ConvTranspose1d   speech-resynthesis/models.py:84-86, models_multi_input.py:40 — one call per output phase
 */
typedef struct l2s_gemm_desc {
  const void* A;      /* [rows_in, lda] 16-bit */
  const void* W;      /* [N, Ktot] 16-bit, Ktot = ntaps*Cin, K contiguous */
  void* C;            /* [rows_out, ldc] 16-bit or fp32 (L2S_F_OUT_F32) */
  void* C2;           /* [rows_out, ldc2] 16-bit, L2S_F_DUAL only */
  const float* bias;  /* [N] fp32 or NULL */
  const float* slope; /* [N] fp32 PReLU slopes (L2S_ACT_PRELU) or NULL */
  const void* R;      /* residual [rows_out, ldr] or NULL */
  const int32_t* lens;/* [clips] valid base-time units per clip (L2S_F_MASK) */
  int32_t M, N, Cin, ntaps;
  int32_t lda, ldc, ldc2, ldr;
  int32_t mode;
  /* CONV1D */
  int32_t T_out, T_in, stride, dil, off;
  /* CONV2D */
  int32_t Ho, Wo, Hi, Wi, KW, pad;
  /* output row remap: o = m*out_row_mul + out_row_add (ConvTranspose1d phases) */
  int32_t out_row_mul, out_row_add;
  /* row mask: clip = o / mask_T, t = o % mask_T, valid iff t < lens[clip]*mask_mul */
  int32_t mask_T, mask_mul;
  int32_t act, flags, dtype;
  float alpha;        /* v = alpha*(acc + bias) */
  float act_slope;    /* L2S_ACT_LRELU slope */
  float slope2;       /* leaky slope of the C2 copy */
  /* grouped conv (blockIdx.z): per-group element offsets added to A cols, W, C cols, bias */
  int32_t groups, a_gstride, c_gstride;
  int64_t w_gstride;
  /* K-block table (L2S_MODE_LINEAR only; NULL = none): the launch is `groups` problems over the same A [M, lda] and W [N, Ktot =
   * ntaps*Cin].  Problem g computes C[:, g*c_gstride + n] = epilogue(sum over j < nblk of A[:, a_off[j] .. + Cin) . W[n, w_off[j] ..
   * + Cin)) - a K that is a LIST of Cin-wide column blocks.  With an H x W map stored as ONE row of A ([images, H*W*Cin]) this is a
   * convolution on a small map in which output position g sums only the taps that fall inside the map (avhubert/resnet.py:15-24
   * conv3x3 with padding 1 on the 6 x 6 / 3 x 3 maps of layer3 / layer4: 21 / 40 % of the taps are padding).
   * ktab: device int32 [groups][2 + 2*L2S_KTAB_MAX] = {nblk, 0, a_off[L2S_KTAB_MAX], w_off[L2S_KTAB_MAX]}, element offsets that
   * are multiples of 8; bias / slope are indexed [g*N + n] (as for grouped convolutions); R is addressed like C.  Served by the
   * phase-staggered kernel for the bias + linear-activation families with or without a 16-bit residual; else L2S_EUNSUPPORTED. */
  const int32_t* ktab;
} l2s_gemm_desc;
#define L2S_KTAB_MAX 9

int l2s_abi_version(void);
const char* l2s_build_info(void);

int l2s_tapgemm(const l2s_gemm_desc* host_desc, void* stream);
/* block tile the launcher picks for this descriptor, as BM*1000+BN (profiling aid: names the kernel instantiation);
 * 256256 = the phase-staggered kernel, 999064 / 999128 = the patch conv kernel at 64 / 128 channels */
int l2s_tapgemm_variant(const l2s_gemm_desc* host_desc);
/* epilogue family (0..9) of the kernel instantiation the launcher picks: every kernel is built once per family of
 * (flags, activation) so that a launch carries one epilogue's code only (profiling aid, names the instantiation) */
int l2s_tapgemm_epilogue_family(const l2s_gemm_desc* host_desc);

/*
 * Stem: Conv3d(1->64,k(5,7,7),s(1,2,2),p(2,3,3)) + BatchNorm3d(eval, folded) + PReLU
 * avhubert/resnet.py:137-140.  x: [B,T,88,88] (fp32 when x_is_f32 else 16-bit); w: [64, 288] 16-bit packed
 * (k = (dt*7+dy)*8+dx, dx==7 zero); y: [B*T, Ho, Wo, 64] 16-bit channels-last.  Frames t >= lens[b] read as zero.
 * slope: [64] fp32 PReLU slopes (zeros = ReLU); NULL selects Swish, the stem of ESPnet's Conv3dResNet
 * (espnet/nets/pytorch_backend/backbones/conv3d_extractor.py:57-78, used by the `multi_target` model, model.py:184-228).
 */
int l2s_stem_conv3d(const void* x, int x_is_f32, const void* w, const float* bias, const float* slope,
                    void* y, int B, int T, int H, int W, int dtype, void* stream);

/*
 * Fused stem + pool: Conv3d + BatchNorm3d + PReLU + MaxPool3d(k(1,3,3),s(1,2,2),p(0,1,1)), avhubert/resnet.py:137-141, in one
 * launch: frame window in an LDS ring, conv tile pooled out of LDS, the [44,44,64] conv activation never reaches HBM.
 * Same x / w / bias / slope as l2s_stem_conv3d (x 16-byte aligned); y: [B*T, 22, 22, 64] 16-bit channels-last.  With all
 * slopes >= 0 the kernel pools the raw conv tile and applies the PReLU to the pooled values (PReLU is then non-decreasing and
 * commutes with the maximum); any negative slope, and Swish, keep the activation in front of the pool as written in the
 * reference.  DEVIATION of the all-slopes->= 0 path from l2s_stem_conv3d + l2s_maxpool2d_3x3s2 (and from the reference's
 * order, fp32 PReLU then ONE rounding): the conv value is rounded to 16 bits before the pool and the activated value again
 * after it, so NEGATIVE outputs are round16(s * round16(a)) instead of round16(s * a) - one 16-bit ulp at most from that;
 * the fused kernel also sums the taps in another order than l2s_stem_conv3d (different tiles), worth one more ulp on a small
 * share of the outputs of either sign (tests/test_kernels_gpu.py::test_stem_pool_fused_vs_two_step_launches: <= 2 ulp, < 2 %
 * of the outputs differ, both dtypes).
 */
int l2s_stem_pool_fused(const void* x, int x_is_f32, const void* w, const float* bias, const float* slope,
                        void* y, int B, int T, int H, int W, int dtype, void* stream);

/*
 * The same fused stem, fed by the raw decoder output: frames [B,T,Hin,Win] uint8 grayscale.  The centre crop to
 * crop x crop (= 88) and the (x/255 - mean)/std normalisation (avhubert/hubert_dataset.py:242-245, utils.py:56-95:
 * CenterCrop offsets truncate) are applied while the frame window is staged into LDS, in l2s_preprocess_frames'
 * arithmetic (results are bit-identical to l2s_preprocess_frames followed by l2s_stem_pool_fused); the normalised
 * frames never exist in HBM.
 */
int l2s_stem_pool_fused_u8(const uint8_t* frames, int Hin, int Win, int crop, float mean, float std, const void* w,
                           const float* bias, const float* slope, void* y, int B, int T, int dtype, void* stream);

/* MaxPool3d(k(1,3,3),s(1,2,2),p(0,1,1)) on channels-last frames, avhubert/resnet.py:141.  x:[N,H,W,C] -> y:[N,Ho,Wo,C] */
int l2s_maxpool2d_3x3s2(const void* x, void* y, int N, int H, int W, int C, int dtype, void* stream);

/* AdaptiveAvgPool2d(1) over HW, avhubert/resnet.py:127-128.  x:[N,HW,C] 16-bit -> y:[N,C] 16-bit */
int l2s_avgpool_hw(const void* x, void* y, int N, int HW, int C, int dtype, void* stream);

/*
 * LayerNorm over the last dim with fp32 statistics (wavefront reduction).
 *   fairseq LayerNorm eps 1e-5: hubert.py:400,720 and every transformer layer ; espnet layer_norm.py:12-33 eps 1e-12.
 * x: [M, ldx] (fp32 if x_is_f32 else 16-bit); y: [M, ldy] (fp32 if y_is_f32 else 16-bit), C columns.
 * zero_prefix > 0 reproduces hubert.py:706-720 (video-only fuse): the normalised vector is [zeros(zero_prefix) ‖ x]
 * of width zero_prefix + C; gamma/beta then have zero_prefix + C entries and y gets zero_prefix + C columns.
 * y2 (optional, 16-bit, ldy2) receives a second copy (used to feed the mel head concat buffer).
 * lens (optional): rows with (row % mask_T) >= lens[row / mask_T]*len_mul are written as zero (the zero padding a
 * clip run alone would see in the temporal convs that follow, SURVEY.md section 7 "Batching semantics").
 */
int l2s_layernorm(const void* x, int x_is_f32, int ldx, const float* gamma, const float* beta, float eps,
                  void* y, int y_is_f32, int ldy, void* y2, int ldy2, int M, int C, int zero_prefix,
                  const int32_t* lens, int len_mul, int mask_T, int dtype, void* stream);

/*
 * Fused multi-head self-attention with key-padding mask, fp32 online softmax.
 *   fairseq MultiheadAttention (hubert.py:739-743; q pre-scaled by d^-0.5 at pack time) when pos == NULL;
 *   espnet RelPositionMultiHeadedAttention (attention.py:240-280,59-90) when pos != NULL:
 *   score[i,j] = (q_i+u).k_j + (q_i+v).P[i-j], P = pos[(T-1)-(i-j)] (rel_shift folded into the index).
 * qkv: [B*T, ldq] 16-bit with q at col 0, k at col H*64, v at col 2*H*64 (head h at +h*64); out: [B*T, ldo] 16-bit.
 * pos: [2T-1, ldp] 16-bit (row k <-> relative position T-1-k), head h at col h*64; bias_u/bias_v: [H,64] fp32.
 * lens: [B] valid keys per clip scaled by len_mul (keys >= lens[b]*len_mul get zero probability).
 */
int l2s_attention(const void* qkv, int ldq, void* out, int ldo, const void* pos, int ldp,
                  const float* bias_u, const float* bias_v, const int32_t* lens, int len_mul,
                  int B, int T, int H, int dtype, void* stream);

/*
 * Conformer conv-module core: GLU(dim=C) -> depthwise Conv1d(k, pad (k-1)/2, groups=C) -> BatchNorm1d(eval, folded)
 * -> Swish.  espnet convolution.py:57-62.  x: [B*T, 2C] 16-bit (pointwise_cov1 output); w: [k, C] fp32 (BN folded);
 * bias: [C] fp32; y: [B*T, C] 16-bit.  Rows t >= lens[b]*len_mul are treated as zero padding and written as zero.
 */
int l2s_glu_dwconv_swish(const void* x, const float* w, const float* bias, void* y, const int32_t* lens,
                         int len_mul, int B, int T, int C, int k, int dtype, void* stream);

/*
 * Greedy unit decode == hypothesis 0 of the reference beam search (multi_target_lip2speech/sequence_generator.py:235-494):
 * per step t < 2*src_len: token = argmax over ids [4, V) of logits/temperature (pad,bos,eos,unk excluded :274-282),
 * lprob = log_softmax over all V ; step t == 2*src_len emits EOS (id 2) with lprob 0 (:286-298).
 * logits: [B*T2, ldl] fp32; tokens: [B, T2+1] int32 (pad id 1 after EOS); lprobs: [B, T2+1] fp32; score: [B] fp32
 * = sum(lprobs)/(L+1)^lenpen (avhubert/sequence_generator.py:650-651).
 */
int l2s_greedy_decode(const float* logits, int ldl, const int32_t* lens, int len_mul, int B, int T2, int V,
                      float temperature, float lenpen, int32_t* tokens, float* lprobs, float* score, void* stream);

/*
 * n-best unit decode: the reference's beam search itself (multi_target_lip2speech/sequence_generator.py:235-494 with fairseq
 * BeamSearch.step at :337-343 and finalize_hypos avhubert/sequence_generator.py:605-721), one wavefront per clip.  Step scores are
 * history-independent (:253-256), so the search is an exact top-`beam` over sequences; hypothesis 0 equals l2s_greedy_decode.
 * beam <= 64, V <= 256.  tokens / pos_scores: [B, beam, T2+1] (hypotheses in finalisation order = descending score; EOS at
 * position L, pad id 1 / 0.0 behind it); score: [B, beam] = cumulative score / (L+1)^lenpen; nhyp: [B] hypotheses produced
 * (beam, or 1 for an empty clip).  workspace: l2s_beam_decode_workspace(B, T2, beam) bytes of device memory.
 */
int l2s_beam_decode(const float* logits, int ldl, const int32_t* lens, int len_mul, int B, int T2, int V, float temperature,
                    float lenpen, int beam, void* workspace, size_t workspace_bytes, int32_t* tokens, float* pos_scores,
                    float* score, int32_t* nhyp, void* stream);
size_t l2s_beam_decode_workspace(int B, int T2, int beam);

/* time-major frame duplication x2 (sequence_generator.py:130-131) fused with a cast: x:[B*T, C] fp32 -> y:[B*2T, C] 16-bit */
int l2s_repeat2_cast(const float* x, void* y, int B, int T, int C, int dtype, void* stream);

/*
 * Split-K tail of a small-M residual-stream Linear (the one-clip-per-request path, multi_target_lip2speech/inference.py:161):
 * with M of 100-500 rows a 64 x 64-tile GEMM fills 32 of the 256 CUs for K / 64 serial K-tiles.  The host then runs the layer
 * as ONE grouped l2s_tapgemm launch (groups = S slices of K: a_gstride = K / S columns of A, the weight pre-packed as
 * [S][N][K / S], c_gstride = N, fp32 partial products P [M, S*N], bias in slice 0 only) and this kernel folds them into the
 * fp32 residual stream: x[m, n] += sum_{s < S} P[m, s*N + n], s ascending (deterministic; fairseq fc2 / out_proj + residual,
 * hubert.py:739; espnet positionwise_feed_forward.py:30 + encoder_layer.py:95).
 */
int l2s_splitk_reduce(const float* P, int ldp, int S, float* x, int ldx, int M, int N, void* stream);

/* l2s_splitk_reduce followed by l2s_layernorm of the updated stream, in one launch (C = 1024 or 512; the pre-LN encoder / conformer
 * layers put a LayerNorm behind every residual update: hubert.py:739-743, encoder_layer.py:89-141): x[m] += sum_s P[m, s*C ..],
 * y[m] = LayerNorm(x[m]) (16-bit, or fp32 when y_is_f32; y == x in fp32 = in place, the un-normalised sum is then not stored).
 * The stream update is bit-identical to l2s_splitk_reduce's, the LayerNorm equal to l2s_layernorm's to fp32 rounding. */
int l2s_splitk_reduce_layernorm(const float* P, int ldp, int S, float* x, int ldx, const float* gamma, const float* beta, float eps,
                                void* y, int y_is_f32, int ldy, int M, int C, const int32_t* lens, int len_mul, int mask_T,
                                int dtype, void* stream);

/* generic cast / layout helpers */
int l2s_cast_f32_to_16(const float* x, int ldx, void* y, int ldy, int M, int C, int dtype, void* stream);
int l2s_cast_16_to_f32(const void* x, int ldx, float* y, int ldy, int M, int C, int dtype, void* stream);
/* y[b*T + t, col0 + c] = v[b, c] for t < T (speaker-embedding time tiling: model_avhubert.py:269, models.py:158-177);
 * rows t >= lens[b]*len_mul are zeroed when lens != NULL */
int l2s_broadcast_rows(const void* v, int ldv, void* y, int ldy, int col0, const int32_t* lens, int len_mul,
                       int B, int T, int C, int v_is_f32, int dtype, void* stream);
/* y[b*T + t, col0 + c] = x[b, c, t] (fp32 [B,C,T] mel -> channels-last 16-bit; models_multi_input.py:65,73) */
int l2s_transpose_ct_to_tc(const float* x, void* y, int ldy, int col0, const int32_t* lens, int len_mul,
                           int B, int C, int T, int dtype, void* stream);
/* y[b*L + l, :] = table[code[b,l], :] (nn.Embedding, models_multi_input.py:67); rows l >= lens[b] zeroed */
int l2s_embedding(const int32_t* code, const void* table, void* y, int ldy, const int32_t* lens,
                  int B, int L, int C, int dtype, void* stream);
/*
 * In-memory stage 1 -> stage 2 hand-off (replaces the file round trip multi_target_lip2speech/inference.py:267-274 ->
 * create_dataset.py:366-428 -> multi_input_vocoder/dataset_multi_input.py:41-110,198-291), all on the caller's stream:
 *  l2s_embedding_tokens: the generator's token rows (tok[b*ldt + l], fairseq ids: unit u is token u + token_offset, 4 specials
 *    first) -> nn.Embedding rows y[b*L + l, :] = table[clamp(tok - token_offset, 0, n_rows-1), :]; rows l >= lens[b]*len_mul zero.
 *  l2s_rows_f32_to_16_masked: time-major fp32 rows x[(b*T + t)*ldx + c] (the mel head's output, model_avhubert.py:276) ->
 *    16-bit y[(b*T + t)*ldy + col0 + c] (the vocoder's concat buffer, models_multi_input.py:65,73); rows t >= lens[b]*len_mul zero.
 *  l2s_lens_from_mask: src_lengths of sequence_generator.py:64-65: lens[b] = T - sum_t mask[b,t] (mask: bool bytes, True = pad;
 *    NULL = no padding).
 */
int l2s_embedding_tokens(const int32_t* tok, int ldt, int token_offset, const void* table, int n_rows, void* y, int ldy,
                         const int32_t* lens, int len_mul, int B, int L, int C, int dtype, void* stream);
int l2s_rows_f32_to_16_masked(const float* x, int ldx, void* y, int ldy, int col0, const int32_t* lens, int len_mul,
                              int B, int T, int C, int dtype, void* stream);
int l2s_lens_from_mask(const uint8_t* mask, int32_t* lens, int B, int T, void* stream);
/*
 * Reference-precision switch of the vocoder (the reference's MelCodeGenerator runs in fp32, multi_input_vocoder/inference.py:
 * 73-82, speech-resynthesis/models.py:98-114): y = act(x) for rows t < lens[b]*len_mul (else 0), written as hi = 16-bit(y) and
 * lo = 16-bit(y - hi).  act: 0 none, 1 leaky_relu(slope), 2 GELU (erf).  x: fp32 [B*T, ldx]; hi, lo: 16-bit [B*T, ld16].
 * A layer is then three l2s_tapgemm launches into one fp32 output (a_hi w_hi, then a_lo w_hi and a_hi w_lo with L2S_F_ACCUM).
 */
int l2s_split_hi_lo(const float* x, int ldx, void* hi, void* lo, int ld16, int act, float slope, const int32_t* lens,
                    int len_mul, int B, int T, int C, int dtype, void* stream);

/*
 * Fused convolution PAIR of ResBlock1 for the wide vocoder stages (speech-resynthesis/models.py:34-41, one (c1, c2, d) step):
 *   x' = c2(leaky_relu(c1(leaky_relu(x)))) + x,  C in {64, 128}, k odd <= 11, (k-1)/2 * dil <= 32.
 * The pair's input and output travel as their LeakyReLU'd 16-bit copies only (x is recovered by the inverse of leaky_relu);
 * the intermediate never leaves LDS.  X: [B*T, C] = leaky_relu(x), rows t >= lens[b]*len_mul zero; W1 / W2: [C][k*C] 16-bit,
 * K = tap*C + c (W1 applied with dilation dil, W2 with dilation 1, both "same" padded); b1 / b2: [C] fp32.
 * last == 0: Y [B*T, C] = leaky_relu(x').   last != 0 (the third pair of a ResBlock): XS [B*T, C] fp32 = x' (+ XS when
 * accumulate: the sum over the stage's ResBlocks, models.py:103-108) and, when Y is given, Y = leaky_relu(XS).
 * last == 2 (the stage's final pair when only Y travels on: `x = xs / num_kernels` feeds leaky_relu + ups, models.py:109,101):
 * XS is read for the sum but NOT written back - Y = leaky_relu(XS + x') is the only output.
 * Rows at or past the clip length are written as zero.
 */
typedef struct l2s_respair_desc {
  const void* X; const void* W1; const void* W2; const float* b1; const float* b2;
  void* Y; float* XS; const int32_t* lens;
  int32_t len_mul, B, T, C, k, dil, last, accumulate, dtype;
  float slope;
} l2s_respair_desc;
int l2s_respair(const l2s_respair_desc* d, void* stream);

/*
 * The LAST conv pairs of a stage's n <= 3 ResBlocks in one launch (speech-resynthesis/models.py:103-109 over :34-41):
 *   Y [B*T, C] = leaky_relu( sum_j ( c2_j(leaky_relu(c1_j(leaky_relu(x_j)))) + x_j ) ),  C in {64, 128, 256}
 * where X[j] = leaky_relu(x_j) is the input of ResBlock j's last (c1, c2, d) pair - what l2s_respair(last = 0) of its previous pair
 * wrote.  Same operand layouts and limits as l2s_respair (k odd <= 11, (k-1)/2 * dil <= 28).  The stage's fp32 sum is kept in
 * accumulators across the ResBlocks and never written: use it where only leaky_relu of the sum travels on (`x = xs / num_kernels`
 * feeds leaky_relu + ups, models.py:109,101 - every stage but the last); results equal n l2s_respair(last != 0) launches up to the
 * order of the fp32 additions.  Rows at or past the clip length are written as zero.
 */
typedef struct l2s_respair_final_desc {
  const void* X[3]; const void* W1[3]; const void* W2[3]; const float* b1[3]; const float* b2[3];
  int32_t k[3], dil[3];
  void* Y; const int32_t* lens;
  int32_t n, len_mul, B, T, C, dtype;
  float slope;
} l2s_respair_final_desc;
int l2s_respair_final(const l2s_respair_final_desc* d, void* stream);

/*
 * Fused BasicBlock of the lip frontend's 64-channel stage (avhubert/resnet.py:43-74 with :15-24 conv3x3, layer1 of ResNet-18;
 * eval BatchNorm folded into weight + bias):  y = prelu(conv2(prelu(conv1(x) + b1, s1)) + b2 + x, s2), both convolutions 3x3,
 * stride 1, padding 1, C -> C channels.  One block keeps one image in LDS: HBM sees x once and y once.
 * x, y: [n_images*H*W, C] 16-bit channels-last; w1, w2: [C][9*C] 16-bit, K index = (ky*3 + kx)*C + cin; b*, s*: fp32[C] (bias,
 * PReLU slope).  Supported: C = 64, (H+2)*(W+2) <= 576, W <= 29 (22 x 22 on the path: csrc/basicblock.hip), and C = 128 with H = W = 11
 * (layer2's second block: csrc/basicblock_phase.hip, two images per 256-row tile); anything else returns L2S_EUNSUPPORTED.
 */
int l2s_basicblock_fused(const void* x, const void* w1, const float* b1, const float* s1, const void* w2, const float* b2,
                         const float* s2, void* y, int n_images, int H, int W, int C, int dtype, void* stream);
/*
 * n_blocks (<= 4) BasicBlocks of that stage back to back on the LDS-resident image (avhubert/resnet.py:101-118 `_make_layer`,
 * layer1 = two blocks): the layer's input is read once, its output written once, the activations in between never leave the
 * CU.  Results are bit-identical to n_blocks l2s_basicblock_fused calls (the hand-over is the same 16-bit rounding).
 * w, bias, slope: HOST arrays of 2*n_blocks device pointers in the order conv1, conv2 of block 0, conv1, conv2 of block 1, ...
 */
int l2s_basiclayer_fused(const void* x, const void* const* w, const float* const* bias, const float* const* slope, int n_blocks,
                         void* y, int n_images, int H, int W, int C, int dtype, void* stream);
/*
 * The rest of ResNet-18's strided 128-channel stage behind its first convolution, in one launch (avhubert/resnet.py:61-74 for a
 * block WITH downsample, :101-118 `_make_layer`, layer2 on the path: 22 x 22 x 64 -> 11 x 11 x 128):
 *   out0 = prelu(conv3x3(t0) + ba + Wd x0[::2, ::2], sa)                  second conv of block 1, residual = the 1x1 stride-2 downsample
 *   y    = prelu(conv3x3(prelu(conv3x3(out0) + b1, s1)) + b2 + out0, s2)  block 2 (= l2s_basicblock_fused, C = 128)
 * x0: the stage input [n_images*(2H)*(2W), 64]; t0: block 1's first conv output [n_images*H*W, 128] (after bn1 + PReLU); both 16-bit
 * channels-last.  wa: [128][9*128 + 64 + 64] 16-bit = the 3x3 weights (K index (ky*3+kx)*128 + cin) | the downsample weights
 * [128][64] | 64 zero columns; ba = the conv's folded bias + the downsample's folded bias; w1, w2: [128][9*128]; b*, s*: fp32[128].
 * The downsample runs as one more K-tile of the same fp32 accumulation (it is not rounded to 16 bits on the way, unlike a separate
 * launch); out0 never leaves the CU.  Supported: H = W = 11 (l2s_basicblock_fused's C = 128 family); else L2S_EUNSUPPORTED.
 */
int l2s_basicstage128_tail_fused(const void* x0, const void* t0, const void* wa, const float* ba, const float* sa, const void* w1,
                                 const float* b1, const float* s1, const void* w2, const float* b2, const float* s2, void* y,
                                 int n_images, int H, int W, int dtype, void* stream);

/*
 * Vocoder tail: leaky_relu(x, 0.01) -> Conv1d(C->1, k7, p3) -> tanh -> *32768 -> int16 truncation.
 * speech-resynthesis/models.py:110-112 ; multi_input_vocoder/inference.py:79-81.
 * x: [B*T, C] fp32 (sum of the three ResBlocks; the /3 is folded into w); w: [k, C] fp32; wav: [B, T] fp32; pcm: [B, T] int16 or NULL.
 */
int l2s_conv_post_tanh(const float* x, const float* w, float bias, float* wav, int16_t* pcm, const int32_t* lens,
                       int len_mul, int B, int T, int C, int k, void* stream);

/*
 * Fused ResBlock1 (speech-resynthesis/models.py:16-47) for the narrow vocoder stages, C in {16,32}, k in {3,7,11}:
 * six convolutions run out of an LDS-resident time tile; HBM sees one read of xl = leaky_relu(x) and one accumulate of
 * the block output into xs (the sum over the stage's three ResBlocks, models.py:103-108).
 * xl: [B*T, C] 16-bit; w: [6][C][Kpad] 16-bit in the order c1(d0),c2,c1(d1),c2,c1(d2),c2 with K = tap*C + c zero-padded to
 * a multiple of 32; bias: [6][C] fp32; xs: [B*T, C] fp32 (overwritten, or added to when accumulate != 0);
 * xl_out (optional): [B*T, C] 16-bit = leaky_relu(xs after this call).  Rows t >= lens[b]*len_mul are zero.
 */
int l2s_resblock_fused(const void* xl, const void* w, const float* bias, float* xs, void* xl_out,
                       const int32_t* lens, int len_mul, int B, int T, int C, int k, int d0, int d1, int d2,
                       int accumulate, float slope, int dtype, void* stream);

/*
 * All ResBlocks of one narrow stage in one launch (speech-resynthesis/models.py:103-109, `xs = sum_j resblocks[i*3+j](x)`):
 * each block runs the n_blocks = 3 ResBlocks (resblock_kernel_sizes = [3, 7, 11]) on its time tile back to back, so xl
 * is read from HBM once; at C = 32 the fp32 running sum stays in registers between the ResBlocks, at C = 16 it passes through
 * xs (L2).  Results are those of three l2s_resblock_fused calls (accumulate = 0, 1, 1; xl_out on the last) in the kernel's
 * order, bit for bit: k = 11, 7, 3 at C = 32 (the widest body first: the sum's registers are live through the later ones)
 * and k = 3, 7, 11 at C = 16, i.e. the stage sum is (first + second) + third in fp32.
 * w[j]: [6][C][Kpad_j] 16-bit and bias[j]: [6][C] fp32 as in l2s_resblock_fused (w, bias, ks, dils are HOST arrays: of
 * device pointers, of kernel sizes, and of the n_blocks x 3 dilations); xs: [B*T, C] fp32; xl_out optional.
 * xs_final != 0: xs holds the stage's sum afterwards (conv_post reads it); xs_final == 0 (xl_out required): only
 * leaky_relu(sum) is written and xs is scratch (content afterwards unspecified; ABI 14 - up to ABI 13 it held the partial
 * sum after the second ResBlock).
 * Other stage layouts return L2S_EUNSUPPORTED (the caller launches the ResBlocks one by one).
 */
int l2s_resstage_fused(const void* xl, const void* const* w, const float* const* bias, const int* ks, const int* dils,
                       int n_blocks, float* xs, void* xl_out, const int32_t* lens, int len_mul, int B, int T, int C,
                       float slope, int xs_final, int dtype, void* stream);

/* frames: uint8 [B,T,Hin,Win] -> centre crop + (x/255-mean)/std, hubert_dataset.py:242-245, utils.py:56-95 -> 16-bit [B,T,crop,crop] */
int l2s_preprocess_frames(const uint8_t* frames, void* y, int B, int T, int Hin, int Win, int crop, float mean,
                          float std, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LIP2SPEECH_HIP_H */
