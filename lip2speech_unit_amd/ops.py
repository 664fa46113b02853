"""Thin torch-tensor front end over the C ABI (include/lip2speech_hip.h).

torch is used for device memory and the current HIP stream only; every function launches hand-written gfx950 kernels
through liblip2speech_hip.so and raises if the library is missing or rejects the call.  No function here computes.
"""
import ctypes
import os
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LRELU, ACT_NONE, ACT_PRELU, ACT_RELU, ACT_SWISH, ACT_TANH, BF16, F16, F_ACCUM,  # noqa: F401
                   F_DUAL, F_MASK, F_OUT_F32, F_RES_F32, F_RES_POST, F_RES_PRE, MODE_CONV1D, MODE_CONV2D,
                   MODE_LINEAR, GemmDesc, L2SError, check)

_TORCH16 = {F16: torch.float16, BF16: torch.bfloat16}


class KernelProfiler:
    """Optional per-launch HIP-event timing (events recorded on the stream the kernels are launched on).
    Used by bench.py to find the dominant kernel and its achieved rate; never active on the product path by default."""

    def __init__(self, detail=False):
        self.records = []  # (key, flops, bytes, ev_start, ev_end, shape)
        self.detail = detail  # tap-GEMM keys also carry the problem shape (tools/kernel_table.py --detail)

    def summary(self):
        """Per kernel key: calls, ms, algorithmic flops / bytes, and the same split by problem shape (`shapes`: one kernel
        instantiation serves several GEMM shapes; a roofline of the pooled launches alone would average their intensities)."""
        torch.cuda.synchronize()
        agg = {}
        for key, fl, by, e0, e1, shape in self.records:
            a = agg.setdefault(key, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "shapes": {}})
            ms = e0.elapsed_time(e1)
            for t in (a, a["shapes"].setdefault(shape, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})):
                t["calls"] += 1
                t["ms"] += ms
                t["flops"] += fl
                t["bytes"] += by
        return agg


_profiler = None


def set_profiler(p):
    global _profiler
    _profiler = p


def _run(key, call, flops=0.0, nbytes=0.0, shape=None):
    if _profiler is None:
        return check(call(), key)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = call()
    e1.record()
    check(rc, key)
    _profiler.records.append((key, flops, nbytes, e0, e1, shape))


def torch_dtype(dtype: int) -> torch.dtype:
    return _TORCH16[dtype]


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float16:
        return F16
    if t == torch.bfloat16:
        return BF16
    raise L2SError(f"unsupported 16-bit dtype {t}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise L2SError("HIP ops need device tensors (there is no CPU path)")
    return t.data_ptr()


def _req(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise L2SError(f"{name}: expected a device tensor (there is no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise L2SError(f"{name}: expected {dtype}, got {t.dtype}")
    if t.dim() >= 2 and t.stride(-1) != 1:
        raise L2SError(f"{name}: innermost dim must be contiguous")
    return t


def tapgemm(A, W, C, *, M, N, Cin, ntaps=1, lda=None, ldc=None, bias=None, slope=None, R=None, ldr=None, C2=None,
            ldc2=None, lens=None, mode=MODE_LINEAR, T_out=0, T_in=0, stride=1, dil=1, off=0, Ho=0, Wo=0, Hi=0, Wi=0,
            KW=1, pad=0, out_row_mul=1, out_row_add=0, mask_T=0, mask_mul=1, act=ACT_NONE, flags=0, dtype=F16,
            alpha=1.0, act_slope=0.0, slope2=0.0, groups=1, a_gstride=0, c_gstride=0, w_gstride=0, ktab=None, kflops=None):
    """One tap-GEMM launch. A/W/C/... are torch device tensors (or tensor views whose data_ptr is the origin).
    ktab: int32 [groups, 2 + 2*9] K-block table (include/lip2speech_hip.h, l2s_gemm_desc::ktab); kflops: the FLOPs such a launch
    really does (the profiler's count; default: the dense M x N x K figure)."""
    lib = _lib.load()
    d = GemmDesc()
    d.A, d.W, d.C = _ptr(A), _ptr(W), _ptr(C)
    d.C2, d.bias, d.slope, d.R, d.lens = _ptr(C2), _ptr(bias), _ptr(slope), _ptr(R), _ptr(lens)
    if bias is not None and bias.dtype != torch.float32:
        raise L2SError("bias must be fp32")
    if lens is not None and lens.dtype != torch.int32:
        raise L2SError("lens must be int32")
    d.M, d.N, d.Cin, d.ntaps = M, N, Cin, ntaps
    d.lda = lda if lda is not None else Cin
    d.ldc = ldc if ldc is not None else N * groups
    d.ldc2 = ldc2 if ldc2 is not None else d.ldc
    d.ldr = ldr if ldr is not None else d.ldc
    d.mode = mode
    d.T_out, d.T_in, d.stride, d.dil, d.off = T_out, T_in, stride, dil, off
    d.Ho, d.Wo, d.Hi, d.Wi, d.KW, d.pad = Ho, Wo, Hi, Wi, KW, pad
    d.out_row_mul, d.out_row_add = out_row_mul, out_row_add
    d.mask_T, d.mask_mul = mask_T, mask_mul
    if C.dtype == torch.float32:
        flags |= F_OUT_F32
    if R is not None and R.dtype == torch.float32:
        flags |= F_RES_F32
    d.act, d.flags, d.dtype = act, flags, dtype
    d.alpha, d.act_slope, d.slope2 = alpha, act_slope, slope2
    d.groups, d.a_gstride, d.c_gstride, d.w_gstride = groups, a_gstride, c_gstride, w_gstride
    if ktab is not None:
        if ktab.dtype != torch.int32 or ktab.dim() != 2 or ktab.shape != (groups, 20) or not ktab.is_contiguous():
            raise L2SError("ktab: int32 [groups, 20], contiguous")
        d.ktab = _ptr(ktab)
    if _profiler is None:
        check(lib.l2s_tapgemm(ctypes.byref(d), _stream()), "l2s_tapgemm")
        return
    var = lib.l2s_tapgemm_variant(ctypes.byref(d))
    key = f"tapgemm<{'f16' if dtype == F16 else 'bf16'},{var // 1000}x{var % 1000},mode{mode}>"
    if _profiler is not None:  # one kernel instantiation per epilogue family: name it like rocprofv3 sees it
        fam = lib.l2s_tapgemm_epilogue_family(ctypes.byref(d))
        if var in (999064, 999128):  # patchconv.hip builds 5 of the 10 families; the others run on a superset
            fam = {0: 2, 1: 3, 2: 2, 3: 3, 6: 6, 7: 7}.get(fam, 9)
        key = key[:-1] + f",e{fam}>"
    if _profiler is not None and _profiler.detail:
        key += f" M{M} N{N} Cin{Cin} taps{ntaps} g{groups} fl{flags:#x} act{act} alpha{alpha:g}"
    ktot = Cin * ntaps
    esz = 4 if (flags & F_OUT_F32) else 2
    # algorithmic bytes: A once, W once, C written once; + the residual read, the second output, the accumulate read
    mn = float(M) * N * groups
    nbytes = 2.0 * M * ktot * groups / max(ntaps, 1) + 2.0 * N * ktot * groups + esz * mn
    if R is not None:
        nbytes += (4 if (flags & F_RES_F32) else 2) * mn
    if flags & F_DUAL:
        nbytes += 2 * mn
    if flags & F_ACCUM:
        nbytes += esz * mn
    _run(key, lambda: lib.l2s_tapgemm(ctypes.byref(d), _stream()), kflops if kflops is not None else 2.0 * M * N * ktot * groups, nbytes,
         shape=f"M{M} N{N * groups} K{ktot}" + (" ktab" if ktab is not None else ""))


def stem_conv3d(x, w, bias, slope, y, B, T, dtype):
    lib = _lib.load()
    _run("l2s_stem_conv3d", lambda: lib.l2s_stem_conv3d(_ptr(x), int(x.dtype == torch.float32), _ptr(w), _ptr(bias), _ptr(slope), _ptr(y), B, T,
                              x.shape[-2], x.shape[-1], dtype, _stream()))


def stem_pool_fused(x, w, bias, slope, y, B, T, dtype):
    lib = _lib.load()
    _run("l2s_stem_pool_fused", lambda: lib.l2s_stem_pool_fused(
        _ptr(x), int(x.dtype == torch.float32), _ptr(w), _ptr(bias), _ptr(slope), _ptr(y), B, T, x.shape[-2], x.shape[-1],
        dtype, _stream()), flops=2.0 * B * T * 44 * 44 * 64 * 245,
        nbytes=B * T * (88 * 88 * (4 if x.dtype == torch.float32 else 2) + 22 * 22 * 64 * 2))


def stem_pool_fused_u8(frames, w, bias, slope, y, B, T, dtype, crop=88, mean=0.421, std=0.165):
    """Fused stem on raw uint8 frames [B,T,Hin,Win]: crop + normalise in the slab fetch (hubert_dataset.py:242-245)."""
    lib = _lib.load()
    if frames.dtype != torch.uint8 or frames.dim() != 4:
        raise ValueError("frames: uint8 [B,T,Hin,Win]")
    Hin, Win = frames.shape[-2], frames.shape[-1]
    _run("l2s_stem_pool_fused", lambda: lib.l2s_stem_pool_fused_u8(
        _ptr(frames), Hin, Win, crop, mean, std, _ptr(w), _ptr(bias), _ptr(slope), _ptr(y), B, T, dtype, _stream()),
        flops=2.0 * B * T * 44 * 44 * 64 * 245, nbytes=B * T * (crop * crop + 22 * 22 * 64 * 2))


def maxpool2d_3x3s2(x, y, N, H, W, C, dtype):
    _run("l2s_maxpool2d_3x3s2", lambda: _lib.load().l2s_maxpool2d_3x3s2(_ptr(x), _ptr(y), N, H, W, C, dtype, _stream()))


def avgpool_hw(x, y, N, HW, C, dtype):
    _run("l2s_avgpool_hw", lambda: _lib.load().l2s_avgpool_hw(_ptr(x), _ptr(y), N, HW, C, dtype, _stream()))


def layernorm(x, gamma, beta, eps, y, *, M, C, ldx=None, ldy=None, y2=None, ldy2=0, zero_prefix=0, lens=None,
              len_mul=1, mask_T=0, dtype=F16):
    ldx = ldx if ldx is not None else C
    ldy = ldy if ldy is not None else C + zero_prefix
    _run("l2s_layernorm", lambda: _lib.load().l2s_layernorm(_ptr(x), int(x.dtype == torch.float32), ldx, _ptr(gamma), _ptr(beta), eps, _ptr(y),
                                    int(y.dtype == torch.float32), ldy, _ptr(y2), ldy2, M, C, zero_prefix,
                                    _ptr(lens), len_mul, mask_T, dtype, _stream()))


def attention(qkv, out, *, B, T, H, ldq=None, ldo=None, pos=None, ldp=0, bias_u=None, bias_v=None, lens=None,
              len_mul=1, dtype=F16):
    ldq = ldq if ldq is not None else 3 * H * 64
    ldo = ldo if ldo is not None else H * 64
    _run("l2s_attention", lambda: _lib.load().l2s_attention(_ptr(qkv), ldq, _ptr(out), ldo, _ptr(pos), ldp, _ptr(bias_u), _ptr(bias_v),
                                    _ptr(lens), len_mul, B, T, H, dtype, _stream()))


def glu_dwconv_swish(x, w, bias, y, *, B, T, C, k, lens=None, len_mul=1, dtype=F16):
    _run("l2s_glu_dwconv_swish", lambda: _lib.load().l2s_glu_dwconv_swish(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(lens), len_mul, B, T, C, k,
                                           dtype, _stream()))


def greedy_decode(logits, tokens, lprobs, score, *, B, T2, V, ldl=None, lens=None, len_mul=1, temperature=1.0,
                  lenpen=1.0):
    ldl = ldl if ldl is not None else V
    _run("l2s_greedy_decode", lambda: _lib.load().l2s_greedy_decode(_ptr(logits), ldl, _ptr(lens), len_mul, B, T2, V, temperature, lenpen,
                                        _ptr(tokens), _ptr(lprobs), _ptr(score), _stream()))


def beam_decode(logits, *, B, T2, V, beam, ldl=None, lens=None, len_mul=1, temperature=1.0, lenpen=1.0):
    """n-best decode (the reference's beam search, one wave per clip).  Returns device tensors (tokens int32 [B,beam,T2+1],
    positional scores fp32 [B,beam,T2+1], score fp32 [B,beam], nhyp int32 [B])."""
    lib = _lib.load()
    dev = logits.device
    ldl = ldl if ldl is not None else V
    nbytes = lib.l2s_beam_decode_workspace(B, T2, beam)
    if nbytes == 0:
        raise L2SError("l2s_beam_decode_workspace: bad shape")
    work = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    tokens = torch.empty(B, beam, T2 + 1, device=dev, dtype=torch.int32)
    pos = torch.empty(B, beam, T2 + 1, device=dev, dtype=torch.float32)
    score = torch.empty(B, beam, device=dev, dtype=torch.float32)
    nhyp = torch.empty(B, device=dev, dtype=torch.int32)
    _run("l2s_beam_decode", lambda: lib.l2s_beam_decode(_ptr(_req(logits, torch.float32, "logits")), ldl, _ptr(lens), len_mul, B, T2, V,
                                                          temperature, lenpen, beam, _ptr(work), nbytes, _ptr(tokens), _ptr(pos),
                                                          _ptr(score), _ptr(nhyp), _stream()))
    return tokens, pos, score, nhyp


def repeat2_cast(x, y, B, T, C, dtype):
    _run("l2s_repeat2_cast", lambda: _lib.load().l2s_repeat2_cast(_ptr(x), _ptr(y), B, T, C, dtype, _stream()))


def splitk_reduce(P, x, *, M, N, S, ldp=None, ldx=None):
    """x[m, n] += sum_{s < S} P[m, s*N + n] (fp32, s ascending): the tail of a split-K residual-stream Linear at small M."""
    _run("l2s_splitk_reduce", lambda: _lib.load().l2s_splitk_reduce(_ptr(P), ldp or S * N, S, _ptr(x), ldx or N, M, N, _stream()))


# Residual-stream Linears at small M (one clip per request: M = 100-500 rows).  A 64 x 64-tile GEMM then fills 32-64 of the 256
# CUs for K / 64 serial K-tiles; cut into S slices of K (one grouped launch, groups = S) it fills them all for K / (64 S), and
# l2s_splitk_reduce folds the fp32 partial products into the stream in a fixed order.
SPLITK_MAX_M = 512
_SPLITK_ON = os.environ.get("L2S_SPLITK", "1") != "0"       # A/B: 0 = the one-launch form at every M


def splitk_slices(M, N, K):
    """K slices for the residual-stream Linear x[M, N] += A[M, K] W^T (0 = one launch as before): the largest power of two <= 8
    that keeps >= 256 columns of K per slice.  M only switches the form on (<= SPLITK_MAX_M rows): the slice count - and with it
    the re-packed weight a layer caches - depends on the layer alone, so whatever batch first takes the split path builds the
    state every later one reads (pipeline.forward_device_u8_streams primes shared state with ONE clip on the caller's stream)."""
    if not _SPLITK_ON or M > SPLITK_MAX_M or K < 1024 or (N & 63):
        return 0
    S = 8
    while S > 1 and (K % (S * 64) or K // S < 256):
        S //= 2
    return S if S > 1 else 0


def splitk_reduce_layernorm(P, x, gamma, beta, eps, y, *, M, C, S, ldp=None, ldx=None, ldy=None, lens=None, len_mul=1, mask_T=0,
                            dtype=F16):
    """splitk_reduce(P, x) and layernorm(x) -> y in one launch (C = 1024 / 512; y may be x itself in fp32)."""
    _run("l2s_splitk_reduce_layernorm", lambda: _lib.load().l2s_splitk_reduce_layernorm(
        _ptr(P), ldp or S * C, S, _ptr(x), ldx or C, _ptr(gamma), _ptr(beta), eps, _ptr(y), int(y.dtype == torch.float32), ldy or C,
        M, C, _ptr(lens), len_mul, mask_T, dtype, _stream()))


def residual_linear(A, W, bias, x, *, M, N, K, dtype, alpha=1.0, cache=None, key=None, ln=None):
    """x (fp32 residual stream, in place) += alpha * (A W^T + bias).  cache / key: where the [S][N][K / S] repack of W and the
    slice-0-only bias of the split-K form are kept (a layer's dict of packed weights).
    ln = (gamma, beta, eps, y): the LayerNorm that follows the update, y = LayerNorm(x) (y may be x) - always applied; in the
    split-K form it rides in the reduction's launch."""
    S = splitk_slices(M, N, K)
    if not S:
        tapgemm(A, W, x, M=M, N=N, Cin=K, bias=bias, alpha=alpha, R=x, ldr=N, flags=F_RES_POST, dtype=dtype)
        if ln is not None:
            layernorm(x, ln[0], ln[1], ln[2], ln[3], M=M, C=N, dtype=dtype)
        return
    ck = (key, S)
    if cache is None or ck not in cache:
        ws = W.view(N, S, K // S).permute(1, 0, 2).contiguous()
        bs = torch.zeros(S * N, device=W.device, dtype=torch.float32)
        if bias is not None:
            bs[:N] = bias
        if cache is not None:
            cache[ck] = (ws, bs)
    else:
        ws, bs = cache[ck]
    P = torch.empty(M, S * N, device=x.device, dtype=torch.float32)
    tapgemm(A, ws, P, M=M, N=N, Cin=K // S, lda=K, ldc=S * N, bias=bs, alpha=alpha, groups=S, a_gstride=K // S, c_gstride=N,
            w_gstride=N * (K // S), dtype=dtype)
    if ln is not None and N in (512, 1024):
        splitk_reduce_layernorm(P, x, ln[0], ln[1], ln[2], ln[3], M=M, C=N, S=S, dtype=dtype)
        return
    splitk_reduce(P, x, M=M, N=N, S=S)
    if ln is not None:
        layernorm(x, ln[0], ln[1], ln[2], ln[3], M=M, C=N, dtype=dtype)


def cast_f32_to_16(x, y, M, C, dtype, ldx=None, ldy=None):
    _run("l2s_cast_f32_to_16", lambda: _lib.load().l2s_cast_f32_to_16(_ptr(x), ldx or C, _ptr(y), ldy or C, M, C, dtype, _stream()))


def cast_16_to_f32(x, y, M, C, dtype, ldx=None, ldy=None):
    _run("l2s_cast_16_to_f32", lambda: _lib.load().l2s_cast_16_to_f32(_ptr(x), ldx or C, _ptr(y), ldy or C, M, C, dtype, _stream()))


def broadcast_rows(v, y, *, B, T, C, ldy, col0=0, ldv=None, lens=None, len_mul=1, dtype=F16):
    _run("l2s_broadcast_rows", lambda: _lib.load().l2s_broadcast_rows(_ptr(v), ldv or C, _ptr(y), ldy, col0, _ptr(lens), len_mul, B, T, C,
                                         int(v.dtype == torch.float32), dtype, _stream()))


def transpose_ct_to_tc(x, y, *, B, C, T, ldy, col0=0, lens=None, len_mul=1, dtype=F16):
    _run("l2s_transpose_ct_to_tc", lambda: _lib.load().l2s_transpose_ct_to_tc(_ptr(x), _ptr(y), ldy, col0, _ptr(lens), len_mul, B, C, T, dtype,
                                             _stream()))


def embedding(code, table, y, *, B, L, C, ldy=None, lens=None, dtype=F16):
    _run("l2s_embedding", lambda: _lib.load().l2s_embedding(_ptr(code), _ptr(table), _ptr(y), ldy or C, _ptr(lens), B, L, C, dtype, _stream()))


def embedding_tokens(tok, table, y, *, B, L, C, token_offset=4, ldt=None, ldy=None, lens=None, len_mul=1, dtype=F16):
    _run("l2s_embedding_tokens", lambda: _lib.load().l2s_embedding_tokens(
        _ptr(tok), ldt or tok.shape[-1], token_offset, _ptr(table), table.shape[0], _ptr(y), ldy or C, _ptr(lens), len_mul, B, L, C,
        dtype, _stream()))


def rows_f32_to_16_masked(x, y, *, B, T, C, ldx=None, ldy=None, col0=0, lens=None, len_mul=1, dtype=F16):
    _run("l2s_rows_f32_to_16_masked", lambda: _lib.load().l2s_rows_f32_to_16_masked(
        _ptr(x), ldx or C, _ptr(y), ldy or C, col0, _ptr(lens), len_mul, B, T, C, dtype, _stream()))


def lens_from_mask(padding_mask, B, T, device):
    """int32 [B] = T - padding_mask.sum(-1) (sequence_generator.py:64-65) on the device, no host sync; None = no padding."""
    import torch
    lens = torch.empty(B, device=device, dtype=torch.int32)
    if padding_mask is not None:
        assert padding_mask.dtype == torch.bool and padding_mask.shape == (B, T) and padding_mask.is_contiguous()
    _run("l2s_lens_from_mask", lambda: _lib.load().l2s_lens_from_mask(_ptr(padding_mask), _ptr(lens), B, T, _stream()))
    return lens


def basicblock_fused(x, w1, b1, s1, w2, b2, s2, y, *, n_images, H, W, C=64, dtype=F16):
    M = float(n_images) * H * W
    _run("l2s_basicblock_fused", lambda: _lib.load().l2s_basicblock_fused(
        _ptr(x), _ptr(w1), _ptr(b1), _ptr(s1), _ptr(w2), _ptr(b2), _ptr(s2), _ptr(y), n_images, H, W, C, dtype, _stream()),
        flops=2.0 * 2 * M * C * C * 9, nbytes=2 * 2 * M * C)


def basicstage128_tail_fused(x0, t0, wa, ba, sa, w1, b1, s1, w2, b2, s2, y, *, n_images, H, W, dtype=F16):
    """The strided 128-channel stage behind its first conv in one launch (csrc/basicblock_phase.hip, TAIL): block 1's second conv
    with the 1x1 stride-2 downsample of the stage input x0 as one more K-tile of the same accumulation, then block 2 on the
    LDS-resident result.  wa: [128][9*128 + 64 + 64] = conv weights | downsample weights | zeros; ba: both folded biases summed."""
    M = float(n_images) * H * W
    _run("l2s_basicstage128_tail_fused", lambda: _lib.load().l2s_basicstage128_tail_fused(
        _ptr(x0), _ptr(t0), _ptr(wa), _ptr(ba), _ptr(sa), _ptr(w1), _ptr(b1), _ptr(s1), _ptr(w2), _ptr(b2), _ptr(s2), _ptr(y),
        n_images, H, W, dtype, _stream()),
        flops=2.0 * M * 128 * (3 * 9 * 128 + 64), nbytes=2 * M * (128 + 64 + 128))


def basiclayer_fused(x, ws, biases, slopes, y, *, n_images, H, W, C=64, dtype=F16):
    """len(ws) / 2 BasicBlocks of the 64-channel stage in one launch (csrc/basicblock.hip): ws / biases / slopes list conv1, conv2
    of block 0, conv1, conv2 of block 1, ..."""
    n = len(ws)
    if n % 2 or len(biases) != n or len(slopes) != n:
        raise L2SError("basiclayer_fused: 2 * n_blocks weights, biases and slopes")
    wp = (ctypes.c_void_p * n)(*[_ptr(_req(w, _TORCH16[dtype], "w")) for w in ws])
    bp = (ctypes.c_void_p * n)(*[_ptr(_req(b_, torch.float32, "bias")) for b_ in biases])
    sp = (ctypes.c_void_p * n)(*[_ptr(_req(s_, torch.float32, "slope")) for s_ in slopes])
    M = float(n_images) * H * W
    _run("l2s_basiclayer_fused", lambda: _lib.load().l2s_basiclayer_fused(
        _ptr(x), wp, bp, sp, n // 2, _ptr(y), n_images, H, W, C, dtype, _stream()),
        flops=2.0 * n * M * C * C * 9, nbytes=2 * 2 * M * C)


def split_hi_lo(x, hi, lo, *, B, T, C, act=0, slope=0.0, ldx=None, ld16=None, lens=None, len_mul=1, dtype=F16):
    _run("l2s_split_hi_lo", lambda: _lib.load().l2s_split_hi_lo(_ptr(x), ldx or C, _ptr(hi), _ptr(lo), ld16 or C, act, float(slope),
                                                                _ptr(lens), len_mul, B, T, C, dtype, _stream()))


def conv_post_tanh(x, w, bias, wav, pcm, *, B, T, C, k, lens=None, len_mul=1):
    _run("l2s_conv_post_tanh", lambda: _lib.load().l2s_conv_post_tanh(_ptr(x), _ptr(w), float(bias), _ptr(wav), _ptr(pcm), _ptr(lens), len_mul, B,
                                         T, C, k, _stream()))


def resblock_fused(xl, w, bias, xs, xl_out, *, B, T, C, k, dil, accumulate, slope, lens=None, len_mul=1, dtype=F16):
    _run(f"l2s_resblock_fused<C{C},k{k}>", lambda: _lib.load().l2s_resblock_fused(
        _ptr(xl), _ptr(w), _ptr(bias), _ptr(xs), _ptr(xl_out), _ptr(lens), len_mul, B, T, C, k, dil[0], dil[1], dil[2],
        int(bool(accumulate)), float(slope), dtype, _stream()),
        flops=2.0 * B * T * C * C * k * 6, nbytes=B * T * C * (2 + 4 + (4 if accumulate else 0) + (2 if xl_out is not None else 0)))


def resstage_fused(xl, ws, biases, xs, xl_out, *, B, T, C, ks, dils, slope, lens=None, len_mul=1, dtype=F16, xs_final=True):
    """The three ResBlocks (k = 3, 7, 11) of a narrow stage in one launch (csrc/resblock.hip): xs = sum_j rb_j(x),
    xl_out = leaky_relu(xs).  ws / biases: per-ResBlock fused layouts of resblock_fused; dils: per-ResBlock dilations.
    xs_final=False: xs is only the running sum's scratch (nobody reads the stage's fp32 sum), its last pass is not written."""
    n = len(ws)
    wp = (ctypes.c_void_p * n)(*[_ptr(_req(w, _TORCH16[dtype], "w")) for w in ws])
    bp = (ctypes.c_void_p * n)(*[_ptr(_req(b_, torch.float32, "bias")) for b_ in biases])
    kk = (ctypes.c_int * n)(*[int(k) for k in ks])
    dd = (ctypes.c_int * (3 * n))(*[int(d) for dl in dils for d in dl])
    if xs.dtype != torch.float32:
        raise L2SError("xs must be fp32")
    _run(f"l2s_resstage_fused<C{C}>", lambda: _lib.load().l2s_resstage_fused(
        _ptr(_req(xl, _TORCH16[dtype], "xl")), wp, bp, kk, dd, n, _ptr(xs), _ptr(xl_out), _ptr(lens), len_mul, B, T, C,
        float(slope), int(bool(xs_final)), dtype, _stream()),
        flops=sum(2.0 * B * T * C * C * k * 6 for k in ks),
        nbytes=B * T * C * (2 + (4 if xs_final else 0) + (2 if xl_out is not None else 0)))


def respair(x_l, w1, b1, w2, b2, *, B, T, C, k, dil, slope, y=None, xs=None, accumulate=False, lens=None, len_mul=1, dtype=F16,
            xs_final=True):
    """One fused conv pair of ResBlock1 (csrc/respair.hip).  y given alone: mid pair, y = leaky_relu(x').  xs given: last
    pair of a ResBlock, xs (+)= x' and (when y is given too) y = leaky_relu(xs).  xs_final=False (with y): xs is read for the
    sum and not written back - nobody reads the stage's fp32 sum."""
    lib = _lib.load()
    d = _lib.RespairDesc()
    d.X, d.W1, d.W2, d.b1, d.b2 = _ptr(_req(x_l, _TORCH16[dtype], "x_l")), _ptr(w1), _ptr(w2), _ptr(_req(b1, torch.float32, "b1")), _ptr(_req(b2, torch.float32, "b2"))
    d.Y, d.XS, d.lens = _ptr(y), _ptr(xs), _ptr(lens)
    if xs is not None and xs.dtype != torch.float32:
        raise L2SError("xs must be fp32")
    if lens is not None and lens.dtype != torch.int32:
        raise L2SError("lens must be int32")
    d.len_mul, d.B, d.T, d.C, d.k, d.dil = len_mul, B, T, C, k, dil
    d.last = 0 if xs is None else (1 if xs_final else 2)
    d.accumulate, d.dtype, d.slope = int(bool(accumulate)), dtype, float(slope)
    kind = "last" if xs is not None else "mid"
    arrays = 2 * 2 + (0 if xs is None else (4 if accumulate else 0) + (4 if xs_final else 0) - (0 if y is not None else 2))
    key = f"l2s_respair<C{C},{kind}>"    # rocprofv3 name: respair_kernel<Elem.., C, KIND> (k is a runtime argument)
    if _profiler is not None and _profiler.detail:
        key += f" k{k} dil{dil}"
    _run(key, lambda: lib.l2s_respair(ctypes.byref(d), _stream()),
         flops=2.0 * B * T * C * C * k * 2, nbytes=float(B) * T * C * arrays + 2.0 * 2 * C * C * k, shape=f"M{B * T} k{k}")


def respair_final(xs_l, w1s, b1s, w2s, b2s, y, *, B, T, C, ks, dils, slope, lens=None, len_mul=1, dtype=F16):
    """The last conv pairs of a stage's ResBlocks in one launch (csrc/respair_phase.hip: respair_final_kernel):
    y = leaky_relu(sum_j (c2_j(leaky_relu(c1_j(x_l[j]))) + x_j)); the stage's fp32 sum stays in accumulators.  xs_l / w1s / b1s /
    w2s / b2s: per-ResBlock lists (n <= 3) of the operands l2s_respair takes; ks, dils: their kernel sizes and dilations."""
    lib = _lib.load()
    n = len(xs_l)
    if not (1 <= n <= 3 and len(w1s) == len(w2s) == len(b1s) == len(b2s) == len(ks) == len(dils) == n):
        raise L2SError("respair_final: 1..3 ResBlocks, one entry per list each")
    d = _lib.RespairFinalDesc()
    for j in range(n):
        d.X[j], d.W1[j], d.W2[j] = _ptr(_req(xs_l[j], _TORCH16[dtype], "x_l")), _ptr(w1s[j]), _ptr(w2s[j])
        d.b1[j], d.b2[j] = _ptr(_req(b1s[j], torch.float32, "b1")), _ptr(_req(b2s[j], torch.float32, "b2"))
        d.k[j], d.dil[j] = int(ks[j]), int(dils[j])
    if lens is not None and lens.dtype != torch.int32:
        raise L2SError("lens must be int32")
    d.Y, d.lens = _ptr(_req(y, _TORCH16[dtype], "y")), _ptr(lens)
    d.n, d.len_mul, d.B, d.T, d.C, d.dtype, d.slope = n, len_mul, B, T, C, dtype, float(slope)
    _run(f"l2s_respair_final<C{C}>", lambda: lib.l2s_respair_final(ctypes.byref(d), _stream()),
         flops=sum(2.0 * B * T * C * C * k * 2 for k in ks), nbytes=float(B) * T * C * 2 * (n + 1) + sum(2.0 * 2 * C * C * k for k in ks),
         shape=f"M{B * T} n{n}")


def preprocess_frames(frames, y, *, B, T, Hin, Win, crop=88, mean=0.421, std=0.165, dtype=F16):
    _run("l2s_preprocess_frames", lambda: _lib.load().l2s_preprocess_frames(_ptr(frames), _ptr(y), B, T, Hin, Win, crop, mean, std, dtype, _stream()))


# ---- torch.library registration ("PyTorch-ROCm custom ops", SURVEY 8b last row) --------------------------------------------------
# Every launcher above is ALSO a dispatcher-visible operator `torch.ops.lip2speech.<name>` (schema below, CUDA = HIP kernel only: a
# CPU tensor finds no kernel and raises; a fake / meta implementation gives shapes to torch.compile and fake-tensor tracing), and
# the module-level names are rebound to route through the dispatcher - host code keeps calling `ops.tapgemm(...)` and ends in the
# same ctypes call.  The operators mutate their output arguments in place (`Tensor(a!)`) and return nothing, exactly as the C ABI
# does; nothing is computed here.  ALIASES are the operator names SURVEY 8(b) lists, bound to the entry that implements them.
_TG = ("(Tensor A, Tensor W, Tensor(a!) C, *, int M, int N, int Cin, int ntaps=1, int? lda=None, int? ldc=None, Tensor? bias=None, "
       "Tensor? slope=None, Tensor? R=None, int? ldr=None, Tensor(b!)? C2=None, int? ldc2=None, Tensor? lens=None, int mode=0, "
       "int T_out=0, int T_in=0, int stride=1, int dil=1, int off=0, int Ho=0, int Wo=0, int Hi=0, int Wi=0, int KW=1, int pad=0, "
       "int out_row_mul=1, int out_row_add=0, int mask_T=0, int mask_mul=1, int act=0, int flags=0, int dtype=0, float alpha=1.0, "
       "float act_slope=0.0, float slope2=0.0, int groups=1, int a_gstride=0, int c_gstride=0, int w_gstride=0, Tensor? ktab=None, "
       "float? kflops=None) -> ()")
_ATT = ("(Tensor qkv, Tensor(a!) out, *, int B, int T, int H, int? ldq=None, int? ldo=None, Tensor? pos=None, int ldp=0, "
        "Tensor? bias_u=None, Tensor? bias_v=None, Tensor? lens=None, int len_mul=1, int dtype=0) -> ()")
_STEM = "(Tensor x, Tensor w, Tensor bias, Tensor? slope, Tensor(a!) y, int B, int T, int dtype) -> ()"
_RP = ("(Tensor x_l, Tensor w1, Tensor b1, Tensor w2, Tensor b2, *, int B, int T, int C, int k, int dil, float slope, "
       "Tensor(a!)? y=None, Tensor(b!)? xs=None, bool accumulate=False, Tensor? lens=None, int len_mul=1, int dtype=0, "
       "bool xs_final=True) -> ()")
_CPT = ("(Tensor x, Tensor w, float bias, Tensor(a!) wav, Tensor(b!)? pcm, *, int B, int T, int C, int k, Tensor? lens=None, "
        "int len_mul=1) -> ()")
_GLU = ("(Tensor x, Tensor w, Tensor bias, Tensor(a!) y, *, int B, int T, int C, int k, Tensor? lens=None, int len_mul=1, "
        "int dtype=0) -> ()")
_LN = ("(Tensor x, Tensor gamma, Tensor beta, float eps, Tensor(a!) y, *, int M, int C, int? ldx=None, int? ldy=None, "
       "Tensor(b!)? y2=None, int ldy2=0, int zero_prefix=0, Tensor? lens=None, int len_mul=1, int mask_T=0, int dtype=0) -> ()")
_GD = ("(Tensor logits, Tensor(a!) tokens, Tensor(b!) lprobs, Tensor(c!) score, *, int B, int T2, int V, int? ldl=None, "
       "Tensor? lens=None, int len_mul=1, float temperature=1.0, float lenpen=1.0) -> ()")
_BL = "(Tensor x, Tensor[] ws, Tensor[] biases, Tensor[] slopes, Tensor(a!) y, *, int n_images, int H, int W, int C=64, int dtype=0) -> ()"
_SCHEMAS = {
    "tapgemm": _TG,
    "stem_conv3d": _STEM,
    "stem_pool_fused": _STEM,
    "stem_pool_fused_u8": "(Tensor frames, Tensor w, Tensor bias, Tensor? slope, Tensor(a!) y, int B, int T, int dtype, int crop=88, "
                          "float mean=0.421, float std=0.165) -> ()",
    "maxpool2d_3x3s2": "(Tensor x, Tensor(a!) y, int N, int H, int W, int C, int dtype) -> ()",
    "avgpool_hw": "(Tensor x, Tensor(a!) y, int N, int HW, int C, int dtype) -> ()",
    "layernorm": _LN,
    "attention": _ATT,
    "glu_dwconv_swish": _GLU,
    "greedy_decode": _GD,
    "beam_decode": "(Tensor logits, *, int B, int T2, int V, int beam, int? ldl=None, Tensor? lens=None, int len_mul=1, "
                   "float temperature=1.0, float lenpen=1.0) -> (Tensor, Tensor, Tensor, Tensor)",
    "repeat2_cast": "(Tensor x, Tensor(a!) y, int B, int T, int C, int dtype) -> ()",
    "splitk_reduce": "(Tensor P, Tensor(a!) x, *, int M, int N, int S, int? ldp=None, int? ldx=None) -> ()",
    "splitk_reduce_layernorm": "(Tensor P, Tensor(a!) x, Tensor gamma, Tensor beta, float eps, Tensor(b!) y, *, int M, int C, int S, "
                               "int? ldp=None, int? ldx=None, int? ldy=None, Tensor? lens=None, int len_mul=1, int mask_T=0, int dtype=0) -> ()",
    "cast_f32_to_16": "(Tensor x, Tensor(a!) y, int M, int C, int dtype, int? ldx=None, int? ldy=None) -> ()",
    "cast_16_to_f32": "(Tensor x, Tensor(a!) y, int M, int C, int dtype, int? ldx=None, int? ldy=None) -> ()",
    "broadcast_rows": "(Tensor v, Tensor(a!) y, *, int B, int T, int C, int ldy, int col0=0, int? ldv=None, Tensor? lens=None, "
                      "int len_mul=1, int dtype=0) -> ()",
    "transpose_ct_to_tc": "(Tensor x, Tensor(a!) y, *, int B, int C, int T, int ldy, int col0=0, Tensor? lens=None, int len_mul=1, "
                          "int dtype=0) -> ()",
    "embedding": "(Tensor code, Tensor table, Tensor(a!) y, *, int B, int L, int C, int? ldy=None, Tensor? lens=None, int dtype=0) -> ()",
    "embedding_tokens": "(Tensor tok, Tensor table, Tensor(a!) y, *, int B, int L, int C, int token_offset=4, int? ldt=None, "
                        "int? ldy=None, Tensor? lens=None, int len_mul=1, int dtype=0) -> ()",
    "rows_f32_to_16_masked": "(Tensor x, Tensor(a!) y, *, int B, int T, int C, int? ldx=None, int? ldy=None, int col0=0, "
                             "Tensor? lens=None, int len_mul=1, int dtype=0) -> ()",
    "lens_from_mask": "(Tensor? padding_mask, int B, int T, Device device) -> Tensor",
    "basicblock_fused": "(Tensor x, Tensor w1, Tensor b1, Tensor s1, Tensor w2, Tensor b2, Tensor s2, Tensor(a!) y, *, int n_images, "
                        "int H, int W, int C=64, int dtype=0) -> ()",
    "basiclayer_fused": _BL,
    "basicstage128_tail_fused": "(Tensor x0, Tensor t0, Tensor wa, Tensor ba, Tensor sa, Tensor w1, Tensor b1, Tensor s1, Tensor w2, "
                                "Tensor b2, Tensor s2, Tensor(a!) y, *, int n_images, int H, int W, int dtype=0) -> ()",
    "split_hi_lo": "(Tensor x, Tensor(a!) hi, Tensor(b!) lo, *, int B, int T, int C, int act=0, float slope=0.0, int? ldx=None, "
                   "int? ld16=None, Tensor? lens=None, int len_mul=1, int dtype=0) -> ()",
    "conv_post_tanh": _CPT,
    "resblock_fused": "(Tensor xl, Tensor w, Tensor bias, Tensor(a!) xs, Tensor(b!)? xl_out, *, int B, int T, int C, int k, int[] dil, "
                      "bool accumulate, float slope, Tensor? lens=None, int len_mul=1, int dtype=0) -> ()",
    "resstage_fused": "(Tensor xl, Tensor[] ws, Tensor[] biases, Tensor(a!) xs, Tensor(b!)? xl_out, *, int B, int T, int C, int[] ks, "
                      "int[] dils, float slope, Tensor? lens=None, int len_mul=1, int dtype=0, bool xs_final=True) -> ()",
    "respair": _RP,
    "respair_final": "(Tensor[] xs_l, Tensor[] w1s, Tensor[] b1s, Tensor[] w2s, Tensor[] b2s, Tensor(a!) y, *, int B, int T, int C, int[] ks, "
                     "int[] dils, float slope, Tensor? lens=None, int len_mul=1, int dtype=0) -> ()",
    "preprocess_frames": "(Tensor frames, Tensor(a!) y, *, int B, int T, int Hin, int Win, int crop=88, float mean=0.421, "
                         "float std=0.165, int dtype=0) -> ()",
}
# C-ABI entry each operator launches (tests/test_torchlib_cpu.py: every device entry of include/lip2speech_hip.h has a twin)
ENTRY_OF = {n: "l2s_" + n for n in _SCHEMAS}
ENTRY_OF.update({"maxpool2d_3x3s2": "l2s_maxpool2d_3x3s2", "avgpool_hw": "l2s_avgpool_hw"})
# host-side queries of the ABI (no launch, nothing for the dispatcher to see)
HOST_QUERIES = ("l2s_abi_version", "l2s_build_info", "l2s_tapgemm_variant", "l2s_tapgemm_epilogue_family", "l2s_beam_decode_workspace")
# SURVEY 8(b)'s operator names -> the entry that implements them (`mel_head` is a composition of linear_epilogue launches,
# conformer.py::Conformer.forward_rows; it has no kernel of its own)
ALIASES = {"frontend3d_stem": "stem_pool_fused", "resnet_trunk": "basiclayer_fused", "linear_epilogue": "tapgemm",
           "posconv_gelu": "tapgemm", "convtranspose1d": "tapgemm", "mel_head": "tapgemm", "mhsa_padmask": "attention",
           "relpos_mhsa": "attention", "conformer_conv_module": "glu_dwconv_swish", "greedy_unit_decode": "greedy_decode",
           "resblock1": "respair", "tanh_to_int16": "conv_post_tanh"}

_TORCH_LIB = torch.library.Library("lip2speech", "DEF")


def _fake_none(*a, **k):
    return None


def _fake_beam_decode(logits, *, B, T2, V, beam, ldl=None, lens=None, len_mul=1, temperature=1.0, lenpen=1.0):
    return (logits.new_empty((B, beam, T2 + 1), dtype=torch.int32), logits.new_empty((B, beam, T2 + 1), dtype=torch.float32),
            logits.new_empty((B, beam), dtype=torch.float32), logits.new_empty((B,), dtype=torch.int32))


def _fake_lens_from_mask(padding_mask, B, T, device):
    return torch.empty(B, device=device, dtype=torch.int32)


def _resstage_fused_flat(xl, ws, biases, xs, xl_out, *, B, T, C, ks, dils, slope, lens=None, len_mul=1, dtype=F16, xs_final=True):
    return _IMPL["resstage_fused"](xl, ws, biases, xs, xl_out, B=B, T=T, C=C, ks=ks, dils=[dils[i:i + 3] for i in range(0, len(dils), 3)],
                                   slope=slope, lens=lens, len_mul=len_mul, dtype=dtype, xs_final=xs_final)


_IMPL = {}


def _register_torch_ops():
    g = globals()
    for name, schema in _SCHEMAS.items():
        _IMPL[name] = g[name]
    for name, schema in _SCHEMAS.items():
        impl = _resstage_fused_flat if name == "resstage_fused" else _IMPL[name]
        fake = {"beam_decode": _fake_beam_decode, "lens_from_mask": _fake_lens_from_mask}.get(name, _fake_none)
        for op in [name] + [a for a, base in ALIASES.items() if base == name]:
            _TORCH_LIB.define(op + schema)
            # lens_from_mask may be called with no tensor at all (padding_mask = None): it cannot dispatch on a device key
            _TORCH_LIB.impl(op, impl, "CompositeExplicitAutograd" if name == "lens_from_mask" else "CUDA")
            torch.library.register_fake("lip2speech::" + op)(fake)

    def router(name):
        op = getattr(torch.ops.lip2speech, name)

        def call(*a, **k):
            try:
                return op(*a, **k)
            except NotImplementedError as e:    # the dispatcher found no kernel: the operators exist for the HIP device only
                raise L2SError(f"lip2speech::{name}: HIP ops need device tensors (there is no CPU path)") from e
        call.__name__ = name
        call.__doc__ = _IMPL[name].__doc__
        return call
    for name in _SCHEMAS:
        if name == "resstage_fused":
            op = torch.ops.lip2speech.resstage_fused

            def resstage_fused(xl, ws, biases, xs, xl_out, *, B, T, C, ks, dils, slope, lens=None, len_mul=1, dtype=F16, xs_final=True):
                try:
                    return op(xl, list(ws), list(biases), xs, xl_out, B=B, T=T, C=C, ks=[int(k) for k in ks],
                              dils=[int(d) for dl in dils for d in dl], slope=slope, lens=lens, len_mul=len_mul, dtype=dtype,
                              xs_final=xs_final)
                except NotImplementedError as e:
                    raise L2SError("lip2speech::resstage_fused: HIP ops need device tensors (there is no CPU path)") from e
            resstage_fused.__doc__ = _IMPL[name].__doc__
            g[name] = resstage_fused
        else:
            g[name] = router(name)


_register_torch_ops()
