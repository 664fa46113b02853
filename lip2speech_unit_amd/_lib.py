"""ctypes binding of liblip2speech_hip.so (the C ABI declared in include/lip2speech_hip.h).

There is no CPU fallback: if the library is missing or a call is rejected the caller gets an exception.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("L2S_LIB_PATH") or os.path.join(_HERE, "liblip2speech_hip.so")  # env override: A/B builds

# mirrors of the header's enums
F16, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SWISH, ACT_PRELU, ACT_LRELU, ACT_TANH = range(7)
F_RES_PRE, F_RES_POST, F_ACCUM, F_DUAL, F_MASK, F_OUT_F32, F_RES_F32 = (1 << i for i in range(7))
MODE_LINEAR, MODE_CONV1D, MODE_CONV2D = 0, 1, 2
ABI_VERSION = 14

_ERR = {-1: "L2S_EINVAL", -2: "L2S_ESHAPE", -3: "L2S_EALIGN", -4: "L2S_EUNSUPPORTED"}


class L2SError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("W", ctypes.c_void_p), ("C", ctypes.c_void_p), ("C2", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("slope", ctypes.c_void_p), ("R", ctypes.c_void_p), ("lens", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("Cin", ctypes.c_int32), ("ntaps", ctypes.c_int32),
        ("lda", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldc2", ctypes.c_int32), ("ldr", ctypes.c_int32),
        ("mode", ctypes.c_int32),
        ("T_out", ctypes.c_int32), ("T_in", ctypes.c_int32), ("stride", ctypes.c_int32), ("dil", ctypes.c_int32),
        ("off", ctypes.c_int32),
        ("Ho", ctypes.c_int32), ("Wo", ctypes.c_int32), ("Hi", ctypes.c_int32), ("Wi", ctypes.c_int32),
        ("KW", ctypes.c_int32), ("pad", ctypes.c_int32),
        ("out_row_mul", ctypes.c_int32), ("out_row_add", ctypes.c_int32),
        ("mask_T", ctypes.c_int32), ("mask_mul", ctypes.c_int32),
        ("act", ctypes.c_int32), ("flags", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("alpha", ctypes.c_float), ("act_slope", ctypes.c_float), ("slope2", ctypes.c_float),
        ("groups", ctypes.c_int32), ("a_gstride", ctypes.c_int32), ("c_gstride", ctypes.c_int32),
        ("w_gstride", ctypes.c_int64),
        ("ktab", ctypes.c_void_p),
    ]


_vp, _i, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float


class RespairDesc(ctypes.Structure):
    _fields_ = [("X", ctypes.c_void_p), ("W1", ctypes.c_void_p), ("W2", ctypes.c_void_p), ("b1", ctypes.c_void_p),
                ("b2", ctypes.c_void_p), ("Y", ctypes.c_void_p), ("XS", ctypes.c_void_p), ("lens", ctypes.c_void_p),
                ("len_mul", ctypes.c_int32), ("B", ctypes.c_int32), ("T", ctypes.c_int32), ("C", ctypes.c_int32),
                ("k", ctypes.c_int32), ("dil", ctypes.c_int32), ("last", ctypes.c_int32), ("accumulate", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("slope", ctypes.c_float)]


class RespairFinalDesc(ctypes.Structure):
    _fields_ = [("X", ctypes.c_void_p * 3), ("W1", ctypes.c_void_p * 3), ("W2", ctypes.c_void_p * 3), ("b1", ctypes.c_void_p * 3),
                ("b2", ctypes.c_void_p * 3), ("k", ctypes.c_int32 * 3), ("dil", ctypes.c_int32 * 3),
                ("Y", ctypes.c_void_p), ("lens", ctypes.c_void_p),
                ("n", ctypes.c_int32), ("len_mul", ctypes.c_int32), ("B", ctypes.c_int32), ("T", ctypes.c_int32),
                ("C", ctypes.c_int32), ("dtype", ctypes.c_int32), ("slope", ctypes.c_float)]


# name -> argtypes; every symbol include/lip2speech_hip.h declares
SIGNATURES = {
    "l2s_abi_version": ([], ctypes.c_int),
    "l2s_build_info": ([], ctypes.c_char_p),
    "l2s_tapgemm": ([ctypes.POINTER(GemmDesc), _vp], _i),
    "l2s_tapgemm_variant": ([ctypes.POINTER(GemmDesc)], _i),
    "l2s_tapgemm_epilogue_family": ([ctypes.POINTER(GemmDesc)], _i),
    "l2s_stem_conv3d": ([_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_stem_pool_fused": ([_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_stem_pool_fused_u8": ([_vp, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _vp], _i),
    "l2s_maxpool2d_3x3s2": ([_vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_avgpool_hw": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "l2s_layernorm": ([_vp, _i, _i, _vp, _vp, _f, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp], _i),
    "l2s_attention": ([_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_glu_dwconv_swish": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_greedy_decode": ([_vp, _i, _vp, _i, _i, _i, _i, _f, _f, _vp, _vp, _vp, _vp], _i),
    "l2s_beam_decode": ([_vp, _i, _vp, _i, _i, _i, _i, _f, _f, _i, _vp, ctypes.c_size_t, _vp, _vp, _vp, _vp, _vp], _i),
    "l2s_beam_decode_workspace": ([_i, _i, _i], ctypes.c_size_t),
    "l2s_repeat2_cast": ([_vp, _vp, _i, _i, _i, _i, _vp], _i),
    "l2s_splitk_reduce": ([_vp, _i, _i, _vp, _i, _i, _i, _vp], _i),
    "l2s_splitk_reduce_layernorm": ([_vp, _i, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _vp], _i),
    "l2s_cast_f32_to_16": ([_vp, _i, _vp, _i, _i, _i, _i, _vp], _i),
    "l2s_cast_16_to_f32": ([_vp, _i, _vp, _i, _i, _i, _i, _vp], _i),
    "l2s_broadcast_rows": ([_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_transpose_ct_to_tc": ([_vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_embedding": ([_vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp], _i),
    "l2s_embedding_tokens": ([_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_rows_f32_to_16_masked": ([_vp, _i, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_lens_from_mask": ([_vp, _vp, _i, _i, _vp], _i),
    "l2s_basicblock_fused": ([_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_basiclayer_fused": ([_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_basicstage128_tail_fused": ([_vp] * 12 + [_i, _i, _i, _i, _vp], _i),
    "l2s_split_hi_lo": ([_vp, _i, _vp, _vp, _i, _i, _f, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_conv_post_tanh": ([_vp, _vp, _f, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "l2s_resblock_fused": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp], _i),
    "l2s_resstage_fused": ([_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _i, _vp], _i),
    "l2s_respair": ([ctypes.POINTER(RespairDesc), _vp], _i),
    "l2s_respair_final": ([ctypes.POINTER(RespairFinalDesc), _vp], _i),
    "l2s_preprocess_frames": ([_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp], _i),
}

_lib = None


def load():
    """Load the HIP library; raises L2SError when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise L2SError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C lip2speech_unit_amd/csrc). The MI355X path has no CPU fallback.")
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (streams and device pointers
    # we pass come from it), so make sure it is loaded before this library's DT_NEEDED entry resolves.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise L2SError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.argtypes = argtypes
        fn.restype = restype
    v = lib.l2s_abi_version()
    if v != ABI_VERSION:
        raise L2SError(f"ABI mismatch: library {v}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise L2SError(f"{what}: rejected with {_ERR.get(rc, rc)}")
    raise L2SError(f"{what}: HIP launch failed with hipError_t={rc}")
