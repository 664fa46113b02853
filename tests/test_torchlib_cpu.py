"""The C ABI's device entries are also dispatcher-visible PyTorch operators (torch.ops.lip2speech.*, SURVEY 8b last row):
registration, schemas and the names SURVEY lists - no compute, no GPU."""
import re

import pytest
import torch

from lip2speech_unit_amd import _lib, ops


def test_every_device_entry_of_the_abi_has_a_torch_op_twin():
    hdr = open(__import__("os").path.join(__import__("os").path.dirname(__file__), "..", "include", "lip2speech_hip.h")).read()
    declared = set(re.findall(r"\b(l2s_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if d in _lib.SIGNATURES}
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    twins = set(ops.ENTRY_OF.values())
    assert twins | set(ops.HOST_QUERIES) == declared, declared - twins - set(ops.HOST_QUERIES)
    for name in ops.ENTRY_OF:
        assert hasattr(torch.ops.lip2speech, name)
        op = getattr(torch.ops.lip2speech, name).default
        schema = str(op._schema)
        # in-place outputs are declared as mutated arguments; only the two allocating entries return tensors
        if name in ("beam_decode", "lens_from_mask"):
            assert "->" in schema and "Tensor" in schema.split("->")[1]
        else:
            assert schema.rstrip().endswith("-> ()") and "!" in schema, schema


def test_survey_operator_names_are_registered():
    want = ["frontend3d_stem", "resnet_trunk", "linear_epilogue", "layernorm", "posconv_gelu", "mhsa_padmask", "relpos_mhsa",
            "conformer_conv_module", "mel_head", "greedy_unit_decode", "convtranspose1d", "resblock1", "tanh_to_int16"]
    for n in want:
        assert hasattr(torch.ops.lip2speech, n), n
        base = ops.ALIASES.get(n, n)
        a, b = getattr(torch.ops.lip2speech, n).default._schema, getattr(torch.ops.lip2speech, base).default._schema
        assert [x.name for x in a.arguments] == [x.name for x in b.arguments]


def test_no_cpu_kernel_and_fake_shapes():
    x = torch.zeros(4, 8)
    with pytest.raises(NotImplementedError):        # the dispatcher has a HIP kernel only
        torch.ops.lip2speech.repeat2_cast(x, torch.zeros(8, 8, dtype=torch.float16), 1, 4, 8, ops.F16)
    with pytest.raises(ops.L2SError):               # ... and the host wrappers turn that into the package's own error
        ops.repeat2_cast(x, torch.zeros(8, 8, dtype=torch.float16), 1, 4, 8, ops.F16)
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        logits = torch.empty(2, 9, 204, device="cuda")
        tok, pos, score, nhyp = torch.ops.lip2speech.beam_decode(logits, B=2, T2=9, V=204, beam=5)
        assert tok.shape == (2, 5, 10) and tok.dtype == torch.int32 and score.shape == (2, 5) and nhyp.shape == (2,)
        y = torch.empty(8, 8, device="cuda", dtype=torch.float16)
        assert torch.ops.lip2speech.repeat2_cast(torch.empty(4, 8, device="cuda"), y, 1, 4, 8, ops.F16) is None


def test_splitk_slices_depend_on_the_layer_only():
    """ops.splitk_slices: M switches the split-K form on (<= SPLITK_MAX_M rows), the slice count follows from K alone (K / S >= 256,
    S <= 8, whole 64-column K-tiles per slice) - so the re-packed weight a layer caches is the same for every batch that takes the path."""
    from lip2speech_unit_amd import ops
    for N, K, S in ((1024, 4096, 8), (512, 2048, 8), (1024, 1024, 4), (1024, 1536, 4), (1024, 512, 0), (1000, 4096, 0)):
        got = {ops.splitk_slices(M, N, K) for M in (1, 37, 100, 250, 500, ops.SPLITK_MAX_M)}
        assert got == {S}, (N, K, got)
        assert ops.splitk_slices(ops.SPLITK_MAX_M + 1, N, K) == 0
        if S:
            assert K % (64 * S) == 0 and K // S >= 256
