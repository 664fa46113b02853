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
