"""torch.ops.lip2speech.* on the device: the operators trace under torch.compile (dynamo, no graph break on any of them) and the
compiled function launches the same kernels with the same bits as the eager call."""
import pytest
import torch

from lip2speech_unit_amd import ops
from lip2speech_unit_amd.ops import F_RES_POST

pytestmark = pytest.mark.gpu


def _layer(x32, g, b, wqkv, bqkv, wo, bo, lens, B, T):
    """One pre-LN self-attention block on rows (LayerNorm -> QKV GEMM -> attention -> out-proj + fp32 residual)."""
    M, C = x32.shape
    h = torch.empty(M, C, device=x32.device, dtype=torch.float16)
    ops.layernorm(x32, g, b, 1e-5, h, M=M, C=C)
    qkv = torch.empty(M, 3 * C, device=x32.device, dtype=torch.float16)
    ops.tapgemm(h, wqkv, qkv, M=M, N=3 * C, Cin=C, bias=bqkv)
    att = torch.empty(M, C, device=x32.device, dtype=torch.float16)
    ops.attention(qkv, att, B=B, T=T, H=C // 64, lens=lens)
    out = x32.clone()
    ops.tapgemm(att, wo, out, M=M, N=C, Cin=C, bias=bo, R=out, ldr=C, flags=F_RES_POST)
    return out


def test_ops_trace_under_torch_compile_without_graph_breaks():
    import torch._dynamo as dynamo
    B, T, C = 3, 40, 256
    g0 = torch.Generator().manual_seed(5)
    dev = "cuda"
    x = torch.randn(B * T, C, generator=g0).to(dev)
    g, b = (1 + 0.1 * torch.randn(C, generator=g0)).to(dev), (0.1 * torch.randn(C, generator=g0)).to(dev)
    wqkv = (torch.randn(3 * C, C, generator=g0) / C ** 0.5).half().to(dev)
    wo = (torch.randn(C, C, generator=g0) / C ** 0.5).half().to(dev)
    bqkv, bo = (0.1 * torch.randn(3 * C, generator=g0)).to(dev), (0.1 * torch.randn(C, generator=g0)).to(dev)
    lens = torch.tensor([T, T - 7, 1], dtype=torch.int32, device=dev)
    want = _layer(x, g, b, wqkv, bqkv, wo, bo, lens, B, T)
    dynamo.reset()
    ex = dynamo.explain(_layer)(x, g, b, wqkv, bqkv, wo, bo, lens, B, T)
    assert ex.graph_break_count == 0, ex.break_reasons
    names = {str(n.target) for gr in ex.graphs for n in gr.graph.nodes if n.op == "call_function"}
    for op in ("lip2speech.layernorm", "lip2speech.tapgemm", "lip2speech.attention"):
        assert any(op in n for n in names), (op, names)
    fn = torch.compile(_layer, backend="eager", fullgraph=True)
    got = fn(x, g, b, wqkv, bqkv, wo, bo, lens, B, T)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert torch.isfinite(got).all() and (got - x).abs().max().item() > 1e-3      # the block did something


def test_survey_alias_runs_the_same_kernel():
    M, C = 64, 128
    a = torch.randn(M, C, device="cuda").half()
    w = (torch.randn(C, C, device="cuda") / C ** 0.5).half()
    c1 = torch.empty(M, C, device="cuda", dtype=torch.float16)
    c2 = torch.empty_like(c1)
    torch.ops.lip2speech.tapgemm(a, w, c1, M=M, N=C, Cin=C)
    torch.ops.lip2speech.linear_epilogue(a, w, c2, M=M, N=C, Cin=C)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)
