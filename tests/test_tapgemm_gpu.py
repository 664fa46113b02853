"""tap-GEMM kernel family vs torch fp32 on CPU (the oracle ops it replaces), through the C ABI."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from lip2speech_unit_amd import ops  # noqa: E402

TOL = {ops.F16: 2e-3, ops.BF16: 1.5e-2}  # relative to max|ref|, inputs pre-rounded to the 16-bit type


def _r16(x, dt):
    return x.to(ops.torch_dtype(dt)).float()


def _check(got, ref, dt, what):
    err = (got.float().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item() + 1e-6
    assert err <= TOL[dt] * scale, f"{what}: max err {err} vs scale {scale}"


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("M,N,K", [(300, 204, 512), (128, 128, 64), (1000, 1024, 1024), (77, 16, 48), (260, 32, 96),
                                   (3200, 64, 576)])
def test_linear(dt, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N)
    a = _r16(torch.randn(M, K, generator=g), dt)
    w = _r16(torch.randn(N, K, generator=g) / K ** 0.5, dt)
    b = torch.randn(N, generator=g)
    ref = F.gelu(a @ w.t() + b)
    t16 = ops.torch_dtype(dt)
    A, W, Bv = a.to(t16).cuda(), w.to(t16).cuda(), b.cuda()
    C = torch.empty(M, N, dtype=t16, device="cuda")
    ops.tapgemm(A, W, C, M=M, N=N, Cin=K, bias=Bv, act=ops.ACT_GELU, dtype=dt)
    torch.cuda.synchronize()
    _check(C, ref, dt, "linear+gelu")
    # fp32 output with fp32 residual, alpha
    r = torch.randn(M, N, generator=g)
    ref2 = r + 0.5 * (a @ w.t() + b)
    C32 = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.tapgemm(A, W, C32, M=M, N=N, Cin=K, bias=Bv, R=r.cuda(), flags=ops.F_RES_POST, alpha=0.5, dtype=dt)
    torch.cuda.synchronize()
    _check(C32, ref2, dt, "linear+res f32")


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("B,T,Cin,Cout,k,dil", [(2, 50, 64, 64, 3, 1), (3, 37, 32, 32, 7, 3), (1, 100, 16, 16, 11, 5),
                                                 (2, 40, 336, 512, 7, 1), (2, 33, 768, 512, 3, 1)])
def test_conv1d(dt, B, T, Cin, Cout, k, dil):
    g = torch.Generator().manual_seed(B * 100 + T + k)
    x = _r16(torch.randn(B, Cin, T, generator=g), dt)
    w = _r16(torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5, dt)
    b = torch.randn(Cout, generator=g)
    pad = (k * dil - dil) // 2
    ref = F.leaky_relu(F.conv1d(x, w, b, 1, pad, dil), 0.1).transpose(1, 2).reshape(B * T, Cout)
    t16 = ops.torch_dtype(dt)
    A = x.transpose(1, 2).contiguous().reshape(B * T, Cin).to(t16).cuda()
    W = w.permute(0, 2, 1).reshape(Cout, k * Cin).contiguous().to(t16).cuda()
    C = torch.empty(B * T, Cout, dtype=t16, device="cuda")
    ops.tapgemm(A, W, C, M=B * T, N=Cout, Cin=Cin, ntaps=k, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil,
                off=-pad, bias=b.cuda(), act=ops.ACT_LRELU, act_slope=0.1, dtype=dt)
    torch.cuda.synchronize()
    _check(C, ref, dt, "conv1d")


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("Cin,Cout,k,s", [(64, 32, 11, 5), (32, 16, 8, 4), (128, 128, 4, 2)])
def test_conv_transpose1d_phases(dt, Cin, Cout, k, s):
    B, L = 2, 23
    g = torch.Generator().manual_seed(k * 10 + s)
    x = _r16(torch.randn(B, Cin, L, generator=g), dt)
    w = _r16(torch.randn(Cin, Cout, k, generator=g) / (Cin * k / s) ** 0.5, dt)
    b = torch.randn(Cout, generator=g)
    p = (k - s) // 2
    ref = F.conv_transpose1d(x, w, b, s, p).transpose(1, 2).reshape(B * L * s, Cout)
    t16 = ops.torch_dtype(dt)
    A = x.transpose(1, 2).contiguous().reshape(B * L, Cin).to(t16).cuda()
    C = torch.empty(B * L * s, Cout, dtype=t16, device="cuda")
    from lip2speech_unit_amd.packing import convtranspose_phases
    for ph in convtranspose_phases(w, s, p):
        W = ph["w"].to(t16).cuda()
        ops.tapgemm(A, W, C, M=B * L, N=Cout, Cin=Cin, ntaps=ph["ntaps"], mode=ops.MODE_CONV1D, T_out=L, T_in=L,
                    stride=1, dil=-1, off=ph["off"], out_row_mul=s, out_row_add=ph["r"], bias=b.cuda(), dtype=dt)
    torch.cuda.synchronize()
    _check(C, ref, dt, "convtranspose1d")


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("N,H,Cin,Cout,k,s", [(5, 22, 64, 64, 3, 1), (5, 22, 64, 128, 3, 2), (7, 11, 128, 256, 1, 2),
                                               (9, 6, 256, 256, 3, 1), (11, 3, 512, 512, 3, 1)])
def test_conv2d(dt, N, H, Cin, Cout, k, s):
    g = torch.Generator().manual_seed(N * 13 + H)
    x = _r16(torch.randn(N, Cin, H, H, generator=g), dt)
    w = _r16(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, dt)
    b = torch.randn(Cout, generator=g)
    sl = torch.rand(Cout, generator=g) * 0.3
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    r = _r16(torch.randn(N, Cout, Ho, Ho, generator=g), dt)
    ref = F.prelu(F.conv2d(x, w, b, s, pad) + r, sl).permute(0, 2, 3, 1).reshape(N * Ho * Ho, Cout)
    t16 = ops.torch_dtype(dt)
    A = x.permute(0, 2, 3, 1).contiguous().reshape(N * H * H, Cin).to(t16).cuda()
    W = w.permute(0, 2, 3, 1).reshape(Cout, k * k * Cin).contiguous().to(t16).cuda()
    R = r.permute(0, 2, 3, 1).contiguous().reshape(N * Ho * Ho, Cout).to(t16).cuda()
    C = torch.empty(N * Ho * Ho, Cout, dtype=t16, device="cuda")
    ops.tapgemm(A, W, C, M=N * Ho * Ho, N=Cout, Cin=Cin, ntaps=k * k, mode=ops.MODE_CONV2D, Ho=Ho, Wo=Ho, Hi=H, Wi=H,
                KW=k, pad=pad, stride=s, bias=b.cuda(), slope=sl.cuda(), act=ops.ACT_PRELU, R=R, flags=ops.F_RES_PRE,
                dtype=dt)
    torch.cuda.synchronize()
    _check(C, ref, dt, "conv2d")


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
def test_grouped_posconv_mask_dual_accum(dt):
    # fairseq pos_conv shape family: groups, k even with trailing step removed; plus MASK / DUAL / ACCUM epilogues
    B, T, C, G, k = 2, 30, 128, 4, 16
    g = torch.Generator().manual_seed(5)
    x = _r16(torch.randn(B, C, T, generator=g), dt)
    w = _r16(torch.randn(C, C // G, k, generator=g) / (C // G * k) ** 0.5, dt)
    b = torch.randn(C, generator=g)
    y = F.conv1d(x, w, b, 1, k // 2, 1, G)[:, :, :-1]
    ref = (x + F.gelu(y)).transpose(1, 2).reshape(B * T, C)
    t16 = ops.torch_dtype(dt)
    A = x.transpose(1, 2).contiguous().reshape(B * T, C).to(t16).cuda()
    cg = C // G
    W = w.view(G, cg, cg, k).permute(0, 1, 3, 2).reshape(G, cg, k * cg).contiguous().to(t16).cuda()
    Cc = torch.empty(B * T, C, dtype=torch.float32, device="cuda")
    Rf = A.float()
    ops.tapgemm(A, W, Cc, M=B * T, N=cg, Cin=cg, ntaps=k, lda=C, ldc=C, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1,
                dil=1, off=-(k // 2), bias=b.cuda(), act=ops.ACT_GELU, R=Rf, ldr=C, flags=ops.F_RES_POST, dtype=dt,
                groups=G, a_gstride=cg, c_gstride=cg, w_gstride=cg * k * cg)
    torch.cuda.synchronize()
    _check(Cc, ref, dt, "grouped pos_conv")
    # mask + dual + accum
    lens = torch.tensor([30, 17], dtype=torch.int32)
    M, N, K = B * T, 64, 128
    a = _r16(torch.randn(M, K, generator=g), dt)
    wl = _r16(torch.randn(N, K, generator=g) / K ** 0.5, dt)
    prev = torch.randn(M, N, generator=g)
    ref = prev + a @ wl.t()
    keep = (torch.arange(T)[None, :] < lens[:, None]).reshape(M, 1).float()
    ref = ref * keep
    ref2 = F.leaky_relu(ref, 0.1)
    Cp = prev.clone().cuda()
    C2 = torch.empty(M, N, dtype=t16, device="cuda")
    ops.tapgemm(a.to(t16).cuda(), wl.to(t16).cuda(), Cp, M=M, N=N, Cin=K, C2=C2, lens=lens.cuda(), mask_T=T, mask_mul=1,
                flags=ops.F_ACCUM | ops.F_DUAL | ops.F_MASK, slope2=0.1, dtype=dt)
    torch.cuda.synchronize()
    _check(Cp, ref, dt, "accum+mask")
    _check(C2, ref2, dt, "dual")


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("C", [64, 128])
@pytest.mark.parametrize("k,dil", [(3, 1), (7, 3), (11, 5), (11, 1)])
def test_patchconv_conv1d(dt, k, dil, C):
    """Cin = N = 64 / 128 stride-1 Conv1d large enough to take the LDS-resident-patch kernel (csrc/patchconv.hip)."""
    B, T = 3, 24000                 # M = 72000 >= 65536; T is not a multiple of the 256-row tile
    g = torch.Generator().manual_seed(k * 10 + dil)
    x = _r16(torch.randn(B, C, T, generator=g), dt)
    w = _r16(torch.randn(C, C, k, generator=g) / (C * k) ** 0.5, dt)
    b = torch.randn(C, generator=g)
    lens = torch.tensor([T, T - 777, 5000])
    pad = (k * dil - dil) // 2
    keep = (torch.arange(T)[None, :] < lens[:, None])
    xm = x * keep[:, None, :]
    res = _r16(torch.randn(B, C, T, generator=g), dt)
    y = F.conv1d(xm, w, b, 1, pad, dil) + res
    y = y * keep[:, None, :]
    ref = y.transpose(1, 2).reshape(B * T, C)
    ref2 = F.leaky_relu(ref, 0.1)
    t16 = ops.torch_dtype(dt)
    A = xm.transpose(1, 2).contiguous().reshape(B * T, C).to(t16).cuda()
    R = res.transpose(1, 2).contiguous().reshape(B * T, C).to(t16).cuda()
    W = w.permute(0, 2, 1).reshape(C, k * C).contiguous().to(t16).cuda()
    Cout = torch.empty(B * T, C, dtype=t16, device="cuda")
    C2 = torch.empty(B * T, C, dtype=t16, device="cuda")
    kw = dict(M=B * T, N=C, Cin=C, ntaps=k, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil, off=-pad,
              bias=b.cuda(), R=R, C2=C2, lens=lens.int().cuda(), mask_T=T, mask_mul=1,
              flags=ops.F_RES_POST | ops.F_DUAL | ops.F_MASK, slope2=0.1, dtype=dt)
    import ctypes
    from lip2speech_unit_amd import _lib
    d = _lib.GemmDesc(M=B * T, N=C, Cin=C, ntaps=k, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil, off=-pad,
                      lda=C, groups=1)
    assert _lib.load().l2s_tapgemm_variant(ctypes.byref(d)) == 999000 + C, "expected the patch kernel to be selected"
    ops.tapgemm(A, W, Cout, **kw)
    torch.cuda.synchronize()
    _check(Cout, ref, dt, "patch conv1d")
    _check(C2, ref2, dt, "patch conv1d dual")


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("H,C", [(22, 64), (22, 128)])
def test_patchconv_conv2d(dt, H, C):
    N = 150                         # 72600 rows: ResNet layer1 shape family (avhubert/resnet.py:61-74), also at 128 channels
    import ctypes
    from lip2speech_unit_amd import _lib
    d = _lib.GemmDesc(M=N * H * H, N=C, Cin=C, ntaps=9, mode=ops.MODE_CONV2D, Ho=H, Wo=H, Hi=H, Wi=H, KW=3, pad=1,
                      stride=1, lda=C, groups=1)
    assert _lib.load().l2s_tapgemm_variant(ctypes.byref(d)) == 999000 + C, "expected the patch kernel to be selected"
    g = torch.Generator().manual_seed(77)
    x = _r16(torch.randn(N, C, H, H, generator=g), dt)
    w = _r16(torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5, dt)
    b = torch.randn(C, generator=g)
    sl = torch.rand(C, generator=g) * 0.3
    r = _r16(torch.randn(N, C, H, H, generator=g), dt)
    ref = F.prelu(F.conv2d(x, w, b, 1, 1) + r, sl).permute(0, 2, 3, 1).reshape(N * H * H, C)
    t16 = ops.torch_dtype(dt)
    A = x.permute(0, 2, 3, 1).contiguous().reshape(N * H * H, C).to(t16).cuda()
    W = w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(t16).cuda()
    R = r.permute(0, 2, 3, 1).contiguous().reshape(N * H * H, C).to(t16).cuda()
    Cout = torch.empty(N * H * H, C, dtype=t16, device="cuda")
    ops.tapgemm(A, W, Cout, M=N * H * H, N=C, Cin=C, ntaps=9, mode=ops.MODE_CONV2D, Ho=H, Wo=H, Hi=H, Wi=H, KW=3, pad=1,
                stride=1, bias=b.cuda(), slope=sl.cuda(), act=ops.ACT_PRELU, R=R, flags=ops.F_RES_PRE, dtype=dt)
    torch.cuda.synchronize()
    _check(Cout, ref, dt, "patch conv2d")
