#!/usr/bin/env python3
"""Checks the phase-staggered 256x256 kernel (csrc/phasegemm.hip, forced with L2S_PHASEGEMM=2) against torch fp32:
every epilogue family it is built for, ragged M / N tails, one to many tiles per block, repeated runs (the kernel's
LDS hand-over is ordered by counted vmcnt + barriers: a race would show as rare wrong tiles).
Used by tests/test_tiles_gpu.py."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from lip2speech_unit_amd import _lib, ops


def rel_err(got, ref):
    return (got.float().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-6)


def check_convs(lib):
    """Conv1d (dilated, transposed-phase style negative dilation, stride 1) and Conv2d 3x3 (stride 1 and 2) with
    Cin % 64 == 0 and N >= 256, lean / residual / dual / mask epilogue families."""
    worst = 0.0
    for dt, tol in ((ops.F16, 2e-3), (ops.BF16, 1.5e-2)):
        t16 = ops.torch_dtype(dt)
        g = torch.Generator().manual_seed(9)
        # ---- Conv1d: B clips x T samples, Cin 128 -> Co 256, k taps, dilation d
        for (B, T, Cin, Co, k, dil) in ((3, 1000, 128, 256, 7, 3), (2, 4100, 64, 320, 11, 5), (5, 700, 256, 256, 3, 1)):
            x = torch.randn(B, Cin, T, generator=g).to(t16).float()
            wc = (torch.randn(Co, Cin, k, generator=g) / (Cin * k) ** 0.5).to(t16).float()
            b = torch.randn(Co, generator=g)
            res = torch.randn(B, Co, T, generator=g).to(t16).float()
            lens = torch.randint(T // 2, T + 1, (B,), generator=g).int()
            keep = (torch.arange(T)[None, :] < lens[:, None]).float()[:, None, :]
            pad = (k * dil - dil) // 2
            conv = F.conv1d(x, wc, b, 1, pad, dil)
            rows = lambda y: y.transpose(1, 2).reshape(B * T, Co)
            A = x.transpose(1, 2).contiguous().reshape(B * T, Cin).to(t16).cuda()
            Wp = wc.permute(0, 2, 1).reshape(Co, k * Cin).contiguous().to(t16).cuda()
            R = rows(res).contiguous().to(t16).cuda()
            base = dict(M=B * T, N=Co, Cin=Cin, ntaps=k, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil, off=-pad,
                        bias=b.cuda(), dtype=dt)
            d = _lib.GemmDesc(M=B * T, N=Co, Cin=Cin, ntaps=k, mode=ops.MODE_CONV1D, T_out=T, T_in=T, stride=1, dil=dil,
                              off=-pad, lda=Cin, ldc=Co, ldr=Co, groups=1, flags=0, act=0)
            assert lib.l2s_tapgemm_variant(ctypes.byref(d)) == 256256, "phase kernel not selected for conv1d"
            cases = [   # the residual / dual families run on the 256x128 kernel: checked here all the same
                ("c1 lrelu+mask", dict(act=ops.ACT_LRELU, act_slope=0.1, lens=lens.cuda(), mask_T=T, mask_mul=1, flags=ops.F_MASK),
                 rows(F.leaky_relu(conv, 0.1) * keep), None),
                ("c2 res+dual+mask", dict(R=R, lens=lens.cuda(), mask_T=T, mask_mul=1, slope2=0.1,
                                          flags=ops.F_RES_POST | ops.F_DUAL | ops.F_MASK), rows((conv + res) * keep), 0.1),
                ("res pre relu", dict(R=R, act=ops.ACT_RELU, flags=ops.F_RES_PRE), rows(F.relu(conv + res)), None),
                ("dual only", dict(slope2=0.2, flags=ops.F_DUAL), rows(conv), 0.2),
            ]
            for name, kw, ref, s2 in cases:
                C = torch.full((B * T, Co), float("nan"), device="cuda", dtype=t16)
                C2 = torch.full((B * T, Co), float("nan"), device="cuda", dtype=t16) if s2 is not None else None
                ops.tapgemm(A, Wp, C, C2=C2, ldc2=Co if s2 is not None else None, **base, **kw)
                torch.cuda.synchronize()
                errs = [rel_err(C, ref)] + ([rel_err(C2, F.leaky_relu(ref, s2))] if s2 is not None else [])
                for e in errs:
                    if not (e <= tol):
                        print(f"FAIL conv1d {name} dt={dt} B{B} T{T} Cin{Cin} Co{Co} k{k} d{dil}: rel err {e:.3e}")
                    worst = max(worst, (e if e == e else 1e9) / tol)
        # ---- Conv2d 3x3: stride 1 (+ residual, PReLU) and stride 2
        for (N, H, Cin, Co, stride) in ((40, 22, 128, 256, 1), (64, 12, 256, 512, 2), (30, 11, 64, 256, 1)):
            x = torch.randn(N, Cin, H, H, generator=g).to(t16).float()
            wc = (torch.randn(Co, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5).to(t16).float()
            b = torch.randn(Co, generator=g)
            sl = torch.rand(Co, generator=g) * 0.3
            Ho = (H + 2 - 3) // stride + 1
            conv = F.conv2d(x, wc, b, stride, 1)
            res = torch.randn(N, Co, Ho, Ho, generator=g).to(t16).float()
            rows = lambda y: y.permute(0, 2, 3, 1).reshape(N * Ho * Ho, Co)
            A = x.permute(0, 2, 3, 1).contiguous().reshape(N * H * H, Cin).to(t16).cuda()
            Wp = wc.permute(0, 2, 3, 1).reshape(Co, 9 * Cin).contiguous().to(t16).cuda()
            R = rows(res).contiguous().to(t16).cuda()
            M = N * Ho * Ho
            base = dict(M=M, N=Co, Cin=Cin, ntaps=9, mode=ops.MODE_CONV2D, Ho=Ho, Wo=Ho, Hi=H, Wi=H, KW=3, pad=1, stride=stride,
                        bias=b.cuda(), dtype=dt)
            d = _lib.GemmDesc(M=M, N=Co, Cin=Cin, ntaps=9, mode=ops.MODE_CONV2D, Ho=Ho, Wo=Ho, Hi=H, Wi=H, KW=3, pad=1,
                              stride=stride, lda=Cin, ldc=Co, ldr=Co, groups=1, flags=0, act=0)
            assert lib.l2s_tapgemm_variant(ctypes.byref(d)) == 256256, "phase kernel not selected for conv2d"
            for name, kw, ref in (("prelu", dict(act=ops.ACT_PRELU, slope=sl.cuda()), rows(F.prelu(conv, sl))),
                                  ("res pre prelu", dict(act=ops.ACT_PRELU, slope=sl.cuda(), R=R, flags=ops.F_RES_PRE),
                                   rows(F.prelu(conv + res, sl)))):
                C = torch.full((M, Co), float("nan"), device="cuda", dtype=t16)
                ops.tapgemm(A, Wp, C, **base, **kw)
                torch.cuda.synchronize()
                e = rel_err(C, ref)
                if not (e <= tol):
                    print(f"FAIL conv2d {name} dt={dt} N{N} H{H} Cin{Cin} Co{Co} s{stride}: rel err {e:.3e}")
                worst = max(worst, (e if e == e else 1e9) / tol)
    return worst


def main():
    worst = 0.0
    lib = _lib.load()
    shapes = [(256, 256, 64), (1000, 384, 128), (2304, 512, 192), (16000, 1024, 512), (8200, 768, 1024)]
    for dt, tol in ((ops.F16, 2e-3), (ops.BF16, 1.5e-2)):
        t16 = ops.torch_dtype(dt)
        for si, (M, N, K) in enumerate(shapes):
            g = torch.Generator().manual_seed(100 + si)
            a = torch.randn(M, K, generator=g).to(t16)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(t16)
            b = torch.randn(N, generator=g)
            sl = torch.rand(N, generator=g) * 0.3
            r32 = torch.randn(M, N, generator=g)
            r16 = torch.randn(M, N, generator=g).to(t16)
            T = 125 if M % 125 == 0 else M                      # rows per "clip" for the length mask
            lens = torch.randint(1, T + 1, (M // T,), generator=g).int()
            keep = (torch.arange(T)[None, :] < lens[:, None]).reshape(M, 1).float()
            lin = a.float() @ w.float().t() + b
            ag, wg, bg = a.cuda(), w.cuda(), b.cuda()
            d = _lib.GemmDesc(M=M, N=N, Cin=K, ntaps=1, mode=ops.MODE_LINEAR, lda=K, ldc=N, groups=1, flags=0, act=0)
            assert lib.l2s_tapgemm_variant(ctypes.byref(d)) == 256256, "phase kernel not selected (L2S_PHASEGEMM=2?)"
            cases = [
                ("none", dict(), lin, None),
                ("relu", dict(act=ops.ACT_RELU), F.relu(lin), None),
                ("lrelu+alpha", dict(act=ops.ACT_LRELU, act_slope=0.1, alpha=0.5), F.leaky_relu(lin * 0.5, 0.1), None),
                ("prelu", dict(act=ops.ACT_PRELU, slope=sl.cuda()), torch.where(lin >= 0, lin, lin * sl), None),
                ("gelu", dict(act=ops.ACT_GELU), F.gelu(lin), None),
                ("gelu+mask", dict(act=ops.ACT_GELU, lens=lens.cuda(), mask_T=T, mask_mul=1, flags=ops.F_MASK), F.gelu(lin) * keep, None),
                ("none+mask", dict(lens=lens.cuda(), mask_T=T, mask_mul=1, flags=ops.F_MASK), lin * keep, None),
                ("res32 post", dict(R=r32.cuda(), flags=ops.F_RES_POST, alpha=0.5), lin * 0.5 + r32, torch.float32),
                # the residual stream updated in place (R is C: x += W h + b), the way the encoder / conformer call it
                ("res32 post in place", dict(R="C", flags=ops.F_RES_POST), lin + r32, torch.float32),
                ("res32 pre relu", dict(R=r32.cuda(), flags=ops.F_RES_PRE, act=ops.ACT_RELU), F.relu(lin + r32), torch.float32),
                ("res16 post f32out", dict(R=r16.cuda(), flags=ops.F_RES_POST), lin + r16.float(), torch.float32),
                ("res16 pre relu", dict(R=r16.cuda(), flags=ops.F_RES_PRE, act=ops.ACT_RELU), F.relu(lin + r16.float()), None),
                # the ResBlock-sum update of the un-fused vocoder stage (family L2S_EPI_X32): fp32 out, 16-bit residual, mask,
                # with / without accumulate and the LeakyReLU'd 16-bit copy
                ("x32 res16 mask", dict(R=r16.cuda(), lens=lens.cuda(), mask_T=T, mask_mul=1, flags=ops.F_RES_POST | ops.F_MASK),
                 (lin + r16.float()) * keep, torch.float32),
                ("x32 accum res16 mask dual", dict(R=r16.cuda(), lens=lens.cuda(), mask_T=T, mask_mul=1, slope2=0.1, dual=True,
                                                   flags=ops.F_RES_POST | ops.F_MASK | ops.F_ACCUM | ops.F_DUAL),
                 (lin + r16.float() + r32) * keep, torch.float32),
                ("x32 accum res16 alpha", dict(R=r16.cuda(), alpha=0.5, flags=ops.F_RES_POST | ops.F_ACCUM),
                 lin * 0.5 + r16.float() + r32, torch.float32),
                ("swish", dict(act=ops.ACT_SWISH), F.silu(lin), None),
                ("tanh", dict(act=ops.ACT_TANH), torch.tanh(lin), None),
                ("accum16 + res16 + mask", dict(R=r16.cuda(), lens=lens.cuda(), mask_T=T, mask_mul=1,
                                                flags=ops.F_ACCUM | ops.F_RES_POST | ops.F_MASK),
                 (lin + 2 * r16.float()) * keep, None),
            ]
            for name, kw, ref, odt in cases:
                for rep in range(3 if M >= 8000 else 1):
                    C = torch.full((M, N), float("nan"), device="cuda", dtype=odt or t16)
                    if kw.get("flags", 0) & ops.F_ACCUM:
                        C = (r32 if odt == torch.float32 else r16).cuda().clone()   # the accumulate operand is the output's previous content
                    kw_run = {k: v for k, v in kw.items() if k != "dual"}
                    C2 = None
                    if kw.get("dual"):
                        C2 = torch.full((M, N), float("nan"), device="cuda", dtype=t16)
                        kw_run.update(C2=C2, ldc2=N)
                    if isinstance(kw.get("R"), str):
                        C = r32.cuda().clone()
                        kw_run = dict(kw_run, R=C)
                    ops.tapgemm(ag, wg, C, M=M, N=N, Cin=K, bias=bg, dtype=dt, **kw_run)
                    torch.cuda.synchronize()
                    e = rel_err(C, ref)
                    if C2 is not None:
                        e = max(e, rel_err(C2, F.leaky_relu(ref, kw["slope2"])) * (2e-3 / tol if dt == ops.F16 else 1.0))
                    if not (e <= tol):
                        print(f"FAIL {name} dt={dt} shape={M}x{N}x{K} rep={rep}: rel err {e:.3e} (tol {tol})")
                    worst = max(worst, (e if e == e else 1e9) / tol)
    worst = max(worst, check_convs(lib))
    print(f"phasegemm worst err/tol = {worst:.3f}")
    sys.exit(0 if worst <= 1.0 else 1)


if __name__ == "__main__":
    main()
