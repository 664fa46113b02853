import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from lip2speech_unit_amd import ops
from lip2speech_unit_amd.packing import pack_conv1d

def r16(x, dt): return x.to(ops.torch_dtype(dt)).float()
def case(C, k, dil, T, lens, dt=ops.F16):
    t16 = ops.torch_dtype(dt)
    B, slope = len(lens), 0.1
    g = torch.Generator().manual_seed(C * 100 + k * 10 + dil)
    x = torch.randn(B, T, C, generator=g)
    w1 = torch.randn(C, C, k, generator=g) * (C * k) ** -0.5
    w2 = torch.randn(C, C, k, generator=g) * (C * k) ** -0.5
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    L = torch.tensor(lens, dtype=torch.int32)
    valid = torch.arange(T)[None, :] < L[:, None]
    xl = r16(F.leaky_relu(x, slope) * valid[:, :, None], dt)
    w1r, w2r = r16(w1, dt), r16(w2, dt)
    ref = torch.zeros(B, T, C)
    for b in range(B):
        n = lens[b]
        if n == 0: continue
        xi = xl[b:b + 1, :n].transpose(1, 2)
        t1 = r16(F.leaky_relu(F.conv1d(xi, w1r, b1, padding=(k - 1) // 2 * dil, dilation=dil), slope), dt)
        xr = torch.where(xi >= 0, xi, xi / slope)
        ref[b, :n] = (F.conv1d(t1, w2r, b2, padding=(k - 1) // 2) + xr)[0].t()
    kw = dict(B=B, T=T, C=C, k=k, dil=dil, slope=slope, lens=L.cuda(), len_mul=1, dtype=dt)
    W1, W2 = pack_conv1d(w1r).to(t16).cuda().contiguous(), pack_conv1d(w2r).to(t16).cuda().contiguous()
    X = xl.reshape(B * T, C).to(t16).cuda().contiguous()
    y = torch.full((B * T, C), 7.0, device="cuda", dtype=t16)
    ops.respair(X, W1, b1.cuda(), W2, b2.cuda(), y=y, **kw)
    e0 = (y.float().cpu().view(B, T, C) - F.leaky_relu(ref, slope)).abs()
    xs = torch.full((B * T, C), 3.0, device="cuda")
    ops.respair(X, W1, b1.cuda(), W2, b2.cuda(), xs=xs, **kw)
    e1 = (xs.cpu().view(B, T, C) - ref).abs()
    print(f"C{C} k{k} d{dil} T{T} lens{lens}: ref max {ref.abs().max():.3f}  mid err {e0.max():.4f}  last err {e1.max():.4f}")
    for name, e in (("mid", e0), ("last", e1)):
        if e.max() > 0.02 * ref.abs().max():
            bad = (e > 0.02 * ref.abs().max())
            idx = bad.nonzero()
            print(f"   {name}: {bad.sum().item()} bad of {bad.numel()}; clips {sorted(set(idx[:,0].tolist()))} t range {idx[:,1].min().item()}..{idx[:,1].max().item()} "
                  f"chan set size {len(set(idx[:,2].tolist()))} first {idx[:6].tolist()}")
            tb = bad.any(2)[idx[0,0]].nonzero().flatten().tolist()
            print("   bad t (first clip w/ errors):", tb[:40], "...", tb[-10:])
            cb = bad.any(1)[idx[0,0]].nonzero().flatten().tolist()
            print("   bad channels:", cb[:64])

for args in [(256, 3, 1, 300, [300, 211]), (256, 7, 3, 257, [257, 40]), (256, 11, 5, 400, [400, 399]), (256, 3, 5, 1, [1, 1]), (256, 11, 3, 129, [128, 129, 118, 117])]:
    case(*args)
