"""Multi-input HiFi-GAN vocoder on gfx950 — host-side mirror of multi_input_vocoder/models_multi_input.py
(`MelCodeGenerator` :26-97) over speech-resynthesis/models.py (`Generator` :72-122, `ResBlock1` :16-47,
`CodeGenerator._upsample` :158-177).

Same constructor (`MelCodeGenerator(h)` with the multi_input.json AttrDict), forward keywords (code, mel, spkr) and
state_dict names (`dict, spkr, layer.0, fc, conv_pre, ups.i, resblocks.j.convs1/2.k, conv_post`, weight-norm keys
`weight_g/weight_v` accepted, `remove_weight_norm()` kept as the reference's call order needs it).
All convolutions are Conv1d-as-GEMM on MFMA (tap-GEMM); ConvTranspose1d runs as `stride` phase GEMMs; the LeakyReLU
feeding each conv is produced by the previous epilogue (dual store), the sum of the three ResBlocks accumulates in fp32
and its 1/3 is folded into the following conv's weights (leaky_relu is positively homogeneous).
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU, ACT_LRELU, F_ACCUM, F_DUAL, F_MASK, F_RES_POST, MODE_CONV1D
from .packing import convtranspose_fused, convtranspose_phases, pack_conv1d

LRELU_SLOPE = 0.1  # speech-resynthesis/models.py:13
# C = 64 / 128 stages: one launch per (c1, c2) conv pair out of LDS (csrc/respair.hip); 0 = the unfused tap-GEMM pairs (A/B)
FUSED_PAIR = os.environ.get("L2S_RESPAIR", "1") != "0"
# ... and the C = 256 stage on the phase-staggered pair kernel (csrc/respair256.hip); 0 = its 18 tap-GEMM launches (A/B)
FUSED_PAIR256 = os.environ.get("L2S_RESPAIR256", "1") != "0"
# The pair kernels walk B * ceil(T / rows) time tiles, one per CU (rows = 128 / 256 / 512 at C = 256 / 128 / 64).  With fewer than
# PAIR_MIN_TILES of them - one clip per request: 16-40 tiles - a pair is two K loops in a row on a handful of CUs and the tap-GEMM
# launches, whose 64 x 64 tiles cut rows AND channels, are faster: batch-1 latency 4.50 -> 4.15 ms per 4-s clip.
PAIR_MIN_TILES = int(os.environ.get("L2S_RESPAIR_MIN_TILES", "64"))
PAIR_ROWS = {256: 128, 128: 256, 64: 512}
# ConvTranspose1d of the late stages (Cin <= 128, HBM-bound) as ONE launch with N = stride*Cout instead of `stride` phase
# launches that each re-read the input (packing.convtranspose_fused); 0 = the phase launches everywhere (A/B)
FUSED_UPS = os.environ.get("L2S_FUSED_UPS", "1") != "0"
FUSED_UPS_MAX_CIN = int(os.environ.get("L2S_FUSED_UPS_MAX_CIN", "128"))
# C = 32 / 16 stages: the three ResBlocks of a stage as one launch (l2s_resstage_fused); 0 = one launch per ResBlock (A/B)
FUSED_STAGE = os.environ.get("L2S_RESSTAGE", "1") != "0"
# wide stages whose fp32 sum nobody reads: the last pairs of the three ResBlocks as ONE launch, the sum kept in accumulators
# (l2s_respair_final); 0 = one launch per last pair with the sum read-modify-written in HBM (A/B)
FUSED_FINAL = os.environ.get("L2S_RESPAIR_FINAL", "1") != "0"
# a stage whose fp32 ResBlock sum nobody reads (every stage but the last: only leaky_relu(xs) travels on) leaves the sum's
# last pass unwritten; 0 = written everywhere (A/B)
SKIP_DEAD_XS = os.environ.get("L2S_SKIP_DEAD_XS", "1") != "0"


class AttrDict(dict):
    """speech-resynthesis/utils.py:77-80."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.__dict__ = self


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)  # speech-resynthesis/utils.py:44-45


class _WN(nn.Module):
    """Parameter holder for a weight-normed conv: exposes weight_g/weight_v until remove_weight_norm(), then weight."""

    def __init__(self, shape, bias_n, g_dim0):
        super().__init__()
        self.weight_g = nn.Parameter(torch.ones(g_dim0, 1, 1))
        self.weight_v = nn.Parameter(torch.zeros(*shape))
        self.bias = nn.Parameter(torch.zeros(bias_n))

    def effective_weight(self):
        if hasattr(self, "weight") and isinstance(getattr(self, "weight"), torch.Tensor):
            return self.weight.detach().float()
        v, g = self.weight_v.detach().float(), self.weight_g.detach().float()
        return v * (g / v.pow(2).sum(dim=(1, 2), keepdim=True).sqrt())

    def remove_weight_norm(self):
        if "weight_g" in self._parameters:
            w = self.effective_weight()
            del self._parameters["weight_g"], self._parameters["weight_v"]
            self.weight = nn.Parameter(w)


class ResBlock1(nn.Module):
    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5)):
        super().__init__()
        self.channels, self.kernel_size, self.dilation = channels, kernel_size, tuple(dilation)
        self.convs1 = nn.ModuleList([_WN((channels, channels, kernel_size), channels, channels) for _ in dilation])
        self.convs2 = nn.ModuleList([_WN((channels, channels, kernel_size), channels, channels) for _ in dilation])

    def remove_weight_norm(self):
        for m in list(self.convs1) + list(self.convs2):
            m.remove_weight_norm()


class Generator(nn.Module):
    """speech-resynthesis/models.py:72-122."""

    def __init__(self, h, dtype=ops.F16):
        super().__init__()
        self.h = h
        if str(h.resblock) != "1":
            raise NotImplementedError("configs/lrs3/multi_input.json uses resblock '1'")
        self.num_kernels = len(h.resblock_kernel_sizes)
        self.num_upsamples = len(h.upsample_rates)
        c0 = h.upsample_initial_channel
        cin = getattr(h, "model_in_dim", None) or h.get("model_in_dim", 128)
        self.conv_pre = _WN((c0, cin, 7), c0, c0)
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            ci, co = c0 // (2 ** i), c0 // (2 ** (i + 1))
            self.ups.append(_WN((ci, co, k), co, ci))  # ConvTranspose1d weight [Cin, Cout, k], weight_norm dim 0
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = c0 // (2 ** (i + 1))
            for k, d in zip(h.resblock_kernel_sizes, h.resblock_dilation_sizes):
                self.resblocks.append(ResBlock1(h, ch, k, d))
        self.conv_post = _WN((1, ch, 7), 1, 1)
        self.dtype = dtype
        self._packed = None

    def remove_weight_norm(self):
        for m in self.ups:
            m.remove_weight_norm()
        for m in self.resblocks:
            m.remove_weight_norm()
        self.conv_pre.remove_weight_norm()
        self.conv_post.remove_weight_norm()
        self._packed = None

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = None
        return r

    # ---- packing -------------------------------------------------------------------------------------------------
    def _pack_generator(self, dev):
        t16 = ops.torch_dtype(self.dtype)
        h = self.h
        third = 1.0 / self.num_kernels
        P = {"pre_w": pack_conv1d(self.conv_pre.effective_weight()).to(dev, t16).contiguous(),
             "pre_b": self.conv_pre.bias.detach().float().to(dev).contiguous(), "stages": []}
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            w = self.ups[i].effective_weight()
            if i > 0:
                w = w * third  # x = xs / num_kernels of the previous stage (models.py:109), folded through leaky_relu
            st = {"u": u, "cout": w.shape[1], "cin": w.shape[0],
                  "phases": [dict(ph, w=ph["w"].to(dev, t16).contiguous()) for ph in
                             convtranspose_phases(w, u, (k - u) // 2)],
                  "b": self.ups[i].bias.detach().float().to(dev).contiguous(), "rbs": []}
            if FUSED_UPS and w.shape[0] <= FUSED_UPS_MAX_CIN:
                # 32 input channels: two time steps per row, so the layer has the 64-channel patch kernel's shape
                fold = 2 if (w.shape[0] == 32 and u * w.shape[1] == 32) else 1
                f = convtranspose_fused(w, u, (k - u) // 2, fold=fold)
                st["fused"] = dict(f, w=f["w"].to(dev, t16).contiguous(), b=st["b"].repeat(u * fold).contiguous())
            for j in range(self.num_kernels):
                rb = self.resblocks[i * self.num_kernels + j]
                convs = []
                for c1, c2, d in zip(rb.convs1, rb.convs2, rb.dilation):
                    convs.append({
                        "w1": pack_conv1d(c1.effective_weight()).to(dev, t16).contiguous(),
                        "b1": c1.bias.detach().float().to(dev).contiguous(), "d": d,
                        "w2": pack_conv1d(c2.effective_weight()).to(dev, t16).contiguous(),
                        "b2": c2.bias.detach().float().to(dev).contiguous()})
                e = {"k": rb.kernel_size, "convs": convs, "dil": tuple(rb.dilation)}
                C = w.shape[1]
                if C in (16, 32) and rb.kernel_size in (3, 7, 11) and len(rb.dilation) == 3:
                    # fused-ResBlock layout: [6][C][Kpad] in the order c1(d0), c2, c1(d1), c2, c1(d2), c2
                    kpad = ((rb.kernel_size * C + 31) // 32) * 32
                    wf = torch.zeros(6, C, kpad)
                    bf = torch.zeros(6, C)
                    for m, (c1, c2) in enumerate(zip(rb.convs1, rb.convs2)):
                        for q, cv in enumerate((c1, c2)):
                            wf[2 * m + q, :, : rb.kernel_size * C] = pack_conv1d(cv.effective_weight())
                            bf[2 * m + q] = cv.bias.detach().float()
                    e["fw"] = wf.to(dev, t16).contiguous()
                    e["fb"] = bf.to(dev).contiguous()
                st["rbs"].append(e)
            P["stages"].append(st)
        wp = self.conv_post.effective_weight()[0] * third                   # [C, 7]
        P["post_w"] = wp.t().contiguous().to(dev)                           # [7, C] fp32
        P["post_b"] = float(self.conv_post.bias.detach().float()[0])
        return P

    def _pack_precise(self, dev):
        """fp32 weights as (hi, lo) pairs in the tap-GEMM layouts; same folds as _pack_generator (1 / num_kernels into the next
        stage's ups and conv_post)."""
        t16 = ops.torch_dtype(self.dtype)
        h, third = self.h, 1.0 / self.num_kernels
        P = {"pre_w": _hi_lo(pack_conv1d(self.conv_pre.effective_weight()), t16, dev),
             "pre_b": self.conv_pre.bias.detach().float().to(dev).contiguous(), "stages": []}
        for i, (u, k) in enumerate(zip(h.upsample_rates, h.upsample_kernel_sizes)):
            w = self.ups[i].effective_weight()
            if i > 0:
                w = w * third
            st = {"u": u, "cout": w.shape[1], "cin": w.shape[0],
                  "phases": [dict(ph, w=_hi_lo(ph["w"], t16, dev)) for ph in convtranspose_phases(w, u, (k - u) // 2)],
                  "b": self.ups[i].bias.detach().float().to(dev).contiguous(), "rbs": []}
            for j in range(self.num_kernels):
                rb = self.resblocks[i * self.num_kernels + j]
                st["rbs"].append({"k": rb.kernel_size, "convs": [
                    {"w1": _hi_lo(pack_conv1d(c1.effective_weight()), t16, dev), "b1": c1.bias.detach().float().to(dev).contiguous(),
                     "d": d, "w2": _hi_lo(pack_conv1d(c2.effective_weight()), t16, dev),
                     "b2": c2.bias.detach().float().to(dev).contiguous()}
                    for c1, c2, d in zip(rb.convs1, rb.convs2, rb.dilation)]})
            P["stages"].append(st)
        P["post_w"] = (self.conv_post.effective_weight()[0] * third).t().contiguous().to(dev)
        P["post_b"] = float(self.conv_post.bias.detach().float()[0])
        return P

    def generator_rows_precise(self, cat32, lens, B, T0, base_mul):
        """speech-resynthesis/models.py:98-114 at reference precision (see _PreciseOps).  cat32: fp32 [B*T0, Cin] generator
        input rows; returns (wav fp32 [B, T0*160], pcm int16)."""
        dev = cat32.device
        if getattr(self, "_packed_precise", None) is None or self._packed_precise["pre_b"].device != dev:
            self._packed_precise = self._pack_precise(dev)
        P, po = self._packed_precise, _PreciseOps(self.dtype, dev)
        Cin, c0 = cat32.shape[1], P["pre_b"].shape[0]
        T, mul = T0, base_mul
        a = po.split(cat32, B, T, Cin, lens=lens, len_mul=mul)
        x = torch.empty(B * T, c0, device=dev, dtype=torch.float32)
        po.gemm3(a, P["pre_w"], x, bias=P["pre_b"], M=B * T, N=c0, Cin=Cin, ntaps=7, mode=MODE_CONV1D, T_out=T, T_in=T,
                 stride=1, dil=1, off=-3)                                                              # conv_pre :99
        xs = None
        for st in P["stages"]:
            u, C = st["u"], st["cout"]
            To, M_in = T * u, B * T
            a = po.split(x if xs is None else xs, B, T, st["cin"], act=1, slope=LRELU_SLOPE, lens=lens, len_mul=mul)   # :101
            mul *= u
            M = B * To
            x = torch.empty(M, C, device=dev, dtype=torch.float32)
            for ph in st["phases"]:                                                                     # ups[i] :102
                po.gemm3(a, ph["w"], x, bias=st["b"], M=M_in, N=C, Cin=st["cin"], ntaps=ph["ntaps"], mode=MODE_CONV1D, T_out=T,
                         T_in=T, stride=1, dil=-1, off=ph["off"], out_row_mul=u, out_row_add=ph["r"])
            xs = torch.zeros(M, C, device=dev, dtype=torch.float32)
            for rb in st["rbs"]:                                                                        # :103-108
                k, cur = rb["k"], x
                for m, cv in enumerate(rb["convs"]):
                    d = cv["d"]
                    a1 = po.split(cur, B, To, C, act=1, slope=LRELU_SLOPE, lens=lens, len_mul=mul)
                    t1 = torch.empty(M, C, device=dev, dtype=torch.float32)
                    po.gemm3(a1, cv["w1"], t1, bias=cv["b1"], M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=To, T_in=To,
                             stride=1, dil=d, off=-get_padding(k, d))
                    a2 = po.split(t1, B, To, C, act=1, slope=LRELU_SLOPE, lens=lens, len_mul=mul)
                    last = m == len(rb["convs"]) - 1
                    o = xs if last else torch.empty(M, C, device=dev, dtype=torch.float32)
                    # xt = c2(.) + x (models.py:39-40); the block's last pair adds straight into the stage sum (:105-108)
                    po.gemm3(a2, cv["w2"], o, bias=cv["b2"], R=cur, ldr=C, accumulate=last, M=M, N=C, Cin=C, ntaps=k,
                             mode=MODE_CONV1D, T_out=To, T_in=To, stride=1, dil=1, off=-get_padding(k, 1))
                    cur = o
            T = To
        wav = torch.empty(B, T, device=dev, dtype=torch.float32)
        pcm = torch.empty(B, T, device=dev, dtype=torch.int16)
        ops.conv_post_tanh(xs, P["post_w"], P["post_b"], wav, pcm, B=B, T=T, C=xs.shape[1], k=7, lens=lens, len_mul=mul)
        return wav, pcm

    def generator_rows(self, x_l, lens, B, T0, base_mul):
        """x_l: [B*T0, 512] 16-bit = leaky_relu(conv_pre(x)) rows; lens: int32 [B] in code frames; base_mul: rows per code
        frame at this rate (2).  Returns (wav fp32 [B, T0*160], pcm int16)."""
        P, dt = self._packed["gen"], self.dtype
        t16 = ops.torch_dtype(dt)
        dev = x_l.device
        T, mul = T0, base_mul
        xs = None
        for si, st in enumerate(P["stages"]):
            u, C = st["u"], st["cout"]
            To, M_in = T * u, B * T
            mul *= u
            M = B * To
            pair_stage = FUSED_PAIR and (C in (64, 128) or (C == 256 and FUSED_PAIR256)) and all(
                rb["k"] <= 11 and max(rb["dil"]) * (rb["k"] - 1) // 2 <= (28 if C == 256 else 32) for rb in st["rbs"])
            pair_stage = pair_stage and B * -(-To // PAIR_ROWS[C]) >= PAIR_MIN_TILES
            fused_stage = all("fw" in rb for rb in st["rbs"])
            xl = torch.empty(M, C, device=dev, dtype=t16)     # leaky_relu(x) (input of every ResBlock)
            if (pair_stage or fused_stage) and "fused" in st:
                # one launch: GEMM row q, column r*C + c is sample q*u + r, channel c - the [M_in, u*C] output IS xl
                x, f = None, st["fused"]
                fd = f["fold"] if T % f["fold"] == 0 and (mul // u) % f["fold"] == 0 else 0
                if fd == 0:
                    raise ops.L2SError("folded ConvTranspose needs an even number of input steps per clip")
                ops.tapgemm(x_l, f["w"], xl, M=M_in // fd, N=f["n"], Cin=st["cin"] * fd, ntaps=f["ntaps"], mode=MODE_CONV1D,
                            T_out=T // fd, T_in=T // fd, stride=1, dil=-1, off=f["off"], bias=f["b"], act=ACT_LRELU,
                            act_slope=LRELU_SLOPE, lens=lens, mask_T=T // fd, mask_mul=mul // u // fd, flags=F_MASK, dtype=dt)
            elif pair_stage or fused_stage:
                # those kernels recover the residual x from leaky_relu(x): the raw ups output is never stored
                x = None
                for ph in st["phases"]:
                    ops.tapgemm(x_l, ph["w"], xl, M=M_in, N=C, Cin=st["cin"], ntaps=ph["ntaps"], mode=MODE_CONV1D, T_out=T,
                                T_in=T, stride=1, dil=-1, off=ph["off"], out_row_mul=u, out_row_add=ph["r"], bias=st["b"],
                                act=ACT_LRELU, act_slope=LRELU_SLOPE, lens=lens, mask_T=To, mask_mul=mul, flags=F_MASK,
                                dtype=dt)
            else:
                x = torch.empty(M, C, device=dev, dtype=t16)  # raw ups output (residual of every ResBlock)
                for ph in st["phases"]:
                    ops.tapgemm(x_l, ph["w"], x, M=M_in, N=C, Cin=st["cin"], ntaps=ph["ntaps"], mode=MODE_CONV1D, T_out=T,
                                T_in=T, stride=1, dil=-1, off=ph["off"], out_row_mul=u, out_row_add=ph["r"], bias=st["b"],
                                C2=xl, ldc2=C, lens=lens, mask_T=To, mask_mul=mul, flags=F_DUAL | F_MASK,
                                slope2=LRELU_SLOPE, dtype=dt)
            nxt = torch.empty(M, C, device=dev, dtype=t16)
            last_stage = si == len(P["stages"]) - 1
            # (the ResBlocks' last pairs in one launch keep the stage's fp32 sum in accumulators: no xs buffer then)
            final_fused = pair_stage and FUSED_FINAL and SKIP_DEAD_XS and not last_stage and len(st["rbs"]) <= 3
            xs = None if final_fused else torch.empty(M, C, device=dev, dtype=torch.float32)
            if pair_stage:
                # wide stages (C = 256, 128, 64): one launch per conv pair, the activation travels as its LeakyReLU'd copy only
                finals = []
                for j, rb in enumerate(st["rbs"]):
                    cur_l = xl
                    for m, cv in enumerate(rb["convs"]):
                        if final_fused and m == len(rb["convs"]) - 1:
                            finals.append((cur_l, cv, rb["k"]))      # the ResBlocks' last pairs go in one launch below
                        elif m < len(rb["convs"]) - 1:
                            ol = torch.empty(M, C, device=dev, dtype=t16)
                            ops.respair(cur_l, cv["w1"], cv["b1"], cv["w2"], cv["b2"], B=B, T=To, C=C, k=rb["k"], dil=cv["d"],
                                        slope=LRELU_SLOPE, y=ol, lens=lens, len_mul=mul, dtype=dt)
                            cur_l = ol
                        else:
                            dual = (j == len(st["rbs"]) - 1) and not last_stage
                            ops.respair(cur_l, cv["w1"], cv["b1"], cv["w2"], cv["b2"], B=B, T=To, C=C, k=rb["k"], dil=cv["d"],
                                        slope=LRELU_SLOPE, xs=xs, y=nxt if dual else None, accumulate=j > 0, lens=lens,
                                        len_mul=mul, dtype=dt, xs_final=not (dual and SKIP_DEAD_XS))   # only nxt travels on
                if finals:
                    ops.respair_final([f[0] for f in finals], [f[1]["w1"] for f in finals], [f[1]["b1"] for f in finals],
                                      [f[1]["w2"] for f in finals], [f[1]["b2"] for f in finals], nxt, B=B, T=To, C=C,
                                      ks=[f[2] for f in finals], dils=[f[1]["d"] for f in finals], slope=LRELU_SLOPE, lens=lens,
                                      len_mul=mul, dtype=dt)
                x_l, T = nxt, To
                continue
            if fused_stage:
                # narrow stages (C = 32, 16): each ResBlock is ONE launch working out of LDS (csrc/resblock.hip)
                if FUSED_STAGE and [rb["k"] for rb in st["rbs"]] == [3, 7, 11]:
                    # ... and the stage's three ResBlocks are one launch: xl is read once, xs never round-trips HBM
                    ops.resstage_fused(xl, [rb["fw"] for rb in st["rbs"]], [rb["fb"] for rb in st["rbs"]], xs,
                                       None if last_stage else nxt, B=B, T=To, C=C, ks=[rb["k"] for rb in st["rbs"]],
                                       dils=[rb["dil"] for rb in st["rbs"]], slope=LRELU_SLOPE, lens=lens, len_mul=mul,
                                       dtype=dt, xs_final=last_stage or not SKIP_DEAD_XS)    # conv_post reads the last stage's sum
                    x_l, T = nxt, To
                    continue
                for j, rb in enumerate(st["rbs"]):
                    dual = (j == len(st["rbs"]) - 1) and not last_stage
                    ops.resblock_fused(xl, rb["fw"], rb["fb"], xs, nxt if dual else None, B=B, T=To, C=C, k=rb["k"],
                                       dil=rb["dil"], accumulate=j > 0, slope=LRELU_SLOPE, lens=lens, len_mul=mul, dtype=dt)
                x_l, T = nxt, To
                continue
            t1 = torch.empty(M, C, device=dev, dtype=t16)
            for j, rb in enumerate(st["rbs"]):
                k = rb["k"]
                cur, cur_l = x, xl
                for m, cv in enumerate(rb["convs"]):
                    d = cv["d"]
                    ops.tapgemm(cur_l, cv["w1"], t1, M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=To, T_in=To,
                                stride=1, dil=d, off=-get_padding(k, d), bias=cv["b1"], act=ACT_LRELU,
                                act_slope=LRELU_SLOPE, lens=lens, mask_T=To, mask_mul=mul, flags=F_MASK, dtype=dt)
                    if m < len(rb["convs"]) - 1:
                        o = torch.empty(M, C, device=dev, dtype=t16)
                        ol = torch.empty(M, C, device=dev, dtype=t16)
                        ops.tapgemm(t1, cv["w2"], o, M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=To, T_in=To,
                                    stride=1, dil=1, off=-get_padding(k, 1), bias=cv["b2"], R=cur, ldr=C, C2=ol, ldc2=C,
                                    lens=lens, mask_T=To, mask_mul=mul, flags=F_RES_POST | F_DUAL | F_MASK,
                                    slope2=LRELU_SLOPE, dtype=dt)
                        cur, cur_l = o, ol
                    else:
                        # last conv of the block: x_j = conv + cur, accumulated into xs (models.py:103-108)
                        fl = F_RES_POST | F_MASK | (F_ACCUM if j > 0 else 0)
                        dual = (j == len(st["rbs"]) - 1) and not last_stage
                        ops.tapgemm(t1, cv["w2"], xs, M=M, N=C, Cin=C, ntaps=k, mode=MODE_CONV1D, T_out=To, T_in=To,
                                    stride=1, dil=1, off=-get_padding(k, 1), bias=cv["b2"], R=cur, ldr=C,
                                    C2=nxt if dual else None, ldc2=C, lens=lens, mask_T=To, mask_mul=mul,
                                    flags=fl | (F_DUAL if dual else 0), slope2=LRELU_SLOPE, dtype=dt)
            x_l, T = nxt, To
        wav = torch.empty(B, T, device=dev, dtype=torch.float32)
        pcm = torch.empty(B, T, device=dev, dtype=torch.int16)
        ops.conv_post_tanh(xs, P["post_w"], P["post_b"], wav, pcm, B=B, T=T, C=xs.shape[1], k=7, lens=lens, len_mul=mul)
        return wav, pcm


def _hi_lo(w, t16, dev):
    """fp32 weight -> (hi, lo) 16-bit pair with hi + lo = w up to 2 x the 16-bit mantissa."""
    w = w.detach().float()
    hi = w.to(t16)
    lo = (w - hi.float()).to(t16)
    return hi.to(dev).contiguous(), lo.to(dev).contiguous()


class _PreciseOps:
    """Reference-precision arithmetic for the parity switch `precise=True` (the reference's vocoder runs fp32,
    multi_input_vocoder/inference.py:73-82): every layer input is an fp32 array split into hi + lo 16-bit arrays
    (l2s_split_hi_lo, which also applies the layer's input activation and row mask), every layer is three tap-GEMM launches
    into one fp32 output (a_hi w_hi [+ bias, + fp32 residual], then a_lo w_hi and a_hi w_lo accumulated).  Slow by design
    (generic tiles, fp32 activations through HBM, 3 x the MFMA work): it exists to separate precision from ordering in the
    PCM error of the 16-bit path, not to be fast."""

    def __init__(self, dtype, dev):
        self.dt, self.t16, self.dev = dtype, ops.torch_dtype(dtype), dev

    def split(self, x, B, T, C, act=0, slope=0.0, lens=None, len_mul=1, ldx=None):
        hi = torch.empty(B * T, C, device=self.dev, dtype=self.t16)
        lo = torch.empty(B * T, C, device=self.dev, dtype=self.t16)
        ops.split_hi_lo(x, hi, lo, B=B, T=T, C=C, act=act, slope=slope, ldx=ldx, lens=lens, len_mul=len_mul, dtype=self.dt)
        return hi, lo

    def gemm3(self, a, w, out, *, bias=None, R=None, accumulate=False, **kw):
        """out (fp32) [+]= a . w^T (+ bias) (+ R fp32); a = (hi, lo), w = (hi, lo)."""
        fl = (F_RES_POST if R is not None else 0) | (F_ACCUM if accumulate else 0)
        ops.tapgemm(a[0], w[0], out, bias=bias, R=R, flags=fl, dtype=self.dt, **kw)
        ops.tapgemm(a[1], w[0], out, flags=F_ACCUM, dtype=self.dt, **kw)
        ops.tapgemm(a[0], w[1], out, flags=F_ACCUM, dtype=self.dt, **kw)


class MelCodeGenerator(Generator):
    """multi_input_vocoder/models_multi_input.py:26-97 (text_supervision branch not built)."""

    def __init__(self, h, dtype=ops.F16):
        super().__init__(h, dtype=dtype)
        if h.get("text_supervision", False):
            raise NotImplementedError("TEXT_SUPERVISION=1 vocoder branch is outside the lip2speech inference path")
        E = h.embedding_dim
        self.dict = nn.Embedding(h.num_embeddings, E)
        self.multispkr = h.get("multispkr", None)
        embedder_dim = h.get("embedder_dim", None)
        if not embedder_dim:
            raise NotImplementedError("speaker-id embedding table variant is not used by configs/lrs3/multi_input.json")
        self.spkr = nn.Linear(embedder_dim, E)
        self.layer = nn.Sequential(nn.ConvTranspose1d(E, E, kernel_size=4, stride=2, padding=1), nn.GELU())
        self.fc = nn.Linear(E, E)
        self.num_mels = h.get("num_mels", 80)

    def pack(self, dev):
        t16 = ops.torch_dtype(self.dtype)
        P = {"gen": self._pack_generator(dev)}
        P["table"] = self.dict.weight.detach().to(dev, t16).contiguous()
        ct = self.layer[0]
        P["up_phases"] = [dict(ph, w=ph["w"].to(dev, t16).contiguous())
                          for ph in convtranspose_phases(ct.weight.detach().float(), 2, 1)]
        P["up_b"] = ct.bias.detach().float().to(dev).contiguous()
        P["fc_w"], P["fc_b"] = self.fc.weight.detach().to(dev, t16).contiguous(), self.fc.bias.detach().float().to(dev).contiguous()
        P["sp_w"], P["sp_b"] = self.spkr.weight.detach().to(dev, t16).contiguous(), self.spkr.bias.detach().float().to(dev).contiguous()
        self._packed = P

    def forward_rows(self, code, mel, spkr, lens=None):
        """code int [B,L]; mel fp32 [B,80,2L]; spkr fp32 [B,256]; lens int32 [B] valid code frames (None = all).
        Returns (wav fp32 [B, 320L], pcm int16 [B, 320L])."""
        dev = code.device
        B, L = code.shape
        nm = self.num_mels
        assert mel.shape == (B, nm, 2 * L), f"mel {tuple(mel.shape)} vs code {tuple(code.shape)}"
        if lens is None:
            lens = torch.full((B,), L, device=dev, dtype=torch.int32)

        def fill(P, cat, emb, Cin, dt):
            ops.embedding(code.to(torch.int32).contiguous(), P["table"], emb, B=B, L=L, C=self.h.embedding_dim, lens=lens,
                          dtype=dt)                                                                      # :67
            ops.transpose_ct_to_tc(mel.float().contiguous(), cat, B=B, C=nm, T=2 * L, ldy=Cin, col0=0, lens=lens, len_mul=2,
                                   dtype=dt)                                                             # :65,:73
        return self._rows(fill, spkr, lens, 1, B, L, dev)

    def forward_rows_precise(self, code, mel, spkr, lens=None):
        """forward_rows at reference precision (parity switch, see _PreciseOps): models_multi_input.py:60-97 with every Linear /
        ConvTranspose as three hi / lo tap-GEMM launches; embedding lookup, concatenation and the layout transposes are torch
        data movement (no arithmetic)."""
        dev = code.device
        B, L = code.shape
        E, nm, t16 = self.h.embedding_dim, self.num_mels, ops.torch_dtype(self.dtype)
        T0 = 2 * L
        if lens is None:
            lens = torch.full((B,), L, device=dev, dtype=torch.int32)
        po = _PreciseOps(self.dtype, dev)
        if getattr(self, "_front_precise", None) is None or self._front_precise["up_b"].device != dev:
            ct = self.layer[0]
            self._front_precise = {
                "up": [dict(ph, w=_hi_lo(ph["w"], t16, dev)) for ph in convtranspose_phases(ct.weight.detach().float(), 2, 1)],
                "up_b": ct.bias.detach().float().to(dev).contiguous(),
                "fc": _hi_lo(self.fc.weight, t16, dev), "fc_b": self.fc.bias.detach().float().to(dev).contiguous(),
                "sp": _hi_lo(self.spkr.weight, t16, dev), "sp_b": self.spkr.bias.detach().float().to(dev).contiguous()}
        F = self._front_precise
        emb = self.dict.weight.detach().float().to(dev)[code.long()].reshape(B * L, E).contiguous()           # :67
        a = po.split(emb, B, L, E, lens=lens, len_mul=1)
        up = torch.empty(B * T0, E, device=dev, dtype=torch.float32)
        for ph in F["up"]:                                                                                    # :68 ConvTranspose1d
            po.gemm3(a, ph["w"], up, bias=F["up_b"], M=B * L, N=E, Cin=E, ntaps=ph["ntaps"], mode=MODE_CONV1D, T_out=L, T_in=L,
                     stride=1, dil=-1, off=ph["off"], out_row_mul=2, out_row_add=ph["r"])
        a = po.split(up, B, T0, E, act=2, lens=lens, len_mul=2)                                               # :68 GELU
        cat = torch.zeros(B * T0, nm + 2 * E, device=dev, dtype=torch.float32)
        po.gemm3(a, F["fc"], cat[:, nm:], bias=F["fc_b"], M=B * T0, N=E, Cin=E, ldc=nm + 2 * E)               # :70-73
        cat[:, :nm] = mel.float().transpose(1, 2).reshape(B * T0, nm)                                         # :65,:73
        sp_in = po.split(spkr.float().contiguous(), B, 1, spkr.shape[1])
        sp = torch.empty(B, E, device=dev, dtype=torch.float32)
        po.gemm3(sp_in, F["sp"], sp, bias=F["sp_b"], M=B, N=E, Cin=spkr.shape[1])                             # :80
        cat[:, nm + E:] = sp.repeat_interleave(T0, dim=0)                                                     # :81-82
        return self.generator_rows_precise(cat, lens, B, T0, 2)

    def forward_tokens_rows(self, tokens, mel_rows, spkr, src_lens, token_offset=4):
        """The in-memory hand-off from stage 1, with no layout or arithmetic left to torch: `tokens` int32
        [B, >= L] are the generator's token rows (unit u = token u + `token_offset`: fairseq's 4 specials come first, and
        dict.unt.txt lists the units in order), `mel_rows` fp32 [B, 2L, 80] the mel head's time-major output
        (model_avhubert.py:276), `src_lens` int32 [B] the clips' VIDEO frame counts (L = 2 x the padded frame count: a unit
        per 20 ms, sequence_generator.py:109).  Equivalent to forward_rows(tokens[:, :L] - 4, mel_rows.transpose(1, 2), spkr,
        2 * src_lens), the file round trip of inference.py:267-274 -> create_dataset.py:366-428 -> dataset_multi_input.py:
        198-291 whose trimming rule cut = min(mel_len * 160, code_len * 320) is the identity here."""
        dev = tokens.device
        B = tokens.shape[0]
        nm = self.num_mels
        L = mel_rows.shape[1] // 2
        assert mel_rows.shape == (B, 2 * L, nm) and mel_rows.dtype == torch.float32 and mel_rows.is_contiguous()
        assert tokens.dtype == torch.int32 and tokens.shape[1] >= L and tokens.stride(1) == 1

        def fill(P, cat, emb, Cin, dt):
            ops.embedding_tokens(tokens, P["table"], emb, B=B, L=L, C=self.h.embedding_dim, token_offset=token_offset,
                                 ldt=tokens.stride(0), lens=src_lens, len_mul=2, dtype=dt)
            ops.rows_f32_to_16_masked(mel_rows, cat, B=B, T=2 * L, C=nm, ldy=Cin, col0=0, lens=src_lens, len_mul=4, dtype=dt)
        return self._rows(fill, spkr, src_lens, 2, B, L, dev)

    def _rows(self, fill, spkr, lens, lm, B, L, dev):
        """models_multi_input.py:60-97 on channels-last rows.  `fill(P, cat, emb, Cin, dt)` writes the unit embeddings and the
        mel columns of the concat buffer; `lens` counts units of 1 / lm code frames (lm = 1: code frames, 2: video frames)."""
        if self._packed is None or self._packed["table"].device != dev:
            self.pack(dev)
        P, dt, h = self._packed, self.dtype, self.h
        t16 = ops.torch_dtype(dt)
        E, nm = h.embedding_dim, self.num_mels
        T0 = 2 * L
        Cin = nm + 2 * E
        cat = torch.empty(B * T0, Cin, device=dev, dtype=t16)
        emb = torch.empty(B * L, E, device=dev, dtype=t16)
        fill(P, cat, emb, Cin, dt)
        up = torch.empty(B * T0, E, device=dev, dtype=t16)
        for ph in P["up_phases"]:                                                                        # :68
            ops.tapgemm(emb, ph["w"], up, M=B * L, N=E, Cin=E, ntaps=ph["ntaps"], mode=MODE_CONV1D, T_out=L, T_in=L,
                        stride=1, dil=-1, off=ph["off"], out_row_mul=2, out_row_add=ph["r"], bias=P["up_b"],
                        act=ACT_GELU, lens=lens, mask_T=T0, mask_mul=2 * lm, flags=F_MASK, dtype=dt)
        ops.tapgemm(up, P["fc_w"], cat[:, nm:], M=B * T0, N=E, Cin=E, ldc=Cin, bias=P["fc_b"], lens=lens, mask_T=T0,
                    mask_mul=2 * lm, flags=F_MASK, dtype=dt)                                             # :70-73
        spkr = spkr.contiguous()
        if spkr.dtype == torch.float32:
            sp16 = torch.empty(B, spkr.shape[1], device=dev, dtype=t16)
            ops.cast_f32_to_16(spkr, sp16, B, spkr.shape[1], dt)
        else:
            sp16 = spkr.to(t16)
        sp = torch.empty(B, E, device=dev, dtype=t16)
        ops.tapgemm(sp16, P["sp_w"], sp, M=B, N=E, Cin=sp16.shape[1], bias=P["sp_b"], dtype=dt)          # :80
        ops.broadcast_rows(sp, cat, B=B, T=T0, C=E, ldy=Cin, col0=nm + E, lens=lens, len_mul=2 * lm, dtype=dt)  # :81-82
        G = P["gen"]
        c0 = G["pre_b"].shape[0]
        x_l = torch.empty(B * T0, c0, device=dev, dtype=t16)
        # conv_pre (models.py:99) + the leaky_relu that opens the first upsample stage (:101)
        ops.tapgemm(cat, G["pre_w"], x_l, M=B * T0, N=c0, Cin=Cin, ntaps=7, mode=MODE_CONV1D, T_out=T0, T_in=T0,
                    stride=1, dil=1, off=-3, bias=G["pre_b"], act=ACT_LRELU, act_slope=LRELU_SLOPE, lens=lens,
                    mask_T=T0, mask_mul=2 * lm, flags=F_MASK, dtype=dt)
        return self.generator_rows(x_l, lens, B, T0, 2 * lm)

    def forward(self, **kwargs):
        """models_multi_input.py:60-97: returns waveform [B,1,320L] in (-1,1)."""
        wav, _ = self.forward_rows(kwargs["code"], kwargs["mel"], kwargs["spkr"], kwargs.get("lens"))
        return wav.unsqueeze(1)
