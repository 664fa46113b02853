"""ResNet-18 3D/2D visual frontend on gfx950 — host-side mirror of avhubert/resnet.py.

Class names, constructor arguments, forward signature and state_dict key layout follow the reference
(`ResEncoder` avhubert/resnet.py:131-169, `ResNet` :77-129, `BasicBlock` :35-74) so a reference checkpoint loads
unchanged; the arithmetic runs in liblip2speech_hip.so (stem implicit-GEMM kernel + tap-GEMM convs, BatchNorm folded).
The torch.nn layers below only hold parameters: their own forward() is never called.
"""
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_PRELU, ACT_SWISH, F_RES_PRE, MODE_CONV2D


def _fold_bn(bn: nn.Module):
    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
    return scale, shift


FUSED_BASICBLOCK = os.environ.get("L2S_BASICBLOCK", "1") != "0"   # A/B switch: 0 = two patch-kernel launches per BasicBlock
FUSED_BASICLAYER = os.environ.get("L2S_BASICLAYER", "1") != "0"   # A/B switch: 0 = one launch per BasicBlock of layer1
FUSED_BASICBLOCK128 = os.environ.get("L2S_BASICBLOCK128", "1") != "0"   # A/B switch: 0 = layer2's second block as two tap-GEMM launches
KTAB_CONV = os.environ.get("L2S_KTAB_CONV", "1") != "0"   # A/B switch: 0 = layer3 / layer4 convolutions as CONV2D tap-GEMMs (padding taps multiplied)
FUSED_STAGE128_TAIL = os.environ.get("L2S_STAGE128_TAIL", "1") != "0"   # A/B switch: 0 = layer2's downsample / conv2 of block 1 as launches of their own



def ktab_conv3x3(Hi, Wi, Cin, stride, device):
    """K-block table (include/lip2speech_hip.h, l2s_gemm_desc::ktab) of a 3x3 / padding-1 convolution on an Hi x Wi map stored as ONE
    row of A ([images, Hi*Wi*Cin]): output position g = (y, x) lists only the taps that fall inside the map - (A element offset of
    the input position, W element offset of the tap) - so padding is skipped instead of multiplied (avhubert/resnet.py:15-24 on the
    6 x 6 / 3 x 3 maps of layer3 / layer4).  Returns (int32 [Ho*Wo, 20], Ho, Wo, total blocks)."""
    Ho, Wo = (Hi + 2 - 3) // stride + 1, (Wi + 2 - 3) // stride + 1
    rows, total = [], 0
    for y in range(Ho):
        for x in range(Wo):
            a, w = [], []
            for ky in range(3):
                for kx in range(3):
                    iy, ix = stride * y + ky - 1, stride * x + kx - 1
                    if 0 <= iy < Hi and 0 <= ix < Wi:
                        a.append((iy * Wi + ix) * Cin)
                        w.append((ky * 3 + kx) * Cin)
            total += len(a)
            rows.append([len(a), 0] + a + [0] * (9 - len(a)) + w + [0] * (9 - len(w)))
    return torch.tensor(rows, dtype=torch.int32, device=device).contiguous(), Ho, Wo, total

class Swish(nn.Module):
    """espnet/nets/pytorch_backend/transformer/convolution.py:68-73 (x * sigmoid(x)); holds no parameters."""


class BasicBlock(nn.Module):
    """Parameter layout of avhubert/resnet.py:35-74 (relu_type 'prelu' or 'relu') and of ESPnet's
    backbones/modules/resnet.py:44-106 (the same block with relu_type 'swish')."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, relu_type="relu"):
        super().__init__()
        assert relu_type in ("relu", "prelu", "swish")
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        if relu_type == "prelu":
            self.relu1 = nn.PReLU(num_parameters=planes)
            self.relu2 = nn.PReLU(num_parameters=planes)
        elif relu_type == "swish":
            self.relu1, self.relu2 = Swish(), Swish()
        else:
            self.relu1 = nn.ReLU()
            self.relu2 = nn.ReLU()
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        self.inplanes, self.planes = inplanes, planes


def _downsample_basic_block(inplanes, outplanes, stride):
    # avhubert/resnet.py:20-24 (the default, non-avgpool variant)
    return nn.Sequential(nn.Conv2d(inplanes, outplanes, 1, stride, bias=False), nn.BatchNorm2d(outplanes))


class ResNet(nn.Module):
    """Trunk of avhubert/resnet.py:77-129: layers [2,2,2,2], planes 64/128/256/512, global average pool."""

    def __init__(self, block=BasicBlock, layers=(2, 2, 2, 2), relu_type="relu"):
        super().__init__()
        self.inplanes = 64
        self.relu_type = relu_type
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = _downsample_basic_block(self.inplanes, planes * block.expansion, stride)
        mods = [block(self.inplanes, planes, stride, downsample, relu_type=self.relu_type)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            mods.append(block(self.inplanes, planes, relu_type=self.relu_type))
        return nn.Sequential(*mods)


class ResEncoder(nn.Module):
    """avhubert/resnet.py:131-169.  forward(x[B,1,T,88,88]) -> [B,512,T]."""

    def __init__(self, relu_type="prelu", weights=None, dtype=ops.F16):
        super().__init__()
        if weights is not None:
            raise NotImplementedError("standalone frontend checkpoints are loaded through load_state_dict")
        self.frontend_nout = 64
        self.backend_out = 512
        act = nn.PReLU(num_parameters=64) if relu_type == "prelu" else (Swish() if relu_type == "swish" else nn.ReLU())
        self.frontend3D = nn.Sequential(
            nn.Conv3d(1, 64, (5, 7, 7), (1, 2, 2), (2, 3, 3), bias=False), nn.BatchNorm3d(64), act,
            nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1)))
        self.trunk = ResNet(BasicBlock, [2, 2, 2, 2], relu_type=relu_type)
        self.relu_type = relu_type
        self.dtype = dtype
        self.u8_transform = (88, 0.421, 0.165)   # image_crop_size / image_mean / image_std (hubert_pretraining.py config)
        self._packed = None

    # ---- packing -------------------------------------------------------------------------------------------------
    def _slopes(self, mod, n, dev):
        if isinstance(mod, nn.PReLU):
            return mod.weight.detach().float().to(dev).contiguous()
        return torch.zeros(n, device=dev)  # ReLU == PReLU with slope 0

    def pack(self, device=None):
        """Fold eval-mode BatchNorm into 16-bit [N, taps*Cin] weights + fp32 bias; call again after loading weights."""
        dev = torch.device(device) if device is not None else self.frontend3D[0].weight.device
        t16 = ops.torch_dtype(self.dtype)
        P = {}
        w = self.frontend3D[0].weight.detach().float().to(dev)  # [64,1,5,7,7]
        sc, sh = _fold_bn(self.frontend3D[1])
        w = w[:, 0] * sc.to(dev)[:, None, None, None]
        wp = torch.zeros(64, 36, 8, device=dev)
        wp[:, :35, :7] = w.reshape(64, 35, 7)
        P["stem_w"] = wp.reshape(64, 288).to(t16).contiguous()
        P["stem_b"] = sh.to(dev).contiguous()
        P["stem_s"] = None if self.relu_type == "swish" else self._slopes(self.frontend3D[2], 64, dev)
        blocks = []
        for layer in (self.trunk.layer1, self.trunk.layer2, self.trunk.layer3, self.trunk.layer4):
            for blk in layer:
                e = {"cin": blk.inplanes, "cout": blk.planes, "stride": blk.stride}
                for i, (conv, bn) in enumerate(((blk.conv1, blk.bn1), (blk.conv2, blk.bn2)), 1):
                    sc, sh = _fold_bn(bn)
                    wt = conv.weight.detach().float().to(dev) * sc.to(dev)[:, None, None, None]
                    e[f"w{i}"] = wt.permute(0, 2, 3, 1).reshape(wt.shape[0], -1).to(t16).contiguous()
                    e[f"b{i}"] = sh.to(dev).contiguous()
                e["s1"] = self._slopes(blk.relu1, blk.planes, dev)
                e["s2"] = self._slopes(blk.relu2, blk.planes, dev)
                if blk.downsample is not None:
                    sc, sh = _fold_bn(blk.downsample[1])
                    wt = blk.downsample[0].weight.detach().float().to(dev)[:, :, 0, 0] * sc.to(dev)[:, None]
                    e["wd"] = wt.to(t16).contiguous()
                    e["bd"] = sh.to(dev).contiguous()
                    if blk.planes == 128 and blk.inplanes == 64:
                        # l2s_basicstage128_tail_fused: the downsample rides behind conv2's nine taps as one more K-tile (+ 64 zero
                        # columns: a whole number of ring turns), its bias joins conv2's
                        e["wa"] = torch.cat([e["w2"], e["wd"], torch.zeros(128, 64, device=dev, dtype=t16)], dim=1).contiguous()
                        e["ba"] = (e["b2"] + e["bd"]).contiguous()
                blocks.append(e)
        P["blocks"] = blocks
        self._packed = P
        return self

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = None
        return r

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward_rows(self, x):
        """x: [B,1,T,88,88] or [B,T,88,88], fp32 or 16-bit - or uint8 [B,(1,)T,Hin,Win] raw frames, cropped and normalised
        with `self.u8_transform` = (crop, mean, std) - -> ([B*T, 512] 16-bit rows (b,t), B, T)."""
        if x.dim() == 5:
            assert x.size(1) == 1
            x = x[:, 0]
        B, T, H, W = x.shape
        x = x.contiguous()
        if self._packed is None or self._packed["stem_w"].device != x.device:
            self.pack(x.device)
        P, dt = self._packed, self.dtype
        t16 = ops.torch_dtype(dt)
        dev = x.device
        N = B * T
        cur = torch.empty(N, 22, 22, 64, device=dev, dtype=t16)
        if x.dtype == torch.uint8:
            # raw decoder frames [B,T,Hin,Win]: centre crop + normalise (hubert_dataset.py:242-245) inside the stem's fetch
            crop, mean, std = self.u8_transform
            ops.stem_pool_fused_u8(x, P["stem_w"], P["stem_b"], P["stem_s"], cur, B, T, dt, crop=crop, mean=mean, std=std)
        else:
            if x.dtype not in (torch.float32, t16):
                x = x.float()
            ops.stem_pool_fused(x, P["stem_w"], P["stem_b"], P["stem_s"], cur, B, T, dt)   # resnet.py:137-141 in one launch
        act = ACT_SWISH if self.relu_type == "swish" else ACT_PRELU
        Hc = 22

        def resident(e):   # a BasicBlock the LDS-resident-image kernel takes (csrc/basicblock.hip): layer1 of ResNet-18
            return (FUSED_BASICBLOCK and e["stride"] == 1 and e["cin"] == 64 and e["cout"] == 64 and "wd" not in e
                    and act == ACT_PRELU and (Hc + 2) * (Hc + 2) <= 576)

        def resident128(e):   # ... and the phase-staggered one of the 128-channel stage (csrc/basicblock_phase.hip): layer2, block 2
            return (FUSED_BASICBLOCK128 and e["stride"] == 1 and e["cin"] == 128 and e["cout"] == 128 and "wd" not in e
                    and act == ACT_PRELU and Hc == 11)

        blocks = P["blocks"]
        bi = 0
        while bi < len(blocks):
            e = blocks[bi]
            s, cin, cout = e["stride"], e["cin"], e["cout"]
            Ho = (Hc + 2 - 3) // s + 1
            M = N * Ho * Ho
            if resident(e):
                # layer1 (resnet.py:61-74, 101-118): its BasicBlocks back to back in ONE launch, the image resident in LDS
                run = [e]
                while FUSED_BASICLAYER and len(run) < 4 and bi + len(run) < len(blocks) and resident(blocks[bi + len(run)]):
                    run.append(blocks[bi + len(run)])
                out = torch.empty(M, cout, device=dev, dtype=t16)
                if len(run) == 1:
                    ops.basicblock_fused(cur, e["w1"], e["b1"], e["s1"], e["w2"], e["b2"], e["s2"], out, n_images=N, H=Hc,
                                         W=Hc, dtype=dt)
                else:
                    ops.basiclayer_fused(cur, [t for r in run for t in (r["w1"], r["w2"])],
                                         [t for r in run for t in (r["b1"], r["b2"])],
                                         [t for r in run for t in (r["s1"], r["s2"])], out, n_images=N, H=Hc, W=Hc, dtype=dt)
                cur = out
                bi += len(run)
                continue
            bi += 1
            if resident128(e):
                out = torch.empty(M, cout, device=dev, dtype=t16)
                ops.basicblock_fused(cur, e["w1"], e["b1"], e["s1"], e["w2"], e["b2"], e["s2"], out, n_images=N, H=Hc, W=Hc,
                                     C=128, dtype=dt)
                cur = out
                continue
            h1 = torch.empty(M, cout, device=dev, dtype=t16)

            def conv3x3(x_in, w, b, sl, y, hin, ci, st, tag, res=None):
                # a 3x3 convolution of this block: on the small maps of layer3 / layer4 (image = one row of A) through the K-block
                # table - every output position sums only its in-map taps - else as a CONV2D tap-GEMM
                ho = (hin + 2 - 3) // st + 1
                if (KTAB_CONV and act == ACT_PRELU and hin <= 11 and ho <= 6 and N >= 256 and N % 8 == 0 and ci % 64 == 0
                        and cout >= 256 and cout % 64 == 0):
                    key = ("ktab", tag)
                    if key not in e:
                        tab, _, _, nblk = ktab_conv3x3(hin, hin, ci, st, dev)
                        e[key] = (tab, nblk, b.repeat(ho * ho).contiguous(), sl.repeat(ho * ho).contiguous())
                    tab, nblk, bg, sg = e[key]
                    ops.tapgemm(x_in, w, y, M=N, N=cout, Cin=ci, ntaps=9, lda=hin * hin * ci, ldc=ho * ho * cout, groups=ho * ho,
                                c_gstride=cout, bias=bg, slope=sg, act=act, R=res, ldr=ho * ho * cout,
                                flags=F_RES_PRE if res is not None else 0, dtype=dt, ktab=tab, kflops=2.0 * N * cout * ci * nblk)
                    return
                ops.tapgemm(x_in, w, y, M=N * ho * ho, N=cout, Cin=ci, ntaps=9, mode=MODE_CONV2D, Ho=ho, Wo=ho, Hi=hin, Wi=hin,
                            KW=3, pad=1, stride=st, bias=b, slope=sl, act=act, R=res, ldr=cout,
                            flags=F_RES_PRE if res is not None else 0, dtype=dt)

            conv3x3(cur, e["w1"], e["b1"], e["s1"], h1, Hc, cin, s, 1)
            if (FUSED_STAGE128_TAIL and FUSED_BASICBLOCK128 and "wa" in e and s == 2 and Ho == 11 and Hc == 22 and act == ACT_PRELU
                    and bi < len(blocks) and blocks[bi]["stride"] == 1 and blocks[bi]["cin"] == 128 and blocks[bi]["cout"] == 128
                    and "wd" not in blocks[bi]):
                # layer2 behind its first conv in ONE launch: conv2 of this block + the downsample as its residual, then the next block
                n = blocks[bi]
                out = torch.empty(M, cout, device=dev, dtype=t16)
                ops.basicstage128_tail_fused(cur, h1, e["wa"], e["ba"], e["s2"], n["w1"], n["b1"], n["s1"], n["w2"], n["b2"], n["s2"],
                                             out, n_images=N, H=Ho, W=Ho, dtype=dt)
                cur, Hc = out, Ho
                bi += 1
                continue
            if "wd" in e:
                res = torch.empty(M, cout, device=dev, dtype=t16)
                ops.tapgemm(cur, e["wd"], res, M=M, N=cout, Cin=cin, ntaps=1, mode=MODE_CONV2D, Ho=Ho, Wo=Ho, Hi=Hc,
                            Wi=Hc, KW=1, pad=0, stride=s, bias=e["bd"], act=ACT_NONE, dtype=dt)
            else:
                res = cur
            out = torch.empty(M, cout, device=dev, dtype=t16)
            conv3x3(h1, e["w2"], e["b2"], e["s2"], out, Ho, cout, 1, 2, res=res)
            cur, Hc = out, Ho
        feat = torch.empty(N, 512, device=dev, dtype=t16)
        ops.avgpool_hw(cur, feat, N, Hc * Hc, 512, dt)
        return feat, B, T

    def forward(self, x):
        feat, B, T = self.forward_rows(x)
        out = torch.empty(B * T, 512, device=feat.device, dtype=torch.float32)
        ops.cast_16_to_f32(feat, out, B * T, 512, self.dtype)
        return out.view(B, T, 512).transpose(1, 2)  # [B, 512, T] like resnet.py:162-163
