"""ESPnet `Conv3dResNet` visual frontend on gfx950 — host-side mirror of
espnet/nets/pytorch_backend/backbones/conv3d_extractor.py:25-101 (backbone 'resnet', relu_type 'swish'): the frontend of
the reference's `multi_target` model (multi_target_lip2speech/model.py:184-228).

Same structure as the AV-HuBERT `ResEncoder` (stem Conv3d k5x7x7 + BN3d + act + MaxPool3d, ResNet-18 trunk [2,2,2,2] of
backbones/modules/resnet.py:44-170, global average pool) with Swish activations and ESPnet's state_dict names
(`frontend3D.{0,1}`, `trunk.layerN.M.{conv1,bn1,conv2,bn2,downsample.{0,1}}`; Swish has no parameters), so the kernels are
the same ones: `l2s_stem_pool_fused` with slope = NULL (Swish before the pool: it is not monotonic) and the Conv2d
tap-GEMMs with the Swish epilogue.  forward(xs_pad [B,T,88,88]) -> [B,T,512].
"""
import torch

from . import ops
from .resnet import ResEncoder


class Conv3dResNet(ResEncoder):
    def __init__(self, backbone_type="resnet", relu_type="swish", dtype=ops.F16):
        if backbone_type != "resnet":
            raise NotImplementedError("the lip2speech models build Conv3dResNet with backbone_type='resnet'")
        super().__init__(relu_type=relu_type, weights=None, dtype=dtype)
        self.backbone_type = backbone_type

    def forward(self, xs_pad):
        """conv3d_extractor.py:86-101: xs_pad [B,T,H,W] -> [B,T,512] fp32."""
        feat, B, T = self.forward_rows(xs_pad)
        out = torch.empty(B * T, 512, device=feat.device, dtype=torch.float32)
        ops.cast_16_to_f32(feat, out, B * T, 512, self.dtype)
        return out.view(B, T, 512)
