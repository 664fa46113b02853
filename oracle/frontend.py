"""Oracle for the ResNet-18 3D/2D frontend (avhubert/resnet.py) and, with relu_type='swish', for ESPnet's Conv3dResNet
(espnet/nets/pytorch_backend/backbones/conv3d_extractor.py:25-101 over backbones/modules/resnet.py:44-170): the same
structure and parameter names with Swish (transformer/convolution.py:68-73) in place of PReLU."""
import torch
import torch.nn.functional as F


def _bn(sd, p, x, eps=1e-5):
    # eval-mode BatchNorm (running statistics), nn.BatchNorm2d/3d defaults
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def _act(sd, p, x, relu_type=None):
    # relu_type='prelu' (hubert.py:249): per-channel nn.PReLU; absent key -> ReLU; 'swish': x * sigmoid(x)
    if relu_type == "swish":
        return x * torch.sigmoid(x)
    k = p + ".weight"
    return F.prelu(x, sd[k]) if k in sd else F.relu(x)


def stem(sd, x, prefix="frontend3D", relu_type=None):
    """avhubert/resnet.py:137-141: Conv3d(1,64,(5,7,7),(1,2,2),(2,3,3)) + BN3d + PReLU (+ MaxPool3d)."""
    y = F.conv3d(x, sd[f"{prefix}.0.weight"], None, (1, 2, 2), (2, 3, 3))
    y = _bn(sd, f"{prefix}.1", y)
    y = _act(sd, f"{prefix}.2", y, relu_type)
    return y


def stem_pool(y):
    return F.max_pool3d(y, (1, 3, 3), (1, 2, 2), (0, 1, 1))


def basic_block(sd, p, x, stride, relu_type=None):
    """avhubert/resnet.py:61-74 with downsample_basic_block :20-24."""
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)
    out = _act(sd, p + ".relu1", _bn(sd, p + ".bn1", out), relu_type)
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, 1, 1))
    res = x
    if p + ".downsample.0.weight" in sd:
        res = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0))
    return _act(sd, p + ".relu2", out + res, relu_type)


def trunk(sd, x, prefix="trunk", taps=None, relu_type=None):
    """avhubert/resnet.py:122-129."""
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        for bi in range(2):
            x = basic_block(sd, f"{prefix}.layer{li}.{bi}", x, stride if bi == 0 else 1, relu_type)
        if taps is not None:
            taps[f"layer{li}"] = x
    x = F.adaptive_avg_pool2d(x, 1)
    return x.view(x.size(0), -1)


def res_encoder(sd, x, taps=None, relu_type=None):
    """ResEncoder.forward avhubert/resnet.py:156-169: x [B,1,T,88,88] -> [B,512,T]."""
    B, C, T, H, W = x.shape
    y = stem(sd, x, relu_type=relu_type)
    if taps is not None:
        taps["stem"] = y
    y = stem_pool(y)
    if taps is not None:
        taps["pool"] = y
    Tn = y.shape[2]
    y = y.transpose(1, 2).contiguous().reshape(B * Tn, y.shape[1], y.shape[3], y.shape[4])  # threeD_to_2D_tensor :166-169
    y = trunk(sd, y, taps=taps, relu_type=relu_type)
    return y.view(B, Tn, y.size(1)).transpose(1, 2).contiguous()


def conv3d_resnet(sd, xs_pad, taps=None):
    """Conv3dResNet.forward conv3d_extractor.py:86-101: xs_pad [B,T,88,88] -> [B,T,512]."""
    return res_encoder(sd, xs_pad.unsqueeze(1), taps, relu_type="swish").transpose(1, 2).contiguous()
