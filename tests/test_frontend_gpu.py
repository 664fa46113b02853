"""ResNet-18 3D/2D frontend: HIP path (through the C ABI) vs the CPU oracle on the same seeded weights/inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lip2speech_unit_amd import ops, weights  # noqa: E402
from lip2speech_unit_amd.resnet import ResEncoder  # noqa: E402
from oracle import frontend as ofe  # noqa: E402


def _frames(B, T, seed=1234):
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (B, T, 88, 88), generator=g)
    return ((u8.float() / 255.0 - 0.421) / 0.165).unsqueeze(1)


@pytest.mark.parametrize("dt,tol", [(ops.F16, 1e-2), (ops.BF16, 6e-2)])
def test_frontend_vs_oracle(dt, tol):
    enc = ResEncoder("prelu", None, dtype=dt)
    sd = weights.synth_state_dict(weights.spec_of(enc), seed=0)
    enc.load_state_dict(sd)
    enc = enc.cuda().eval()
    x = _frames(2, 7)
    x[1, :, 5:] = 0  # second clip padded after 5 frames (collater zero fill)
    taps = {}
    with torch.no_grad():
        ref = ofe.res_encoder(sd, x, taps)        # [B,512,T]
        got = enc(x.cuda()).float().cpu()
    assert got.shape == ref.shape
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale, (err, scale)


def test_stem_and_pool_vs_oracle():
    dt = ops.F16
    enc = ResEncoder("prelu", None, dtype=dt)
    sd = weights.synth_state_dict(weights.spec_of(enc), seed=1)
    enc.load_state_dict(sd)
    enc.pack("cuda")
    x = _frames(1, 6, seed=7)
    with torch.no_grad():
        ref = ofe.stem(sd, x)                       # [1,64,T,44,44]
        refp = ofe.stem_pool(ref)
    B, T = 1, 6
    P = enc._packed
    y = torch.empty(B * T, 44, 44, 64, dtype=torch.float16, device="cuda")
    ops.stem_conv3d(x[:, 0].contiguous().cuda(), P["stem_w"], P["stem_b"], P["stem_s"], y, B, T, dt)
    yp = torch.empty(B * T, 22, 22, 64, dtype=torch.float16, device="cuda")
    ops.maxpool2d_3x3s2(y, yp, B * T, 44, 44, 64, dt)
    torch.cuda.synchronize()
    got = y.float().cpu().view(B, T, 44, 44, 64).permute(0, 4, 1, 2, 3)
    gotp = yp.float().cpu().view(B, T, 22, 22, 64).permute(0, 4, 1, 2, 3)
    s = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 4e-3 * s
    assert (gotp - refp).abs().max().item() <= 4e-3 * s


@pytest.mark.parametrize("dt,tol", [(ops.F16, 1e-2), (ops.BF16, 6e-2)])
def test_conv3d_resnet_swish_vs_reference_fixture(golden_dir, dt, tol):
    """the `multi_target` frontend (ESPnet Conv3dResNet, Swish) on the stem / tap-GEMM kernels against the
    output of the reference's own module (tests/golden/frontend_swish.npz)."""
    import os
    import numpy as np
    from lip2speech_unit_amd.conv3d_extractor import Conv3dResNet
    d = np.load(os.path.join(golden_dir, "frontend_swish.npz"))
    enc = Conv3dResNet(dtype=dt)
    enc.load_state_dict(weights.synth_state_dict(weights.spec_of(enc), seed=int(d["seed"])))
    enc = enc.cuda().eval()
    x = (torch.from_numpy(d["frames_u8"]).float() / 255.0 - 0.421) / 0.165
    with torch.no_grad():
        got = enc(x.cuda()).cpu()
    ref = torch.from_numpy(d["out"])
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()
