"""Mixed-length batch with a 10-s clip (BASELINE configs[4] shape family): every stage incl. the vocoder against the
clip-alone oracle.  Exercises the big-tile / patch / fused-ResBlock kernels with row masks at realistic sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lip2speech_unit_amd import ops, weights  # noqa: E402
from lip2speech_unit_amd.conformer import ConformerConfig  # noqa: E402
from lip2speech_unit_amd.hubert import AVHubertConfig  # noqa: E402
from lip2speech_unit_amd.model_avhubert import MultiTargetAVHubertEncoderModel  # noqa: E402
from lip2speech_unit_amd.pipeline import LipToSpeechPipeline  # noqa: E402
from lip2speech_unit_amd.vocoder import AttrDict, MelCodeGenerator  # noqa: E402
from oracle import stage1 as os1  # noqa: E402
from oracle import vocoder as ov  # noqa: E402
from tests.test_models_gpu import VOC_H, _frames  # noqa: E402


def test_mixed_length_batch_with_10s_clip():
    dt = ops.F16
    model = MultiTargetAVHubertEncoderModel.build_model(dtype=dt, w2v_cfg=AVHubertConfig(encoder_layers=2),
                                                        conformer_cfg=ConformerConfig(conformer_layers=2))
    sd = weights.synth_state_dict(weights.spec_of(model), seed=41)
    model.load_state_dict(sd)
    voc = MelCodeGenerator(AttrDict(VOC_H), dtype=dt)
    vsd = weights.synth_state_dict(weights.spec_of(voc), seed=42)
    voc.load_state_dict(vsd)
    voc.remove_weight_norm()
    model.cuda().eval()
    voc.cuda().eval()
    lens = [250, 111, 25]                      # 10 s, 4.4 s, 1 s
    B, T = len(lens), max(lens)
    video = _frames(B, T, 123)
    pad = torch.zeros(B, T, dtype=torch.bool)
    for b, n in enumerate(lens):
        pad[b, n:] = True
        video[b, :, n:] = 0
    spk = torch.rand(B, 256, generator=torch.Generator().manual_seed(5))
    out = LipToSpeechPipeline(model, voc).forward_device(video.cuda(), pad.cuda(), spk.cuda())
    torch.cuda.synchronize()
    vsd_r = {k: v.detach().float().cpu() for k, v in voc.state_dict().items()}
    for b, n in enumerate(lens):
        with torch.no_grad():
            ref = os1.generate(sd, video[b:b + 1, :, :n], torch.zeros(1, n, dtype=torch.bool), spk[b:b + 1],
                               enc_layers=2, conf_layers=2)
        L = 2 * n
        lr = ref["logits"][:, 0, 4:]
        top2 = lr.topk(2, -1).values
        safe = (top2[:, 0] - top2[:, 1]) > 2e-2
        toks = out["tokens"][b].cpu().long()
        assert torch.equal(toks[:L][safe], ref["tokens"][0][:L][safe]), f"clip {b}"
        assert toks[L].item() == 2 and (toks[L + 1:] == 1).all()
        assert int((~safe).sum()) <= 0.1 * L
        mel = out["mel"][b, : 2 * L].cpu()
        assert (mel - ref["mels"][0]).abs().max().item() < 3e-2
        if bool(safe.all()):
            with torch.no_grad():
                code = (ref["tokens"][0][:-1] - 4).unsqueeze(0)
                wav = ov.mel_code_generator(vsd_r, VOC_H, code, ref["mels"][0].t().unsqueeze(0), spk[b:b + 1])[0, 0]
            got = out["wav"][b, : 320 * L].cpu()
            assert (got - wav).abs().max().item() < 2e-3, f"clip {b} wav"     # fp16: ~8x the measured 2.6e-4
        if 320 * L < out["wav"].shape[1]:
            assert out["wav"][b, 320 * L:].abs().max().item() == 0.0
