"""Non-GEMM kernels of the C ABI vs the torch-fp32 CPU computation they replace."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from lip2speech_unit_amd import ops, weights  # noqa: E402
from oracle import conformer as oc  # noqa: E402
from oracle import decode as od  # noqa: E402


def _r16(x, dt):
    return x.to(ops.torch_dtype(dt)).float()


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("M,C,zp,eps", [(37, 1024, 0, 1e-5), (50, 512, 0, 1e-12), (21, 1024, 1024, 1e-5), (9, 2048, 0, 1e-5)])
def test_layernorm(dt, M, C, zp, eps):
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn(M, C, generator=g) * 2 + 0.3
    gamma, beta = torch.rand(C + zp, generator=g) + 0.5, torch.randn(C + zp, generator=g) * 0.1
    xin = torch.cat([torch.zeros(M, zp), x], 1)
    ref = F.layer_norm(xin, (C + zp,), gamma, beta, eps)
    t16 = ops.torch_dtype(dt)
    y32 = torch.empty(M, C + zp, device="cuda")
    y16 = torch.empty(M, C + zp, device="cuda", dtype=t16)
    ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), eps, y32, M=M, C=C, zero_prefix=zp, dtype=dt)
    ops.layernorm(x.cuda(), gamma.cuda(), beta.cuda(), eps, y16, M=M, C=C, zero_prefix=zp, dtype=dt)
    torch.cuda.synchronize()
    assert (y32.cpu() - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5
    tol = 2e-3 if dt == ops.F16 else 1.2e-2
    assert (y16.float().cpu() - ref).abs().max().item() < tol * ref.abs().max().item()
    # masked rows + in-place fp32
    lens = torch.tensor([M - 5], dtype=torch.int32).cuda()
    xi = x.clone().cuda()
    ops.layernorm(xi, gamma[zp:].cuda() if zp == 0 else gamma.cuda(), beta[zp:].cuda() if zp == 0 else beta.cuda(), eps,
                  xi if zp == 0 else y32, M=M, C=C, zero_prefix=zp, lens=lens, len_mul=1, mask_T=M, dtype=dt)
    torch.cuda.synchronize()
    got = (xi if zp == 0 else y32).cpu()
    assert got[M - 5:].abs().max().item() == 0.0
    assert (got[: M - 5] - ref[: M - 5]).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-5


def _attn_ref(q, k, v, lens, pos=None, u=None, vb=None):
    """q,k,v [B,H,T,64] fp32 (q already scaled); pos [H,2T-1,64] -> [B,T,H*64]"""
    B, H, T, d = q.shape
    if pos is None:
        s = q @ k.transpose(-1, -2)
    else:
        ac = (q + u[None, :, None, :]) @ k.transpose(-1, -2)
        bd = oc.rel_shift((q + vb[None, :, None, :]) @ pos.transpose(-1, -2)[None])
        s = ac + bd
    mask = torch.arange(T)[None, :] >= lens[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, H * d)


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("B,T,H,relpos", [(2, 100, 16, False), (3, 37, 4, False), (2, 200, 8, True), (2, 70, 8, True),
                                          (1, 300, 2, True), (1, 130, 2, False), (1, 600, 2, True),
                                          (2, 250, 4, False), (2, 64, 2, False), (2, 65, 2, True), (2, 128, 2, True),
                                          (1, 1, 1, False), (1, 513, 1, True),
                                          # sequence-resident rel-pos kernel (T <= 224): more clips than block slots, ragged
                                          # lengths, the largest T it takes, tiny T
                                          (70, 200, 8, True), (3, 224, 8, True), (2, 17, 8, True), (5, 1, 2, True),
                                          (40, 33, 8, True), (2, 225, 8, True)])
def test_attention(dt, B, T, H, relpos):
    g = torch.Generator().manual_seed(B * 1000 + T)
    d = 64
    qkv = _r16(torch.randn(B, T, 3, H, d, generator=g), dt)
    qkv[:, :, 0] *= 0.35
    lens = torch.tensor([T] + [max(1, T - 13 * (i + 1)) for i in range(B - 1)])
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    t16 = ops.torch_dtype(dt)
    dev_qkv = qkv.reshape(B * T, 3 * H * d).to(t16).cuda()
    out = torch.empty(B * T, H * d, device="cuda", dtype=t16)
    if relpos:
        pos = _r16(torch.randn(2 * T - 1, H, d, generator=g) * 0.5, dt)
        u, vb = torch.randn(H, d, generator=g) * 0.1, torch.randn(H, d, generator=g) * 0.1
        # the kernel rounds q+u / q+v to 16-bit before the MFMA: `ref` mirrors that rounding (tight tolerance: everything
        # else of the kernel), `ref_plain` is the un-mirrored fp32 q+u / q+v of attention.py:240-280 (tolerance x2: the
        # operand rounding adds 2^-11 / 2^-8 of |q+u| per score term on top of the 16-bit rounding of P and the output)
        ref = _attn_ref_rounded(q, k, v, lens, pos.permute(1, 0, 2), u, vb, dt)
        ref_plain = _attn_ref(q, k, v, lens, pos.permute(1, 0, 2), u, vb)
        ops.attention(dev_qkv, out, B=B, T=T, H=H, pos=pos.reshape(2 * T - 1, H * d).to(t16).cuda(), ldp=H * d,
                      bias_u=u.cuda(), bias_v=vb.cuda(), lens=lens.int().cuda(), dtype=dt)
    else:
        ref = ref_plain = _attn_ref(q, k, v, lens)
        ops.attention(dev_qkv, out, B=B, T=T, H=H, lens=lens.int().cuda(), dtype=dt)
    torch.cuda.synchronize()
    got = out.float().cpu().view(B, T, H * d)
    tol = 4e-3 if dt == ops.F16 else 2.5e-2
    for b in range(B):
        n = int(lens[b])
        err = (got[b, :n] - ref[b, :n]).abs().max().item()
        assert err < tol * ref.abs().max().item(), (b, err)
        err = (got[b, :n] - ref_plain[b, :n]).abs().max().item()
        assert err < 2 * tol * ref_plain.abs().max().item(), ("un-mirrored", b, err)
    assert torch.isfinite(got).all()


def _attn_ref_rounded(q, k, v, lens, pos, u, vb, dt):
    B, H, T, d = q.shape
    qu = _r16(q + u[None, :, None, :], dt)
    qv = _r16(q + vb[None, :, None, :], dt)
    ac = qu @ k.transpose(-1, -2)
    bd = oc.rel_shift(qv @ pos.transpose(-1, -2)[None])
    s = ac + bd
    mask = torch.arange(T)[None, :] >= lens[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, T, H * d)


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("T,lens", [(150, [150, 97]), (200, [200, 101]), (100, [99, 100]), (257, [257, 3]), (31, [31, 1])])
def test_glu_dwconv_swish(dt, T, lens):
    """T = 150, 257, 31 run on the 128-step tile, T = 200 / 100 on the 100-step tile (the launcher takes whichever wastes
    fewer rows); ragged lengths, a clip shorter than the conv's half-width."""
    B, C, k = 2, 128, 31
    g = torch.Generator().manual_seed(3 + T)
    x = _r16(torch.randn(B, T, 2 * C, generator=g), dt)
    w, b = torch.randn(C, k, generator=g) / k ** 0.5, torch.randn(C, generator=g) * 0.1
    lens = torch.tensor(lens)
    refs = []
    for i in range(B):
        n = int(lens[i])
        xi = F.glu(x[i:i + 1, :n].transpose(1, 2), dim=1)
        yi = F.conv1d(xi, w[:, None, :], b, padding=15, groups=C)
        yi = (yi * torch.sigmoid(yi)).transpose(1, 2)[0]
        refs.append(torch.cat([yi, torch.zeros(T - n, C)], 0))
    ref = torch.stack(refs)
    t16 = ops.torch_dtype(dt)
    y = torch.empty(B * T, C, device="cuda", dtype=t16)
    ops.glu_dwconv_swish(x.reshape(B * T, 2 * C).to(t16).cuda(), w.t().contiguous().cuda(), b.cuda(), y, B=B, T=T, C=C,
                         k=k, lens=lens.int().cuda(), len_mul=1, dtype=dt)
    torch.cuda.synchronize()
    tol = 2e-3 if dt == ops.F16 else 1.2e-2
    assert (y.float().cpu().view(B, T, C) - ref).abs().max().item() < tol * ref.abs().max().item()


def test_greedy_decode_matches_beam_search_oracle():
    B, T2, V = 3, 24, 204
    g = torch.Generator().manual_seed(9)
    logits = torch.randn(T2, B, V, generator=g) * 3
    lens_half = torch.tensor([12, 7, 3])  # video frames; target lengths 24, 14, 6
    tl = (lens_half * 2).tolist()
    fin = od.beam_search_decode(logits, tl, beam_size=5, temperature=1.3)
    rows = logits.transpose(0, 1).contiguous().view(B * T2, V).cuda()
    tokens = torch.empty(B, T2 + 1, dtype=torch.int32, device="cuda")
    lprobs = torch.empty(B, T2 + 1, device="cuda")
    score = torch.empty(B, device="cuda")
    ops.greedy_decode(rows, tokens, lprobs, score, B=B, T2=T2, V=V, lens=lens_half.int().cuda(), len_mul=2,
                      temperature=1.3, lenpen=1.0)
    torch.cuda.synchronize()
    for b in range(B):
        n = tl[b]
        assert tokens[b, : n + 1].cpu().tolist() == fin[b][0]["tokens"].tolist()          # bit-exact ids
        assert tokens[b, n + 1:].cpu().eq(1).all()
        assert abs(score[b].item() - float(fin[b][0]["score"])) < 1e-4
        assert (lprobs[b, : n + 1].cpu() - fin[b][0]["positional_scores"]).abs().max().item() < 1e-4


@pytest.mark.parametrize("beam", [1, 5, 50, 64])
def test_beam_decode_nbest_matches_beam_search_oracle(beam):
    """a12 n-best: l2s_beam_decode against the oracle's restatement of the reference loop (BeamSearch.step, finalize_hypos) -
    tokens and order of every hypothesis exact, scores / positional scores to fp32 rounding; hypothesis 0 == greedy kernel."""
    B, T2, V = 4, 30, 204
    g = torch.Generator().manual_seed(beam)
    logits = torch.randn(T2, B, V, generator=g) * 2.5
    logits[3, 0, 2] = 40.0      # specials never win (:276-282)
    logits[4, 1, 1] = 40.0
    lens_half = torch.tensor([15, 9, 1, 0])       # target lengths 30, 18, 2, 0 (empty clip: EOS only)
    tl = (lens_half * 2).tolist()
    temp, lenpen = 0.9, 1.0
    fin = od.beam_search_decode(logits, tl, beam_size=beam, temperature=temp, len_penalty=lenpen)
    rows = logits.transpose(0, 1).contiguous().view(B * T2, V).cuda()
    tok, pos, score, nhyp = ops.beam_decode(rows, B=B, T2=T2, V=V, beam=beam, lens=lens_half.int().cuda(), len_mul=2,
                                            temperature=temp, lenpen=lenpen)
    gt = torch.empty(B, T2 + 1, dtype=torch.int32, device="cuda")
    gl = torch.empty(B, T2 + 1, device="cuda")
    gs = torch.empty(B, device="cuda")
    ops.greedy_decode(rows, gt, gl, gs, B=B, T2=T2, V=V, lens=lens_half.int().cuda(), len_mul=2, temperature=temp, lenpen=lenpen)
    torch.cuda.synchronize()
    tok, pos, score, nhyp = tok.cpu(), pos.cpu(), score.cpu(), nhyp.cpu().tolist()
    for b in range(B):
        n = tl[b]
        assert nhyp[b] == len(fin[b]) == (beam if n > 0 else 1)
        for h, hyp in enumerate(fin[b]):
            assert tok[b, h, : n + 1].tolist() == hyp["tokens"].tolist(), (b, h)
            assert tok[b, h, n + 1:].eq(1).all()
            assert abs(score[b, h].item() - float(hyp["score"])) < 1e-4 * max(1.0, abs(float(hyp["score"])))
            assert (pos[b, h, : n + 1] - hyp["positional_scores"]).abs().max().item() < 2e-4
        assert torch.equal(tok[b, 0], gt[b].cpu())
        s = score[b, : nhyp[b]]
        assert bool((s[:-1] >= s[1:]).all())          # finalized sorted by score (:497-505)
        assert len({tuple(tok[b, h].tolist()) for h in range(nhyp[b])}) == nhyp[b]   # distinct hypotheses


def test_misc_layout_kernels():
    dt, t16 = ops.F16, torch.float16
    g = torch.Generator().manual_seed(1)
    B, T, C = 2, 13, 64
    x = torch.randn(B * T, C, generator=g)
    y = torch.empty(B * 2 * T, C, device="cuda", dtype=t16)
    ops.repeat2_cast(x.cuda(), y, B, T, C, dt)
    ref = x.view(B, T, C).repeat_interleave(2, dim=1).reshape(B * 2 * T, C).half()
    assert torch.equal(y.cpu(), ref)
    # mel transpose + mask, speaker broadcast, embedding
    mel = torch.randn(B, 80, T, generator=g)
    cat = torch.zeros(B * T, 336, device="cuda", dtype=t16)
    lens = torch.tensor([T, 5], dtype=torch.int32).cuda()
    ops.transpose_ct_to_tc(mel.cuda(), cat, B=B, C=80, T=T, ldy=336, col0=0, lens=lens, len_mul=1, dtype=dt)
    spk = torch.randn(B, 128, generator=g)
    ops.broadcast_rows(spk.cuda(), cat, B=B, T=T, C=128, ldy=336, col0=208, lens=lens, len_mul=1, dtype=dt)
    c = cat.cpu().view(B, T, 336)
    assert torch.equal(c[0, :, :80], mel[0].t().half()) and c[1, 5:, :80].abs().max() == 0
    assert torch.equal(c[1, :5, 208:], spk[1].half().expand(5, 128)) and c[1, 5:, 208:].abs().max() == 0
    table = torch.randn(200, 128, generator=g).half()
    code = torch.randint(0, 200, (B, T), generator=g).int()
    e = torch.empty(B * T, 128, device="cuda", dtype=t16)
    ops.embedding(code.cuda(), table.cuda(), e, B=B, L=T, C=128, lens=lens, dtype=dt)
    ev = e.cpu().view(B, T, 128)
    assert torch.equal(ev[0], table[code[0].long()]) and ev[1, 5:].abs().max() == 0
    # uint8 frame preprocessing (hubert_dataset.py:242-245)
    u8 = torch.randint(0, 256, (1, 3, 96, 96), generator=g).to(torch.uint8)
    o = torch.empty(1, 3, 88, 88, device="cuda", dtype=t16)
    ops.preprocess_frames(u8.cuda(), o, B=1, T=3, Hin=96, Win=96, dtype=dt)
    ref = ((u8[:, :, 4:92, 4:92].float() / 255.0 - 0.421) / 0.165)
    assert (o.float().cpu() - ref).abs().max().item() < 3e-3


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
def test_stage_handoff_kernels(dt):
    """The in-memory stage 1 -> stage 2 hand-off (inference.py:267-274 -> dataset_multi_input.py:41-110,198-291): token rows ->
    embedding rows with the -4 offset, time-major mel rows -> the concat buffer's 16-bit columns, lens from the padding mask."""
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(5)
    B, T = 3, 11                       # video frames; L = 2T units, 4T mel rows
    L = 2 * T
    pad = torch.zeros(B, T, dtype=torch.bool)
    pad[1, 7:] = True
    pad[2, 1:] = True
    lens = ops.lens_from_mask(pad.cuda(), B, T, torch.device("cuda"))
    assert lens.cpu().tolist() == [11, 7, 1]
    assert ops.lens_from_mask(None, B, T, torch.device("cuda")).cpu().tolist() == [T] * B
    table = torch.randn(200, 128, generator=g).to(t16)
    tok = torch.randint(4, 204, (B, L + 1), generator=g).int()
    tok[0, 3], tok[1, 2] = 2, 0        # a special inside the valid range clamps to unit 0 instead of indexing out of range
    e = torch.empty(B * L, 128, device="cuda", dtype=t16)
    ops.embedding_tokens(tok.cuda(), table.cuda(), e, B=B, L=L, C=128, token_offset=4, lens=lens, len_mul=2, dtype=dt)
    ev = e.cpu().view(B, L, 128)
    for b, n in enumerate((11, 7, 1)):
        assert torch.equal(ev[b, : 2 * n], table[(tok[b, : 2 * n].long() - 4).clamp(min=0)]) and not ev[b, 2 * n:].any()
    mel = torch.randn(B, 2 * L, 80, generator=g)
    cat = torch.full((B * 2 * L, 336), 7.0, device="cuda", dtype=t16)
    ops.rows_f32_to_16_masked(mel.cuda(), cat, B=B, T=2 * L, C=80, ldy=336, col0=0, lens=lens, len_mul=4, dtype=dt)
    c = cat.cpu().view(B, 2 * L, 336)
    for b, n in enumerate((11, 7, 1)):
        assert torch.equal(c[b, : 4 * n, :80], mel[b, : 4 * n].to(t16)) and not c[b, 4 * n:, :80].any()
    assert (c[:, :, 80:] == 7.0).all()                   # the other columns are not touched
    with pytest.raises(ops.L2SError):
        ops.rows_f32_to_16_masked(mel.cuda(), cat, B=B, T=2 * L, C=80, ldy=336, col0=2, lens=lens, len_mul=4, dtype=dt)


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
def test_vocoder_token_handoff_equals_code_mel_entry(dt):
    """MelCodeGenerator.forward_tokens_rows (tokens, time-major mel, video-frame lengths: what stage 1 holds on the device) is
    bit-identical to forward_rows on (tokens - 4, mel transposed to [B,80,2L], code-frame lengths: what the reference's vocoder
    dataset would have read back from pred_unit / pred_mel), ragged lengths included."""
    from bench import VOC_H
    from lip2speech_unit_amd.vocoder import AttrDict, MelCodeGenerator
    voc = MelCodeGenerator(AttrDict(VOC_H), dtype=dt)
    voc.load_state_dict(weights.synth_state_dict(weights.spec_of(voc), seed=1))
    voc.remove_weight_norm()
    voc.cuda().eval()
    g = torch.Generator().manual_seed(9)
    B, T = 3, 6
    L = 2 * T
    src_lens = torch.tensor([6, 4, 1], dtype=torch.int32)
    tok = torch.randint(4, 204, (B, L + 1), generator=g).int()
    mel = -11.5 + 11.6 * torch.rand(B, 2 * L, 80, generator=g)
    spk = torch.rand(B, 256, generator=g)
    with torch.no_grad():
        w1, p1 = voc.forward_tokens_rows(tok.cuda(), mel.cuda(), spk.cuda(), src_lens.cuda())
        w2, p2 = voc.forward_rows((tok[:, :L] - 4).cuda(), mel.transpose(1, 2).contiguous().cuda(), spk.cuda(),
                                  (2 * src_lens).cuda())
    torch.cuda.synchronize()
    assert torch.equal(w1, w2) and torch.equal(p1, p2)
    for b, n in enumerate(src_lens.tolist()):
        assert w1[b, : 640 * n].abs().max() > 0 and not w1[b, 640 * n:].any()


@pytest.mark.parametrize("dt,tol", [(ops.F16, 2e-3), (ops.BF16, 1.5e-2)])
@pytest.mark.parametrize("N,H", [(300, 22), (5, 22), (7, 10), (3, 21)])
def test_basicblock_fused_vs_torch(dt, tol, N, H):
    """l2s_basicblock_fused = prelu(conv2(prelu(conv1(x) + b1)) + b2 + x) (avhubert/resnet.py:61-74, BatchNorm folded) against
    torch fp32 on the 16-bit-rounded operands; more images than blocks (300 > 256), image sizes below the 22 x 22 of the path."""
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(N * 100 + H)
    C = 64
    x = torch.randn(N, C, H, H, generator=g).to(t16).float()
    w1 = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(t16).float()
    w2 = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(t16).float()
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    s1, s2 = torch.rand(C, generator=g) * 0.4, torch.rand(C, generator=g) * 0.4
    t1 = F.prelu(F.conv2d(x, w1, b1, padding=1), s1).to(t16).float()          # the kernel keeps t1 in 16 bits (as two launches do)
    ref = F.prelu(F.conv2d(t1, w2, b2, padding=1) + x, s2)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(N * H * H, C).contiguous()
    pack = lambda w: w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(t16).cuda()      # K = (ky*3 + kx)*C + cin
    y = torch.full((N * H * H, C), float("nan"), device="cuda", dtype=t16)
    ops.basicblock_fused(rows(x).to(t16).cuda(), pack(w1), b1.cuda(), s1.cuda(), pack(w2), b2.cuda(), s2.cuda(), y, n_images=N,
                         H=H, W=H, dtype=dt)
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - rows(ref)).abs().max().item()
    assert err < tol * rows(ref).abs().max().item(), err


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("nb,N,H", [(2, 300, 22), (3, 37, 13), (4, 5, 22)])
def test_basiclayer_fused_equals_block_launches(dt, nb, N, H):
    """l2s_basiclayer_fused (nb BasicBlocks back to back on the LDS-resident image, avhubert/resnet.py:101-118) == nb
    l2s_basicblock_fused launches, bit for bit: the hand-over between blocks is the same 16-bit rounding.  The single block is
    checked against torch in test_basicblock_fused_vs_torch.  More images than blocks, a smaller image, the maximum depth."""
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(nb * 1000 + N)
    C = 64
    x = torch.randn(N * H * H, C, generator=g).to(t16).cuda()
    ws = [(torch.randn(C, 9 * C, generator=g) / (9 * C) ** 0.5).to(t16).cuda() for _ in range(2 * nb)]
    bs = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2 * nb)]
    ss = [(torch.rand(C, generator=g) * 0.4).cuda() for _ in range(2 * nb)]
    cur = x
    for b in range(nb):
        out = torch.full_like(x, float("nan"))
        ops.basicblock_fused(cur, ws[2 * b], bs[2 * b], ss[2 * b], ws[2 * b + 1], bs[2 * b + 1], ss[2 * b + 1], out, n_images=N,
                             H=H, W=H, dtype=dt)
        cur = out
    y = torch.full_like(x, float("nan"))
    ops.basiclayer_fused(x, ws, bs, ss, y, n_images=N, H=H, W=H, dtype=dt)
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and float(y.float().abs().max()) > 0
    assert torch.equal(y.view(torch.int16), cur.view(torch.int16))
    with pytest.raises(ops.L2SError):      # at most four blocks per launch
        ops.basiclayer_fused(x, ws + ws[:2] * (5 - nb), bs + bs[:2] * (5 - nb), ss + ss[:2] * (5 - nb), y, n_images=N, H=H, W=H,
                             dtype=dt)


@pytest.mark.parametrize("dt,tol", [(ops.F16, 2e-3), (ops.BF16, 1.5e-2)])
@pytest.mark.parametrize("N", [600, 5, 2, 1])
def test_basicblock128_fused_vs_torch(dt, tol, N):
    """The 128-channel family of l2s_basicblock_fused (csrc/basicblock_phase.hip: layer2's second block, 11 x 11 maps, two images
    per tile, avhubert/resnet.py:61-74) against torch fp32 on the 16-bit-rounded operands: more tiles than blocks (600 images =
    300 tiles > 256), an odd image count (the last tile holds one image), a single tile, a single image."""
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(N * 100 + 11)
    C, H = 128, 11
    x = torch.randn(N, C, H, H, generator=g).to(t16).float()
    w1 = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(t16).float()
    w2 = (torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5).to(t16).float()
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    s1, s2 = torch.rand(C, generator=g) * 0.4, torch.rand(C, generator=g) * 0.4
    t1 = F.prelu(F.conv2d(x, w1, b1, padding=1), s1).to(t16).float()          # the kernel keeps t1 in 16 bits (as two launches do)
    ref = F.prelu(F.conv2d(t1, w2, b2, padding=1) + x, s2)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(N * H * H, C).contiguous()
    pack = lambda w: w.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous().to(t16).cuda()      # K = (ky*3 + kx)*C + cin
    y = torch.full((N * H * H + 64, C), float("nan"), device="cuda", dtype=t16)                 # + a guard band behind the output
    ops.basicblock_fused(rows(x).to(t16).cuda(), pack(w1), b1.cuda(), s1.cuda(), pack(w2), b2.cuda(), s2.cuda(), y, n_images=N,
                         H=H, W=H, C=C, dtype=dt)
    torch.cuda.synchronize()
    assert torch.isnan(y[N * H * H:].float()).all()                            # nothing written past the last image
    got = y[: N * H * H].float().cpu()
    assert torch.isfinite(got).all()
    err = (got - rows(ref)).abs().max().item()
    assert err < tol * rows(ref).abs().max().item(), err


@pytest.mark.parametrize("dt,tol", [(ops.F16, 2e-3), (ops.BF16, 1.5e-2)])
@pytest.mark.parametrize("N", [520, 3, 1])
def test_basicstage128_tail_fused_vs_torch(dt, tol, N):
    """l2s_basicstage128_tail_fused (csrc/basicblock_phase.hip, TAIL): layer2 behind its first convolution - conv2 of the strided
    block with the 1x1 stride-2 downsample of the 22 x 22 x 64 stage input as residual (avhubert/resnet.py:61-74 with downsample),
    then the second block - against torch fp32 on the 16-bit-rounded operands.  260 tiles > 256 blocks, an odd count, one image."""
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(N * 7 + 3)
    C, H = 128, 11
    r16 = lambda t: t.to(t16).float()
    x0 = r16(torch.randn(N, 64, 2 * H, 2 * H, generator=g))
    t0 = r16(torch.randn(N, C, H, H, generator=g))
    wa, w1, w2 = (r16(torch.randn(C, C, 3, 3, generator=g) / (9 * C) ** 0.5) for _ in range(3))
    wd = r16(torch.randn(C, 64, 1, 1, generator=g) / 8)
    ba_c, bd, b1, b2 = (torch.randn(C, generator=g) * 0.1 for _ in range(4))
    sa, s1, s2 = (torch.rand(C, generator=g) * 0.4 for _ in range(3))
    out0 = r16(F.prelu(F.conv2d(t0, wa, ba_c, padding=1) + F.conv2d(x0, wd, bd, stride=2), sa))   # 16 bits where it leaves the registers
    t1 = r16(F.prelu(F.conv2d(out0, w1, b1, padding=1), s1))
    ref = F.prelu(F.conv2d(t1, w2, b2, padding=1) + out0, s2)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()
    pack = lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()
    wa_cat = torch.cat([pack(wa), wd[:, :, 0, 0], torch.zeros(C, 64)], dim=1).to(t16).cuda()
    y = torch.full((N * H * H + 64, C), float("nan"), device="cuda", dtype=t16)
    ops.basicstage128_tail_fused(rows(x0).to(t16).cuda(), rows(t0).to(t16).cuda(), wa_cat, (ba_c + bd).cuda(), sa.cuda(),
                                 pack(w1).to(t16).cuda(), b1.cuda(), s1.cuda(), pack(w2).to(t16).cuda(), b2.cuda(), s2.cuda(), y,
                                 n_images=N, H=H, W=H, dtype=dt)
    torch.cuda.synchronize()
    assert torch.isnan(y[N * H * H:].float()).all()
    got = y[: N * H * H].float().cpu()
    assert torch.isfinite(got).all()
    err = (got - rows(ref)).abs().max().item()
    assert err < tol * rows(ref).abs().max().item(), err
    with pytest.raises(ops.L2SError):      # built for the 11 x 11 maps of the path only
        ops.basicstage128_tail_fused(rows(x0).to(t16).cuda(), rows(t0).to(t16).cuda(), wa_cat, (ba_c + bd).cuda(), sa.cuda(),
                                     pack(w1).to(t16).cuda(), b1.cuda(), s1.cuda(), pack(w2).to(t16).cuda(), b2.cuda(), s2.cuda(), y,
                                     n_images=N, H=10, W=10, dtype=dt)


@pytest.mark.parametrize("dt,tol", [(ops.F16, 2e-3), (ops.BF16, 1.5e-2)])
@pytest.mark.parametrize("Hi,st,Cin,Co,res", [(6, 1, 256, 256, True), (11, 2, 128, 256, False), (3, 1, 512, 512, True),
                                              (6, 2, 256, 512, False), (3, 1, 512, 512, False)])
def test_ktab_conv3x3_small_maps_vs_torch(dt, tol, Hi, st, Cin, Co, res):
    """K-block-table launches of the phase-staggered kernel (l2s_gemm_desc::ktab, csrc/phasegemm_kernel.h MODE 3): a 3x3 / padding-1
    convolution on a small map with the image as one row of A, every output position summing only its in-map taps
    (avhubert/resnet.py:15-24, 61-74 on layer3 / layer4's maps), bias + PReLU (+ 16-bit residual in front of it), against
    torch conv2d in fp32 on the 16-bit-rounded operands.  520 images: 3 M-tiles, ragged last tile; strides 1 and 2."""
    from lip2speech_unit_amd.resnet import ktab_conv3x3
    from lip2speech_unit_amd.ops import ACT_PRELU, F_RES_PRE
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(Hi * 100 + Cin + st)
    N = 520
    x = torch.randn(N, Cin, Hi, Hi, generator=g).to(t16).float()
    w = (torch.randn(Co, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).to(t16).float()
    b, sl = torch.randn(Co, generator=g) * 0.1, torch.rand(Co, generator=g) * 0.4
    tab, Ho, Wo, nblk = ktab_conv3x3(Hi, Hi, Cin, st, "cuda")
    assert tab.shape == (Ho * Wo, 20) and nblk < 9 * Ho * Wo                 # some taps are padding
    r = torch.randn(N, Co, Ho, Wo, generator=g).to(t16).float() if res else None
    pre = F.conv2d(x, w, b, stride=st, padding=1) + (r if res else 0)
    ref = F.prelu(pre, sl)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(t.shape[0], -1).contiguous()      # one image = one row
    y = torch.full((N, Ho * Wo * Co), float("nan"), device="cuda", dtype=t16)
    G = Ho * Wo
    ops.tapgemm(rows(x).to(t16).cuda(), w.permute(0, 2, 3, 1).reshape(Co, 9 * Cin).contiguous().to(t16).cuda(), y, M=N, N=Co, Cin=Cin,
                ntaps=9, lda=Hi * Hi * Cin, ldc=G * Co, groups=G, c_gstride=Co, bias=b.repeat(G).cuda(), slope=sl.repeat(G).cuda(),
                act=ACT_PRELU, R=rows(r).to(t16).cuda() if res else None, ldr=G * Co, flags=F_RES_PRE if res else 0, dtype=dt, ktab=tab)
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - rows(ref)).abs().max().item()
    assert err < tol * rows(ref).abs().max().item(), err


def test_basicblock_fused_rejects_other_layouts():
    x = torch.zeros(4 * 30 * 30, 64, device="cuda", dtype=torch.float16)
    w = torch.zeros(64, 576, device="cuda", dtype=torch.float16)
    v = torch.zeros(64, device="cuda")
    with pytest.raises(ops.L2SError):      # 32 x 32 padded positions do not fit the 576-position block
        ops.basicblock_fused(x, w, v, v, w, v, v, x.clone(), n_images=4, H=30, W=30)
    with pytest.raises(ops.L2SError):      # the 128-channel family is built for the 11 x 11 maps of the path only
        ops.basicblock_fused(x, w, v, v, w, v, v, x.clone(), n_images=4, H=10, W=10, C=128)
    with pytest.raises(ops.L2SError):      # no other channel count
        ops.basicblock_fused(x, w, v, v, w, v, v, x.clone(), n_images=4, H=10, W=10, C=256)


@pytest.mark.parametrize("C,k", [(16, 7), (32, 7), (16, 5)])
def test_conv_post_tanh_and_pcm(C, k):
    """speech-resynthesis/models.py:110-112 + the int16 conversion of multi_input_vocoder/inference.py:79-81.  (16, 7) is the
    generator's own shape and has its own kernel (float4 tile, padded LDS rows); the others run the generic one."""
    B, T = 2, 700
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, T, C, generator=g)
    w, b = torch.randn(1, C, k, generator=g) * 0.2, 0.05
    lens = torch.tensor([700, 300])
    wav = torch.empty(B, T, device="cuda")
    pcm = torch.empty(B, T, device="cuda", dtype=torch.int16)
    ops.conv_post_tanh(x.reshape(B * T, C).cuda(), w[0].t().contiguous().cuda(), b, wav, pcm, B=B, T=T, C=C, k=k,
                       lens=lens.int().cuda(), len_mul=1)
    for i in range(B):
        n = int(lens[i])
        xi = F.leaky_relu(x[i:i + 1, :n].transpose(1, 2))
        ref = torch.tanh(F.conv1d(xi, w, torch.tensor([b]), padding=(k - 1) // 2))[0, 0]
        got = wav[i, :n].cpu()
        assert (got - ref).abs().max().item() < 2e-5
        assert wav[i, n:].abs().max().item() == 0 if n < T else True
        # truncating astype('int16') of the kernel's own fp32 samples (multi_input_vocoder/inference.py:79-81)
        assert torch.equal(pcm[i, :n].cpu(), torch.from_numpy((got * 32768.0).numpy().astype("int16")))


def test_pipeline_u8_input_equals_fp32_input():
    """uint8 frames + on-device crop/normalise == the reference's CPU-normalised fp32 frames."""
    from bench import build, synth_inputs
    from lip2speech_unit_amd.pipeline import LipToSpeechPipeline
    dev = torch.device("cuda")
    model, voc, _, _ = build(ops.F16, dev, 2, 1)
    pipe = LipToSpeechPipeline(model, voc)
    B, T = 3, 14
    video, spk = synth_inputs(B, T, seed=77)
    g = torch.Generator().manual_seed(77)
    u8 = torch.randint(0, 256, (B, T, 96, 96), generator=g, dtype=torch.uint8)
    a = pipe.forward_device(video.to(dev), None, spk.to(dev))
    b = pipe.forward_device_u8(u8.to(dev), None, spk.to(dev))
    torch.cuda.synchronize()
    assert torch.equal(a["tokens"], b["tokens"])
    assert float((a["mel"] - b["mel"]).abs().max()) < 2e-3
    assert float((a["wav"] - b["wav"]).abs().max()) < 2e-3
    # crop + normalise inside the stem's frame fetch (l2s_stem_pool_fused_u8) == the separate l2s_preprocess_frames launch
    c = pipe.forward_device_u8(u8.to(dev), None, spk.to(dev), fused=False)
    torch.cuda.synchronize()
    for k in ("tokens", "mel", "wav", "pcm", "logits"):
        assert torch.equal(b[k], c[k]), k


def test_pipeline_sub_batches_on_streams_equal_one_batch():
    """forward_device_u8_streams: the batch as 2 / 3 independent sub-batches on their own HIP streams (eager and captured
    into one hipGraph) gives the results of the one-stream batch bit for bit - clips are independent."""
    from bench import build, synth_inputs
    from lip2speech_unit_amd.pipeline import LipToSpeechPipeline
    dev = torch.device("cuda")
    model, voc, _, _ = build(ops.F16, dev, 2, 1)
    pipe = LipToSpeechPipeline(model, voc)
    B, T = 6, 12
    _, spk, u8 = synth_inputs(B, T, seed=5, with_u8=True)
    u8, spk = u8.to(dev), spk.to(dev)
    ref = pipe.forward_device_u8(u8, None, spk)
    torch.cuda.synchronize()
    for n in (2, 3):
        got = pipe.forward_device_u8_streams(u8, None, spk, n)
        torch.cuda.synchronize()
        for k in ("tokens", "lens", "mel", "wav", "pcm"):
            assert torch.equal(ref[k], got[k]), (n, k)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipe.forward_device_u8_streams(u8, None, spk, 2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = pipe.forward_device_u8_streams(u8, None, spk, 2)
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    for k in ("tokens", "wav", "pcm"):
        assert torch.equal(ref[k], out[k]), k


def test_pipeline_streams_first_call_on_fresh_model():
    """The sub-batches share lazily built state (packed weights, the conformer's position table): the very first call on a
    freshly built model - nothing packed yet - must already give the one-stream results, also for a second clip length."""
    from bench import build, synth_inputs
    from lip2speech_unit_amd.pipeline import LipToSpeechPipeline
    dev = torch.device("cuda")
    model, voc, _, _ = build(ops.F16, dev, 2, 1)
    pipe = LipToSpeechPipeline(model, voc)
    for T in (12, 9):
        _, spk, u8 = synth_inputs(6, T, seed=11 + T, with_u8=True)
        u8, spk = u8.to(dev), spk.to(dev)
        got = pipe.forward_device_u8_streams(u8, None, spk, 3)      # FIRST use of this model / this T
        torch.cuda.synchronize()
        ref = pipe.forward_device_u8(u8, None, spk)
        torch.cuda.synchronize()
        for k in ("tokens", "lens", "mel", "wav", "pcm"):
            assert torch.equal(ref[k], got[k]), (T, k)


@pytest.mark.parametrize("Hin,Win", [(96, 96), (88, 88), (97, 120), (89, 91)])
@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
def test_stem_u8_fetch_bit_identical(Hin, Win, dt):
    """l2s_stem_pool_fused_u8 == l2s_preprocess_frames + l2s_stem_pool_fused, bit for bit, for even and odd crop margins
    (utils.py:90-91 truncates the offset), PReLU and Swish stems, clips that do not fill the last 10-frame block."""
    dev = torch.device("cuda")
    t16 = ops.torch_dtype(dt)
    B, T = 2, 13
    g = torch.Generator().manual_seed(Hin * 131 + Win)
    u8 = torch.randint(0, 256, (B, T, Hin, Win), generator=g, dtype=torch.uint8).to(dev)
    w = (torch.randn(64, 36 * 8, generator=g) * 0.05).to(dev, t16)
    bias = (torch.randn(64, generator=g) * 0.1).to(dev)
    slope = torch.rand(64, generator=g).to(dev)
    x = torch.empty(B, T, 88, 88, device=dev, dtype=t16)
    ops.preprocess_frames(u8, x, B=B, T=T, Hin=Hin, Win=Win, crop=88, mean=0.421, std=0.165, dtype=dt)
    for sl in (slope, None):
        ya = torch.empty(B * T, 22, 22, 64, device=dev, dtype=t16)
        yb = torch.empty_like(ya)
        ops.stem_pool_fused(x, w, bias, sl, ya, B, T, dt)
        ops.stem_pool_fused_u8(u8, w, bias, sl, yb, B, T, dt)
        torch.cuda.synchronize()
        assert torch.equal(ya.view(torch.int16), yb.view(torch.int16))
        assert ya.float().abs().max().item() > 0


@pytest.mark.parametrize("slopes", ["positive", "mixed", "swish"])
@pytest.mark.parametrize("dt,tol", [(ops.F16, 3e-3), (ops.BF16, 2e-2)])
def test_stem_pool_fused_vs_torch(slopes, dt, tol):
    """Conv3d(1->64, 5x7x7, s 1x2x2, p 2x3x3) + bias + PReLU / Swish + MaxPool3d(1x3x3, s 1x2x2, p 0x1x1)
    (avhubert/resnet.py:137-141, BatchNorm folded) against torch fp32 on the kernel's own 16-bit operands.  Three paths:
    slopes all >= 0 (the kernel pools the raw conv tile and activates the pooled quarter - PReLU is then non-decreasing),
    mixed-sign slopes and Swish (activation before the pool, as written).  T = 13 leaves a short last 10-frame block;
    B = 2 clips must not leak into each other through the 5-frame window."""
    import torch.nn.functional as F
    dev = torch.device("cuda")
    t16 = ops.torch_dtype(dt)
    B, T = 2, 13
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, T, 88, 88, generator=g).to(dev, t16)
    w = (torch.randn(64, 5, 7, 7, generator=g) * 0.06).to(dev, t16)
    bias = (torch.randn(64, generator=g) * 0.2).to(dev)
    slope = torch.rand(64, generator=g) * 0.5
    if slopes == "mixed":
        slope[5::7] = -slope[5::7] - 0.1
    slope = slope.to(dev)
    wk = torch.zeros(64, 5, 7, 8, device=dev, dtype=t16)          # k = (dt*7 + dy)*8 + dx, dx == 7 zero, K 280 -> 288
    wk[..., :7] = w
    wp = torch.zeros(64, 288, device=dev, dtype=t16)
    wp[:, :280] = wk.reshape(64, 280)
    y = torch.empty(B * T, 22, 22, 64, device=dev, dtype=t16)
    ops.stem_pool_fused(x, wp, bias, None if slopes == "swish" else slope, y, B, T, dt)
    torch.cuda.synchronize()
    ref = F.conv3d(x.float()[:, None], w.float()[:, None], bias, stride=(1, 2, 2), padding=(2, 3, 3))   # [B,64,T,44,44]
    ref = ref * torch.sigmoid(ref) if slopes == "swish" else torch.where(ref >= 0, ref, ref * slope.view(1, 64, 1, 1, 1))
    ref = F.max_pool3d(ref, (1, 3, 3), (1, 2, 2), (0, 1, 1))                                            # [B,64,T,22,22]
    got = y.float().view(B, T, 22, 22, 64).permute(0, 4, 1, 2, 3)
    err = (got - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("C,T,lens", [(256, 300, [300, 211, 0]), (128, 700, [700, 246, 245]), (64, 1100, [1100, 502, 1]),
                                      (256, 117, [117, 118 - 1, 1])])
def test_respair_final_equals_the_last_pairs_summed(dt, C, T, lens):
    """l2s_respair_final (the last (c1, c2, d = 5) pairs of a stage's three ResBlocks, k = 3 / 7 / 11, in one launch with the stage
    sum in accumulators) against (a) torch fp32 on each clip alone and (b) the three l2s_respair(last) launches it replaces
    (same kernels' arithmetic; only the order of the fp32 additions of the sum differs)."""
    t16 = ops.torch_dtype(dt)
    B, slope, ks, dil = len(lens), 0.1, (3, 7, 11), 5
    g = torch.Generator().manual_seed(C + T)
    L = torch.tensor(lens, dtype=torch.int32)
    valid = torch.arange(T)[None, :] < L[:, None]
    from lip2speech_unit_amd.packing import pack_conv1d
    xs_l, w1s, w2s, b1s, b2s, ref = [], [], [], [], [], torch.zeros(B, T, C)
    for k in ks:
        x = torch.randn(B, T, C, generator=g)
        xl = _r16(F.leaky_relu(x, slope) * valid[:, :, None], dt)
        w1 = _r16(torch.randn(C, C, k, generator=g) * (C * k) ** -0.5, dt)
        w2 = _r16(torch.randn(C, C, k, generator=g) * (C * k) ** -0.5, dt)
        b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
        for b in range(B):
            n = lens[b]
            if n == 0:
                continue
            xi = xl[b:b + 1, :n].transpose(1, 2)
            t1 = _r16(F.leaky_relu(F.conv1d(xi, w1, b1, padding=(k - 1) // 2 * dil, dilation=dil), slope), dt)
            ref[b, :n] += (F.conv1d(t1, w2, b2, padding=(k - 1) // 2) + torch.where(xi >= 0, xi, xi / slope))[0].t()
        xs_l.append(xl.reshape(B * T, C).to(t16).cuda().contiguous())
        w1s.append(pack_conv1d(w1).to(t16).cuda().contiguous())
        w2s.append(pack_conv1d(w2).to(t16).cuda().contiguous())
        b1s.append(b1.cuda())
        b2s.append(b2.cuda())
    kw = dict(B=B, T=T, C=C, slope=slope, lens=L.cuda(), len_mul=1, dtype=dt)
    y = torch.full((B * T, C), 7.0, device="cuda", dtype=t16)
    ops.respair_final(xs_l, w1s, b1s, w2s, b2s, y, ks=list(ks), dils=[dil] * 3, **kw)
    got = y.float().cpu().view(B, T, C)
    want = F.leaky_relu(ref, slope)
    tol = (3e-3 if dt == ops.F16 else 2e-2) * ref.abs().max().item()
    assert (got - want).abs().max().item() < tol
    if (~valid).any():
        assert got[~valid].abs().max().item() == 0.0
    # (b) the launches it replaces
    xs = torch.zeros(B * T, C, device="cuda")
    y2 = torch.full((B * T, C), 7.0, device="cuda", dtype=t16)
    for j, k in enumerate(ks):
        ops.respair(xs_l[j], w1s[j], b1s[j], w2s[j], b2s[j], k=k, dil=dil, xs=xs, y=y2 if j == 2 else None, accumulate=j > 0, **kw)
    torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs().max().item()
    assert d <= (2e-3 if dt == ops.F16 else 1.6e-2) * ref.abs().max().item(), d
    with pytest.raises(ops.L2SError):      # a fourth ResBlock is not built
        ops.respair_final(xs_l + xs_l[:1], w1s + w1s[:1], b1s + b1s[:1], w2s + w2s[:1], b2s + b2s[:1], y, ks=[3, 7, 11, 3], dils=[5] * 4, **kw)


@pytest.mark.parametrize("slopes", ["positive", "mixed"])
@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
def test_stem_pool_fused_vs_two_step_launches(slopes, dt):
    """The fused stem against l2s_stem_conv3d + l2s_maxpool2d_3x3s2 (the reference's order: fp32 PReLU on the conv value, one
    rounding, then the pool; avhubert/resnet.py:137-141).  With every slope >= 0 the fused kernel stores the conv tile rounded to
    16 bits, pools, then applies PReLU and rounds again: NEGATIVE outputs are rounded twice - round16(s * round16(a)) instead of
    round16(s * a).  The two kernels also sum the 245 taps in different orders (different tiles), which shows as one 16-bit ulp
    on a small share of the outputs of either sign: the bound is 2 ulp, < 2 % of the outputs differing."""
    dev = torch.device("cuda")
    t16 = ops.torch_dtype(dt)
    B, T = 2, 7
    g = torch.Generator().manual_seed(4321)
    x = torch.randn(B, T, 88, 88, generator=g).to(dev, t16)
    w = (torch.randn(64, 5, 7, 7, generator=g) * 0.06).to(dev, t16)
    bias = (torch.randn(64, generator=g) * 0.2 - 0.3).to(dev)          # shifted down: plenty of negative pooled maxima
    slope = torch.rand(64, generator=g) * 0.5
    if slopes == "mixed":
        slope[5::7] = -slope[5::7] - 0.1
    slope = slope.to(dev)
    wk = torch.zeros(64, 5, 7, 8, device=dev, dtype=t16)
    wk[..., :7] = w
    wp = torch.zeros(64, 288, device=dev, dtype=t16)
    wp[:, :280] = wk.reshape(64, 280)
    y_f = torch.empty(B * T, 22, 22, 64, device=dev, dtype=t16)
    ops.stem_pool_fused(x, wp, bias, slope, y_f, B, T, dt)
    conv = torch.empty(B * T, 44, 44, 64, device=dev, dtype=t16)
    ops.stem_conv3d(x, wp, bias, slope, conv, B, T, dt)
    y_2 = torch.empty(B * T, 22, 22, 64, device=dev, dtype=t16)
    ops.maxpool2d_3x3s2(conv, y_2, B * T, 44, 44, 64, dt)
    torch.cuda.synchronize()
    a, b = y_f.view(torch.int16).int(), y_2.view(torch.int16).int()
    diff = (a - b).abs()                     # same sign, adjacent 16-bit patterns differ by exactly 1
    neg = y_2.float() < 0
    assert int(neg.sum()) > 1000, "the case must exercise negative outputs"
    same_sign = (a < 0) == (b < 0)
    assert bool(same_sign[(y_2.float() != 0) & (y_f.float() != 0)].all())
    # First run of this test (round 4): the two kernels also tile the conv's K loop differently, so their fp32 sums differ in the
    # last bit now and then and ONE 16-bit ulp shows on either sign (the "bit-identical for non-negative outputs" expectation
    # failed); the double rounding of negative outputs adds at most one more.  Bound: 2 ulp, and almost all outputs equal.
    d = diff[same_sign]
    assert int(d.max()) <= 2, int(d.max())
    frac = float((d > 0).float().mean())
    print(f"fused stem vs two launches, dtype {dt}, slopes {slopes}: {int((d > 0).sum())} of {d.numel()} outputs differ "
          f"({int((d[neg[same_sign]] > 0).sum())} of them negative), max {int(d.max())} ulp")
    assert frac < 0.02, frac


@pytest.mark.parametrize("T", [1, 2, 3, 27])
def test_stem_pool_fused_short_and_chunked_clips(T, tmp_path):
    """Clips shorter than the 5-frame window (every slab of the first window partly outside the clip) and a clip of two and a bit
    10-frame chunks, against torch; and the same launch with 25 frames per block (`L2S_STEM_FT=25`, read once per process:
    a subprocess) is bit-identical - the chunking only changes which block computes a frame."""
    import subprocess, sys
    import torch.nn.functional as F
    dev = torch.device("cuda")
    B = 3
    g = torch.Generator().manual_seed(T)
    x = torch.randn(B, T, 88, 88, generator=g).half()
    w = (torch.randn(64, 5, 7, 7, generator=g) * 0.06).half()
    bias = torch.randn(64, generator=g) * 0.2
    slope = torch.rand(64, generator=g) * 0.5
    wk = torch.zeros(64, 5, 7, 8, dtype=torch.float16)
    wk[..., :7] = w
    wp = torch.zeros(64, 288, dtype=torch.float16)
    wp[:, :280] = wk.reshape(64, 280)
    y = torch.empty(B * T, 22, 22, 64, device=dev, dtype=torch.float16)
    ops.stem_pool_fused(x.to(dev), wp.to(dev), bias.to(dev), slope.to(dev), y, B, T, ops.F16)
    torch.cuda.synchronize()
    ref = F.conv3d(x.float()[:, None], w.float()[:, None], bias, stride=(1, 2, 2), padding=(2, 3, 3))
    ref = torch.where(ref >= 0, ref, ref * slope.view(1, 64, 1, 1, 1))
    ref = F.max_pool3d(ref, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    got = y.float().cpu().view(B, T, 22, 22, 64).permute(0, 4, 1, 2, 3)
    assert (got - ref).abs().max().item() <= 3e-3 * ref.abs().max().item()
    if T > 10:
        torch.save({"x": x, "wp": wp, "bias": bias, "slope": slope}, tmp_path / "in.pt")
        code = ("import sys, torch; sys.path.insert(0, %r); from lip2speech_unit_amd import ops; d = torch.load(%r); "
                "B, T = d['x'].shape[:2]; y = torch.empty(B * T, 22, 22, 64, device='cuda', dtype=torch.float16); "
                "ops.stem_pool_fused(d['x'].cuda(), d['wp'].cuda(), d['bias'].cuda(), d['slope'].cuda(), y, B, T, ops.F16); "
                "torch.cuda.synchronize(); torch.save(y.cpu(), %r)") % (
                    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "in.pt"), str(tmp_path / "out.pt"))
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, L2S_STEM_FT="25"), capture_output=True, text=True,
                           timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert torch.equal(torch.load(tmp_path / "out.pt").view(torch.int16), y.cpu().view(torch.int16))


def test_stem_u8_rejects_bad_arguments():
    dev = torch.device("cuda")
    w = torch.zeros(64, 288, device=dev, dtype=torch.float16)
    bias = torch.zeros(64, device=dev)
    y = torch.empty(4, 22, 22, 64, device=dev, dtype=torch.float16)
    small = torch.zeros(1, 4, 80, 96, device=dev, dtype=torch.uint8)
    with pytest.raises(ops.L2SError):
        ops.stem_pool_fused_u8(small, w, bias, None, y, 1, 4, ops.F16)                 # frame smaller than the crop
    ok = torch.zeros(1, 4, 96, 96, device=dev, dtype=torch.uint8)
    with pytest.raises(ops.L2SError):
        ops.stem_pool_fused_u8(ok, w, bias, None, y, 1, 4, ops.F16, crop=80)           # only image_crop_size = 88 is built
    with pytest.raises(ValueError):
        ops.stem_pool_fused_u8(ok.float(), w, bias, None, y, 1, 4, ops.F16)


@pytest.mark.parametrize("C,k", [(32, 3), (32, 7), (32, 11), (16, 3), (16, 7), (16, 11)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_resblock_fused_kernel(C, k, accumulate):
    """l2s_resblock_fused (csrc/resblock.hip) against the six convolutions of one HiFi-GAN ResBlock1
    (speech-resynthesis/models.py:34-41) run per clip at its own length; T is not a multiple of the time tile."""
    import torch.nn.functional as F
    dt, slope, dil = ops.F16, 0.1, (1, 3, 5)
    B, T = 3, 1500
    g = torch.Generator().manual_seed(C * 100 + k)
    lens = torch.tensor([T, T - 333, 700])
    x = torch.randn(B, C, T, generator=g)
    xl = _r16(F.leaky_relu(x, slope), dt)
    xin = torch.where(xl >= 0, xl, xl / slope)            # what the kernel recovers from the 16-bit leaky_relu(x)
    ws = [_r16(torch.randn(C, C, k, generator=g) / (C * k) ** 0.5, dt) for _ in range(6)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(6)]
    prev = torch.randn(B, T, C, generator=g)
    ref = torch.zeros(B, T, C)
    for b in range(B):
        L = int(lens[b])
        cur = xin[b:b + 1, :, :L]
        for m, d in enumerate(dil):
            t1 = F.conv1d(_r16(F.leaky_relu(cur, slope), dt), ws[2 * m], bs[2 * m], 1, (k * d - d) // 2, d)
            t2 = F.conv1d(_r16(F.leaky_relu(t1, slope), dt), ws[2 * m + 1], bs[2 * m + 1], 1, (k - 1) // 2, 1)
            cur = t2 + cur
        ref[b, :L] = cur[0].t()
    if accumulate:
        ref = ref + prev
    kpad = ((k * C + 31) // 32) * 32
    wf = torch.zeros(6, C, kpad)
    for i in range(6):
        wf[i, :, : k * C] = ws[i].permute(0, 2, 1).reshape(C, k * C)     # K index = tap * C + c_in
    bf = torch.stack(bs)
    xs = (prev.clone() if accumulate else torch.full((B, T, C), float("nan"))).reshape(B * T, C).cuda()
    nxt = torch.empty(B * T, C, device="cuda", dtype=torch.float16)
    ops.resblock_fused(xl.transpose(1, 2).contiguous().reshape(B * T, C).half().cuda(), wf.half().cuda(), bf.cuda(), xs, nxt,
                       B=B, T=T, C=C, k=k, dil=dil, accumulate=accumulate, slope=slope, lens=lens.int().cuda(), len_mul=1,
                       dtype=dt)
    torch.cuda.synchronize()
    got = xs.cpu().view(B, T, C)
    scale = ref.abs().max().item()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() < 6e-3 * scale, (got - ref).abs().max().item() / scale
    ref_l = F.leaky_relu(ref, slope)
    assert (nxt.float().cpu().view(B, T, C) - ref_l).abs().max().item() < 6e-3 * scale


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("C", [32, 16])
@pytest.mark.parametrize("T,lens,with_out", [(1500, [1500, 1167, 700], True), (2049, [2049, 0, 513, 1], False),
                                             (37, [37, 5], True), (1024, [1024, 513, 512], True)])
def test_resstage_fused_equals_three_resblocks(dt, C, T, lens, with_out):
    """l2s_resstage_fused == l2s_resblock_fused x 3 (in the stage kernel's order; accumulate 0, 1, 1; xl_out on the last), bit for bit:
    the stage kernel runs the same per-ResBlock body on the k = 11 tile geometry, the running sum in registers.  The per-ResBlock kernel is checked
    against torch in test_resblock_fused_kernel.  Ragged / empty clips, T not a multiple of the 512-sample tile."""
    t16 = ops.torch_dtype(dt)
    B, slope = len(lens), 0.1
    ks, dils = (3, 7, 11), ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    g = torch.Generator().manual_seed(C * 1000 + T)
    L = torch.tensor(lens, dtype=torch.int32)
    valid = (torch.arange(T)[None, :] < L[:, None])[:, :, None]
    xl = (F.leaky_relu(torch.randn(B, T, C, generator=g), slope) * valid).reshape(B * T, C).to(t16).cuda()
    ws, bs = [], []
    for k in ks:
        kpad = ((k * C + 31) // 32) * 32
        wf = torch.zeros(6, C, kpad)
        wf[:, :, : k * C] = torch.randn(6, C, k * C, generator=g) / (C * k) ** 0.5
        ws.append(wf.to(t16).cuda())
        bs.append((torch.randn(6, C, generator=g) * 0.1).cuda())
    lens_d = L.cuda()
    xs_a = torch.full((B * T, C), float("nan"), device="cuda")
    nxt_a = torch.zeros(B * T, C, device="cuda", dtype=t16)
    # the stage kernel's order (the fp32 sum is (first + second) + third): k = 11, 7, 3 at C = 32; k = 3, 7, 11 at C = 16
    for n, j in enumerate((2, 1, 0) if C == 32 else (0, 1, 2)):
        ops.resblock_fused(xl, ws[j], bs[j], xs_a, nxt_a if (with_out and n == 2) else None, B=B, T=T, C=C, k=ks[j], dil=dils[j],
                           accumulate=n > 0, slope=slope, lens=lens_d, len_mul=1, dtype=dt)
    xs_b = torch.full((B * T, C), float("nan"), device="cuda")
    nxt_b = torch.zeros(B * T, C, device="cuda", dtype=t16)
    ops.resstage_fused(xl, ws, bs, xs_b, nxt_b if with_out else None, B=B, T=T, C=C, ks=ks, dils=dils, slope=slope,
                       lens=lens_d, len_mul=1, dtype=dt)
    torch.cuda.synchronize()
    assert torch.isfinite(xs_b).all()
    assert torch.equal(xs_a, xs_b)
    assert torch.equal(nxt_a.view(torch.int16), nxt_b.view(torch.int16))
    assert float(xs_b.abs().max()) > 0
    # rows past each clip's length are zero (the reference's padding convention downstream)
    assert float((xs_b.view(B, T, C) * (~valid).cuda()).abs().max()) == 0
    if with_out:
        # xs_final = 0 (nobody reads the stage's fp32 sum): the same xl_out; xs is scratch (ABI 14: its content afterwards
        # is unspecified - at C = 32 the running sum lives in registers, at C = 16 the partial sums pass through xs)
        xs_c = torch.full((B * T, C), 123.0, device="cuda")
        nxt_c = torch.zeros(B * T, C, device="cuda", dtype=t16)
        ops.resstage_fused(xl, ws, bs, xs_c, nxt_c, B=B, T=T, C=C, ks=ks, dils=dils, slope=slope, lens=lens_d, len_mul=1,
                           dtype=dt, xs_final=False)
        torch.cuda.synchronize()
        assert torch.equal(nxt_b.view(torch.int16), nxt_c.view(torch.int16))
        assert torch.isfinite(xs_c).all()
    else:
        with pytest.raises(ops.L2SError):       # xs_final = 0 without xl_out: nothing would be written
            ops.resstage_fused(xl, ws, bs, xs_b, None, B=B, T=T, C=C, ks=ks, dils=dils, slope=slope, lens=lens_d, len_mul=1,
                               dtype=dt, xs_final=False)


def test_resstage_fused_rejects_other_layouts():
    C, T, B = 32, 64, 1
    xl = torch.zeros(B * T, C, device="cuda", dtype=torch.float16)
    xs = torch.zeros(B * T, C, device="cuda")
    ws = [torch.zeros(6, C, ((k * C + 31) // 32) * 32, device="cuda", dtype=torch.float16) for k in (3, 7, 11)]
    bs = [torch.zeros(6, C, device="cuda") for _ in range(3)]
    d = ((1, 3, 5),) * 3
    with pytest.raises(ops.L2SError):
        ops.resstage_fused(xl, ws, bs, xs, None, B=B, T=T, C=C, ks=(3, 5, 11), dils=d, slope=0.1)      # other kernel sizes
    with pytest.raises(ops.L2SError):
        ops.resstage_fused(xl, ws[:2], bs[:2], xs, None, B=B, T=T, C=C, ks=(3, 7), dils=d[:2], slope=0.1)
    with pytest.raises(ops.L2SError):
        ops.resstage_fused(xl, ws, bs, xs, None, B=B, T=T, C=C, ks=(3, 7, 11), dils=((1, 3, 9),) * 3, slope=0.1)
    with pytest.raises(ops.L2SError):
        ops.resstage_fused(xl, ws, bs, xs, None, B=B, T=T, C=C, ks=(3, 7, 11), dils=d, slope=1.5)     # LeakyReLU slope in (0, 1]
    with pytest.raises(ops.L2SError):
        ops.resblock_fused(xl, ws[0], bs[0], xs, None, B=B, T=T, C=C, k=3, dil=(1, 3, 5), accumulate=False, slope=1.5)


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("C,k,dil,T,lens", [(64, 3, 1, 500, [500, 311]), (64, 7, 3, 700, [700, 17]), (64, 11, 5, 400, [390, 400]),
                                            (128, 3, 5, 300, [300, 1]), (128, 7, 1, 555, [200, 555]), (128, 11, 3, 193, [193, 100]),
                                            (128, 11, 5, 2000, [2000, 1999, 1217]),
                                            # edge cases: clip shorter than one tile / than the halo, an empty clip, one sample
                                            (64, 11, 5, 33, [33, 0, 7]), (128, 3, 1, 1, [1, 1]), (64, 7, 5, 191, [0, 191]),
                                            (128, 7, 3, 187, [187, 186, 185, 1]),
                                            # C = 256: csrc/respair256.hip (128-row tiles, phase-staggered weight stream)
                                            (256, 3, 1, 300, [300, 211]), (256, 7, 3, 257, [257, 40]), (256, 11, 5, 400, [400, 399]),
                                            (256, 11, 1, 2000, [2000, 1999, 1217]), (256, 7, 5, 119, [119, 0, 1]),
                                            (256, 3, 5, 1, [1, 1]), (256, 11, 3, 129, [128, 129, 118, 117]),
                                            # C = 128 takes the same kernel with 256-row tiles: lengths around the tile edges
                                            (128, 11, 5, 600, [600, 246, 247, 245]), (128, 3, 3, 257, [257, 256, 254, 0])])
def test_respair_fused_conv_pair(dt, C, k, dil, T, lens):
    """csrc/respair.hip against torch fp32 on each clip ALONE: x' = c2(lrelu(c1(lrelu(x)))) + x with the input / output
    carried as LeakyReLU'd 16-bit copies; mid pair, last pair (overwrite, accumulate, with and without the second output)."""
    t16 = ops.torch_dtype(dt)
    B, slope = len(lens), 0.1
    g = torch.Generator().manual_seed(C * 100 + k * 10 + dil)
    x = torch.randn(B, T, C, generator=g)
    w1 = torch.randn(C, C, k, generator=g) * (C * k) ** -0.5
    w2 = torch.randn(C, C, k, generator=g) * (C * k) ** -0.5
    b1, b2 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    L = torch.tensor(lens, dtype=torch.int32)
    valid = torch.arange(T)[None, :] < L[:, None]
    xl = _r16(F.leaky_relu(x, slope) * valid[:, :, None], dt)          # what the producer stores
    xs0 = torch.randn(B, T, C, generator=g) * valid[:, :, None]
    w1r, w2r = _r16(w1, dt), _r16(w2, dt)
    from lip2speech_unit_amd.packing import pack_conv1d
    dev = dict(w1=pack_conv1d(w1r).to(t16).cuda().contiguous(), w2=pack_conv1d(w2r).to(t16).cuda().contiguous(),
               b1=b1.cuda(), b2=b2.cuda(), x=xl.reshape(B * T, C).to(t16).cuda().contiguous())
    ref = torch.zeros(B, T, C)
    for b in range(B):
        n = lens[b]
        if n == 0:
            continue
        xi = xl[b:b + 1, :n].transpose(1, 2)
        t1 = _r16(F.leaky_relu(F.conv1d(xi, w1r, b1, padding=(k - 1) // 2 * dil, dilation=dil), slope), dt)
        xr = torch.where(xi >= 0, xi, xi / slope)
        ref[b, :n] = (F.conv1d(t1, w2r, b2, padding=(k - 1) // 2) + xr)[0].t()
    kw = dict(B=B, T=T, C=C, k=k, dil=dil, slope=slope, lens=L.cuda(), len_mul=1, dtype=dt)
    tol = (3e-3 if dt == ops.F16 else 2e-2) * ref.abs().max().item()
    # mid pair
    y = torch.full((B * T, C), 7.0, device="cuda", dtype=t16)
    ops.respair(dev["x"], dev["w1"], dev["b1"], dev["w2"], dev["b2"], y=y, **kw)
    got = y.float().cpu().view(B, T, C)
    assert (got - F.leaky_relu(ref, slope)).abs().max().item() < tol
    assert got[~valid].abs().max().item() == 0.0 if (~valid).any() else True
    # last pair: overwrite, then accumulate with the second output
    xs = torch.full((B * T, C), 3.0, device="cuda")
    ops.respair(dev["x"], dev["w1"], dev["b1"], dev["w2"], dev["b2"], xs=xs, **kw)
    assert (xs.cpu().view(B, T, C) - ref).abs().max().item() < tol
    xs = xs0.reshape(B * T, C).cuda().contiguous()
    y2 = torch.full((B * T, C), 7.0, device="cuda", dtype=t16)
    ops.respair(dev["x"], dev["w1"], dev["b1"], dev["w2"], dev["b2"], xs=xs, y=y2, accumulate=True, **kw)
    tot = ref + xs0
    assert (xs.cpu().view(B, T, C) - tot).abs().max().item() < tol
    assert (y2.float().cpu().view(B, T, C) - F.leaky_relu(tot, slope)).abs().max().item() < tol + (2e-3 if dt == ops.F16 else 1.6e-2) * tot.abs().max().item()
    assert torch.isfinite(xs).all() and xs.cpu().view(B, T, C)[~valid].abs().max().item() == 0.0 if (~valid).any() else True
    # final pair of a stage (last = 2): the same y, xs read for the sum and left as it was
    xs3 = xs0.reshape(B * T, C).cuda().contiguous()
    y3 = torch.full((B * T, C), 7.0, device="cuda", dtype=t16)
    ops.respair(dev["x"], dev["w1"], dev["b1"], dev["w2"], dev["b2"], xs=xs3, y=y3, accumulate=True, xs_final=False, **kw)
    torch.cuda.synchronize()
    assert torch.equal(y3.view(torch.int16), y2.view(torch.int16))
    assert torch.equal(xs3.cpu(), xs0.reshape(B * T, C))
    with pytest.raises(ops.L2SError):           # last = 2 without y: nothing would be written
        ops.respair(dev["x"], dev["w1"], dev["b1"], dev["w2"], dev["b2"], xs=xs3, accumulate=True, xs_final=False, **kw)


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("M,N,K,alpha", [(100, 1024, 4096, 1.0), (250, 1024, 1024, 1.0), (200, 512, 2048, 0.5), (500, 512, 2048, 0.5),
                                         (37, 1024, 4096, 1.0)])
def test_residual_linear_splitk_small_m(dt, M, N, K, alpha):
    """ops.residual_linear at one-clip M (split-K: one grouped tap-GEMM launch over S slices of K + l2s_splitk_reduce) against
    the single-launch fp32-residual-stream epilogue: the same products, the slices' fp32 sums added in a fixed order - equal to
    fp32 rounding of a different summation order, and bit-identical from run to run."""
    t16 = ops.torch_dtype(dt)
    S = ops.splitk_slices(M, N, K)
    assert S >= 2 and K % (64 * S) == 0
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(t16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(t16).cuda()
    b = torch.randn(N, generator=g).cuda()
    x0 = torch.randn(M, N, generator=g).cuda()
    ref = x0.clone()
    ops.tapgemm(a, w, ref, M=M, N=N, Cin=K, bias=b, alpha=alpha, R=ref, ldr=N, flags=ops.F_RES_POST, dtype=dt)
    cache = {}
    got = x0.clone()
    ops.residual_linear(a, w, b, got, M=M, N=N, K=K, alpha=alpha, dtype=dt, cache=cache, key="w")
    again = x0.clone()
    ops.residual_linear(a, w, b, again, M=M, N=N, K=K, alpha=alpha, dtype=dt, cache=cache, key="w")
    torch.cuda.synchronize()
    assert ("w", S) in cache
    assert torch.equal(got, again)
    exact = x0.double() + alpha * (a.double() @ w.double().t() + b.double())
    scale = float(exact.abs().max())
    assert float((ref.double() - exact).abs().max()) < 2e-5 * scale
    assert float((got.double() - exact).abs().max()) < 2e-5 * scale
    # above the small-M limit nothing changes: the one-launch form
    assert ops.splitk_slices(ops.SPLITK_MAX_M + 1, N, K) == 0 and ops.splitk_slices(M, N, 512) == 0


def test_splitk_reduce_rejects_bad_arguments():
    P = torch.zeros(8, 4 * 64, device="cuda")
    x = torch.zeros(8, 64, device="cuda")
    ops.splitk_reduce(P, x, M=8, N=64, S=4)
    with pytest.raises(ops.L2SError):
        ops.splitk_reduce(P, x, M=8, N=62, S=4)                 # N % 4
    with pytest.raises(ops.L2SError):
        ops.splitk_reduce(P, x, M=8, N=64, S=4, ldp=128)        # ldp < S * N


@pytest.mark.parametrize("dt", [ops.F16, ops.BF16])
@pytest.mark.parametrize("M,C,S,inplace", [(100, 1024, 8, False), (250, 1024, 4, False), (200, 512, 8, False), (200, 512, 8, True),
                                           (3, 512, 2, True)])
def test_splitk_reduce_layernorm_equals_the_two_launches(dt, M, C, S, inplace):
    """l2s_splitk_reduce_layernorm against l2s_splitk_reduce then l2s_layernorm (16-bit output, and the conformer's norm_final:
    fp32 written over the stream itself): the updated stream bit for bit (same additions, same order), the LayerNorm to fp32
    rounding (hipcc contracts the two kernels' multiply-adds differently: first run 3e-7 relative, <= 1 ulp of the 16-bit
    output); rows past a clip's length come out zero as from l2s_layernorm."""
    t16 = ops.torch_dtype(dt)
    g = torch.Generator().manual_seed(M * 7 + C + S)
    P = torch.randn(M, S * C, generator=g).cuda()
    x0 = (torch.randn(M, C, generator=g) * 3 + 0.5).cuda()
    gamma, beta = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    lens = torch.tensor([M - M // 3], dtype=torch.int32).cuda()
    for use_lens in (False, True):
        kw = dict(lens=lens, len_mul=1, mask_T=M) if use_lens else {}
        xa = x0.clone()
        ops.splitk_reduce(P, xa, M=M, N=C, S=S)
        ya = xa if inplace else torch.empty(M, C, device="cuda", dtype=t16)
        ops.layernorm(xa, gamma, beta, 1e-5, ya, M=M, C=C, dtype=dt, **kw)
        xb = x0.clone()
        yb = xb if inplace else torch.empty(M, C, device="cuda", dtype=t16)
        ops.splitk_reduce_layernorm(P, xb, gamma, beta, 1e-5, yb, M=M, C=C, S=S, dtype=dt, **kw)
        torch.cuda.synchronize()
        if inplace:
            assert float((xa - xb).abs().max()) <= 2e-6 * float(xa.abs().max())
        else:
            assert torch.equal(xa, xb)
            d16 = (ya.view(torch.int16).int() - yb.view(torch.int16).int()).abs()
            assert int(d16.max()) <= 1 and float((d16 > 0).float().mean()) < 0.01      # at most 1 ulp, on < 1 % of the elements
            if use_lens:
                assert float(yb[int(lens[0]):].float().abs().max()) == 0
    with pytest.raises(ops.L2SError):
        ops.splitk_reduce_layernorm(P[:, :S * 256], x0[:, :256].contiguous(), gamma[:256], beta[:256], 1e-5,
                                    torch.empty(M, 256, device="cuda", dtype=t16), M=M, C=256, S=S, dtype=dt)   # other widths
