"""GPU parity of the FastSpeech2 / FastSpeech2MIDI HIP path: each operator of include/dsf.h against
torch-CPU fp32 on seeded inputs, then the whole modules against the fixtures generated from the reference's own modules.

Tolerances: fp32 throughout; the differences are reduction order (MFMA k-ordered fmaf chain vs oneDNN blocking), erf / exp
implementations and the online softmax.  Operators: <= 2e-5 max-abs on O(1) outputs; whole model (8 FFT blocks deep, values up
to ~5): <= 1e-4 max-abs on decoder_inp and mel_out, <= 1e-4 on the predictor outputs; integer outputs (mel2ph) exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.fs2_cases import CASES
from tests import fs2_helpers as FH

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda', 0)


def _keep(B, T, g):
    lens = [T] + [max(1, T - 5 * b - 2) for b in range(1, B)]
    keep = torch.zeros(B, T)
    for b, n in enumerate(lens):
        keep[b, :n] = 1
    return keep


@pytest.mark.parametrize('B,T,Ci,Co,K,act,scale,use_res,use_keep', [
    (2, 77, 256, 768, 1, 'none', 1.0, False, False),          # attention in-projection
    (2, 77, 256, 256, 1, 'none', 1.0, True, True),            # out-projection + residual + mask
    (2, 77, 256, 1024, 9, 'gelu', 9 ** -0.5, False, False),   # ffn_1
    (2, 77, 1024, 256, 1, 'none', 1.0, True, True),           # ffn_2
    (3, 40, 128, 256, 5, 'none', 1.0, False, False),          # cwt predictor conv (idim 128)
    (2, 33, 256, 80, 1, 'none', 1.0, False, True),            # mel_out
    (2, 64, 256, 1, 1, 'none', 1.0, False, True),             # duration linear
    (3, 1, 256, 128, 1, 'relu', 1.0, False, False),           # cwt_stats_layers on one frame
    (1, 200, 256, 256, 3, 'relu', 1.0, False, False),         # duration predictor conv
])
def test_conv1d(B, T, Ci, Co, K, act, scale, use_res, use_keep):
    from diffsinger_amd import fs2
    g = torch.Generator().manual_seed(B * 1000 + T + Ci + Co + K)
    x = torch.randn(B, T, Ci, generator=g)
    w = torch.randn(Co, Ci, K, generator=g) * (Ci * K) ** -0.5
    bias = torch.randn(Co, generator=g) * 0.1
    res = torch.randn(B, T, Co, generator=g) if use_res else None
    keep = _keep(B, T, g) if use_keep else None
    ref = F.conv1d(x.transpose(1, 2), w, bias, padding=K // 2) * scale
    ref = F.gelu(ref) if act == 'gelu' else (F.relu(ref) if act == 'relu' else ref)
    ref = ref.transpose(1, 2)
    if res is not None:
        ref = ref + res
    if keep is not None:
        ref = ref * keep[:, :, None]
    d = _dev()
    xc = fs2.to_cm(x.to(d))
    assert float(xc[:, :, T:].abs().max() if xc.shape[2] > T else 0) == 0
    outs = {}
    try:
        # both kernels on every shape: 256-row workgroups (k_fs_conv) and the small-grid form whose waves split the contraction (k_fs_conv_ks)
        for mode in (0, 1):
            fs2.set_conv_split(mode)
            out = fs2.conv1d_cm(xc, T, w.to(d), fs2.PackedWeight(), bias.to(d), scale=scale, act=act,
                                residual=fs2.to_cm(res.to(d)) if res is not None else None, keep=keep.to(d).contiguous() if keep is not None else None)
            assert out.shape == (B, Co, fs2.padded_frames(T))
            if out.shape[2] > T:
                assert float(out[:, :, T:].abs().max()) == 0                # the zero-tail invariant
            got = outs[mode] = fs2.from_cm(out, T).cpu()
            err = float((got - ref).abs().max())
            print(f'conv Ci={Ci} Co={Co} K={K} act={act} split={mode}: max-abs err {err:.3e} (max|ref| {float(ref.abs().max()):.2f})')
            assert err <= 2e-5
    finally:
        fs2.set_conv_split(-1)
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-5


def test_conv1d_kernel_choice_by_grid_size():
    """Automatic mode: a launch with few workgroups takes the K-split kernel, a chip-filling one the 256-row kernel - each bit-identical to its
    forced form."""
    from diffsinger_amd import fs2
    d = _dev()
    g = torch.Generator(device=d).manual_seed(3)
    w = torch.randn(256, 256, 3, device=d, generator=g) * 0.03
    bias = torch.randn(256, device=d, generator=g) * 0.1
    for B, T, forced in ((2, 100, 1), (8, 1024, 0)):
        xc = fs2.to_cm(torch.randn(B, T, 256, device=d, generator=g))
        auto = fs2.conv1d_cm(xc, T, w, fs2.PackedWeight(), bias)
        try:
            fs2.set_conv_split(forced)
            same = fs2.conv1d_cm(xc, T, w, fs2.PackedWeight(), bias)
            fs2.set_conv_split(1 - forced)
            other = fs2.conv1d_cm(xc, T, w, fs2.PackedWeight(), bias)
        finally:
            fs2.set_conv_split(-1)
        assert torch.equal(auto, same), (B, T)
        assert not torch.equal(auto, other) and float((auto - other).abs().max()) <= 2e-5


@pytest.mark.parametrize('eps,relu_in,use_keep', [(1e-5, False, False), (1e-5, False, True), (1e-12, True, True)])
def test_layer_norm(eps, relu_in, use_keep):
    from diffsinger_amd import fs2
    g = torch.Generator().manual_seed(11)
    B, T, Cc = 3, 70, 256
    x = torch.randn(B, T, Cc, generator=g) * 2 + 0.3
    x[1, 50:] = 0                                                       # all-zero (padded) frames -> LN gives beta
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    keep = _keep(B, T, g) if use_keep else None
    ref = F.layer_norm(F.relu(x) if relu_in else x, (Cc,), gamma, beta, eps)
    if keep is not None:
        ref = ref * keep[:, :, None]
    d = _dev()
    out = fs2.layer_norm_cm(fs2.to_cm(x.to(d)), T, gamma.to(d), beta.to(d), eps, relu_in=relu_in, keep=keep.to(d).contiguous() if keep is not None else None)
    got = fs2.from_cm(out, T).cpu()
    err = float((got - ref).abs().max())
    print(f'layer_norm eps={eps} relu_in={relu_in}: max-abs err {err:.3e}')
    assert err <= 2e-5
    assert float(out[:, :, T:].abs().max()) == 0


@pytest.mark.parametrize('B,T', [(2, 45), (3, 130)])
def test_attention(B, T):
    from diffsinger_amd import fs2
    g = torch.Generator().manual_seed(T)
    Cc, heads, hd = 256, 2, 128
    qkv = torch.randn(B, T, 3 * Cc, generator=g)
    pad = _keep(B, T, g) == 0
    q, k, v = [t.reshape(B, T, heads, hd).permute(0, 2, 1, 3) for t in qkv.chunk(3, -1)]
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    s = s.masked_fill(pad[:, None, None, :], float('-inf'))
    ref = (F.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B, T, Cc)
    d = _dev()
    out = fs2.attention_cm(fs2.to_cm(qkv.to(d)), T, pad.to(torch.uint8).to(d).contiguous(), heads)
    got = fs2.from_cm(out, T).cpu()
    err = float((got - ref).abs().max())
    print(f'attention B={B} T={T}: max-abs err {err:.3e}')
    assert err <= 2e-5


def _run_hip(name):
    case, m, hp, params, inp = FH.case_setup(name)
    d = _dev()
    m = m.to(d)
    kw = {k: v.to(d) for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        r = m(inp['txt_tokens'].to(d), infer=True, **kw)
    torch.cuda.synchronize()
    from diffsinger_amd import _lib
    assert _lib._lib is not None, 'libdsdenoise.so was not loaded'
    return {k: v.detach().cpu().numpy() for k, v in r.items() if isinstance(v, torch.Tensor)}


@pytest.mark.parametrize('name', list(CASES))
def test_fs2_matches_reference(name):
    g = FH.load_golden(name)
    out = _run_hip(name)
    np.testing.assert_array_equal(out['mel2ph'], g['mel2ph'])
    for k in ('dur', 'pitch_pred', 'cwt', 'energy_pred', 'decoder_inp', 'mel_out'):
        if k not in g:
            continue
        assert out[k].shape == g[k].shape, (k, out[k].shape, g[k].shape)
        err = float(np.abs(out[k] - g[k]).max())
        print(f'{name}:{k}: max-abs err {err:.3e} (max|ref| {float(np.abs(g[k]).max()):.2f})')
        assert err <= 1e-4, (name, k, err)
    f = np.abs(out['f0_denorm'] - g['f0_denorm']) if 'f0_denorm' in g else np.zeros(1)
    assert float((f / np.maximum(np.abs(g.get('f0_denorm', np.ones(1))), 1.0)).max()) <= 1e-4


def test_fs2_feeds_the_diffusion_hot_path():
    """FastSpeech2 -> GaussianDiffusion.forward(infer=True) stand-alone (outside the reference tree): shapes, masks, finiteness."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    case, fs2m, hp, params, inp = FH.case_setup('fs2_popcs_teacher')
    d = _dev()
    from diffsinger_amd.synth import presets
    pre = presets()[case['preset']]
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=pre['timesteps'], K_step=pre['K_step'], loss_type='l1',
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max'], fs2=fs2m).to(d).eval()
    kw = {k: v.to(d) for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        ret = gd(inp['txt_tokens'].to(d), infer=True, **kw)
    B, T = inp['mel2ph'].shape
    assert ret['mel_out'].shape == (B, T, 80) and ret['fs2_mel'].shape == (B, T, 80)
    assert bool(torch.isfinite(ret['mel_out']).all())
    assert float(ret['mel_out'][inp['mel2ph'].to(d) == 0].abs().max()) == 0       # `* (mel2ph > 0)` (:273)


@pytest.mark.parametrize('name', list(CASES))
def test_fused_glue_equals_the_torch_op_sequence_bit_for_bit(name):
    """Round 6: the index / mask glue of the forward (positions, padding masks, embeddings, the length regulator's gather, `* nonpadding`, the
    [B,T,C] -> channel-major transposition) runs as four HIP operators (include/dsf.h dsf_positions / dsf_input_cm / dsf_gather_frames /
    dsf_sum_embed) instead of ~110 torch launches.  They perform the reference's operations in the reference's order: every output of the
    forward - all 11 reference-generated cases: frame / phone / cwt pitch, speaker ids / d-vectors, energy, MIDI encoder, predicted
    durations - must carry the SAME BITS as with the torch op sequence (fs2.set_glue(False))."""
    from diffsinger_amd import fs2
    try:
        fs2.set_glue(True)
        fast = _run_hip(name)
        fs2.set_glue(False)
        slow = _run_hip(name)
    finally:
        fs2.set_glue(True)
    assert set(fast) == set(slow)
    for k in sorted(fast):
        assert fast[k].shape == slow[k].shape and fast[k].dtype == slow[k].dtype, k
        np.testing.assert_array_equal(fast[k], slow[k], err_msg=f'{name}:{k}')


@torch.no_grad()
def test_token_masks_operator():
    """dsf_token_masks: (v > 0).float(), v == 0, (~(v == 0)).float() of an int64 index tensor in one launch (fs2.py:98, :127, :157, :199;
    tts_modules.py:109) - zeros, negative values, a length that is no multiple of the workgroup."""
    from diffsinger_amd import fs2
    d = _dev()
    g = torch.Generator().manual_seed(3)
    v = torch.randint(-2, 5, (3, 1001), generator=g)
    gt, eq, ne = fs2.token_masks_op(v.to(d), gt0=True, eq0=True, ne0=True)
    assert eq.dtype == torch.bool
    assert torch.equal(gt.cpu(), (v > 0).float()) and torch.equal(eq.cpu(), v == 0) and torch.equal(ne.cpu(), (~(v == 0)).float())
    gt2, eq2, ne2 = fs2.token_masks_op(v.to(d), eq0=True)
    assert gt2 is None and ne2 is None and torch.equal(eq2.cpu(), v == 0)


@pytest.mark.parametrize('norm', ['log', 'standard'])
@torch.no_grad()
def test_pitch_coarse_operator_equals_the_torch_op_sequence_bit_for_bit(norm):
    """dsf_pitch_coarse = denorm_f0 + f0_to_coarse (utils/pitch_utils.py:64-77, :21-30; 22 elementwise launches of the reference's op sequence on
    the device) as one launch: f0_denorm and the quantised pitch carry the SAME BITS - voiced / unvoiced (float and bool uv), padding frames,
    a strided f0 view (the predictor's column), values below and above the quantiser's range, both normalisations; and the split forms
    (torch's own pow / log kernels around the operator's stages)."""
    from diffsinger_amd import fs2
    d = _dev()
    g = torch.Generator().manual_seed(11)
    B, T = 5, 1531
    hp = dict(pitch_norm=norm, use_uv=True, f0_mean=214.0, f0_std=63.5)
    if norm == 'log':
        f0 = torch.rand(B, T, generator=g) * 9 + 3                      # 8 Hz ... 4 kHz: both clamps of the quantiser
    else:
        f0 = torch.randn(B, T, generator=g) * 3
    f0[0, :7] = torch.tensor([0.0, -0.0, 1e-30, -5.0, 20.0, 1.0, 7.123])
    pred = torch.stack([f0, torch.randn(B, T, generator=g)], -1).to(d)    # [B,T,2]: column 0 is a strided view like pitch_pred[:, :, 0]
    uv_f = (torch.rand(B, T, generator=g) > 0.7).float()
    mel2ph = torch.randint(0, 4, (B, T), generator=g)
    for f0_in, uv, m2p in ((f0.to(d), uv_f.to(d), mel2ph.to(d)), (pred[:, :, 0], pred[:, :, 1] > 0, None), (f0.to(d), None, mel2ph.to(d)),
                           (f0.to(d), uv_f.to(d).bool(), None)):
        want_den = fs2.denorm_f0(f0_in.clone(), uv, hp, pitch_padding=(m2p == 0) if m2p is not None else None)
        want = fs2.f0_to_coarse(want_den.clone())
        assert fs2._pitch_fusable(f0_in, uv, hp)
        for native in ((True, True), (False, True), (True, False), (False, False)):
            try:
                fs2.set_pitch_native(*native)
                den, got = fs2.pitch_coarse_op(f0_in, uv, m2p, hp)
            finally:
                fs2.set_pitch_native(True, True)
            assert got.dtype == torch.int64 and int(got.min()) >= 1 and int(got.max()) <= 255
            assert torch.equal(den, want_den), (native, float((den - want_den).abs().max()))
            assert torch.equal(got, want), (native, int((got != want).sum()))
    assert int(want.min()) == 1 and (norm != 'log' or int(want.max()) == 255)


@torch.no_grad()
def test_pitch_coarse_every_float_of_the_working_range():
    """The two transcendental steps of dsf_pitch_coarse go through the device library (__ocml_pow_f32 / __ocml_log_f32) as ATen's kernels do.
    EVERY float32 in [1, 14] as the exponent of `2 ** f0` (2 Hz ... 16 kHz: 31.5 M values) and, through them, the arguments of the logarithm
    (1 + f0 / 700 in [1, 24.4]) gives the bits of the torch op sequence; if a torch build ever links a device library that rounds differently,
    this is the test that says so (and fs2.set_pitch_native switches to torch's own kernels around the operator's stages)."""
    from diffsinger_amd import fs2
    d = _dev()
    hp = dict(pitch_norm='log', use_uv=False)
    lo, hi = 0x3F800000, 0x41600000
    step = 1 << 22
    bad_den = bad_idx = 0
    for a in range(lo, hi + 1, step):
        n = min(step, hi + 1 - a)
        f0 = (torch.arange(n, device=d, dtype=torch.int32) + a).view(torch.float32).view(1, n)
        want_den = fs2.denorm_f0(f0, None, hp)
        want = fs2.f0_to_coarse(want_den.clone())
        den, got = fs2.pitch_coarse_op(f0, None, None, hp)
        bad_den += int((den.view(torch.int32) != want_den.view(torch.int32)).sum())
        bad_idx += int((got != want).sum())
    assert bad_den == 0 and bad_idx == 0, (bad_den, bad_idx)


def test_glue_operators_edge_cases():
    """dsf_positions / dsf_input_cm on ragged, partly and fully padded utterances, T not a multiple of 32, a given padding mask; against
    the torch expressions they replace (utils/__init__.py:145-157, tts_modules.py:288-296)."""
    from diffsinger_amd import fs2
    d = _dev()
    g = torch.Generator().manual_seed(5)
    B, T, C = 4, 77, 256
    tok = torch.randint(1, 60, (B, T), generator=g)
    tok[1, 50:] = 0
    tok[2, :] = 0                                              # an empty utterance
    tok[3, 10:20] = 0                                          # padding in the middle (positions skip it)
    want_pos = fs2.make_positions(tok, 0)
    got = fs2.positions_op(tokens=tok.to(d), padding_idx=0).cpu()
    assert torch.equal(got.long(), want_pos)
    x = torch.randn(B, T, C, generator=g)
    x[1, 50:] = 0
    x[2] = 0
    x[3, 5, 0] = 0                                             # channel 0 zero, the frame is not: positions skip it, the mask keeps it
    got = fs2.positions_op(x=x.to(d), padding_idx=0).cpu()
    assert torch.equal(got.long(), fs2.make_positions(x[..., 0], 0))
    emb = fs2.SinusoidalPositionalEmbedding(C, 0, init_size=200)
    alpha = torch.tensor([0.7])
    for mask in (None, tok.eq(0)):
        pm = x.abs().sum(-1).eq(0) if mask is None else mask
        keep = (~pm).float()
        want = (x + alpha * emb(x[..., 0])) * keep[:, :, None]
        tab = emb.table(T).to(d)
        xc, kp, pu = fs2.input_cm_op(x=x.to(d), pos=fs2.positions_op(x=x.to(d)), pos_table=tab, alpha=alpha.to(d),
                                     padding_mask=None if mask is None else mask.to(d))
        assert torch.equal(fs2.from_cm(xc, T).cpu(), want) and torch.equal(kp.cpu(), keep) and torch.equal(pu.cpu().bool(), pm)
        assert xc.shape[2] == 96 and float(xc[:, :, T:].abs().max()) == 0
