"""GPU: diffsinger_amd.graphs.GraphedForward - the rows around the hot path replayed as ONE hipGraph (the
denoiser loop has its own cached graphs inside the C library).  The replay must be bit-identical to the eager forward, follow new input
VALUES (the inputs are copied into the graph's static buffers) and re-capture on new shapes."""
import pytest
import torch

from diffsinger_amd.graphs import GraphedForward
from tests import fs2_helpers as FH

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_hifigan_forward_replayed_as_one_graph():
    from diffsinger_amd.vocoder import HifiGanGenerator
    from oracle.make_golden_hifigan import CONFIG
    m = HifiGanGenerator(dict(CONFIG, use_pitch_embed=False))
    m.remove_weight_norm()
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith('weight'):
                p.copy_(torch.randn(p.shape, generator=g) / max(1, p[0].numel()) ** 0.5)
    m = m.to(DEV).eval()
    gm = GraphedForward(m)
    mels = [torch.randn(2, 80, 40, generator=g).to(DEV), torch.randn(2, 80, 40, generator=g).to(DEV), torch.randn(1, 80, 57, generator=g).to(DEV)]
    for mel in mels:
        want = m(mel)
        got = gm(mel)
        assert got.shape == want.shape and torch.equal(got, want)
    assert gm.captures == 2                                  # (2, 80, 40) captured once and replayed for the second mel; (1, 80, 57) is a new graph


def test_fastspeech2_teacher_forced_forward_replayed_as_one_graph():
    case, m, hp, params, inp = FH.case_setup('fs2_lj_teacher')
    m = m.to(DEV)
    tok = inp['txt_tokens'].to(DEV)
    kw = {k: v.to(DEV) for k, v in inp.items() if k != 'txt_tokens'}
    with torch.no_grad():
        want = m(tok, infer=True, **{k: v.clone() for k, v in kw.items()})
    gm = GraphedForward(lambda t, mel2ph, f0, uv: m(t, infer=True, mel2ph=mel2ph, f0=f0, uv=uv))
    for _ in range(2):
        got = gm(tok, kw['mel2ph'], kw['f0'], kw['uv'])
        for k in ('decoder_inp', 'mel_out', 'f0_denorm'):
            assert torch.equal(got[k], want[k]), k
    # new values, same shapes: the graph follows the inputs
    f0b = kw['f0'] * 1.01
    with torch.no_grad():
        want_b = m(tok, infer=True, mel2ph=kw['mel2ph'], f0=f0b.clone(), uv=kw['uv'])
    got_b = gm(tok, kw['mel2ph'], f0b, kw['uv'])
    assert torch.equal(got_b['mel_out'], want_b['mel_out']) and not torch.equal(got_b['mel_out'], want['mel_out'])
    assert gm.captures == 1
    with pytest.raises(TypeError):
        gm(tok, [1, 2], kw['f0'], kw['uv'])
