"""GPU: the persistent K-step loop in its DIRECT-convolution form (csrc/dsd_loop.hpp, k_loop: one kernel for the whole loop, x and the skip
sum resident in registers, halo exchange between neighbouring workgroups) against the per-layer-kernel hipGraph path.  Same arithmetic in the
same order -> BIT-identical; every inter-workgroup wait satisfied (no timeout); also against the reference fixtures.  The Winograd form -
the default of the persistent path - is tests/test_gpu_wino.py; this file is the bit-identity anchor."""
import numpy as np
import pytest
import torch

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _direct_convolution(monkeypatch):
    monkeypatch.setenv('DSD_CONV', 'direct')         # read at dsd_create: every engine of this file runs k_loop, not k_loop_wino


def _run(name, loop_mode):
    from tests.gpu_helpers import build_hip
    case, pre, cfg, k_step, inp, smin, smax = H.case_setup(name)
    gd, _, _ = build_hip(case['preset'], k_step, legacy=bool(case.get('legacy')))
    cond = inp['cond'].transpose(1, 2).contiguous().cuda().transpose(1, 2)
    eng = gd._engine(cond)
    eng.set_loop_mode(loop_mode)
    with torch.no_grad():
        if case['kind'] == 'plms':
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), K_step=k_step, pndm_speedup=case['interval'])
        elif case['gaussian']:
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), noise=inp['noise'].cuda(), K_step=k_step, pndm_speedup=0)
        else:
            out = gd.inference(cond, fs2_mels=inp['fs2_mel'].cuda(), q_noise=inp['q_noise'].cuda(), noise=inp['noise'].cuda(),
                               K_step=k_step, pndm_speedup=0, gaussian_start=False)
    used = eng.loop_mode()
    tmo = eng.loop_timeouts()
    return out.cpu().numpy(), used, tmo


@pytest.mark.parametrize('name', ['ddpm_lj_k100', 'shallow_opencpop_k60', 'shallow_popcs_k51', 'plms_opencpop_i40', 'plms_opencpop_i250'])
def test_persistent_loop_equals_per_layer_kernels(name):
    a, used_a, tmo_a = _run(name, 1)
    b, used_b, _ = _run(name, 0)
    assert used_a == 1 and used_b == 0
    assert tmo_a == 0, 'an inter-workgroup wait timed out'
    np.testing.assert_array_equal(a, b)
    g = H.load_golden(name)['out']
    scale = max(1.0, float(np.abs(g).max())) if 'plms' in name else 1.0
    assert float(np.abs(a - g).max()) / scale <= 1e-4


@pytest.mark.parametrize('B,T,K', [(8, 1024, 12), (5, 2048, 6), (3, 1000, 8), (2, 33, 5), (1, 5, 3)])
def test_persistent_loop_full_width(B, T, K):
    """Bench-size batches: 256 workgroups at once (8 x 1024), chunks of whole utterances (5 x 2048 -> 4 + 1), ragged T, a two-tile utterance
    with a one-frame tail, a one-tile utterance."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    pre = presets()['opencpop_ds60_rel']                       # dilation cycle 4: halos up to 8 frames
    hparams.clear()
    diffsinger_amd.use_preset('opencpop_ds60_rel')
    torch.manual_seed(3)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(5)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)
    outs = []
    for mode in (1, 0):
        eng.set_loop_mode(mode)
        with torch.no_grad():
            outs.append(gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).cpu().numpy())
        assert eng.loop_mode() == mode
        assert eng.loop_timeouts() == 0
    np.testing.assert_array_equal(outs[0], outs[1])
    assert np.isfinite(outs[0]).all()


def test_persistent_loop_under_concurrent_load():
    """Hand-offs under UNEVEN load: a side stream keeps the chip busy with unrelated GEMMs (delays workgroup start-up, skews the
    tiles, thrashes the L2s) while the persistent loop runs - results must stay bit-identical and no wait may time out."""
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    pre = presets()['opencpop_ds60_rel']
    hparams.clear()
    diffsinger_amd.use_preset('opencpop_ds60_rel')
    torch.manual_seed(4)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    B, T, K = 6, 1024, 10
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(6)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)
    eng.set_loop_mode(1)
    with torch.no_grad():
        ref = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).cpu().numpy()
    assert eng.loop_mode() == 1 and eng.loop_timeouts() == 0
    a = torch.randn(4096, 4096, device='cuda')
    big = torch.randn(64 * 1024 * 1024, device='cuda')
    side = torch.cuda.Stream()
    for rep in range(4):
        with torch.cuda.stream(side):
            for _ in range(12):
                a = (a @ a).clamp_(-1, 1)          # MFMA-heavy
                big.mul_(1.0001)                   # streams 256 MB through the L2s / HBM
        with torch.no_grad():
            out = gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0).cpu().numpy()
        assert eng.loop_timeouts() == 0, 'an inter-workgroup wait timed out under load'
        np.testing.assert_array_equal(out, ref, err_msg=f'repetition {rep}')
    torch.cuda.synchronize()


def test_starved_persistent_loop_is_loud_and_the_retry_succeeds():
    """a foreign kernel that holds compute units the persistent loop needs (here: 64 CUs for 10 s on a side stream,
    dsd_debug_hold_cus) drives the loop's inter-workgroup waits into their spin bound - the mel tiles are NaN.  That must RAISE:
    with check=True at the call itself (what forward(infer=True) does), without it at the NEXT call into the engine, never rc = 0 with
    NaNs handed on.  After the report the engine is parked on the hipGraph path: the retry succeeds while the foreign kernel is still
    resident and equals the clean result bit for bit."""
    import time
    import diffsinger_amd
    from diffsinger_amd import hparams
    from diffsinger_amd.synth import presets
    pre = presets()['lj_ds_beta6']
    hparams.clear()
    diffsinger_amd.use_preset('lj_ds_beta6')
    torch.manual_seed(5)
    net = diffsinger_amd.DIFF_DECODERS['wavenet'](hparams)
    torch.nn.init.normal_(net.output_projection.weight, std=0.02)
    B, T, K = 8, 1024, 3                                        # 256 tiles: the loop needs every CU of the chip
    gd = diffsinger_amd.GaussianDiffusion(None, 80, net, timesteps=100, K_step=K, loss_type='l1', spec_min=pre['spec_min'],
                                          spec_max=pre['spec_max']).cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(8)
    cond = torch.randn(B, T, 256, device='cuda', generator=g).transpose(1, 2)
    x_T = torch.randn(B, 1, 80, T, device='cuda', generator=g)
    noise = torch.randn(K, B, 1, 80, T, device='cuda', generator=g)
    eng = gd._engine(cond)
    eng.set_loop_mode(1)
    run = lambda **kw: gd.inference(cond, x_T=x_T, noise=noise, K_step=K, pndm_speedup=0, **kw)
    ref = run(check=True).cpu().numpy()
    assert eng.loop_mode() == 1 and np.isfinite(ref).all()

    # (a) check=True: the call itself raises
    # the holders stay until release_cus() (bounded at 120 s), on a stream hold_cus has PROBED to run beside the current one (inside the
    # whole GPU suite a fresh torch stream can share its hardware queue with the current stream: the loop then simply waits behind the
    # holders and finishes clean - the earlier form of this test failed that way in full-suite runs only)
    side = eng.hold_cus(64, 120000)                            # returns when all 64 holders are resident beside this stream
    t0 = time.time()
    try:
        with pytest.raises(RuntimeError, match='spin bound'):
            run(check=True)
        print(f'starved loop reported after {time.time() - t0:.1f} s (side stream found at attempt {eng._hold_attempts})')
        assert eng.loop_mode() == 0                             # parked on the hipGraph path
        out = run(check=True).cpu().numpy()                     # the retry, with the holders still resident
        np.testing.assert_array_equal(out, ref)
    finally:
        eng.release_cus()
        side.synchronize()

    # (b) without check: nothing waits, the NaNs come back - and the next call into the engine raises
    eng.set_loop_mode(1)
    assert eng.loop_mode() == 1
    side = eng.hold_cus(64, 120000)
    try:
        mel = run()
        torch.cuda.current_stream().synchronize()               # (not the device: the holders' stream is still busy)
        assert not bool(torch.isfinite(mel).all()), 'the starved loop should have poisoned its tiles'
        with pytest.raises(RuntimeError, match='spin bound'):
            run()
        out = run(check=True).cpu().numpy()
        np.testing.assert_array_equal(out, ref)
    finally:
        eng.release_cus()
        side.synchronize()
    eng.set_loop_mode(1)                                        # re-armed: the persistent loop works again once the chip is free
    np.testing.assert_array_equal(run(check=True).cpu().numpy(), ref)
    assert eng.loop_mode() == 1 and eng.loop_timeouts() == 0
