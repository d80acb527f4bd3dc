"""GPU-side test plumbing: build the HIP modules with the oracle's seeded weights for a golden case."""
import torch

import diffsinger_amd
from diffsinger_amd import hparams
from tests import helpers as H


def build_hip(preset_name, k_step=None, legacy=False):
    pre = H.presets()[preset_name]
    hparams.clear()
    diffsinger_amd.use_preset(preset_name)
    cfg = H.net_config(pre)
    net = diffsinger_amd.DIFF_DECODERS[pre['diff_decoder_type']](hparams)
    missing = net.load_state_dict(H.oracle_params(cfg), strict=True)
    if legacy:
        from diffsinger_amd.legacy import GaussianDiffusion as LegacyGD
        gd = LegacyGD(None, pre['audio_num_mel_bins'], net, timesteps=pre['timesteps'], loss_type=pre['diff_loss_type'],
                      spec_min=pre['spec_min'], spec_max=pre['spec_max'])
        return gd.cuda().eval(), cfg, pre
    gd = diffsinger_amd.GaussianDiffusion(None, pre['audio_num_mel_bins'], net, timesteps=pre['timesteps'],
                                          K_step=(k_step or pre['K_step']), loss_type=pre['diff_loss_type'],
                                          spec_min=pre['spec_min'], spec_max=pre['spec_max'])
    return gd.cuda().eval(), cfg, pre


def run_hip_case(name, use_graph=True, tile=0, split=False, loop_mode=None, lat_split=None, conv=None):
    case, pre, cfg, k_step, inp, smin, smax = H.case_setup(name)
    gd, _, _ = build_hip(case['preset'], k_step, legacy=bool(case.get('legacy')))
    cond = inp['cond'].transpose(1, 2).contiguous().cuda().transpose(1, 2)      # [B,H,T] view of [B,T,H], like :238
    eng = gd._engine(cond)
    eng.set_use_graph(use_graph)
    eng.set_layer_tile(tile)
    if loop_mode is not None:
        eng.set_loop_mode(loop_mode)
    if lat_split is not None:
        eng.set_lat_split(lat_split)
    if conv is not None:
        eng.set_conv_mode(conv)                      # convolution of the persistent loop: 'winograd' (default) / 'direct'
    if split:
        eng.set_split_mode(True)                     # EXPERIMENT: layers on the bf16 matrix pipe, six plane products per fp32 product
    kind = case['kind']
    with torch.no_grad():
        if kind == 'denoise':
            out = gd.denoise_fn(inp['x_T'].cuda(), torch.tensor(case['t']).cuda(), cond)
        elif kind == 'ddpm':
            if case.get('legacy'):
                out = gd.sample(cond, x_T=inp['x_T'].cuda(), noise=inp['noise'].cuda())
            elif case['gaussian']:
                out = gd.inference(cond, x_T=inp['x_T'].cuda(), noise=inp['noise'].cuda(), K_step=k_step, pndm_speedup=0)
            else:
                out = gd.inference(cond, fs2_mels=inp['fs2_mel'].cuda(), q_noise=inp['q_noise'].cuda(), noise=inp['noise'].cuda(),
                                   K_step=k_step, pndm_speedup=0, gaussian_start=False)
        else:
            out = gd.inference(cond, x_T=inp['x_T'].cuda(), K_step=k_step, pndm_speedup=case['interval'])
    torch.cuda.synchronize()
    return out.cpu().numpy()
