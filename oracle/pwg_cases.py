"""ParallelWaveGAN fixture cases: what oracle/make_golden_pwg.py runs THROUGH THE REFERENCE generator and what the tests replay through
oracle/pwg_oracle.py and the HIP module.  TEST INFRASTRUCTURE ONLY.  No checkpoints: seeded synthetic weight-normed state (`synth_state`),
seeded inputs (`make_inputs`)."""
import torch

CASES = {
    # the shipped generator (30 layers, 3 stacks: dilations 1 ... 512), two utterances of 10 and 7 mel frames at hop 256
    'pwg_default': dict(gen=dict(), B=2, T_mel=10, seed=301),
    # pitch-conditioned variant (use_pitch_embed: c_proj over [mel; pitch embedding]) with a shorter stack and other scales (hop 64)
    'pwg_pitch': dict(gen=dict(use_pitch_embed=True, layers=12, stacks=2, upsample_params={'upsample_scales': [4, 2, 8]}), B=3, T_mel=23, seed=302),
}


def gen_config(case: dict) -> dict:
    g = dict(case['gen'])
    return {'layers': g.get('layers', 30), 'stacks': g.get('stacks', 3), 'aux_context_window': g.get('aux_context_window', 2),
            'upsample_scales': list(g.get('upsample_params', {'upsample_scales': [4, 4, 4, 4]})['upsample_scales']),
            'use_pitch_embed': bool(g.get('use_pitch_embed', False))}


def synth_state(shapes: dict, seed: int) -> dict:
    """shapes: {state_dict key: shape}.  Weight-normed state (weight_g / weight_v) + biases, deterministic and non-trivial."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(shapes):
        shape = tuple(shapes[k])
        if k.endswith('weight_g'):
            out[k] = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif k.endswith('weight_v'):
            out[k] = torch.randn(shape, generator=g)
        elif k.endswith('bias'):
            out[k] = 0.1 * torch.randn(shape, generator=g)
        elif k == 'pitch_embed.weight':
            out[k] = torch.randn(shape, generator=g) * 0.5
            out[k][0] = 0
        else:                                           # c_proj.weight
            out[k] = torch.randn(shape, generator=g) * shape[1] ** -0.5
    # (weight_norm gives every output row of a layer the norm |g|: g ~ 1 keeps the signal alive through the 30 layers whatever the fan-in)
    return out


def make_inputs(case: dict, cfg: dict) -> dict:
    g = torch.Generator().manual_seed(case['seed'] + 50)
    hop = 1
    for s in cfg['upsample_scales']:
        hop *= s
    B, Tm, ctx = case['B'], case['T_mel'], cfg['aux_context_window']
    inp = {'x': torch.randn(B, 1, Tm * hop, generator=g), 'c': torch.randn(B, 80, Tm + 2 * ctx, generator=g) * 1.5}
    if cfg['use_pitch_embed']:
        inp['pitch'] = torch.randint(0, 300, (B, Tm + 2 * ctx), generator=g)
    return inp
