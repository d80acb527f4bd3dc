"""Generate tests/golden/*.npz by running the REAL reference (build container only).

    python -m oracle.make_golden            # all cases
    python -m oracle.make_golden NAME ...   # selected cases

One subprocess per case (the reference reads hparams at import time, oracle/ref_driver.py).  Each
fixture holds the reference's OUTPUT, the reference's schedule tables, a few weight probes and input
checksums; inputs and weights themselves are re-derived from seeds (diffsinger_amd/synth.py,
oracle/diffnet_oracle.init_diffnet_params) - 60 MB of weights cannot be committed."""
from __future__ import annotations

import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def checksum(t) -> np.ndarray:
    a = t.detach().double()
    return np.array([a.sum().item(), a.abs().sum().item(), float(a.flatten()[0]), float(a.flatten()[-1])])


def weight_probe(state: dict) -> np.ndarray:
    """A few float64 statistics per parameter tensor, in sorted key order."""
    rows = []
    for k in sorted(state):
        a = state[k].detach().double().flatten()
        rows.append([a.sum().item(), a.abs().sum().item(), a[0].item(), a[-1].item()])
    return np.array(rows)


def run_case(name: str):
    import torch
    sys.path.insert(0, ROOT)
    from oracle.golden_cases import CASES, WEIGHT_SEED, FINAL_PROJ_STD
    from oracle.ref_driver import Reference
    from diffsinger_amd.synth import presets, make_inputs

    case = CASES[name]
    pre = presets()[case['preset']]
    ref = Reference(pre['source'])
    hp = ref.hparams
    k_step = case.get('k_step', hp['K_step'])
    if case.get('legacy'):
        net, gd = ref.build_legacy(WEIGHT_SEED, FINAL_PROJ_STD, hp['timesteps'], hp['spec_min'], hp['spec_max'])
    else:
        net, gd = ref.build(WEIGHT_SEED, FINAL_PROJ_STD, hp['timesteps'], k_step, hp['spec_min'], hp['spec_max'])
    B, T = case['B'], case['T']
    kind = case['kind']
    n_noise = k_step if kind == 'ddpm' else 0
    inp = make_inputs(case['seed'], B, T, n_noise=n_noise, with_fs2_mel=(kind == 'ddpm' and not case['gaussian']),
                      spec_min=pre['spec_min'], spec_max=pre['spec_max'])
    cond = inp['cond']
    out = {}
    with torch.no_grad():
        if kind == 'denoise':
            t = torch.tensor(case['t'], dtype=torch.long)
            out['out'] = net(inp['x_T'], t, cond=cond).numpy()
        elif kind == 'ddpm':
            if case['gaussian']:
                x = inp['x_T']
            else:       # shallow_diffusion_tts.py:250-255
                f = gd.norm_spec(inp['fs2_mel']).transpose(1, 2)[:, None, :, :]
                x = gd.q_sample(x_start=f, t=torch.tensor([k_step - 1]).long(), noise=inp['q_noise'])
                out['x_start'] = x.numpy()
            if case.get('legacy'):
                x = ref.sample_ddpm_legacy(gd, x, cond, list(inp['noise']))
            else:
                x = ref.sample_ddpm(gd, x, cond, list(inp['noise']), k_step)
            out['x_final'] = x.numpy()
            out['out'] = gd.denorm_spec(x[:, 0].transpose(1, 2)).numpy()       # :271,:275
        elif kind == 'plms':
            x = ref.sample_plms(gd, inp['x_T'], cond, k_step, case['interval'])
            out['x_final'] = x.numpy()
            out['out'] = gd.denorm_spec(x[:, 0].transpose(1, 2)).numpy()
    for k in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
              'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod',
              'sqrt_recipm1_alphas_cumprod', 'posterior_variance', 'posterior_log_variance_clipped',
              'posterior_mean_coef1', 'posterior_mean_coef2'):
        out['sched_' + k] = getattr(gd, k).numpy()
    out['spec_min'] = gd.spec_min.numpy()
    out['spec_max'] = gd.spec_max.numpy()
    out['weight_probe'] = weight_probe(net.state_dict())
    out['checksum_cond'] = checksum(cond)
    out['checksum_x_T'] = checksum(inp['x_T'])
    if n_noise:
        out['checksum_noise'] = checksum(inp['noise'])
    out['torch_version'] = np.array(torch.__version__)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(GOLDEN_DIR, name + '.npz'), **out)
    print(f'{name}: out shape {out["out"].shape} max|out| {np.abs(out["out"]).max():.4f}')


def main(argv):
    sys.path.insert(0, ROOT)
    from oracle.golden_cases import CASES
    if len(argv) >= 2 and argv[0] == '--child':
        run_case(argv[1])
        return
    names = argv or list(CASES)
    for n in names:
        subprocess.run([sys.executable, '-m', 'oracle.make_golden', '--child', n], cwd=ROOT, check=True)


if __name__ == '__main__':
    main(sys.argv[1:])
