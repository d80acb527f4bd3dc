"""`usr/diff/diffusion.py::GaussianDiffusion` - the reference's FIRST sampler class (only `usr/task.py::DiffFsTask`
builds it, usr/task.py:19-24), SURVEY.md section 8 row a15.

Same math as the shallow-diffusion class, with these differences (usr/diff/diffusion.py:181-334):
  * constructor has no `K_step` (:182-183): the loop always runs all `num_timesteps` steps (:314-318);
  * the schedule is `betas` if given, else ALWAYS the cosine schedule (:192-195) - `hparams['schedule_type']`
    is not consulted;
  * `self.fs2.decoder = None` (:190) and `fs2(..., skip_decoder=True)` (:303-304): no aux mel, always a Gaussian
    start `x = torch.randn(shape)` (:315), DDPM only, no `mel2ph > 0` mask on the output (:319-320).
The loop itself is `GaussianDiffusion.inference` -> one hipGraph replay of the fused HIP kernels."""
from __future__ import annotations

import torch

from .diffusion import GaussianDiffusion as _ShallowGaussianDiffusion, cosine_beta_schedule


class GaussianDiffusion(_ShallowGaussianDiffusion):
    def __init__(self, phone_encoder, out_dims, denoise_fn, timesteps=1000, loss_type='l1', betas=None, spec_min=None,
                 spec_max=None, fs2=None):
        if betas is None:
            betas = cosine_beta_schedule(timesteps)                         # :192-195
        super().__init__(phone_encoder, out_dims, denoise_fn, timesteps=timesteps, K_step=None, loss_type=loss_type,
                         betas=betas, spec_min=spec_min, spec_max=spec_max, fs2=fs2)
        self.K_step = self.num_timesteps                                    # :313 `t = self.num_timesteps`
        if self.fs2 is not None and hasattr(self.fs2, 'decoder'):
            self.fs2.decoder = None                                         # :190

    @torch.no_grad()
    def sample(self, cond, *, x_T=None, noise=None, return_x=False):
        """:313-320 from `cond` on: Gaussian start, `num_timesteps` ancestral steps, de-normalise."""
        return self.inference(cond, x_T=x_T, noise=noise, K_step=self.num_timesteps, pndm_speedup=0, gaussian_start=True,
                              return_x=return_x)

    def forward(self, txt_tokens, mel2ph=None, spk_embed=None, ref_mels=None, f0=None, uv=None, energy=None, infer=False):
        if self.fs2 is None:
            raise RuntimeError('no FastSpeech2 attached (self.fs2): run inside the reference tree, pass fs2=, or call '
                               'sample(cond) with a precomputed conditioner')
        if not infer:
            # :296-311 - fs2(skip_decoder=True, infer=False) UNDER AUTOGRAD (the reference's inside its tree, or the HIP one: its backward runs on
            # HIP kernels since round 3, diffsinger_amd/fs2.py), t ~ U[0, num_timesteps), L1 / L2 on the predicted noise with the `mel2ph != 0`
            # non-padding factor: the diffusion loss back-propagates through `cond` into a trainable FastSpeech2, as in DiffFsTask
            # (usr/task.py:19-24, :56-84); a frozen one (requires_grad False) costs no graph
            ret = self.fs2(txt_tokens, mel2ph, spk_embed, ref_mels, f0, uv, energy, skip_decoder=True, infer=False)
            cond = ret['decoder_inp'].transpose(1, 2)
            b = txt_tokens.shape[0]
            t = torch.randint(0, self.num_timesteps, (b,), device=txt_tokens.device).long()
            x = self.norm_spec(ref_mels).transpose(1, 2)[:, None, :, :]
            ret['diff_loss'] = self.p_losses(x, t, cond, nonpadding=(mel2ph != 0).float())
            return ret
        ret = self.fs2(txt_tokens, mel2ph, spk_embed, ref_mels, f0, uv, energy, skip_decoder=True, infer=infer)
        cond = ret['decoder_inp'].transpose(1, 2)
        ret['mel_out'] = self.sample(cond)
        return ret
