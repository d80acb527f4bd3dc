"""The golden cases: what `oracle/make_golden.py` runs THROUGH THE REFERENCE and what the tests replay
through the oracle and the HIP path.  TEST INFRASTRUCTURE ONLY.

Every case is the full-width model of a shipped config (C = H = 256, 20 layers, 80 bins - there are no
checkpoints, weights come from the seeded reference init with the final projection re-drawn
N(0, 0.02^2)) on small B/T so the CPU oracle finishes in seconds."""

WEIGHT_SEED = 1234          # reference default `seed: 1234` (configs/config_base.yaml:5)
FINAL_PROJ_STD = 0.02

CASES = {
    # single denoiser evaluations (usr/diff/net.py:107-130), per-utterance t, odd T
    'denoise_lj': dict(preset='lj_ds_beta6', kind='denoise', B=3, T=77, t=[99, 37, 0], seed=101),
    'denoise_opencpop': dict(preset='opencpop_ds60_rel', kind='denoise', B=2, T=130, t=[59, 3], seed=102),
    # BASELINE config 1/2 shape family: DiffSpeech, Gaussian start, full K=100 DDPM
    'ddpm_lj_k100': dict(preset='lj_ds_beta6', kind='ddpm', B=2, T=96, k_step=100, gaussian=True, seed=103),
    # BASELINE config 3: shallow diffusion from the aux-decoder mel, K=60, dilation cycle 4
    'shallow_opencpop_k60': dict(preset='opencpop_ds60_rel', kind='ddpm', B=2, T=80, k_step=60, gaussian=False, seed=104),
    # PopCS's own shallow setting (K_step 51, cycle 1), T not a multiple of anything
    'shallow_popcs_k51': dict(preset='popcs_ds_beta6', kind='ddpm', B=1, T=50, k_step=51, gaussian=False, seed=105),
    # BASELINE config 4: PNDM/PLMS, 1000-step schedule, pndm_speedup 40 (26 evaluations) and the literal
    # 4-iteration case (pndm_speedup 250, 5 evaluations); reference runs per utterance (B=1 quirk)
    'plms_opencpop_i40': dict(preset='opencpop_ds1000', kind='plms', B=2, T=64, k_step=1000, interval=40, seed=106),
    # row a15: the legacy class usr/diff/diffusion.py::GaussianDiffusion (DiffFsTask): cosine schedule, no K_step,
    # Gaussian start, all `timesteps` steps
    'ddpm_legacy_cosine': dict(preset='lj_ds_beta6', kind='ddpm', B=2, T=72, k_step=100, gaussian=True, legacy=True, seed=108),
    'plms_opencpop_i250': dict(preset='opencpop_ds1000', kind='plms', B=2, T=64, k_step=1000, interval=250, seed=107),
}
