"""numpy restatement of ``Convofusion._diffusion_reverse`` -- TEST INFRASTRUCTURE (oracle/__init__.py).

Reference: convofusion/models/modeltype/convofusion.py:391-549 (loop), :499-501 (7x replication),
:527-541 (modality guidance combine), :544 (scheduler step), :548 (permute on return); in-painting
variant unbounded_synthesis.py:28-187 (:70-76 overwrite of the first ``preseq_len`` tokens).
The WEG branch (:437-496) enters through ``pre_step`` (restated in oracle/weg_ref.py).

The reference module itself cannot be imported here (pytorch_lightning, torchmetrics, omegaconf ...
are missing), so the loop is restated and drives either ``oracle.denoiser_ref`` or -- in
tests/golden/make_golden.py -- the imported reference ``Denoiser``.
"""
import numpy as np

F32 = np.float32
CFG_CHUNKS = 7  # clf_guidance_drops + 1, convofusion.py:60,399


def cfg_combine(noise_pred, guidance_scale):
    """convofusion.py:527-541.  Chunk order: all_drop, text_only, audio_only, spk_only,
    apb_only, lsnid_only, full."""
    u, t, a, s, p, i, f = np.split(noise_pred, CFG_CHUNKS, axis=0)
    g = F32(guidance_scale)
    n_text = g * F32(1) * (t - u)
    n_audio = g * F32(1) * (a - u)
    n_spk = g * F32(1) * (s - u)
    n_apb = g * F32(1) * (p - u)
    n_lsn = g * F32(1) * (i - u)
    n_all = g * F32(0) * (f - u)
    return (u + (n_text + n_audio + n_spk + n_apb + n_lsn + n_all)).astype(F32)


def diffusion_reverse(denoise_fn, scheduler, encoder_hidden_states, cond_masks, init_latents,
                      step_noise, guidance_scale=7.5, num_inference_steps=1000, eta=0.0,
                      preseq=None, keep_steps=(), return_att=False, pre_step=None, stop_after=None):
    """denoise_fn(sample[7B,L,128], t, enc, masks) -> (eps[7B,L,128], att_mats).
    ``step_noise(i, t)`` returns the [B,L,128] N(0,1) draw for loop index i (used when t > 0
    for DDPM, when eta > 0 for DDIM).  Returns (latents [L,B,128], snapshots, att dict)."""
    latents = (np.asarray(init_latents, dtype=F32) * F32(scheduler.init_noise_sigma)).astype(F32)
    init_noise = latents.copy()
    scheduler.set_timesteps(num_inference_steps)
    is_ddim = hasattr(scheduler, "final_alpha_cumprod")
    snaps, atts = {}, {}
    for i, t in enumerate(scheduler.timesteps):
        if stop_after is not None and i >= stop_after:   # tests that only look at the first snapshots
            break
        if preseq is not None:  # unbounded_synthesis.py:70-76
            pl = preseq.shape[1]
            latents = latents.copy()
            noised = scheduler.add_noise(preseq, init_noise[:, :pl], np.full((preseq.shape[0],), t))
            latents[:, :pl] = noised
            if i == 0:
                # reference aliasing quirk: ``latents = init_noise`` (unbounded_synthesis.py:63) is
                # the SAME tensor at i == 0, so the overwrite at :76 also rewrites the noise that
                # every later iteration re-clones at :72.
                init_noise[:, :pl] = noised
        if pre_step is not None:  # the WEG branch (:437-496): latents = pre_step(i, t, latents)
            latents = pre_step(i, int(t), latents)
        model_in = np.concatenate([latents] * CFG_CHUNKS, axis=0)
        noise_pred, att = denoise_fn(model_in, int(t), encoder_hidden_states, cond_masks)
        if return_att:
            atts[int(t)] = [np.split(a, CFG_CHUNKS, axis=0)[-1] for a in att]
        eps = cfg_combine(noise_pred, guidance_scale)
        if is_ddim:
            latents = scheduler.step(eps, t, latents, eta=eta, noise=step_noise(i, t) if eta > 0 else None)
        else:
            latents = scheduler.step(eps, t, latents, noise=step_noise(i, t) if t > 0 else None)
        if (i + 1) in keep_steps:
            snaps[i + 1] = latents.copy()
    return latents.transpose(1, 0, 2).copy(), snaps, atts
