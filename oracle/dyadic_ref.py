"""numpy restatement of the dyadic reactive loop (BASELINE.json configs[4]) -- TEST
INFRASTRUCTURE (oracle/__init__.py).

Not a reference feature: two denoising loops (convofusion.py:391-549, restated in oracle.sampler_ref) run in
lock-step; before iteration i the conditional ``spkemb`` memory of each side is
``TextAudioMotionFuser.latent_proj`` (condfuser.py:22-27) applied to the PARTNER's current latents [B, L, 128]
(so its key length is L), placed in the guidance chunks that carry the conditional speaker memory (3 and 6,
convofusion.py:909-929).  Parity per denoiser call is defined against the reference ``Denoiser`` fed the same
memories (oracle.denoiser_ref, pinned by tests/golden/denoiser_*.npz).
"""
import numpy as np

from . import conditioning_ref, sampler_ref
from .inputs import COND_CHUNKS

F32 = np.float32


def guidance_batch(cond, uncond):
    """7x replicated memories from per-utterance ``cond`` 5x[B,S_j,512] and shared ``uncond`` 5x[1,S_j,512]."""
    B = cond[0].shape[0]
    out = []
    for j in range(5):
        chunks = [cond[j] if c in COND_CHUNKS[j] else np.repeat(uncond[j], B, axis=0) for c in range(sampler_ref.CFG_CHUNKS)]
        out.append(np.concatenate(chunks, axis=0).astype(F32))
    return out


def dyadic_reverse(denoise_a, denoise_b, scheduler_a, scheduler_b, fuser_sd, cond_a, cond_b, uncond, init_a, init_b,
                   noise_a, noise_b, guidance_scale=7.5, num_inference_steps=20):
    """denoise_x(sample[7B,L,128], t, memories, masks) -> (eps, att).  cond_x[0] is ignored (replaced every step by
    the projection of the partner's latents); noise_x(i, t) -> [B,L,128].  Returns (latents_a, latents_b) [B,L,128]."""
    la = (np.asarray(init_a, F32) * F32(scheduler_a.init_noise_sigma)).astype(F32)
    lb = (np.asarray(init_b, F32) * F32(scheduler_b.init_noise_sigma)).astype(F32)
    scheduler_a.set_timesteps(num_inference_steps)
    scheduler_b.set_timesteps(num_inference_steps)
    masks = {k: None for k in ("spkemb", "alsn", "tlsn", "apb", "lsnemb")}
    for i, t in enumerate(scheduler_a.timesteps):
        spk_a = conditioning_ref.latent_proj(fuser_sd, lb)     # A attends to B's current latents, and vice versa
        spk_b = conditioning_ref.latent_proj(fuser_sd, la)
        new = []
        for lat, den, sch, cond, spk, noise in ((la, denoise_a, scheduler_a, cond_a, spk_a, noise_a),
                                                (lb, denoise_b, scheduler_b, cond_b, spk_b, noise_b)):
            mems = guidance_batch([spk] + list(cond[1:]), uncond)
            eps, _ = den(np.concatenate([lat] * sampler_ref.CFG_CHUNKS, axis=0), int(t), mems, masks)
            eps = sampler_ref.cfg_combine(eps, guidance_scale)
            new.append(sch.step(eps, t, lat, noise=noise(i, t) if t > 0 else None))
        la, lb = new
    return la, lb
