"""CPU oracle for the ConvoFusion denoising loop -- TEST INFRASTRUCTURE ONLY.

This package is a numpy (float32) restatement of the reference's hot path:

  * ``denoiser_ref``  -- ``Denoiser.forward`` and everything below it
                         (reference: convofusion/models/architectures/denoiser.py:173-386,
                          convofusion/models/operator/cross_attention.py:204-247,411-439,556-664,
                          convofusion/models/architectures/tools/embeddings.py:245-322,
                          convofusion/models/operator/position_encoding.py:113-163)
  * ``scheduler_ref`` -- diffusers==0.14.0 DDPMScheduler / DDIMScheduler arithmetic
                         (third-party, un-vendored; pinned in reference environment.yml:85;
                          call sites convofusion/models/modeltype/convofusion.py:104-106,419-423,544)
  * ``sampler_ref``   -- ``Convofusion._diffusion_reverse`` (convofusion.py:391-549), 7-way CFG
  * ``philox_ref``    -- CPU restatement of the product's device RNG (Philox4x32-10 + Box-Muller)
  * ``weights``       -- deterministic state-dict generator with the reference's 537 keys/shapes
  * ``conditioning_ref`` -- AudioConvEncoder / TextAudioMotionFuser (audioenc.py:9-34, condfuser.py:8-50)
  * ``dyadic_ref``    -- two loops in lock-step with partner-projected speaker memories (BASELINE configs[4])
  * ``weg_ref``       -- word-excitation guidance: the attention-focus objective and a hand-written backward pass through
                         the denoiser (word_excitation_guidance.py:11-81, convofusion.py:298-388,437-496), pinned against
                         torch autograd through the imported reference (``make_golden_weg.py`` -> ``weg.npz``)
  * ``vae_ref``, ``vae_weights`` -- ``ConvoFusionVae.decode`` (vae.py:268-372) and its seeded 337-key state dict

Parity pinning: the reference ships no tests or golden vectors.  The
denoiser restatement is pinned against outputs of the reference ``Denoiser`` class itself,
imported from /root/reference in the build container by ``tests/golden/make_golden.py`` and
committed as fixtures under ``tests/golden/``; the conditioning producers and the VAE decoder
likewise (``make_golden_conditioning.py``, ``make_golden_vae.py``).  The scheduler restatement (diffusers is absent
from the reference tree and from this image) is pinned by closed-form known-answer tests only:
**scheduler parity unpinned** against the third-party package itself.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``convofusion_amd``) never imports it and has no CPU fallback.
"""
