"""Dyadic reactive sampling (BASELINE.json configs[4]): two denoising loops in lock-step,
each conditioned on its partner.

Before every iteration the conditional ``spkemb`` memory of side A is ``TextAudioMotionFuser.latent_proj``
(reference condfuser.py:22-27: Linear 128->128, GELU, Linear 128->512, GELU) of side B's CURRENT latents
[B, L, 128] -- a memory of L keys -- and vice versa.  This is not a reference feature (the reference declares
``latent_proj`` and never calls it); per denoiser call it is the reference ``Denoiser.forward`` on those memories.

Each side is an ordinary ``SamplingRun`` (one captured hipGraph per handle) over the structured guidance batch
(``build_guidance_batch``: B + 1 distinct memories, no 7x materialisation).  The graph reads the memories
through the pointers given at capture time, so the partner projection writes straight into the conditional rows of
the speaker memory between replays: two small ``linear_act`` launches per side per step, no re-capture; the library enqueues a
lock-step iteration (four projection launches, two graph replays) on one stream without any host synchronisation
(``cfd_dyadic_steps``).  The speaker
memory is declared dynamic (``cfd_sample_args.dynamic_memory_mask``), so its projections stay inside the captured iteration; the
other four memories are constants of the run and are projected once.
"""
import ctypes as C

import torch

from . import _lib
from .sampler import SamplingRun, build_guidance_batch


def _cat_masks(ma, mb, B, S_of, dev):
    """Key-padding masks of the two sides stacked along the batch (a side without a mask for a memory contributes zeros)."""
    if ma is None and mb is None:
        return None
    out = {}
    for k in set(ma or {}) | set(mb or {}):
        a, b = (ma or {}).get(k), (mb or {}).get(k)
        if a is None and b is None:
            out[k] = None
            continue
        z = lambda: torch.zeros((B, S_of[k]), dtype=torch.bool, device=dev)   # noqa: E731
        out[k] = torch.cat([a if a is not None else z(), b if b is not None else z()], 0)
    return out


class DyadicRun:
    def __init__(self, denoiser_a, denoiser_b, scheduler, fuser, cond_a, cond_b, uncond, B, L, num_inference_steps, *,
                 cond_masks_a=None, cond_masks_b=None, uncond_masks=None, guidance_scale=7.5, eta=0.0,
                 init_latents_a=None, init_latents_b=None, step_noise_a=None, step_noise_b=None, seed=0,
                 first_utterance=0, shared_weights=False):
        """cond_x: 5 tensors [B, S_j, 512] (entry 0, the speaker memory, is replaced by the partner projection and may
        be None); uncond: 5 tensors [1, S_j, 512] with uncond[0] of L keys.

        ``shared_weights=True`` (``denoiser_b`` None or ``denoiser_a``): both sides use ONE denoiser, so the pair is one sampling run
        of 2 B utterances -- side A's followed by side B's -- whose speaker rows point at the partner's projection: one captured
        iteration of the double batch per lock-step iteration instead of two of B (13.9 instead of 18 ms at B = 16 per side, L = 196).
        Without injected noise the sides then draw from the Philox streams of utterances ``first_utterance`` .. ``+ 2 B - 1`` of ONE seed
        (the two-handle form gives side B the seed ``seed + 1``)."""
        self.merged = bool(shared_weights)
        if self.merged:
            if denoiser_b is not None and denoiser_b is not denoiser_a:
                raise ValueError("shared_weights=True runs both sides on denoiser_a: pass denoiser_b=None (or the same module)")
        elif denoiser_a is denoiser_b or denoiser_b is None:
            raise ValueError("the two sides need two Denoiser modules (one open sampling run per libcfdenoise handle); "
                             "for two sides with the same weights pass shared_weights=True")
        dev = uncond[1].device
        if tuple(uncond[0].shape) != (1, L, 512):
            raise ValueError(f"uncond[0] (speaker memory) must be [1, L={L}, 512]: the partner projection has L keys")
        self.fuser = fuser
        self.B, self.L, self.N = B, L, num_inference_steps
        self.runs, self.spk = [], []
        if self.merged:
            cond = [None] + [torch.cat([cond_a[j], cond_b[j]], 0) for j in range(1, 5)]
            cond[0] = torch.zeros((2 * B, L, 512), dtype=torch.float32, device=dev)
            names = ("spkemb", "alsn", "tlsn", "apb", "lsnemb")
            cmask = _cat_masks(cond_masks_a, cond_masks_b, B, {n: int(cond[j].shape[1]) for j, n in enumerate(names)}, dev)
            both = lambda x, y: None if x is None and y is None else torch.cat([x, y], 0 if x.dim() == 3 else 1)   # noqa: E731
            if (init_latents_a is None) != (init_latents_b is None) or (step_noise_a is None) != (step_noise_b is None):
                raise ValueError("give the initial latents / the step noise of both sides or of neither")
            uniq, maps, masks = build_guidance_batch(cond, uncond, cmask, uncond_masks)
            run = SamplingRun(denoiser_a, scheduler, uniq, masks, 2 * B, L, num_inference_steps, guidance_scale=guidance_scale, eta=eta,
                              init_latents=both(init_latents_a, init_latents_b), step_noise=both(step_noise_a, step_noise_b), seed=seed,
                              first_utterance=first_utterance, dedup=False, row_maps=maps, dynamic_memories=(0,))
            self.runs.append(run)
            self.spk = [uniq[0][1:1 + B], uniq[0][1 + B:]]
        sides = () if self.merged else ((denoiser_a, cond_a, cond_masks_a, init_latents_a, step_noise_a, 0),
                                        (denoiser_b, cond_b, cond_masks_b, init_latents_b, step_noise_b, 1))
        for den, cond, cmask, init, noise, side in sides:
            cond = list(cond)
            cond[0] = torch.zeros((B, L, 512), dtype=torch.float32, device=dev)
            uniq, maps, masks = build_guidance_batch(cond, uncond, cmask, uncond_masks)
            run = SamplingRun(den, scheduler, uniq, masks, B, L, num_inference_steps, guidance_scale=guidance_scale, eta=eta,
                              init_latents=init, step_noise=noise, seed=seed + side, first_utterance=first_utterance,
                              dedup=False, row_maps=maps, dynamic_memories=(0,))
            self.runs.append(run)
            self.spk.append(uniq[0][1:])     # rows 1..B of the distinct speaker memories = the conditional ones (a view)
        lp = fuser.latent_proj
        if lp[0].in_features != 128 or lp[2].out_features != 512:
            raise ValueError("the partner projection must map 128-wide latents to 512-wide keys (condfuser.py:22-27)")
        self._w = [t.detach().to(dev, torch.float32).contiguous() for t in (lp[0].weight, lp[0].bias, lp[2].weight, lp[2].bias)]
        self._tmp = torch.empty((B * L, lp[0].out_features), dtype=torch.float32, device=dev)
        self._proj = _lib.DyadicProj(w1=self._w[0].data_ptr(), b1=self._w[1].data_ptr(), w2=self._w[2].data_ptr(), b2=self._w[3].data_ptr(),
                                     hidden=lp[0].out_features, out_dim=512, spk_a=self.spk[0].data_ptr(), spk_b=self.spk[1].data_ptr(),
                                     tmp=self._tmp.data_ptr())
        self.N = self.runs[0].N      # loop iterations = len(scheduler.timesteps): differs from num_inference_steps for a count that does not divide the schedule
        self.position = 0

    def steps(self, n):
        """n lock-step iterations, enqueued by the library on side A's stream with no host synchronisation in between
        (``cfd_dyadic_steps``): per iteration the two partner projections (A's speaker memory from B's current latents and vice
        versa, both taken at the start of the iteration), then side A's captured iteration, then side B's -- one queue, one after
        the other (two graphs replaying at once are not reliable on this stack: DESIGN.md sections 6 and 7.2)."""
        n = int(n)
        if n <= 0:
            return
        _lib.check(_lib.load().cfd_dyadic_steps(self.runs[0].handle, None if self.merged else self.runs[1].handle, C.byref(self._proj), n))
        _lib.wrote(*self.spk)     # (the partner projections landed in these speaker-memory rows)
        self.position += n

    def read(self, close=False):
        if self.merged:
            lat = self.runs[0].read(close)
            return lat[:self.B], lat[self.B:]
        return self.runs[0].read(close), self.runs[1].read(close)


def sample_dyadic(denoiser_a, denoiser_b, scheduler, fuser, cond_a, cond_b, uncond, *, B, L, num_inference_steps, **kw):
    """Run both loops to the end; returns (latents_a, latents_b), each [B, L, 128]."""
    run = DyadicRun(denoiser_a, denoiser_b, scheduler, fuser, cond_a, cond_b, uncond, B, L, num_inference_steps, **kw)
    run.steps(run.N)
    assert run.position == run.N
    return run.read(close=True)
