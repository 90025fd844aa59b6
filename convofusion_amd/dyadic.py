"""Dyadic reactive sampling (BASELINE.json configs[4]): two denoising loops in lock-step,
each conditioned on its partner.

Before every iteration the conditional ``spkemb`` memory of side A is ``TextAudioMotionFuser.latent_proj``
(reference condfuser.py:22-27: Linear 128->128, GELU, Linear 128->512, GELU) of side B's CURRENT latents
[B, L, 128] -- a memory of L keys -- and vice versa.  This is not a reference feature (the reference declares
``latent_proj`` and never calls it); per denoiser call it is the reference ``Denoiser.forward`` on those memories.

Each side is an ordinary ``SamplingRun`` (one captured hipGraph per handle) over the structured guidance batch
(``build_guidance_batch``: B + 1 distinct memories, no 7x materialisation).  The graph reads the memories
through the pointers given at capture time, so the partner projection writes straight into the conditional rows of
the speaker memory between replays: two small ``cfd_linear_act`` launches per side per step, no re-capture.  The speaker
memory is declared dynamic (``cfd_sample_args.dynamic_memory_mask``), so its projections stay inside the captured iteration; the
other four memories are constants of the run and are projected once.
"""
import torch

from .sampler import SamplingRun, build_guidance_batch


class DyadicRun:
    def __init__(self, denoiser_a, denoiser_b, scheduler, fuser, cond_a, cond_b, uncond, B, L, num_inference_steps, *,
                 cond_masks_a=None, cond_masks_b=None, uncond_masks=None, guidance_scale=7.5, eta=0.0,
                 init_latents_a=None, init_latents_b=None, step_noise_a=None, step_noise_b=None, seed=0,
                 first_utterance=0):
        """cond_x: 5 tensors [B, S_j, 512] (entry 0, the speaker memory, is replaced by the partner projection and may
        be None); uncond: 5 tensors [1, S_j, 512] with uncond[0] of L keys."""
        if denoiser_a is denoiser_b:
            raise ValueError("the two sides need two Denoiser modules (one open sampling run per libcfdenoise handle); "
                             "they may hold the same weights")
        dev = uncond[1].device
        if tuple(uncond[0].shape) != (1, L, 512):
            raise ValueError(f"uncond[0] (speaker memory) must be [1, L={L}, 512]: the partner projection has L keys")
        self.fuser = fuser
        self.B, self.L, self.N = B, L, num_inference_steps
        self.runs, self.spk = [], []
        sides = ((denoiser_a, cond_a, cond_masks_a, init_latents_a, step_noise_a, 0),
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
        self.N = self.runs[0].N      # loop iterations = len(scheduler.timesteps): differs from num_inference_steps for a count that does not divide the schedule
        self.position = 0
        self._lat = None

    def steps(self, n):
        for _ in range(n):
            la, lb = self._lat if self._lat is not None else (self.runs[0].read(), self.runs[1].read())   # (read syncs the run's stream)
            self.fuser.project_latents(lb, out=self.spk[0])          # A attends to B
            self.fuser.project_latents(la, out=self.spk[1])          # B attends to A
            torch.cuda.current_stream(la.device).synchronize()       # the graphs replay on the runs' own streams
            # One side after the other: side A's replay is waited for (its latents are needed for the next iteration anyway) before
            # side B's is launched.  Two captured graphs replaying at the same time are not reliable on this stack (see
            # DESIGN.md sections 6 and 7.2, tools/concurrency_soak.py); the overlap was worth ~5 % of an iteration.
            self.runs[0].steps(1)
            la = self.runs[0].read()
            self.runs[1].steps(1)
            lb = self.runs[1].read()
            self._lat = (la, lb)
            self.position += 1

    def read(self, close=False):
        return self.runs[0].read(close), self.runs[1].read(close)


def sample_dyadic(denoiser_a, denoiser_b, scheduler, fuser, cond_a, cond_b, uncond, *, B, L, num_inference_steps, **kw):
    """Run both loops to the end; returns (latents_a, latents_b), each [B, L, 128]."""
    run = DyadicRun(denoiser_a, denoiser_b, scheduler, fuser, cond_a, cond_b, uncond, B, L, num_inference_steps, **kw)
    run.steps(run.N)
    assert run.position == run.N
    return run.read(close=True)
