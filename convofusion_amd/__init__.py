"""convofusion_amd -- MI355X-native denoising loop for ConvoFusion (HIP/CDNA4 behind a C ABI).

Host-side mirror of the reference interfaces for the one hot path this package accelerates:

  convofusion_amd.denoiser.Denoiser          <- convofusion.models.architectures.denoiser.Denoiser
  convofusion_amd.scheduler.DDPMScheduler    <- diffusers.DDPMScheduler (0.14.0)
  convofusion_amd.scheduler.DDIMScheduler    <- diffusers.DDIMScheduler (0.14.0)
  convofusion_amd.sampler.diffusion_reverse  <- Convofusion._diffusion_reverse
  convofusion_amd.sampler.diffusion_reverse_forecast <- unbounded_synthesis.diffusion_reverse_forecast
  convofusion_amd.install(model) / patch_rollout(module): bind the two loop entry points without editing reference sources

Everything numerical runs in libcfdenoise.so (csrc/, built by convofusion_amd.build); there is no
CPU or PyTorch fallback -- a missing library or a missing MI355X is an error.
"""
from .installer import install, patch_rollout, uninstall  # noqa: E402  (light: the heavy modules load on first use)

__all__ = ["denoiser", "scheduler", "sampler", "distributed", "build", "install", "uninstall", "patch_rollout"]
