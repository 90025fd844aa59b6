"""Host-side mirror of the diffusers==0.14.0 schedulers the reference instantiates by dotted path
(``target: diffusers.DDPMScheduler``, reference configs/modules/scheduler.yaml:2,14; used at
convofusion/models/modeltype/convofusion.py:104-106,419-423,544,574 and unbounded_synthesis.py:49-75).

diffusers is a third-party dependency that is neither vendored in the reference nor installed here;
these classes restate its public surface for the epsilon-prediction / fixed_small / clip_sample
configuration.  Tables are built with the same torch float32 ops diffusers uses; ``step`` and
``add_noise`` run on the device through libcfdenoise (cfd_scheduler_step / cfd_add_noise).  The
fused sampling loop (convofusion_amd.sampler) reads only the tables and config from these objects.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib

_ops_handles = {}


def _ops_handle(device):
    """A weight-less libcfdenoise handle per device for the stand-alone scheduler kernels."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ops_handles:
        _ops_handles[idx] = _lib.create_handle(idx)
    return _ops_handles[idx]


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


class _Config(dict):
    __getattr__ = dict.__getitem__


class _SchedulerBase:
    KIND = 0

    def _init_tables(self, num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas):
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps

    def _acp_host(self):
        a = self.alphas_cumprod.detach().to("cpu", torch.float32).contiguous()
        return a, C.c_void_p(a.data_ptr())

    def add_noise(self, original_samples, noise, timesteps):
        t = torch.as_tensor(timesteps).reshape(-1)
        if t.numel() != 1 and not bool((t == t[0]).all()):
            # per-sample timesteps (training, convofusion.py:574): broadcast on the host side in slices
            out = torch.empty_like(original_samples)
            for i in range(original_samples.shape[0]):
                out[i] = self.add_noise(original_samples[i:i + 1], noise[i:i + 1], t[i])
            return out
        if not original_samples.is_cuda:
            raise RuntimeError("convofusion_amd schedulers operate on device tensors (no CPU fallback)")
        x = original_samples.detach().to(torch.float32).contiguous()
        n = noise.detach().to(torch.float32).contiguous()
        out = torch.empty_like(x)
        acp, acp_p = self._acp_host()
        lib = _lib.load()
        with torch.cuda.device(x.device):
            _lib.check(lib.cfd_add_noise(_ops_handle(x.device), acp_p, int(t[0]), C.c_void_p(x.data_ptr()), C.c_void_p(n.data_ptr()),
                                         C.c_void_p(out.data_ptr()), x.numel(), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return out

    def _step(self, model_output, timestep, sample, eta, noise, generator):
        if not sample.is_cuda:
            raise RuntimeError("convofusion_amd schedulers operate on device tensors (no CPU fallback)")
        t = int(timestep)
        n_inf = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        eps = model_output.detach().to(torch.float32).contiguous()
        x = sample.detach().to(torch.float32).clone().contiguous()
        needs_noise = (t > 0) if self.KIND == 0 else (eta > 0)
        if needs_noise and noise is None:
            noise = torch.randn(eps.shape, generator=generator, device=eps.device, dtype=eps.dtype)
        acp, acp_p = self._acp_host()
        x0 = torch.empty_like(x)     # the x0 estimate the step forms on the way (diffusers: SchedulerOutput.pred_original_sample)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            _lib.check(lib.cfd_scheduler_step(
                _ops_handle(x.device), self.KIND, acp_p, self.config.num_train_timesteps, n_inf, t,
                1 if self.config.clip_sample else 0, float(eta), 1 if self.config.get("set_alpha_to_one", True) else 0,
                C.c_void_p(eps.data_ptr()), C.c_void_p(noise.data_ptr()) if noise is not None else None,
                C.c_void_p(x.data_ptr()), x.numel(), C.c_void_p(x0.data_ptr()), C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)))
        return x, x0


class DDPMScheduler(_SchedulerBase):
    """diffusers 0.14.0 DDPMScheduler (epsilon prediction, variance_type fixed_small)."""
    KIND = 0

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 allow_unpinned_timesteps=False, **kwargs):
        if variance_type != "fixed_small":
            raise NotImplementedError("only variance_type='fixed_small' (configs/modules/scheduler.yaml:10)")
        if prediction_type != "epsilon":
            raise NotImplementedError("only prediction_type='epsilon' (TRAIN.ABLATION.PREDICT_EPSILON)")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, variance_type=variance_type, clip_sample=clip_sample,
                              prediction_type=prediction_type, allow_unpinned_timesteps=bool(allow_unpinned_timesteps))
        self.variance_type = variance_type
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)

    def timestep_table(self, num_inference_steps):
        """(clamped step count, int64 timestep array) of ``set_timesteps`` without touching the scheduler's state.

        A count that divides ``num_train_timesteps`` (the shipped schedule is 1000 of 1000, configs/modules/scheduler.yaml) gives
        ``(arange(N) * (T // N))[::-1]``, on which every diffusers release agrees.  For other counts the releases disagree:
        0.14.0 (the reference's pin, environment.yml:85) builds ``arange(0, T, T // N)[::-1]`` -- which has MORE than N entries,
        e.g. 334 for N = 300 -- later ones ``(arange(N) * (T // N)).round()[::-1]``.  The package is not available here to pin
        either, so such counts are refused unless the scheduler was built with ``allow_unpinned_timesteps=True``, which selects
        the 0.14.0 form (restated from the release's published source, NOT checked against it: parity unpinned)."""
        T = self.config.num_train_timesteps
        n = min(T, int(num_inference_steps))
        if n < 1:
            raise ValueError(f"num_inference_steps = {num_inference_steps}")
        if T % n == 0:
            return n, (np.arange(0, n) * (T // n)).round()[::-1].copy().astype(np.int64)
        if not self.config.get("allow_unpinned_timesteps", False):
            raise ValueError(f"DDPM num_inference_steps = {n} does not divide num_train_timesteps = {T}: the timestep table for such "
                             "counts differs between diffusers releases; build the scheduler with allow_unpinned_timesteps=True to "
                             "get the 0.14.0 table arange(0, T, T // N)[::-1] (unpinned)")
        return n, np.arange(0, T, T // n)[::-1].copy().astype(np.int64)

    def set_timesteps(self, num_inference_steps, device=None):
        """``num_inference_steps`` is clamped to the training schedule like diffusers 0.14.0 does; see ``timestep_table``."""
        self.num_inference_steps, timesteps = self.timestep_table(num_inference_steps)
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, variance_noise=None):
        prev, x0 = self._step(model_output, timestep, sample, 0.0, variance_noise, generator)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0) if return_dict else (prev,)


class DDIMScheduler(_SchedulerBase):
    """diffusers 0.14.0 DDIMScheduler (epsilon prediction)."""
    KIND = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", **kwargs):
        if prediction_type != "epsilon":
            raise NotImplementedError("only prediction_type='epsilon'")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, clip_sample=clip_sample, set_alpha_to_one=set_alpha_to_one,
                              steps_offset=steps_offset, prediction_type=prediction_type)
        self._init_tables(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def timestep_table(self, num_inference_steps):
        """(step count, int64 timestep array) of ``set_timesteps`` without touching the scheduler's state."""
        n = int(num_inference_steps)
        step_ratio = self.config.num_train_timesteps // n
        return n, (np.arange(0, n) * step_ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps, timesteps = self.timestep_table(num_inference_steps)
        self.timesteps = torch.from_numpy(timesteps).to(device)

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True):
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output")
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        prev, x0 = self._step(model_output, timestep, sample, eta, variance_noise, generator)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0) if return_dict else (prev,)
