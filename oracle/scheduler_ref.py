"""numpy float32 restatement of diffusers==0.14.0 DDPMScheduler / DDIMScheduler arithmetic.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The scheduler is a third-party dependency of the
reference: ``diffusers.DDPMScheduler`` (configs/modules/scheduler.yaml:2,14), pinned
``diffusers==0.14.0`` in environment.yml:85; its source is NOT under /root/reference and the
package is not installed in this image, so this file restates its published algorithm
(epsilon prediction, ``variance_type='fixed_small'``, ``clip_sample=True``).  Call sites in the
reference: convofusion/models/modeltype/convofusion.py:104-106 (instantiate), :419
(``init_noise_sigma``), :421-423 (``set_timesteps`` / ``timesteps``), :544 (``step``), :574
(``add_noise``); unbounded_synthesis.py:49,56-58,75,181.

PARITY UNPINNED against the third-party package: the reference holds no test or golden vector at
this boundary.  Pinned instead by closed-form known-answer tests (tests/test_oracle_scheduler.py)
and against tables generated with torch in the build container (tests/golden/scheduler_tables.npz).

Scalar coefficient math is float32, as in diffusers (0-dim float32 tensors indexed out of
``alphas_cumprod``).  diffusers 0.14.0 uses the *ratio* form ``alpha_t = abar_t / abar_prev``
(version caveat).
"""
import numpy as np

F32 = np.float32


def _linspace_f32(start, end, steps):
    """torch.linspace(float32) on CPU: symmetric fill from both ends, ``start + step*i`` fused
    (one rounding) -- emulated by evaluating in float64 from the float32 scalars."""
    start, end = F32(start), F32(end)
    step = F32((np.float64(end) - np.float64(start)) / (steps - 1))
    i = np.arange(steps)
    half = steps // 2
    lo = (np.float64(start) + np.float64(step) * i).astype(F32)
    hi = (np.float64(end) - np.float64(step) * (steps - 1 - i)).astype(F32)
    return np.where(i < half, lo, hi).astype(F32)


class _Tables:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear"):
        self.num_train_timesteps = num_train_timesteps
        if beta_schedule == "scaled_linear":
            r = _linspace_f32(np.float64(beta_start) ** 0.5, np.float64(beta_end) ** 0.5, num_train_timesteps)
            self.betas = (r * r).astype(F32)
        elif beta_schedule == "linear":
            self.betas = _linspace_f32(beta_start, beta_end, num_train_timesteps)
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas = (F32(1.0) - self.betas).astype(F32)
        self.alphas_cumprod = np.cumprod(self.alphas, dtype=F32)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64)

    def add_noise(self, original, noise, timesteps):
        ac = self.alphas_cumprod[np.asarray(timesteps).reshape(-1)]
        sa = np.sqrt(ac).astype(F32)
        sb = np.sqrt(F32(1.0) - ac).astype(F32)
        shape = (-1,) + (1,) * (np.ndim(original) - 1)
        return (sa.reshape(shape) * original + sb.reshape(shape) * noise).astype(F32)


class DDPMSchedulerRef(_Tables):
    def __init__(self, clip_sample=True, variance_type="fixed_small", allow_unpinned_timesteps=False, **kw):
        super().__init__(**kw)
        assert variance_type == "fixed_small"
        self.clip_sample = clip_sample
        self.allow_unpinned_timesteps = allow_unpinned_timesteps

    def set_timesteps(self, num_inference_steps):
        T = self.num_train_timesteps
        num_inference_steps = min(T, num_inference_steps)
        if num_inference_steps < 1:
            raise ValueError("num_inference_steps")
        self.num_inference_steps = num_inference_steps
        ratio = T // num_inference_steps
        if T % num_inference_steps == 0:
            self.timesteps = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        elif self.allow_unpinned_timesteps:
            # diffusers 0.14.0 scheduling_ddpm.set_timesteps as published: arange(0, T, T // N)[::-1] -- MORE than N entries
            # when N does not divide T; later releases build (arange(N) * (T // N))[::-1].  Restated, not checked against the
            # package (absent here): parity unpinned for such counts.
            self.timesteps = np.arange(0, T, ratio)[::-1].copy().astype(np.int64)
        else:
            raise ValueError("DDPM num_inference_steps must divide num_train_timesteps (or allow_unpinned_timesteps=True)")

    def coefficients(self, t):
        """(sqrt_beta_prod_t, sqrt_alpha_prod_t, x0_coeff, x_coeff, sigma) as float32 scalars."""
        n = self.num_inference_steps or self.num_train_timesteps
        prev_t = int(t) - self.num_train_timesteps // n
        ap_t = self.alphas_cumprod[int(t)]
        ap_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else F32(1.0)
        bp_t = F32(1.0) - ap_t
        bp_prev = F32(1.0) - ap_prev
        cur_alpha = F32(ap_t / ap_prev)
        cur_beta = F32(1.0) - cur_alpha
        c0 = F32(np.sqrt(ap_prev) * cur_beta / bp_t)
        cx = F32(np.sqrt(cur_alpha) * bp_prev / bp_t)
        var = F32(bp_prev / bp_t * cur_beta)
        var = max(var, F32(1e-20))
        sigma = F32(np.sqrt(var)) if int(t) > 0 else F32(0.0)
        return F32(np.sqrt(bp_t)), F32(np.sqrt(ap_t)), c0, cx, sigma

    def step(self, model_output, t, sample, noise=None):
        """prev_sample; ``noise`` is the N(0,1) draw the reference takes from the global
        generator when t > 0 (injected here so trajectories are comparable)."""
        sb, sa, c0, cx, sigma = self.coefficients(t)
        x0 = ((sample - sb * model_output) / sa).astype(F32)
        if self.clip_sample:
            x0 = np.clip(x0, F32(-1.0), F32(1.0))
        prev = (c0 * x0 + cx * sample).astype(F32)
        if int(t) > 0:
            prev = (prev + sigma * noise).astype(F32)
        self.pred_original_sample = x0      # diffusers: SchedulerOutput.pred_original_sample (read by convofusion.py:619)
        return prev


class DDIMSchedulerRef(_Tables):
    def __init__(self, clip_sample=True, set_alpha_to_one=True, steps_offset=0, **kw):
        super().__init__(**kw)
        self.clip_sample = clip_sample
        self.final_alpha_cumprod = F32(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.steps_offset = steps_offset

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = ts + self.steps_offset

    def coefficients(self, t, eta=0.0):
        """(sqrt_beta_prod_t, sqrt_alpha_prod_t, sqrt_alpha_prev, dir_coeff, sigma)."""
        prev_t = int(t) - self.num_train_timesteps // self.num_inference_steps
        ap_t = self.alphas_cumprod[int(t)]
        ap_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        bp_t = F32(1.0) - ap_t
        bp_prev = F32(1.0) - ap_prev
        var = F32((bp_prev / bp_t) * (F32(1.0) - ap_t / ap_prev))
        std = F32(F32(eta) * np.sqrt(var))
        dirc = F32(np.sqrt(F32(1.0) - ap_prev - std * std))
        return F32(np.sqrt(bp_t)), F32(np.sqrt(ap_t)), F32(np.sqrt(ap_prev)), dirc, std

    def step(self, model_output, t, sample, eta=0.0, noise=None):
        sb, sa, sp, dirc, std = self.coefficients(t, eta)
        x0 = ((sample - sb * model_output) / sa).astype(F32)
        if self.clip_sample:
            x0 = np.clip(x0, F32(-1.0), F32(1.0))
        prev = (sp * x0 + dirc * model_output).astype(F32)
        if eta > 0:
            prev = (prev + std * noise).astype(F32)
        self.pred_original_sample = x0
        return prev
