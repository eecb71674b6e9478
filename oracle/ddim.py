"""Oracle DDIMScheduler (test infrastructure; see oracle/__init__.py).

Restates diffusers==0.32.2 `DDIMScheduler` as used by the reference:
  * `DDIMScheduler.from_pretrained(..., subfolder="scheduler")`        [REF train:367]
  * `noise_scheduler.config.num_train_timesteps`                       [REF train:503]
  * `noise_scheduler.add_noise(latents, noise, timesteps)`             [REF train:504]
  * `set_timesteps / step / init_noise_sigma / scale_model_input` inside
    `AudioLDMPipeline.__call__`       [REF generate_audio.py:47-52] [REF app.py:14]
Arithmetic spec: SURVEY.md Appendix B.1.  Integer results (timesteps, prev
timestep) are the bit-exact part of the parity contract.
"""
from types import SimpleNamespace

import numpy as np
import torch

from .configs import SCHEDULER


class DDIMScheduler:
    def __init__(self, **overrides):
        cfg = dict(SCHEDULER)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        n = self.config.num_train_timesteps
        assert self.config.beta_schedule == "scaled_linear"
        # diffusers: torch.linspace(beta_start**0.5, beta_end**0.5, n, dtype=float32) ** 2
        self.betas = torch.linspace(self.config.beta_start ** 0.5, self.config.beta_end ** 0.5, n,
                                    dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = (torch.tensor(1.0) if self.config.set_alpha_to_one
                                    else self.alphas_cumprod[0])
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        assert num_inference_steps <= n
        self.num_inference_steps = num_inference_steps
        assert self.config.timestep_spacing == "leading"
        step_ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def prev_timestep(self, timestep):
        return int(timestep) - self.config.num_train_timesteps // self.num_inference_steps

    def step_coefficients(self, timestep):
        """(alpha_prod_t, alpha_prod_t_prev) as python floats from the fp32 table."""
        prev = self.prev_timestep(timestep)
        a_t = self.alphas_cumprod[int(timestep)]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, model_output, timestep, sample, eta=0.0):
        a_t, a_prev = self.step_coefficients(timestep)
        beta_t = 1 - a_t
        assert self.config.prediction_type == "epsilon"
        pred_x0 = (sample - beta_t ** 0.5 * model_output) / a_t ** 0.5
        pred_eps = model_output
        variance = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
        std = eta * variance ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * pred_eps
        prev_sample = a_prev ** 0.5 * pred_x0 + direction
        assert eta == 0.0, "oracle covers the deterministic eta=0 path the reference uses"
        return SimpleNamespace(prev_sample=prev_sample, pred_original_sample=pred_x0)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(dtype=original_samples.dtype)
        sa = ac[timesteps] ** 0.5
        sb = (1 - ac[timesteps]) ** 0.5
        while sa.dim() < original_samples.dim():
            sa = sa.unsqueeze(-1)
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise
