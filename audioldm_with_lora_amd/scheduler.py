"""DDIMScheduler for the HIP path (host-side integer logic + device step kernel).

Same surface the reference touches: `from_pretrained(id, subfolder="scheduler")`, `.config.num_train_timesteps`,
`.add_noise` [REF script/train/train_audioldm_lora.py:367,503-504] and, through the pipeline,
`set_timesteps / timesteps / step / init_noise_sigma / scale_model_input` [REF script/inference/generate_audio.py:47-52].
Arithmetic spec: SURVEY.md Appendix B.1 (diffusers 0.32.2).  Timestep indices are int64 and computed exactly as
diffusers does ("leading" spacing, steps_offset); the fp32 alpha-bar table uses the same torch ops as diffusers.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .configs import SCHEDULER


class DDIMScheduler:
    def __init__(self, **over):
        cfg = dict(SCHEDULER)
        cfg.update({k: v for k, v in over.items() if k in SCHEDULER})
        self.config = SimpleNamespace(**cfg)
        if cfg["beta_schedule"] != "scaled_linear" or cfg["prediction_type"] != "epsilon" or cfg["timestep_spacing"] != "leading":
            raise NotImplementedError("only the AudioLDM scheduler configuration is implemented")
        n = cfg["num_train_timesteps"]
        self.betas = torch.linspace(cfg["beta_start"] ** 0.5, cfg["beta_end"] ** 0.5, n, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if cfg["set_alpha_to_one"] else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, n)[::-1].copy().astype(np.int64))
        self._dev = {}

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        f = os.path.join(d, "scheduler_config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f} not found: hub downloads are unavailable, pass a local directory")
        return cls(**json.load(open(f)))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps, device=None):
        n = self.config.num_train_timesteps
        if num_inference_steps > n:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = n // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)
        if device is not None:
            self.timesteps = self.timesteps.to(device)

    def prev_timestep(self, t):
        return int(t) - self.config.num_train_timesteps // self.num_inference_steps

    def step_coefficients(self, t):
        """fp32 {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} for timestep t (eta = 0)."""
        p = self.prev_timestep(t)
        a_t = self.alphas_cumprod[int(t)]
        a_p = self.alphas_cumprod[p] if p >= 0 else self.final_alpha_cumprod
        return torch.stack([a_t ** 0.5, (1 - a_t) ** 0.5, a_p ** 0.5, (1 - a_p) ** 0.5]).float()

    def coefficient_table(self):
        return torch.stack([self.step_coefficients(t) for t in self.timesteps.tolist()])

    def step(self, model_output, timestep, sample, eta=0.0, **kw):
        """x_{t-1} from eps (eta = 0) via the fused device kernel; fp32 tensors of any layout."""
        if eta != 0.0:
            raise NotImplementedError("the reference path uses eta = 0")
        dev = sample.device
        coef = self.step_coefficients(timestep).to(dev)
        idx = self._dev.setdefault(("zero", dev), torch.zeros(1, dtype=torch.int32, device=dev))
        x = sample.detach().float().contiguous().clone()
        ops.cfg_ddim_step(model_output.detach().float().contiguous(), x, False, 0.0, coef, idx, None)
        return SimpleNamespace(prev_sample=x.to(sample.dtype))

    def add_noise(self, original_samples, noise, timesteps):
        dev = original_samples.device
        ac = self._dev.get(("ac", dev))
        if ac is None:
            ac = self._dev[("ac", dev)] = self.alphas_cumprod.to(dev, torch.float32).contiguous()
        return ops.add_noise_t(original_samples, noise, ac, timesteps.to(dev, torch.int64).reshape(-1))
