"""Oracle AudioLDMPipeline.__call__ and the LoRA train step (test infrastructure).

Restates diffusers==0.32.2 `AudioLDMPipeline.__call__` steps 1-8 of SURVEY.md
section 3.1 as the reference invokes it
  [REF app.py:14] [REF script/inference/generate_audio.py:47-52]
  [REF script/train/train_audioldm_lora.py:142,161]
and the train-loop body [REF script/train/train_audioldm_lora.py:495-565]
(VAE encode and the CLAP text encoder are outside the north_star path: the
step starts from latents and L2-normalised prompt embeddings).
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F


def audio_geometry(audio_length_in_s, vocoder_cfg, vae_scale_factor=4):
    """height (mel frames) and output sample count: pipeline steps 1 / 8."""
    up = float(np.prod(vocoder_cfg.upsample_rates)) / vocoder_cfg.sampling_rate
    height = int(audio_length_in_s / up)
    original_waveform_length = int(audio_length_in_s * vocoder_cfg.sampling_rate)
    if height % vae_scale_factor != 0:
        height = int(np.ceil(height / vae_scale_factor)) * vae_scale_factor
    return height, original_waveform_length


def cfg_combine(eps, g):
    u, t = eps.chunk(2)
    return u + g * (t - u)


def denoise_loop(unet, scheduler, latents, prompt_embeds, negative_prompt_embeds, num_inference_steps,
                 guidance_scale, trace=None):
    """The hot loop (section 3.1 step 5).  `trace`, if a list, receives per-step latents."""
    cfg = guidance_scale > 1.0
    emb = torch.cat([negative_prompt_embeds, prompt_embeds]) if cfg else prompt_embeds
    scheduler.set_timesteps(num_inference_steps)
    latents = latents * scheduler.init_noise_sigma
    for t in scheduler.timesteps:
        x_in = torch.cat([latents] * 2) if cfg else latents
        x_in = scheduler.scale_model_input(x_in, t)
        eps = unet(x_in, t, encoder_hidden_states=None, class_labels=emb)[0]
        if cfg:
            eps = cfg_combine(eps, guidance_scale)
        latents = scheduler.step(eps, t, latents, eta=0.0).prev_sample
        if trace is not None:
            trace.append(latents.clone())
    return latents


class AudioLDMPipeline:
    """Embedding-level oracle pipeline: prompt_embeds in, audio out."""

    def __init__(self, unet, vae, vocoder, scheduler):
        self.unet, self.vae, self.vocoder, self.scheduler = unet, vae, vocoder, scheduler
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds=None, audio_length_in_s=5.12,
                 num_inference_steps=10, guidance_scale=2.5, latents=None, generator=None):
        vc = self.vocoder.config
        height, n_samples = audio_geometry(audio_length_in_s, vc, self.vae_scale_factor)
        b = prompt_embeds.shape[0]
        shape = (b, self.unet.cfg["in_channels"], height // self.vae_scale_factor,
                 vc.model_in_dim // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator)
        assert tuple(latents.shape) == shape
        if negative_prompt_embeds is None:
            negative_prompt_embeds = torch.zeros_like(prompt_embeds)
        latents = denoise_loop(self.unet, self.scheduler, latents, prompt_embeds, negative_prompt_embeds,
                               num_inference_steps, guidance_scale)
        mel = self.vae.decode(latents / self.vae.config.scaling_factor).sample
        wav = self.vocoder(mel.squeeze(1)).float()
        return SimpleNamespace(audios=wav[:, :n_samples].numpy(), mel=mel, latents=latents)


def polynomial_lr(step, lr_init, num_training_steps, lr_end=1e-7, power=1.0, num_warmup_steps=0):
    """diffusers get_scheduler("polynomial") lambda [REF train:438-443]; same rule as
    transformers.optimization.get_polynomial_decay_schedule_with_warmup."""
    if step < num_warmup_steps:
        return lr_init * step / max(1, num_warmup_steps)
    if step > num_training_steps:
        return lr_end
    lr_range = lr_init - lr_end
    remaining = 1 - (step - num_warmup_steps) / (num_training_steps - num_warmup_steps)
    return lr_range * remaining ** power + lr_end


def train_step(unet, scheduler, optimizer, latents, noise, timesteps, prompt_embeds):
    """One LoRA step: add_noise -> UNet -> eps-MSE -> backward -> AdamW  [REF train:499-565]."""
    noisy = scheduler.add_noise(latents, noise, timesteps)
    pred = unet(noisy, timesteps, encoder_hidden_states=None, class_labels=prompt_embeds,
                cross_attention_kwargs={"scale": 1.0}, return_dict=False)[0]
    loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach()
