"""AudioLDMPipeline on the MI355X HIP path.

Drop-in for diffusers' `AudioLDMPipeline` as the reference drives it:
  `AudioLDMPipeline.from_pretrained(id, unet=unet, torch_dtype=...)`, `.to(device)`,
  `pipe(prompt, num_inference_steps=, audio_length_in_s=, guidance_scale=).audios[0]`
  [REF app.py:7-14] [REF script/inference/generate_audio.py:42-52] [REF script/train/train_audioldm_lora.py:365,599,142]
Steps 1-8 of SURVEY.md section 3.1.  The text encoder (CLAP, outside the north_star path) is stock transformers
host code when a local checkpoint provides it; `prompt_embeds=` bypasses it.
"""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .engine import DenoiseEngine
from .scheduler import DDIMScheduler
from .unet import UNet2DConditionModel
from .vae import AutoencoderKL
from .vocoder import SpeechT5HifiGan


class AudioPipelineOutput(SimpleNamespace):
    pass


class AudioLDMPipeline:
    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler, vocoder):
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        # the reference hands the pipeline either the bare UNet [REF generate_audio.py:42] or the peft wrapper
        # (accelerator.unwrap_model(unet) is the PeftModel [REF train:598-599]); both keep `.unet` as given
        self.unet, self.scheduler, self.vocoder = unet, scheduler, vocoder
        self.vae_scale_factor = 2 ** (len(vae.cfg["block_out_channels"]) - 1)
        self.device = torch.device("cpu")
        self._engines = {}
        self._progress = {}

    @classmethod
    def from_pretrained(cls, path, unet=None, torch_dtype=None, **kw):
        """Loads a local diffusers-format directory (unet/ vae/ vocoder/ scheduler/ [text_encoder/ tokenizer/])."""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: hub downloads are unavailable offline, pass a local model directory")
        if unet is None:
            unet = UNet2DConditionModel.from_pretrained(path, subfolder="unet")
        vae = AutoencoderKL.from_pretrained(path, subfolder="vae")
        vocoder = SpeechT5HifiGan.from_pretrained(path, subfolder="vocoder")
        scheduler = DDIMScheduler.from_pretrained(path, subfolder="scheduler")
        text_encoder = tokenizer = None
        if os.path.isdir(os.path.join(path, "text_encoder")):
            from transformers import RobertaTokenizerFast          # host-side string -> ids only
            from .clap_text import ClapTextModelWithProjection    # the tower itself runs on the HIP kernels
            text_encoder = ClapTextModelWithProjection.from_pretrained(os.path.join(path, "text_encoder"))
            tokenizer = RobertaTokenizerFast.from_pretrained(os.path.join(path, "tokenizer"))
        return cls(vae, text_encoder, tokenizer, unet, scheduler, vocoder)

    @property
    def _unet(self):
        """the UNet2DConditionModel behind `.unet` (unwraps a lora.PeftModel)"""
        return getattr(getattr(self.unet, "base_model", None), "model", self.unet)

    def to(self, device):
        self.device = torch.device(device)
        for m in (self.vae, self.unet, self.vocoder, self.text_encoder):
            if m is not None:
                m.to(self.device)
        self._engines.clear()
        return self

    def set_progress_bar_config(self, **kw):
        self._progress = kw

    # ---- step 2: prompt -> L2-normalised CLAP text embedding (tokeniser on the host, tower on the GPU: clap_text.py) ----
    def _encode_prompt(self, prompt, batch):
        if self.text_encoder is None or self.tokenizer is None:
            raise ValueError("no text encoder loaded: pass prompt_embeds= / negative_prompt_embeds=")
        tok = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                             return_tensors="pt")
        with torch.no_grad():
            emb = self.text_encoder(tok.input_ids.to(self.device), attention_mask=tok.attention_mask.to(self.device)).text_embeds
        return torch.nn.functional.normalize(emb.float(), dim=-1)

    def geometry(self, audio_length_in_s):
        vc = self.vocoder.config
        up = float(np.prod(vc.upsample_rates)) / vc.sampling_rate
        height = int(audio_length_in_s / up)
        n_samples = int(audio_length_in_s * vc.sampling_rate)
        if height % self.vae_scale_factor != 0:
            height = int(np.ceil(height / self.vae_scale_factor)) * self.vae_scale_factor
        return height, n_samples

    def engine(self, batch, h, w, steps, guidance):
        key = (batch, h, w, steps, float(guidance))
        eng = self._engines.get(key)
        if eng is not None and (eng.unet is not self._unet or eng.stale()):
            eng = None                                  # weights / adapter changed since the capture: never replay the old graph
        if eng is None:
            eng = self._engines[key] = DenoiseEngine(self._unet, self.scheduler, batch, h, w, steps, guidance, device=self.device)
        return eng

    def decode_latents_nhwc(self, x_nhwc_f32):
        """steps 6-7: latents [B, h, w, 8] fp32 channels-last -> waveform [B, 160*4h + 32] fp32 (device)."""
        z = ops.f32_to_bf16(x_nhwc_f32, 1.0 / self.vae.config.scaling_factor)
        mel = self.vae.decode_nhwc(z)                          # [B, T, 64, 1] fp32 == [B, 1, T, 64] channels-last
        B, T, F, _ = mel.shape
        melb = ops.f32_to_bf16(mel).view(B, 1, T, F)
        return self.vocoder.forward_nhwc(melb), mel

    @torch.no_grad()
    def __call__(self, prompt=None, audio_length_in_s=None, num_inference_steps=10, guidance_scale=2.5,
                 negative_prompt=None, num_waveforms_per_prompt=1, eta=0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, return_dict=True, output_type="np", **kw):
        if self.device.type != "cuda":
            raise ops._lib.AldmError("AudioLDMPipeline runs on the MI355X only: call .to('cuda') (no CPU fallback)")
        if eta != 0.0:
            raise NotImplementedError("the reference path uses eta = 0")
        vc = self.vocoder.config
        if audio_length_in_s is None:
            # diffusers: unet.config.sample_size * vae_scale_factor * prod(upsample_rates) / sampling_rate
            inner = getattr(getattr(self.unet, "base_model", None), "model", self.unet)          # a PeftModel wraps the UNet
            audio_length_in_s = inner.config.sample_size * self.vae_scale_factor * float(np.prod(vc.upsample_rates)) / vc.sampling_rate
        height, n_samples = self.geometry(audio_length_in_s)
        if prompt_embeds is None:
            if prompt is None:
                raise ValueError("pass prompt or prompt_embeds")
            prompts = [prompt] if isinstance(prompt, str) else list(prompt)
            prompt_embeds = self._encode_prompt(prompts, len(prompts))
        batch = prompt_embeds.shape[0]
        cfg = guidance_scale > 1.0
        if cfg and negative_prompt_embeds is None:
            if self.text_encoder is not None:
                neg = [""] * batch if negative_prompt is None else ([negative_prompt] * batch if isinstance(negative_prompt, str) else list(negative_prompt))
                negative_prompt_embeds = self._encode_prompt(neg, batch)
            else:
                negative_prompt_embeds = torch.zeros_like(prompt_embeds)
        if num_waveforms_per_prompt > 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_waveforms_per_prompt, dim=0)
            if negative_prompt_embeds is not None:
                negative_prompt_embeds = negative_prompt_embeds.repeat_interleave(num_waveforms_per_prompt, dim=0)
            batch = prompt_embeds.shape[0]
        h, w = height // self.vae_scale_factor, vc.model_in_dim // self.vae_scale_factor
        shape = (batch, self._unet.cfg["in_channels"], h, w)
        if latents is None:
            gdev = generator.device if generator is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"Unexpected latents shape, got {tuple(latents.shape)}, expected {shape}")
        latents = latents.to(self.device, torch.float32) * self.scheduler.init_noise_sigma

        eng = self.engine(batch, h, w, num_inference_steps, guidance_scale)
        eng.set_condition(prompt_embeds, negative_prompt_embeds)
        eng.set_latents(latents)
        if eng.graph is None and eng.use_graph:
            eng.capture()
        eng.run()
        wav, mel = self.decode_latents_nhwc(eng.x)
        audio = wav[:, :n_samples]
        if output_type == "np":
            audio = audio.float().cpu().numpy()
        if not return_dict:
            return (audio,)
        return AudioPipelineOutput(audios=audio)
