"""MI355X-native AudioLDM + LoRA hot path (gfx950 HIP kernels behind a C-ABI).

Drop-in surface for the reference's entry points:
AudioLDMPipeline / UNet2DConditionModel / AutoencoderKL / SpeechT5HifiGan / DDIMScheduler and the
peft-shaped LoraConfig / get_peft_model helpers.  No CPU fallback: ops raise if libaldm_hip.so is missing.
"""
__version__ = "0.1.0"
