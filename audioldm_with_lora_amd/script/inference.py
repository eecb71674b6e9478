"""LoRA inference driver -- MI355X mirror of  script/inference/generate_audio.py:main  [REF generate_audio.py:11-59]
(and app.py [REF app.py:6-16] with --no-lora --steps 200).

Loads the UNet, injects the peft-style LoRA structure, loads adapter weights from a safetensors file with peft key
names (`base_model.model.<path>.lora_{A,B}.default.weight`, strict=False as in the reference), builds the pipeline and
writes a wav.  LoRA stays un-merged and is applied inside the fused projection GEMMs (reference quirk Q4).
Defaults follow the script: r=2, 50 DDIM steps, 10 s, guidance 5.0; alpha defaults to the TRAINED value 2 rather than the
script's inconsistent 4 (quirk Q3) -- pass --lora-alpha 4 to reproduce the script literally.
"""
import argparse
import os

import numpy as np
import torch

from ..lora import LoraConfig, get_peft_model
from ..pipeline import AudioLDMPipeline
from ..unet import UNet2DConditionModel


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", required=True, help="local diffusers-format directory of cvssp/audioldm-s-full-v2")
    ap.add_argument("--lora-weights", default=None, help="checkpoint-*/model.safetensors written by the trainer")
    ap.add_argument("--no-lora", action="store_true")
    ap.add_argument("--rank", type=int, default=2)
    ap.add_argument("--lora-alpha", type=int, default=2)
    ap.add_argument("--target-modules", default="to_q,to_v")
    ap.add_argument("--prompt", default="An instrumental hip-hop track in the subgenre of boom bap")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--audio-length", type=float, default=10.0)
    ap.add_argument("--guidance-scale", type=float, default=5.0)
    ap.add_argument("--output", default="./generated_audio_LoRA/ex.wav")
    ap.add_argument("--seed", type=int, default=None, help="seed of the initial-noise generator (the reference seeds nothing, quirk Q5)")
    args = ap.parse_args(argv)

    device = "cuda"
    unet = UNet2DConditionModel.from_pretrained(args.model_dir, subfolder="unet")
    if not args.no_lora:
        unet_lora = get_peft_model(unet, LoraConfig(r=args.rank, lora_alpha=args.lora_alpha, init_lora_weights="gaussian",
                                                    target_modules=args.target_modules.split(",")))
        if args.lora_weights:
            from safetensors.torch import load_file
            unet_lora.load_state_dict(load_file(args.lora_weights), strict=False)
    pipe = AudioLDMPipeline.from_pretrained(args.model_dir, unet=unet).to(device)
    generator = torch.Generator().manual_seed(args.seed) if args.seed is not None else None
    audio = pipe(prompt=args.prompt, num_inference_steps=args.steps, audio_length_in_s=args.audio_length,
                 guidance_scale=args.guidance_scale, generator=generator).audios[0]
    os.makedirs(os.path.dirname(os.path.abspath(args.output)), exist_ok=True)
    from scipy.io import wavfile
    wavfile.write(args.output, 16000, np.asarray(audio, dtype=np.float32))
    print(f"Generated audio saved to: {args.output}")


if __name__ == "__main__":
    main()
