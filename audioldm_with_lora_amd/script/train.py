"""LoRA fine-tuning driver -- MI355X mirror of  script/train/train_audioldm_lora.py:main  [REF train:324-626].

Same structure and the same default hyper-parameters as the reference's hard-coded literals:
base model cvssp/audioldm-s-full-v2, LoRA r=2 / alpha=2 / gaussian init on to_q,to_v [REF train:378-383], AdamW lr 1e-5
betas (0.9, 0.999) wd 1e-5 eps 1e-8 [REF train:396-403], batch 2 per process [REF train:406], polynomial LR with no
warm-up [REF train:438-443], max 97 000 steps, checkpoint every 19 400 steps [REF train:410-411,574-576].
One process per GPU (torchrun); gradients are all-reduced as ONE flat buffer over RCCL.

Two input contracts:
  --input latents (default)  the north_star boundary: VAE latents [B, 8, 256, 16] and L2-normalised CLAP prompt
                             embeddings [B, 512]; `--latents-file` loads a .pt dict {latents, prompt_embeds}.
  --input mel                the reference's collate_fn batch [REF train:415-420]: {log_mel_spec [B,1,1024,64] fp32,
                             input_ids [B,1,512], attention_mask [B,1,512]}; the loop body then runs, as the reference
                             does, `vae.encode(mel).latent_dist.sample() * scaling_factor` [REF train:495-496] and
                             `normalize(text_encoder(ids, mask).text_embeds)` [REF train:510-524] on the HIP kernels
                             (vae.py, clap_text.py).  `--batches-file` loads a .pt dict of those three tensors.
Without a file the driver trains on synthetic tensors of the same shapes (no datasets or checkpoints are available
offline).  The dataset classes, the librosa mel front end and the CLAP / KAD metrics stay out of scope.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m audioldm_with_lora_amd.script.train \
        --max-train-steps 100
"""
import argparse
import json
import os
import time

import torch

from .. import dp
from ..clap_text import ClapTextModelWithProjection
from ..lora import LoraConfig, get_peft_model
from ..scheduler import DDIMScheduler
from ..training import LoraTrainer
from ..unet import UNet2DConditionModel
from ..vae import AutoencoderKL


def encode_batch(vae, text_encoder, batch, generator=None, mel_frontend=None):
    """collate_fn batch -> (latents [B,8,H/4,16], prompt_embeds [B,512]) exactly as [REF train:495-524].  A batch that
    carries `waveform` [B,1,T] instead of `log_mel_spec` goes through the GPU log-mel front end first (mel.py)."""
    dev = vae.post_quant_conv.weight.device
    if "log_mel_spec" in batch:
        mel = batch["log_mel_spec"].to(dev, torch.float32)
    else:
        mel = mel_frontend(batch["waveform"].to(dev, torch.float32).flatten(1))
    latents = vae.encode(mel).latent_dist.sample(generator) * vae.config.scaling_factor
    input_ids = batch["input_ids"].squeeze(1)
    attention_mask = batch["attention_mask"].squeeze(1)
    text_embeds = text_encoder(input_ids=input_ids, attention_mask=attention_mask, return_dict=True).text_embeds
    return latents, torch.nn.functional.normalize(text_embeds, dim=-1)


def synthetic_batch(B, g, vocab=50265, max_len=512, pad=1):
    """Shapes and dtypes of the reference's collate_fn output; captions of 8..64 tokens right-padded to 512."""
    ids = torch.full((B, 1, max_len), pad, dtype=torch.long)
    mask = torch.zeros(B, 1, max_len, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(8, 65, (1,), generator=g))
        ids[b, 0, :n] = torch.randint(3, vocab, (n,), generator=g)
        ids[b, 0, 0], ids[b, 0, n - 1] = 0, 2
        mask[b, 0, :n] = 1
    mel = torch.randn(B, 1, 1024, 64, generator=g) * 2.0 - 5.0          # log-mel-like range
    return {"log_mel_spec": mel, "input_ids": ids, "attention_mask": mask}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-dir", default=None, help="local diffusers-format directory of cvssp/audioldm-s-full-v2")
    ap.add_argument("--input", choices=("latents", "mel"), default="latents")
    ap.add_argument("--latents-file", default=None)
    ap.add_argument("--batches-file", default=None)
    ap.add_argument("--output-dir", default="data/LoRA_weight/r2_alpha2")
    ap.add_argument("--rank", type=int, default=2)
    ap.add_argument("--lora-alpha", type=int, default=2)
    ap.add_argument("--target-modules", default="to_q,to_v")
    ap.add_argument("--learning-rate", type=float, default=1.0e-5)
    ap.add_argument("--weight-decay", type=float, default=1e-5)
    ap.add_argument("--train-batch-size", type=int, default=2)
    ap.add_argument("--max-train-steps", type=int, default=97000)
    ap.add_argument("--checkpointing-steps", type=int, default=9700 * 2)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tiny", action="store_true", help="shrunken random-init models of the same topology (smoke tests)")
    args = ap.parse_args(argv)

    accelerator = dp.Accelerator(gradient_accumulation_steps=1, mixed_precision=None)
    device = accelerator.device
    torch.manual_seed(1234)                                       # identical base weights on every rank
    if args.model_dir:
        unet = UNet2DConditionModel.from_pretrained(args.model_dir, subfolder="unet")
        noise_scheduler = DDIMScheduler.from_pretrained(args.model_dir, subfolder="scheduler")
    elif args.tiny:
        from .. import configs
        unet, noise_scheduler = UNet2DConditionModel(**configs.tiny_unet()), DDIMScheduler()
    else:
        unet, noise_scheduler = UNet2DConditionModel(), DDIMScheduler()
    unet.requires_grad_(False)
    cfg = LoraConfig(r=args.rank, lora_alpha=args.lora_alpha, init_lora_weights="gaussian",
                     target_modules=args.target_modules.split(","))
    peft_unet = get_peft_model(unet, cfg)
    unet.to(device)
    trainer = LoraTrainer(unet, noise_scheduler, lr=args.learning_rate, betas=(0.9, 0.999), weight_decay=args.weight_decay,
                          eps=1e-08, max_train_steps=args.max_train_steps, device=device)

    vae = text_encoder = None
    if args.input == "mel":
        if args.model_dir:
            vae = AutoencoderKL.from_pretrained(args.model_dir, subfolder="vae")
            text_encoder = ClapTextModelWithProjection.from_pretrained(args.model_dir, subfolder="text_encoder")
        elif args.tiny:
            vae = AutoencoderKL(**configs.tiny_vae())
            text_encoder = ClapTextModelWithProjection(**dict(configs.tiny_clap_text(), max_position_embeddings=514,
                                                              projection_dim=unet.cfg["class_embed_input_dim"]))
        else:
            vae, text_encoder = AutoencoderKL(), ClapTextModelWithProjection()
        vae.requires_grad_(False).to(device)
        text_encoder.requires_grad_(False).to(device)
    data = torch.load(args.latents_file) if args.latents_file else (torch.load(args.batches_file) if args.batches_file else None)
    g = torch.Generator().manual_seed(args.seed + 1000 * accelerator.process_index)      # per-rank noise / timesteps
    g_dev = torch.Generator(device=device).manual_seed(args.seed + 1000 * accelerator.process_index + 1)
    B = args.train_batch_size
    t0, train_loss = time.time(), 0.0
    for global_step in range(1, args.max_train_steps + 1):
        if args.input == "mel":
            if data is not None:
                idx = torch.randint(0, data["log_mel_spec"].shape[0], (B,), generator=g)
                batch = {k: data[k][idx] for k in ("log_mel_spec", "input_ids", "attention_mask")}
            else:
                batch = synthetic_batch(B, g, vocab=text_encoder.cfg["vocab_size"])
            latents, prompt_embeds = encode_batch(vae, text_encoder, batch, g_dev)
        elif data is not None:
            idx = torch.randint(0, data["latents"].shape[0], (B,), generator=g)
            latents, prompt_embeds = data["latents"][idx], data["prompt_embeds"][idx]
        else:
            latents = torch.randn(B, 8, 256, 16, generator=g) * 0.9228
            prompt_embeds = torch.nn.functional.normalize(torch.randn(B, unet.cfg["class_embed_input_dim"], generator=g), dim=-1)
        noise = torch.randn(latents.shape, generator=g)
        timesteps = torch.randint(0, noise_scheduler.config.num_train_timesteps, (B,), generator=g).long()
        loss = trainer.step(latents, noise, timesteps, prompt_embeds)
        if global_step % 10 == 0 or global_step == args.max_train_steps:
            train_loss = float(loss)
            if accelerator.is_main_process:
                print(json.dumps({"step": global_step, "train_loss": train_loss, "lr": trainer.lr(global_step),
                                  "clips_per_sec": global_step * B * accelerator.num_processes / (time.time() - t0)}), flush=True)
        if global_step % args.checkpointing_steps == 0 and accelerator.is_main_process:
            accelerator.save_state(os.path.join(args.output_dir, f"checkpoint-{global_step}"), trainer)
    accelerator.wait_for_everyone()
    if accelerator.is_main_process:
        accelerator.save_state(os.path.join(args.output_dir, f"checkpoint-{args.max_train_steps}"), trainer)
    accelerator.end_training()
    return train_loss


if __name__ == "__main__":
    main()
