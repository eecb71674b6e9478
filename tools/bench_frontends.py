"""Times the two training-input front ends that sit before the hot path on one MI355X:
the CLAP text tower (RoBERTa-base size, captions right-padded to 512 tokens) and the one-kernel log-mel front end."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
from audioldm_with_lora_amd.mel import LogMelFrontEnd
from audioldm_with_lora_amd.script.train import synthetic_batch
from audioldm_with_lora_amd.vae import AutoencoderKL


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


torch.manual_seed(0)
g = torch.Generator().manual_seed(0)
B = 8
batch = synthetic_batch(B, g)
clap = ClapTextModelWithProjection().cuda()
ids, mask = batch["input_ids"].squeeze(1), batch["attention_mask"].squeeze(1)
print(f"CLAP text tower, {B} captions (8-64 tokens, padded to 512): {timed(lambda: clap(input_ids=ids, attention_mask=mask)):.2f} ms", flush=True)
full = torch.ones_like(mask)
print(f"  same, no padding cut (all 512 tokens valid): {timed(lambda: clap(input_ids=ids, attention_mask=full)):.2f} ms", flush=True)
mel = LogMelFrontEnd()
wav = (torch.rand(B, 163840, generator=g) * 2 - 1).cuda()
print(f"log-mel front end, {B} x 10.24 s: {timed(lambda: mel(wav)):.3f} ms", flush=True)
vae = AutoencoderKL().cuda()
x = batch["log_mel_spec"].cuda()
print(f"VAE encode, {B} x [1, 1024, 64] mel -> latents: {timed(lambda: vae.encode(x).latent_dist.mode(), reps=5):.2f} ms", flush=True)
