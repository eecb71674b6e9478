"""Measure the igemm launch configuration (tile, LDS ring depth, split-K) of every distinct GEMM on the benchmarked paths
and write audioldm_with_lora_amd/tuned_gfx950.json (see the Tuner notes in ops.py).  Run on an MI355X:

    python tools/autotune.py [--passes 3]

Workloads: config 2 inference (UNet batch 8, 250x16, rank-4 LoRA), VAE decode + vocoder of 4 x 10 s clips, config 3 / 4
training (batch 8, 256x16, rank 8 and 16).  Keys carry the GEMM's M, so other batch sizes fall back to the heuristics."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ALDM_NO_TUNED"] = "1"                      # start from the heuristics, not from an older table
import bench  # noqa: E402
from audioldm_with_lora_amd import ops  # noqa: E402


def tune(step, tuner, passes):
    """stage 1 on first sight of each GEMM (plain eager call), then stage 2: every shortlist slot `passes` times, the whole
    step queued behind a sleep kernel so that its launches run back to back."""
    tuner.slot = None
    step()
    torch.cuda.synchronize()
    nslots = 1 + tuner.shortlist
    for _ in range(passes):
        for slot in range(nslots):
            tuner.slot = slot
            ops.sleep_us(60000)
            step()
            torch.cuda.synchronize()
    tuner.slot = None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=3)
    ap.add_argument("--shortlist", type=int, default=5)
    args = ap.parse_args()
    ops.TUNED.clear()
    tuner = ops.TUNER = ops.Tuner(shortlist=args.shortlist, verbose=bool(os.environ.get("ALDM_TUNE_VERBOSE")))
    t0 = time.time()
    from audioldm_with_lora_amd.engine import DenoiseEngine
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    unet, _ = bench.build_unet(4)
    eng = DenoiseEngine(unet, DDIMScheduler(), 4, 250, 16, 200, 2.5, use_graph=False)
    lat, pe, ne = bench.synth_inputs(4, 250, 16)
    eng.set_condition(pe, ne)
    eng.set_latents(lat)
    tune(eng.step, tuner, args.passes)
    print(f"[autotune] inference: {len(tuner.cands)} GEMMs, {time.time() - t0:.1f} s", flush=True)
    # the reference's own inference call: one prompt (UNet batch 2), 50 steps [REF script/inference/generate_audio.py:47-52]
    eng1 = DenoiseEngine(unet, DDIMScheduler(), 1, 250, 16, 50, 5.0, use_graph=False)
    lat1, pe1, ne1 = bench.synth_inputs(1, 250, 16)
    eng1.set_condition(pe1, ne1)
    eng1.set_latents(lat1)
    tune(eng1.step, tuner, args.passes)
    print(f"[autotune] + inference at batch 1: {len(tuner.cands)} GEMMs, {time.time() - t0:.1f} s", flush=True)
    del eng1

    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    torch.manual_seed(0)
    vae, voc = AutoencoderKL().cuda(), SpeechT5HifiGan().cuda()
    z = torch.randn(4, 8, 250, 16, device="cuda")
    tune(lambda: voc(vae.decode(z / vae.config.scaling_factor).sample.squeeze(1)), tuner, args.passes)
    print(f"[autotune] + VAE decode / vocoder: {len(tuner.cands)} GEMMs, {time.time() - t0:.1f} s", flush=True)
    # the training loop body's input side [REF train:495-524]: VAE encode of 8 log-mels, CLAP text tower on captions of <= 64 tokens
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.script.train import synthetic_batch
    clap = ClapTextModelWithProjection().cuda()
    b = synthetic_batch(8, torch.Generator().manual_seed(21), vocab=clap.cfg["vocab_size"])
    mel = b["log_mel_spec"].cuda()
    ids = b["input_ids"].squeeze(1)[:, :64].cuda().contiguous()
    kv = clap._lengths(b["input_ids"].squeeze(1), b["attention_mask"].squeeze(1)).cuda()
    tune(lambda: (vae.encode(mel), clap.forward_device(ids, kv)), tuner, args.passes)
    print(f"[autotune] + VAE encode / CLAP tower: {len(tuner.cands)} GEMMs, {time.time() - t0:.1f} s", flush=True)
    del eng, unet, vae, voc, clap
    torch.cuda.empty_cache()

    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model
    from audioldm_with_lora_amd.training import LoraTrainer
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    for r in (8, 16):
        torch.manual_seed(1234)
        unet = UNet2DConditionModel()
        get_peft_model(unet, LoraConfig(r=r, lora_alpha=r, init_lora_weights="gaussian", target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
        unet.cuda()
        tr = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=100000, use_graph=False)
        g = torch.Generator().manual_seed(5)
        batch = ((torch.randn(8, 8, 256, 16, generator=g) * 0.92).cuda(), torch.randn(8, 8, 256, 16, generator=g).cuda(),
                 torch.randint(0, 1000, (8,), generator=g).cuda(), torch.nn.functional.normalize(torch.randn(8, 512, generator=g), dim=-1).cuda())
        tune(lambda: tr.step(*batch), tuner, args.passes)
        del tr, unet
        torch.cuda.empty_cache()
    print(f"[autotune] + training: {len(tuner.cands)} GEMMs, {time.time() - t0:.1f} s", flush=True)

    res = tuner.finish()
    ops.TUNER = None
    changed = 0
    gain = 0.0
    for key, (best, t_best, t_def) in sorted(res.items()):
        default = tuner.cands[key][0]
        if best != default:
            changed += 1
            if t_def is not None:                            # (the default slot of a launch whose tile the caller forces -- gn_in -- is never timed)
                gain += (t_def - t_best)
            ops.TUNED[key] = best
            print(f"[autotune] {key}: {default} {(t_def or 0.0) * 1e3:.1f} us -> {best} {t_best * 1e3:.1f} us", flush=True)
    ops.save_tuned()
    print(f"[autotune] wrote {ops.TUNED_PATH}: {changed} of {len(res)} GEMMs leave the heuristic (sum of per-launch gains {gain * 1e3:.0f} us)")


if __name__ == "__main__":
    main()
