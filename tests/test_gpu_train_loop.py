"""GPU: the reference trainer's own loop body, imports swapped, and its train -> validate flow in one process.

(1) the call sequence of [REF script/train/train_audioldm_lora.py:479-565] run against this package's drop-in objects
    (`unet(...)[0]` with a grad_fn, `F.mse_loss`, `accelerator.backward`, `clip_grad_norm_`, `optimizer.step`, `lr_scheduler.step`,
    `optimizer.zero_grad`, `accelerator.log / save_state`) and compared, step by step, with the same loop on the CPU oracle
    (torch autograd + torch.optim.AdamW + the transformers polynomial schedule).
(2) [REF train:597-603,142]: after training, `AudioLDMPipeline(unet=<the live, just-trained UNet>)` must run with the TRAINED
    adapter (the packed LoRA operands and any captured denoise graph follow the optimiser), compared with the oracle pipeline
    loaded from `get_peft_model_state_dict` and asserted different from the base model's audio.
"""
import json
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _models(seed, r=2, alpha=2, targets=("to_q", "to_v")):
    from audioldm_with_lora_amd import lora as plora
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from oracle import configs
    from oracle import lora as olora
    from oracle.unet import UNet2DConditionModel as OUNet
    cfg = configs.tiny_unet()
    torch.manual_seed(seed)
    ref = OUNet(**cfg)
    unet = UNet2DConditionModel(**cfg)
    unet.load_state_dict(ref.state_dict())
    unet.requires_grad_(False)
    ref.requires_grad_(False)
    pref = olora.get_peft_model(ref, olora.LoraConfig(r=r, lora_alpha=alpha, target_modules=list(targets), init_lora_weights="gaussian"))
    punet = plora.get_peft_model(unet, plora.LoraConfig(r=r, lora_alpha=alpha, target_modules=list(targets), init_lora_weights="gaussian"))
    g = torch.Generator().manual_seed(seed + 1)
    sd = pref.state_dict()
    for k in sd:
        if "lora_B" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    pref.load_state_dict(sd)
    punet.load_state_dict(sd)
    return pref, punet, unet


def test_reference_loop_call_sequence_matches_oracle_loop(tmp_path):
    from audioldm_with_lora_amd import dp, optim
    from audioldm_with_lora_amd.lora import convert_state_dict_to_diffusers, get_peft_model_state_dict
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from oracle.ddim import DDIMScheduler as ODDIM
    from transformers.optimization import get_polynomial_decay_schedule_with_warmup
    pref, unet, _ = _models(0)
    g = torch.Generator().manual_seed(7)
    steps, bsz = 4, 2
    data = [dict(latents=torch.randn(bsz, 8, 16, 16, generator=g) * 0.92, noise=torch.randn(bsz, 8, 16, 16, generator=g),
                 timesteps=torch.randint(0, 1000, (bsz,), generator=g),
                 prompt_embeds=F.normalize(torch.randn(bsz, 64, generator=g), dim=-1)) for _ in range(steps)]
    lr0, max_train_steps = 1.0e-3, 20

    # ---- oracle: the reference loop with torch / transformers objects on the CPU ----
    osched = ODDIM()
    oopt = torch.optim.AdamW([p for p in pref.parameters() if p.requires_grad], lr=lr0, betas=(0.9, 0.999), weight_decay=1e-5, eps=1e-08)
    olr = get_polynomial_decay_schedule_with_warmup(oopt, 0, max_train_steps, lr_end=1e-7, power=1.0)
    want_losses = []
    pref.train()
    for batch in data:
        noisy = osched.add_noise(batch["latents"], batch["noise"], batch["timesteps"])
        pred = pref(noisy, batch["timesteps"], encoder_hidden_states=None, class_labels=batch["prompt_embeds"])[0]
        loss = F.mse_loss(pred.float(), batch["noise"].float(), reduction="mean")
        loss.backward()
        oopt.step(); olr.step(); oopt.zero_grad()
        want_losses.append(float(loss))

    # ---- the same call sequence against this package's drop-in objects (own wording; the API calls and their order are the
    #      reference's [REF train:325-346,396-447,479-576]: tracker init, optimiser / schedule construction, prepare, then per batch
    #      add_noise -> unet(...)[0] -> mse -> gather -> backward -> clip -> step -> schedule -> zero_grad -> log, then save_state) ----
    acc = dp.Accelerator(gradient_accumulation_steps=1, mixed_precision=None, log_with="wandb",
                         project_config=dp.ProjectConfiguration(project_dir=str(tmp_path), logging_dir=str(tmp_path / "log")))
    acc.init_trackers(project_name="AudioLDM-with-LoRA", config=None,
                      init_kwargs={"wandb": {"group": "gpu-exp-group-1", "tags": ["lora"], "name": "r = 2, alpha = 2"}})
    ddim = DDIMScheduler()
    unet.to(acc.device, dtype=torch.float32)
    trainable = filter(lambda p: p.requires_grad, unet.parameters())          # a one-shot iterator, as in the reference (quirk Q1)
    optimizer = optim.AdamW(trainable, lr=lr0, betas=(0.9, 0.999), weight_decay=1e-5, eps=1e-08)
    lr_scheduler = optim.get_scheduler("polynomial", optimizer=optimizer, num_warmup_steps=0,
                                       num_training_steps=max_train_steps * acc.num_processes)
    unet, optimizer, _dl, lr_scheduler = acc.prepare(unet, optimizer, data, lr_scheduler)
    got_losses, running, done = [], 0.0, 0
    unet.train()
    optimizer.zero_grad()
    for batch in data:
        with acc.accumulate(unet):
            x0, eps = batch["latents"].to(acc.device), batch["noise"].to(acc.device)
            t = batch["timesteps"].to(x0.device).long()
            cond = batch["prompt_embeds"].to(x0.device)
            pred = unet(ddim.add_noise(x0, eps, t), t, encoder_hidden_states=None, class_labels=cond,
                        cross_attention_kwargs={"scale": 1.0}, return_dict=False)[0]
            assert pred.requires_grad and pred.grad_fn is not None
            loss = F.mse_loss(pred.float(), eps.float(), reduction="mean")
            running += acc.gather(loss).mean().item()
            acc.backward(loss)
            if acc.sync_gradients:
                acc.clip_grad_norm_(trainable, 1.0)                            # exhausted iterator: clips nothing, as in the reference
            optimizer.step()
            lr_scheduler.step()
            optimizer.zero_grad()
        if acc.sync_gradients:
            acc.log({"train_loss": running}, step=done)
            got_losses.append(running)
            running, done = 0.0, done + 1
    save_path = os.path.join(str(tmp_path), f"checkpoint-{done}")
    acc.save_state(save_path)                                                  # one argument, as the reference calls it
    unwrapped_unet = acc.unwrap_model(unet)
    unet_lora_state_dict = convert_state_dict_to_diffusers(get_peft_model_state_dict(unwrapped_unet))
    acc.wait_for_everyone()
    acc.end_training()

    # losses step by step (the first is before any update: forward parity; later ones also check backward + AdamW + LR)
    for i, (a, b) in enumerate(zip(got_losses, want_losses)):
        assert abs(a - b) < 2e-2 * b + 1e-4, (i, got_losses, want_losses)
    assert abs(lr_scheduler.get_last_lr()[0] - olr.get_last_lr()[0]) < 1e-12
    # final adapter vs the oracle's
    want = {k.replace(".default", ""): v for k, v in pref.state_dict().items() if "lora_" in k}
    got = get_peft_model_state_dict(unwrapped_unet)
    assert set(got) == set(want) and len(unet_lora_state_dict) == len(got)
    num = sum(float(((got[k].float().cpu() - want[k]) ** 2).sum()) for k in want)
    den = sum(float((want[k] ** 2).sum()) for k in want)
    assert (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5
    # what the run left on disk
    from safetensors.torch import load_file
    sd = load_file(os.path.join(save_path, "model.safetensors"))
    assert len(sd) == 128 and all(k.startswith("base_model.model.") for k in sd)
    assert os.path.isfile(os.path.join(save_path, "optimizer.bin")) and os.path.isfile(os.path.join(save_path, "scheduler.bin"))
    lines = [json.loads(l) for l in open(os.path.join(str(tmp_path), "log", "AudioLDM-with-LoRA.metrics.jsonl"))]
    assert lines[0]["event"] == "init" and [l["step"] for l in lines[1:]] == list(range(steps))


def test_clip_grad_norm_really_clips_with_a_real_parameter_list():
    from audioldm_with_lora_amd import dp
    pref, punet, unet = _models(3)
    unet.cuda().train()
    acc = dp.Accelerator()
    acc.prepare(punet)
    x = torch.randn(2, 8, 16, 16).cuda()
    pred = punet(x, torch.tensor([10, 500]).cuda(), encoder_hidden_states=None, class_labels=F.normalize(torch.randn(2, 64), dim=-1).cuda())[0]
    acc.backward((pred.float() ** 2).mean())
    params = [p for p in unet.parameters() if p.requires_grad]
    total = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params)))
    assert total > 0
    ret = float(acc.clip_grad_norm_(params, total / 4))
    assert abs(ret - total) < 1e-4 * total
    after = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in params)))
    assert abs(after - total / 4) < 1e-3 * total


def test_train_then_validate_uses_the_trained_adapter():
    """[REF train:597-603]: pipeline around the live UNet after training; also a SECOND round of training + validation, which
    must not replay the first validation's captured graph."""
    from audioldm_with_lora_amd.lora import get_peft_model_state_dict
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.training import LoraTrainer
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    from oracle import configs
    from oracle.ddim import DDIMScheduler as ODDIM
    from oracle.hifigan import SpeechT5HifiGan as OVoc
    from oracle.pipeline import AudioLDMPipeline as OPipe
    from oracle.vae import AutoencoderKL as OVae
    pref, punet, unet = _models(5, r=4, alpha=4, targets=("to_q", "to_k", "to_v", "to_out.0"))
    torch.manual_seed(9)
    ov, oh = OVae(**configs.tiny_vae()).eval(), OVoc(**configs.tiny_vocoder()).eval()
    g = torch.Generator().manual_seed(8)
    hsd = oh.state_dict()
    for k, v in hsd.items():
        if k.endswith("weight"):
            fan_in = v[0].numel() if "upsampler" not in k else v.shape[0] * v.shape[2] / 2
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)
    oh.load_state_dict(hsd)
    vae, voc = AutoencoderKL(**configs.tiny_vae()), SpeechT5HifiGan(**configs.tiny_vocoder())
    vae.load_state_dict(ov.state_dict()); voc.load_state_dict(oh.state_dict())
    unet.cuda()
    pipe = AudioLDMPipeline(vae, None, None, punet, DDIMScheduler(), voc).to("cuda")     # the peft wrapper, as unwrap_model returns it
    pe = F.normalize(torch.randn(1, 64, generator=g), dim=-1)
    ne = F.normalize(torch.randn(1, 64, generator=g), dim=-1)
    lat0 = torch.randn(1, 8, 16, 16, generator=g)
    call = dict(prompt_embeds=pe, negative_prompt_embeds=ne, audio_length_in_s=0.64, num_inference_steps=4, guidance_scale=2.5)
    before = torch.from_numpy(pipe(latents=lat0.clone(), **call).audios)      # captures a graph with the INITIAL adapter
    import conftest
    rel = lambda a, b: conftest.record(float((a - b).norm() / b.norm()))

    tr = LoraTrainer(unet, DDIMScheduler(), lr=2e-2, weight_decay=0.0, max_train_steps=100)
    lat = torch.randn(2, 8, 16, 16, generator=g) * 0.9
    noise = torch.randn(2, 8, 16, 16, generator=g)
    t = torch.randint(0, 1000, (2,), generator=g)
    emb = F.normalize(torch.randn(2, 64, generator=g), dim=-1)
    prev = before
    for round_ in range(2):
        for _ in range(5):
            tr.step(lat, noise, t, emb)
        got = torch.from_numpy(pipe(latents=lat0.clone(), **call).audios)
        # oracle pipeline with the trained adapter loaded from the peft state dict
        sd = {k.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight"): v.float().cpu()
              for k, v in get_peft_model_state_dict(punet).items()}
        pref.load_state_dict(sd, strict=False)
        want = torch.from_numpy(OPipe(pref.base_model.model, ov, oh, ODDIM())(pe, ne, audio_length_in_s=0.64, num_inference_steps=4,
                                                                              guidance_scale=2.5, latents=lat0.clone()).audios)
        assert rel(got, want) < 8e-2, (round_, rel(got, want))
        assert rel(got, prev) > 2 * rel(got, want), (round_, rel(got, prev), rel(got, want))   # NOT the previous adapter's audio
        prev = got
