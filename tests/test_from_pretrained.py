"""`from_pretrained` of every class and the inference driver, on a synthetic diffusers-format directory (tests/synth_checkpoint.py).

The reference's loading sequence [REF script/inference/generate_audio.py:14-42]:
    unet = UNet2DConditionModel.from_pretrained(id, subfolder="unet") ; unet_lora = get_peft_model(unet, LoraConfig(r=2, ...))
    unet_lora.load_state_dict(load_file("checkpoint-19400/model.safetensors"), strict=False)      # accelerate's FULL peft-wrapped UNet
    pipe = DiffusionPipeline.from_pretrained(id, unet=unet).to(device) ; pipe(prompt, ...).audios[0] ; write wav
and the trainer's [REF script/train/train_audioldm_lora.py:364-371].  Loading is host logic (runs without a GPU); running the
loaded pipeline is the -m gpu half.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth_checkpoint  # noqa: E402


def test_every_class_loads_from_a_diffusers_format_directory(tmp_path):
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    root = str(tmp_path / "audioldm-s-tiny")
    src = synth_checkpoint.write_model_dir(root)
    unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
    vae = AutoencoderKL.from_pretrained(root, subfolder="vae")
    voc = SpeechT5HifiGan.from_pretrained(root, subfolder="vocoder")
    sch = DDIMScheduler.from_pretrained(root, subfolder="scheduler")
    clap = ClapTextModelWithProjection.from_pretrained(root, subfolder="text_encoder")
    for got, want in ((unet, src["unet"]), (vae, src["vae"]), (voc, src["vocoder"]), (clap, src["text_encoder"])):
        a, b = got.state_dict(), want.state_dict()
        assert set(a) == set(b)
        assert all(torch.equal(a[k], b[k]) for k in a)
    assert unet.cfg["num_heads"] == 4 and unet.cfg["class_embed_input_dim"] == 64 and tuple(unet.cfg["block_out_channels"]) == (32, 64, 96, 160)
    assert sch.config.num_train_timesteps == 1000 and sch.config.steps_offset == 1
    assert abs(vae.config.scaling_factor - 0.9227914214134216) < 1e-12
    assert tuple(voc.config.upsample_rates) == (5, 4, 2, 2, 2) and voc.config.sampling_rate == 16000
    # the pipeline loader, with and without a caller-supplied UNet [REF generate_audio.py:42] [REF train:365]
    pipe = AudioLDMPipeline.from_pretrained(root, unet=unet)
    assert pipe.unet is unet and pipe.text_encoder is not None and pipe.tokenizer is not None
    assert pipe.tokenizer.model_max_length == 32 and pipe.tokenizer.pad_token_id == 1
    tok = pipe.tokenizer(["a dog barking", ""], padding="max_length", max_length=pipe.tokenizer.model_max_length, truncation=True,
                         return_tensors="pt")
    assert tok.input_ids.shape == (2, 32) and int(tok.input_ids[0, 0]) == 0 and tok.attention_mask[1].sum() == 2
    pipe2 = AudioLDMPipeline.from_pretrained(root)
    assert pipe2.unet is not unet and pipe2.vae_scale_factor == 4
    h, n = pipe2.geometry(10.0)
    assert (h, n) == (1000, 160000)
    with pytest.raises(FileNotFoundError):
        UNet2DConditionModel.from_pretrained(str(tmp_path / "nope"), subfolder="unet")
    with pytest.raises(FileNotFoundError):
        AudioLDMPipeline.from_pretrained("cvssp/audioldm-s-full-v2")              # hub ids cannot resolve offline


def test_accelerate_full_unet_checkpoint_loads_through_the_peft_wrapper(tmp_path):
    """[REF generate_audio.py:21-33]: the checkpoint holds base keys (`...to_q.base_layer.weight`) AND LoRA keys; strict=False
    must take all of them with nothing unexpected, and the adapter must equal what was written."""
    from safetensors.torch import load_file
    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model, get_peft_model_state_dict
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    root = str(tmp_path / "m")
    src = synth_checkpoint.write_model_dir(root, with_text=False)
    ck = str(tmp_path / "checkpoint-19400" / "model.safetensors")
    lora_written, n_keys = synth_checkpoint.write_accelerate_unet_checkpoint(ck, src["unet"])
    unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
    v0 = unet.plan_version
    unet_lora = get_peft_model(unet, LoraConfig(r=2, lora_alpha=4, init_lora_weights="gaussian", target_modules=["to_q", "to_v"]))
    sd = load_file(ck)
    assert len(sd) == n_keys > len(lora_written) == 128
    res = unet_lora.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not res.missing_keys
    assert unet.plan_version > v0                                 # packed operands / captured graphs must follow the load
    got = get_peft_model_state_dict(unet_lora)
    for k, v in lora_written.items():
        assert torch.equal(got[k.replace(".default", "")], v)


@pytest.mark.gpu
def test_inference_script_main_end_to_end(tmp_path):
    """script/inference.py::main on the synthetic directory: from_pretrained -> get_peft_model -> accelerate checkpoint ->
    pipeline from prompt STRING (tokenizer + CLAP tower + loop + VAE + vocoder) -> wav; compared with a pipeline assembled by
    hand from the same weights, and required to differ from the base model's audio."""
    import numpy as np
    from scipy.io import wavfile
    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.script import inference
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from safetensors.torch import load_file
    root = str(tmp_path / "m")
    src = synth_checkpoint.write_model_dir(root)
    ck = str(tmp_path / "checkpoint-19400" / "model.safetensors")
    synth_checkpoint.write_accelerate_unet_checkpoint(ck, src["unet"], r=2, alpha=2)
    out = str(tmp_path / "generated_audio_LoRA" / "ex.wav")
    prompt = "An instrumental hip-hop track in the subgenre of boom bap"
    args = ["--model-dir", root, "--lora-weights", ck, "--rank", "2", "--lora-alpha", "2", "--target-modules", "to_q,to_v",
            "--prompt", prompt, "--steps", "5", "--audio-length", "1.28", "--guidance-scale", "5.0", "--output", out, "--seed", "77"]
    inference.main(args)
    sr, wav = wavfile.read(out)
    assert sr == 16000 and wav.shape == (20480,) and wav.dtype == np.float32 and np.isfinite(wav).all()

    def by_hand(with_lora):
        unet = UNet2DConditionModel.from_pretrained(root, subfolder="unet")
        if with_lora:
            pm = get_peft_model(unet, LoraConfig(r=2, lora_alpha=2, init_lora_weights="gaussian", target_modules=["to_q", "to_v"]))
            pm.load_state_dict(load_file(ck), strict=False)
        pipe = AudioLDMPipeline.from_pretrained(root, unet=unet).to("cuda")
        return pipe(prompt=prompt, num_inference_steps=5, audio_length_in_s=1.28, guidance_scale=5.0,
                    generator=torch.Generator().manual_seed(77)).audios[0]

    same, base = by_hand(True), by_hand(False)
    assert np.array_equal(np.asarray(same, dtype=np.float32), wav)        # deterministic: same weights, same seed, same kernels
    rel = float(np.linalg.norm(wav - base) / np.linalg.norm(base))
    assert rel > 1e-2, rel                                                # the adapter was really applied
