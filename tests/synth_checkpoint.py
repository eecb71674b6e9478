"""Test helper: writes a synthetic diffusers-format model directory (the layout of cvssp/audioldm-s-full-v2 that the reference
loads with `from_pretrained(base_model_id, subfolder=...)` [REF script/train/train_audioldm_lora.py:364-371],
[REF script/inference/generate_audio.py:18,42]) with shrunken random-init weights -- no checkpoints are available offline.

    <dir>/unet/{config.json, diffusion_pytorch_model.safetensors}
    <dir>/vae/{config.json, diffusion_pytorch_model.safetensors}
    <dir>/vocoder/{config.json, model.safetensors}
    <dir>/scheduler/scheduler_config.json
    <dir>/text_encoder/{config.json, model.safetensors}
    <dir>/tokenizer/{tokenizer.json, tokenizer_config.json}      (byte-level BPE trained on three captions)
The config.json files carry the diffusers / transformers key NAMES (e.g. `attention_head_dim`,
`projection_class_embeddings_input_dim`, `text_config`), which is what the loaders must parse.
"""
import json
import os

import torch
from safetensors.torch import save_file


def _dump(d, name, obj):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, name), "w") as f:
        json.dump(obj, f)


def write_tokenizer(d, model_max_length=32):
    from tokenizers import ByteLevelBPETokenizer
    from tokenizers.processors import RobertaProcessing
    os.makedirs(d, exist_ok=True)
    tok = ByteLevelBPETokenizer()
    tok.train_from_iterator(["An instrumental hip-hop track in the subgenre of boom bap", "a dog barking in the rain",
                             "techno music with heavy bass"] * 10, vocab_size=190, min_frequency=1,
                            special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>"])
    tok._tokenizer.post_processor = RobertaProcessing(sep=("</s>", 2), cls=("<s>", 0))
    tok.save(os.path.join(d, "tokenizer.json"))
    _dump(d, "tokenizer_config.json", {"model_max_length": model_max_length, "tokenizer_class": "RobertaTokenizerFast",
                                      "bos_token": "<s>", "eos_token": "</s>", "pad_token": "<pad>", "unk_token": "<unk>",
                                      "cls_token": "<s>", "sep_token": "</s>", "mask_token": "<mask>"})
    return tok.get_vocab_size()


def write_model_dir(root, seed=0, with_text=True):
    """Returns the source modules {unet, vae, vocoder, text_encoder} (CPU, random init) whose weights were written."""
    from audioldm_with_lora_amd import configs
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    torch.manual_seed(seed)
    ucfg, vcfg, hcfg = configs.tiny_unet(), configs.tiny_vae(), configs.tiny_vocoder()
    unet, vae, voc = UNet2DConditionModel(**ucfg), AutoencoderKL(**vcfg), SpeechT5HifiGan(**hcfg)
    g = torch.Generator().manual_seed(seed + 1)
    hsd = voc.state_dict()
    for k, v in hsd.items():                                  # O(1) activations through the vocoder stack
        if k.endswith("weight"):
            fan_in = v[0].numel() if "upsampler" not in k else v.shape[0] * v.shape[2] / 2
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)
    voc.load_state_dict(hsd)
    cont = lambda sd: {k: v.detach().contiguous() for k, v in sd.items()}
    d = os.path.join(root, "unet")
    _dump(d, "config.json", {
        "_class_name": "UNet2DConditionModel", "in_channels": ucfg["in_channels"], "out_channels": ucfg["out_channels"],
        "block_out_channels": list(ucfg["block_out_channels"]), "layers_per_block": ucfg["layers_per_block"],
        "attention_head_dim": ucfg["num_heads"], "cross_attention_dim": list(ucfg["cross_attention_dim"]),
        "projection_class_embeddings_input_dim": ucfg["class_embed_input_dim"], "class_embed_type": "simple_projection",
        "class_embeddings_concat": True, "norm_num_groups": ucfg["norm_num_groups"], "norm_eps": ucfg["norm_eps"],
        "down_block_types": list(ucfg["down_block_types"]), "up_block_types": list(ucfg["up_block_types"]),
        "mid_block_type": "UNetMidBlock2DCrossAttn", "act_fn": "silu", "sample_size": 128})
    save_file(cont(unet.state_dict()), os.path.join(d, "diffusion_pytorch_model.safetensors"))
    d = os.path.join(root, "vae")
    _dump(d, "config.json", {"_class_name": "AutoencoderKL", "in_channels": 1, "out_channels": 1,
                             "latent_channels": vcfg["latent_channels"], "block_out_channels": list(vcfg["block_out_channels"]),
                             "layers_per_block": vcfg["layers_per_block"], "norm_num_groups": vcfg["norm_num_groups"],
                             "scaling_factor": vcfg["scaling_factor"], "act_fn": "silu", "sample_size": 512})
    save_file(cont(vae.state_dict()), os.path.join(d, "diffusion_pytorch_model.safetensors"))
    d = os.path.join(root, "vocoder")
    _dump(d, "config.json", {k: (list(map(list, v)) if k == "resblock_dilation_sizes" else list(v) if isinstance(v, tuple) else v)
                             for k, v in hcfg.items()})
    save_file(cont(voc.state_dict()), os.path.join(d, "model.safetensors"))
    _dump(os.path.join(root, "scheduler"), "scheduler_config.json", dict(configs.SCHEDULER, _class_name="DDIMScheduler"))
    out = dict(unet=unet, vae=vae, vocoder=voc, text_encoder=None)
    if with_text:
        vocab = write_tokenizer(os.path.join(root, "tokenizer"))
        ccfg = dict(configs.tiny_clap_text(), vocab_size=vocab, projection_dim=ucfg["class_embed_input_dim"], max_position_embeddings=40)
        clap = ClapTextModelWithProjection(**ccfg)
        d = os.path.join(root, "text_encoder")
        _dump(d, "config.json", {"model_type": "clap", "projection_dim": ccfg["projection_dim"],
                                 "text_config": {k: ccfg[k] for k in ccfg if k != "projection_dim"} | {"projection_dim": ccfg["projection_dim"]}})
        sd = cont(clap.state_dict())
        sd["text_model.embeddings.position_ids"] = torch.arange(ccfg["max_position_embeddings"]).unsqueeze(0)   # persistent buffer in 4.29 checkpoints
        save_file(sd, os.path.join(d, "model.safetensors"))
        out["text_encoder"] = clap
    return out


def write_accelerate_unet_checkpoint(path, unet, r=2, alpha=2, targets=("to_q", "to_v"), seed=3):
    """`accelerator.save_state` of the reference [REF train:576]: the FULL peft-wrapped UNet (frozen base under `.base_layer.` for
    wrapped modules + LoRA tensors), prefix `base_model.model.`, adapter name `default`.  Returns the LoRA tensors written."""
    import copy
    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model
    u = copy.deepcopy(unet)
    pm = get_peft_model(u, LoraConfig(r=r, lora_alpha=alpha, target_modules=list(targets), init_lora_weights="gaussian"))
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.detach().clone().contiguous() for k, v in pm.state_dict().items()}
    for k in sd:
        if "lora_B" in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    os.makedirs(os.path.dirname(path), exist_ok=True)
    save_file(sd, path)
    return {k: v for k, v in sd.items() if "lora_" in k}, len(sd)
