"""Oracle building blocks vs independent compositions of torch primitives."""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import configs
from oracle.lora import (LoraConfig, LoraLinear, convert_state_dict_to_diffusers, get_peft_model,
                         get_peft_model_state_dict, merged_weight)
from oracle.pipeline import audio_geometry, polynomial_lr
from oracle.unet import Attention, GEGLU, UNet2DConditionModel, Upsample2D, timestep_embedding
from oracle.vae import AutoencoderKL


def test_unet_structure_known_answers():
    u = UNet2DConditionModel()
    n = sum(p.numel() for p in u.parameters())
    assert n == 185_036_552                      # AudioLDM-S UNet (185 M)
    sd = u.state_dict()
    assert sd["conv_in.weight"].shape == (128, 8, 3, 3)
    assert sd["up_blocks.0.resnets.2.conv1.weight"].shape == (640, 1024, 3, 3)
    assert sd["up_blocks.2.resnets.0.norm1.weight"].shape == (640,)
    assert sd["down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape == (256, 256)
    assert sd["down_blocks.3.attentions.1.transformer_blocks.0.ff.net.0.proj.weight"].shape == (5120, 640)
    assert sd["mid_block.resnets.0.time_emb_proj.weight"].shape == (640, 1024)
    assert "down_blocks.3.downsamplers.0.conv.weight" not in sd
    assert "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.bias" not in sd
    assert "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_out.0.bias" in sd


def test_timestep_embedding_cos_first():
    t = torch.tensor([0, 7, 996])
    e = timestep_embedding(t, 128, True, 0)
    k = torch.arange(64, dtype=torch.float64)
    f = torch.exp(-math.log(10000.0) * k / 64)
    arg = t.double()[:, None] * f[None]
    want = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
    torch.testing.assert_close(e.double(), want, atol=2e-4, rtol=0)
    assert torch.all(e[0, :64] == 1) and torch.all(e[0, 64:] == 0)


def test_attention_vs_manual_softmax():
    torch.manual_seed(0)
    a = Attention(64, 4, 16).eval()
    x = torch.randn(2, 10, 64)
    q, k, v = a.to_q(x), a.to_k(x), a.to_v(x)
    sp = lambda z: z.view(2, 10, 4, 16).permute(0, 2, 1, 3)
    s = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / 4.0, -1) @ sp(v)
    want = a.to_out[0](s.permute(0, 2, 1, 3).reshape(2, 10, 64))
    torch.testing.assert_close(a(x), want, rtol=1e-5, atol=1e-5)


def test_geglu_value_first_gate_second_erf_gelu():
    torch.manual_seed(0)
    g = GEGLU(8, 16)
    x = torch.randn(3, 8)
    y = g.proj(x)
    want = y[:, :16] * (0.5 * y[:, 16:] * (1 + torch.erf(y[:, 16:] / math.sqrt(2))))
    torch.testing.assert_close(g(x), want, rtol=1e-5, atol=1e-6)


def test_upsample_size_targeted_nearest_indices():
    x = torch.arange(32 * 2, dtype=torch.float32).view(1, 1, 32, 2)
    y = F.interpolate(x, size=(63, 4), mode="nearest")
    ih = torch.tensor([(i * 32) // 63 for i in range(63)])
    iw = torch.tensor([(j * 2) // 4 for j in range(4)])
    assert torch.equal(y[0, 0], x[0, 0][ih][:, iw])
    for (i, o) in ((63, 125), (125, 250), (32, 64)):
        z = F.interpolate(torch.arange(i, dtype=torch.float32).view(1, 1, i, 1), size=(o, 1), mode="nearest")
        assert z.view(-1).tolist() == [float((d * i) // o) for d in range(o)]


def test_unet_odd_sizes_and_batch_independence():
    torch.manual_seed(0)
    u = UNet2DConditionModel(**configs.tiny_unet()).eval()
    x = torch.randn(2, 8, 63, 16)
    c = torch.randn(2, 64)
    with torch.no_grad():
        y = u(x, torch.tensor(501), class_labels=c)[0]
        y0 = u(x[:1], torch.tensor([501]), class_labels=c[:1])[0]
    assert y.shape == x.shape
    torch.testing.assert_close(y[:1], y0, rtol=1e-4, atol=1e-5)


def test_lora_known_answers():
    torch.manual_seed(0)
    u = UNet2DConditionModel(**configs.tiny_unet()).eval()
    x, c, t = torch.randn(1, 8, 16, 16), torch.randn(1, 64), torch.tensor([10])
    with torch.no_grad():
        base = u(x, t, class_labels=c)[0]
    pm = get_peft_model(u, LoraConfig(r=2, lora_alpha=2, target_modules=["to_q", "to_v"],
                                      init_lora_weights="gaussian"))
    wrapped = [n for n, m in u.named_modules() if isinstance(m, LoraLinear)]
    assert len(wrapped) == 64
    with torch.no_grad():
        y = pm(x, t, class_labels=c)[0]
    assert torch.equal(y, base)                                    # B = 0  =>  bit-identical
    trainable = [n for n, p in pm.named_parameters() if p.requires_grad]
    assert len(trainable) == 128 and all("lora_" in n for n in trainable)
    k = "base_model.model.down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.lora_A.default.weight"
    assert k in pm.state_dict()
    sd = get_peft_model_state_dict(pm)
    assert k.replace(".default", "") in sd and len(sd) == 128
    dsd = convert_state_dict_to_diffusers(sd)
    assert any(k.endswith("to_q.lora.down.weight") for k in dsd)

    u2 = UNet2DConditionModel(**configs.tiny_unet())
    get_peft_model(u2, LoraConfig(r=4, lora_alpha=4, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    assert sum(isinstance(m, LoraLinear) for m in u2.modules()) == 128
    # full-size count: 112 640 * r trainable parameters for q/k/v/o
    full = UNet2DConditionModel()
    get_peft_model(full, LoraConfig(r=8, lora_alpha=8, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    assert sum(p.numel() for p in full.parameters() if p.requires_grad) == 112_640 * 8


def test_lora_merged_equals_unmerged():
    torch.manual_seed(1)
    lin = torch.nn.Linear(32, 48, bias=True)
    l = LoraLinear(lin, LoraConfig(r=4, lora_alpha=8, init_lora_weights="gaussian"))
    torch.nn.init.normal_(l.lora_B["default"].weight, std=0.02)
    assert abs(float(l.lora_A["default"].weight.std()) - 0.25) < 0.05       # std = 1/r
    x = torch.randn(5, 32)
    want = F.linear(x, merged_weight(l), lin.bias)
    torch.testing.assert_close(l(x), want, rtol=1e-5, atol=1e-5)


def test_polynomial_lr_matches_transformers_golden():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "poly_lr.npz"))
    for s, lr in zip(z["steps"], z["lr"]):
        assert abs(polynomial_lr(int(s), 1e-5, 97000) - lr) < 1e-15


def test_audio_geometry():
    from types import SimpleNamespace
    vc = SimpleNamespace(upsample_rates=(5, 4, 2, 2, 2), sampling_rate=16000)
    assert audio_geometry(10.0, vc) == (1000, 160000)
    assert audio_geometry(5.0, vc) == (500, 80000)
    assert audio_geometry(4.0, vc) == (400, 64000)
    assert audio_geometry(10.24, vc) == (1024, 163840)
    assert audio_geometry(5.025, vc)[0] == 504           # 502 frames -> next multiple of 4


def test_vae_shapes_and_scale_factor():
    torch.manual_seed(0)
    v = AutoencoderKL(**configs.tiny_vae()).eval()
    z = torch.randn(1, 8, 6, 4)
    with torch.no_grad():
        mel = v.decode(z).sample
        lat = v.encode(torch.randn(1, 1, 24, 16)).latent_dist
    assert mel.shape == (1, 1, 24, 16)
    assert lat.mean.shape == (1, 8, 6, 4)
    full = AutoencoderKL()
    assert "decoder.mid_block.attentions.0.to_q.bias" in full.state_dict()
    assert full.state_dict()["decoder.conv_in.weight"].shape == (512, 8, 3, 3)
