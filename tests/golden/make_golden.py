"""Regenerates the committed golden vectors in tests/golden/.

Sources of truth (none of them is a reference file):
  * vocoder_tiny.npz   -- transformers.SpeechT5HifiGan (installed package; the very class the
                          reference loads at [REF script/train/train_audioldm_lora.py:371]) with a
                          shrunken config and seeded weights: weights, mel input, waveform output.
  * ddim_tables.npz    -- closed-form DDIM known answers, computed here in
                          float64 numpy independently of oracle/ddim.py.
  * poly_lr.npz        -- transformers.optimization.get_polynomial_decay_schedule_with_warmup values.
  * clap_text_tiny.npz -- transformers.ClapTextModelWithProjection (installed package; the class the reference
                          calls at [REF script/train/train_audioldm_lora.py:513-518]) with a shrunken config and
                          seeded weights: weights, right-padded input_ids / attention_mask, text_embeds and
                          last_hidden_state.
Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def vocoder():
    from transformers import SpeechT5HifiGan, SpeechT5HifiGanConfig
    from oracle.configs import tiny_vocoder
    c = tiny_vocoder()
    cfg = SpeechT5HifiGanConfig(
        model_in_dim=c["model_in_dim"], sampling_rate=c["sampling_rate"],
        upsample_initial_channel=c["upsample_initial_channel"], upsample_rates=list(c["upsample_rates"]),
        upsample_kernel_sizes=list(c["upsample_kernel_sizes"]),
        resblock_kernel_sizes=list(c["resblock_kernel_sizes"]),
        resblock_dilation_sizes=[list(d) for d in c["resblock_dilation_sizes"]],
        leaky_relu_slope=c["leaky_relu_slope"], normalize_before=c["normalize_before"])
    torch.manual_seed(20250824)
    m = SpeechT5HifiGan(cfg).eval()
    # HF initialises convs with a tiny std; re-draw so activations are O(1) and the test is meaningful.
    g = torch.Generator().manual_seed(7)
    sd = m.state_dict()
    for k, v in sd.items():
        if k.endswith("weight"):
            fan_in = v[0].numel() if "upsampler" not in k else v.shape[0] * v.shape[2]
            v.copy_(torch.randn(v.shape, generator=g) * (1.0 / fan_in) ** 0.5)
        elif k.endswith("bias"):
            v.copy_(torch.randn(v.shape, generator=g) * 0.05)
    m.load_state_dict(sd)
    mel = torch.randn(2, 12, c["model_in_dim"], generator=g)
    with torch.no_grad():
        wav = m(mel)
    out = {"w::" + k: v.numpy() for k, v in sd.items()}
    out["mel"] = mel.numpy()
    out["wav"] = wav.numpy()
    np.savez_compressed(os.path.join(HERE, "vocoder_tiny.npz"), **out)
    print("vocoder_tiny", wav.shape, float(wav.abs().mean()))


def ddim():
    n = 1000
    betas = np.linspace(0.0015 ** 0.5, 0.0195 ** 0.5, n, dtype=np.float64) ** 2
    ac = np.cumprod(1.0 - betas)
    out = {"alphas_cumprod_f64": ac}
    for steps in (10, 50, 200):
        ratio = n // steps
        ts = (np.arange(steps) * ratio).round()[::-1].astype(np.int64) + 1
        out[f"timesteps_{steps}"] = ts
        out[f"prev_{steps}"] = ts - ratio
    np.savez_compressed(os.path.join(HERE, "ddim_tables.npz"), **out)
    print("ddim", out["timesteps_10"])


def poly():
    from transformers.optimization import get_polynomial_decay_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-5)
    sch = get_polynomial_decay_schedule_with_warmup(opt, 0, 97000, lr_end=1e-7, power=1.0)
    steps = [0, 1, 2, 10, 1000, 48500, 96999, 97000, 97001, 100000]
    vals = []
    cur = 0
    for s in steps:
        while cur < s:
            opt.step(); sch.step(); cur += 1
        vals.append(sch.get_last_lr()[0])
    np.savez_compressed(os.path.join(HERE, "poly_lr.npz"), steps=np.array(steps), lr=np.array(vals, dtype=np.float64))
    print("poly", vals[:3], vals[-3:])


def hf_clap_text(cfg):
    from transformers import ClapTextConfig, ClapTextModelWithProjection
    hc = ClapTextConfig(vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"],
                        num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                        intermediate_size=cfg["intermediate_size"], max_position_embeddings=cfg["max_position_embeddings"],
                        type_vocab_size=cfg["type_vocab_size"], pad_token_id=cfg["pad_token_id"],
                        layer_norm_eps=cfg["layer_norm_eps"], projection_dim=cfg["projection_dim"], hidden_act="gelu",
                        projection_hidden_act="relu")
    return ClapTextModelWithProjection(hc).eval()


def clap_text():
    from oracle.configs import tiny_clap_text
    cfg = tiny_clap_text()
    torch.manual_seed(20250825)
    m = hf_clap_text(cfg)
    g = torch.Generator().manual_seed(11)
    sd = {k: v for k, v in m.state_dict().items() if not k.endswith(("position_ids", "token_type_ids"))}
    for k, v in sd.items():                     # HF draws N(0, 0.02): re-draw so activations are O(1)
        if k.endswith("weight") and v.dim() == 2 and "embeddings" not in k:
            v.copy_(torch.randn(v.shape, generator=g) / v.shape[1] ** 0.5)
        elif k.endswith("bias"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif "embeddings" in k and v.dim() == 2:
            v.copy_(torch.randn(v.shape, generator=g))
    m.load_state_dict(sd, strict=False)
    B, L = 3, 24
    ids = torch.randint(3, cfg["vocab_size"], (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    for b, n in enumerate((24, 9, 1)):
        ids[b, n:] = cfg["pad_token_id"]
        mask[b, n:] = 0
    with torch.no_grad():
        r = m(input_ids=ids, attention_mask=mask)
    out = {"w::" + k: v.numpy() for k, v in sd.items()}
    out.update(input_ids=ids.numpy(), attention_mask=mask.numpy(), text_embeds=r.text_embeds.numpy(),
               last_hidden_state=r.last_hidden_state.numpy())
    np.savez_compressed(os.path.join(HERE, "clap_text_tiny.npz"), **out)
    print("clap_text_tiny.npz", {k: v.shape for k, v in out.items() if not k.startswith("w::")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["vocoder", "ddim", "poly", "clap_text"]
    for name in which:
        globals()[name]()
