"""Oracle LoRA (test infrastructure; see oracle/__init__.py).

Restates peft==0.13.2 `LoraConfig` / `get_peft_model` / `lora.Linear.forward` /
`get_peft_model_state_dict` and diffusers' `convert_state_dict_to_diffusers`
as the reference uses them:
  [REF script/train/train_audioldm_lora.py:378-385,578]
  [REF script/inference/generate_audio.py:21-36]
Spec: SURVEY.md Appendix B.7 --  y = W x + (alpha/r) * B(A x);  A ~ N(0, 1/r) (std 1/r),
B = 0 for init_lora_weights="gaussian"; module-name suffix matching; adapter "default".
"""
from dataclasses import dataclass, field
from typing import Sequence

import torch
from torch import nn


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: int = 8
    target_modules: Sequence[str] = field(default_factory=lambda: ["to_q", "to_v"])
    init_lora_weights: object = True      # True (kaiming-uniform A) | "gaussian"
    lora_dropout: float = 0.0
    bias: str = "none"


class LoraLinear(nn.Module):
    """peft.tuners.lora.Linear: keys base_layer.*, lora_A.default.weight [r,in], lora_B.default.weight [out,r]."""

    def __init__(self, base: nn.Linear, cfg: LoraConfig):
        super().__init__()
        self.base_layer = base
        self.r = cfg.r
        self.scaling = cfg.lora_alpha / cfg.r
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, cfg.r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(cfg.r, base.out_features, bias=False)})
        if cfg.init_lora_weights == "gaussian":
            nn.init.normal_(self.lora_A["default"].weight, std=1.0 / cfg.r)
        else:
            nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=5 ** 0.5)
        nn.init.zeros_(self.lora_B["default"].weight)

    def forward(self, x):
        result = self.base_layer(x)
        a = self.lora_A["default"]
        b = self.lora_B["default"]
        return result + b(a(x.to(a.weight.dtype))) * self.scaling


def _match(name, targets):
    return any(name == t or name.endswith("." + t) for t in targets)


class PeftModel(nn.Module):
    """get_peft_model wrapper: state-dict keys are prefixed `base_model.model.`."""

    def __init__(self, model):
        super().__init__()
        self.base_model = nn.Module()
        self.base_model.model = model

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)


def get_peft_model(model: nn.Module, cfg: LoraConfig) -> PeftModel:
    """In-place injection (the caller's `model` object is mutated, as in peft)."""
    for p in model.parameters():
        p.requires_grad_(False)
    targets = []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear) and _match(name, cfg.target_modules):
            targets.append(name)
    for name in targets:
        parent_name, _, leaf = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        wrapped = LoraLinear(getattr(parent, leaf), cfg)
        if leaf.isdigit():
            parent[int(leaf)] = wrapped
        else:
            setattr(parent, leaf, wrapped)
    return PeftModel(model)


def get_peft_model_state_dict(peft_model: PeftModel):
    """LoRA tensors only, adapter name dropped (`...lora_A.weight`)."""
    out = {}
    for k, v in peft_model.state_dict().items():
        if "lora_" in k:
            out[k.replace(".default", "")] = v
    return out


def convert_state_dict_to_diffusers(sd):
    """peft -> diffusers LoRA key names (lora_A -> lora.down, lora_B -> lora.up)."""
    out = {}
    for k, v in sd.items():
        k = k.replace(".lora_A.weight", ".lora.down.weight").replace(".lora_B.weight", ".lora.up.weight")
        out[k] = v
    return out


def merged_weight(layer: LoraLinear):
    return layer.base_layer.weight + layer.scaling * (layer.lora_B["default"].weight @ layer.lora_A["default"].weight)
