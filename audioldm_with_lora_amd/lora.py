"""peft-shaped LoRA helpers for the HIP path.

Mirrors the surface the reference uses -- `LoraConfig`, `get_peft_model`, `get_peft_model_state_dict`
[REF script/train/train_audioldm_lora.py:35-36,378-385,578] [REF script/inference/generate_audio.py:8,21-36]
and diffusers' `convert_state_dict_to_diffusers` -- with peft 0.13.2 semantics: in-place
injection by module-name suffix, adapter name "default", state-dict prefix `base_model.model.`,
A ~ N(0, (1/r)^2) / B = 0 for init_lora_weights="gaussian", scaling = lora_alpha / r.

The wrapped layers are parameter containers: the arithmetic y = Wx + (alpha/r) B(Ax) runs fused inside
aldm_igemm (one launch per projection GEMM), never in Python.
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
    init_lora_weights: object = True
    lora_dropout: float = 0.0
    bias: str = "none"


class LoraLinear(nn.Module):
    """Container with peft's key layout: base_layer.{weight,bias}, lora_A.default.weight [r, in],
    lora_B.default.weight [out, r]."""

    def __init__(self, base: nn.Linear, cfg: LoraConfig):
        super().__init__()
        assert cfg.lora_dropout == 0.0 and cfg.bias == "none", "reference uses lora_dropout=0, bias='none'"
        self.base_layer = base
        self.r = cfg.r
        self.lora_alpha = cfg.lora_alpha
        self.scaling = cfg.lora_alpha / cfg.r
        dev = base.weight.device
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, cfg.r, bias=False, device=dev)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(cfg.r, base.out_features, bias=False, device=dev)})
        if cfg.init_lora_weights == "gaussian":
            nn.init.normal_(self.lora_A["default"].weight, std=1.0 / cfg.r)
        else:
            nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=5 ** 0.5)
        nn.init.zeros_(self.lora_B["default"].weight)

    @property
    def in_features(self):
        return self.base_layer.in_features

    @property
    def out_features(self):
        return self.base_layer.out_features

    def forward(self, *a, **k):
        raise RuntimeError("LoraLinear is a parameter container; the fused HIP GEMM computes it")


def _match(name, targets):
    return any(name == t or name.endswith("." + t) for t in targets)


class PeftModel(nn.Module):
    def __init__(self, model, cfg):
        super().__init__()
        self.base_model = nn.Module()
        self.base_model.model = model
        self.peft_config = {"default": cfg}

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)

    def load_state_dict(self, *a, **k):
        """[REF script/inference/generate_audio.py:32-33] loads the adapter through the wrapper: the wrapped UNet's packed
        operands (and any captured denoise graph) must follow."""
        out = super().load_state_dict(*a, **k)
        inner = self.base_model.model
        if hasattr(inner, "invalidate_packed"):
            inner.invalidate_packed()
        return out

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model.model, name)


def get_peft_model(model: nn.Module, cfg: LoraConfig) -> PeftModel:
    """Freeze the base, wrap every nn.Linear whose name ends with a target.  Mutates `model` in place."""
    for p in model.parameters():
        p.requires_grad_(False)
    names = [n for n, m in model.named_modules() if isinstance(m, nn.Linear) and _match(n, cfg.target_modules)]
    for name in names:
        parent_name, _, leaf = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        wrapped = LoraLinear(getattr(parent, leaf), cfg)
        if leaf.isdigit():
            parent[int(leaf)] = wrapped
        else:
            setattr(parent, leaf, wrapped)
    if hasattr(model, "invalidate_packed"):
        model.invalidate_packed()
    return PeftModel(model, cfg)


def get_peft_model_state_dict(peft_model):
    return {k.replace(".default", ""): v for k, v in peft_model.state_dict().items() if "lora_" in k}


def convert_state_dict_to_diffusers(sd):
    return {k.replace(".lora_A.weight", ".lora.down.weight").replace(".lora_B.weight", ".lora.up.weight"): v
            for k, v in sd.items()}


def lora_parameters(model):
    return [p for n, p in model.named_parameters() if "lora_" in n]
