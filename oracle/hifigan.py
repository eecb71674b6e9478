"""Oracle HiFi-GAN vocoder (test infrastructure; see oracle/__init__.py).

Restates `transformers.SpeechT5HifiGan.forward` -- loaded by the reference at
[REF script/train/train_audioldm_lora.py:371] and run inside
`AudioLDMPipeline.__call__` [REF generate_audio.py:47-52].  Spec: SURVEY.md
Appendix B.6 (from transformers/models/speecht5/modeling_speecht5.py:2887-3069).
PINNED: tests compare this module against the importable transformers class on
identical weights (tests/test_oracle_vocoder.py).

Written functionally over a flat state dict so the same key names
(`conv_pre`, `upsampler.{i}`, `resblocks.{i}.convs{1,2}.{j}`, `conv_post`,
`mean`, `scale`) are shared with the product path.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .configs import VOCODER


class HifiGanResidualBlock(nn.Module):
    def __init__(self, ch, k, dil, slope):
        super().__init__()
        self.slope = slope
        self.convs1 = nn.ModuleList(
            [nn.Conv1d(ch, ch, k, dilation=d, padding=(k * d - d) // 2) for d in dil])
        self.convs2 = nn.ModuleList(
            [nn.Conv1d(ch, ch, k, dilation=1, padding=(k - 1) // 2) for _ in dil])

    def forward(self, h):
        for c1, c2 in zip(self.convs1, self.convs2):
            r = h
            h = c1(F.leaky_relu(h, self.slope))
            h = c2(F.leaky_relu(h, self.slope))
            h = h + r
        return h


class SpeechT5HifiGan(nn.Module):
    def __init__(self, **over):
        super().__init__()
        cfg = dict(VOCODER)
        cfg.update(over)
        self.config = SimpleNamespace(**cfg)
        c0 = cfg["upsample_initial_channel"]
        self.conv_pre = nn.Conv1d(cfg["model_in_dim"], c0, 7, padding=3)
        self.upsampler = nn.ModuleList()
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            self.upsampler.append(
                nn.ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, stride=u, padding=(k - u) // 2))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.upsampler)):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]):
                self.resblocks.append(HifiGanResidualBlock(ch, k, d, cfg["leaky_relu_slope"]))
        self.conv_post = nn.Conv1d(ch, 1, 7, padding=3)
        self.register_buffer("mean", torch.zeros(cfg["model_in_dim"]))
        self.register_buffer("scale", torch.ones(cfg["model_in_dim"]))
        self.num_kernels = len(cfg["resblock_kernel_sizes"])

    def forward(self, spectrogram):
        cfg = self.config
        if cfg.normalize_before:
            spectrogram = (spectrogram - self.mean) / self.scale
        h = self.conv_pre(spectrogram.transpose(2, 1))
        for i, up in enumerate(self.upsampler):
            h = up(F.leaky_relu(h, cfg.leaky_relu_slope))
            acc = self.resblocks[i * self.num_kernels](h)
            for j in range(1, self.num_kernels):
                acc = acc + self.resblocks[i * self.num_kernels + j](h)
            h = acc / self.num_kernels
        h = F.leaky_relu(h)                     # torch default slope 0.01 (NOT 0.1) -- modeling_speecht5.py:3058
        h = torch.tanh(self.conv_post(h))
        return h.squeeze(1)
