"""Oracle AutoencoderKL (test infrastructure; see oracle/__init__.py).

Restates diffusers==0.32.2 `AutoencoderKL.decode` (hot path, inside
`AudioLDMPipeline.__call__` [REF generate_audio.py:47-52]) and `.encode`
("next" row: [REF script/train/train_audioldm_lora.py:495-496]).
Spec: SURVEY.md Appendix B.5 / B.4; keys per Appendix A.5.  GroupNorm eps is
1e-6 everywhere in the VAE; resnets carry no time embedding; the mid-block
attention is single-head (d = 512) with biased q/k/v/out, its own GroupNorm and
an internal residual.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .configs import VAE
from .unet import ResnetBlock2D, Upsample2D

EPS = 1e-6


class VaeAttention(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=EPS)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(b, c, h, w) + res


class VaeMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, EPS) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])

    def forward(self, h):
        h = self.resnets[0](h)
        h = self.attentions[0](h)
        return self.resnets[1](h)


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, EPS) for i in range(layers)])
        self.has_up = upsample
        if upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.has_up:
            h = self.upsamplers[0](h)
        return h


class EncDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, EPS) for i in range(layers)])
        self.has_down = downsample
        if downsample:
            self.downsamplers = nn.ModuleList([EncDownsample(cout)])

    def forward(self, h):
        for r in self.resnets:
            h = r(h)
        if self.has_down:
            h = self.downsamplers[0](h)
        return h


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["latent_channels"], boc[-1], 3, padding=1)
        self.mid_block = VaeMid(boc[-1], g)
        rev = list(reversed(boc))
        ups, out_c = [], rev[0]
        for i in range(len(boc)):
            prev, out_c = out_c, rev[i]
            ups.append(UpDecoderBlock(prev, out_c, cfg["layers_per_block"] + 1, g, i != len(boc) - 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=EPS)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    def forward(self, z):
        h = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            h = u(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc, g = cfg["block_out_channels"], cfg["norm_num_groups"]
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        downs, out_c = [], boc[0]
        for i in range(len(boc)):
            in_c, out_c = out_c, boc[i]
            downs.append(DownEncoderBlock(in_c, out_c, cfg["layers_per_block"], g, i != len(boc) - 1))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = VaeMid(boc[-1], g)
        self.conv_norm_out = nn.GroupNorm(g, boc[-1], eps=EPS)
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg["latent_channels"], 3, padding=1)

    def forward(self, x):
        h = self.conv_in(x)
        for d in self.down_blocks:
            h = d(h)
        h = self.mid_block(h)
        return self.conv_out(F.silu(self.conv_norm_out(h)))


class DiagonalGaussian:
    def __init__(self, params):
        self.mean, logvar = params.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        return self.mean + self.std * torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, **over):
        super().__init__()
        cfg = dict(VAE)
        cfg.update(over)
        self.config = SimpleNamespace(**cfg)
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        lc = cfg["latent_channels"]
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussian(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))
