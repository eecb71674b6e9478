"""Oracle UNet2DConditionModel (test infrastructure; see oracle/__init__.py).

Restates diffusers==0.32.2 `UNet2DConditionModel.forward` for the AudioLDM
configuration, the call the reference makes at
  [REF script/train/train_audioldm_lora.py:539-546]  and, through
`AudioLDMPipeline.__call__`, at [REF script/inference/generate_audio.py:47-52].
Graph / arithmetic spec: SURVEY.md section 3.2 and Appendix B.2-B.4; module and
parameter names follow the diffusers state-dict scheme (Appendix A.5) so real
checkpoints would load with strict=True.

Layout is NCHW fp32 and every op is a torch CPU primitive (F.conv2d,
F.group_norm, F.scaled_dot_product_attention, F.layer_norm, F.gelu,
F.interpolate) -- the primitives diffusers itself composes.
"""
import math

import torch
import torch.nn.functional as F
from torch import nn

from .configs import UNET


def timestep_embedding(timesteps, dim, flip_sin_to_cos=True, freq_shift=0, max_period=10000):
    """diffusers get_timestep_embedding (Appendix B.2)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - freq_shift)
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_channels, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout) if temb_channels else None
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        if self.time_emb_proj is not None:
            h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """diffusers Attention + AttnProcessor2_0, no mask, no dropout."""

    def __init__(self, query_dim, heads, dim_head, cross_dim=None, bias=False, out_bias=True):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        cross_dim = cross_dim or query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross_dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])

    def forward(self, x, context=None):
        context = x if context is None else context
        b, n, _ = x.shape
        q, k, v = self.to_q(x), self.to_k(context), self.to_v(context)
        d = q.shape[-1] // self.heads
        q = q.view(b, -1, self.heads, d).transpose(1, 2)
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.transpose(1, 2).reshape(b, n, self.heads * d)
        return self.to_out[1](self.to_out[0](o))


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, h, encoder_hidden_states=None):
        h = self.attn1(self.norm1(h)) + h
        h = self.attn2(self.norm2(h), encoder_hidden_states) + h
        h = self.ff(self.norm3(h)) + h
        return h


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(channels, heads, channels // heads, cross_dim)])
        self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x, encoder_hidden_states=None):
        b, c, hh, ww = x.shape
        res = x
        h = self.proj_in(self.norm(x))
        h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
        for blk in self.transformer_blocks:
            h = blk(h, encoder_hidden_states)
        h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2).contiguous()
        return self.proj_out(h) + res


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x, output_size=None):
        if output_size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=output_size, mode="nearest")
        return self.conv(x)


class DownBlock(nn.Module):
    """DownBlock2D / CrossAttnDownBlock2D."""

    def __init__(self, cin, cout, temb, layers, groups, eps, heads, cross_dim, attn, downsample):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(layers)])
        self.has_attn = attn
        if downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_down = downsample

    def forward(self, h, temb):
        outs = []
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h)
            outs.append(h)
        if self.has_down:
            h = self.downsamplers[0](h)
            outs.append(h)
        return h, outs


class UpBlock(nn.Module):
    """UpBlock2D / CrossAttnUpBlock2D."""

    def __init__(self, cin, cout, cprev, temb, layers, groups, eps, heads, cross_dim, attn, upsample):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = cprev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(layers)])
        self.has_attn = attn
        if upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_up = upsample

    def forward(self, h, skips, temb, upsample_size=None):
        for i, r in enumerate(self.resnets):
            h = torch.cat([h, skips.pop()], dim=1)
            h = r(h, temb)
            if self.has_attn:
                h = self.attentions[i](h)
        if self.has_up:
            h = self.upsamplers[0](h, upsample_size)
        return h


class MidBlock(nn.Module):
    def __init__(self, c, temb, groups, eps, heads, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, cross_dim, groups)])

    def forward(self, h, temb):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h)
        return self.resnets[1](h, temb)


class UNet2DConditionModel(nn.Module):
    def __init__(self, **cfg_over):
        super().__init__()
        cfg = dict(UNET)
        cfg.update(cfg_over)
        self.cfg = cfg
        boc = cfg["block_out_channels"]
        groups, eps, heads = cfg["norm_num_groups"], cfg["norm_eps"], cfg["num_heads"]
        ted = boc[0] * 4
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.class_embedding = nn.Linear(cfg["class_embed_input_dim"], ted)
        temb = ted * 2 if cfg["class_embeddings_concat"] else ted
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)

        downs = []
        out_c = boc[0]
        for i, typ in enumerate(cfg["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            final = i == len(boc) - 1
            downs.append(DownBlock(in_c, out_c, temb, cfg["layers_per_block"], groups, eps, heads,
                                   cfg["cross_attention_dim"][i], typ.startswith("CrossAttn"), not final))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb, groups, eps, heads, cfg["cross_attention_dim"][-1])

        ups = []
        rev = list(reversed(boc))
        rev_cross = list(reversed(cfg["cross_attention_dim"]))
        out_c = rev[0]
        for i, typ in enumerate(cfg["up_block_types"]):
            prev = out_c
            out_c = rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            ups.append(UpBlock(in_c, out_c, prev, temb, cfg["layers_per_block"] + 1, groups, eps, heads,
                               rev_cross[i], typ.startswith("CrossAttn"), not final))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)
        self.num_upsamplers = len(boc) - 1

    def embed(self, sample, timestep, class_labels):
        cfg = self.cfg
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.int64, device=sample.device)
        elif t.dim() == 0:
            t = t[None].to(sample.device)
        t = t.expand(sample.shape[0])
        t_emb = timestep_embedding(t, cfg["block_out_channels"][0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
        emb = self.time_embedding(t_emb.to(sample.dtype))
        class_emb = self.class_embedding(class_labels).to(sample.dtype)
        return torch.cat([emb, class_emb], dim=-1) if cfg["class_embeddings_concat"] else emb + class_emb

    def forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None,
                cross_attention_kwargs=None, return_dict=False):
        assert encoder_hidden_states is None, "AudioLDM conditions through class_labels only"
        factor = 2 ** self.num_upsamplers
        forward_upsample_size = any(d % factor != 0 for d in sample.shape[-2:])
        emb = self.embed(sample, timestep, class_labels)
        h = self.conv_in(sample)
        skips = [h]
        for blk in self.down_blocks:
            h, outs = blk(h, emb)
            skips.extend(outs)
        h = self.mid_block(h, emb)
        for i, blk in enumerate(self.up_blocks):
            final = i == len(self.up_blocks) - 1
            n = len(blk.resnets)
            mine, skips = skips[-n:], skips[:-n]
            size = skips[-1].shape[2:] if (not final and forward_upsample_size) else None
            h = blk(h, mine, emb, size)
        h = self.conv_out(F.silu(self.conv_norm_out(h)))
        return (h,)
