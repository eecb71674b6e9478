"""AutoencoderKL on the MI355X HIP path (decode = hot path, encode = the "next" row before training).

Drop-in for the diffusers class: `AutoencoderKL.from_pretrained(id, subfolder="vae")`, `.decode(z).sample`,
`.encode(x).latent_dist.sample()`, `.config.scaling_factor`
  [REF script/train/train_audioldm_lora.py:370,495-496]; decode runs inside AudioLDMPipeline.__call__
  [REF script/inference/generate_audio.py:47-52].  Spec: SURVEY.md B.5 / B.4; keys per A.5.

Launch sequence (channels-last bf16): the resnets / up-samplers reuse the UNet's GroupNorm and implicit-GEMM
kernels (nearest x2 folded into the conv gather).  The mid-block attention is single-head with d = 512 over
N = H*W = 4000 tokens: it runs as three GEMMs through the same MFMA kernel -- S = Q K^T (K as the "weight"
matrix), row softmax, O = P V (V^T token-major straight out of the projection GEMM's transposed store).
"""
import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from ._lib import ACT_NONE, ACT_SILU
from .configs import VAE
from .unet import ResnetBlock2D, Upsample2D, _f32, pack_resnet, run_resnet

EPS = 1e-6


class VaeAttention(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=EPS)
        self.to_q = nn.Linear(c, c)
        self.to_k = nn.Linear(c, c)
        self.to_v = nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])
        self.groups = groups


class VaeMid(nn.Module):
    def __init__(self, c, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, None, groups, EPS) for _ in range(2)])
        self.attentions = nn.ModuleList([VaeAttention(c, groups)])


class UpDecoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, EPS) for i in range(layers)])
        if upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_up = upsample


class EncDownsample(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)


class DownEncoderBlock(nn.Module):
    def __init__(self, cin, cout, layers, groups, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, None, groups, EPS) for i in range(layers)])
        if downsample:
            self.downsamplers = nn.ModuleList([EncDownsample(cout)])
        self.has_down = downsample


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
        self.groups = g


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
        self.groups = g


def pack_vae_attention(a: VaeAttention):
    c = a.to_q.weight.shape[0]
    # fused q | k | v projection for the flash kernel (ops.attention_wide): the softmax scale C^-0.5 (and log2 e) rides in to_q
    qs = ops.LOG2E / (c ** 0.5)
    wqkv = torch.cat([a.to_q.weight.detach().float() * qs, a.to_k.weight.detach().float(), a.to_v.weight.detach().float()])
    bqkv = torch.cat([a.to_q.bias.detach().float() * qs, a.to_k.bias.detach().float(), a.to_v.bias.detach().float()])
    return SimpleNamespace(
        g=_f32(a.group_norm.weight), b=_f32(a.group_norm.bias), groups=a.groups,
        out=ops.pack_linear(a.to_out[0].weight, a.to_out[0].bias),
        qkv=ops.pack_linear(wqkv, bqkv) if c in ops.WIDE_HEAD_DIMS else None, c=c)


def run_vae_attention(P, x):
    B, H, W, C = x.shape
    N = H * W
    hn = ops.groupnorm(x, P.g, P.b, P.groups, EPS, ACT_NONE).view(B * N, C)
    if P.qkv is not None:
        # diffusers Attention of the VAE mid block (1 head, d = C): ONE fused projection GEMM (Q | K row-major, V^T token-major
        # straight from the epilogue) -> flash attention over the wide head -> out-projection + residual.  The N x N score
        # matrix (64 MB fp32 per 10 s clip) never exists.
        npad = (N + 31) // 32 * 32
        vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device=x.device)       # keys beyond N must read as zeros
        qk = ops.conv(hn.view(B, 1, N, C), P.qkv, vt=vt, vt_col0=2 * C, vt_ld=npad, vt_batch_stride=C * npad)
        o = ops.attention_wide(qk.view(B * N, 2 * C), vt, B, N, C)
        return ops.conv(o.view(B, H, W, C), P.out, res=x, qstats=True)
    raise ops._lib.AldmError(f"AutoencoderKL mid-block attention: head width {C} is not one the flash kernel is built for "
                             f"{ops.WIDE_HEAD_DIMS} (cvssp/audioldm-s-full-v2 uses 512); there is no slower fallback")


class DiagonalGaussian:
    """diffusers' DiagonalGaussianDistribution over encoder moments [B, 2C, H, W] (mean | logvar).  `sample()` is one HIP kernel
    (clamp, exp, multiply-add); `mean` / `mode()` are views; `logvar` / `std` exist for inspection and are derived lazily."""

    def __init__(self, params_nchw):
        self.parameters = params_nchw
        self.mean = params_nchw.chunk(2, dim=1)[0]

    @property
    def logvar(self):
        return self.parameters.chunk(2, dim=1)[1].clamp(-30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, dtype=torch.float32, device=self.mean.device)
        return ops.gaussian_sample(self.parameters, noise).to(self.mean.dtype)

    def mode(self):
        return self.mean


class AutoencoderKL(nn.Module):
    def __init__(self, **over):
        super().__init__()
        cfg = dict(VAE)
        cfg.update({k: v for k, v in over.items() if k in VAE})
        self.cfg = cfg
        self.config = SimpleNamespace(**cfg)
        self.encoder = Encoder(cfg)
        self.decoder = Decoder(cfg)
        lc = cfg["latent_channels"]
        self.quant_conv = nn.Conv2d(2 * lc, 2 * lc, 1)
        self.post_quant_conv = nn.Conv2d(lc, lc, 1)
        self._plan = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        f = os.path.join(d, "config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f} not found: hub downloads are unavailable, pass a local directory")
        raw = json.load(open(f))
        m = cls(in_channels=raw["in_channels"], out_channels=raw["out_channels"], latent_channels=raw["latent_channels"],
                block_out_channels=tuple(raw["block_out_channels"]), layers_per_block=raw["layers_per_block"],
                norm_num_groups=raw["norm_num_groups"], scaling_factor=raw["scaling_factor"])
        from safetensors.torch import load_file
        m.load_state_dict(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")), strict=True)
        return m

    def _apply(self, fn, *a, **k):
        self._plan = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._plan = None
        return super().load_state_dict(*a, **k)

    def plan(self):
        if self._plan is not None:
            return self._plan
        if self.post_quant_conv.weight.device.type != "cuda":
            raise ops._lib.AldmError("AutoencoderKL runs on the MI355X only: call .to('cuda') first (no CPU fallback)")
        pc = lambda m: ops.pack_conv(m.weight, m.bias)
        d, e = self.decoder, self.encoder
        P = SimpleNamespace()
        P.post_quant = pc(self.post_quant_conv)
        P.quant = pc(self.quant_conv)
        P.dec = SimpleNamespace(
            conv_in=pc(d.conv_in), mid_res=[pack_resnet(r) for r in d.mid_block.resnets],
            mid_attn=pack_vae_attention(d.mid_block.attentions[0]),
            ups=[SimpleNamespace(resnets=[pack_resnet(r) for r in u.resnets],
                                 up=pc(u.upsamplers[0].conv) if u.has_up else None) for u in d.up_blocks],
            gn=(_f32(d.conv_norm_out.weight), _f32(d.conv_norm_out.bias)), conv_out=pc(d.conv_out), groups=d.groups)
        P.enc = SimpleNamespace(
            conv_in=pc(e.conv_in), mid_res=[pack_resnet(r) for r in e.mid_block.resnets],
            mid_attn=pack_vae_attention(e.mid_block.attentions[0]),
            downs=[SimpleNamespace(resnets=[pack_resnet(r) for r in u.resnets],
                                   down=pc(u.downsamplers[0].conv) if u.has_down else None) for u in e.down_blocks],
            gn=(_f32(e.conv_norm_out.weight), _f32(e.conv_norm_out.bias)), conv_out=pc(e.conv_out), groups=e.groups)
        self._plan = P
        return P

    # ---- decode: latent [B, h, w, 8] channels-last bf16 -> mel [B, 4h, 4w, 1] fp32 ----
    def decode_nhwc(self, z):
        P = self.plan()
        D = P.dec
        h = ops.conv(z, P.post_quant)
        h = ops.conv(h, D.conv_in, pad=(1, 1), qstats=True)
        h = run_resnet(D.mid_res[0], h)
        h = run_vae_attention(D.mid_attn, h)
        h = run_resnet(D.mid_res[1], h)
        for u in D.ups:
            for r in u.resnets:
                h = run_resnet(r, h)
            if u.up is not None:
                h = ops.conv(h, u.up, pad=(1, 1), up_size=(h.shape[1] * 2, h.shape[2] * 2), qstats=True)
        h = ops.groupnorm(h, D.gn[0], D.gn[1], D.groups, EPS, ACT_SILU)
        return ops.conv(h, D.conv_out, pad=(1, 1), out_f32=True)

    def decode(self, z, return_dict=True, **kw):
        """z: NCHW latents (already divided by scaling_factor by the caller, as in diffusers)."""
        if not z.is_cuda:
            raise ops._lib.AldmError("AutoencoderKL.decode needs CUDA/HIP tensors (no CPU fallback)")
        mel = self.decode_nhwc(ops.nchw_to_nhwc(z.float()))
        out = ops.nhwc_to_nchw_f32(mel).to(z.dtype)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    # ---- encode: mel [B, H, W, 1] -> moments [B, H/4, W/4, 16] fp32 ----
    def encode_nhwc(self, x):
        P = self.plan()
        E = P.enc
        B, H, W, C = x.shape
        if C % 8:                                            # the gather works on 16-byte channel chunks
            xp = torch.zeros(B, H, W, 8, dtype=torch.bfloat16, device=x.device)
            xp[..., :C] = x
            x = xp
            if not hasattr(E, "conv_in8"):
                w = self.encoder.conv_in.weight
                w8 = torch.zeros(w.shape[0], 8, 3, 3, device=w.device, dtype=w.dtype)
                w8[:, :C] = w
                E.conv_in8 = ops.pack_conv(w8, self.encoder.conv_in.bias)
            h = ops.conv(x, E.conv_in8, pad=(1, 1), qstats=True)
        else:
            h = ops.conv(x, E.conv_in, pad=(1, 1), qstats=True)
        for d in E.downs:
            for r in d.resnets:
                h = run_resnet(r, h)
            if d.down is not None:                           # F.pad(x, (0,1,0,1)) + stride-2 conv, pad folded into bounds
                oh, ow = h.shape[1] // 2, h.shape[2] // 2
                h = ops.conv(h, d.down, stride=(2, 2), pad=(0, 0), out_hw=(oh, ow), qstats=True)
        h = run_resnet(E.mid_res[0], h)
        h = run_vae_attention(E.mid_attn, h)
        h = run_resnet(E.mid_res[1], h)
        h = ops.groupnorm(h, E.gn[0], E.gn[1], E.groups, EPS, ACT_SILU)
        h = ops.conv(h, E.conv_out, pad=(1, 1))
        return ops.conv(h, P.quant, out_f32=True)

    def encode(self, x, return_dict=True, **kw):
        if not x.is_cuda:
            raise ops._lib.AldmError("AutoencoderKL.encode needs CUDA/HIP tensors (no CPU fallback)")
        mom = ops.nhwc_to_nchw_f32(self.encode_nhwc(ops.nchw_to_nhwc(x.float())))
        return SimpleNamespace(latent_dist=DiagonalGaussian(mom.to(x.dtype)))
