"""UNet2DConditionModel on the MI355X HIP path.

Drop-in for the diffusers class the reference calls at
  [REF script/train/train_audioldm_lora.py:364,374,539-546] and, through AudioLDMPipeline.__call__,
  [REF script/inference/generate_audio.py:18,42,47-52] / [REF app.py:14]:
same constructor config, same module / state-dict names, same
`forward(sample, timestep, encoder_hidden_states, class_labels=, cross_attention_kwargs=, return_dict=)`.

The nn.Module tree below only HOLDS parameters (so `state_dict`, `load_state_dict`, `requires_grad_`,
peft-style injection by module name all behave as in diffusers).  The arithmetic is a launch sequence of
hand-written gfx950 kernels over channels-last bf16 activations:
  ResnetBlock2D      = groupnorm(+SiLU) -> igemm 3x3 (+bias +time-emb row bias) -> groupnorm(+SiLU)
                       -> [igemm 1x1 shortcut over the virtual concat] -> igemm 3x3 (+residual)
  Transformer2DModel = groupnorm -> igemm 1x1 -> {layernorm -> fused QKV igemm with LoRA side channel
                       (Q|K row-major, V^T token-major) -> flash attention -> out-proj igemm (+LoRA +bias
                       +residual)} x2 -> layernorm -> GEGLU igemm -> igemm (+residual) -> igemm 1x1 (+residual)
Skip connections are never concatenated in memory: GroupNorm and the GEMM gather read both sources.
All 22 time_emb_proj layers run as ONE GEMM per forward (their input is the same silu(emb) vector).
"""
import json
import math
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from ._lib import ACT_NONE, ACT_SILU
from .configs import UNET
from .lora import LoraLinear


# ----------------------------------------------------------------------------------------------
# parameter containers (diffusers names)
# ----------------------------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        if temb:
            self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        if cin != cout:
            self.conv_shortcut = nn.Conv2d(cin, cout, 1)
        self.cin, self.cout, self.groups, self.eps = cin, cout, groups, eps


class Attention(nn.Module):
    def __init__(self, query_dim, heads, dim_head, cross_dim=None, bias=False, out_bias=True):
        super().__init__()
        inner = heads * dim_head
        cross_dim = cross_dim or query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(cross_dim, inner, bias=bias)
        self.to_v = nn.Linear(cross_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=out_bias), nn.Dropout(0.0)])


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, heads, dim_head, cross_dim=cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, cross_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6)
        self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, channels // heads, cross_dim)])
        self.proj_out = nn.Conv2d(channels, channels, 1)
        self.channels, self.heads, self.groups = channels, heads, groups


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, groups, eps, heads, cross_dim, attn, downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups, eps) for i in range(layers)])
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(layers)])
        if downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])
        self.has_attn, self.has_down = attn, downsample


class UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, layers, groups, eps, heads, cross_dim, attn, upsample):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = cprev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb, groups, eps))
        self.resnets = nn.ModuleList(rs)
        if attn:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cross_dim, groups) for _ in range(layers)])
        if upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])
        self.has_attn, self.has_up = attn, upsample


class MidBlock(nn.Module):
    def __init__(self, c, temb, groups, eps, heads, cross_dim):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, groups, eps) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, cross_dim, groups)])


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)


# ----------------------------------------------------------------------------------------------
# packing helpers
# ----------------------------------------------------------------------------------------------
def _f32(t):
    return t.detach().float().contiguous()


def _lin(mod):
    """(weight, bias, lora) of an nn.Linear or a LoraLinear wrapper."""
    if isinstance(mod, LoraLinear):
        b = mod.base_layer
        return b.weight, b.bias, (mod.lora_A["default"].weight, mod.lora_B["default"].weight, mod.scaling)
    return mod.weight, mod.bias, None


def pack_resnet(r: ResnetBlock2D):
    sc = r.conv_shortcut if hasattr(r, "conv_shortcut") else None
    # conv_shortcut(input) + conv2(h) as ONE implicit GEMM (ops.pack_conv_shortcut) wherever the LDS-DMA path can take it
    fuse = sc is not None and r.cout % 64 == 0 and r.cin % 64 == 0
    return SimpleNamespace(
        g1=_f32(r.norm1.weight), b1=_f32(r.norm1.bias), g2=_f32(r.norm2.weight), b2=_f32(r.norm2.bias),
        conv1=ops.pack_conv(r.conv1.weight, r.conv1.bias), conv2=ops.pack_conv(r.conv2.weight, r.conv2.bias),
        shortcut=ops.pack_conv(sc.weight, sc.bias) if sc is not None else None,
        conv2s=ops.pack_conv_shortcut(r.conv2.weight, r.conv2.bias, sc.weight, sc.bias) if fuse else None,
        groups=r.groups, eps=r.eps, cout=r.cout, temb_off=0)


# LayerNorm folded into the consuming GEMM (ops.pack_linear_ln).  With the row statistics accumulated inside the consumer's K loop
# the fold only paid at C = 640 (measured in a replayed graph, tools/bench_qkv_graph.py: -0.7 / -1.5 us there, +2.7-3.5 us on the
# C = 384 QKV and C = 256 GEGLU GEMMs: every one of the consumer's N-tiles re-derives the same statistics).  Now the PRODUCER of the
# hidden state hands them over (ops.conv rowstats= / ln_parts=): its epilogue writes per-row partial sums, the consumer's prologue
# folds them -- neither GEMM's K loop pays, and all 48 LayerNorm launches of a UNet forward are gone at every width.
LN_FOLD_MIN_C = 64


def pack_attention(a: Attention, ln=None):
    """ln = (gamma, beta) of the LayerNorm feeding this attention: folded into the fused QKV GEMM when the LDS-DMA
    path can take it (C % 64 == 0); the caller then passes the RAW hidden state."""
    wq, bq, lq = _lin(a.to_q)
    wk, bk, lk = _lin(a.to_k)
    wv, bv, lv = _lin(a.to_v)
    wo, bo, lo = _lin(a.to_out[0])
    c = wq.shape[0]
    # the softmax scale d^-0.5 (and log2 e: the kernel exponentiates with v_exp_f32 = 2^x) rides in the Q projection: W_q, its
    # bias and the LoRA-B rows of to_q are multiplied once, here, BEFORE their bf16 rounding (the rounding error of w * s is the
    # rounding error of w), so the attention kernel never multiplies a score (ops.attention prescaled=True)
    qs = ops.LOG2E / math.sqrt(a.dim_head)
    wq = wq.detach().float() * qs
    bq = None if bq is None else bq.detach().float() * qs
    lq = None if lq is None else (lq[0], lq[1], lq[2] * qs)
    bias = None
    if bq is not None:
        bias = torch.cat([bq, bk.detach().float(), bv.detach().float()])
    fold = ln is not None and c % 64 == 0 and c >= LN_FOLD_MIN_C
    wqkv = torch.cat([wq, wk.detach().float(), wv.detach().float()])
    qkv = ops.pack_linear_ln(wqkv, bias, ln[0], ln[1]) if fold else ops.pack_linear(wqkv, bias)
    ops.attach_lora(qkv, [None if l is None else (i * c, c, l[0], l[1], l[2]) for i, l in enumerate((lq, lk, lv))])
    out = ops.pack_linear(wo, bo)
    ops.attach_lora(out, [None if lo is None else (0, wo.shape[0], lo[0], lo[1], lo[2])])
    return SimpleNamespace(qkv=qkv, out=out, heads=a.heads, d=a.dim_head, c=c, ln_folded=fold)


def _merge_ff2_proj_out(t: Transformer2DModel):
    """FeedForward's output linear and Transformer2DModel.proj_out are two linear maps with nothing but a residual add between
    them:  proj_out(ff2(g) + h) + x  =  [g | h] . [Wp W2 | Wp]^T + (Wp b2 + bp) + x.  One GEMM over the virtual concat [g | h]
    (K = 4C + C: the same FLOPs as the two it replaces) saves a launch and the round trip of the block output through HBM.
    Both layers are frozen base weights (LoRA wraps the attention projections only), so the merge is exact up to bf16 rounding
    of the merged weight."""
    blk = t.transformer_blocks[0]
    w2, b2 = blk.ff.net[2].weight.detach().float(), blk.ff.net[2].bias.detach().float()
    wp = t.proj_out.weight.detach().float().flatten(1)                     # [C, C] (1x1 conv)
    bp = t.proj_out.bias.detach().float()
    wm = torch.cat([wp @ w2, wp], dim=1)                                    # [C, 4C + C]
    return ops.pack_conv(wm[:, :, None, None].contiguous(), (wp @ b2 + bp).contiguous())


def pack_transformer(t: Transformer2DModel):
    blk = t.transformer_blocks[0]
    merged = (t.channels % 64 == 0 and not isinstance(blk.ff.net[2], LoraLinear) and not isinstance(t.proj_out, LoraLinear))
    return SimpleNamespace(
        ff2_proj=_merge_ff2_proj_out(t) if merged else None,
        gn_g=_f32(t.norm.weight), gn_b=_f32(t.norm.bias), groups=t.groups,
        proj_in=ops.pack_conv(t.proj_in.weight, t.proj_in.bias), proj_out=ops.pack_conv(t.proj_out.weight, t.proj_out.bias),
        ln=[(_f32(n.weight), _f32(n.bias)) for n in (blk.norm1, blk.norm2, blk.norm3)],
        attn1=pack_attention(blk.attn1, (blk.norm1.weight, blk.norm1.bias)),
        attn2=pack_attention(blk.attn2, (blk.norm2.weight, blk.norm2.bias)),
        ff1=(ops.pack_linear_ln(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias, blk.norm3.weight, blk.norm3.bias, geglu=True)
             if (t.channels % 64 == 0 and t.channels >= LN_FOLD_MIN_C) else ops.pack_geglu(blk.ff.net[0].proj.weight, blk.ff.net[0].proj.bias)),
        ff2=ops.pack_linear(blk.ff.net[2].weight, blk.ff.net[2].bias))


# ----------------------------------------------------------------------------------------------
# launch sequences
# ----------------------------------------------------------------------------------------------
GN_IN = os.environ.get("ALDM_NO_GN_IN") is None    # GroupNorm of a convolution's input inside its halo tile (ops.conv gn_in=); the switch is an A/B aid


def _gn_in_pays(x, x2, pw):
    """Policy on top of ops.gn_in_ok (capability): the fold forces a halo tile, which is the right tile for the UNet's 4000-pixel level
    (128 output channels, 128 - 384 input channels: -3 .. -12 us per convolution) but not for the VAE's 512 -> 512 convolutions at
    250 x 16 (the 128x128 tile runs them at 898 TFLOP/s, the halo tile at 574: +47 us per launch against 23 us of apply pass saved)."""
    if not (GN_IN and pw.N <= 128 and pw.Cin <= 384 and ops.gn_in_ok(x, x2, pw)):
        return False
    # every workgroup sums its image's producer tiles in its prologue: fine for the latent's 32 - 63 tiles, ruinous for a VAE mel image
    # of 512 (measured: 630 us per convolution against 217 + 81 for convolution + apply pass) -- same bound as aldm_groupnorm_apply's
    # finalize launch
    hw = x.shape[1] * x.shape[2]
    return all((q.tpi if q.tpi > 0 else hw // q.bm + 1) < ops.GN_FINALIZE_MIN_TILES for q in (x.qstats, x2.qstats if x2 is not None else None) if q is not None)


def run_resnet(P, x, x2=None, rowbias=None, rowbias_ld=0, next_gn=None, defer=None):
    """ResnetBlock2D over x (| x2).  next_gn = (gamma, beta, groups, eps, act) of a GroupNorm that consumes the block output
    (the Transformer2DModel behind it): returns (output, its GroupNorm), the norm fused with conv2's split-K reduce.
    x may be an ops.Deferred (producer's split-K reduce pending): norm1 performs it.  defer = (channels, groups) of the norm
    that will consume THIS block's output next: conv2 may then return an ops.Deferred in turn."""
    rb = rowbias[:, P.temb_off:] if rowbias is not None else None
    gn2 = (P.g2, P.b2, P.groups, P.eps, ACT_SILU)
    h1 = None                                  # conv1's RAW output when norm2 is left to conv2 (statistics attached)
    if _gn_in_pays(x, x2, P.conv1):
        # the 4000-pixel level: x (| x2) are raw convolution outputs with their statistics tables -- norm1 + SiLU happen inside conv1's
        # halo tile (no groupnorm_apply launch, the normalised tensor never exists in HBM); conv1 leaves ITS statistics for norm2
        h1 = ops.conv(x, P.conv1, x2=x2, pad=(1, 1), rowbias=rb, rowbias_ld=rowbias_ld, qstats=True,
                      gn_in=(P.g1, P.b1, P.groups, P.eps, ACT_SILU))
        if not (P.conv2s is None and getattr(h1, "qstats", None) is not None and _gn_in_pays(h1, None, P.conv2)):
            h, h1 = ops.groupnorm(h1, *gn2), None          # conv2 carries the fused shortcut segment (no halo tile): norm2 as a launch
    else:
        h = ops.groupnorm(x, P.g1, P.b1, P.groups, P.eps, ACT_SILU, x2=x2)
        x = ops.tensor_of(x)
        # conv1 -> norm2 -> SiLU; a split-K conv1 leaves its partial tiles to the GroupNorm kernel (no reduce launch)
        # (qstats=True: a launch that does not split K leaves GroupNorm statistics next to its output, so the norm that consumes it --
        #  here norm2, below the next block's norm1 / the Transformer2DModel's norm -- is one coalesced apply pass)
        h = ops.conv(h, P.conv1, pad=(1, 1), rowbias=rb, rowbias_ld=rowbias_ld, gn=gn2, qstats=True)
    x = ops.tensor_of(x)
    if P.conv2s is not None and (x2 is None or x2.shape[3] % 64 == 0) and x.shape[3] % 64 == 0:
        # the 1x1 shortcut over the block input (| skip) rides at the end of conv2's K loop: one launch instead of two
        if next_gn is not None:
            return ops.conv(h, P.conv2s, pad=(1, 1), x3=x, x4=x2, gn=next_gn, gn_keep=True, qstats=True)
        return ops.conv(h, P.conv2s, pad=(1, 1), x3=x, x4=x2, defer=(defer or False), qstats=True)
    if P.shortcut is not None:
        xs = ops.conv(x, P.shortcut, x2=x2)
    else:
        assert x2 is None
        xs = x
    src, fold = (h1, dict(gn_in=gn2)) if h1 is not None else (h, {})      # (norm2 + SiLU inside conv2's halo tile)
    if next_gn is not None:
        return ops.conv(src, P.conv2, pad=(1, 1), res=xs, gn=next_gn, gn_keep=True, qstats=True, **fold)
    return ops.conv(src, P.conv2, pad=(1, 1), res=xs, defer=(defer or False), qstats=True, **fold)


def run_attention(P, hn, h_res, B, N, fp8=False, ln_parts=None, rowstats=False):
    """hn = LayerNorm(h) [B*N, C] (or h itself when the norm is folded into the QKV GEMM; ln_parts = its row statistics from the
    producer of h); returns h_res + to_out(attention(q, k, v)) with LoRA fused in both GEMMs -- and, with rowstats, the row
    statistics of that sum for the LayerNorm that follows."""
    C = P.c
    ops.SITE = f"attn C{C} N{N}"                      # bench.py prices these launches as ONE fused-LoRA attention module (K1)
    kind = ops.attn_block_ok(P.qkv, N, P.heads, P.d, ln_parts)
    if kind == 64 or (kind and not fp8):
        # the 64- and 252-token levels: projection (+ LoRA, folded LayerNorm) and attention of a (sample, head) in ONE launch; Q | K | V stay in
        # LDS (config 5's e4m3 operands: the 64-token launch has an fp8 form, the 252-token one stays on the two launches)
        a = ops.attn_block(hn, P.qkv, ln_parts, B, N, P.heads, P.d, fp8=fp8)
        y = ops.linear(a, P.out, res=h_res, rowstats=rowstats)
        ops.SITE = None
        return y
    npad = (N + 7) // 8 * 8
    vt = torch.empty(B, C, npad, dtype=torch.bfloat16, device=hn.device)
    qk = ops.conv(hn.view(B, 1, N, C), P.qkv, vt=vt, vt_col0=2 * C, vt_ld=npad, vt_batch_stride=C * npad, ln_parts=ln_parts)
    a = ops.attention(qk.view(B * N, 2 * C), vt, B, N, P.heads, P.d, fp8=fp8, prescaled=True)
    y = ops.linear(a, P.out, res=h_res, rowstats=rowstats)
    ops.SITE = None
    return y


def transformer_gn(P):
    return (P.gn_g, P.gn_b, P.groups, 1e-6, ACT_NONE)


def run_transformer(P, x, fp8=False, xn=None, defer=None):
    """Transformer2DModel over x; xn = its GroupNorm when the producer already computed it (run_resnet next_gn); defer as in
    run_resnet (the last GEMM may leave its split-K reduce to the next block's norm1)."""
    B, H, W, C = x.shape
    N = H * W
    h = xn if xn is not None else ops.groupnorm(x, *transformer_gn(P))
    # The three LayerNorms are folded into the GEMMs that consume them; each producer of the hidden state (proj_in, the two
    # out-projections) hands the row statistics over (st): no LayerNorm launch, no statistics pass inside a K loop.
    f1, f2, f3 = P.attn1.ln_folded, P.attn2.ln_folded, P.ff1.ln_s is not None
    st = None
    h = ops.conv(h, P.proj_in, rowstats=f1)
    if f1:
        h, st = h
    h = h.view(B * N, C)
    h = run_attention(P.attn1, h if f1 else ops.layernorm(h, *P.ln[0]), h, B, N, fp8, ln_parts=st, rowstats=f2)
    st = None
    if f2:
        h, st = h
    h = run_attention(P.attn2, h if f2 else ops.layernorm(h, *P.ln[1]), h, B, N, fp8, ln_parts=st, rowstats=f3)   # encoder_hidden_states=None: self-attention
    st = None
    if f3:
        h, st = h
    g = ops.linear(h if f3 else ops.layernorm(h, *P.ln[2]), P.ff1, ln_parts=st)
    if P.ff2_proj is not None:                                # ff2 and proj_out as ONE GEMM over the virtual concat [g | h]
        return ops.conv(g.view(B, H, W, 4 * C), P.ff2_proj, x2=h.view(B, H, W, C), res=x, defer=(defer or False), qstats=True)
    h = ops.linear(g, P.ff2, res=h)
    return ops.conv(h.view(B, H, W, C), P.proj_out, res=x, defer=(defer or False), qstats=True)


# ----------------------------------------------------------------------------------------------
class UNet2DConditionModel(nn.Module):
    config_name = "config.json"

    def __init__(self, **cfg_over):
        super().__init__()
        cfg = dict(UNET)
        cfg.update({k: v for k, v in cfg_over.items() if k in UNET})
        self.cfg = cfg
        self.config = SimpleNamespace(**cfg)
        boc = cfg["block_out_channels"]
        groups, eps, heads = cfg["norm_num_groups"], cfg["norm_eps"], cfg["num_heads"]
        assert cfg["class_embeddings_concat"], "AudioLDM concatenates the class embedding"
        ted = boc[0] * 4
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.class_embedding = nn.Linear(cfg["class_embed_input_dim"], ted)
        temb = ted * 2
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        downs, out_c = [], boc[0]
        for i, typ in enumerate(cfg["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            downs.append(DownBlock(in_c, out_c, temb, cfg["layers_per_block"], groups, eps, heads,
                                   cfg["cross_attention_dim"][i], typ.startswith("CrossAttn"), i != len(boc) - 1))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = MidBlock(boc[-1], temb, groups, eps, heads, cfg["cross_attention_dim"][-1])
        ups, rev, rev_cross = [], list(reversed(boc)), list(reversed(cfg["cross_attention_dim"]))
        out_c = rev[0]
        for i, typ in enumerate(cfg["up_block_types"]):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            ups.append(UpBlock(in_c, out_c, prev, temb, cfg["layers_per_block"] + 1, groups, eps, heads, rev_cross[i],
                               typ.startswith("CrossAttn"), i != len(boc) - 1))
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=eps)
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)
        self.num_upsamplers = len(boc) - 1
        self._plan = None
        self._plan_version = 0        # bumped whenever the packed operands may be stale (see invalidate_packed)
        self._weights_version = 0     # bumped when the FROZEN weights may have changed (state load, .to()): training.trainer_of

    # ---- drop-in plumbing ----
    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kw):
        """Loads a local diffusers-format directory (config.json + diffusion_pytorch_model.safetensors)."""
        d = os.path.join(path, subfolder) if subfolder else path
        cfg_path = os.path.join(d, cls.config_name)
        if not os.path.isfile(cfg_path):
            raise FileNotFoundError(f"{cfg_path} not found: hub downloads are unavailable, pass a local directory")
        raw = json.load(open(cfg_path))
        over = dict(in_channels=raw["in_channels"], out_channels=raw["out_channels"],
                    block_out_channels=tuple(raw["block_out_channels"]), layers_per_block=raw["layers_per_block"],
                    num_heads=raw.get("num_attention_heads") or raw["attention_head_dim"],
                    cross_attention_dim=tuple(raw["cross_attention_dim"]),
                    class_embed_input_dim=raw["projection_class_embeddings_input_dim"],
                    norm_num_groups=raw["norm_num_groups"], norm_eps=raw["norm_eps"],
                    down_block_types=tuple(raw["down_block_types"]), up_block_types=tuple(raw["up_block_types"]),
                    sample_size=raw.get("sample_size", 128))
        m = cls(**over)
        from safetensors.torch import load_file
        m.load_state_dict(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")), strict=True)
        return m

    def _has_trainable_lora(self):
        return any(isinstance(m, LoraLinear) and m.lora_A["default"].weight.requires_grad for m in self.modules())

    def invalidate_packed(self):
        """Forget the packed bf16 operands: call after ANY change of parameter values the module hooks cannot see (in-place
        edits of `.data`).  State loads, `.to()`, peft injection, PeftModel.load_state_dict and LoraTrainer steps call it
        themselves.  The version counter lets holders of captured graphs (engine.DenoiseEngine) notice that the plan they
        captured pointers into is no longer the live one."""
        self._plan = None
        self._plan_version += 1

    @property
    def plan_version(self):
        return self._plan_version

    def _apply(self, fn, *a, **k):
        # (a training engine attached to this UNet survives a no-op .to(): training.trainer_of() checks that the LoRA
        # parameters still alias its flat buffer and rebuilds only when they were really moved; it re-packs its copies of
        # the frozen weights whenever _weights_version moved)
        self.invalidate_packed()
        self._weights_version += 1
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.invalidate_packed()
        self._weights_version += 1
        return super().load_state_dict(*a, **k)

    # ---- packing ----
    def plan(self):
        if self._plan is not None:
            return self._plan
        dev = self.conv_in.weight.device
        if dev.type != "cuda":
            raise ops._lib.AldmError("UNet2DConditionModel runs on the MI355X only: call .to('cuda') first (no CPU fallback)")
        P = SimpleNamespace()
        P.te1 = ops.pack_linear(self.time_embedding.linear_1.weight, self.time_embedding.linear_1.bias)
        P.te2 = ops.pack_linear(self.time_embedding.linear_2.weight, self.time_embedding.linear_2.bias)
        P.cls = ops.pack_linear(self.class_embedding.weight, self.class_embedding.bias)
        P.conv_in = ops.pack_conv(self.conv_in.weight, self.conv_in.bias)
        resnets = []

        def res(r):
            p = pack_resnet(r)
            resnets.append((p, r))
            return p

        P.down = []
        for blk in self.down_blocks:
            b = SimpleNamespace(resnets=[res(r) for r in blk.resnets],
                                attns=[pack_transformer(t) for t in blk.attentions] if blk.has_attn else None,
                                down=ops.pack_conv(blk.downsamplers[0].conv.weight, blk.downsamplers[0].conv.bias) if blk.has_down else None)
            P.down.append(b)
        P.mid = SimpleNamespace(resnets=[res(r) for r in self.mid_block.resnets],
                                attns=[pack_transformer(self.mid_block.attentions[0])])
        P.up = []
        for blk in self.up_blocks:
            b = SimpleNamespace(resnets=[res(r) for r in blk.resnets],
                                attns=[pack_transformer(t) for t in blk.attentions] if blk.has_attn else None,
                                up=ops.pack_conv(blk.upsamplers[0].conv.weight, blk.upsamplers[0].conv.bias) if blk.has_up else None)
            P.up.append(b)
        # all time_emb_proj layers as one GEMM: rows concatenated in forward order
        off = 0
        ws, bs = [], []
        for p, r in resnets:
            p.temb_off = off
            off += r.cout
            ws.append(r.time_emb_proj.weight)
            bs.append(r.time_emb_proj.bias)
        P.temb_all = ops.pack_linear(torch.cat(ws), torch.cat(bs))
        P.temb_total = off
        P.gn_out = (_f32(self.conv_norm_out.weight), _f32(self.conv_norm_out.bias))
        P.conv_out = ops.pack_conv(self.conv_out.weight, self.conv_out.bias)
        P.version = self._plan_version
        self._plan = P
        return P

    # ---- the launch sequence ----
    def temb_table(self, timesteps_f32, class_labels_bf16):
        """The time-embedding projections of EVERY step at once: [n_steps, b, temb_total] fp32 for timesteps [n] (same scalar
        timestep for the whole batch, as in the DDIM loop) and class labels [b, D] -- four GEMMs per prompt instead of five
        launches per step."""
        cfg, P = self.cfg, self.plan()
        boc = cfg["block_out_channels"]
        ted = boc[0] * 4
        n, b = timesteps_f32.numel(), class_labels_bf16.shape[0]
        temb = ops.timestep_embedding(timesteps_f32.contiguous(), n, boc[0])
        e2 = ops.linear(ops.linear(temb, P.te1, out_act=ACT_SILU), P.te2, out_act=ACT_SILU)           # [n, ted]
        c = ops.linear(class_labels_bf16, P.cls, out_act=ACT_SILU)                                      # [b, ted]
        semb = torch.cat([e2[:, None, :].expand(n, b, ted), c[None].expand(n, b, ted)], dim=2).reshape(n * b, 2 * ted).contiguous()
        return ops.linear(semb, P.temb_all, out_f32=True).view(n, b, P.temb_total)

    def forward_nhwc(self, x, t_dev, class_labels_bf16, rowbias=None):
        """x [b, H, W, Cin] bf16 channels-last, t_dev fp32 [1] or [b] (device), class_labels [b, D] bf16.
        rowbias: optional precomputed time-embedding projections [b, temb_total] fp32 (temb_table); t_dev / class_labels are
        then unused.  Returns eps fp32 [b, H, W, Cout]."""
        ops.drop_pending(x)
        cfg, P = self.cfg, self.plan()
        fp8 = bool(getattr(self, "attention_fp8", False))       # BASELINE config 5: e4m3 Q / K / V / P attention operands
        b, H, W, _ = x.shape
        boc = cfg["block_out_channels"]
        ted = boc[0] * 4
        groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
        factor = 2 ** self.num_upsamplers
        forward_upsample_size = (H % factor != 0) or (W % factor != 0)

        # embeddings: silu(cat[time_embedding(t), class_embedding(c)]) -> one GEMM for all 22 time_emb_proj
        ld = P.temb_total
        if rowbias is None:
            temb = ops.timestep_embedding(t_dev, b, boc[0])
            e1 = ops.linear(temb, P.te1, out_act=ACT_SILU)
            semb = torch.empty(b, 2 * ted, dtype=torch.bfloat16, device=x.device)
            ops.linear(e1, P.te2, out_act=ACT_SILU, out=semb, out_ld=2 * ted)
            ops.linear(class_labels_bf16, P.cls, out_act=ACT_SILU, out=semb[:, ted:], out_ld=2 * ted)
            rowbias = ops.linear(semb, P.temb_all, out_f32=True)

        h = ops.conv(x, P.conv_in, pad=(1, 1), qstats=True)
        skips = [h]
        # `nxt` = (channels, groups) of the GroupNorm that consumes a block's output next (norm1 of the following ResnetBlock2D,
        # over torch.cat([h, skip]) on the way up): a split-K producer then leaves its reduce to that norm (ops.Deferred)
        def norm1_of(r):
            return (r.g1.numel(), r.groups)

        for bi, blk in enumerate(P.down):
            nr = len(blk.resnets)
            for i, r in enumerate(blk.resnets):
                if i + 1 < nr:
                    nxt = norm1_of(blk.resnets[i + 1])
                elif blk.down is not None:
                    nxt = None                                        # a down-sampling conv follows
                else:
                    nxt = norm1_of(P.mid.resnets[0])
                if blk.attns is not None:
                    h, hn = run_resnet(r, h, None, rowbias, ld, next_gn=transformer_gn(blk.attns[i]))
                    h = run_transformer(blk.attns[i], h, fp8, xn=hn, defer=nxt)
                else:
                    h = run_resnet(r, h, None, rowbias, ld, defer=nxt)
                skips.append(ops.tensor_of(h))
            if blk.down is not None:
                h = ops.conv(h, blk.down, stride=(2, 2), pad=(1, 1), defer=norm1_of(P.down[bi + 1].resnets[0]), qstats=True)
                skips.append(ops.tensor_of(h))
        h, hn = run_resnet(P.mid.resnets[0], h, None, rowbias, ld, next_gn=transformer_gn(P.mid.attns[0]))
        h = run_transformer(P.mid.attns[0], h, fp8, xn=hn, defer=norm1_of(P.mid.resnets[1]))
        h = run_resnet(P.mid.resnets[1], h, None, rowbias, ld, defer=norm1_of(P.up[0].resnets[0]))
        for bi, blk in enumerate(P.up):
            nr = len(blk.resnets)
            for i, r in enumerate(blk.resnets):
                if i + 1 < nr:
                    nxt = norm1_of(blk.resnets[i + 1])
                elif blk.up is not None:
                    nxt = None                                        # the up-sampling conv follows
                else:
                    nxt = (P.gn_out[0].numel(), groups)               # conv_norm_out
                if blk.attns is not None:
                    h, hn = run_resnet(r, h, skips.pop(), rowbias, ld, next_gn=transformer_gn(blk.attns[i]))
                    h = run_transformer(blk.attns[i], h, fp8, xn=hn, defer=nxt)
                else:
                    h = run_resnet(r, h, skips.pop(), rowbias, ld, defer=nxt)
            if blk.up is not None:
                if forward_upsample_size:
                    size = (skips[-1].shape[1], skips[-1].shape[2])
                else:
                    size = (h.shape[1] * 2, h.shape[2] * 2)
                h = ops.conv(h, blk.up, pad=(1, 1), up_size=size, defer=norm1_of(P.up[bi + 1].resnets[0]), qstats=True)
        if ops.gn_silu_conv_out_ok(h, P.conv_out, groups):        # norm + SiLU + the 128 -> 8 channel convolution as one launch
            return ops.gn_silu_conv_out(h, P.gn_out[0], P.gn_out[1], groups, eps, P.conv_out)
        h = ops.groupnorm(h, P.gn_out[0], P.gn_out[1], groups, eps, ACT_SILU)
        return ops.conv(h, P.conv_out, pad=(1, 1), out_f32=True)

    def forward(self, sample, timestep, encoder_hidden_states=None, class_labels=None, cross_attention_kwargs=None,
                return_dict=False, **kw):
        """diffusers signature: NCHW `sample`, scalar / [b] `timestep`, `class_labels` [b, D]."""
        assert encoder_hidden_states is None, "AudioLDM conditions through class_labels only"
        if not sample.is_cuda:
            raise ops._lib.AldmError("UNet2DConditionModel.forward needs CUDA/HIP tensors (no CPU fallback)")
        b = sample.shape[0]
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], device=sample.device)
        t = timestep.to(device=sample.device, dtype=torch.float32).reshape(-1)
        if t.numel() not in (1, b):
            raise ValueError("timestep must be a scalar or have one entry per sample")
        if self.training and torch.is_grad_enabled() and self._has_trainable_lora():
            # the reference's training call [REF train:479,539-546]: return a tensor autograd can differentiate -- forward and
            # backward both run on the HIP launch tape (training._UNetTrainFn)
            from .training import trainer_of
            out = trainer_of(self).autograd_forward(sample, t, class_labels)
            return SimpleNamespace(sample=out) if return_dict else (out,)
        x = ops.nchw_to_nhwc(sample.float())
        cls_bf16 = ops.f32_to_bf16(class_labels.to(device=sample.device, dtype=torch.float32).contiguous())
        eps = self.forward_nhwc(x, t.contiguous(), cls_bf16)
        out = ops.nhwc_to_nchw_f32(eps).to(sample.dtype)
        if return_dict:
            return SimpleNamespace(sample=out)
        return (out,)
