"""CLAP text tower on the HIP kernels -- SURVEY.md 8(f) row 2.

Mirror of `transformers.ClapTextModelWithProjection` as the reference uses it: once per training step on the
tokenised captions [REF script/train/train_audioldm_lora.py:513-524] (`text_encoder(input_ids, attention_mask,
return_dict=True).text_embeds`, then F.normalize) and once per prompt inside AudioLDMPipeline._encode_prompt
[REF script/inference/generate_audio.py:47-52].  Parameter names equal the transformers checkpoint's
(`text_model.embeddings.*`, `text_model.encoder.layer.{i}.attention.self.query.*`, ..., `text_projection.linear{1,2}.*`),
so `text_encoder/model.safetensors` loads unchanged.

Launch sequence per forward (bf16 activations, fp32 accumulation / statistics):
    embed_layernorm                                   1 kernel   (gather word+type+position, LayerNorm)
    12 x [ fused QKV GEMM (V written token-major) -> flash attention with per-item key length
           -> dense + residual -> LayerNorm -> dense + erf-GELU -> dense + residual -> LayerNorm ]
    pooler dense+tanh on token 0 -> linear+ReLU -> linear (fp32 out)
The reference pads every caption to 512 tokens [REF script/data/datasets.py:128-134]; padding never influences a valid
token (keys are masked, rows are independent everywhere else), so the batch is cut to its longest caption rounded up
to 8 tokens and the attention key loop additionally stops at each item's own length.
"""
import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from ._lib import ACT_GELU, ACT_LRELU, ACT_TANH, AldmError
from .configs import CLAP_TEXT


def _f32(t):
    return t.detach().float().contiguous()


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)


class _DenseLN(nn.Module):
    def __init__(self, cin, c, eps):
        super().__init__()
        self.dense = nn.Linear(cin, c)
        self.LayerNorm = nn.LayerNorm(c, eps=eps)


class _Attention(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _DenseLN(c, c, eps)


class _Dense(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.dense = nn.Linear(cin, cout)


class _Layer(nn.Module):
    def __init__(self, c, i, eps):
        super().__init__()
        self.attention = _Attention(c, eps)
        self.intermediate = _Dense(c, i)
        self.output = _DenseLN(i, c, eps)


class _Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        c = cfg["hidden_size"]
        self.word_embeddings = nn.Embedding(cfg["vocab_size"], c, padding_idx=cfg["pad_token_id"])
        self.position_embeddings = nn.Embedding(cfg["max_position_embeddings"], c, padding_idx=cfg["pad_token_id"])
        self.token_type_embeddings = nn.Embedding(cfg["type_vocab_size"], c)
        self.LayerNorm = nn.LayerNorm(c, eps=cfg["layer_norm_eps"])


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([_Layer(cfg["hidden_size"], cfg["intermediate_size"], cfg["layer_norm_eps"])
                                    for _ in range(cfg["num_hidden_layers"])])


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.pooler = _Dense(cfg["hidden_size"], cfg["hidden_size"])


class _Projection(nn.Module):
    def __init__(self, c, p):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(c, p), nn.Linear(p, p)


class ClapTextModelWithProjection(nn.Module):
    def __init__(self, **over):
        super().__init__()
        cfg = dict(CLAP_TEXT)
        cfg.update({k: v for k, v in over.items() if k in CLAP_TEXT})
        self.cfg = cfg
        self.config = SimpleNamespace(**cfg)
        self.text_model = _TextModel(cfg)
        self.text_projection = _Projection(cfg["hidden_size"], cfg["projection_dim"])
        self._plan = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        f = os.path.join(d, "config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f} not found: hub downloads are unavailable, pass a local directory")
        raw = json.load(open(f))
        raw = raw.get("text_config", raw)
        m = cls(**{k: raw[k] for k in CLAP_TEXT if k in raw})
        from safetensors.torch import load_file
        sd = load_file(os.path.join(d, "model.safetensors"))
        sd = {k: v for k, v in sd.items() if not k.endswith(("position_ids", "token_type_ids"))}   # persistent index buffers
        m.load_state_dict(sd, strict=True)
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
        tm = self.text_model
        if tm.pooler.dense.weight.device.type != "cuda":
            raise AldmError("ClapTextModelWithProjection runs on the MI355X only: call .to('cuda') first (no CPU fallback)")
        pl = lambda m: ops.pack_linear(m.weight, m.bias)
        e = tm.embeddings
        P = SimpleNamespace(
            word=_f32(e.word_embeddings.weight), pos=_f32(e.position_embeddings.weight),
            type0=_f32(e.token_type_embeddings.weight[0]), emb_ln=(_f32(e.LayerNorm.weight), _f32(e.LayerNorm.bias)),
            layers=[], pool=pl(tm.pooler.dense), p1=pl(self.text_projection.linear1), p2=pl(self.text_projection.linear2))
        for lyr in tm.encoder.layer:
            sa = lyr.attention.self
            P.layers.append(SimpleNamespace(
                qkv=ops.pack_linear(torch.cat([sa.query.weight, sa.key.weight, sa.value.weight]),
                                    torch.cat([sa.query.bias, sa.key.bias, sa.value.bias])),
                ao=pl(lyr.attention.output.dense),
                ln1=(_f32(lyr.attention.output.LayerNorm.weight), _f32(lyr.attention.output.LayerNorm.bias)),
                ff1=pl(lyr.intermediate.dense), ff2=pl(lyr.output.dense),
                ln2=(_f32(lyr.output.LayerNorm.weight), _f32(lyr.output.LayerNorm.bias))))
        self._plan = P
        return P

    @staticmethod
    def _lengths(input_ids, attention_mask):
        """Valid-token count per item; the padding mask must be a prefix mask (right padding, the RoBERTa tokenizer's)."""
        B, L = input_ids.shape
        if attention_mask is None:
            return torch.full((B,), L, dtype=torch.int32)
        m = attention_mask.to("cpu").reshape(B, L) != 0
        if bool((m[:, 1:] & ~m[:, :-1]).any()):
            raise ValueError("attention_mask must be right-padded (ones then zeros); left/inner padding is not supported")
        lens = m.sum(1).to(torch.int32)
        if int(lens.min()) < 1:
            raise ValueError("every item needs at least one unmasked token")
        return lens

    @torch.no_grad()
    def forward(self, input_ids=None, attention_mask=None, return_dict=True, **kw):
        P, cfg = self.plan(), self.cfg
        dev = self.text_model.pooler.dense.weight.device
        B, L = input_ids.shape
        C, H, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
        lens = self._lengths(input_ids, attention_mask)
        Le = min(L, (int(lens.max()) + 7) // 8 * 8)                  # longest caption, 8-token granules
        ids = input_ids[:, :Le].to(dev, torch.int64).contiguous()
        kv_len = lens.to(dev)
        emb, x, pooled = self.forward_device(ids, kv_len)
        if not return_dict:
            return (emb, x)
        return SimpleNamespace(text_embeds=emb, last_hidden_state=x, pooler_output=pooled)

    @torch.no_grad()
    def forward_device(self, ids, kv_len):
        """The launch sequence alone: ids int64 [B, Le] and kv_len int32 [B] already on the device, Le >= every length.  No host
        synchronisation, so it can sit inside a captured hipGraph (training.LoraTrainer.step_from_batch)."""
        P, cfg = self.plan(), self.cfg
        dev = ids.device
        B, Le = ids.shape
        C, H, eps = cfg["hidden_size"], cfg["num_attention_heads"], cfg["layer_norm_eps"]
        x = ops.embed_layernorm(ids, P.word, P.pos, P.type0, P.emb_ln[0], P.emb_ln[1], eps, cfg["pad_token_id"])
        for lp in P.layers:
            vt = torch.empty(B, C, Le, dtype=torch.bfloat16, device=dev)
            qk = ops.conv(x.view(B, 1, Le, C), lp.qkv, vt=vt, vt_col0=2 * C, vt_ld=Le, vt_batch_stride=C * Le)
            a = ops.attention(qk.view(B * Le, 2 * C), vt, B, Le, H, C // H, kv_len=kv_len)
            x = ops.layernorm(ops.linear(a, lp.ao, res=x), lp.ln1[0], lp.ln1[1], eps)
            h = ops.linear(x, lp.ff1, out_act=ACT_GELU)
            x = ops.layernorm(ops.linear(h, lp.ff2, res=x), lp.ln2[0], lp.ln2[1], eps)
        first = x.view(B, Le, C)[:, 0].contiguous()
        pooled = ops.linear(first, P.pool, out_act=ACT_TANH)
        t = ops.linear(pooled, P.p1, out_act=ACT_LRELU, out_slope=0.0)                 # ReLU
        emb = ops.linear(t, P.p2, out_f32=True)
        return emb, x.view(B, Le, C), pooled

