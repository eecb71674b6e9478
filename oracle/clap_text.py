"""Oracle CLAP text tower (test infrastructure; see oracle/__init__.py).

Restates `transformers.ClapTextModelWithProjection.forward` -> `.text_embeds`, which the reference calls once per
training step [REF script/train/train_audioldm_lora.py:513-524] and `AudioLDMPipeline._encode_prompt` calls once per
prompt [REF script/inference/generate_audio.py:47-52]: RoBERTa-base encoder (post-LN), tanh pooler on token 0, and the
Linear-ReLU-Linear projection head.  Spec followed: transformers/models/clap/modeling_clap.py (ClapTextEmbeddings,
ClapTextSelfAttention, ClapTextSelfOutput, ClapTextIntermediate, ClapTextOutput, ClapTextPooler, ClapProjectionLayer).
PINNED: tests/test_oracle_clap_text.py compares this module with the importable transformers class on identical
weights (same state-dict keys), and tests/golden/clap_text_tiny.npz holds vectors produced by that class.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F
from torch import nn

from .configs import CLAP_TEXT


class _SelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)


class _SelfOutput(nn.Module):
    def __init__(self, cin, c, eps):
        super().__init__()
        self.dense = nn.Linear(cin, c)
        self.LayerNorm = nn.LayerNorm(c, eps=eps)


class _Attention(nn.Module):
    def __init__(self, c, eps):
        super().__init__()
        self.self = _SelfAttention(c)
        self.output = _SelfOutput(c, c, eps)


class _Intermediate(nn.Module):
    def __init__(self, c, i):
        super().__init__()
        self.dense = nn.Linear(c, i)


class _Layer(nn.Module):
    def __init__(self, c, i, eps):
        super().__init__()
        self.attention = _Attention(c, eps)
        self.intermediate = _Intermediate(c, i)
        self.output = _SelfOutput(i, c, eps)


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


class _Pooler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.dense = nn.Linear(c, c)


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings = _Embeddings(cfg)
        self.encoder = _Encoder(cfg)
        self.pooler = _Pooler(cfg["hidden_size"])


class _Projection(nn.Module):
    def __init__(self, c, p):
        super().__init__()
        self.linear1, self.linear2 = nn.Linear(c, p), nn.Linear(p, p)


def position_ids(input_ids, pad):
    """fairseq make_positions: non-pad tokens count from pad+1, pads stay at pad."""
    mask = input_ids.ne(pad).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad


class ClapTextModelWithProjection(nn.Module):
    def __init__(self, **over):
        super().__init__()
        cfg = dict(CLAP_TEXT)
        cfg.update(over)
        self.cfg = cfg
        self.text_model = _TextModel(cfg)
        self.text_projection = _Projection(cfg["hidden_size"], cfg["projection_dim"])

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, **kw):
        cfg, tm = self.cfg, self.text_model
        B, L = input_ids.shape
        H = cfg["num_attention_heads"]
        e = tm.embeddings
        x = e.word_embeddings(input_ids) + e.token_type_embeddings.weight[0] + e.position_embeddings(position_ids(input_ids, cfg["pad_token_id"]))
        x = e.LayerNorm(x)
        bias = None
        if attention_mask is not None:                                    # additive mask over KEYS, broadcast over heads / queries
            bias = torch.zeros(B, 1, 1, L).masked_fill(attention_mask[:, None, None, :] == 0, float("-inf"))
        for lyr in tm.encoder.layer:
            sa = lyr.attention.self
            split = lambda t: t.view(B, L, H, -1).transpose(1, 2)
            a = F.scaled_dot_product_attention(split(sa.query(x)), split(sa.key(x)), split(sa.value(x)), attn_mask=bias)
            a = a.transpose(1, 2).reshape(B, L, -1)
            x = lyr.attention.output.LayerNorm(lyr.attention.output.dense(a) + x)
            h = F.gelu(lyr.intermediate.dense(x))
            x = lyr.output.LayerNorm(lyr.output.dense(h) + x)
        pooled = torch.tanh(tm.pooler.dense(x[:, 0]))
        p = self.text_projection
        emb = p.linear2(F.relu(p.linear1(pooled)))
        return SimpleNamespace(text_embeds=emb, last_hidden_state=x, pooler_output=pooled)
