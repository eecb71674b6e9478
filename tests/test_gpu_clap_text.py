"""GPU parity for SURVEY.md 8(f) row 2: the CLAP text tower on the HIP kernels vs the transformers-pinned oracle
(tests/test_oracle_clap_text.py) and the transformers-made golden vectors; plus the two ops added for it."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    import conftest
    return conftest.record(float((a.float() - b.float()).norm() / b.float().norm()))


def test_embed_layernorm_matches_torch():
    from audioldm_with_lora_amd import ops
    from oracle.clap_text import position_ids
    g = torch.Generator().manual_seed(0)
    V, P, C, B, L, pad = 300, 70, 768, 3, 64, 1
    word, pos, typ = torch.randn(V, C, generator=g), torch.randn(P, C, generator=g), torch.randn(1, C, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    ids = torch.randint(2, V, (B, L), generator=g)
    ids[1, 20:] = pad
    ids[2, 1:] = pad
    want = F.layer_norm(word[ids] + typ[0] + pos[position_ids(ids, pad)], (C,), gamma, beta, 1e-12)
    got = ops.embed_layernorm(ids.cuda(), word.cuda(), pos.cuda(), typ[0].contiguous().cuda(), gamma.cuda(), beta.cuda(), 1e-12, pad)
    assert got.shape == (B * L, C)
    torch.testing.assert_close(got.float().cpu().view(B, L, C), want, rtol=1e-2, atol=1e-2)     # bf16 output rounding only


def test_embed_layernorm_rejects_too_many_positions():
    from audioldm_with_lora_amd import ops
    from audioldm_with_lora_amd._lib import AldmError
    C = 64
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(AldmError):
        ops.embed_layernorm(torch.zeros(1, 40, dtype=torch.int64, device="cuda"), z(10, C), z(41, C), z(C), z(C), z(C), 1e-5, 1)


@pytest.mark.parametrize("d,H,N,lens", [(64, 12, 64, (64, 33, 1, 8)), (16, 4, 24, (24, 9, 1)), (64, 12, 512, (512, 77, 300, 5))])
def test_attention_varlen_matches_masked_sdpa(d, H, N, lens):
    from audioldm_with_lora_amd import ops
    g = torch.Generator().manual_seed(1)
    B, C = len(lens), H * d
    q, k, v = (torch.randn(B, N, C, generator=g).bfloat16() for _ in range(3))
    qk = torch.cat([q, k], dim=2).reshape(B * N, 2 * C).cuda()
    vt = v.transpose(1, 2).contiguous().cuda()                       # [B, C, N]
    kv = torch.tensor(lens, dtype=torch.int32).cuda()
    got = ops.attention(qk, vt, B, N, H, d, kv_len=kv).float().cpu().view(B, N, C)
    bias = torch.zeros(B, 1, 1, N)
    for b, n in enumerate(lens):
        bias[b, ..., n:] = float("-inf")
    sp = lambda t: t.float().view(B, N, H, d).transpose(1, 2)
    want = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=bias).transpose(1, 2).reshape(B, N, C)
    for b, n in enumerate(lens):                                   # valid queries: parity; all-padding workgroups: zeros
        assert rel_l2(got[b, :n], want[b, :n]) < 1.5e-2
    full = ops.attention(qk, vt, B, N, H, d, kv_len=torch.full((B,), N, dtype=torch.int32).cuda())
    assert torch.equal(full, ops.attention(qk, vt, B, N, H, d))     # kv_len = N is the unmasked kernel


def _load_tiny():
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.configs import tiny_clap_text
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "clap_text_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    m = ClapTextModelWithProjection(**tiny_clap_text())
    m.load_state_dict(sd, strict=True)
    return m.cuda(), z


def test_clap_text_matches_transformers_golden():
    m, z = _load_tiny()
    ids, mask = torch.from_numpy(z["input_ids"]), torch.from_numpy(z["attention_mask"])
    out = m(input_ids=ids.cuda(), attention_mask=mask.cuda())
    want = torch.from_numpy(z["text_embeds"])
    assert out.text_embeds.dtype == torch.float32 and out.text_embeds.shape == want.shape
    assert rel_l2(out.text_embeds.cpu(), want) < 3e-2
    hs, ref = out.last_hidden_state.float().cpu(), torch.from_numpy(z["last_hidden_state"])
    for b in range(ids.shape[0]):
        n = int(mask[b].sum())
        assert rel_l2(hs[b, :n], ref[b, :n]) < 3e-2
    # the same captions padded to a longer max_length give the same embeddings (padding is cut / masked)
    pad = torch.ones(ids.shape[0], 8, dtype=ids.dtype)
    out2 = m(input_ids=torch.cat([ids, pad], 1).cuda(), attention_mask=torch.cat([mask, 0 * pad], 1).cuda())
    torch.testing.assert_close(out2.text_embeds, out.text_embeds, rtol=1e-3, atol=1e-3)


def test_clap_text_rejects_left_padding_and_cpu():
    m, z = _load_tiny()
    ids = torch.from_numpy(z["input_ids"])
    mask = torch.from_numpy(z["attention_mask"]).clone()
    mask[0, 0] = 0
    with pytest.raises(ValueError):
        m(input_ids=ids.cuda(), attention_mask=mask)
    from audioldm_with_lora_amd._lib import AldmError
    with pytest.raises(AldmError):
        m.cpu()(input_ids=ids, attention_mask=torch.from_numpy(z["attention_mask"]))


def test_clap_text_full_config_matches_oracle_on_512_token_padding():
    """RoBERTa-base size, captions padded to 512 tokens as the reference's dataset does [REF script/data/datasets.py:128-134]."""
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from oracle.clap_text import ClapTextModelWithProjection as OClap
    torch.manual_seed(21)
    ref = OClap().eval()
    g = torch.Generator().manual_seed(22)
    sd = ref.state_dict()
    for k, v in sd.items():
        if k.endswith("weight") and v.dim() == 2 and "embeddings" not in k:
            v.copy_(torch.randn(v.shape, generator=g) / v.shape[1] ** 0.5)
        elif k.endswith("bias"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
    ref.load_state_dict(sd)
    B, L, lens = 4, 512, (37, 12, 64, 5)
    ids = torch.randint(3, 50265, (B, L), generator=g)
    mask = torch.ones(B, L, dtype=torch.long)
    for b, n in enumerate(lens):
        ids[b, 0] = 0
        ids[b, n - 1] = 2
        ids[b, n:] = 1
        mask[b, n:] = 0
    want = F.normalize(ref(ids[:, :64], mask[:, :64]).text_embeds, dim=-1)     # oracle on the cut batch (== padded, see CPU test)
    m = ClapTextModelWithProjection()
    m.load_state_dict(sd, strict=True)
    m = m.cuda()
    got = F.normalize(m(input_ids=ids.cuda(), attention_mask=mask.cuda()).text_embeds, dim=-1).cpu()
    assert got.shape == (B, 512)
    assert rel_l2(got, want) < 3e-2
    cos = (got * want).sum(-1)
    assert float(cos.min()) > 0.999
