"""PIN: oracle CLAP text tower == transformers.ClapTextModelWithProjection (golden vectors + live class)."""
import os
import sys

import numpy as np
import torch

from oracle import configs
from oracle.clap_text import ClapTextModelWithProjection, position_ids

G = os.path.join(os.path.dirname(__file__), "golden", "clap_text_tiny.npz")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))


def _golden():
    z = np.load(G)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    return sd, {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("w::")}


def test_position_ids_fairseq_rule():
    ids = torch.tensor([[0, 5, 6, 2, 1, 1], [0, 9, 2, 1, 1, 1]])
    assert position_ids(ids, 1).tolist() == [[2, 3, 4, 5, 1, 1], [2, 3, 4, 1, 1, 1]]


def test_oracle_matches_golden_from_transformers():
    sd, v = _golden()
    m = ClapTextModelWithProjection(**configs.tiny_clap_text()).eval()
    m.load_state_dict(sd, strict=True)
    out = m(v["input_ids"], v["attention_mask"])
    torch.testing.assert_close(out.text_embeds, v["text_embeds"], rtol=1e-5, atol=1e-5)
    valid = v["attention_mask"].bool()
    torch.testing.assert_close(out.last_hidden_state[valid], v["last_hidden_state"][valid], rtol=1e-4, atol=1e-4)


def test_oracle_matches_live_transformers_with_and_without_mask():
    from make_golden import hf_clap_text
    cfg = dict(configs.tiny_clap_text(), num_hidden_layers=3, hidden_size=96, num_attention_heads=6)
    torch.manual_seed(3)
    hf = hf_clap_text(cfg)
    m = ClapTextModelWithProjection(**cfg).eval()
    m.load_state_dict({k: v for k, v in hf.state_dict().items() if not k.endswith(("position_ids", "token_type_ids"))}, strict=True)
    ids = torch.randint(3, cfg["vocab_size"], (2, 16))
    with torch.no_grad():
        ref = hf(input_ids=ids).text_embeds
    torch.testing.assert_close(m(ids).text_embeds, ref, rtol=1e-5, atol=1e-6)
    mask = torch.ones(2, 16, dtype=torch.long)
    ids[1, 7:] = cfg["pad_token_id"]
    mask[1, 7:] = 0
    with torch.no_grad():
        ref = hf(input_ids=ids, attention_mask=mask).text_embeds
    torch.testing.assert_close(m(ids, mask).text_embeds, ref, rtol=1e-5, atol=1e-6)


def test_padding_never_changes_valid_tokens():
    """The property the HIP path relies on to cut the 512-token padding: truncating the batch to its longest item
    leaves text_embeds unchanged."""
    sd, v = _golden()
    m = ClapTextModelWithProjection(**configs.tiny_clap_text()).eval()
    m.load_state_dict(sd, strict=True)
    ids, mask = v["input_ids"][1:], v["attention_mask"][1:]            # lengths 9 and 1
    full = m(ids, mask).text_embeds
    cut = m(ids[:, :16], mask[:, :16]).text_embeds
    torch.testing.assert_close(cut, full, rtol=1e-5, atol=1e-6)


def test_full_config_param_count():
    m = ClapTextModelWithProjection()
    assert sum(p.numel() for p in m.parameters()) == 125302016          # RoBERTa-base 124.6 M + projection head 0.66 M
