"""PIN: oracle HiFi-GAN == transformers.SpeechT5HifiGan (golden vectors + live class)."""
import os

import numpy as np
import pytest
import torch

from oracle import configs
from oracle.hifigan import SpeechT5HifiGan

G = os.path.join(os.path.dirname(__file__), "golden", "vocoder_tiny.npz")


def _golden():
    z = np.load(G)
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    return sd, torch.from_numpy(z["mel"]), torch.from_numpy(z["wav"])


def test_oracle_matches_golden_from_transformers():
    sd, mel, wav = _golden()
    m = SpeechT5HifiGan(**configs.tiny_vocoder()).eval()
    m.load_state_dict(sd, strict=True)
    with torch.no_grad():
        got = m(mel)
    assert got.shape == wav.shape == (2, 160 * 12 + 32)
    torch.testing.assert_close(got, wav, rtol=1e-5, atol=1e-5)


def test_oracle_matches_live_transformers_full_config():
    tr = pytest.importorskip("transformers")
    c = configs.VOCODER
    cfg = tr.SpeechT5HifiGanConfig(
        model_in_dim=c["model_in_dim"], sampling_rate=c["sampling_rate"],
        upsample_initial_channel=c["upsample_initial_channel"], upsample_rates=list(c["upsample_rates"]),
        upsample_kernel_sizes=list(c["upsample_kernel_sizes"]),
        resblock_kernel_sizes=list(c["resblock_kernel_sizes"]),
        resblock_dilation_sizes=[list(d) for d in c["resblock_dilation_sizes"]],
        leaky_relu_slope=c["leaky_relu_slope"], normalize_before=False)
    torch.manual_seed(3)
    ref = tr.SpeechT5HifiGan(cfg).eval()
    mine = SpeechT5HifiGan().eval()
    assert sum(p.numel() for p in mine.parameters()) == 55264897
    assert len(mine.state_dict()) == 196
    torch.manual_seed(4)
    sd = {k: (torch.randn_like(v) * 0.02 if v.dim() > 1 else torch.randn_like(v) * 0.01)
          for k, v in mine.state_dict().items()}
    mine.load_state_dict(sd, strict=True)
    ref.load_state_dict(sd, strict=True)
    mel = torch.randn(1, 6, 64)
    with torch.no_grad():
        a, b = mine(mel), ref(mel)
    assert a.shape == (1, 160 * 6 + 32)
    torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)
