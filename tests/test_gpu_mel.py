"""GPU parity for SURVEY.md 8(f) row 4: the one-kernel log-mel front end vs the oracle (torch.stft + pinned Slaney basis)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(y, atol=2e-3):
    from audioldm_with_lora_amd.mel import LogMelFrontEnd
    from oracle import mel as omel
    want = omel.log_mel_spec(y)
    got = LogMelFrontEnd()(y.cuda()).cpu()
    assert got.shape == want.shape and got.dtype == torch.float32
    # log domain: absolute tolerance; bins sitting at the 1e-5 clamp are exact on both sides
    torch.testing.assert_close(got, want, rtol=0, atol=atol)
    return got, want


def test_noise_clip_10s24_full_length():
    g = torch.Generator().manual_seed(0)
    y = (torch.rand(2, 163840, generator=g) * 2 - 1) * 0.8
    got, want = _check(y)
    assert got.shape == (2, 1, 1024, 64)


def test_short_clip_zero_padded_and_long_clip_cropped():
    g = torch.Generator().manual_seed(1)
    got, _ = _check(torch.randn(3, 16000 * 3 + 37, generator=g).clamp(-1, 1) * 0.3)   # ragged length, not a hop multiple
    n = 1 + (16000 * 3 + 37 + 864 - 1024) // 160
    assert torch.equal(got[:, :, n:], torch.zeros_like(got[:, :, n:])) and float(got[:, :, n - 1].abs().sum()) > 0
    _check(torch.randn(1, 160 * 1100, generator=g).clamp(-1, 1) * 0.3)


def test_tones_and_silence():
    t = torch.arange(32000) / 16000.0
    y = torch.stack([0.5 * torch.sin(2 * math.pi * 440.0 * t), 0.25 * torch.sin(2 * math.pi * 3000.0 * t) + 0.1 * torch.sin(2 * math.pi * 7000.0 * t),
                     torch.zeros(32000)])
    got, want = _check(y, atol=5e-3)                                        # deep spectral valleys: fp32 FFT round-off differs slightly
    assert torch.allclose(got[2, 0, :200], torch.full((200, 64), math.log(1e-5)))


def test_rejects_other_fft_sizes_and_cpu():
    from audioldm_with_lora_amd._lib import AldmError
    from audioldm_with_lora_amd.mel import LogMelFrontEnd
    with pytest.raises(AldmError):
        LogMelFrontEnd(filter_length=512, win_length=512)(torch.zeros(1, 16000).cuda())
    with pytest.raises(AldmError):
        LogMelFrontEnd()(torch.zeros(1, 16000))
