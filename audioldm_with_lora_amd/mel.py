"""Log-mel front end on the GPU -- SURVEY.md 8(f) row 4.

`LogMelFrontEnd()(waveform [B, T])` returns the tensor the reference's dataset hands to collate_fn as `log_mel_spec`
[B, 1, 1024, 64] [REF script/data/datasets.py:301-354, 385-398] [REF script/train/train_audioldm_lora.py:415-420],
computed by one HIP kernel (csrc/mel.hip: reflect pad, 1024-point LDS FFT, magnitude, Slaney mel, log-clamp, pad/crop)
instead of four CPU dataloader workers running librosa/torch.stft.  Constants are the reference's `build_dsp` literals
[REF script/data/datasets.py:69-83].
"""
import numpy as np
import torch

from . import _lib
from .ops import _p, _require_gpu, _stream, check

DSP = dict(sampling_rate=16000, filter_length=1024, hop_length=160, win_length=1024, n_mel=64, mel_fmin=0, mel_fmax=8000,
           target_length=1024)


def _slaney_scale(x, inverse):
    x = np.asarray(x, dtype=np.float64)
    lin, brk, step = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    if inverse:
        return np.where(x >= brk / lin, brk * np.exp(step * (x - brk / lin)), lin * x)
    return np.where(x >= brk, brk / lin + np.log(np.maximum(x, 1e-10) / brk) / step, x / lin)


def mel_filter_bank(sr, n_fft, n_mels, fmin, fmax):
    """Slaney-scale, area-normalised triangular filters [n_mels, n_fft//2+1] (what librosa.filters.mel returns by default)."""
    bins = np.linspace(0.0, sr / 2.0, n_fft // 2 + 1)
    edges = _slaney_scale(np.linspace(_slaney_scale(fmin, False), _slaney_scale(fmax, False), n_mels + 2), True)
    up = (bins[None, :] - edges[:-2, None]) / (edges[1:-1] - edges[:-2])[:, None]
    down = (edges[2:, None] - bins[None, :]) / (edges[2:] - edges[1:-1])[:, None]
    tri = np.clip(np.minimum(up, down), 0.0, None)
    return (tri * (2.0 / (edges[2:] - edges[:-2]))[:, None]).astype(np.float32)


class LogMelFrontEnd:
    def __init__(self, device="cuda", **over):
        d = dict(DSP)
        d.update(over)
        self.dsp = d
        if d["win_length"] != d["filter_length"]:
            raise ValueError("win_length must equal filter_length (as in the reference's build_dsp)")
        basis = mel_filter_bank(d["sampling_rate"], d["filter_length"], d["n_mel"], d["mel_fmin"], d["mel_fmax"])
        rng = np.zeros((d["n_mel"], 2), dtype=np.int32)
        for m, row in enumerate(basis):
            nz = np.nonzero(row)[0]
            rng[m] = (nz.min(), nz.max() + 1) if nz.size else (0, 0)
        self.device = torch.device(device)
        self.basis = torch.from_numpy(basis).to(self.device)
        self.ranges = torch.from_numpy(rng).to(self.device)
        self.window = torch.hann_window(d["win_length"], periodic=True, dtype=torch.float32).to(self.device)

    def __call__(self, waveform):
        """waveform fp32 [B, T] (or [T]) on the GPU -> log_mel_spec fp32 [B, 1, target_length, n_mel]."""
        if waveform.dim() == 1:
            waveform = waveform[None]
        _require_gpu(waveform)
        w = waveform.to(torch.float32).contiguous()
        B, T = w.shape
        d = self.dsp
        out = torch.empty(B, 1, d["target_length"], d["n_mel"], dtype=torch.float32, device=w.device)
        check(_lib.load().aldm_log_mel(_p(w), B, T, d["filter_length"], d["hop_length"], _p(self.window), _p(self.basis),
                                       _p(self.ranges), d["n_mel"], d["target_length"], 1e-5, _p(out), _stream()),
              "aldm_log_mel")
        return out
