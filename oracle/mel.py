"""Oracle log-mel front end (test infrastructure; see oracle/__init__.py).

Restates the reference dataloader's feature extraction: `build_dsp` constants [REF script/data/datasets.py:69-83],
`mel_spectrogram_train` [REF script/data/datasets.py:301-354] (reflect pad (1024-160)/2, torch.stft n_fft 1024 /
hop 160 / periodic Hann 1024 / center=False, magnitude, librosa Slaney mel basis 64 bins 0-8 kHz, log(clamp(., 1e-5)))
and `pad_spec` [REF script/data/datasets.py:385-398] (zero-pad or crop to target_length = 1024 frames).
librosa is absent from the image; `slaney_mel_basis` restates librosa.filters.mel(htk=False, norm="slaney") and is
PINNED against transformers.audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") in
tests/test_oracle_mel.py; the STFT itself is torch.stft, the very call the reference makes.
"""
import numpy as np
import torch

DSP = dict(sampling_rate=16000, filter_length=1024, hop_length=160, win_length=1024, n_mel=64, mel_fmin=0, mel_fmax=8000,
           target_length=1024)


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax):
    """[n_mels, n_fft//2+1] float32 triangular filters, area-normalised (librosa.filters.mel defaults)."""
    fftfreqs = np.linspace(0, sr / 2, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def mel_spectrogram_train(y, dsp=DSP):
    """y [B, T] float32 in [-1, 1] -> (log-mel [B, n_mel, frames], |STFT| [B, 513, frames])."""
    n_fft, hop = dsp["filter_length"], dsp["hop_length"]
    basis = torch.from_numpy(slaney_mel_basis(dsp["sampling_rate"], n_fft, dsp["n_mel"], dsp["mel_fmin"], dsp["mel_fmax"]))
    p = int((n_fft - hop) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop, win_length=dsp["win_length"], window=torch.hann_window(dsp["win_length"]),
                      center=False, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.abs(spec)
    return torch.log(torch.clamp(torch.matmul(basis, spec), min=1e-5)), spec


def pad_spec(spec_tm, target_length):
    """[frames, bins] -> [target_length, bins]: zero rows appended or extra frames cut; an odd bin count loses its last bin."""
    n = spec_tm.shape[0]
    if n < target_length:
        spec_tm = torch.nn.functional.pad(spec_tm, (0, 0, 0, target_length - n))
    elif n > target_length:
        spec_tm = spec_tm[:target_length]
    if spec_tm.shape[-1] % 2:
        spec_tm = spec_tm[..., :-1]
    return spec_tm


def log_mel_spec(waveform, dsp=DSP):
    """waveform [B, T] -> the collate_fn tensor `log_mel_spec` [B, 1, target_length, n_mel] [REF script/train/train_audioldm_lora.py:415-420]."""
    mel, _ = mel_spectrogram_train(waveform, dsp)
    return torch.stack([pad_spec(m.T, dsp["target_length"]) for m in mel]).unsqueeze(1)
