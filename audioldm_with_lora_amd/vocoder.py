"""SpeechT5HifiGan (HiFi-GAN vocoder) on the MI355X HIP path.

Drop-in for `transformers.SpeechT5HifiGan` as the reference uses it: `from_pretrained(id, subfolder="vocoder")`,
`forward(spectrogram [B, T, 64]) -> waveform [B, 160 T + 32]`, `.config.{upsample_rates, sampling_rate, model_in_dim}`
  [REF script/train/train_audioldm_lora.py:371]; runs inside AudioLDMPipeline.__call__
  [REF script/inference/generate_audio.py:47-52].  Spec: SURVEY.md B.6
(transformers/models/speecht5/modeling_speecht5.py:2887-3069); state-dict keys identical to the HF class.

The 32- / 64-channel stages (320 k - 640 k samples, 18 of the 45 residual (conv1, conv2) pairs) run each pair as ONE launch of the
vocoder's own kernel (csrc/hifigan.hip, aldm_hifigan_respair: the run is activated on its way into LDS, conv1's output never leaves
LDS, both convolutions' weights stay in registers as MFMA fragments, residual / MRF mean / next activation in the epilogue).
Everything else is the implicit-GEMM MFMA kernel over channels-last [B, T, C] sequences:
  * Conv1d(k, dilation d)            -> 1 x k filter, dil_w = d
  * ConvTranspose1d(k, stride u, p)  -> u phase convolutions: output t = u q + phi - p takes taps j = phi + u i from
    input q - i; each phase is a plain conv with n = ceil((k - phi)/u) taps, run with dil_w = -1 and written with an
    output row stride of u (no zero-stuffing, no wasted MFMA work)
  * leaky-relu never runs as its own pass over the residual stream: each conv epilogue stores the stream AND its
    leaky-relu'd copy (the next conv's input) in one pass; the MRF mean (sum of 3 resblocks / 3) accumulates in the
    last conv of each resblock.
"""
import json
import os
from types import SimpleNamespace

import torch
from torch import nn

from . import ops
from ._lib import ACT_LRELU, ACT_NONE, ACT_TANH
from .configs import VOCODER


class HifiGanResidualBlock(nn.Module):
    def __init__(self, ch, k, dil, slope):
        super().__init__()
        self.convs1 = nn.ModuleList([nn.Conv1d(ch, ch, k, dilation=d, padding=(k * d - d) // 2) for d in dil])
        self.convs2 = nn.ModuleList([nn.Conv1d(ch, ch, k, dilation=1, padding=(k - 1) // 2) for _ in dil])
        self.k, self.dil = k, tuple(dil)


def _pad_c(t, dim, mult=8):
    """zero-pad dimension `dim` up to a multiple of 8 (the gather moves 16-byte channel chunks)."""
    n = t.shape[dim]
    m = (n + mult - 1) // mult * mult
    if m == n:
        return t
    shape = list(t.shape)
    shape[dim] = m - n
    return torch.cat([t, torch.zeros(shape, dtype=t.dtype, device=t.device)], dim=dim)


def pack_conv1d(w, b):
    """Conv1d weight [Cout, Cin, k]; channel counts zero-padded to multiples of 8."""
    w = _pad_c(_pad_c(w.detach(), 0), 1)
    b = _pad_c(b.detach(), 0)
    return ops.pack_conv(w.unsqueeze(2), b)


def pack_conv_transpose1d(w, b, stride, padding):
    """ConvTranspose1d weight [Cin, Cout, k] -> one packed conv per output phase."""
    cin, cout, k = w.shape
    w = _pad_c(_pad_c(w.detach(), 0), 1)
    b = _pad_c(b.detach(), 0)
    phases = []
    ntap = (k + stride - 1) // stride
    for phi in range(stride):
        taps = list(range(phi, k, stride))
        wp = torch.zeros(w.shape[1], w.shape[0], 1, ntap, dtype=w.dtype, device=w.device)   # [Cout, Cin, 1, ntap]
        for i, j in enumerate(taps):
            wp[:, :, 0, i] = w[:, :, j].t()
        phases.append(ops.pack_conv(wp, b))
    return SimpleNamespace(phases=phases, stride=stride, padding=padding, k=k, ntap=ntap)


def run_conv_transpose1d(P, x, out_len, post_act=ACT_NONE, post_slope=0.0, out2=False):
    """x [B, 1, T, Cin] -> y [B, 1, out_len, Cout] (+ optional leaky-relu'd copy)."""
    B, _, T, _ = x.shape
    co = P.phases[0].N
    y = torch.empty(B, 1, out_len, co, dtype=torch.bfloat16, device=x.device)
    y2 = torch.empty_like(y) if out2 else None
    u, p = P.stride, P.padding
    for phi, pw in enumerate(P.phases):
        # outputs t = u*q + phi - p, q >= q0 ; input index = q - i
        q0 = max(0, -((phi - p) // u))           # smallest q with t >= 0  (ceil((p - phi)/u))
        t0 = u * q0 + phi - p
        if t0 >= out_len:
            continue
        nq = (out_len - 1 - t0) // u + 1
        ops.conv(x, pw, pad=(0, -q0), dil=(1, -1), out_hw=(1, nq), out=y, out_ld=co, out_batch_stride=out_len * co,
                 out_pix_stride=u, out_pix_offset=t0, post_act=post_act, post_slope=post_slope, out2=y2)
    return (y, y2) if out2 else y


class SpeechT5HifiGan(nn.Module):
    def __init__(self, **over):
        super().__init__()
        cfg = dict(VOCODER)
        cfg.update({k: v for k, v in over.items() if k in VOCODER})
        self.config = SimpleNamespace(**cfg)
        c0 = cfg["upsample_initial_channel"]
        self.conv_pre = nn.Conv1d(cfg["model_in_dim"], c0, 7, padding=3)
        self.upsampler = nn.ModuleList()
        for i, (u, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            self.upsampler.append(nn.ConvTranspose1d(c0 // 2 ** i, c0 // 2 ** (i + 1), k, stride=u, padding=(k - u) // 2))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.upsampler)):
            ch = c0 // 2 ** (i + 1)
            for k, d in zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"]):
                self.resblocks.append(HifiGanResidualBlock(ch, k, d, cfg["leaky_relu_slope"]))
        self.conv_post = nn.Conv1d(ch, 1, 7, padding=3)
        self.register_buffer("mean", torch.zeros(cfg["model_in_dim"]))
        self.register_buffer("scale", torch.ones(cfg["model_in_dim"]))
        self.num_kernels = len(cfg["resblock_kernel_sizes"])
        self._plan = None

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        f = os.path.join(d, "config.json")
        if not os.path.isfile(f):
            raise FileNotFoundError(f"{f} not found: hub downloads are unavailable, pass a local directory")
        raw = json.load(open(f))
        m = cls(**{k: (tuple(map(tuple, v)) if k == "resblock_dilation_sizes" else tuple(v) if isinstance(v, list) else v)
                   for k, v in raw.items() if k in VOCODER})
        from safetensors.torch import load_file
        m.load_state_dict(load_file(os.path.join(d, "model.safetensors")), strict=True)
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
        if self.conv_pre.weight.device.type != "cuda":
            raise ops._lib.AldmError("SpeechT5HifiGan runs on the MI355X only: call .to('cuda') first (no CPU fallback)")
        P = SimpleNamespace()
        P.pre = pack_conv1d(self.conv_pre.weight, self.conv_pre.bias)
        P.ups = [pack_conv_transpose1d(u.weight, u.bias, u.stride[0], u.padding[0]) for u in self.upsampler]
        P.res = [SimpleNamespace(c1=[pack_conv1d(c.weight, c.bias) for c in rb.convs1],
                                 c2=[pack_conv1d(c.weight, c.bias) for c in rb.convs2], k=rb.k, dil=rb.dil)
                 for rb in self.resblocks]
        P.post = pack_conv1d(self.conv_post.weight, self.conv_post.bias)
        self._plan = P
        return P

    def forward_nhwc(self, mel):
        """mel [B, 1, T, 64] channels-last bf16 -> waveform [B, 160 T + 32] fp32."""
        cfg, P = self.config, self.plan()
        slope = cfg.leaky_relu_slope
        if cfg.normalize_before:
            raise NotImplementedError("normalize_before=True is not used by cvssp/audioldm-s-full-v2")
        # conv_pre; only leaky_relu(h) is consumed downstream (by the first up-sampler)
        a = ops.conv(mel, P.pre, pad=(0, 3), out_act=ACT_LRELU, out_slope=slope)
        T = a.shape[2]
        nk = self.num_kernels
        nstage = len(P.ups)
        for i, up in enumerate(P.ups):
            T = (T - 1) * up.stride - 2 * up.padding + up.k
            last = i == nstage - 1
            # the 32- / 64-channel stages: every (convs1[q], convs2[q]) pair is ONE launch (aldm_hifigan_respair: the run is activated on
            # its way into LDS, conv1's output never leaves LDS) -- no activated copy of the stream is needed from the up-sampler
            fused = all(ops.hifigan_respair_ok(torch.empty(0, 1, 0, up.phases[0].N), rb.c1[q], rb.c2[q], rb.dil[q])
                        for rb in P.res[i * nk:(i + 1) * nk] for q in range(len(rb.c1)))
            if fused:
                h = run_conv_transpose1d(up, a, T)
                acc = None
                for j in range(nk):
                    rb = P.res[i * nk + j]
                    r, npairs = h, len(rb.c1)
                    for q, (c1, c2, d) in enumerate(zip(rb.c1, rb.c2, rb.dil)):
                        if q < npairs - 1:
                            r = ops.hifigan_respair(r, c1, c2, d, slope)
                        else:
                            fin = j == nk - 1
                            acc = ops.hifigan_respair(r, c1, c2, d, slope, alpha=1.0 / nk, res2=acc,
                                                      post_act=(ACT_LRELU if fin else ACT_NONE), post_slope=(0.01 if last else slope))
                a = acc
                continue
            h, ha = run_conv_transpose1d(up, a, T, post_act=ACT_LRELU, post_slope=slope, out2=True)
            acc = None
            for j in range(nk):
                rb = P.res[i * nk + j]
                r, ra = h, ha                                   # residual stream and its leaky-relu'd copy
                npairs = len(rb.c1)
                for q, (c1, c2, d) in enumerate(zip(rb.c1, rb.c2, rb.dil)):
                    t = ops.conv(ra, c1, pad=(0, (rb.k * d - d) // 2), dil=(1, d), out_act=ACT_LRELU, out_slope=slope)
                    if q < npairs - 1:
                        out2 = torch.empty_like(r)
                        r = ops.conv(t, c2, pad=(0, (rb.k - 1) // 2), res=r, out2=out2, post_act=ACT_LRELU, post_slope=slope)
                        ra = out2
                    else:
                        # last conv of the resblock: accumulate the MRF mean  acc = (acc) + (conv + r)/nk ; the final
                        # accumulation also applies the leaky-relu feeding the next stage (slope 0.1) or conv_post
                        # (torch default slope 0.01, modeling_speecht5.py:3058)
                        fin = j == nk - 1
                        acc = ops.conv(t, c2, pad=(0, (rb.k - 1) // 2), res=r, alpha=1.0 / nk, res2=acc,
                                       post_act=(ACT_LRELU if fin else ACT_NONE), post_slope=(0.01 if last else slope))
            a = acc
        if a.shape[3] % 8 == 0 and a.shape[3] <= 128 and P.post.KW % 2 == 1 and P.post.KW <= 15:
            return ops.conv1d_to1(a, P.post, act=ACT_TANH)       # one output channel: an HBM-bound stencil, not a GEMM with 1 live column
        w = ops.conv(a, P.post, pad=(0, (P.post.KW - 1) // 2), out_act=ACT_TANH, out_f32=True)
        return w[:, 0, :, 0].contiguous()

    def forward(self, spectrogram):
        if not spectrogram.is_cuda:
            raise ops._lib.AldmError("SpeechT5HifiGan.forward needs CUDA/HIP tensors (no CPU fallback)")
        batched = spectrogram.dim() == 3
        x = spectrogram if batched else spectrogram.unsqueeze(0)
        mel = ops.f32_to_bf16(x.float().contiguous()).unsqueeze(1)      # [B, 1, T, 64] channels-last
        wav = self.forward_nhwc(mel)
        return wav if batched else wav[0]
