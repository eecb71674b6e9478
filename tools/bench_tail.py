"""Times the once-per-clip tail of the pipeline (VAE decode + HiFi-GAN vocoder) at config 2: 4 x 10 s clips."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops  # noqa: E402
from audioldm_with_lora_amd.vae import AutoencoderKL  # noqa: E402
from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan  # noqa: E402

torch.manual_seed(0)
vae, voc = AutoencoderKL().cuda(), SpeechT5HifiGan().cuda()
z = torch.randn(4, 250, 16, 8, device="cuda").to(torch.bfloat16)
for name, fn in (("vae.decode", lambda: vae.decode_nhwc(z)),):
    for _ in range(2):
        mel = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        mel = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms  (4 clips; 2616 GFLOP algorithmic)")
melb = ops.f32_to_bf16(mel).view(4, 1, 1000, 64)
for _ in range(2):
    w = voc.forward_nhwc(melb)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    w = voc.forward_nhwc(melb)
torch.cuda.synchronize()
print(f"vocoder: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms  (4 clips; 4011 GFLOP algorithmic) out {tuple(w.shape)}")
ops.PROFILE = []
voc.forward_nhwc(melb)
vae.decode_nhwc(z)
torch.cuda.synchronize()
agg = {}
for label, fl, by, s, e, _site in ops.PROFILE:
    a = agg.setdefault(label, [0.0, 0, 0.0])
    a[0] += s.elapsed_time(e); a[1] += 1; a[2] += fl
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{k:70s} {v[1]:3d}x {v[0]:8.2f} ms {v[2] / v[0] / 1e9 if v[0] else 0:7.1f} TF/s")
