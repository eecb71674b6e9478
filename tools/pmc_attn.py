"""Micro-driver for PMC runs: launches the attention kernel at the three UNet sites a few times."""
import os, sys, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops
for (B, N, H, d) in ((8, 1000, 8, 32), (8, 252, 8, 48), (8, 64, 8, 80)):
    C = H * d
    qk = torch.randn(B * N, 2 * C, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, C, (N + 7) // 8 * 8, device="cuda").to(torch.bfloat16)
    for _ in range(4):
        ops.attention(qk, vt, B, N, H, d)
torch.cuda.synchronize()
