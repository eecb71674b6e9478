"""A/B timer for the flash-attention kernel at the three UNet self-attention sites (config 2: CFG batch 8).
    ALDM_LIB=/path/to/libaldm_hip.so python tools/bench_attn.py      # time another build of the library in the same box
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops

for (B, N, H, d) in ((8, 1000, 8, 32), (8, 252, 8, 48), (8, 64, 8, 80), (8, 1000, 4, 32), (16, 1024, 8, 32)):
    C = H * d
    qk = torch.randn(B * N, 2 * C, device="cuda").to(torch.bfloat16)
    vt = torch.randn(B, C, (N + 7) // 8 * 8, device="cuda").to(torch.bfloat16)
    out = torch.empty(B * N, C, dtype=torch.bfloat16, device="cuda")
    for ps, f8 in ((False, False), (True, False), (True, True)):
        for _ in range(5):
            ops.attention(qk, vt, B, N, H, d, out=out, prescaled=ps, fp8=f8)
        reps = 50
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                ops.attention(qk, vt, B, N, H, d, out=out, prescaled=ps, fp8=f8)
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        fl = 4.0 * B * N * N * C
        print(f"attention B{B} N{N} H{H} d{d} prescaled={int(ps)} fp8={int(f8)}: {best * 1e3:8.2f} us  {fl / best / 1e9:7.1f} TF/s", flush=True)
