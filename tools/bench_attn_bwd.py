"""Attention backward (aldm_attention_bwd: dQ kernel + dK/dV kernel) at the UNet's three sites, inside a replayed graph.
ALDM_ATTN_BWD_NW=2|4|8 overrides the waves per workgroup.  usage: python tools/bench_attn_bwd.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops  # noqa: E402


def main():
    B, H = 8, 8
    for N, d in ((1024, 32), (256, 48), (64, 80)):
        C = H * d
        g = torch.Generator().manual_seed(0)
        qkv = torch.randn(B * N, 3 * C, generator=g).to(torch.bfloat16).cuda()
        npad = (N + 7) // 8 * 8
        qkvT = ops.transpose_tokens(qkv, B, N, 3 * C, npad)
        O, lse = ops.attention_train(qkv, qkvT, B, N, H, d)
        dO = torch.randn(B * N, C, generator=g).to(torch.bfloat16).cuda()
        ops.attention_bwd(qkv, qkvT, dO, O, lse, B, N, H, d)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                ops.attention_bwd(qkv, qkvT, dO, O, lse, B, N, H, d)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
        print(f"N{N} d{d} NW={os.environ.get('ALDM_ATTN_BWD_NW', 'default')}: {best:7.1f} us per backward (transpose of dO + dQ + dK/dV)", flush=True)


if __name__ == "__main__":
    main()
