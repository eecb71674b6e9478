"""Micro-benchmark of aldm_igemm on the UNet's layer shapes (config 2: UNet batch 8, latent 250x16).
usage: python tools/bench_igemm.py [--tiles 1,2,3,4] [--reps 30]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops  # noqa: E402

SHAPES = [  # (name, B, H, W, Cin, Cin2, Cout, k, stride, up)
    ("L0 conv 128->128", 8, 250, 16, 128, 0, 128, 3, 1, None),
    ("L0 conv 256+128->128", 8, 250, 16, 256, 128, 128, 3, 1, None),
    ("L0 upconv 256->256 (from 125x8)", 8, 125, 8, 256, 0, 256, 3, 1, (250, 16)),
    ("L0 down 128->128 s2", 8, 250, 16, 128, 0, 128, 3, 2, None),
    ("L1 conv 256->256", 8, 125, 8, 256, 0, 256, 3, 1, None),
    ("L1 conv 384+256->256", 8, 125, 8, 384, 256, 256, 3, 1, None),
    ("L1 lin 256->768", 8, 125, 8, 256, 0, 768, 1, 1, None),
    ("L1 lin 256->2048", 8, 125, 8, 256, 0, 2048, 1, 1, None),
    ("L1 lin 1024->256", 8, 125, 8, 1024, 0, 256, 1, 1, None),
    ("L1 lin 256->256", 8, 125, 8, 256, 0, 256, 1, 1, None),
    ("L2 conv 384->384", 8, 63, 4, 384, 0, 384, 3, 1, None),
    ("L2 conv 640+384->384", 8, 63, 4, 640, 384, 384, 3, 1, None),
    ("L2 lin 384->3072", 8, 63, 4, 384, 0, 3072, 1, 1, None),
    ("L2 lin 1536->384", 8, 63, 4, 1536, 0, 384, 1, 1, None),
    ("L3 conv 640->640", 8, 32, 2, 640, 0, 640, 3, 1, None),
    ("L3 conv 640+640->640", 8, 32, 2, 640, 640, 640, 3, 1, None),
    ("L3 lin 640->5120", 8, 32, 2, 640, 0, 5120, 1, 1, None),
    ("L3 lin 2560->640", 8, 32, 2, 2560, 0, 640, 1, 1, None),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="0,1,3,2")
    ap.add_argument("--splits", default="0")
    ap.add_argument("--ring", default="0")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    splits = [int(t) for t in a.splits.split(",")]
    rings = [int(t) for t in a.ring.split(",")]
    dev = "cuda"
    tot = {}
    for name, B, H, W, c1, c2, co, k, st, up in SHAPES:
        if a.only and a.only not in name:
            continue
        x = torch.randn(B, H, W, c1, device=dev).to(torch.bfloat16)
        x2 = torch.randn(B, H, W, c2, device=dev).to(torch.bfloat16) if c2 else None
        w = torch.randn(co, c1 + c2, k, k, device=dev) / math.sqrt(k * k * (c1 + c2))
        pw = ops.pack_conv(w, torch.randn(co, device=dev))
        pad = (k // 2, k // 2)
        line = f"{name:34s}"
        for t in tiles:
            for sp in [(a_, b_) for a_ in splits for b_ in rings]:
                sp, rg = sp
                kw = dict(x2=x2, stride=(st, st), pad=pad, up_size=up, tile=t, splits=(None if sp == 0 else sp), ring=rg)
                y = ops.conv(x, pw, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    ops.conv(x, pw, **kw)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / a.reps
                M = y.shape[0] * y.shape[1] * y.shape[2]
                fl = 2.0 * M * co * k * k * (c1 + c2)
                line += f" | t{t}s{sp}r{rg}: {us:7.1f}us {fl / us / 1e6:6.0f}TF"
                tot[(t, sp, rg)] = tot.get((t, sp, rg), 0.0) + us
        print(line, flush=True)
    print("sum us:", {k: round(v, 1) for k, v in tot.items()})
    if a.only and "qkv" not in a.only:
        return
    # transformer projections with the fused LoRA side channel (rank 4 on q,k,v / out) and the V^T store
    for B, N, Cc in ((8, 1000, 256), (8, 252, 384), (8, 64, 640)):
        M = B * N
        x = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
        res = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
        wq = torch.randn(3 * Cc, Cc, device=dev) / math.sqrt(Cc)
        A = [torch.randn(4, Cc, device=dev) / 4 for _ in range(3)]
        Bm = [torch.randn(Cc, 4, device=dev) * 0.02 for _ in range(3)]
        npad = (N + 7) // 8 * 8
        vt = torch.empty(B, Cc, npad, device=dev, dtype=torch.bfloat16)
        variants = {}
        pw = ops.pack_linear(wq, None); variants["qkv plain"] = (pw, {})
        pw = ops.pack_linear(wq, None); variants["qkv vt"] = (pw, dict(vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad))
        pw = ops.pack_linear(wq, None); ops.attach_lora(pw, [(i * Cc, Cc, A[i], Bm[i], 1.0) for i in range(3)]); variants["qkv lora"] = (pw, {})
        variants["qkv lora+vt"] = (pw, dict(vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad))
        po = ops.pack_linear(wq[:Cc], torch.zeros(Cc, device=dev)); variants["out plain+res"] = (po, dict(res=res.view(B, 1, N, Cc)))
        po2 = ops.pack_linear(wq[:Cc], torch.zeros(Cc, device=dev)); ops.attach_lora(po2, [(0, Cc, A[0], Bm[0], 1.0)]); variants["out lora+res"] = (po2, dict(res=res.view(B, 1, N, Cc)))
        for t in tiles:
            line = f"C={Cc} N={N} tile{t}: "
            for name, (pw, kw) in variants.items():
                xx = x.view(B, 1, N, Cc)
                ops.conv(xx, pw, tile=t, **kw)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    ops.conv(xx, pw, tile=t, **kw)
                e1.record()
                torch.cuda.synchronize()
                line += f"{name}: {e0.elapsed_time(e1) * 1e3 / a.reps:6.1f}us | "
            print(line, flush=True)


if __name__ == "__main__":
    main()
