"""The VAE's 64- / 32-wide 3x3 convolutions on the 128x128 tile, the 8-wave 128x128 tile and the halo tile (five-pass halo at 64 wide),
in a replayed graph.  usage: python tools/bench_vae_conv.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops


def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


g = torch.Generator().manual_seed(0)
for name, B, H, W, ci, co in (("enc 128->128 @1024x64", 8, 1024, 64, 128, 128), ("dec 128->128 @1000x64", 4, 1000, 64, 128, 128),
                              ("enc 256->256 @512x32", 8, 512, 32, 256, 256), ("dec 256->256 @500x32", 4, 500, 32, 256, 256),
                              ("dec 256->128 @1000x64", 4, 1000, 64, 256, 128)):
    x = torch.randn(B, H, W, ci, generator=g).to(torch.bfloat16).cuda()
    pw = ops.pack_conv((torch.randn(co, ci, 3, 3, generator=g) * 0.03).cuda(), torch.randn(co, generator=g).cuda())
    fl = 2.0 * B * H * W * co * 9 * ci
    row = []
    for tile, ring in ((1, 2), (1, 3), (6, 2), (7, 2), (7, 3), (7, 4)):
        try:
            us = t(lambda: ops.conv(x, pw, pad=(1, 1), tile=tile, ring=ring, splits=1))
            row.append(f"t{tile}r{ring} {us:7.1f} us {fl / us / 1e6:6.0f} TF/s")
        except Exception as e:
            row.append(f"t{tile}r{ring} -- {str(e)[:40]}")
    print(f"{name:24s} " + " | ".join(row), flush=True)
