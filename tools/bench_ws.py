"""A/B of the wave-specialised 256x128 tile (12) against the other big tiles on the VAE / vocoder / level-0 shapes, in a replayed graph with
rotating operands.  usage: python tools/bench_ws.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops

def run(name, B, H, W, Cin, Cout, kh, kw, cfgs, nrot=4, dil=1):
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda() for _ in range(nrot)]
    pws = [ops.pack_conv((torch.randn(Cout, Cin, kh, kw, generator=g) / 50).cuda(), torch.zeros(Cout).cuda()) for _ in range(nrot)]
    pad = (kh // 2, (kw // 2) * dil)
    fl = 2.0 * B * H * W * Cout * Cin * kh * kw
    ref = None
    for (tile, ring) in cfgs:
        try:
            y = ops.conv(xs[0], pws[0], pad=pad, dil=(1, dil), tile=tile, ring=ring, splits=1)
        except Exception as e:
            print(f"{name} tile {tile}: {str(e)[:90]}"); continue
        if ref is None:
            ref = y.clone()
        ok = float((y.float() - ref.float()).abs().max()) <= 2 ** -6 * float(ref.float().abs().max())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for r in range(3):
                for i in range(nrot):
                    ops.conv(xs[i], pws[i], pad=pad, dil=(1, dil), tile=tile, ring=ring, splits=1)
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (3 * nrot))
        print(f"{name} tile {ops.TILE_NAMES[tile]:10s} ring {ring}: {best * 1e3:8.2f} us  {fl / best / 1e9:7.1f} TF/s{'' if ok else '  MISMATCH'}", flush=True)

C = [(1, 2), (1, 3), (6, 2), (9, 2), (9, 3), (12, 3)]
run("VAE  M64000  N512 K4608", 4, 250, 64, 512, 512, 3, 3, C)
run("VAE  M256000 N256 K2304", 4, 1000, 64, 256, 256, 3, 3, C, nrot=2)
run("VAE  M256000 N128 K1152", 4, 1000, 64, 128, 128, 3, 3, C, nrot=2)
run("voc  M80016  N256 K2816", 4, 1, 20004, 256, 256, 1, 11, C, dil=3)
run("voc  M20004  N512 K5632", 4, 1, 5001, 512, 512, 1, 11, C, dil=5)
run("voc  M160032 N128 K1408", 4, 1, 40008, 128, 128, 1, 11, C, nrot=2)
run("UNet M32000  N256 K2304", 8, 250, 16, 256, 256, 3, 3, C)
