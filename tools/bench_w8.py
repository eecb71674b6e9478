"""A/B of the 8-wave small tiles (10: 64x128w8, 11: 128x64w8) against the 4-wave ones on the split-K convolutions of the UNet's low-resolution
levels, in a replayed graph with rotating weights (cold weight stream, as in the step).  usage: python tools/bench_w8.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops

def run(name, B, H, W, Cin, Cout, cfgs, nrot=8, ext=0):
    g = torch.Generator().manual_seed(0)
    xs = [torch.randn(B, H, W, Cin, generator=g).to(torch.bfloat16).cuda() for _ in range(nrot)]
    if ext:      # conv2 + conv_shortcut over `ext` more channels of a second tensor (the fused 1x1 segment)
        x3s = [torch.randn(B, H, W, ext, generator=g).to(torch.bfloat16).cuda() for _ in range(nrot)]
        pws = [ops.pack_conv_shortcut((torch.randn(Cout, Cin, 3, 3, generator=g) / 50).cuda(), torch.zeros(Cout).cuda(),
                                      (torch.randn(Cout, ext, 1, 1, generator=g) / 20).cuda(), torch.zeros(Cout).cuda()) for _ in range(nrot)]
        conv0 = ops.conv
        def conv_ext(x, pw, **kw):
            return conv0(x, pw, x3=x3s[[id(t) for t in xs].index(id(x))], **kw)
        ops_conv = conv_ext
    else:
        ops_conv = ops.conv
        pws = [ops.pack_conv((torch.randn(Cout, Cin, 3, 3, generator=g) / 50).cuda(), torch.zeros(Cout).cuda()) for _ in range(nrot)]
    ref = None
    for (tile, ring, sp) in cfgs:
        try:
            y = ops_conv(xs[0], pws[0], pad=(1, 1), tile=tile, ring=ring, splits=sp)
        except Exception as e:
            print(f"{name} tile {tile} ring {ring} splits {sp}: {str(e)[:80]}"); continue
        if ref is None:
            ref = y.clone()
        same = bool(torch.equal(y, ref)) or float((y.float() - ref.float()).abs().max()) < 2 ** -6 * float(ref.float().abs().max())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for r in range(3):
                for i in range(nrot):
                    ops_conv(xs[i], pws[i], pad=(1, 1), tile=tile, ring=ring, splits=sp)
        gr.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / (3 * nrot))
        print(f"{name} tile {ops.TILE_NAMES[tile]:10s} ring {ring} splits {sp:2d}: {best * 1e3:7.2f} us (conv + reduce){'' if same else '  MISMATCH'}", flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "haloext":   # conv2 + conv_shortcut: the fused-shortcut halo-ws tiles against the generic ws tiles
    run("M8000 N256 C256 e384", 8, 125, 8, 256, 256, [(13, 3, 2), (14, 3, 2), (15, 3, 2), (15, 3, 1), (16, 3, 1), (16, 3, 2)], ext=384)
    run("M8000 N256 C256 e128", 8, 125, 8, 256, 256, [(14, 4, 1), (13, 3, 1), (15, 3, 1), (16, 3, 1), (15, 3, 2)], ext=128)
    run("M32000 N128 C128 e256", 8, 250, 16, 128, 128, [(4, 3, 1), (13, 3, 1), (15, 3, 1), (16, 3, 1)], ext=256)
    run("M2016 N384 C384 e640", 8, 63, 4, 384, 384, [(13, 3, 4), (16, 3, 3), (16, 3, 2), (16, 3, 4)], ext=640)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "halows":   # wave-specialised halo tiles (15, 16) against the plain halo tiles (7, 8)
    run("M512 N640 K5760", 8, 32, 2, 640, 640, [(14, 4, 6), (7, 3, 5), (15, 3, 5), (15, 4, 5), (15, 3, 10)])
    run("M2016 N384 K3456", 8, 63, 4, 384, 384, [(13, 3, 4), (7, 3, 3), (15, 3, 3), (15, 3, 6), (8, 3, 2), (16, 3, 2), (16, 3, 3), (16, 4, 3)])
    run("M2016 N384 K9216(C1024)", 8, 63, 4, 1024, 384, [(7, 3, 4), (15, 3, 4), (15, 4, 4), (15, 3, 8), (16, 3, 4)])
    run("M8000 N256 K2304", 8, 125, 8, 256, 256, [(8, 3, 1), (16, 3, 1), (16, 4, 1), (7, 3, 1), (15, 3, 1), (15, 3, 2), (16, 3, 2)])
    run("M8000 N256 K4608", 8, 125, 8, 512, 256, [(7, 3, 2), (15, 3, 2), (15, 4, 2), (16, 3, 2), (15, 3, 4)])
    run("M32000 N128 K1152", 8, 250, 16, 128, 128, [(7, 3, 1), (15, 3, 1), (15, 4, 1), (16, 3, 1)])
    run("M32000 N256 K2304", 8, 250, 16, 256, 256, [(7, 3, 1), (15, 3, 1), (15, 4, 1), (12, 3, 1)])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "halo":   # split-K halo tiles against the adopted wave-specialised tiles
    run("M512 N640 K5760", 8, 32, 2, 640, 640, [(14, 4, 6), (7, 3, 10), (7, 3, 5), (7, 4, 10), (7, 3, 4)])
    run("M512 N1280 K11520", 8, 32, 2, 1280, 1280, [(14, 4, 3), (7, 3, 10), (7, 3, 5), (7, 3, 4), (7, 4, 5)])
    run("M2016 N384 K3456", 8, 63, 4, 384, 384, [(13, 3, 4), (7, 3, 6), (7, 3, 3), (7, 4, 6), (8, 3, 6), (8, 3, 3), (8, 4, 3), (8, 3, 2)])
    run("M2016 N384 K9216(C1024)", 8, 63, 4, 1024, 384, [(13, 3, 4), (7, 3, 8), (7, 3, 6), (7, 3, 4), (8, 3, 4), (8, 3, 8)])
    run("M8000 N256 K2304", 8, 125, 8, 256, 256, [(13, 3, 1), (7, 3, 1), (7, 3, 2), (7, 4, 2), (7, 3, 4), (8, 3, 1), (8, 3, 2), (8, 4, 2)])
    run("M8000 N256 K4608", 8, 125, 8, 512, 256, [(13, 3, 2), (7, 3, 2), (7, 3, 4), (8, 3, 2), (8, 3, 4)])
    run("M32000 N128 K1152", 8, 250, 16, 128, 128, [(7, 3, 1), (7, 3, 2), (12, 3, 1)])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "rs":   # register-staged loaders (ring 2) against LDS-DMA loaders (ring 3)
    run("M512 N640 K5760", 8, 32, 2, 640, 640, [(11, 3, 6), (14, 3, 6), (14, 4, 6), (14, 2, 6), (13, 3, 6), (13, 2, 6), (14, 2, 4), (14, 2, 3)])
    run("M512 N1280 K11520", 8, 32, 2, 1280, 1280, [(14, 4, 3), (14, 2, 3), (14, 2, 6), (13, 2, 3)])
    run("M2016 N384 K3456", 8, 63, 4, 384, 384, [(13, 3, 4), (13, 2, 4), (14, 2, 4), (13, 2, 3), (13, 2, 2)])
    run("M2016 N384 K9216(C1024)", 8, 63, 4, 1024, 384, [(3, 3, 8), (13, 3, 4), (13, 2, 4), (14, 2, 4), (13, 2, 8)])
    run("M8000 N256 K2304", 8, 125, 8, 256, 256, [(13, 3, 1), (13, 2, 1), (14, 2, 1), (13, 2, 2)])
    run("M32000 N256 K2304", 8, 250, 16, 256, 256, [(12, 3, 1), (12, 2, 1), (9, 3, 1)])
    run("M32000 N128 K1152", 8, 250, 16, 128, 128, [(12, 3, 1), (12, 2, 1), (7, 3, 1)])
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "ws":   # the wave-specialised small tiles (13: 64x128ws, 14: 128x64ws)
    run("M512 N640 K5760", 8, 32, 2, 640, 640, [(4, 3, 6), (11, 3, 6), (13, 3, 6), (13, 4, 6), (13, 4, 4), (13, 4, 8), (13, 4, 12), (14, 3, 6), (14, 4, 6), (14, 4, 12), (14, 4, 8)])
    run("M512 N1280 K11520", 8, 32, 2, 1280, 1280, [(4, 3, 6), (11, 3, 6), (13, 4, 6), (13, 4, 4), (13, 4, 3), (14, 4, 6), (14, 4, 4), (14, 4, 3)])
    run("M2016 N384 K3456", 8, 63, 4, 384, 384, [(2, 3, 4), (10, 3, 4), (13, 3, 4), (13, 4, 4), (13, 4, 6), (13, 4, 3), (14, 4, 4), (14, 4, 6), (14, 4, 3)])
    run("M2016 N384 K9216(C1024)", 8, 63, 4, 1024, 384, [(3, 3, 8), (11, 3, 8), (13, 4, 8), (14, 4, 8), (14, 4, 6), (14, 4, 12), (13, 4, 12)])
    run("M8000 N256 K2304", 8, 125, 8, 256, 256, [(2, 2, 1), (10, 3, 1), (13, 3, 1), (13, 4, 1), (14, 3, 1), (14, 4, 1), (13, 4, 2), (14, 4, 2)])
    run("M8000 N128 K1152", 8, 125, 8, 128, 128, [(2, 2, 1), (10, 3, 1), (13, 4, 1), (14, 4, 1), (14, 3, 1)])
    sys.exit(0)
run("M512 N640 K5760", 8, 32, 2, 640, 640, [(4, 3, 6), (4, 4, 6), (10, 3, 6), (10, 4, 6), (10, 3, 4), (10, 3, 8), (10, 3, 12), (3, 3, 6), (11, 3, 6), (11, 4, 6), (11, 3, 12), (2, 3, 4), (1, 2, 12)])
run("M2016 N384 K3456", 8, 63, 4, 384, 384, [(2, 3, 4), (4, 3, 4), (10, 3, 4), (10, 4, 4), (10, 3, 6), (10, 3, 3), (11, 3, 4), (11, 3, 6), (3, 3, 4)])
run("M2016 N384 K9216(C1024)", 8, 63, 4, 1024, 384, [(3, 3, 8), (11, 3, 8), (10, 3, 8), (11, 4, 8), (10, 3, 6), (11, 3, 12)])
run("M8000 N256 K2304", 8, 125, 8, 256, 256, [(2, 2, 1), (4, 3, 2), (10, 3, 2), (10, 3, 1), (11, 3, 1), (11, 3, 2)])
