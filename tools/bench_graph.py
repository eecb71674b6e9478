"""In-graph micro-timer: microseconds per launch of one igemm configuration inside a replayed hipGraph of `reps` launches
(kernels then run back to back as in the captured UNet step; eager per-launch timings sit on a ~12-15 us Python floor).
usage: python tools/bench_graph.py M N K [--k 1|3] [--tile T] [--ring R] [--splits S] [--res] [--rowbias] [--lora r]"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("B", type=int); ap.add_argument("H", type=int); ap.add_argument("W", type=int)
    ap.add_argument("Cin", type=int); ap.add_argument("Cout", type=int)
    ap.add_argument("--k", type=int, default=1)
    ap.add_argument("--tile", type=int, default=0); ap.add_argument("--ring", type=int, default=0)
    ap.add_argument("--splits", type=int, default=0)
    ap.add_argument("--res", action="store_true"); ap.add_argument("--rowbias", action="store_true")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    xs = [torch.randn(a.B, a.H, a.W, a.Cin, device=dev).to(torch.bfloat16) for _ in range(2)]
    pw = ops.pack_conv(torch.randn(a.Cout, a.Cin, a.k, a.k, device=dev) / math.sqrt(a.k * a.k * a.Cin), torch.randn(a.Cout, device=dev))
    res = torch.randn(a.B, a.H, a.W, a.Cout, device=dev).to(torch.bfloat16) if a.res else None
    rb = torch.randn(a.B, a.Cout, device=dev) if a.rowbias else None
    kw = dict(pad=(a.k // 2, a.k // 2), tile=a.tile, ring=a.ring, splits=(a.splits or None), res=res, rowbias=rb, rowbias_ld=(a.Cout if a.rowbias else 0))
    outs = [torch.empty(a.B, a.H, a.W, a.Cout, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    ops.conv(xs[0], pw, out=outs[0], **kw)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(a.reps):
            ops.conv(xs[i & 1], pw, out=outs[i & 1], **kw)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / a.reps)
    M = a.B * a.H * a.W
    fl = 2.0 * M * a.Cout * a.k * a.k * a.Cin
    print(f"B{a.B} {a.H}x{a.W} {a.Cin}->{a.Cout} k{a.k} tile{a.tile} ring{a.ring} splits{a.splits} dbg={os.environ.get('ALDM_DBG', '0')}: "
          f"{best:7.2f} us/launch  {fl / best / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
