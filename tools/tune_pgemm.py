"""Launch-shape sweep for aldm_pgemm (csrc/pgemm.hip): every valid (rows per workgroup, tile width, tiles per range) of every
projection GEMM of the UNet's transformer blocks, timed inside a replayed hipGraph of `reps` launches over rotating buffers (the
kernels then run back to back as in the captured denoise step).  Writes audioldm_with_lora_amd/pgemm_gfx950.json, which ops.py
loads into PGEMM_CFG.
usage: python tools/tune_pgemm.py [--batch 8] [--reps 24] [--write]"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def make_case(kind, M, N, K, B, nbuf=3):
    """operands of one GEMM as the UNet issues it; returns (key, run(i)) with run launching on buffer set i"""
    g = torch.Generator(device="cpu").manual_seed(0)
    xs = [torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV) for _ in range(nbuf)]
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    gm, bt = (torch.randn(K, generator=g) * 0.3 + 1).to(DEV), (torch.randn(K, generator=g) * 0.2).to(DEV)
    lora = lambda n0, n: (n0, n, (torch.randn(4, K, generator=g) / 4).to(DEV), (torch.randn(n, 4, generator=g) * 0.05).to(DEV), 1.0)
    parts = torch.randn(M, 4, 2, generator=g).abs().to(DEV)
    parts[:, :, 1] = parts[:, :, 0] ** 2 + K / 4.0
    if kind == "proj_in":
        pw = ops.pack_linear(w, b)
        return "s", lambda i: ops.linear(xs[i], pw, rowstats=True)
    if kind == "plain":
        pw = ops.pack_linear(w, b)
        return "", lambda i: ops.linear(xs[i], pw)
    if kind == "out_t":                                      # the trainer's to_out.0: rank 8, T copy, residual
        pw = ops.pack_linear(w, b)
        ops.attach_lora(pw, [(0, N, (torch.randn(8, K, generator=g) / 8).to(DEV), (torch.randn(N, 8, generator=g) * 0.05).to(DEV), 1.0)])
        res = [torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV) for _ in range(nbuf)]
        T = torch.empty(M, pw.Rp, dtype=torch.bfloat16, device=DEV)
        return f"rl{pw.Rp}t", lambda i: ops.linear(xs[i], pw, res=res[i], lora_t_out=T)
    if kind == "dual":                                       # the trainer's q | k | v forward (N = 3C, rank 8 each) / out-projection dX (N = C)
        pw = ops.pack_linear(w, b)
        nl = 3 if N > K else 1
        ops.attach_lora(pw, [(j * (N // nl), N // nl, (torch.randn(8, K, generator=g) / 8).to(DEV), (torch.randn(N // nl, 8, generator=g) * 0.05).to(DEV), 1.0)
                             for j in range(nl)])
        pw.ranks_used = 8 * nl
        n = M // B
        npad = (n + 7) // 8 * 8
        yTs = [torch.zeros(B, N, npad, dtype=torch.bfloat16, device=DEV) for _ in range(nbuf)]
        T = torch.empty(M, pw.Rp, dtype=torch.bfloat16, device=DEV)
        return f"vl{pw.Rp}td", lambda i: ops.conv(xs[i].view(B, 1, n, K), pw, lora_t_out=T, splits=1, vt=yTs[i], vt_col0=0, vt_ld=npad,
                                                  vt_batch_stride=N * npad, vt_dual=True)
    if kind == "out":
        pw = ops.pack_linear(w, b)
        ops.attach_lora(pw, [lora(0, N)])
        res = [torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV) for _ in range(nbuf)]
        return f"rl{pw.Rp}s", lambda i: ops.linear(xs[i], pw, res=res[i], rowstats=True)
    if kind == "qkv":
        C = N // 3
        pw = ops.pack_linear_ln(w, None, gm, bt)
        ops.attach_lora(pw, [lora(j * C, C) for j in range(3)])
        n = M // B
        npad = (n + 7) // 8 * 8
        vts = [torch.zeros(B, C, npad, dtype=torch.bfloat16, device=DEV) for _ in range(nbuf)]
        return f"vl{pw.Rp}", lambda i: ops.conv(xs[i].view(B, 1, n, K), pw, vt=vts[i], vt_col0=2 * C, vt_ld=npad, vt_batch_stride=C * npad, ln_parts=parts)
    if kind == "ff1":
        pw = ops.pack_linear_ln(w, b, gm, bt, geglu=True)
        return "g", lambda i: ops.linear(xs[i], pw, ln_parts=parts)
    raise ValueError(kind)


def configs(kind, M, N, K):
    out = []
    for nw in (4, 8):
        for mi in ((1, 2) if ((K <= 384 or kind in ("proj_in", "ff1", "plain")) and nw == 4) else (1,)):
            for nt in ((64,) if kind == "ff1" else (32, 64)):
                if kind == "qkv" and (2 * N // 3) % nt:
                    continue
                nti = N // nt
                res = kind in ("out", "out_t")
                rp = 32 if kind in ("out", "qkv", "out_t", "dual") else 0
                bm = 16 * mi * nw
                stage = nt * K * 2 + (bm * nt * 2 if res else 0)
                for t in range(1, nti + 1):
                    if nti % t or t * nt > 512:
                        continue
                    if kind in ("proj_in", "out") and nti // t > 16:
                        continue
                    if min(t, 3) * stage + 2 * bm * 4 + 2 * t * nt * 4 + t * nt * rp * 2 > 160 * 1024:
                        continue
                    if math.ceil(M / bm) * (nti // t) > 2048:
                        continue
                    out.append((mi, nt, t, nw))
    return out


def time_cfg(run, reps, nbuf):
    run(0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            run(i % nbuf)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[8, 2])
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--fresh", action="store_true", help="start from an empty table")
    ap.add_argument("--train", action="store_true", help="also the taped training step's shapes (batch 8 x 256x16 latents)")
    a = ap.parse_args()
    table = {}
    nbuf = 3
    path = os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "pgemm_gfx950.json")
    if os.path.exists(path) and not a.fresh:                 # keep what earlier runs measured
        table.update(json.load(open(path))["pgemm"])
    jobs = [(B, C, n, kinds) for B in a.batch for C, n, kinds in
            ((256, 1000, None), (384, 252, None), (640, 64, None))]
    if a.train:                                              # proj_in / its dX / ff2's dX (plain), to_out.0 forward (LoRA + residual + T copy)
        jobs += [(8, C, n, (("plain", C), ("plain", 4 * C), ("out_t", C), ("dual", 3 * C), ("dual", C))) for C, n in ((256, 1024), (384, 256), (640, 64))]
    for B, C, n, kinds in jobs:
        if True:
            M = B * n
            for kind, N in (kinds or (("proj_in", C), ("qkv", 3 * C), ("out", C), ("ff1", 8 * C))):
                if a.only and a.only != kind:
                    continue
                sfx, run = make_case(kind, M, N, C, B, nbuf)
                key = (M, N, C, sfx)
                ops.PGEMM_CFG.pop(key, None)
                base = time_cfg(run, a.reps, nbuf)
                res = []
                for cfg in configs(kind, M, N, C):
                    ops.PGEMM_CFG[key] = cfg
                    try:
                        res.append((time_cfg(run, a.reps, nbuf), cfg))
                    except _lib.AldmError as e:
                        print(f"  {cfg}: {e}", flush=True)
                ops.PGEMM_CFG.pop(key, None)
                res.sort()
                os.environ["ALDM_NO_PGEMM"] = "1"
                ops.PGEMM = False
                old = time_cfg(run, a.reps, nbuf)
                ops.PGEMM = True
                print(f"{kind:8s} M{M} N{N} K{C}: plan {base:6.2f} us | igemm {old:6.2f} us | best " +
                      "  ".join(f"{c} {t:5.2f}" for t, c in res[:5]) + f" | worst {res[-1][1]} {res[-1][0]:5.2f}", flush=True)
                # (a GEMM that measures faster on the convolution kernel stays there: [0, 0, 0, 0])
                table["|".join(map(str, key))] = list(res[0][1]) if res[0][0] <= 0.98 * old else [0, 0, 0, 0]
    if a.write:
        with open(path, "w") as f:
            json.dump({"device": "MI355X gfx950", "format": "M|N|K|kind -> [mi, nt, tiles_per_range, waves]; [0, 0, 0, 0] = keep aldm_igemm", "pgemm": table}, f, indent=0, sort_keys=True)
        print("wrote", path)


if __name__ == "__main__":
    main()
