"""Debug helper: where does aldm_attn_block64 differ from the two-launch path?"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops
DEV = "cuda"
def run(B, N, r, seed=31):
    H, d = 8, 80; Cc = 640
    g = torch.Generator().manual_seed(seed)
    bf = lambda t: t.to(torch.bfloat16).float()
    x = bf(torch.randn(B * N, Cc, generator=g) * 1.3 + 0.2)
    wq, wk, wv = (bf(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc)) for _ in range(3))
    gm, bt = torch.randn(Cc, generator=g) * 0.3 + 1, torch.randn(Cc, generator=g) * 0.2
    lor = [(bf(torch.randn(r, Cc, generator=g) / math.sqrt(Cc)), bf(torch.randn(Cc, r, generator=g) * 0.3), 2.0) if r else None for _ in range(3)]
    qs = ops.LOG2E / math.sqrt(d)
    pw = ops.pack_linear_ln(torch.cat([wq * qs, wk, wv]).to(DEV), None, gm.to(DEV), bt.to(DEV))
    ops.attach_lora(pw, [None if l is None else (i * Cc, Cc, l[0].to(DEV), l[1].to(DEV), l[2] * (qs if i == 0 else 1.0)) for i, l in enumerate(lor)])
    xd = x.to(torch.bfloat16).to(DEV)
    xs = xd.float().view(B * N, Cc // 64, 64)
    parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
    out = ops.attn_block64(xd, pw, parts, B, N, H, d).float().cpu()
    npad = (N + 7) // 8 * 8
    vt = torch.zeros(B, Cc, npad, dtype=torch.bfloat16, device=DEV)
    qk = ops.conv(xd.view(B, 1, N, Cc), pw, vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad, ln_parts=parts)
    ref = ops.attention(qk.view(B * N, 2 * Cc), vt, B, N, H, d, prescaled=True).float().cpu()
    bad = ~((out - ref).abs() <= 0.08 + 0.02 * ref.abs())
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
    print(f"B{B} N{N} r{r}: bad {int(bad.sum())} nan {int(out.isnan().sum())} rows {rows[:12]}..{rows[-3:] if rows else ''} ({len(rows)}) cols {cols[:4]}..{cols[-2:] if cols else ''} ({len(cols)})"
          f" maxerr {float((out - ref).abs().nan_to_num().max()):.4f}", flush=True)
def stamps(B=8, N=64, r=4):
    """in-kernel s_memtime stamps (100 MHz) of workgroup (0, 0)"""
    import ctypes
    from audioldm_with_lora_amd import _lib
    lib = _lib.load()
    buf = torch.zeros(24, dtype=torch.int64, device=DEV)
    lib.aldm_attn_block64_set_diag.argtypes = [ctypes.c_void_p]
    lib.aldm_attn_block64_set_diag(buf.data_ptr())
    for _ in range(3):
        run(B, N, r)
    torch.cuda.synchronize()
    t = buf.cpu().tolist()
    lib.aldm_attn_block64_set_diag(None)
    d = [(b - a) for a, b in zip(t, t[1:])]
    print(f"B{B} stamps (cycles): issue0+X issue", d[0], "| X landed", d[1], "| xf read + ring issue", d[2], "| tiles", d[3:20], "| softmax", d[20:22], "| PV+store", d[22],
          "| total", (t[23] - t[0]), flush=True)


for cfg in ((2, 64, 4), (2, 40, 0), (2, 40, 4), (2, 40, 8), (1, 64, 10), (3, 17, 8), (8, 64, 4), (1, 1, 4)):
    run(*cfg)
def timed(ablate, B=8, N=64, r=4, reps=40):
    """graph of `reps` launches, device time per launch"""
    import ctypes
    from audioldm_with_lora_amd import _lib
    lib = _lib.load()
    lib.aldm_attn_block64_set_diag.argtypes = [ctypes.c_void_p]
    H, d, Cc = 8, 80, 640
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(B * N, Cc, generator=g)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(3 * Cc, Cc, generator=g) / 25).to(DEV)
    pw = ops.pack_linear_ln(w, None, torch.ones(Cc, device=DEV), torch.zeros(Cc, device=DEV))
    if r:
        ops.attach_lora(pw, [(i * Cc, Cc, torch.randn(r, Cc, device=DEV) / 25, torch.randn(Cc, r, device=DEV) / 10, 1.0) for i in range(3)])
    xs = x.float().view(B * N, Cc // 64, 64)
    parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
    for _ in range(3):
        ops.attn_block64(x, pw, parts, B, N, H, d)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            ops.attn_block64(x, pw, parts, B, N, H, d)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    lib.aldm_attn_block64_set_diag(None)
    print(f"ablate {ablate:2d} (1 gload 2 lstore 4 ds_read 8 mfma 32 barrier): {best * 1e3:7.2f} us / launch", flush=True)


timed(0)
timed(0, r=0)
timed(0, r=8)
timed(0, B=16)
