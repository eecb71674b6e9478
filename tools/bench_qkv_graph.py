"""In-graph timing of the fused QKV / out projections with and without the LoRA side channel and the V^T store."""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import ops


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


dev = "cuda"
for B, N, Cc in ((8, 1000, 256), (8, 252, 384), (8, 64, 640)):
    M = B * N
    x = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
    res = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
    wq = torch.randn(3 * Cc, Cc, device=dev) / math.sqrt(Cc)
    r = int(os.environ.get("LORA_R", "4"))
    A = [torch.randn(r, Cc, device=dev) / 4 for _ in range(3)]
    Bm = [torch.randn(Cc, r, device=dev) * 0.02 for _ in range(3)]
    npad = (N + 7) // 8 * 8
    vt = torch.empty(B, Cc, npad, device=dev, dtype=torch.bfloat16)
    vkw = dict(vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad)
    p0 = ops.pack_linear(wq, None)
    p1 = ops.pack_linear(wq, None); ops.attach_lora(p1, [(i * Cc, Cc, A[i], Bm[i], 1.0) for i in range(3)])
    o0 = ops.pack_linear(wq[:Cc], torch.zeros(Cc, device=dev))
    o1 = ops.pack_linear(wq[:Cc], torch.zeros(Cc, device=dev)); ops.attach_lora(o1, [(0, Cc, A[0], Bm[0], 1.0)])
    gam, bet = torch.randn(Cc, device=dev) * 0.1 + 1, torch.randn(Cc, device=dev) * 0.1
    p2 = ops.pack_linear_ln(wq, None, gam, bet); ops.attach_lora(p2, [(i * Cc, Cc, A[i], Bm[i], 1.0) for i in range(3)])
    f0 = ops.pack_geglu(torch.randn(8 * Cc, Cc, device=dev) / math.sqrt(Cc), torch.zeros(8 * Cc, device=dev))
    f1 = ops.pack_linear_ln(torch.randn(8 * Cc, Cc, device=dev) / math.sqrt(Cc), torch.zeros(8 * Cc, device=dev), gam, bet, geglu=True)
    xx = x.view(B, 1, N, Cc)
    rr = res.view(B, 1, N, Cc)
    line = f"C={Cc} N={N} r={r}: "
    for name, fn in (("qkv", lambda: ops.conv(xx, p0)), ("qkv+vt", lambda: ops.conv(xx, p0, **vkw)), ("qkv+lora", lambda: ops.conv(xx, p1)),
                     ("qkv+lora+vt", lambda: ops.conv(xx, p1, **vkw)), ("out+res", lambda: ops.conv(xx, o0, res=rr)),
                     ("out+lora+res", lambda: ops.conv(xx, o1, res=rr)), ("LN kernel", lambda: ops.layernorm(x, gam, bet)),
                     ("qkv+lora+vt+LNfold", lambda: ops.conv(xx, p2, **vkw)), ("ff1 geglu", lambda: ops.conv(xx, f0)),
                     ("ff1 geglu+LNfold", lambda: ops.conv(xx, f1))):
        line += f"{name} {timeit(fn):6.2f} us | "
    print(line, flush=True)
