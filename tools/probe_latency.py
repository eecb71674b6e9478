"""What does a kernel's first memory access cost on this part?  (in-kernel s_memtime around dependent loads; one wave)
a: a line of a buffer the PREVIOUS kernel wrote; b: another line of the same 4 KB page; c: a line 64 MB away in a big read-only
buffer nobody touched recently; d: the line next to c."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import _lib
lib = _lib.load()
lib.aldm_probe_latency.argtypes = [ctypes.c_void_p] * 7
lib.aldm_probe_latency.restype = ctypes.c_int
dev = "cuda"
big = torch.zeros(1 << 28, dtype=torch.int32, device=dev)         # 1 GiB
act = torch.zeros(1 << 20, dtype=torch.int32, device=dev)
out = torch.zeros(8, dtype=torch.int64, device=dev)
sink = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
def probe(a, b, c, d, label, prep=None):
    res = []
    for it in range(6):
        if prep is not None:
            prep()
        lib.aldm_probe_latency(a, b, c, d, out.data_ptr(), sink.data_ptr(), st)
        torch.cuda.synchronize()
        t = out.cpu().tolist()
        res.append([t[i + 1] - t[i] for i in range(4)])
    print(f"{label}: cycles per dependent load (a, b, c, d), 6 launches: {res}", flush=True)
P = lambda t, off: t.data_ptr() + 4 * off
probe(P(act, 0), P(act, 512), P(big, 1 << 24), P(big, (1 << 24) + 64), "a/b: just written by the previous kernel (fill); c/d: cold 1 GiB buffer", prep=lambda: act.add_(1))
probe(P(act, 0), P(act, 512), P(big, 1 << 24), P(big, (1 << 24) + 64), "same addresses, nothing in between (warm from the last launch)")
probe(P(big, 3 << 24), P(big, (3 << 24) + (1 << 19)), P(big, 5 << 24), P(big, 7 << 24), "four cold pages of the 1 GiB buffer, 2 MB+ apart")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    act.add_(1)
    lib.aldm_probe_latency(P(act, 0), P(act, 512), P(big, 9 << 24), P(big, (9 << 24) + 64), out.data_ptr(), sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
for _ in range(4):
    g.replay(); torch.cuda.synchronize()
    t = out.cpu().tolist()
    print("in a replayed graph behind a producer kernel:", [t[i + 1] - t[i] for i in range(4)], flush=True)
