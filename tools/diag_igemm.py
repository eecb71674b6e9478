"""Diagnostic build only (make DIAG=1; ALDM_LIB=.../libaldm_hip_diag.so): where a wave's cycles go in the igemm main loop.
Never quote this build's run time -- read the SHARES (stamps serialise the loop)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ALDM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "audioldm_with_lora_amd", "libaldm_hip_diag.so"))
from audioldm_with_lora_amd import ops  # noqa: E402

ops._ws_cache.clear()
dev = "cuda"
for (name, B, H, W, ci, co, k, tile, ring) in [("L0 128->128", 8, 250, 16, 128, 128, 3, 1, 3), ("L0 128->128", 8, 250, 16, 128, 128, 3, 3, 3),
                                               ("L0 128->128", 8, 250, 16, 128, 128, 3, 2, 2), ("L1 256->256", 8, 125, 8, 256, 256, 3, 2, 2),
                                               ("L1 lin 256->768", 8, 125, 8, 256, 768, 1, 2, 4)]:
    x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
    w = torch.randn(co, ci, k, k, device=dev) / math.sqrt(k * k * ci)
    pw = ops.pack_conv(w, None)
    ws = ops._workspace(1 << 26, x.device)
    ws.zero_()
    # splits=1 path never touches the workspace: the diag build writes stamps there; pass it explicitly
    import ctypes as C
    from audioldm_with_lora_amd import _lib
    orig = _lib.load().aldm_igemm

    def hooked(argp, st, _orig=orig, _ws=ws):
        argp._obj.workspace = _ws.data_ptr()
        return _orig(argp, st)
    _lib._lib.aldm_igemm_hook = hooked
    a_saved = ops._lib.load
    y = None
    # call through ops.conv but patch workspace by monkeypatching C.byref result: simplest is to replicate the call
    import types
    real_check = ops.check
    def conv_diag():
        lib = _lib.load()
        real = lib.aldm_igemm
        def wrapped(ref, st):
            ref._obj.workspace = ws.data_ptr()
            return real(ref, st)
        lib_aldm = types.SimpleNamespace(aldm_igemm=wrapped)
        old_load = _lib.load
        _lib.load = lambda: lib_aldm
        try:
            return ops.conv(x, pw, pad=(k // 2, k // 2), tile=tile, ring=ring, splits=1)
        finally:
            _lib.load = old_load
    for _ in range(3):
        y = conv_diag()
    torch.cuda.synchronize()
    M = B * H * W
    bm, bn = {1: (128, 128), 2: (64, 64), 3: (128, 64)}[tile]
    nwg = math.ceil(M / bm) * math.ceil(co / bn)
    d = ws[: nwg * 4 * 4 * 2].view(torch.int64).view(nwg, 4, 4).double()     # uint64 stamps as int64 pairs of fp32 slots
    tot = d.sum(-1)
    share = (d / tot.unsqueeze(-1)).mean((0, 1))
    nkt = k * k * ci // 64
    print(f"{name} tile{tile} ring{ring}: per-wave main-loop cycles {tot.mean():.0f} ({tot.mean() / nkt:.0f}/K-tile)  "
          f"wait_vmcnt {share[0]:.2f}  barrier {share[1]:.2f}  issue {share[2]:.2f}  mma(+lds) {share[3]:.2f}")
