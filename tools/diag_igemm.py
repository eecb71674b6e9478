"""Diagnostic build only: where a wave's cycles go in the igemm LDS-DMA main loop (s_memtime stamps).

    make -C audioldm_with_lora_amd/csrc DIAG=1          # -> audioldm_with_lora_amd/libaldm_hip_diag.so
    python tools/diag_igemm.py

The diag library writes {wait_vmcnt, barrier, dma_issue, mma+lds} cycle sums per wave into the (otherwise unused)
split-K workspace.  Never quote this build's run time -- stamps serialise the loop; read the SHARES."""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ALDM_LIB", os.path.join(ROOT, "audioldm_with_lora_amd", "libaldm_hip_diag.so"))
from audioldm_with_lora_amd import _lib, ops  # noqa: E402

CASES = [("L0 conv 128->128", 8, 250, 16, 128, 128, 3, 1, 3), ("L0 conv 128->128", 8, 250, 16, 128, 128, 3, 3, 2),
         ("L0 conv 128->128", 8, 250, 16, 128, 128, 3, 2, 2), ("L1 conv 256->256", 8, 125, 8, 256, 256, 3, 2, 2),
         ("L1 lin 256->768", 8, 125, 8, 256, 768, 1, 2, 2), ("L3 conv 640->640", 8, 32, 2, 640, 640, 3, 2, 3),
         ("L3 lin 640->640", 8, 32, 2, 640, 640, 1, 2, 2), ("L3 lin 640->640", 8, 32, 2, 640, 640, 1, 2, 4),
         ("L3 lin 640->1920", 8, 32, 2, 640, 1920, 1, 2, 2), ("L3 lin 640->1920", 8, 32, 2, 640, 1920, 1, 2, 4)]


def main():
    dev = "cuda"
    lib = _lib.load()
    real = lib.aldm_igemm
    ws = torch.zeros(1 << 24, dtype=torch.float32, device=dev)

    class Shim:          # same call, but hands the stamp buffer to the kernel through the workspace field
        def __getattr__(self, n):
            return getattr(lib, n)

        @staticmethod
        def aldm_igemm(ref, st):
            ref._obj.workspace = ws.data_ptr()
            return real(ref, st)
    _lib._lib = Shim()
    def report(name, nwg, nkt, bm, bn, ring):
        raw = ws[: nwg * 4 * 8 * 2].view(torch.int64).view(nwg, 4, 8)
        d = raw[:, :, :4].double()
        tot = d.sum(-1)
        share = (d / tot.unsqueeze(-1)).mean((0, 1))
        pro, epi = raw[:, :, 4].double().mean(), raw[:, :, 5].double().mean()
        setup, fill = raw[:, :, 6].double().mean(), raw[:, :, 7].double().mean()
        print(f"{name:18s} tile {bm}x{bn} ring {ring}: {tot.mean() / nkt:6.0f} cycles/K-tile/wave | wait_vmcnt {share[0]:.2f} "
              f"barrier {share[1]:.2f} dma_issue {share[2]:.2f} mma+lds {share[3]:.2f} || per wave: entry->ring fill {setup:.0f} + fill issue {fill:.0f} "
              f"(prologue incl. stamp overhead {pro:.0f})  loop {tot.mean():.0f}  tail+epilogue {epi:.0f} cycles", flush=True)

    # the fused QKV GEMM of an attention module: LoRA side channel (Rp = 32), V^T transposed store, folded LayerNorm
    for (name, B, N, Cc, ring) in (("QKV+LoRA+Vt C256", 8, 1000, 256, 2), ("QKV+LoRA+Vt C384", 8, 252, 384, 2), ("out+LoRA C256", 8, 1000, 256, 3)):
        M = B * N
        x = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
        is_qkv = name.startswith("QKV")
        nout = 3 * Cc if is_qkv else Cc
        w = torch.randn(nout, Cc, device=dev) / math.sqrt(Cc)
        if is_qkv:
            pw = ops.pack_linear_ln(w, None, torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev))
            ops.attach_lora(pw, [(i * Cc, Cc, torch.randn(4, Cc, device=dev) / 16, torch.randn(Cc, 4, device=dev) / 16, 1.0) for i in range(3)])
            xs = x.float().view(M, Cc // 64, 64)
            parts = torch.stack([xs.sum(-1), (xs * xs).sum(-1)], -1).contiguous()
            npad = (N + 7) // 8 * 8
            vt = torch.empty(B, Cc, npad, dtype=torch.bfloat16, device=dev)
            run = lambda: ops.conv(x.view(B, 1, N, Cc), pw, vt=vt, vt_col0=2 * Cc, vt_ld=npad, vt_batch_stride=Cc * npad, ln_parts=parts, tile=2, ring=ring, splits=1)
        else:
            pw = ops.pack_linear(w, torch.zeros(nout, device=dev))
            ops.attach_lora(pw, [(0, Cc, torch.randn(4, Cc, device=dev) / 16, torch.randn(Cc, 4, device=dev) / 16, 1.0)])
            res = torch.randn(M, Cc, device=dev).to(torch.bfloat16)
            run = lambda: ops.linear(x, pw, res=res, tile=2, ring=ring, splits=1)
        ws.zero_()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        report(name, math.ceil(M / 64) * math.ceil(nout / 64), Cc // 64, 64, 64, ring)

    for (name, B, H, W, ci, co, k, tile, ring) in CASES:
        x = torch.randn(B, H, W, ci, device=dev).to(torch.bfloat16)
        pw = ops.pack_conv(torch.randn(co, ci, k, k, device=dev) / math.sqrt(k * k * ci), None)
        ws.zero_()
        for _ in range(3):
            ops.conv(x, pw, pad=(k // 2, k // 2), tile=tile, ring=ring, splits=1)
        torch.cuda.synchronize()
        bm, bn = {1: (128, 128), 2: (64, 64), 3: (128, 64)}[tile]
        nwg = math.ceil(B * H * W / bm) * math.ceil(co / bn)
        report(name, nwg, k * k * ci // 64, bm, bn, ring)


if __name__ == "__main__":
    main()
