"""In-kernel s_memtime stamps of aldm_pgemm (diagnostic build: make -C audioldm_with_lora_amd/csrc DIAG=1, then
ALDM_LIB=audioldm_with_lora_amd/libaldm_hip_diag.so python tools/diag_pgemm.py): where one workgroup's time goes --
entry -> prologue loads issued -> T = x A^T done (x fragments landed) -> first barrier (tile 0 landed) -> per tile -> end.
Stamps of workgroup 0 and of the last workgroup, wave 0; shader-clock cycles."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audioldm_with_lora_amd import _lib, ops  # noqa: E402
from tune_pgemm import make_case  # noqa: E402


def main():
    lib = _lib.load()
    lib.aldm_pgemm_set_diag.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(128, dtype=torch.int64, device="cuda")
    B = 8
    for C, n in ((256, 1000), (384, 252), (640, 64)):
        M = B * n
        for kind, N in (("proj_in", C), ("qkv", 3 * C), ("out", C), ("ff1", 8 * C)):
            sfx, run = make_case(kind, M, N, C, B, 3)
            for i in range(6):                                  # warm, then the stamped launch is the last one
                run(i % 3)
            torch.cuda.synchronize()
            buf.zero_()
            lib.aldm_pgemm_set_diag(buf.data_ptr())
            run(0)
            torch.cuda.synchronize()
            lib.aldm_pgemm_set_diag(None)
            t = buf.cpu().tolist()
            for off, name in ((0, "wg0"), (64, "wgN")):
                s = t[off:off + 64]
                if s[0] == 0:
                    continue
                tiles = [(s[3 + 2 * k] - s[0], s[4 + 2 * k] - s[0]) for k in range(12) if s[4 + 2 * k]]
                if s[44]:
                    print(f"    tile 2: wait {s[41] - s[40]}  barrier {s[5 + 2] - s[41]}  dma issue {s[43] - s[7]}  main mfma {s[44] - s[43]}  lora+epilogue {s[8] - s[44]}")
                print(f"{kind:8s} M{M} N{N} K{C} {name}: issued {s[1] - s[0]:6d}  T done {s[2] - s[0]:6d}  "
                      + " ".join(f"[{a}-{b}]" for a, b in tiles) + f"  end {s[30] - s[0]:6d}", flush=True)


if __name__ == "__main__":
    main()
