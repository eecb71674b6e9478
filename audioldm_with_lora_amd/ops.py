"""Tensor-level wrappers over the C-ABI (include/aldm_hip.h).

torch is used for device memory and the current stream only; all arithmetic happens in the HIP
kernels.  Activations are channels-last bf16 tensors: images [B, H, W, C], token sequences [B*N, C].
"""
import ctypes as C
import json
import math
import os
import sys
from dataclasses import dataclass
from typing import Optional

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LRELU, ACT_NONE, ACT_SILU, ACT_TANH, OUT_BF16, OUT_F32, IgemmArgs, check)

BK = 64


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _require_gpu(t):
    if not t.is_cuda:
        raise _lib.AldmError("the product path runs on the MI355X only (got a CPU tensor); there is no CPU fallback")


@dataclass
class PackedW:
    """bf16 weights packed for aldm_igemm: w [N][Kpad] (K = (kh*KW+kw)*Cin + c), fp32 bias."""
    w: torch.Tensor
    bias: Optional[torch.Tensor]
    N: int
    Cin: int            # total input channels (both concat sources)
    KH: int = 1
    KW: int = 1
    geglu: bool = False
    lora_a: Optional[torch.Tensor] = None      # [Rp][Kpad] bf16
    lora_b: Optional[torch.Tensor] = None      # [N][Rp] bf16, pre-scaled by alpha/r
    Rp: int = 0
    ln_s: Optional[torch.Tensor] = None        # LayerNorm folded into the GEMM: fp32 [N] column sums of W diag(gamma)
    ln_sa: Optional[torch.Tensor] = None       # fp32 [Rp]: row sums of A diag(gamma)
    ln_ca: Optional[torch.Tensor] = None       # fp32 [Rp]: A beta
    ln_eps: float = 1e-5
    Cext: int = 0                               # channels of the fused 1x1 second-source segment appended to every weight row

    @property
    def Kpad(self):
        return self.w.shape[1]


def _pad_k(w2d: torch.Tensor) -> torch.Tensor:
    n, k = w2d.shape
    kp = (k + BK - 1) // BK * BK
    out = torch.zeros(n, kp, dtype=torch.bfloat16, device=w2d.device)
    out[:, :k] = w2d.to(torch.bfloat16)
    return out.contiguous()


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedW:
    """Conv2d weight [Cout, Cin, KH, KW] -> [Cout][(kh, kw, cin)]."""
    co, ci, kh, kw = weight.shape
    w = _pad_k(weight.detach().permute(0, 2, 3, 1).reshape(co, kh * kw * ci))
    return PackedW(w, None if bias is None else bias.detach().float().contiguous(), co, ci, kh, kw)


def pack_conv_shortcut(weight, bias, sc_weight, sc_bias) -> PackedW:
    """ResnetBlock2D.conv2 [Cout, C, 3, 3] and conv_shortcut [Cout, Cin, 1, 1] as ONE weight matrix [Cout][(kh, kw, c) | cin]:
    conv2(h) + conv_shortcut(x) is then a single implicit GEMM whose K loop ends with a 1x1 segment over the block input
    (aldm_igemm x3 / x4).  Needs C % 64 == 0 and Cin % 64 == 0 (the LDS-DMA path); the caller falls back to two launches otherwise."""
    co, ci, kh, kw = weight.shape
    w2 = weight.detach().permute(0, 2, 3, 1).reshape(co, kh * kw * ci)
    ws = sc_weight.detach().reshape(co, -1)
    pw = PackedW(_pad_k(torch.cat([w2, ws], dim=1)), (bias.detach().float() + sc_bias.detach().float()).contiguous(), co, ci, kh, kw)
    pw.Cext = ws.shape[1]
    return pw


def pack_conv1d(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedW:
    """Conv1d weight [Cout, Cin, K] -> KH = 1, KW = K."""
    return pack_conv(weight.unsqueeze(2), bias)


def pack_linear(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PackedW:
    n, k = weight.shape
    return PackedW(_pad_k(weight.detach()), None if bias is None else bias.detach().float().contiguous(), n, k)


def pack_geglu(weight: torch.Tensor, bias: torch.Tensor) -> PackedW:
    """GEGLU.proj [2*inner, C]: rows re-ordered into blocks of (16 value | 16 gate) so that a lane holds
    a value and its gate in adjacent MFMA tiles (value = first half, gate = second half)."""
    two_inner, k = weight.shape
    inner = two_inner // 2
    assert inner % 16 == 0
    idx = torch.arange(inner, device=weight.device).view(-1, 16)
    order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
    pw = PackedW(_pad_k(weight.detach()[order]), bias.detach().float()[order].contiguous(), two_inner, k)
    pw.geglu = True
    return pw


def pack_linear_ln(weight, bias, gamma, beta, eps=1e-5, geglu=False):
    """Linear applied to LayerNorm(x) with the norm folded in:  LN(x) W^T + b = rstd (x W'^T - mean s) + c,
    W' = W diag(gamma) (bf16), s = row sums of the bf16 W' (what the MFMA really sums), c = W beta + b."""
    w = weight.detach().float()
    g, bt = gamma.detach().float(), beta.detach().float()
    wp = (w * g[None, :]).to(torch.bfloat16)
    s = wp.float().sum(1)
    c = w @ bt + (bias.detach().float() if bias is not None else 0.0)
    if geglu:
        inner = w.shape[0] // 2
        idx = torch.arange(inner, device=w.device).view(-1, 16)
        order = torch.cat([idx, idx + inner], dim=1).reshape(-1)
        wp, s, c = wp[order], s[order], c[order]
    pw = PackedW(_pad_k(wp), c.contiguous(), w.shape[0], w.shape[1])
    pw.geglu = geglu
    pw.ln_s, pw.ln_eps = s.contiguous(), eps
    pw._ln = (g, bt)
    return pw


def attach_lora(pw: PackedW, parts):
    """parts: list of (row_offset, n_rows, A [r, K], B [n_rows, r], scaling) -- one per LoRA-wrapped target
    that shares this GEMM.  Builds A_cat [Rp][Kpad] and the block-structured, pre-scaled B_ext [N][Rp]."""
    parts = [p for p in parts if p is not None]
    if not parts:
        pw.lora_a = pw.lora_b = None
        pw.Rp = 0
        return pw
    rtot = sum(p[2].shape[0] for p in parts)
    rp = 32 if rtot <= 32 else 64
    if rtot > 64:
        raise _lib.AldmError(f"fused LoRA supports a combined rank <= 64 per GEMM (got {rtot})")
    dev = pw.w.device
    a = torch.zeros(rp, pw.Kpad, dtype=torch.bfloat16, device=dev)
    b = torch.zeros(pw.N, rp, dtype=torch.bfloat16, device=dev)
    ln = getattr(pw, "_ln", None) if pw.ln_s is not None else None
    sa = torch.zeros(rp, dtype=torch.float32, device=dev)
    ca = torch.zeros(rp, dtype=torch.float32, device=dev)
    col = 0
    for row0, nrows, A, Bm, s in parts:
        r, k = A.shape
        Af = A.detach().float()
        if ln is not None:                      # LayerNorm folded: A' = A diag(gamma), sA = row sums, cA = A beta
            Ap = (Af * ln[0][None, :]).to(torch.bfloat16)
            sa[col:col + r] = Ap.float().sum(1)
            ca[col:col + r] = Af @ ln[1]
        else:
            Ap = Af.to(torch.bfloat16)
        a[col:col + r, :k] = Ap
        b[row0:row0 + nrows, col:col + r] = (Bm.detach().float() * s).to(torch.bfloat16)
        col += r
    pw.lora_a, pw.lora_b, pw.Rp = a.contiguous(), b.contiguous(), rp
    pw.ranks_used = col
    if ln is not None:
        pw.ln_sa, pw.ln_ca = sa, ca
    return pw


def auto_splits(M, N, ktiles):
    """Split-K heuristic (tools/bench_igemm.py sweep): the low-resolution UNet levels have few 64x64 output
    tiles but deep K; split until ~512 workgroups are in flight, at most 4 ways, >= 8 K-tiles per split."""
    tiles = math.ceil(M / 64) * math.ceil(N / 64)
    if tiles >= 256 or ktiles < 16:
        return 1
    return max(1, min(4, math.ceil(512 / tiles), ktiles // 8))


def pick_tile(M, N):
    """Tile heuristic from tools/bench_igemm.py sweeps on MI355X: the 8-wave 128x128 tile once it still yields ~2
    workgroups per CU (256 CUs), 128x64 down to ~2 per CU, else 64x64 (more, smaller workgroups hide the short K loops
    of the low-resolution levels)."""
    if N >= 256 and N % 128 == 0 and math.ceil(M / 128) * (N // 128) >= 448:
        return 6
    if N > 64 and math.ceil(M / 128) * math.ceil(N / 64) >= 448:
        return _lib.TILE_128x64
    if N <= 64 and M >= 4096:
        return _lib.TILE_128x64
    return _lib.TILE_64x64


def pick_ring(tile, bm, bn, rp, nwg, ktiles):
    """LDS-DMA ring depth (2..4 K-tiles in flight).  Rule from tools/bench_igemm.py --ring sweeps on MI355X:
    residency first -- the ring must leave room for ceil(nwg / 256 CUs) workgroups per CU inside the 160 KiB of LDS,
    otherwise the grid runs in two rounds (128x64 tile, 500 workgroups: ring 4 = 98 KiB/WG -> 38.9 us, ring 3 -> 23.9 us);
    then depth -- few workgroups (low-resolution levels) have nothing else to hide the weight stream's latency behind, so
    they take the deepest ring that fits; big grids gain nothing past 2-3."""
    if tile in (6, 9):
        return 2
    if tile == 12:
        return 3
    stage = (bm + bn + rp) * 128
    extra = (bn * 128 if rp else 0) + 2 * bm * 4
    per_cu = min(8, max(1, math.ceil(nwg / 256)))
    fit = ((160 * 1024) // per_cu - extra) // stage
    want = 4 if nwg <= 512 else (2 if ktiles <= 24 else 3)
    return max(2, min(want, fit))


TILE_DIMS = {1: (128, 128), 2: (64, 64), 3: (128, 64), 4: (64, 128), 6: (128, 128), 7: (128, 128), 8: (64, 128), 9: (256, 128), 10: (64, 128), 11: (128, 64), 12: (256, 128), 13: (64, 128), 14: (128, 64), 15: (128, 128), 16: (64, 128)}
HALO_ROWS = {7: 320, 8: 128, 15: 320, 16: 128}          # LDS halo capacity (rows) of the two halo tiles (csrc/igemm_halo.hip; 128x128: three or five DMA passes)


def halo_tiles(OW, eligible):
    """Halo tiles able to take a 3x3/s1/p1 conv whose output is OW wide: BM a multiple of OW and the halo within capacity."""
    if not eligible:
        return []
    return [t for t in (7, 8, 15, 16) if TILE_DIMS[t][0] % OW == 0 and (TILE_DIMS[t][0] // OW + 2) * (OW + 2) <= HALO_ROWS[t]]

# ---- measured launch configurations ------------------------------------------------------------------------------
# "Measure, don't guess": tools/autotune.py picks (tile, ring, splits) per distinct GEMM in two stages and writes
# tuned_gfx950.json next to this file.  Stage 1 times every valid triple on the live operands in isolation and keeps a
# shortlist; stage 2 replays the whole step (all launches queued behind a sleep kernel, so they run back to back as in the
# captured graph) once per shortlist slot and times each GEMM IN CONTEXT -- repeated isolated launches see warm caches
# and mis-rank configurations (a table built from stage 1 alone measured 4.64 ms/step against 4.52 for the heuristics).
# At run time the table is only looked up; GEMMs it does not hold fall back to the heuristics.
TUNED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")
TUNED = {}
TUNED_LEGACY_KEYS = False
TUNER = None
if os.path.exists(TUNED_PATH) and os.environ.get("ALDM_NO_TUNED") != "1":
    with open(TUNED_PATH) as _f:
        TUNED = {k: tuple(v) for k, v in json.load(_f)["igemm"].items()}
if os.environ.get("ALDM_TUNED_PATCH"):                           # A/B aid: a JSON {key: [tile, ring, splits]} laid over the table
    with open(os.environ["ALDM_TUNED_PATCH"]) as _f:
        TUNED.update({k: tuple(v) for k, v in json.load(_f).items()})
TUNED_LEGACY_KEYS = any(" k3x3 " in k and " ow" not in k for k in TUNED)


def tune_key(M, N, c1, c2, kh, kw, stride, up, dilate, rp, vt, geglu, ln, fast, ow=None, pad=None, dil=None):
    """Key of one GEMM in the measured table.  The geometry suffix (output width, padding, filter dilation) is part of the key for
    every filter larger than 1x1: two convolutions with equal M / N / C can differ in OW (UNet level 0 is 16 wide, the VAE
    decoder's last level 64) and then differ in which tiles are even legal (the halo tiles need BM % OW == 0)."""
    key = (f"M{M} N{N} C{c1}+{c2} k{kh}x{kw} s{stride[0]} up{int(up)} d{dilate} r{rp} vt{int(vt)} g{int(bool(geglu))} "
           f"ln{int(ln)} f{int(fast)}")
    if ow is not None and (kh > 1 or kw > 1):
        key += f" ow{ow} p{pad[0]}x{pad[1]} dl{dil[0]}x{dil[1]}"
    return key


def heuristic_cfg(M, pw, ktiles, can_split, fast_path, has_vt, splits=None):
    if splits is None:
        splits = auto_splits(M, pw.N, ktiles) if can_split else 1
    tile = _lib.TILE_64x64 if pw.Rp else pick_tile(M, pw.N)      # LoRA GEMMs are short-K: favour many workgroups
    if tile == 6 and (not fast_path or has_vt or ktiles < 16):
        tile = _lib.TILE_128x64           # the 8-wave tile only exists on the LDS-DMA path and only pays on deep K loops
    bm, bn = TILE_DIMS[tile]
    nwg = math.ceil(M / bm) * math.ceil(pw.N / bn) * max(1, splits)
    return tile, pick_ring(tile, bm, bn, pw.Rp, nwg, ktiles), splits


class Tuner:
    """Two-stage launch-configuration search driven by tools/autotune.py."""

    def __init__(self, shortlist=5, reps=4, verbose=False):
        self.shortlist, self.reps, self.verbose = shortlist, reps, verbose
        self.cands = {}          # key -> [cfg, ...]   (slot 0 = heuristic)
        self.times = {}          # key -> {cfg: [event pairs]}
        self.slot = None         # stage 2: which shortlist slot this pass runs

    # ---- stage 1 ----
    def isolated(self, a, key, M, pw, ktiles, can_split, forced_splits, fast_path, has_vt, device, halo=()):
        lib = _lib.load()
        default = heuristic_cfg(M, pw, ktiles, can_split, fast_path, has_vt, forced_splits)
        cands = []
        for t in [2, 3, 1, 4] + ([6] if (fast_path and not has_vt and ktiles >= 8) else []) + (
                [9, 12] if (fast_path and not has_vt and not pw.Rp and ktiles >= 8 and M >= 32768 and pw.N % 128 == 0 and pw.ln_s is None
                            and not pw.geglu and not pw.Cext) else []) + (
                [10, 11] if (fast_path and not has_vt and not pw.Rp and ktiles >= 8 and M <= 8192) else []) + (
                [13, 14] if (fast_path and not has_vt and not pw.Rp and ktiles >= 8 and M <= 8192 and pw.ln_s is None and not pw.geglu) else []):
            bm, bn = TILE_DIMS[t]
            if has_vt and a.vt_col0 % bn:
                continue
            base = math.ceil(M / bm) * math.ceil(pw.N / bn)
            sp_list = [forced_splits or 1]
            if can_split and base <= 640:
                sp_list += [sp for sp in (2, 3, 4, 6, 8, 12, 16)
                            if sp <= ktiles // 2 and base * sp <= 2560 and sp * M * pw.N * 4 <= (1 << 28)]
            for sp in sp_list:
                for rg in ((3,) if t == 12 else (3, 4) if t in (13, 14) else (2, 3) if t in (6, 9) else (2, 3, 4)):
                    cands.append((t, rg, sp))
        if forced_splits in (None, 1):
            cands += [(t, rg, 1) for t in halo for rg in (2, 3, 4)]
            if can_split and " gi" not in key:              # the halo tiles split by 64-channel chunk (csrc/igemm_halo.hip)
                nch = ktiles // 9
                for t in halo:
                    base = math.ceil(M / TILE_DIMS[t][0]) * math.ceil(pw.N / TILE_DIMS[t][1])
                    cands += [(t, rg, sp) for sp in (2, 3, 4, 5, 6, 8, 10) for rg in (3, 4)
                              if sp <= nch and base * sp <= 2560 and sp * M * pw.N * 4 <= (1 << 28)]
        max_sp = max(c[2] for c in cands)
        ws = _workspace(max_sp * M * pw.N * 4, device) if max_sp > 1 else None
        results = []
        for t, rg, sp in cands:
            a.tile, a.ring, a.splits = t, rg, sp
            a.workspace = ws.data_ptr() if sp > 1 else None
            a.defer_reduce = 1 if (sp > 1 and key.endswith(" gn")) else 0   # the consumer GroupNorm pays the reduce
            if lib.aldm_igemm(C.byref(a), _stream()) != 0:              # warm-up doubles as the validity check
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sleep_us(60)                                                 # let the timed launches queue up behind it
            e0.record()
            for _ in range(self.reps):
                lib.aldm_igemm(C.byref(a), _stream())
            e1.record()
            results.append(((t, rg, sp), e0, e1))
        torch.cuda.synchronize()
        ranked = sorted(results, key=lambda r: r[1].elapsed_time(r[2]))
        short = [default] + [r[0] for r in ranked if r[0] != default][: self.shortlist]
        self.cands[key] = short
        if self.verbose:
            print(f"[autotune] {key}: isolated best {ranked[0][0]} {ranked[0][1].elapsed_time(ranked[0][2]) / self.reps * 1e3:.1f} us, "
                  f"default {default}, {len(results)} valid", flush=True)
        return default

    # ---- called by conv() ----
    def choose(self, key, *args):
        if key not in self.cands:
            return self.isolated(args[0], key, *args[1:])
        c = self.cands[key]
        return c[self.slot % len(c)] if self.slot is not None else c[0]

    def record(self, key, cfg, e0, e1):
        self.times.setdefault(key, {}).setdefault(cfg, []).append((e0, e1))

    # ---- stage 2 result ----
    def finish(self):
        torch.cuda.synchronize()
        out = {}
        for key, per in self.times.items():
            avg = {cfg: sum(a.elapsed_time(b) for a, b in evs) / len(evs) for cfg, evs in per.items()}
            best = min(avg, key=avg.get)
            default = self.cands[key][0]
            # keep the heuristic unless the measured gain is clear (>3 %): the table then only lists real wins
            if default in avg and avg[best] > 0.97 * avg[default]:
                best = default
            out[key] = (best, avg[best], avg.get(default))
        return out


def save_tuned(path=TUNED_PATH):
    with open(path, "w") as f:
        json.dump({"device": "MI355X gfx950", "format": "key -> [tile, ring, splits]",
                   "igemm": {k: list(v) for k, v in sorted(TUNED.items())}}, f, indent=0)


TILE_NAMES = {1: "128x128", 2: "64x64", 3: "128x64", 4: "64x128", 6: "128x128w8", 7: "halo128x128", 8: "halo64x128", 9: "256x128w8", 10: "64x128w8", 11: "128x64w8", 12: "256x128ws", 13: "64x128ws", 14: "128x64ws", 15: "halo128x128ws", 16: "halo64x128ws"}

# Optional launch profiler (bench.py): a list that receives (label, flops, bytes, start_event, end_event, site) per C-ABI call.
# Events are recorded on the stream the kernel is launched on.  SITE tags the launches of one fused-LoRA attention module
# (unet.run_attention) so that bench.py can price the module as a whole.
PROFILE = None
SITE = None
KEYLOG = None        # with PROFILE: (row index, (tuning-table key, (tile, ring, splits) used)) of every igemm launch (tools/ab_overlay.py)


def _launch(label, flops, nbytes, fn):
    if PROFILE is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    rc = fn()
    e.record()
    PROFILE.append((label, flops, nbytes, s, e, SITE))
    return rc


_ws_cache = {}


def _workspace(nbytes, device):
    key = (device, torch.cuda.current_stream().cuda_stream)
    t = _ws_cache.get(key)
    if t is None or t.numel() * 4 < nbytes:
        t = torch.empty(max(nbytes // 4, 1 << 22), dtype=torch.float32, device=device)
        _ws_cache[key] = t
    return t


GN_FINALIZE_MIN_TILES = 96          # = ALDM_GN_FINALIZE_MIN_TILES (csrc/norm.hip)


@dataclass
class QStats:
    """GroupNorm statistics a convolution left next to its output (aldm_igemm qstat_out): per M-tile, image slot and 4-channel quad
    the (sum, sum of squares) of the stored values.  Travels as the attribute `.qstats` of the output tensor; groupnorm() then
    runs as one coalesced apply pass (aldm_groupnorm_apply).  bm: rows per generic M-tile; tpi > 0: image-aligned halo tiles."""
    table: torch.Tensor
    bm: int
    tpi: int


# Stand-alone GroupNorms read the statistics their producer handed over where that pays: big images (the strip-per-workgroup kernel
# is L2-request-bound there) -- below this many pixels per image the register-resident kernel is as fast (HW = 1000: 6.7 us against
# 10.0 us for fold + apply) and the producers' epilogues stay free of the statistics code.
QSTATS_MIN_HW = 2048


# aldm_igemm_t.xcd_map: 0 = the library decides per launch which operand an XCD's L2 keeps (weight-stationary where the weight matrix
# is the larger operand), 1 / 2 force activation- / weight-stationary (A/B measurements only).
XCD_MAP = int(os.environ.get("ALDM_XCD_MAP", "0"))


class Deferred:
    """A split-K convolution whose reduce is still pending (conv(..., defer=True)): the fp32 partial tiles sit in the shared
    workspace, `out` is the bf16 tensor the consumer will fill.  The ONLY valid consumer is the next groupnorm() call on this
    stream -- it sums the tiles, adds bias / row bias / residual, writes `out` and returns the norm; no other split-K conv may
    run in between (the workspace is shared; conv() refuses)."""
    __slots__ = ("out", "ws", "eff", "bias", "rowbias", "rowbias_ld", "res", "keep")

    def __init__(self, out, ws, eff, bias, rowbias, rowbias_ld, res, keep):
        global DEFERRED_COUNT
        DEFERRED_COUNT += 1
        self.out, self.ws, self.eff, self.bias, self.rowbias, self.rowbias_ld, self.res = out, ws, eff, bias, rowbias, rowbias_ld, res
        self.keep = keep                                   # tensors the pointers above refer to

    @property
    def shape(self):
        return self.out.shape


DEFERRED_COUNT = 0                                         # diagnostics / tests: how many split-K launches deferred their reduce


class _PendingMap(dict):
    """(device, stream) -> the Deferred whose partial tiles currently own THAT stream's split-K workspace.  The workspace is per
    (device, stream) (_workspace), so the producer -> consumer hand-over is too: two engines on two streams (or two devices) never
    see each other's pending reduce."""

    def key(self, t):
        return (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)


_PENDING_BY_STREAM = _PendingMap()


def _pending_get(t):
    return _PENDING_BY_STREAM.get(_PENDING_BY_STREAM.key(t))


def _pending_set(t, d):
    k = _PENDING_BY_STREAM.key(t)
    if d is None:
        _PENDING_BY_STREAM.pop(k, None)
    else:
        _PENDING_BY_STREAM[k] = d


def drop_pending(t=None):
    """Forget a deferred reduce nobody consumed (an exception between producer and consumer): called at the top of a forward with
    that forward's input (its device / current stream); without an argument, every stream's."""
    if t is None:
        _PENDING_BY_STREAM.clear()
    else:
        _pending_set(t, None)


def tensor_of(x):
    """the bf16 tensor behind a conv result (for a Deferred: valid on the stream once its consumer norm has been launched)"""
    return x.out if isinstance(x, Deferred) else x


def conv(x: torch.Tensor, pw: PackedW, *, x2: Optional[torch.Tensor] = None, stride=(1, 1), pad=(0, 0), dil=(1, 1),
         up_size=None, in_dilate=0, out_hw=None, in_act=ACT_NONE, in_slope=0.0, rowbias=None, rowbias_ld=0, out_act=ACT_NONE,
         out_slope=0.0, res=None, res2=None, alpha=1.0, post_act=ACT_NONE, post_slope=0.0, out2=None, out=None, out_f32=False, out_ld=None, out_batch_stride=None,
         out_pix_stride=1, out_pix_offset=0, vt=None, vt_col0=0, vt_ld=0, vt_batch_stride=0, lora_t_out=None,
         splits=None, tile=0, ring=0, gn=None, gn_keep=False, defer=False, rowstats=False, ln_parts=None, x3=None, x4=None,
         vt_dual=False, qstats=False, gn_in=None):
    """Implicit-GEMM convolution over channels-last x [B, IH, IW, C1] (+ x2 [B, IH, IW, C2]).

    gn=(gamma, beta, groups, eps, act) returns GroupNorm(+act) of the convolution instead of the convolution: when the launch
    is split-K the partial tiles are summed by the GroupNorm kernel itself (aldm_groupnorm_partials) and the convolution's
    bf16 output is never written (ResnetBlock2D.conv1 -> norm2 -> SiLU).  gn_keep=True returns (convolution, its GroupNorm)
    -- ResnetBlock2D.conv2 (+ shortcut `res`) in front of a Transformer2DModel's norm: the block output stays on the residual
    stream and the fused kernel writes both.  defer=True: the same for a norm that is launched later by the caller -- returns
    a Deferred (see there) when the launch is split-K, else the tensor as usual.

    gn_in=(gamma, beta, groups, eps, act): x (| x2) are RAW convolution outputs carrying `.qstats`; their GroupNorm (+ act) is applied
    inside this launch, on the halo tile's way into LDS (gn_in_ok() says whether a launch can take it; a halo tile is then forced).

    LayerNorm hand-over (BasicTransformerBlock: h -> LayerNorm -> projection): rowstats=True returns (out, stats), stats fp32
    [M, N / BN, 2] = per-row partial (sum, sum of squares) written by the epilogue; a consumer packed with pack_linear_ln takes
    them as ln_parts= and derives mean / rstd from them -- no LayerNorm launch, no statistics pass in either GEMM's K loop."""
    _require_gpu(x)
    if _pending_get(x) is not None:
        raise _lib.AldmError("conv: a deferred split-K reduce is pending on this stream -- its consumer groupnorm() must be the next launch")
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4
    B, IH, IW, C1 = x.shape
    C2 = 0
    if x2 is not None:
        assert x2.dtype == torch.bfloat16 and x2.is_contiguous() and x2.shape[:3] == x.shape[:3]
        C2 = x2.shape[3]
    assert C1 + C2 == pw.Cin, f"conv: input channels {C1}+{C2} != packed {pw.Cin}"
    C3 = x3.shape[3] if x3 is not None else 0
    C4 = x4.shape[3] if x4 is not None else 0
    assert C3 + C4 == pw.Cext, f"conv: fused 1x1 segment channels {C3}+{C4} != packed {pw.Cext}"
    KH, KW = pw.KH, pw.KW
    VH, VW = (up_size if up_size is not None else (IH, IW))
    if out_hw is None:
        OH = (VH + 2 * pad[0] - dil[0] * (KH - 1) - 1) // stride[0] + 1
        OW = (VW + 2 * pad[1] - dil[1] * (KW - 1) - 1) // stride[1] + 1
    else:
        OH, OW = out_hw
    ncols = pw.N // 2 if pw.geglu else pw.N
    if vt is not None and not vt_dual:
        ncols = vt_col0
    if out_ld is None:
        out_ld = ncols
    if out is None:
        out = torch.empty(B, OH, OW, out_ld, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    if out_batch_stride is None:
        assert out_pix_stride == 1 and out_pix_offset == 0, "strided output rows need an explicit batch stride"
        out_batch_stride = OH * OW * out_ld
    if (PGEMM and tile == 0 and ring == 0 and KH == 1 and KW == 1 and stride == (1, 1) and pad == (0, 0) and up_size is None
            and not in_dilate and in_act == ACT_NONE and x2 is None and x3 is None and pw.Kpad == C1 and C1 in PGEMM_K
            and pw.N % 64 == 0 and out.dtype == torch.bfloat16 and out_pix_stride == 1 and out_pix_offset == 0
            and out_batch_stride == OH * OW * out_ld and out_ld % 8 == 0 and rowbias is None and out_act == ACT_NONE
            and post_act == ACT_NONE and res2 is None and out2 is None and alpha == 1.0 and (lora_t_out is None or pw.Rp)
            and splits in (None, 1) and gn is None and (not vt_dual or vt is not None)
            and not (qstats and OH * OW >= QSTATS_MIN_HW)
            and (not pw.Rp or getattr(pw, "ranks_used", 99) <= 32)
            and (pw.ln_s is None or (ln_parts is not None and ln_parts.shape[1] <= 16))
            and not (pw.geglu and (res is not None or vt is not None or pw.Rp or rowstats))
            and not (vt is not None and (res is not None or rowstats or vt_col0 % 32 or (vt_col0 <= 0 and not vt_dual) or vt_col0 < 0))
            and (res is None or (res.is_contiguous() and res.numel() == B * OH * OW * out_ld))
            and PGEMM_CFG.get(_pgemm_key(B * OH * OW, pw, C1, res, vt, rowstats, lora_t_out is not None, vt_dual)) != PGEMM_USE_IGEMM):
        return _pgemm(x, pw, out, out_ld, B * OH * OW, C1, OH * OW, res, vt, vt_col0, vt_ld, vt_batch_stride, rowstats, ln_parts, lora_t_out, vt_dual)
    a = IgemmArgs()
    if x3 is not None:
        for t in (x3, x4):
            assert t is None or (t.dtype == torch.bfloat16 and t.is_contiguous() and tuple(t.shape[:3]) == (B, OH, OW))
        a.x3, a.Cin3 = x3.data_ptr(), C3
        if x4 is not None:
            a.x4, a.Cin4 = x4.data_ptr(), C4
    a.x, a.x2 = x.data_ptr(), (x2.data_ptr() if x2 is not None else None)
    a.B, a.IH, a.IW, a.Cin, a.Cin2 = B, IH, IW, C1, C2
    a.UH, a.UW = (up_size if up_size is not None else (0, 0))
    a.in_dilate = in_dilate
    a.w, a.Kpad = pw.w.data_ptr(), pw.Kpad
    a.KH, a.KW = KH, KW
    a.stride_h, a.stride_w = stride
    a.pad_h, a.pad_w = pad
    a.dil_h, a.dil_w = dil
    a.OH, a.OW, a.Cout = OH, OW, pw.N
    a.in_act, a.in_slope = in_act, in_slope
    if pw.Rp:
        a.lora_a, a.lora_b, a.Rp = pw.lora_a.data_ptr(), pw.lora_b.data_ptr(), pw.Rp
        a.lora_t_out = lora_t_out.data_ptr() if lora_t_out is not None else None
    a.bias = pw.bias.data_ptr() if pw.bias is not None else None
    if pw.ln_s is not None:
        a.ln_s, a.ln_eps = pw.ln_s.data_ptr(), pw.ln_eps
        if pw.Rp:
            a.ln_sa, a.ln_ca = pw.ln_sa.data_ptr(), pw.ln_ca.data_ptr()
        if ln_parts is not None:
            assert ln_parts.dtype == torch.float32 and ln_parts.is_contiguous() and ln_parts.dim() == 3 and ln_parts.shape[2] == 2
            assert ln_parts.shape[0] == B * IH * IW and KH == 1 and KW == 1, "ln_parts: one row of partials per GEMM row"
            a.ln_parts, a.ln_nparts = ln_parts.data_ptr(), ln_parts.shape[1]
    elif ln_parts is not None:
        raise _lib.AldmError("conv: ln_parts without a LayerNorm-folded weight pack (pack_linear_ln)")
    if rowstats:
        if vt is not None or pw.geglu or out_f32 or out2 is not None or pw.N % 64 or gn is not None or defer:
            raise _lib.AldmError("conv: rowstats needs the standard bf16 epilogue and N % 64 == 0")
        if splits is None:
            splits = 1                                   # the statistics come out of the (single) epilogue
    if rowbias is not None:
        assert rowbias.dtype == torch.float32
        a.rowbias, a.rowbias_ld = rowbias.data_ptr(), rowbias_ld
    a.geglu = 1 if pw.geglu else 0
    a.out_act, a.out_slope = out_act, out_slope
    a.res = res.data_ptr() if res is not None else None
    a.res2 = res2.data_ptr() if res2 is not None else None
    a.alpha = alpha
    a.post_act, a.post_slope = post_act, post_slope
    a.out2 = out2.data_ptr() if out2 is not None else None
    a.out, a.out_dtype, a.out_ld = out.data_ptr(), (OUT_F32 if out.dtype == torch.float32 else OUT_BF16), out_ld
    a.out_batch_stride = out_batch_stride
    a.out_pix_stride, a.out_pix_offset = out_pix_stride, out_pix_offset
    if vt is not None:
        a.vt, a.vt_col0, a.vt_ld, a.vt_batch_stride = vt.data_ptr(), vt_col0, vt_ld, vt_batch_stride
        a.vt_dual = 1 if vt_dual else 0
    M = B * OH * OW
    ktiles = pw.Kpad // BK
    can_split = not (vt is not None or pw.N % 4 or pw.ln_s is not None)
    gn_defer = ((gn is not None or defer) and can_split and not pw.geglu and res2 is None and out2 is None
                and out_act == ACT_NONE and post_act == ACT_NONE and alpha == 1.0 and out.dtype == torch.bfloat16
                and out_ld == pw.N and out_pix_stride == 1)
    if gn_defer and gn is not None:
        gn_defer = pw.N % gn[2] == 0 and (pw.N // gn[2]) % 4 == 0 and OH * OW * (pw.N // gn[2] // 4) <= 4096
    elif gn_defer:                                         # defer = (channels, groups) of the consuming norm (it may append a skip tensor)
        ct, ng = defer if isinstance(defer, tuple) else (pw.N, 32)
        cg = ct // ng
        gn_defer = ct % ng == 0 and cg % 4 == 0 and pw.N % cg == 0 and OH * OW * (cg // 4) <= 4096
    fast_path = not (pw.Cin % 64 or (x2 is not None and x2.shape[3] % 64) or in_act)
    tuning = False
    key = None
    if tile == 0 and ring == 0:
        # launch configuration: the measured table (tuned_gfx950.json, written by tools/autotune.py) where it has this
        # GEMM, else the heuristics below.  A caller-fixed split count stays fixed (it is part of the key).
        sfx = (("" if splits is None else f" sp{splits}") + (" gn" if gn_defer else "") + (" rs" if rowstats else "")
               + (" lp" if ln_parts is not None else "") + (f" e{C3}+{C4}" if x3 is not None else "") + (" vd" if vt_dual else "")
               + (" gi" if gn_in is not None else ""))
        key = tune_key(M, pw.N, C1, C2, KH, KW, stride, up_size is not None, in_dilate, pw.Rp, vt is not None, pw.geglu,
                       pw.ln_s is not None, fast_path, OW, pad, dil) + sfx
        halo = halo_tiles(OW, KH == 3 and KW == 3 and stride == (1, 1) and pad == (1, 1) and dil == (1, 1) and fast_path
                          and not in_dilate and not pw.Rp and vt is None and not pw.geglu and pw.ln_s is None)   # (any nearest up-sampling size)
        if (KH == 1 and IH == 1 and KW % 2 == 1 and KW > 1 and stride == (1, 1) and pad == (0, (KW - 1) * dil[1] // 2) and dil[0] == 1 and fast_path
                and not in_dilate and not pw.Rp and vt is None and not pw.geglu and pw.ln_s is None and up_size is None and x3 is None and gn_in is None):
            # conv1d (the vocoder): the wave-specialised halo tiles read it as a KW x 1 filter over a T x 1 image (csrc/igemm_halo.hip)
            halo = [t for t in (15, 16) if TILE_DIMS[t][0] + (KW - 1) * dil[1] <= HALO_ROWS[t]]
        if x3 is not None:      # conv2 + conv_shortcut as one GEMM: the wave-specialised halo tiles only, three-pass halo, no up-sampling
            halo = [t for t in halo if t in (15, 16) and up_size is None and C3 % 64 == 0 and C4 % 64 == 0
                    and (TILE_DIMS[t][0] // OW + 2) * (OW + 2) <= (192 if t == 15 else 128)]
        cfg = TUNED.get(key)
        if cfg is None and TUNED_LEGACY_KEYS:           # tables written before the geometry suffix existed
            cfg = TUNED.get(tune_key(M, pw.N, C1, C2, KH, KW, stride, up_size is not None, in_dilate, pw.Rp, vt is not None,
                                     pw.geglu, pw.ln_s is not None, fast_path) + sfx)
        if cfg is not None and cfg[0] in HALO_ROWS and cfg[0] not in halo:
            if os.environ.get("ALDM_VERBOSE_TUNED") == "1":
                print(f"[ops] table entry {cfg} of '{key}' names a halo tile this geometry cannot take: heuristics instead", file=sys.stderr)
            cfg = None                                  # never trust a table entry into a tile this geometry cannot take
        tuning = TUNER is not None and not torch.cuda.is_current_stream_capturing()
        if tuning:
            cfg = TUNER.choose(key, a, M, pw, ktiles, can_split and splits is None, splits, fast_path, vt is not None, x.device, halo)
        if cfg is not None:
            tile, ring, splits = cfg
            if x3 is not None and tile in HALO_ROWS:
                ring = 3                                    # (the fused-shortcut form of the halo tiles has one ring depth)
    if gn_in is not None:
        # GroupNorm of the input inside the launch: halo tiles only (the tile is normalised once in LDS); statistics from the producers
        q1, q2 = getattr(x, "qstats", None), (getattr(x2, "qstats", None) if x2 is not None else None)
        if not gn_in_ok(x, x2, pw, stride, pad, dil, up_size, x3):
            raise _lib.AldmError("conv: gn_in needs a 3x3 / stride 1 / pad 1 launch that fits a halo tile and inputs with .qstats (gn_in_ok)")
        if tile not in HALO_ROWS:
            tile, ring = (15 if (128 % OW == 0 and (128 // OW + 2) * (OW + 2) <= HALO_ROWS[15]) else 16), GNIN_RING   # the loader waves normalise
        splits = 1
        gm_, bt_, a.gnin_groups, a.gnin_eps, a.gnin_act = gn_in
        a.gnin_gamma, a.gnin_beta = gm_.data_ptr(), bt_.data_ptr()
        a.gnin_q1, a.gnin_bm1, a.gnin_tpi1 = q1.table.data_ptr(), q1.bm, q1.tpi
        if q2 is not None:
            a.gnin_q2, a.gnin_bm2, a.gnin_tpi2 = q2.table.data_ptr(), q2.bm, q2.tpi
    if splits is None:
        splits = auto_splits(M, pw.N, ktiles) if (can_split and tile not in HALO_ROWS) else 1
    if tile == 0:
        tile = heuristic_cfg(M, pw, ktiles, can_split, fast_path, vt is not None, splits)[0]
    if not ring:
        bm, bn = TILE_DIMS[tile]
        ring = pick_ring(tile, bm, bn, pw.Rp, math.ceil(M / bm) * math.ceil(pw.N / bn) * max(1, splits), ktiles)
    qs = None
    if (qstats and splits == 1 and vt is None and not pw.geglu and not out_f32 and out2 is None and out_pix_stride == 1
            and not pw.Rp and pw.ln_s is None and not rowstats
            and pw.N % 8 == 0 and pw.N % TILE_DIMS[tile][1] == 0 and out_ld == pw.N and OH * OW >= QSTATS_MIN_HW):
        bm = TILE_DIMS[tile][0]
        if tile in HALO_ROWS:
            tpi = math.ceil(OH / (bm // OW))
            qs = QStats(torch.empty(B * tpi, 2, pw.N // 4, 2, dtype=torch.float32, device=x.device), 0, tpi)
        elif OH * OW >= bm:
            qs = QStats(torch.empty(math.ceil(M / bm), 2, pw.N // 4, 2, dtype=torch.float32, device=x.device), bm, 0)
        if qs is not None:
            a.qstat_out = qs.table.data_ptr()
    stats = None
    if rowstats:
        if tile in HALO_ROWS or pw.N % TILE_DIMS[tile][1] or splits != 1:
            tile, splits = _lib.TILE_64x64, 1             # always legal: N % 64 == 0 was checked above
            ring = pick_ring(tile, 64, 64, pw.Rp, math.ceil(M / 64) * (pw.N // 64), ktiles)
        stats = torch.empty(M, pw.N // TILE_DIMS[tile][1], 2, dtype=torch.float32, device=x.device)
        a.rowstat_out = stats.data_ptr()
    a.splits = splits
    if splits > 1:
        a.workspace = _workspace(splits * M * pw.N * 4, x.device).data_ptr()
    a.tile = tile
    a.ring = ring
    a.xcd_map = XCD_MAP
    lib = _lib.load()
    eff = lib.aldm_igemm_effective_splits(C.byref(a)) if (gn_defer and splits > 1) else 1
    a.defer_reduce = 1 if eff > 1 else 0
    ktot = KH * KW * pw.Cin + pw.Cext
    flops = 2.0 * M * pw.N * ktot + (2.0 * M * pw.Rp * (ktot + pw.N) if pw.Rp else 0.0)
    nbytes = 2.0 * (B * IH * IW * pw.Cin + M * pw.Cext + pw.N * ktot + M * ncols)
    label = (f"igemm_{TILE_NAMES[tile]}_r{pw.Rp}{'_vt' if vt is not None else ''}{'_sk' if splits > 1 else ''}"
             f"|M{M} N{pw.N} K{ktot}{' geglu' if pw.geglu else ''}")

    if qs is not None:
        out.qstats = qs                                     # (a Python attribute of this tensor object: skip lists keep the object)
    elif hasattr(out, "qstats"):
        del out.qstats                                      # a caller-supplied buffer rewritten without statistics: drop the stale table

    def finish():
        # the GroupNorm of gn=: over the partial tiles when the launch deferred its reduce, else over the bf16 output
        if gn is None:
            if eff > 1:
                d = Deferred(out, a.workspace, eff, a.bias, a.rowbias, a.rowbias_ld, a.res, (pw, rowbias, res, x3, x4))
                _pending_set(out, d)
                return d
            return (out, stats) if rowstats else out
        gamma, beta, groups, eps, act = gn
        if eff <= 1:
            y = groupnorm(out, gamma, beta, groups, eps, act)
            return (out, y) if gn_keep else y
        y = torch.empty(B, OH, OW, pw.N, dtype=torch.bfloat16, device=x.device)
        n = M * pw.N
        check(_launch(f"groupnorm_partials|HW{OH * OW} C{pw.N} S{eff}", 10.0 * n, (4.0 * eff + 2.0) * n, lambda: lib.aldm_groupnorm_partials(
            a.workspace, eff, B, OH * OW, pw.N, a.bias, a.rowbias, a.rowbias_ld, a.res, (a.out if gn_keep else None), None, 0,
            groups, eps, _p(gamma), _p(beta), act, _p(y), _stream())), "aldm_groupnorm_partials")
        return (out, y) if gn_keep else y

    if tuning and TUNER.slot is not None:                      # stage 2 of the tuner: time this launch in context
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = lib.aldm_igemm(C.byref(a), _stream())
        check(rc, "aldm_igemm")
        y = finish() if gn_defer else None                     # a deferred reduce is paid by the GroupNorm: time both
        e1.record()
        TUNER.record(key, (tile, ring, splits), e0, e1)
        return y if gn_defer else finish()
    if KEYLOG is not None and PROFILE is not None:
        KEYLOG.append((len(PROFILE), (key, (tile, ring, splits))))
    check(_launch(label, flops, nbytes, lambda: lib.aldm_igemm(C.byref(a), _stream())), "aldm_igemm")
    return finish()


# ---- the transformer blocks' projection GEMMs (csrc/pgemm.hip) ---------------------------------------------------------------
PGEMM = os.environ.get("ALDM_NO_PGEMM") != "1"
PGEMM_K = (256, 384, 640)
PGEMM_CFG = {}                                               # (M, N, K, kind) -> (mi, nt, tiles_per_range): tools/tune_pgemm.py
PGEMM_TUNED_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pgemm_gfx950.json")
if os.path.exists(PGEMM_TUNED_PATH) and os.environ.get("ALDM_NO_TUNED") != "1":
    with open(PGEMM_TUNED_PATH) as _f:
        for _k, _v in json.load(_f)["pgemm"].items():
            _m, _n, _kk, _kind = _k.split("|")
            PGEMM_CFG[(int(_m), int(_n), int(_kk), _kind)] = tuple(_v)


if os.environ.get("ALDM_PGEMM_PATCH"):                        # A/B aid: {"M|N|K|kind": [mi, nt, tpr, nw] | null (= no entry: the planner decides)}
    with open(os.environ["ALDM_PGEMM_PATCH"]) as _f:
        for _k, _v in json.load(_f).items():
            _m, _n, _kk, _kind = _k.split("|")
            if _v is None:
                PGEMM_CFG.pop((int(_m), int(_n), int(_kk), _kind), None)
            else:
                PGEMM_CFG[(int(_m), int(_n), int(_kk), _kind)] = tuple(_v)
PGEMM_USE_IGEMM = (0, 0, 0, 0)                               # table entry: this GEMM measured faster on aldm_igemm, keep it there


def _pgemm_key(M, pw, K, res, vt, rowstats=False, lora_t=False, dual=False):
    kind = (("g" if pw.geglu else "") + ("v" if vt is not None else "") + ("r" if res is not None else "") + (f"l{pw.Rp}" if pw.Rp else "")
            + ("s" if rowstats else "") + ("t" if lora_t else "") + ("d" if dual else ""))
    return (M, pw.N, K, kind)


def _pgemm(x, pw, out, out_ld, M, K, OHW, res, vt, vt_col0, vt_ld, vt_bs, rowstats, ln_parts, lora_t_out=None, vt_dual=False):
    """conv()'s 1x1 / short-K case on aldm_pgemm: same operands, same results (to rounding), a kernel built for it."""
    lib = _lib.load()
    a = _lib.PgemmArgs()
    a.x, a.w, a.M, a.N, a.K = x.data_ptr(), pw.w.data_ptr(), M, pw.N, K
    a.bias = pw.bias.data_ptr() if pw.bias is not None else None
    if pw.ln_s is not None:
        assert ln_parts.dtype == torch.float32 and ln_parts.is_contiguous() and ln_parts.dim() == 3 and ln_parts.shape[2] == 2
        assert ln_parts.shape[0] == M, "ln_parts: one row of partials per GEMM row"
        a.ln_s, a.ln_eps, a.ln_parts, a.ln_nparts = pw.ln_s.data_ptr(), pw.ln_eps, ln_parts.data_ptr(), ln_parts.shape[1]
        if pw.Rp:
            a.ln_sa, a.ln_ca = pw.ln_sa.data_ptr(), pw.ln_ca.data_ptr()
    elif ln_parts is not None:
        raise _lib.AldmError("conv: ln_parts without a LayerNorm-folded weight pack (pack_linear_ln)")
    if pw.Rp:
        a.lora_a, a.lora_b, a.Rp, a.ranks_used = pw.lora_a.data_ptr(), pw.lora_b.data_ptr(), pw.Rp, pw.ranks_used
        if lora_t_out is not None:
            assert lora_t_out.dtype == torch.bfloat16 and lora_t_out.is_contiguous() and tuple(lora_t_out.shape) == (M, pw.Rp)
            a.lora_t_out = lora_t_out.data_ptr()
    a.geglu = 1 if pw.geglu else 0
    a.res = res.data_ptr() if res is not None else None
    a.out, a.out_ld = out.data_ptr(), out_ld
    if vt is not None:
        a.vt, a.vt_col0, a.vt_ld, a.vt_batch_stride, a.OHW = vt.data_ptr(), vt_col0, vt_ld, vt_bs, OHW
        a.vt_dual = 1 if vt_dual else 0
    cfg = PGEMM_CFG.get(_pgemm_key(M, pw, K, res, vt, rowstats, lora_t_out is not None, vt_dual))
    if cfg is not None and rowstats and pw.N // (cfg[1] * cfg[2]) > 16:
        cfg = None                                           # (consumers take at most 16 partial pairs per row)
    if cfg is not None:
        a.mi, a.nt, a.tiles_per_range = cfg[:3]
        a.waves = cfg[3] if len(cfg) > 3 else 4
    if rowstats:
        a.max_ranges = 16                                    # consumers take at most 16 partial pairs per row (aldm_attn_block64)
    check(lib.aldm_pgemm_plan(C.byref(a)), "aldm_pgemm_plan")
    stats = None
    if rowstats:
        stats = torch.empty(M, pw.N // (a.nt * a.tiles_per_range), 2, dtype=torch.float32, device=x.device)
        a.rowstat_out = stats.data_ptr()
    ncols = pw.N // 2 if pw.geglu else (vt_col0 if (vt is not None and not vt_dual) else pw.N)
    flops = 2.0 * M * pw.N * K + (2.0 * M * pw.Rp * (K + pw.N) if pw.Rp else 0.0)
    nbytes = 2.0 * (M * K + pw.N * K + M * ncols)
    label = (f"pgemm_{16 * a.mi * a.waves}x{a.nt}x{a.tiles_per_range}w{a.waves}_r{pw.Rp}{'_vt' if vt is not None else ''}{'d' if vt_dual else ''}"
             f"|M{M} N{pw.N} K{K}{' geglu' if pw.geglu else ''}")
    if hasattr(out, "qstats"):
        del out.qstats                                       # (see conv(): never leave a stale GroupNorm table on a rewritten buffer)
    if KEYLOG is not None and PROFILE is not None:
        KEYLOG.append((len(PROFILE), ("pg:" + "|".join(str(v) for v in _pgemm_key(M, pw, K, res, vt, rowstats, lora_t_out is not None, vt_dual)),
                                      (a.mi, a.nt, a.tiles_per_range, a.waves))))
    check(_launch(label, flops, nbytes, lambda: lib.aldm_pgemm(C.byref(a), _stream())), "aldm_pgemm")
    return (out, stats) if rowstats else out


def linear(x2d: torch.Tensor, pw: PackedW, **kw):
    """x2d [M, K] bf16 -> [M, N]; thin view over conv with a 1x1 filter."""
    M, K = x2d.shape
    out = kw.pop("out", None)
    y = conv(x2d.view(1, 1, M, K), pw, out=(None if out is None else out), **kw)
    if kw.get("rowstats"):
        return (y[0] if out is not None else y[0].view(M, -1)), y[1]
    return y if out is not None else y.view(M, -1)


def groupnorm(x, gamma, beta, groups, eps, act=ACT_NONE, x2=None):
    """GroupNorm(+SiLU) over channels-last x (| x2).  x may be a Deferred: its split-K reduce then happens here (and fills x.out)."""
    if isinstance(x, Deferred):
        if x is not _pending_get(x.out):
            raise _lib.AldmError("groupnorm: this deferred convolution's partial tiles are gone (not the one pending on this stream)")
        B, H, W, C1 = x.out.shape
        C2 = x2.shape[3] if x2 is not None else 0
        Cg = (C1 + C2) // groups
        if (C1 + C2) % groups or Cg % 4 or C1 % Cg or H * W * (Cg // 4) > 4096:
            raise _lib.AldmError(f"groupnorm: deferred input [{B} x {H * W} x {C1}+{C2} / {groups} groups] does not fit aldm_groupnorm_partials")
        y = torch.empty(B, H, W, C1 + C2, dtype=torch.bfloat16, device=x.out.device)
        lib = _lib.load()
        n = B * H * W * (C1 + C2)
        check(_launch(f"groupnorm_partials|HW{H * W} C{C1}+{C2} S{x.eff}", 10.0 * n, (4.0 * x.eff + 2.0) * n, lambda: lib.aldm_groupnorm_partials(
            x.ws, x.eff, B, H * W, C1, x.bias, x.rowbias, x.rowbias_ld, x.res, _p(x.out), _p(x2), C2, groups, eps,
            _p(gamma), _p(beta), act, _p(y), _stream())), "aldm_groupnorm_partials")
        _pending_set(x.out, None)
        return y
    _require_gpu(x)
    B, H, W, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    y = torch.empty(B, H, W, C1 + C2, dtype=torch.bfloat16, device=x.device)
    lib = _lib.load()
    n = B * H * W * (C1 + C2)
    q1, q2 = getattr(x, "qstats", None), (getattr(x2, "qstats", None) if x2 is not None else None)
    Cg = (C1 + C2) // groups
    if (q1 is not None and (x2 is None or q2 is not None) and (C1 + C2) % groups == 0 and Cg % 4 == 0
            and C1 % 8 == 0 and C2 % 8 == 0 and groups <= 64 and (C1 + C2) // 8 <= 256
            and x.is_contiguous() and (x2 is None or x2.is_contiguous())):
        # one coalesced pass: the producing convolution(s) handed the statistics over
        # big images (the VAE's mels): the (mean, rstd) table is summed once, by a launch of its own, instead of by every workgroup
        tiles = max(q.tpi if q.tpi > 0 else H * W // q.bm + 1 for q in (q1, q2) if q is not None)
        ws = torch.empty(B, 64, 2, dtype=torch.float32, device=x.device) if tiles >= GN_FINALIZE_MIN_TILES else None
        check(_launch(f"groupnorm_apply|HW{H * W} C{C1 + C2}", 10.0 * n, 4.0 * n, lambda: lib.aldm_groupnorm_apply(
            _p(x), _p(q1.table), q1.bm, q1.tpi, _p(x2), _p(q2.table) if q2 is not None else None, q2.bm if q2 is not None else 0,
            q2.tpi if q2 is not None else 0, B, H * W, C1, C2, groups, eps, _p(gamma), _p(beta), act, _p(y), _p(ws), _stream())),
            "aldm_groupnorm_apply")
        return y
    check(_launch(f"groupnorm|HW{H * W} C{C1 + C2}", 10.0 * n, 4.0 * n, lambda: lib.aldm_groupnorm(
        _p(x), _p(x2), B, H * W, C1, C2, groups, eps, _p(gamma), _p(beta), act, _p(y), _stream())), "aldm_groupnorm")
    return y


GNIN_RING = int(os.environ.get("ALDM_GNIN_RING", "0"))     # LDS-DMA ring depth of the input-norm-folding halo launches (0: pick_ring); tuning aid


def gn_in_ok(x, x2, pw, stride=(1, 1), pad=(1, 1), dil=(1, 1), up_size=None, x3=None):
    """Can a convolution apply the GroupNorm of its input itself (conv(gn_in=))?  3x3 / stride 1 / pad 1 on a halo tile, every source a
    raw convolution output with its statistics table, channel counts multiples of 64, at most 512 input channels."""
    if not isinstance(x, torch.Tensor) or getattr(x, "qstats", None) is None or x.dim() != 4:
        return False
    if x2 is not None and (not isinstance(x2, torch.Tensor) or getattr(x2, "qstats", None) is None):
        return False
    B, IH, IW, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    hw = IH * IW
    for q in (x.qstats, x2.qstats if x2 is not None else None):
        if q is not None and not (q.tpi > 0 or q.bm <= hw):
            return False
    fits = any(bm % IW == 0 and (bm // IW + 2) * (IW + 2) <= HALO_ROWS[t] for t, bm in ((7, 128), (8, 64)))
    return (pw.KH == 3 and pw.KW == 3 and tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and tuple(dil) == (1, 1) and up_size is None
            and x3 is None and not pw.Rp and not pw.geglu and pw.ln_s is None and C1 % 64 == 0 and C2 % 64 == 0 and C1 + C2 <= 512
            and pw.Cin == C1 + C2 and fits)


def gn_silu_conv_out_ok(x, pw, groups):
    """Can conv_norm_out -> SiLU -> conv_out run as the one fused launch?  (the AudioLDM latent: width 16, 128 channels, <= 16 outputs,
    3x3 / pad 1, GroupNorm statistics handed over by the convolution that produced x)"""
    q = getattr(x, "qstats", None) if isinstance(x, torch.Tensor) else None
    return (q is not None and x.dim() == 4 and x.shape[2] == 16 and x.shape[3] == 128 and x.is_contiguous() and x.dtype == torch.bfloat16
            and pw.KH == 3 and pw.KW == 3 and pw.N <= 16 and pw.Cin == 128 and not pw.Rp and 128 % groups == 0 and (128 // groups) % 4 == 0
            and groups <= 64 and (q.tpi > 0 or q.bm <= x.shape[1] * x.shape[2]))


def gn_silu_conv_out(x, gamma, beta, groups, eps, pw):
    """F.silu(F.group_norm(x)) -> 3x3 conv (pad 1) to pw.N channels, fp32 NHWC out, ONE launch (aldm_gn_silu_conv3x3_small)."""
    _require_gpu(x)
    B, H, W, Cc = x.shape
    q = x.qstats
    out = torch.empty(B, H, W, pw.N, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    n = B * H * W
    check(_launch(f"gn_silu_conv3x3|HW{H * W} C{Cc} N{pw.N}", 2.0 * n * pw.N * 9 * Cc + 10.0 * n * Cc, 2.0 * n * Cc + 4.0 * n * pw.N,
                  lambda: lib.aldm_gn_silu_conv3x3_small(_p(x), _p(q.table), q.bm, q.tpi, B, H, W, Cc, groups, eps, _p(gamma), _p(beta), _p(pw.w),
                                                         pw.Kpad, _p(pw.bias), pw.N, _p(out), _stream())), "aldm_gn_silu_conv3x3_small")
    return out


def layernorm(x2d, gamma, beta, eps=1e-5):
    _require_gpu(x2d)
    M, Cc = x2d.shape
    y = torch.empty_like(x2d)
    lib = _lib.load()
    check(_launch(f"layernorm|M{M} C{Cc}", 10.0 * M * Cc, 4.0 * M * Cc, lambda: lib.aldm_layernorm(
        _p(x2d), M, Cc, _p(gamma), _p(beta), eps, _p(y), _stream())), "aldm_layernorm")
    return y


LOG2E = 1.4426950408889634


def attention(qk, vt, B, N, H, d, out=None, kv_len=None, fp8=False, prescaled=False):
    """qk [B*N, 2C] (Q | K), vt [B, C, Npad]; returns [B*N, C].  kv_len: optional int32 [B] valid-key counts
    (right-padded batches).  prescaled: Q already carries d^-0.5 * log2(e) (folded into the to_q weights at pack time)."""
    _require_gpu(qk)
    Cc = H * d
    if out is None:
        out = torch.empty(B * N, Cc, dtype=torch.bfloat16, device=qk.device)
    q_ptr = C.c_void_p(qk.data_ptr())
    k_ptr = C.c_void_p(qk.data_ptr() + Cc * 2)
    lib = _lib.load()
    if kv_len is not None:
        if prescaled or fp8:
            raise _lib.AldmError("attention: the varlen kernel applies d^-0.5 itself and has no fp8 form -- a pre-scaled Q (or "
                                 "fp8=True) with kv_len would be scaled twice / silently ignored")
        assert kv_len.dtype == torch.int32 and kv_len.numel() == B and kv_len.is_cuda
        check(_launch(f"attention_varlen_d{d}_n{N}", 4.0 * B * N * N * Cc, 2.0 * 4 * B * N * Cc, lambda: lib.aldm_attention_varlen(
            q_ptr, qk.shape[1], k_ptr, qk.shape[1], _p(vt), vt.shape[2], vt.stride(0), B, N, H, d, 1.0 / math.sqrt(d),
            _p(kv_len), _p(out), Cc, _stream())), "aldm_attention_varlen")
        return out
    if fp8:                                                   # config 5: e4m3 Q / K / V / P operands
        check(_launch(f"attention_fp8_d{d}_n{N}", 4.0 * B * N * N * Cc, 2.0 * 4 * B * N * Cc, lambda: lib.aldm_attention_fp8(
            q_ptr, qk.shape[1], k_ptr, qk.shape[1], _p(vt), vt.shape[2], vt.stride(0), B, N, H, d,
            (1.0 / LOG2E) if prescaled else 1.0 / math.sqrt(d), _p(out), Cc, _stream())), "aldm_attention_fp8")
        return out
    if prescaled:
        check(_launch(f"attention_d{d}_n{N}", 4.0 * B * N * N * Cc, 2.0 * 4 * B * N * Cc, lambda: lib.aldm_attention_prescaled(
            q_ptr, qk.shape[1], k_ptr, qk.shape[1], _p(vt), vt.shape[2], vt.stride(0), B, N, H, d, _p(out), Cc, _stream())),
            "aldm_attention_prescaled")
        return out
    check(_launch(f"attention_d{d}_n{N}", 4.0 * B * N * N * Cc, 2.0 * 4 * B * N * Cc, lambda: lib.aldm_attention(
        q_ptr, qk.shape[1], k_ptr, qk.shape[1], _p(vt), vt.shape[2], vt.stride(0), B, N, H, d, 1.0 / math.sqrt(d),
        _p(out), Cc, _stream())), "aldm_attention")
    return out


def attn_block64_ok(pw, N, H, d, ln_parts):
    """Can the (sample, head)-fused projection + attention launch take this module?  (C = 640 = 8 x 80, N <= 64, LayerNorm folded
    with the producer's statistics at hand, combined LoRA rank <= 32.)"""
    return (N <= 64 and H == 8 and d == 80 and pw.N == 1920 and pw.Kpad == 640 and pw.ln_s is not None and ln_parts is not None
            and (pw.Rp == 0 or getattr(pw, "ranks_used", 99) <= 32))


# The 252-token level's fused projection + attention launch (csrc/attn_block256.hip) is parity-green and OFF by default: measured in the
# replayed step (round 4, gpurun_out/r4_step_b256.txt) 25.4 us per module against 12.9 + 8.1 us for the projection launch + the attention
# launch.  A (sample, head) workgroup carries 31.5 MFLOP of projection + 12 MFLOP of attention on ONE CU (64 of 256 CUs busy): 4.4 us at
# the MFMA peak of a CU before any latency, where the two launches spread the same work over the whole chip.  ALDM_ATTN_BLOCK256=1 turns it on.
ATTN_BLOCK256 = os.environ.get("ALDM_ATTN_BLOCK256") == "1"


def attn_block_ok(pw, N, H, d, ln_parts):
    """Which (sample, head)-fused projection + attention launch takes this module: 64 (C = 640, N <= 64), 256 (C = 384, N <= 256) or 0."""
    if attn_block64_ok(pw, N, H, d, ln_parts):
        return 64
    if (ATTN_BLOCK256 and N <= 256 and H == 8 and d == 48 and pw.N == 1152 and pw.Kpad == 384 and pw.ln_s is not None and ln_parts is not None
            and ln_parts.shape[1] <= 16 and (pw.Rp == 0 or getattr(pw, "ranks_used", 99) <= 32)):
        return 256
    return 0


def attn_block(x2d, pw, ln_parts, B, N, H, d, fp8=False):
    """x2d [B*N, C] raw hidden state -> attention output [B*N, C] through the fused launch attn_block_ok() names (fp8: config 5's e4m3
    attention operands; the 64-token launch has that form, the 256-token one does not)."""
    kind = attn_block_ok(pw, N, H, d, ln_parts)
    if kind == 64:
        return attn_block64(x2d, pw, ln_parts, B, N, H, d, fp8=fp8)
    if fp8:
        raise _lib.AldmError("attn_block: the 256-token fused launch has no fp8 form")
    if kind != 256:
        raise _lib.AldmError(f"attn_block: no fused projection + attention launch for N {N}, {H} heads x {d}")
    _require_gpu(x2d)
    Cc = H * d
    assert x2d.dtype == torch.bfloat16 and x2d.is_contiguous() and tuple(x2d.shape) == (B * N, Cc)
    assert ln_parts.dtype == torch.float32 and ln_parts.is_contiguous() and ln_parts.shape[0] == B * N
    out = torch.empty(B * N, Cc, dtype=torch.bfloat16, device=x2d.device)
    fl = B * (6.0 * N * Cc * Cc + 4.0 * N * N * Cc + (12.0 * N * Cc * getattr(pw, "ranks_used", 0) if pw.Rp else 0.0))
    check(_launch(f"attn_block256_d{d}_n{N}", fl, 2.0 * (2 * B * N * Cc + 3 * Cc * Cc), lambda: _lib.load().aldm_attn_block256(
        _p(x2d), _p(ln_parts), ln_parts.shape[1], _p(pw.w), pw.Kpad, _p(pw.bias), _p(pw.ln_s), _p(pw.lora_a), _p(pw.lora_b), pw.Rp,
        getattr(pw, "ranks_used", 0) if pw.Rp else 0, _p(pw.ln_sa), _p(pw.ln_ca), pw.ln_eps, B, N, H, d, _p(out), _stream())),
        "aldm_attn_block256")
    return out


def attn_block64(x2d, pw, ln_parts, B, N, H, d, fp8=False):
    """x2d [B*N, C] raw hidden state -> attention output [B*N, C]: LayerNorm-folded QKV projection with LoRA + 64-token attention,
    one launch (aldm_attn_block64)."""
    _require_gpu(x2d)
    Cc = H * d
    assert x2d.dtype == torch.bfloat16 and x2d.is_contiguous() and tuple(x2d.shape) == (B * N, Cc)
    assert ln_parts.dtype == torch.float32 and ln_parts.is_contiguous() and ln_parts.shape[0] == B * N
    out = torch.empty(B * N, Cc, dtype=torch.bfloat16, device=x2d.device)
    fl = B * (6.0 * N * Cc * Cc + 4.0 * N * N * Cc + (12.0 * N * Cc * getattr(pw, "ranks_used", 0) if pw.Rp else 0.0))
    fn = _lib.load().aldm_attn_block64_fp8 if fp8 else _lib.load().aldm_attn_block64
    check(_launch(f"attn_block64{'_fp8' if fp8 else ''}_d{d}_n{N}", fl, 2.0 * (2 * B * N * Cc + 3 * Cc * Cc), lambda: fn(
        _p(x2d), _p(ln_parts), ln_parts.shape[1], _p(pw.w), pw.Kpad, _p(pw.bias), _p(pw.ln_s), _p(pw.lora_a), _p(pw.lora_b), pw.Rp,
        getattr(pw, "ranks_used", 0) if pw.Rp else 0, _p(pw.ln_sa), _p(pw.ln_ca), pw.ln_eps, B, N, H, d, _p(out), _stream())),
        "aldm_attn_block64")
    return out


HIFIGAN_PAIR = os.environ.get("ALDM_NO_HIFIGAN_PAIR") != "1"        # A/B aid: the vocoder's low-channel stages back on aldm_igemm


def hifigan_respair_ok(x, c1, c2, dil):
    """Can one aldm_hifigan_respair launch take this (convs1[q], convs2[q]) pair?  (C = 32 / 64 channels, 3 / 7 / 11 taps, dilation <= 5.)"""
    Cc = x.shape[3]
    return (HIFIGAN_PAIR and x.shape[1] == 1 and c1.N == Cc and c2.N == Cc and c1.Cin == Cc and c2.Cin == Cc and c1.KH == 1 and c2.KH == 1
            and c1.KW == c2.KW and c1.bias is not None and c2.bias is not None
            and bool(_lib.load().aldm_hifigan_respair_supported(Cc, c1.KW, dil)))


def hifigan_respair(x, c1, c2, dil, slope, alpha=1.0, res2=None, post_act=ACT_NONE, post_slope=0.0):
    """x [B, 1, T, C] bf16 residual stream -> post_act(alpha (x + conv2(lrelu(conv1(lrelu(x))))) + res2), one launch."""
    _require_gpu(x)
    B, _, T, Cc = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and (res2 is None or (res2.shape == x.shape and res2.is_contiguous()))
    out = torch.empty_like(x)
    K = c1.KW
    fl = 2.0 * 2.0 * B * T * Cc * Cc * K
    check(_launch(f"hifigan_pair_c{Cc}_k{K}_d{dil}|M{B * T}", fl, 2.0 * B * T * Cc * (2 + (1 if res2 is not None else 0)), lambda: _lib.load().aldm_hifigan_respair(
        _p(x), B, T, Cc, _p(c1.w), c1.Kpad, _p(c1.bias), dil, _p(c2.w), c2.Kpad, _p(c2.bias), K, slope, alpha, _p(res2), post_act, post_slope,
        _p(out), _stream())), "aldm_hifigan_respair")
    return out


def conv1d_to1(x, pw, act=ACT_NONE):
    """x [B, 1, T, C] bf16 (activated) -> fp32 [B, T]: Conv1d(C -> 1, K taps, same padding) (+ tanh) as one HBM-bound stencil launch.
    pw = ops.pack_conv of the [1 (padded to 8), C, 1, K] weight: row 0 holds the filter in (tap, cin) order."""
    _require_gpu(x)
    B, _, T, Cc = x.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and pw.Cin == Cc and pw.KH == 1
    out = torch.empty(B, T, dtype=torch.float32, device=x.device)
    check(_launch(f"conv1d_to1_c{Cc}_k{pw.KW}|M{B * T}", 2.0 * B * T * Cc * pw.KW, 2.0 * B * T * Cc + 4.0 * B * T, lambda: _lib.load().aldm_conv1d_to1(
        _p(x), B, T, Cc, _p(pw.w), _p(pw.bias), pw.KW, act, _p(out), _stream())), "aldm_conv1d_to1")
    return out


WIDE_HEAD_DIMS = (128, 256, 512)


def attention_wide(qk, vt, B, N, d, out=None):
    """Single wide head (AutoencoderKL mid-block): qk [B*N, 2d] (pre-scaled Q | K), vt [B, d, npad] zero beyond N, npad % 32 == 0."""
    _require_gpu(qk)
    assert d in WIDE_HEAD_DIMS and vt.shape[2] % 32 == 0 and vt.shape[2] >= N
    if out is None:
        out = torch.empty(B * N, d, dtype=torch.bfloat16, device=qk.device)
    check(_launch(f"attention_wide_d{d}_n{N}", 4.0 * B * N * N * d, 2.0 * 4 * B * N * d, lambda: _lib.load().aldm_attention_wide(
        C.c_void_p(qk.data_ptr()), qk.shape[1], C.c_void_p(qk.data_ptr() + 2 * d), qk.shape[1], _p(vt), vt.shape[2], vt.stride(0),
        B, N, d, _p(out), d, _stream())), "aldm_attention_wide")
    return out


def embed_layernorm(ids, word, pos, type0, gamma, beta, eps, pad_idx):
    """ids int64 [B, L] (device) -> LayerNorm(word[ids] + type0 + pos[position_ids]) as bf16 [B*L, C]."""
    _require_gpu(ids)
    assert ids.dtype == torch.int64 and ids.is_contiguous()
    B, L = ids.shape
    Cc = word.shape[1]
    for t in (word, pos, type0, gamma, beta):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    y = torch.empty(B * L, Cc, dtype=torch.bfloat16, device=ids.device)
    check(_lib.load().aldm_embed_layernorm(_p(ids), B, L, Cc, _p(word), word.shape[0], _p(pos), pos.shape[0], _p(type0),
                                           _p(gamma), _p(beta), eps, pad_idx, _p(y), _stream()), "aldm_embed_layernorm")
    return y


def softmax_rows(s, scale, cols, ld_out):
    rows = s.shape[0]
    p = torch.empty(rows, ld_out, dtype=torch.bfloat16, device=s.device)
    check(_lib.load().aldm_softmax_rows(_p(s), rows, cols, s.shape[1], scale, _p(p), ld_out, _stream()),
          "aldm_softmax_rows")
    return p


def timestep_embedding(t_dev, B, dim):
    out = torch.empty(B, dim, dtype=torch.bfloat16, device=t_dev.device)
    stride = 0 if t_dev.numel() == 1 else 1
    check(_lib.load().aldm_timestep_embedding(_p(t_dev), stride, B, dim, _p(out), _stream()), "aldm_timestep_embedding")
    return out


def nchw_to_nhwc(x_f32, out_f32=False):
    _require_gpu(x_f32)
    B, Cc, H, W = x_f32.shape
    y = torch.empty(B, H, W, Cc, dtype=torch.float32 if out_f32 else torch.bfloat16, device=x_f32.device)
    check(_lib.load().aldm_nchw_f32_to_nhwc(_p(x_f32.contiguous()), B, Cc, H * W, _p(y), int(out_f32), _stream()),
          "aldm_nchw_f32_to_nhwc")
    return y


def nhwc_to_nchw_f32(x):
    _require_gpu(x)
    B, H, W, Cc = x.shape
    y = torch.empty(B, Cc, H, W, dtype=torch.float32, device=x.device)
    check(_lib.load().aldm_nhwc_to_nchw_f32(_p(x.contiguous()), int(x.dtype == torch.float32), B, Cc, H * W, _p(y),
                                            _stream()), "aldm_nhwc_to_nchw_f32")
    return y


def f32_to_bf16(x, mul=1.0, out=None):
    _require_gpu(x)
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib.load().aldm_f32_to_bf16(_p(x), x.numel(), mul, _p(out), _stream()), "aldm_f32_to_bf16")
    return out


def cfg_ddim_step(eps, x, cfg, guidance, coef, step_idx, x_in):
    B = x.shape[0]
    n = x.numel() // B
    check(_launch("cfg_ddim_step", 6.0 * x.numel(), (4.0 * (2 if cfg else 1) + 8.0 + 2.0 * (2 if cfg else 1)) * x.numel(),
                  lambda: _lib.load().aldm_cfg_ddim_step(_p(eps), _p(x), B, n, int(cfg), guidance, _p(coef), _p(step_idx), _p(x_in),
                                                          _stream())), "aldm_cfg_ddim_step")


def ddim_step_fused(eps, x, cfg, guidance, coef, step_idx, x_in, table, rowbias, timesteps_f32, t_out, ticket):
    """cfg_ddim_step + gather_row(next step) + advance_step as one launch (aldm_ddim_step_fused)."""
    B = x.shape[0]
    n = x.numel() // B
    row = table[0].numel() if table is not None else 0
    assert ticket.dtype == torch.int32 and step_idx.dtype == torch.int32
    check(_launch("ddim_step_fused", 6.0 * x.numel(), (4.0 * (2 if cfg else 1) + 8.0 + 2.0 * (2 if cfg else 1)) * x.numel() + 8.0 * row,
                  lambda: _lib.load().aldm_ddim_step_fused(_p(eps), _p(x), B, n, int(cfg), guidance, _p(coef), _p(step_idx), _p(x_in), _p(table),
                                                           row, _p(rowbias), _p(timesteps_f32), timesteps_f32.numel(), _p(t_out), _p(ticket),
                                                           _stream())), "aldm_ddim_step_fused")


def add_noise(x, noise, coef):
    _require_gpu(x)
    B = x.shape[0]
    xf, nf = x.float().contiguous(), noise.float().contiguous()
    out = torch.empty_like(xf)
    check(_lib.load().aldm_add_noise(_p(xf), _p(nf), _p(coef), B, xf.numel() // B, _p(out), _stream()), "aldm_add_noise")
    return out.to(x.dtype)


def add_noise_t(x, noise, alphas_cumprod, timesteps):
    """DDIMScheduler.add_noise with device-side coefficients: x, noise fp32 [B, ...], alphas_cumprod fp32 [T], timesteps int64 [B]."""
    _require_gpu(x)
    B = x.shape[0]
    xf, nf = x.float().contiguous(), noise.float().contiguous()
    assert alphas_cumprod.dtype == torch.float32 and timesteps.dtype == torch.int64 and timesteps.numel() == B
    out = torch.empty_like(xf)
    check(_lib.load().aldm_add_noise_t(_p(xf), _p(nf), _p(alphas_cumprod), _p(timesteps.contiguous()), alphas_cumprod.numel(), B,
                                       xf.numel() // B, _p(out), _stream()), "aldm_add_noise_t")
    return out.to(x.dtype)


def gaussian_sample(params_nchw, noise):
    """mean + exp(0.5 clamp(logvar)) * noise for params [B, 2C, H, W] fp32 (mean | logvar), noise [B, C, H, W] fp32."""
    _require_gpu(params_nchw)
    B, C2 = params_nchw.shape[:2]
    pf, nf = params_nchw.float().contiguous(), noise.float().contiguous()
    out = torch.empty_like(nf)
    check(_lib.load().aldm_gaussian_sample(_p(pf), _p(nf), B, nf.numel() // B, _p(out), _stream()), "aldm_gaussian_sample")
    return out


def sleep_us(us):
    check(_lib.load().aldm_sleep_us(int(us), _stream()), "aldm_sleep_us")


def gather_row(table, idx, out):
    """out[...] = table[idx[0]] (device-side index); table [n, ...] fp32 contiguous, out one row."""
    row = table[0].numel()
    assert table.dtype == torch.float32 and out.dtype == torch.float32 and out.numel() == row and idx.dtype == torch.int32
    check(_launch("gather_row", 0.0, 8.0 * row, lambda: _lib.load().aldm_gather_row(_p(table), _p(idx), row, _p(out), _stream())),
          "aldm_gather_row")


def advance_step(step_idx, timesteps_f32, t_out):
    check(_launch("advance_step", 0.0, 16.0, lambda: _lib.load().aldm_advance_step(_p(step_idx), _p(timesteps_f32), timesteps_f32.numel(),
                                                                                    _p(t_out), _stream())), "aldm_advance_step")


def adamw_flat(p, g, m, v, lr, beta1, beta2, eps, wd, step, grad_scale=1.0):
    check(_lib.load().aldm_adamw_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, step,
                                      grad_scale, _stream()), "aldm_adamw_flat")


# ---------------------------------------------------------------------------------------------------------
# training-side wrappers (LoRA fine-tune step)
# ---------------------------------------------------------------------------------------------------------
def transpose_tokens(x2d, B, N, Cc, npad=None):
    """rows [B*N, C] -> token-major [B, C, Npad] (zero padded)."""
    _require_gpu(x2d)
    npad = npad or (N + 7) // 8 * 8
    out = torch.empty(B, Cc, npad, dtype=torch.bfloat16, device=x2d.device)
    check(_lib.load().aldm_transpose_tokens(_p(x2d), x2d.stride(0), B, N, Cc, npad, _p(out), _stream()), "aldm_transpose_tokens")
    return out


def attention_train(qkv, qkvT, B, N, H, d):
    """qkv [B*N, 3C] row-major, qkvT [B, 3C, Npad] token-major -> (out [B*N, C], lse [B, H, N] fp32)."""
    Cc = H * d
    out = torch.empty(B * N, Cc, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    base, tb = qkv.data_ptr(), qkvT.data_ptr()
    npad = qkvT.shape[2]
    check(_lib.load().aldm_attention_lse(C.c_void_p(base), 3 * Cc, C.c_void_p(base + 2 * Cc), 3 * Cc,
                                         C.c_void_p(tb + 2 * (2 * Cc) * npad), npad, 3 * Cc * npad, B, N, H, d,
                                         1.0 / math.sqrt(d), _p(out), Cc, _p(lse), _stream()), "aldm_attention_lse")
    return out, lse


def attention_bwd(qkv, qkvT, dO, O, lse, B, N, H, d, dOT=None):
    """-> dqkv [B*N, 3C] (dQ | dK | dV).  dOT: dO token-major [B, C, Npad] when its producer already stored it that way."""
    Cc = H * d
    npad = qkvT.shape[2]
    if dOT is None:
        dOT = transpose_tokens(dO, B, N, Cc, npad)
    assert tuple(dOT.shape) == (B, Cc, npad)
    dqkv = torch.empty(B * N, 3 * Cc, dtype=torch.bfloat16, device=qkv.device)
    delta = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    base, tb, gb = qkv.data_ptr(), qkvT.data_ptr(), dqkv.data_ptr()
    check(_lib.load().aldm_attention_bwd(
        C.c_void_p(base), C.c_void_p(base + 2 * Cc), C.c_void_p(base + 4 * Cc), 3 * Cc,
        C.c_void_p(tb), C.c_void_p(tb + 2 * Cc * npad), _p(dOT), npad, 3 * Cc * npad, 3 * Cc * npad, Cc * npad,
        _p(dO), _p(O), Cc, _p(lse), _p(delta), B, N, H, d, 1.0 / math.sqrt(d),
        C.c_void_p(gb), C.c_void_p(gb + 2 * Cc), C.c_void_p(gb + 4 * Cc), 3 * Cc, _stream()), "aldm_attention_bwd")
    return dqkv


def _like(buf, like):
    assert buf.is_contiguous() and buf.numel() == like.numel() and buf.dtype == like.dtype
    return buf


def groupnorm_bwd(x, dy, gamma, beta, groups, eps, act, x2=None, need_dx2=True, dx_add=None, dx2_add=None):
    """dx (dx2) of act(group_norm(cat[x, x2])).  dx_add / dx2_add: gradients x / x2 already hold from their other consumers;
    they are added inside the kernel (fresh output tensors, the inputs are not modified)."""
    B, H, W, C1 = x.shape
    C2 = x2.shape[3] if x2 is not None else 0
    dx = torch.empty_like(x)
    dx2 = torch.empty_like(x2) if (x2 is not None and need_dx2) else None
    a1 = _like(dx_add, x) if dx_add is not None else None
    a2 = _like(dx2_add, x2) if (dx2 is not None and dx2_add is not None) else None
    if isinstance(dy, Deferred):                           # dY = partial tiles of the dX convolution launched just before
        if dy is not _pending_get(dy.out) or dy.bias is not None or dy.rowbias is not None or dy.res is not None:
            raise _lib.AldmError("groupnorm_bwd: deferred dY must be the pending split-K launch, without bias / residual")
        assert tuple(dy.out.shape) == (B, H, W, C1 + C2)
        check(_lib.load().aldm_groupnorm_bwd_partials(_p(x), _p(x2), dy.ws, dy.eff, B, H * W, C1, C2, groups, eps, _p(gamma), _p(beta),
                                                      act, _p(dx), _p(dx2), _p(a1), _p(a2), _stream()), "aldm_groupnorm_bwd_partials")
        _pending_set(dy.out, None)
        return dx, dx2
    check(_lib.load().aldm_groupnorm_bwd(_p(x), _p(x2), _p(dy), B, H * W, C1, C2, groups, eps, _p(gamma), _p(beta), act,
                                         _p(dx), _p(dx2), _p(a1), _p(a2), _stream()), "aldm_groupnorm_bwd")
    return dx, dx2


def layernorm_bwd(x2d, dy, gamma, eps=1e-5, dx_add=None):
    M, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    a = _like(dx_add, x2d) if dx_add is not None else None
    check(_lib.load().aldm_layernorm_bwd(_p(x2d), _p(dy), M, Cc, _p(gamma), eps, _p(dx), _p(a), _stream()), "aldm_layernorm_bwd")
    return dx


def geglu_fwd(h):
    M, two_i = h.shape
    out = torch.empty(M, two_i // 2, dtype=torch.bfloat16, device=h.device)
    check(_lib.load().aldm_geglu_fwd(_p(h), M, two_i // 2, _p(out), _stream()), "aldm_geglu_fwd")
    return out


def geglu_bwd(h, dout):
    dh = torch.empty_like(h)
    check(_lib.load().aldm_geglu_bwd(_p(h), _p(dout), h.shape[0], h.shape[1] // 2, _p(dh), _stream()), "aldm_geglu_bwd")
    return dh


def add_bf16(a, b):
    c = torch.empty_like(a)
    check(_lib.load().aldm_add_bf16(_p(a), _p(b), a.numel(), _p(c), _stream()), "aldm_add_bf16")
    return c


def upsample_nearest_bwd(dy, ih, iw):
    B, OH, OW, Cc = dy.shape
    dx = torch.empty(B, ih, iw, Cc, dtype=torch.bfloat16, device=dy.device)
    check(_lib.load().aldm_upsample_nearest_bwd(_p(dy), B, ih, iw, OH, OW, Cc, _p(dx), _stream()), "aldm_upsample_nearest_bwd")
    return dx


def tn_small(P, Q, rows_dev, Qc=None):
    """flat_grad rows += P^T Q  through the device row table (see aldm_tn_small)."""
    M, Rp = P.shape
    check(_lib.load().aldm_tn_small(_p(P), Rp, _p(Q), Q.stride(0), Qc or Q.shape[1], M, _p(rows_dev), _stream()), "aldm_tn_small")



class TnBatch:
    """Deferred LoRA-gradient products: collects (P, Q, rows) jobs during the backward pass and runs them in ONE
    aldm_tn_batched launch per rank padding (Rp = 32 / 64).  The device job tables are allocated once (stable pointers for
    hipGraph capture); under capture the launch is recorded first and `upload()` fills the tables after the capture ends."""
    REC = 48                         # sizeof(TnJob): 3 pointers + 6 ints

    def __init__(self, capacity, device):
        self.capacity = capacity
        self.tab = {rp: torch.zeros(capacity * self.REC, dtype=torch.uint8, device=device) for rp in (32, 64)}
        self.jobs = {32: [], 64: []}
        self.keep = []               # operand tensors stay referenced until the launch is recorded
        self._packed = {}
        self.owner = None            # the snapshot whose records the device tables currently hold (None: an eager launch's)

    def add(self, P, Q, rows_dev, Qc):
        M, Rp = P.shape
        ldq = Q.stride(0)
        if ldq % 8 or Qc % 8 or P.data_ptr() % 16 or Q.data_ptr() % 16 or len(self.jobs[Rp]) >= self.capacity:
            tn_small(P, Q, rows_dev, Qc=Qc)                  # odd shapes: the per-call (scalar) path
            return
        qt = -(-Qc // 128)
        msplit = max(1, 24 // qt)                            # ~24 workgroups per job: the whole batch fills the GPU
        mpb = max(64, -(-(-(-M // msplit)) // 32) * 32)
        self.jobs[Rp].append((P.data_ptr(), Q.data_ptr(), rows_dev.data_ptr(), M, ldq, Qc, qt, mpb, qt * (-(-M // mpb))))
        self.keep += [P, Q, rows_dev]

    def _pack(self):
        import struct
        self._packed = {}
        for rp, jobs in self.jobs.items():
            if not jobs:
                continue
            wg0, recs = 0, []
            for (pp, qq, rr, M, ldq, Qc, qt, mpb, nwg) in jobs:
                recs.append(struct.pack("<qqqiiiiii", pp, qq, rr, M, ldq, Qc, qt, mpb, wg0))
                wg0 += nwg
            self._packed[rp] = (b"".join(recs), len(jobs), wg0)

    def upload(self):
        """Copy the packed tables to the device (outside any capture)."""
        for rp, (blob, n, _) in self._packed.items():
            self.tab[rp][: len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        self.owner = None

    def snapshot(self):
        """The records of the launch just recorded (call right after a capture ends): a captured graph replays aldm_tn_batched
        against the ONE device table, so it must be able to put ITS records back (restore) after anything else used the table."""
        return {rp: blob for rp, (blob, _, _) in self._packed.items()}

    def restore(self, snap):
        """Make the device tables hold `snap`'s records (no-op when they already do).  Stream-ordered H2D copies on the current
        stream, i.e. in front of the replay that needs them."""
        if snap is None or self.owner is snap:
            return
        for rp, blob in snap.items():
            self.tab[rp][: len(blob)].copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
        self.owner = snap

    def launch(self):
        self._pack()
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            self.upload()
        for rp, (_, n, total) in self._packed.items():
            check(_lib.load().aldm_tn_batched(_p(self.tab[rp]), n, total, rp, _stream()), "aldm_tn_batched")
        self.jobs = {32: [], 64: []}
        self.keep = []

def lora_pack(jobs_dev, njobs):
    check(_lib.load().aldm_lora_pack(_p(jobs_dev), njobs, _stream()), "aldm_lora_pack")


def mse_grad(pred, target, loss, grad_scale=1.0):
    dp = torch.empty(pred.shape, dtype=torch.bfloat16, device=pred.device)
    check(_lib.load().aldm_mse_grad(_p(pred), _p(target), pred.numel(), grad_scale, _p(dp), _p(loss), _stream()), "aldm_mse_grad")
    return dp


def pack_conv_bwd(weight, lo=0, hi=None):
    """dX weights of a stride-1 'same' conv: Wt[cin][(kh, kw, cout)] = W[cout][cin][KH-1-kh][KW-1-kw] for cin in [lo, hi)."""
    w = weight.detach()[:, lo:hi]
    return pack_conv(w.flip(2, 3).permute(1, 0, 2, 3).contiguous(), None)


def pack_linear_bwd(weight, lo=0, hi=None):
    return pack_linear(weight.detach().t()[lo:hi].contiguous(), None)
