"""In-step A/B of aldm_pgemm launch shapes (rows per workgroup, tile width, tiles per range, waves): every legal shape of every projection GEMM
of the config-2 denoise step, measured in the replayed step as tools/ab_overlay.py does for the convolution tiles (variant i = the i-th
candidate of every key at once; per-key kineto durations joined by launch index; the per-key winners re-measured together).
usage: python tools/ab_pgemm.py out.json        (out.json: an ALDM_PGEMM_PATCH overlay to merge into pgemm_gfx950.json)"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench
import tune_pgemm
from audioldm_with_lora_amd import ops
from audioldm_with_lora_amd.engine import DenoiseEngine
from audioldm_with_lora_amd.scheduler import DDIMScheduler

B = int(os.environ.get("ALDM_AB_BATCH", "4"))
unet, _ = bench.build_unet(4)
lat, pe, ne = bench.synth_inputs(B, 250, 16)
BASE = dict(ops.PGEMM_CFG)
KIND = {"s": "proj_in", "vl32": "qkv", "rl32s": "out", "g": "ff1", "": "plain"}


def measure(overlay):
    ops.PGEMM_CFG.clear(); ops.PGEMM_CFG.update(BASE); ops.PGEMM_CFG.update(overlay)
    eng = DenoiseEngine(unet, DDIMScheduler(), B, 250, 16, 200, 2.5)
    eng.set_condition(pe, ne); eng.set_latents(lat)
    eng.capture()
    for _ in range(5):
        eng.step()
    ops.KEYLOG = []
    rows = bench.launch_rows(eng)
    keylog, ops.KEYLOG = dict(ops.KEYLOG), None
    kern, span = bench.trace_replays(eng)
    j = bench.join_trace(rows, kern)
    per, used = {}, {}
    for i, r in enumerate(j):
        k, cfg = keylog.get(i, (None, None))
        if k is not None and k.startswith("pg:"):
            per[k] = per.get(k, 0.0) + r[5]
            used[k] = cfg
    return per, used, sum(r[5] for r in j)


base, used, base_tot = measure({})
print(f"base: kernel time {base_tot:.1f} us; {len(base)} pgemm keys, {sum(base.values()):.1f} us", flush=True)
cands = {}
for k in base:
    M, N, K, kind = k[3:].split("|")
    if kind not in KIND:
        continue
    cands[k] = [c for c in tune_pgemm.configs(KIND[kind], int(M), int(N), int(K)) if tuple(c) != tuple(used[k])]
nvar = max(len(v) for v in cands.values())
results = {k: {tuple(used[k]): base[k]} for k in cands}
for i in range(nvar):
    ov = {}
    for k, cs in cands.items():
        if i < len(cs):
            M, N, K, kind = k[3:].split("|")
            ov[(int(M), int(N), int(K), kind)] = tuple(cs[i])
    try:
        per, _, tot = measure(ov)
    except Exception as e:
        print(f"variant {i}: {str(e)[:120]}", flush=True)
        continue
    for k, cs in cands.items():
        if i < len(cs) and k in per:
            results[k][tuple(cs[i])] = per[k]
    print(f"variant {i}: kernel time {tot:.1f} us", flush=True)
patch = {}
for k, r in results.items():
    best = min(r, key=r.get)
    line = "  ".join(f"{c}:{v:.1f}" for c, v in sorted(r.items(), key=lambda kv: kv[1])[:6])
    print(f"{k:32s} base {tuple(used[k])}:{base[k]:.1f}  ->  {line}")
    if best != tuple(used[k]) and r[best] < 0.97 * base[k]:
        patch[k[3:]] = list(best)
per, _, tot = measure({tuple(int(v) if i < 3 else v for i, v in enumerate(k.split("|"))): tuple(c) for k, c in patch.items()})
print(f"merged overlay ({len(patch)} keys): kernel time {tot:.1f} us (base {base_tot:.1f}); pgemm {sum(per.values()):.1f} us (base {sum(base.values()):.1f})")
json.dump(patch, open(sys.argv[1], "w"), indent=0)
