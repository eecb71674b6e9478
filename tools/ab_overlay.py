"""In-step A/B of launch configurations for a family of igemm keys (config 2 denoise step, replayed graph, kineto durations).

The isolated tuner (tools/autotune.py) ranks a configuration by one launch behind a sleep kernel; inside the step the neighbours'
L2 / Infinity-Cache footprint and the tail of the previous launch change the ranking (round 4: the re-tuned table was 3 % slower
in the step than the old one).  This tool therefore measures IN the step: every variant re-captures the step's graph with a rule
applied to all eligible keys, the per-launch durations are joined by launch index, and the best variant per KEY (summed over its
launches) goes to an overlay JSON that ALDM_TUNED_PATCH or a merge into tuned_gfx950.json can adopt.

usage: python tools/ab_overlay.py out.json [small|big|all] [infer|train]     (train: config 3's LoRA step, batch 8, rank 8; per-key times
from hipEvent pairs of an eager step queued behind a sleep kernel, totals from the replayed graph)"""
import json, os, re, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audioldm_with_lora_amd import ops
from audioldm_with_lora_amd.engine import DenoiseEngine
from audioldm_with_lora_amd.scheduler import DDIMScheduler

out_path = sys.argv[1]
family = sys.argv[2] if len(sys.argv) > 2 else "small"
workload = sys.argv[3] if len(sys.argv) > 3 else "infer"
B = int(os.environ.get("ALDM_AB_BATCH", "4"))
unet, _ = bench.build_unet(4)
lat, pe, ne = bench.synth_inputs(B, 250, 16)
BASE = dict(ops.TUNED)
USED = {}


def key_m(key):
    return int(re.match(r"M(\d+) ", key).group(1))


def eligible(key):
    if not re.match(r"M(\d+) N(\d+) ", key) or any(t not in key for t in (" f1", " r0 ", " vt0 ", " g0 ", " ln0 ")):
        return False
    if family == "halo" and " gi" in key and " d0 " in key:
        return True                                         # GroupNorm-of-the-input launches: halo tiles only, never split
    conv1d = family == "halo" and re.search(r" k1x(3|5|7|11) s1 up0 d0 ", key) is not None
    if any(t in key for t in (" gi", " lp", " rs", " vd")) or (family != "all" and " k3x3 s1 " not in key and not conv1d) or (family == "halo" and " d0 " not in key):
        return False
    M = key_m(key)
    return (256 <= M <= 8192) if family == "small" else M > 8192 if family == "big" else M >= 256


def measure_train(overlay, total=False):
    from audioldm_with_lora_amd.training import LoraTrainer
    ops.TUNED.clear(); ops.TUNED.update(BASE); ops.TUNED.update(overlay)
    g = torch.Generator().manual_seed(0)
    lat, noise = torch.randn(8, 8, 256, 16, generator=g).cuda(), torch.randn(8, 8, 256, 16, generator=g).cuda()
    t = torch.randint(0, 1000, (8,), generator=g).cuda()
    emb = torch.nn.functional.normalize(torch.randn(8, 512, generator=g), dim=-1).cuda()
    tr = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=97000, use_graph=False)
    for _ in range(2):
        tr.step(lat, noise, t, emb)
    torch.cuda.synchronize()
    ops.PROFILE, ops.KEYLOG = [], []
    ops.sleep_us(100000)
    tr.step(lat, noise, t, emb)
    torch.cuda.synchronize()
    rows, keylog, ops.PROFILE, ops.KEYLOG = ops.PROFILE, dict(ops.KEYLOG), None, None
    per_key = {}
    for i, r in enumerate(rows):
        k, cfg = keylog.get(i, (None, None))
        if k is not None:
            per_key[k] = per_key.get(k, 0.0) + r[3].elapsed_time(r[4]) * 1e3
            USED[k] = cfg
    tot = span = float("nan")
    if total:
        trg = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=97000)
        for _ in range(5):
            trg.step(lat, noise, t, emb)
        torch.cuda.synchronize()
        per, span, kept = bench.trace_steps(lambda: trg.step(lat, noise, t, emb), "adamw_flat", reps=6)
        tot = sum(us for _, us in per)
    return per_key, tot, span



def measure(overlay):
    ops.TUNED.clear(); ops.TUNED.update(BASE); ops.TUNED.update(overlay)
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
    per_key = {}
    for i, r in enumerate(j):
        k, cfg = keylog.get(i, (None, None))
        if k is not None:
            us = r[5]
            if " gn" in k and i + 1 < len(j) and j[i + 1][0].startswith("groupnorm"):
                us += j[i + 1][5]                           # a deferred split-K reduce is paid by the GroupNorm that follows
            per_key[k] = per_key.get(k, 0.0) + us
            USED[k] = cfg
    return per_key, sum(r[5] for r in j), span


def measure_vae(overlay, total=False):
    """AutoencoderKL.decode of 4 x 10 s latents, eager, hipEvent pairs queued behind a sleep kernel (bench.py times the same call)."""
    from audioldm_with_lora_amd.vae import AutoencoderKL
    ops.TUNED.clear(); ops.TUNED.update(BASE); ops.TUNED.update(overlay)
    global _vae
    if "_vae" not in globals():
        torch.manual_seed(0)
        _vae = AutoencoderKL().cuda()
    z = torch.randn(4, 250, 16, 8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).to(torch.bfloat16)
    for _ in range(2):
        _vae.decode_nhwc(z)
    torch.cuda.synchronize()
    ops.PROFILE, ops.KEYLOG = [], []
    ops.sleep_us(100000)
    _vae.decode_nhwc(z)
    torch.cuda.synchronize()
    rows, keylog, ops.PROFILE, ops.KEYLOG = ops.PROFILE, dict(ops.KEYLOG), None, None
    per_key, tot = {}, 0.0
    for i, r in enumerate(rows):
        us = r[3].elapsed_time(r[4]) * 1e3
        tot += us
        k, cfg = keylog.get(i, (None, None))
        if k is not None:
            per_key[k] = per_key.get(k, 0.0) + us
            USED[k] = cfg
    return per_key, tot, tot


def measure_voc(overlay, total=False):
    """SpeechT5HifiGan.forward of 4 x 10 s mels, eager, hipEvent pairs queued behind a sleep kernel."""
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    ops.TUNED.clear(); ops.TUNED.update(BASE); ops.TUNED.update(overlay)
    global _voc
    if "_voc" not in globals():
        torch.manual_seed(0)
        _voc = SpeechT5HifiGan().cuda()
    mel = torch.randn(4, 1, 1000, 64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)).to(torch.bfloat16)
    for _ in range(2):
        _voc.forward_nhwc(mel)
    torch.cuda.synchronize()
    ops.PROFILE, ops.KEYLOG = [], []
    ops.sleep_us(100000)
    _voc.forward_nhwc(mel)
    torch.cuda.synchronize()
    rows, keylog, ops.PROFILE, ops.KEYLOG = ops.PROFILE, dict(ops.KEYLOG), None, None
    per_key, tot = {}, 0.0
    for i, r in enumerate(rows):
        us = r[3].elapsed_time(r[4]) * 1e3
        tot += us
        k, cfg = keylog.get(i, (None, None))
        if k is not None:
            per_key[k] = per_key.get(k, 0.0) + us
            USED[k] = cfg
    return per_key, tot, tot


if workload == "voc":
    measure = lambda ov, total=False: measure_voc(ov, total)
    base_keys, base_tot, base_span = measure({})
elif workload == "vae":
    measure = lambda ov, total=False: measure_vae(ov, total)
    base_keys, base_tot, base_span = measure({})
elif workload == "train":
    unet, _ = bench.build_unet(8)
    infer_measure, measure = measure, (lambda ov, total=False: measure_train(ov, total))
    base_keys, base_tot, base_span = measure({}, total=True)
else:
    base_keys, base_tot, base_span = measure({})
keys = [k for k in base_keys if eligible(k)]
print(f"base: kernel time {base_tot:.1f} us, span {base_span:.1f} us; {len(keys)} eligible keys", flush=True)
variants = {"base": (base_keys, {})}
base_used = dict(USED)
SMALL = [("64x128ws r3", 13, 3, 1), ("128x64ws r3", 14, 3, 1), ("128x64ws r4", 14, 4, 1), ("64x128ws r3 /2", 13, 3, 2), ("128x64ws r3 /2", 14, 3, 2),
         ("128x64ws r4 /2", 14, 4, 2)]
BIG = [("256x128ws", 12, 3, 1), ("64x128ws r3", 13, 3, 1), ("128x64ws r3", 14, 3, 1)]
# the halo tiles with split-K by channel chunk: the key's current split count, half of it, twice it
HALO = [(f"{n}{sfx}", t, 3, d) for n, t in (("halo128ws", 15), ("halo64ws", 16), ("halo128", 7), ("halo64", 8))
        for sfx, d in (("", 1), (" /2", 2), (" x2", 0.5), (" 1", 1000))] + [("halo128ws r4", 15, 4, 1), ("halo64ws r4", 16, 4, 1)]
rules = SMALL if family == "small" else BIG if family == "big" else HALO if family == "halo" else SMALL + [("256x128ws", 12, 3, 1)]
for name, tile, ring, spdiv in rules:
    ov = {}
    for k in keys:
        sp = base_used[k][2]
        if tile in (7, 8, 15, 16):
            ow = int(re.search(r" ow(\d+)", k).group(1))
            m1 = re.search(r" k1x(\d+) s1 .* dl1x(\d+)", k)
            if m1:                                              # conv1d: the wave-specialised halo tiles, halo = BM + (K - 1) dil rows
                if tile not in (15, 16) or ops.TILE_DIMS[tile][0] + (int(m1.group(1)) - 1) * int(m1.group(2)) > ops.HALO_ROWS[tile]:
                    continue
            elif tile not in ops.halo_tiles(ow, True):
                continue
            if re.search(r" e\d+\+\d+", k):                # conv2 + conv_shortcut: wave-specialised halo tiles, three-pass halo, ring 3
                if tile not in (15, 16) or ring != 3 or " up0 " not in k or (tile == 15 and (128 // ow + 2) * (ow + 2) > 192):
                    continue
            nch = sum(int(v) for v in re.search(r" C(\d+)\+(\d+) ", k).groups()) // 64
            nsp = 1 if " gi" in k else max(1, min(nch, int(round(sp / spdiv))))
            if (nsp == sp and spdiv != 1) or (tile, ring, nsp) == tuple(base_used[k]):
                continue
            ov[k] = (tile, ring, nsp)
            continue
        if (spdiv > 1 and sp < 2 * spdiv) or (tile == 12 and key_m(k) <= 8192):
            continue
        ov[k] = (tile, ring, max(1, sp // spdiv))
    if not ov:
        continue
    try:
        pk, tot, span = measure(ov)
    except Exception as e:                                  # a key the tile refuses: drop the variant, say why
        print(f"{name}: {str(e)[:160]}", flush=True)
        continue
    variants[name] = (pk, ov)
    print(f"{name}: kernel time {tot:.1f} us, span {span:.1f} us", flush=True)
patch = {}
for k in keys:
    best = min((v for v in variants if v == "base" or k in variants[v][1]), key=lambda v: variants[v][0].get(k, 1e9))
    line = "  ".join(f"{v} {variants[v][0].get(k, float('nan')) if (v == 'base' or k in variants[v][1]) else float('nan'):7.2f}" for v in variants)
    print(f"{k:100s} {line}  -> {best}")
    if best != "base" and variants[best][0][k] < 0.985 * base_keys[k]:
        patch[k] = list(variants[best][1][k])
pk, tot, span = measure({k: tuple(v) for k, v in patch.items()}, True) if workload != "infer" else measure({k: tuple(v) for k, v in patch.items()})
print(f"merged overlay ({len(patch)} keys): kernel time {tot:.1f} us (base {base_tot:.1f}), span {span:.1f} us (base {base_span:.1f})")
with open(out_path, "w") as f:
    json.dump(patch, f, indent=0)
