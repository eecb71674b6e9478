#!/usr/bin/env python
"""Headline benchmark: UNet denoise-steps/sec (BASELINE.json metric, config[1]).

Workload (config 2 of BASELINE.json / SURVEY.md 8d): cvssp/audioldm-s-full-v2 architecture with random-init
weights (no checkpoints offline), rank-4 LoRA fused into attention to_q/to_k/to_v/to_out.0, batch 4 prompts
x 10 s clips => latents [4, 8, 250, 16], classifier-free guidance (UNet batch 8), bf16, DDIM eta = 0.
A "step" = one denoise step for the whole per-GPU batch: CFG-doubled UNet forward + guidance + scheduler
step, inputs resident in HBM.  One process per GPU; with N > 1 every rank runs its own prompts (prompt-sharded,
no collective in the loop -- SURVEY.md 8e), `value` = total steps / max-over-ranks time.

    python bench.py --gpus 1 --steps 200 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Per-kernel durations come from two sources, both printed:
  * `trace`  -- the device-side begin/end timestamps of every kernel of the REPLAYED graph (torch.profiler = roctracer in
                process; the same timestamps `rocprofv3 --kernel-trace` reports -- profiles/r02*_kernel_trace_summary.txt is
                that command's summary).  `roofline.achieved` and `k1` are computed from these.
  * `events` -- hipEvent pairs around every launch of one eager step queued behind a sleep kernel (events on the launch stream).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def build_unet(rank_lora=4, device="cuda"):
    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    torch.manual_seed(1234)
    unet = UNet2DConditionModel()
    peft = get_peft_model(unet, LoraConfig(r=rank_lora, lora_alpha=rank_lora,
                                           target_modules=["to_q", "to_k", "to_v", "to_out.0"],
                                           init_lora_weights="gaussian"))
    g = torch.Generator().manual_seed(4)
    for n, p in unet.named_parameters():
        if "lora_B" in n:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.02)     # non-zero, else the adapter is a no-op
    unet.to(device)
    return unet, peft


def synth_inputs(batch, height, width, seed_off=0):
    g = torch.Generator().manual_seed(0 + seed_off)
    lat = torch.randn(batch, 8, height, width, generator=g)
    pe = torch.nn.functional.normalize(torch.randn(batch, 512, generator=torch.Generator().manual_seed(1 + seed_off)), dim=-1)
    ne = torch.nn.functional.normalize(torch.randn(batch, 512, generator=torch.Generator().manual_seed(2 + seed_off)), dim=-1)
    return lat, pe, ne


# ---- per-launch measurement -----------------------------------------------------------------------------------------
def launch_rows(engine):
    """One eager denoise step with a hipEvent pair (on the launch stream) around every C-ABI call, queued behind a sleep kernel
    so that the pairs time kernels and not host gaps.  Returns [(label, flops, bytes, event_us, site)] in launch order."""
    from audioldm_with_lora_amd import ops
    ops.PROFILE = []
    ops.sleep_us(60000)
    engine._one_step()
    torch.cuda.synchronize()
    rows, ops.PROFILE = ops.PROFILE, None
    return [(label, flops, nbytes, s.elapsed_time(e) * 1e3, site) for label, flops, nbytes, s, e, site in rows]


def trace_steps(step_fn, end_marker, start_marker=None, reps=6):
    """Device timestamps of every kernel of `reps` calls of step_fn (kineto / roctracer), cut into steps at the kernel whose name
    contains end_marker (the step's LAST launch; start_marker = its first, used when no end marker shows up).  Only COMPLETE steps
    with identical kernel sequences are kept -- the tracer may drop or add events at the edges of its window, so dividing totals by
    `reps` is wrong.  Returns (kernels per step as [(name, avg duration us)], avg wall span of one step in us, steps kept) or
    (None, None, 0) when the tracer is unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile
        step_fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(reps):
                step_fn()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name and "Memset" not in e.name]
        evs.sort(key=lambda e: e.time_range.start)
        ends = [i for i, e in enumerate(evs) if end_marker in e.name]
        if ends:
            starts = [0] + [i + 1 for i in ends[:-1]]
            segs = [evs[a:b + 1] for a, b in zip(starts, ends)][1:]      # the first segment may be cut by the tracer's window
        elif start_marker:
            starts = [i for i, e in enumerate(evs) if start_marker in e.name]
            segs = [evs[a:b] for a, b in zip(starts, starts[1:] + [len(evs)])]
        else:
            segs = []
        if not segs:
            print(f"[bench] kernel trace: no step boundary among {len(evs)} device events", file=sys.stderr)
            return None, None, 0
        lens = sorted(len(g) for g in segs)
        n = lens[len(lens) // 2]
        segs = [g for g in segs if len(g) == n and [e.name for e in g] == [e.name for e in segs[[len(x) for x in segs].index(n)]]]
        if len(segs) < 2:
            print(f"[bench] kernel trace: steps do not line up ({lens})", file=sys.stderr)
            return None, None, 0
        per = [(segs[0][i].name, sum(g[i].time_range.elapsed_us() for g in segs) / len(segs)) for i in range(n)]
        span = sum(g[-1].time_range.end - g[0].time_range.start for g in segs) / len(segs)
        return per, span, len(segs)
    except Exception as ex:                                  # measurement aid only: never let it take the bench line down
        print(f"[bench] kernel trace unavailable: {type(ex).__name__}: {ex}", file=sys.stderr)
        return None, None, 0


def trace_replays(engine, reps=6):
    """One denoise step's kernels from replayed graphs: the step's LAST kernel is the fused guidance / DDIM / counter launch; with
    several chains it still opens with gather_row."""
    per, span, _ = trace_steps(engine.step, "ddim_step_fused_kernel", "gather_row_kernel", reps)
    return per, span


def launch_floor_us(n=200, full_grid=False):
    """Cost of one kernel boundary inside a replayed hipGraph: n launches captured back to back, replay time / n.  Default: a
    one-wave kernel that does nothing (aldm_sleep_us(0)) -- the bare boundary.  full_grid: the cheapest USEFUL launch, a 512-workgroup
    elementwise pass over 1 MB (fp32 -> bf16), i.e. boundary + filling and draining the whole chip once."""
    from audioldm_with_lora_amd import ops
    src = torch.zeros(512, 512, device="cuda")
    dst = torch.empty(512, 512, dtype=torch.bfloat16, device="cuda")
    one = (lambda: ops.f32_to_bf16(src, out=dst)) if full_grid else (lambda: ops.sleep_us(0))
    one()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            one()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


def floor_model(rows, step_ms, boundary_us, small_us):
    """Where the step's time can NOT go away at this launch structure: every launch pays the graph's kernel boundary, then its own
    roofline time max(flops / MFMA peak, algorithmic bytes / HBM peak).  `structure_floor_ms` keeps the 246-launch structure and
    makes every kernel perfect; `fused_floor_ms` is the same work as ONE launch.  ms_per_step / structure_floor_ms is the number
    to push towards 1 by kernel work; structure_floor_ms itself only moves by removing launches."""
    roof_us = sum(max(r[1] / (PEAK_BF16_TFLOPS * 1e6), r[2] / (PEAK_HBM_GBS * 1e3)) for r in rows)
    n = len(rows)
    structure = n * boundary_us + roof_us
    return {"launches": n, "boundary_us_per_launch": round(boundary_us, 2), "boundaries_ms": round(n * boundary_us / 1e3, 3),
            "smallest_useful_launch_us": round(small_us, 2), "launches_at_smallest_ms": round(n * small_us / 1e3, 3),
            "roofline_sum_ms": round(roof_us / 1e3, 3), "structure_floor_ms": round(structure / 1e3, 3),
            "fused_floor_ms": round((boundary_us + roof_us) / 1e3, 3),
            "step_over_structure_floor": round(step_ms / (structure / 1e3), 2),
            "peaks": {"mfma_bf16_tflops": PEAK_BF16_TFLOPS, "hbm_gbs": PEAK_HBM_GBS}}


def join_trace(rows, kernels):
    """Attach the traced kernel durations to the launch rows (same launch order; a split-K igemm call is two kernels)."""
    out, k = [], 0
    for label, flops, nbytes, ev_us, site in rows:
        if k >= len(kernels):
            print(f"[bench] kernel trace: {len(kernels)} kernels per replay < {len(rows)} launch rows", file=sys.stderr)
            return None
        us, names = kernels[k][1], [kernels[k]]
        k += 1
        if k < len(kernels) and "igemm_reduce_kernel" in kernels[k][0] and label.startswith("igemm"):
            us += kernels[k][1]
            names.append(kernels[k])
            k += 1
        out.append((label, flops, nbytes, ev_us, site, us, names))
    if k != len(kernels):
        print(f"[bench] kernel trace: {len(kernels)} kernels per replay, {k} matched to {len(rows)} launch rows; next: {kernels[k][0][:80]}", file=sys.stderr)
        return None
    return out


def short_symbol(name):
    """Kernel symbol as rocprofv3's summary prints it: template arguments kept, namespaces / parameter list dropped."""
    import re
    name = name.replace("(anonymous namespace)::", "").replace("aldm_igemm_detail::", "").replace("void ", "")
    m = re.match(r"(\w+<[^>]*>)", name)
    return m.group(1) if m else name.split("(")[0]


def aggregate(rows, fine=False, by_symbol=False):
    """Totals per label family (default), per label (fine) or per kernel SYMBOL (by_symbol: one entry per template instantiation,
    exactly the rows of a rocprofv3 --stats summary; flops / bytes of a launch go to its first kernel, a split-K reduce kernel
    is its own entry)."""
    agg = {}
    for label, flops, nbytes, ev_us, site, us, names in rows:
        if by_symbol:
            for i, (nm, kus) in enumerate(names):
                a = agg.setdefault(short_symbol(nm), dict(us=0.0, ev_us=0.0, flops=0.0, bytes=0.0, launches=0, kernel=nm, label=label.split("|")[0]))
                a["us"] += kus
                a["launches"] += 1
                if i == 0:
                    a["ev_us"] += ev_us
                    a["flops"] += flops
                    a["bytes"] += nbytes
            continue
        key = label if fine else label.split("|")[0]
        a = agg.setdefault(key, dict(us=0.0, ev_us=0.0, flops=0.0, bytes=0.0, launches=0, kernel=names[0][0]))
        a["us"] += us
        a["ev_us"] += ev_us
        a["flops"] += flops
        a["bytes"] += nbytes
        a["launches"] += 1
    return agg


def k1_sites(rows, ubatch, rank_lora):
    """The fused-LoRA attention module: QKV GEMM + LoRA -> flash attention -> out-proj + LoRA + residual,
    priced per site with the survey's algorithmic work  b(8NC^2 + 4N^2C + 16NCr) flops,  2bNCs + 4C^2 s + 8Crs bytes (s = 2)."""
    sites = {}
    for label, flops, nbytes, ev_us, site, us, names in rows:
        if not site:
            continue
        s = sites.setdefault(site, dict(modules=0, us=0.0, ev_us=0.0, parts={}))
        part = ("qkv_lora+attention" if label.startswith("attn_block") else "attention" if label.startswith("attention")
                else ("qkv_lora" if "_vt" in label else "out_proj_lora"))
        s["parts"][part] = s["parts"].get(part, 0.0) + us
        s["us"] += us
        s["ev_us"] += ev_us
        s["modules"] += 1 if "attention" in part else 0
    out = []
    for site, s in sorted(sites.items()):
        C, N = int(site.split()[1][1:]), int(site.split()[2][1:])
        m = s["modules"]
        flops = ubatch * (8.0 * N * C * C + 4.0 * N * N * C + 16.0 * N * C * rank_lora)
        nbytes = 2.0 * ubatch * N * C * 2 + 4.0 * C * C * 2 + 8.0 * C * rank_lora * 2
        t = s["us"] / m * 1e-6
        tf, gbs = flops / t / 1e12, nbytes / t / 1e9
        bound = "mfma" if flops / nbytes > PEAK_BF16_TFLOPS * 1e3 / PEAK_HBM_GBS * 1.5 else "hbm"
        out.append({"site": site, "modules_per_step": m, "us_per_module": round(s["us"] / m, 2),
                    "us_per_module_events": round(s["ev_us"] / m, 2),
                    "parts_us": {k: round(v / m, 2) for k, v in s["parts"].items()},
                    "algorithmic_gflop": round(flops / 1e9, 3), "algorithmic_mb": round(nbytes / 1e6, 3),
                    "tflops": round(tf, 1), "frac_mfma": round(tf / PEAK_BF16_TFLOPS, 4),
                    "gb_per_s": round(gbs, 1), "frac_hbm": round(gbs / PEAK_HBM_GBS, 4), "bound": bound,
                    "frac": round(tf / PEAK_BF16_TFLOPS if bound == "mfma" else gbs / PEAK_HBM_GBS, 4)})
    return out


def traffic_from_profile(kernel_name):
    """HBM bytes per launch of a kernel from the committed PMC summary (tools/pmc_traffic.py over two rocprofv3 --pmc passes of
    this same command; MI355X_MICROARCH.md correction applied).  None when no summary holds that kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    prof = json.load(open(files[-1]))
    short = short_symbol(kernel_name)
    if short in prof:
        return prof[short]["hbm_bytes_per_launch"], os.path.basename(files[-1])
    return None, os.path.basename(files[-1])


def cpu_baseline(batch, height, width, rank_lora, max_seconds=30.0):
    """The CPU oracle (kind "port": our restatement of the diffusers/peft path) timed on this box's host cores."""
    from oracle import lora as olora
    from oracle.unet import UNet2DConditionModel as OracleUNet
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))          # a 1-GPU box's CPU share is 16 cores; never oversubscribe
    torch.set_num_threads(cores)
    torch.manual_seed(1234)
    u = OracleUNet().eval()
    olora.get_peft_model(u, olora.LoraConfig(r=rank_lora, lora_alpha=rank_lora,
                                             target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    lat, pe, ne = synth_inputs(batch, height, width)
    x = torch.cat([lat, lat])
    emb = torch.cat([ne, pe])
    times = []
    with torch.no_grad():
        t0 = time.time()
        u(x, torch.tensor(996), class_labels=emb)           # warm-up
        budget = max_seconds - (time.time() - t0)
        while len(times) < 3 and (not times or sum(times) + times[-1] < budget):
            t1 = time.time()
            eps = u(x, torch.tensor(991), class_labels=emb)[0]
            eu, et = eps.chunk(2)
            _ = eu + 2.5 * (et - eu)
            times.append(time.time() - t1)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 1.0 / med, "unit": "denoise_steps/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} CFG-doubled UNet forwards (b=8, latent {height}x{width}, fp32, rank-{rank_lora} LoRA) "
                      f"after 1 warm-up; median {med:.2f} s/step"}


def bench_pipeline(unet, batch, seconds, nsteps, guidance, seed_off=0):
    """A REAL `pipe(...)` call, timed end to end: prompt embeddings + initial noise in, host numpy audio out -- set_condition (the
    time-embedding table of all steps), the `nsteps` graph replays, VAE decode, vocoder and the D2H copy all inside the timed
    call [REF script/inference/generate_audio.py:47-52] [REF app.py:14].  The first call (which captures the graph) is timed
    separately.  Random-init weights of the real VAE / vocoder architecture."""
    from audioldm_with_lora_amd.pipeline import AudioLDMPipeline
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    torch.manual_seed(99)
    pipe = AudioLDMPipeline(AutoencoderKL(), None, None, unet, DDIMScheduler(), SpeechT5HifiGan())
    pipe.device = torch.device("cuda")
    pipe.vae.cuda(); pipe.vocoder.cuda()                 # (the UNet is already on the device: keep its packed plan)
    h = int(seconds * 100) // 4
    lat, pe, ne = synth_inputs(batch, h, 16, seed_off)
    call = dict(prompt_embeds=pe, negative_prompt_embeds=ne, audio_length_in_s=seconds, num_inference_steps=nsteps,
                guidance_scale=guidance)
    t0 = time.perf_counter()
    pipe(latents=lat.clone(), **call)
    first = time.perf_counter() - t0
    times = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        audio = pipe(latents=lat.clone(), **call).audios
        times.append(time.perf_counter() - t0)
    assert audio.shape == (batch, int(seconds * 16000)) and bool((audio == audio).all())
    # the decode legs alone (eager launches)
    z = torch.randn(batch, 8, h, 16, device="cuda")
    legs = {}
    mel = pipe.vae.decode(z).sample.squeeze(1)
    for name, fn in (("vae_decode_ms", lambda: pipe.vae.decode(z).sample), ("vocoder_ms", lambda: pipe.vocoder(mel))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        legs[name] = (time.perf_counter() - t0) / 3 * 1e3
    del pipe
    torch.cuda.empty_cache()
    return min(times), first, legs


TRAIN_FAMILIES = (("attention_bwd", ("attn_bwd",)), ("attention_fwd", ("attention_kernel", "attn_block")),
                  ("gemm_lora_site", ("igemm_pipe_kernel<64, 64, 2, 2, 32", "igemm_pipe_kernel<128, 64, 2, 2, 32", "igemm_pipe_kernel<64, 128, 2, 2, 32",
                                      "igemm_pipe_kernel<128, 128, 2, 2, 32", "igemm_pipe_kernel<64, 64, 2, 2, 64", "igemm_pipe_kernel<128, 64, 2, 2, 64",
                                      "igemm_pipe_kernel<64, 128, 2, 2, 64", "igemm_pipe_kernel<128, 128, 2, 2, 64", "pgemm_kernel")),
                  ("gemm", ("igemm_pipe_kernel", "igemm_kernel", "igemm_halo_kernel", "igemm_halo_ws_kernel", "igemm_ws_kernel")), ("splitk_reduce", ("igemm_reduce",)),
                  ("groupnorm_bwd", ("groupnorm_bwd",)), ("groupnorm_fwd", ("groupnorm", "gn_silu")),
                  ("layernorm_geglu", ("layernorm", "geglu")), ("lora_grad", ("tn_mfma", "tn_small", "lora_pack")),
                  ("transpose", ("transpose_tokens",)), ("optimizer", ("adamw",)))


def train_families(per):
    """ms per step by kernel family (first matching entry of TRAIN_FAMILIES; everything else -> other)."""
    out = {}
    for name, us in per:
        fam = next((f for f, keys in TRAIN_FAMILIES if any(k in name for k in keys)), "other")
        out[fam] = out.get(fam, 0.0) + us / 1e3
    return {k: round(v, 3) for k, v in sorted(out.items(), key=lambda kv: -kv[1])}


def bench_train(world, rank, steps=8, warmup=3, batch=8, rank_lora=8, trace=False):
    """Config 3/4: LoRA fine-tune step (add_noise -> UNet fwd -> MSE -> bwd -> flat all-reduce -> AdamW) on synthetic
    10.24 s mel latents [8, 8, 256, 16] per GPU; clips/s = world * batch / step time."""
    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.training import LoraTrainer
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    import torch.distributed as dist
    torch.manual_seed(1234)
    unet = UNet2DConditionModel()
    get_peft_model(unet, LoraConfig(r=rank_lora, lora_alpha=rank_lora, init_lora_weights="gaussian",
                                    target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    unet.cuda()
    tr = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=97000)
    g = torch.Generator().manual_seed(5 + rank)
    lat = (torch.randn(batch, 8, 256, 16, generator=g) * 0.9228).cuda()
    noise = torch.randn(batch, 8, 256, 16, generator=g).cuda()
    t = torch.randint(0, 1000, (batch,), generator=g).cuda()
    emb = torch.nn.functional.normalize(torch.randn(batch, 512, generator=g), dim=-1).cuda()
    for _ in range(warmup):
        tr.step(lat, noise, t, emb)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(lat, noise, t, emb)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt = float(tm.item())
    final = float(loss)
    extra = {}
    if trace and rank == 0:
        # the step's kernels from complete replays only (its last launch is the flat AdamW); families by symbol
        per, span, kept = trace_steps(lambda: tr.step(lat, noise, t, emb), "adamw_flat", reps=5)
        if per:
            extra = {"launches_per_step_total": len(per), "kernel_time_sum_ms": round(sum(u for _, u in per) / 1e3, 3),
                     "graph_span_ms": round(span / 1e3, 3), "trace_steps_kept": kept, "families_ms": train_families(per)}
    del tr, unet
    torch.cuda.empty_cache()
    return {"metric": "lora_train_clips_per_sec", "value": round(world * batch * steps / dt, 3), "unit": "10.24s-clips/s",
            "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "per_gpu_batch": batch, "lora_rank": rank_lora,
            "lora_params": 112640 * rank_lora, "final_loss": round(final, 5), **extra,
            "collective": "1 flat fp32 all-reduce/step (RCCL)" if world > 1 else "none",
            "dtype": TRAIN_DTYPE, "gradient_tolerance": TRAIN_TOL}


def bench_variant(unet, batch, guidance, steps, warmup, fp8, rank_lora, k1=False):
    """A secondary denoise-loop leg on the same UNet (own engine / graph): ms per step, optionally the K1 sites."""
    from audioldm_with_lora_amd.engine import DenoiseEngine
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    unet.attention_fp8 = fp8
    try:
        eng = DenoiseEngine(unet, DDIMScheduler(), batch, 250, 16, 200 if steps > 50 else 50, guidance)
        lat, pe, ne = synth_inputs(batch, 250, 16, seed_off=7)
        eng.set_condition(pe, ne)
        eng.set_latents(lat)
        eng.capture()
        for _ in range(warmup):
            eng.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert torch.isfinite(eng.x).all(), "latents diverged"
        out = {"ms_per_step": round(dt / steps * 1e3, 4), "denoise_steps_per_sec": round(steps / dt, 2), "steps": steps,
               "per_gpu_batch": batch, "unet_batch": 2 * batch if guidance > 1.0 else batch, "guidance_scale": guidance}
        if k1:
            kernels, _ = trace_replays(eng)
            ev_rows = launch_rows(eng)
            rows = join_trace(ev_rows, kernels) if kernels else None
            if rows is None:
                rows = [(l, f, b, e, s, e, [(l.split("|")[0], e)]) for l, f, b, e, s in ev_rows]
            out["k1_sites"] = [{k: v for k, v in site.items() if k in ("site", "us_per_module", "parts_us", "tflops", "frac", "bound")}
                               for site in k1_sites(rows, 2 * batch, rank_lora)]
        eng.graph = None
        return out
    finally:
        unet.attention_fp8 = False


def bench_loop_body(steps=6, warmup=4, batch=8, rank_lora=8):
    """The reference's whole loop body per collate_fn batch [REF script/train/train_audioldm_lora.py:495-565]: log-mel [8,1,1024,64]
    -> vae.encode -> latent_dist.sample() * scaling_factor; token ids (captions of 8..64 tokens padded to 512) -> CLAP text tower ->
    normalize; add_noise -> UNet fwd -> MSE -> bwd -> AdamW -- as ONE captured hipGraph (LoraTrainer.step_from_batch)."""
    from audioldm_with_lora_amd.clap_text import ClapTextModelWithProjection
    from audioldm_with_lora_amd.lora import LoraConfig, get_peft_model
    from audioldm_with_lora_amd.scheduler import DDIMScheduler
    from audioldm_with_lora_amd.script.train import synthetic_batch
    from audioldm_with_lora_amd.training import LoraTrainer
    from audioldm_with_lora_amd.unet import UNet2DConditionModel
    from audioldm_with_lora_amd.vae import AutoencoderKL
    torch.manual_seed(1234)
    unet = UNet2DConditionModel()
    get_peft_model(unet, LoraConfig(r=rank_lora, lora_alpha=rank_lora, init_lora_weights="gaussian",
                                    target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    unet.cuda()
    vae = AutoencoderKL().requires_grad_(False).cuda()
    clap = ClapTextModelWithProjection().requires_grad_(False).cuda()
    tr = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=97000)
    g = torch.Generator().manual_seed(21)
    b = synthetic_batch(batch, g, vocab=clap.cfg["vocab_size"])
    noise = torch.randn(batch, 8, 256, 16, generator=g)
    eps = torch.randn(batch, 8, 256, 16, generator=g)
    t = torch.randint(0, 1000, (batch,), generator=g)
    for _ in range(warmup):
        tr.step_from_batch(vae, clap, b, noise, t, eps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step_from_batch(vae, clap, b, noise, t, eps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    final = float(loss)
    assert final == final
    del tr, unet, vae, clap
    torch.cuda.empty_cache()
    return {"what": "mel -> VAE encode -> sample, ids -> CLAP text tower -> normalize, add_noise -> UNet fwd -> MSE -> bwd -> AdamW; "
                    "one captured hipGraph per step [REF train:495-565]",
            "ms_per_step": round(dt / steps * 1e3, 2), "clips_per_sec": round(batch * steps / dt, 2), "per_gpu_batch": batch,
            "lora_rank": rank_lora, "steps": steps, "final_loss": round(final, 5), "dtype": TRAIN_DTYPE}


TRAIN_DTYPE = ("bf16 activations and activation gradients, fp32 accumulation / statistics / loss, fp32 LoRA master weights, gradients "
               "and AdamW (the reference trains in fp32 [REF train:329,389]: this leg is a mixed-precision number)")
TRAIN_TOL = "flat LoRA gradient vs fp32 autograd: rel. L2 <= 6e-2 (tests/test_gpu_train_fullsize.py); AdamW vs torch.optim.AdamW 2e-6"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--rank", type=int, default=4)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--chains", type=int, default=None, help="independent sub-batch chains captured as parallel graph branches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the LoRA-training and end-to-end pipeline legs")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--fine", action="store_true", help="print the per-kernel-per-shape table to stderr")
    ap.add_argument("--no-trace", action="store_true", help="skip the in-process kernel tracer (use under rocprofv3: one tracer at a time)")
    ap.add_argument("--fp8-attention", action="store_true", help="BASELINE config 5: fp8 (e4m3) Q/K/V/P attention operands")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from audioldm_with_lora_amd.engine import DenoiseEngine
    from audioldm_with_lora_amd.scheduler import DDIMScheduler

    H, W, NSTEPS, G = 250, 16, 200, 2.5
    unet, _ = build_unet(args.rank)
    unet.attention_fp8 = args.fp8_attention
    eng = DenoiseEngine(unet, DDIMScheduler(), args.batch, H, W, NSTEPS, G, use_graph=not args.no_graph, chains=args.chains)
    lat, pe, ne = synth_inputs(args.batch, H, W, seed_off=100 * rank)
    eng.set_condition(pe, ne)
    eng.set_latents(lat)
    eng.capture()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(eng.x).all(), "latents diverged"

    # ---- per-kernel measurement on rank 0 (after the timed region: the tracer must not perturb `value`) ----
    rows = kernels = span_us = None
    if rank == 0:
        print(f"[bench] {args.steps} steps in {dt:.3f} s -> {dt / args.steps * 1e3:.3f} ms/step", file=sys.stderr, flush=True)
        if eng.graph is not None and not args.no_trace:
            kernels, span_us = trace_replays(eng)
        ev_rows = launch_rows(eng)
        rows = join_trace(ev_rows, kernels) if kernels else None
        source = "trace"
        if rows is None:                                     # tracer missing or the sequence did not line up: events only
            source = "events"
            rows = [(l, f, b, e, s, e, [(l.split("|")[0], e)]) for l, f, b, e, s in ev_rows]
            print("[bench] per-kernel durations fall back to hipEvent pairs", file=sys.stderr)

    boundary_us = launch_floor_us() if rank == 0 else 0.0
    small_us = launch_floor_us(full_grid=True) if rank == 0 else 0.0
    pipe_t = None
    if not args.no_train and rank == 0 and not args.fp8_attention and args.chains in (None, 1):
        pipe_t = bench_pipeline(unet, args.batch, 10.0, NSTEPS, G, seed_off=100 * rank)
    eng.graph = None
    fp8_leg = b1_leg = b1_pipe = body_leg = None
    if not args.no_train and rank == 0 and world == 1 and not args.fp8_attention and args.chains in (None, 1):
        # BASELINE config 5's per-GPU work: the same loop with e4m3 attention operands (prompt-sharded: no collective)
        fp8_leg = bench_variant(unet, args.batch, G, 50, 5, True, args.rank, k1=True)
        # what the reference's entry points run: one prompt, 50 steps, guidance 5.0 [REF script/inference/generate_audio.py:47-52]
        b1_leg = bench_variant(unet, 1, 5.0, 50, 5, False, args.rank)
        b1_pipe = bench_pipeline(unet, 1, 10.0, 50, 5.0, seed_off=3)
    train = train16 = None
    if not args.no_train:
        torch.cuda.empty_cache()
        # config 3 (rank 8) on one GPU, config 4 (rank 16, one flat all-reduce per step) data-parallel; on one GPU a short
        # rank-16 leg also runs so that config 4's per-GPU path (the Rp = 64 kernels) is exercised before any N > 1 run
        train = bench_train(world, rank, rank_lora=(8 if world == 1 else 16), trace=not args.no_trace)
        if world == 1:
            train16 = bench_train(world, rank, steps=4, warmup=3, rank_lora=16)
            body_leg = bench_loop_body()
    if rank == 0:
        agg = aggregate(rows)
        total_us = sum(v["us"] for v in agg.values())
        # the dominant kernel = the SYMBOL (template instantiation) with the largest share of the step -- one row of the rocprofv3
        # summary in profiles/, so `frac` can be recomputed from that file: achieved = flops per launch / its average duration
        sym = aggregate(rows, by_symbol=True)
        dom_label, dom = max(((k, v) for k, v in sym.items() if v["flops"] > 0), key=lambda kv: kv[1]["us"])
        if args.fine:
            for k, v in sorted(aggregate(rows, fine=True).items(), key=lambda kv: -kv[1]["us"]):
                print(f"{k:64s} {v['launches']:3d}x {v['us'] / v['launches']:7.1f} us (events {v['ev_us'] / v['launches']:7.1f})  "
                      f"{v['flops'] / v['us'] / 1e6 if v['us'] else 0:7.1f} TF/s", file=sys.stderr)
        if args.breakdown:
            for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
                print(f"{k:32s} {v['launches']:4d} launches {v['us'] / 1e3:8.3f} ms  {v['flops'] / v['us'] / 1e6 if v['us'] else 0:8.1f} TF/s "
                      f"{v['bytes'] / v['us'] / 1e3 if v['us'] else 0:8.1f} GB/s", file=sys.stderr)
            print(f"kernel sum {total_us / 1e3:.3f} ms over {sum(v['launches'] for v in agg.values())} launches ({source})", file=sys.stderr)
        achieved = dom["flops"] / (dom["us"] * 1e-6) / 1e12
        traffic, traffic_src = traffic_from_profile(dom["kernel"])
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": (round(traffic) if traffic else None),
                    "traffic_unit": "HBM bytes/launch (PMC, profiles/%s)" % traffic_src if traffic else None,
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "algorithmic_flops_per_launch": round(dom["flops"] / dom["launches"]),
                    "kernel": dom_label, "kernel_family": dom.get("label"), "launches_per_step": dom["launches"],
                    "avg_launch_us": round(dom["us"] / dom["launches"], 2),
                    "avg_launch_us_events": round(dom["ev_us"] / dom["launches"], 2),
                    "duration_source": ("device timestamps of the replayed graph's kernels (roctracer; = rocprofv3 --kernel-trace)"
                                        if source == "trace" else "hipEvent pairs, eager step behind a sleep kernel"),
                    "share_of_step": round(dom["us"] / total_us, 3),
                    "launches_per_step_total": len(rows), "kernel_time_sum_ms": round(total_us / 1e3, 4),
                    "graph_span_ms": round(span_us / 1e3, 4) if span_us else None,
                    "whole_step_tflops": round(8 * 101.2e9 * (args.batch / 4) / (dt / args.steps) / 1e12, 1)}
        floor = floor_model(rows, dt / args.steps * 1e3, boundary_us, small_us)
        out = {
            "metric": "unet_denoise_steps_per_sec", "value": round(world * args.steps / dt, 3), "unit": "denoise_steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if not args.fp8_attention else "bf16 (fp8 e4m3 attention operands)", "data": "synthetic",
            "config": {"workload": "audioldm-s-full-v2 UNet + rank-%d LoRA (q,k,v,out), DDIM 200-step schedule, "
                                   "batch %d x 10 s clips (latent 250x16), CFG 2.5 (UNet batch %d), bf16" % (args.rank, args.batch, 2 * args.batch),
                       "per_gpu_batch": args.batch, "parallelism": f"prompt-sharded x{world}", "hip_graph": not args.no_graph, "graph_chains": eng.chains,
                       "sample_steps_per_sec": round(world * args.steps * args.batch / dt, 2)},
            "roofline": roofline,
            "floor": floor,
            "k1": {"what": "fused-LoRA attention module = QKV GEMM + LoRA (Q|K, V^T) -> flash attention -> out-proj + LoRA + residual; "
                           "kernel durations only (the two kernel boundaries inside a module are not counted)",
                   "flops": "b(8NC^2 + 4N^2C + 16NCr)", "bytes": "2bNCs + 4C^2 s + 8Crs, s = 2",
                   "sites": k1_sites(rows, 2 * args.batch, args.rank)},
        }
        if train is not None:
            out["train"] = train
        if train16 is not None:
            out["train_rank16"] = train16
        if pipe_t is not None:
            call_s, first_s, legs = pipe_t
            out["end_to_end"] = {"clips_per_sec": round(world * args.batch / call_s, 3), "clip_seconds": 10.0, "ddim_steps": NSTEPS,
                                 "pipe_call_ms": round(call_s * 1e3, 1), "first_call_ms_incl_capture": round(first_s * 1e3, 1),
                                 "vae_decode_ms": round(legs["vae_decode_ms"], 2), "vocoder_ms": round(legs["vocoder_ms"], 2),
                                 # 654.0 / 1002.8 GFLOP per 10 s clip; fraction of the dense bf16 MFMA peak
                                 "vae_frac": round(args.batch * 654.0e9 / (legs["vae_decode_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                 "vocoder_frac": round(args.batch * 1002.8e9 / (legs["vocoder_ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                 "note": "pipe_call_ms = one timed AudioLDMPipeline.__call__ (prompt embeddings + noise in, host audio "
                                         "out): set_condition + 200 graph replays + VAE decode + vocoder + D2H"}
        if fp8_leg is not None:
            fp8_leg["what"] = ("BASELINE config 5 per GPU: the timed loop with fp8 (OCP e4m3) Q / K / V / P attention operands, fp32 accumulation; "
                               "a precision variant, not a faster one (compare ms_per_step with the headline)")
            fp8_leg["tolerance"] = "full-width UNet vs fp32 oracle rel. L2 <= 6e-2 (tests/test_gpu_unet.py)"
            out["fp8"] = fp8_leg
        if b1_leg is not None:
            call_s, first_s, _ = b1_pipe
            b1_leg.update({"what": "generate_audio.py's call: 1 prompt, 50 DDIM steps, guidance 5.0, 10 s [REF script/inference/generate_audio.py:47-52]",
                           "pipe_call_ms": round(call_s * 1e3, 1), "first_call_ms_incl_capture": round(first_s * 1e3, 1)})
            out["infer_b1"] = b1_leg
        if body_leg is not None:
            out["train_loop_body"] = body_leg
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch, H, W, args.rank)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
