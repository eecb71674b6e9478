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


def kernel_profile(engine, fine=False):
    """One eager denoise step with an event pair (on the launch stream) around every launch; returns per-kernel totals."""
    from audioldm_with_lora_amd import ops
    ops.PROFILE = []
    ops.sleep_us(60000)          # queue the whole step behind a 60 ms sleep: event pairs then time kernels, not host gaps
    engine._one_step()
    empty = []                   # the event pair itself costs time on the stream: calibrate it with empty pairs in the same queue
    for _ in range(64):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        e.record()
        empty.append((s, e))
    torch.cuda.synchronize()
    pair_ms = sorted(s.elapsed_time(e) for s, e in empty)[len(empty) // 2]
    rows, ops.PROFILE = ops.PROFILE, None
    agg = {}
    for label, flops, nbytes, s, e in rows:
        if not fine:
            label = label.split("|")[0]
        a = agg.setdefault(label, dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        a["ms"] += max(s.elapsed_time(e) - pair_ms, 1e-4)
        a["flops"] += flops
        a["bytes"] += nbytes
        a["launches"] += 1
    return agg


def traffic_from_profile(label):
    """HBM bytes per launch of the dominant kernel family from the committed PMC summary (tools/pmc_traffic.py over two
    rocprofv3 --pmc passes of this same command; MI355X_MICROARCH.md correction applied).  None when no summary exists."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    m = re.match(r"igemm_(\d+)x(\d+)_r(\d+)(_vt)?", label)
    if not files or not m:
        return None, None
    prof = json.load(open(files[-1]))
    bm, bn, rp, vt = m.group(1), m.group(2), m.group(3), "true" if m.group(4) else "false"
    tot = n = 0
    for k, v in prof.items():       # every ring depth / wave count of the tile family (split-K launches share the kernel)
        km = re.match(r"igemm_pipe_kernel<(\d+), (\d+), \d+, \d+, (\d+), (\w+), (\d+)(?:, \w+)?>", k)
        if km and (km.group(1), km.group(2), km.group(3), km.group(4)) == (bm, bn, rp, vt):
            tot += v["hbm_bytes_per_launch"] * v["launches_sampled"]
            n += v["launches_sampled"]
    return (tot / n if n else None), os.path.basename(files[-1])


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



def bench_decode(batch, height, width, reps=3):
    """VAE latent->mel decode + HiFi-GAN vocoder for the `batch` clips (random-init weights of the real architecture):
    the remainder of AudioLDMPipeline.__call__ after the loop [REF script/inference/generate_audio.py:47-52]."""
    from audioldm_with_lora_amd.vae import AutoencoderKL
    from audioldm_with_lora_amd.vocoder import SpeechT5HifiGan
    torch.manual_seed(99)
    vae, voc = AutoencoderKL().cuda(), SpeechT5HifiGan().cuda()
    z = torch.randn(batch, 8, height, width, device="cuda")
    t = {}
    for name, fn in (("vae_decode_ms", lambda: vae.decode(z).sample), ("vocoder_ms", None)):
        if fn is None:
            mel = vae.decode(z).sample.squeeze(1)
            fn = lambda: voc(mel)
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        t[name] = (time.perf_counter() - t0) / reps * 1e3
    assert torch.isfinite(out).all()
    del vae, voc
    torch.cuda.empty_cache()
    return t


def bench_train(world, rank, steps=8, warmup=3, batch=8, rank_lora=8):
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
    del tr, unet
    torch.cuda.empty_cache()
    return {"metric": "lora_train_clips_per_sec", "value": round(world * batch * steps / dt, 3), "unit": "10.24s-clips/s",
            "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps, "per_gpu_batch": batch, "lora_rank": rank_lora,
            "lora_params": tr_params(rank_lora), "final_loss": round(float(loss), 5),
            "collective": "1 flat fp32 all-reduce/step (RCCL)" if world > 1 else "none"}


def tr_params(r):
    return 112640 * r


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
    ap.add_argument("--no-train", action="store_true", help="skip the short LoRA-training measurement")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel table to stderr")
    ap.add_argument("--fine", action="store_true", help="print the per-kernel-per-shape table to stderr")
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

    decode = None
    if not args.no_train and rank == 0:
        eng.graph = None
        torch.cuda.empty_cache()
        decode = bench_decode(args.batch, H, W)
    train = None
    if not args.no_train:
        eng.graph = None
        torch.cuda.empty_cache()
        train = bench_train(world, rank, rank_lora=(8 if world == 1 else 16))     # config 3 on one GPU, config 4 (rank 16) data-parallel
    if rank == 0:
        print(f"[bench] {args.steps} steps in {dt:.3f} s -> {dt / args.steps * 1e3:.3f} ms/step", file=sys.stderr, flush=True)
        prof = kernel_profile(eng)
        dom_label, dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        total_ms = sum(v["ms"] for v in prof.values())
        if args.fine:
            for k, v in sorted(kernel_profile(eng, fine=True).items(), key=lambda kv: -kv[1]["ms"]):
                print(f"{k:64s} {v['launches']:3d}x {v['ms'] * 1e3 / v['launches']:7.1f} us  {v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:7.1f} TF/s", file=sys.stderr)
        if args.breakdown:
            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                print(f"{k:32s} {v['launches']:4d} launches {v['ms']:8.3f} ms  {v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:8.1f} TF/s "
                      f"{v['bytes'] / v['ms'] / 1e6 if v['ms'] else 0:8.1f} GB/s", file=sys.stderr)
            print(f"eager sum {total_ms:.3f} ms over {sum(v['launches'] for v in prof.values())} launches", file=sys.stderr)
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        traffic, traffic_src = traffic_from_profile(dom_label)
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": (round(traffic) if traffic else None),
                    "traffic_unit": "HBM bytes/launch (PMC, profiles/%s)" % traffic_src if traffic else None,
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                    "algorithmic_flops_per_launch": round(dom["flops"] / dom["launches"]),
                    "kernel": dom_label, "launches_per_step": dom["launches"],
                    "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                    "share_of_step": round(dom["ms"] / total_ms, 3)}
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
        }
        if train is not None:
            out["train"] = train
        if decode is not None:
            loop_ms = NSTEPS * dt / args.steps * 1e3
            total_ms = loop_ms + decode["vae_decode_ms"] + decode["vocoder_ms"]
            out["end_to_end"] = {"clips_per_sec": round(world * args.batch / (total_ms * 1e-3), 3), "clip_seconds": 10.0,
                                 "ddim_steps": NSTEPS, "loop_ms": round(loop_ms, 1), "vae_decode_ms": round(decode["vae_decode_ms"], 2),
                                 "vocoder_ms": round(decode["vocoder_ms"], 2),
                                 "note": "loop_ms = 200 x the measured step; decode legs run eagerly (not graph-captured)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch, H, W, args.rank)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
