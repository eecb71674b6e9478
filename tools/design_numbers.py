"""Rewrites the two tables of DESIGN.md section 5.1 (between the `<!-- 5.1 tables -->` markers) from a bench.py JSON line.
usage: python tools/design_numbers.py gpurun_out/<tag>_bench.json"""
import json, re, sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r, fl, e2e, tr = j["roofline"], j["floor"], j["end_to_end"], j["train"]
cpu = j.get("cpu_baseline") or {}
gflop = 809.6
tf = gflop / j["ms_per_step"]
rows = [
    ("denoise step (config 2: batch 4 x 10 s, CFG -> UNet batch 8, rank-4 LoRA, bf16)",
     f"**{j['ms_per_step']:.3f} ms = {j['value']:.1f} steps/s** ({gflop} GFLOP per step = {tf:.1f} TFLOP/s = {tf / 2500:.3f} of the dense bf16 MFMA peak)", "`value`, `ms_per_step`"),
    ("launches per step / kernel-time sum / graph span", f"{r['launches_per_step_total']} / {r['kernel_time_sum_ms']:.3f} ms / {r['graph_span_ms']:.3f} ms",
     "`roofline.launches_per_step_total`, `kernel_time_sum_ms`, `graph_span_ms`"),
    (f"dominant symbol `{r['kernel']}`", f"{r['launches_per_step']} launches, {r['avg_launch_us']:.2f} us average, {r['achieved']:.2f} TFLOP/s = **{r['frac']:.4f}** of the MFMA peak; "
     f"share of the step {r['share_of_step']:.3f}", "`roofline`"),
    (f"structure floor ({fl['launches']} boundaries x {fl['boundary_us_per_launch']:.2f} us + the launches' own rooflines)",
     f"{fl['structure_floor_ms']:.3f} ms; step / floor = {fl['step_over_structure_floor']:.2f}", "`floor`"),
    (f"CPU port (the oracle, fp32, {cpu.get('cores', '?')} host cores, 3 timed steps)",
     f"{cpu.get('value', 0):.2f} steps/s -> GPU / CPU = {j['value'] / cpu['value']:.0f}x (a reported baseline, not the target)" if cpu else "not run", "`cpu_baseline`"),
    ("generate_audio.py's call (1 prompt, 50 steps, guidance 5)", f"{j['infer_b1']['ms_per_step']:.4f} ms per step at UNet batch 2", "`infer_b1`"),
    ("config 5 per GPU (e4m3 attention operands)", f"{j['fp8']['ms_per_step']:.4f} ms per step", "`fp8`"),
    ("end to end, 4 x 10 s clips, 200 steps", f"{e2e['pipe_call_ms']:.1f} ms per call; VAE decode {e2e['vae_decode_ms']:.2f} ms ({e2e['vae_frac']:.4f} of peak), "
     f"vocoder {e2e['vocoder_ms']:.2f} ms ({e2e['vocoder_frac']:.4f}; 9.2 ms in round 3)", "`end_to_end`"),
    ("LoRA training step (config 3: 8 x 10.24 s latents, rank 8)", f"**{tr['ms_per_step']:.2f} ms = {tr['value']:.0f} clips/s**; {tr['launches_per_step_total']} kernels, "
     f"kernel-time sum {tr['kernel_time_sum_ms']:.3f} ms; rank 16: {j['train_rank16']['ms_per_step']:.2f} ms", "`train`, `train_rank16`"),
    ("the reference's whole loop body as one graph (mel -> VAE encode -> CLAP -> step)", f"{j['train_loop_body']['ms_per_step']:.2f} ms per 8 clips", "`train_loop_body`"),
]
out = ["| quantity | value | where |", "|---|---|---|"] + [f"| {a} | {b} | {c} |" for a, b, c in rows]
out += ["", "**K1 -- the fused-LoRA attention module** (QKV GEMM + LoRA -> flash attention -> out-projection + LoRA + residual; SURVEY 8d work",
        "`b(8NC^2 + 4N^2C + 16NCr)`, bytes `2bNCs + 4C^2s + 8Crs`; UNet batch 8, rank 4; durations from the replayed graph, `bench.py` `k1`):", "",
        "| site (C, N) | parts (us) | module | rate | fraction of the bound |", "|---|---|---|---|---|"]
for s in j["k1"]["sites"]:
    m = re.match(r"attn C(\d+) N(\d+)", s["site"])
    parts = " + ".join(f"{k} {v}" for k, v in s["parts_us"].items())
    rate = f"{s['tflops']} TF/s" + (f", {s['gb_per_s']} GB/s" if s["bound"] == "hbm" else "")
    out.append(f"| ({m.group(1)}, {m.group(2)}) | {parts} | {s['us_per_module']} us | {rate} | **{s['frac']}** ({'MFMA' if s['bound'] == 'mfma' else 'HBM'}) |")
fams = tr.get("families_ms")
text = open("DESIGN.md").read()
a, b = text.index("<!-- 5.1 tables -->"), text.index("<!-- /5.1 tables -->")
text = text[:a] + "<!-- 5.1 tables -->\n" + "\n".join(out) + "\n" + text[b:]
text = re.sub(r"### 5\.1 Headline \(one MI355X, `bench.py --steps 200`, round 4: `[^`]*`", f"### 5.1 Headline (one MI355X, `bench.py --steps 200`, round 4: `{sys.argv[1]}`", text)
open("DESIGN.md", "w").write(text)
print("\n".join(out))
if fams:
    print("train families:", fams)
