"""One denoise step (config 2) as a table in launch order: label, K1 site, in-graph duration (kineto timestamps of replays).
usage: python tools/step_table.py [batch] [fp8]       (fp8: config 5, e4m3 attention operands)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audioldm_with_lora_amd.engine import DenoiseEngine
from audioldm_with_lora_amd.scheduler import DDIMScheduler

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
unet, _ = bench.build_unet(4)
unet.attention_fp8 = len(sys.argv) > 2 and sys.argv[2] == "fp8"
eng = DenoiseEngine(unet, DDIMScheduler(), B, 250, 16, 200, 2.5)
lat, pe, ne = bench.synth_inputs(B, 250, 16)
eng.set_condition(pe, ne)
eng.set_latents(lat)
eng.capture()
for _ in range(5):
    eng.step()
rows = bench.launch_rows(eng)
kern, span = bench.trace_replays(eng)
j = bench.join_trace(rows, kern) if kern else None
tot = 0.0
for i, r in enumerate(j or rows):
    us = r[5] if j else r[3]
    tot += us
    print(f"{i:4d} {us:7.2f} us  {r[0]:60s} {r[4] or ''}")
print(f"{len(rows)} launches, kernel time {tot:.1f} us, replay span {span if span else 0:.1f} us")
