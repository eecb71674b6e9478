"""Kernels of one replayed LoRA training step (config 3: batch 8, rank 8), aggregated by symbol: launches, average and total us per step.
Only COMPLETE replays are counted (bench.trace_steps cuts the device-event stream at the step's last launch, the flat AdamW, and keeps
the steps whose kernel sequences are identical): every per-step launch count is an integer by construction.
usage: python tools/train_table.py [rank]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audioldm_with_lora_amd.scheduler import DDIMScheduler
from audioldm_with_lora_amd.training import LoraTrainer

rank = int(sys.argv[1]) if len(sys.argv) > 1 else 8
unet, _ = bench.build_unet(rank)
tr = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=97000)
g = torch.Generator().manual_seed(0)
lat, noise = torch.randn(8, 8, 256, 16, generator=g).cuda(), torch.randn(8, 8, 256, 16, generator=g).cuda()
t = torch.randint(0, 1000, (8,), generator=g).cuda()
emb = torch.nn.functional.normalize(torch.randn(8, 512, generator=g), dim=-1).cuda()
for _ in range(5):
    tr.step(lat, noise, t, emb)
torch.cuda.synchronize()
per, span, kept = bench.trace_steps(lambda: tr.step(lat, noise, t, emb), "adamw_flat", reps=6)
assert per, "no complete step in the trace"
agg = collections.defaultdict(lambda: [0, 0.0])
for name, us in per:
    agg[name][0] += 1
    agg[name][1] += us
tot = sum(v[1] for v in agg.values())
print(f"{len(per)} kernels per step, {tot:.0f} us of kernel time per step, span {span:.0f} us ({kept} complete steps averaged)")
for fam, ms in bench.train_families(per).items():
    print(f"  family {fam:18s} {ms:7.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{v[1]:8.1f} us {v[0]:4d}x {v[1] / v[0]:7.1f}  {k[:110]}")
