"""Kernels of one replayed LoRA training step (config 3: batch 8, rank 8), aggregated by symbol: launches, average and total us per step.
usage: python tools/train_table.py [rank]"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import ProfilerActivity, profile
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
reps = 4
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(reps):
        tr.step(lat, noise, t, emb)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
first, last = None, None
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        agg[e.name][0] += 1
        agg[e.name][1] += e.time_range.elapsed_us()
tot = sum(v[1] for v in agg.values()) / reps
n = sum(v[0] for v in agg.values()) / reps
print(f"{n:.0f} device activities per step, {tot:.0f} us of kernel time per step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{v[1] / reps:8.1f} us {v[0] / reps:6.1f}x {v[1] / v[0]:7.1f}  {k[:110]}")
