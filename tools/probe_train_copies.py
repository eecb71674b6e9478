"""Which Python lines of the training step launch torch's own copy / fill kernels?  (one eager _fwd_bwd under torch.profiler)"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from audioldm_with_lora_amd.training import LoraTrainer
from audioldm_with_lora_amd.scheduler import DDIMScheduler

unet, _ = bench.build_unet(8)
tr = LoraTrainer(unet, DDIMScheduler(), lr=1e-5, weight_decay=1e-5, max_train_steps=97000, use_graph=False)
g = torch.Generator().manual_seed(0)
lat, noise = torch.randn(8, 8, 256, 16, generator=g).cuda(), torch.randn(8, 8, 256, 16, generator=g).cuda()
t = torch.randint(0, 1000, (8,), generator=g).cuda()
emb = torch.nn.functional.normalize(torch.randn(8, 512, generator=g), dim=-1).cuda()
for _ in range(2):
    tr.step(lat, noise, t, emb)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.step(lat, noise, t, emb)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::zeros", "aten::cat", "aten::add", "aten::mul", "aten::to", "aten::_to_copy"):
        st = [s for s in (e.stack or []) if "audioldm_with_lora_amd" in s or "bench.py" in s]
        cnt[(e.name, st[0] if st else "?")] += 1
for (name, where), c in cnt.most_common(40):
    print(f"{c:5d}  {name:18s} {where}")
