"""Does torch.profiler (kineto / roctracer) see kernels launched through ctypes, including graph replays?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from audioldm_with_lora_amd import ops

x = torch.randn(8, 4000, 1, 256, device="cuda").to(torch.bfloat16)
w = ops.pack_linear(torch.randn(256, 256, device="cuda") * 0.05, None)
for _ in range(3):
    y = ops.conv(x.view(8, 1, 4000, 256), w)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(4):
            y = ops.conv(x.view(8, 1, 4000, 256), w)
torch.cuda.synchronize()
t0 = time.time()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        y = ops.conv(x.view(8, 1, 4000, 256), w)
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
print("profile wall", time.time() - t0)
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
print(len(evs), "device events")
for e in evs[:20]:
    print(e.name[:80], e.time_range.start, e.device_time if hasattr(e, "device_time") else e.cuda_time)
