"""Experiment: the CFG batch of 8 as ONE graph vs TWO independent half-batch graphs replayed on two streams.
Every launch below the top UNet level is latency-bound (2-3 us kernel boundary + a first memory round trip of the same size), so two
command streams should overlap where one leaves the CUs idle."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_unet, synth_inputs
from audioldm_with_lora_amd.engine import DenoiseEngine
from audioldm_with_lora_amd.scheduler import DDIMScheduler

H, W, NSTEPS, G = 250, 16, 200, 2.5
unet, _ = build_unet(4)
lat, pe, ne = synth_inputs(4, H, W)


def make(sl, stream):
    e = DenoiseEngine(unet, DDIMScheduler(), sl.stop - sl.start, H, W, NSTEPS, G)
    with torch.cuda.stream(stream):
        e.set_condition(pe[sl], ne[sl])
        e.set_latents(lat[sl])
        e.capture()
    torch.cuda.synchronize()
    return e


def timed(engs, streams, steps=60):
    for _ in range(5):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


s0 = torch.cuda.Stream()
one = make(slice(0, 4), s0)
ms1 = timed([one], [s0])
print(f"one graph, 4 clips (CFG batch 8):           {ms1:.3f} ms/step", flush=True)
x_one = one.x.clone()
nsplit = int(os.environ.get("NSPLIT", "2"))
streams = [torch.cuda.Stream() for _ in range(nsplit)]
per = 4 // nsplit
halves = [make(slice(i * per, (i + 1) * per), s) for i, s in enumerate(streams)]
ms2 = timed(halves, streams)
print(f"{nsplit} graphs x {per} clips on {nsplit} streams:               {ms2:.3f} ms/step  ({ms1 / ms2:.2f}x)", flush=True)
ms_seq = timed(halves, [streams[0]] * nsplit)
print(f"{nsplit} graphs x {per} clips on ONE stream (no overlap): {ms_seq:.3f} ms/step", flush=True)
# same trajectory?  (both ran 65 steps from the same start; 3 timed() calls for the halves -> compare after equal step counts)
one2 = make(slice(0, 4), s0)
h2 = [make(slice(i * per, (i + 1) * per), s) for i, s in enumerate(streams)]
for _ in range(10):
    with torch.cuda.stream(s0):
        one2.step()
    for e, s in zip(h2, streams):
        with torch.cuda.stream(s):
            e.step()
torch.cuda.synchronize()
xs = torch.cat([e.x for e in h2])
print("max |x_split - x_one| after 10 steps:", float((xs - one2.x).abs().max()), " (|x| max", float(one2.x.abs().max()), ")", flush=True)
