#!/usr/bin/env python3
"""Kernel time and fraction of the HBM roofline for every channel count 1..16, up- and downsampling (device-resident,
synthetic noise, ~10 M input samples per launch, HIP events around 60 launches after ~200 ms of warm-up launches)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clownresampler_amd as cr
from bench import device_noise

# usage: channel_table.py [radius] [in:out ...] [channels=1,2,...] [samples=N]   (default: four ratios, channels 1..16; N input samples
# per launch instead of 52.9 M - 26.5 M from 2:1 downwards, a fifth above 3x upsampling)
radius = int(sys.argv[1]) if len(sys.argv) > 1 else 3
RATES = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:] if ":" in a]
RATES = [(i, o, min(i, o)) for i, o in RATES] or [(44100, 48000, 44100), (48000, 44100, 44100), (44100, 8000, 8000), (8000, 44100, 8000)]
CHANNELS = [int(x) for a in sys.argv[2:] if a.startswith("channels=") for x in a[9:].split(",")] or list(range(1, 17))
SAMPLES = [int(a[8:]) for a in sys.argv[2:] if a.startswith("samples=")]
api = cr.load(radius); dev = torch.device("cuda", 0); pre = api.precomputed()
stream = torch.cuda.current_stream(dev)
print("radius %d" % radius)
print("ch | rates           | kernel slots tile | us/launch | Msamples/s | GB/s | frac of 8 TB/s")
for rates in RATES:
    for ch in CHANNELS:
        frames = 52920000 // ch if rates[0] <= rates[1] * 2 else 26460000 // ch
        if rates[1] > 3 * rates[0]:
            frames //= 5
        if SAMPLES:
            frames = SAMPLES[0] // ch
        st0 = api.LowLevel_State(); assert api.LowLevel_Init(st0, ch, *rates)
        R = st0.lowest_level.integer_stretched_kernel_radius
        n_out = api.CountOutputFrames(st0, frames)
        sets = []
        for k in range(3):
            sets.append((device_noise((frames + 2 * R) * ch, -R * ch + k * 977, dev), torch.empty(n_out * ch, dtype=torch.int32, device=dev)))
        plan = api.PlanCreate(st0, pre)
        info = api.PlanGetInfo(plan)
        kernel = api.PlanKernelAt(plan, 0)   # (the kernel a launch at fraction 0 takes: k_int is chosen per launch)

        def launch(k):
            st = cr.LowLevel_State.from_buffer_copy(st0)
            pcm, out = sets[k % 3]
            api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
        import time
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 0.2:      # sustained clocks: ~200 ms of launches before the timed ones
            for _ in range(10):
                launch(k); k += 1
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(60):
            launch(k)
        e1.record(stream); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 60
        nbytes = frames * ch * 2 + n_out * ch * 4
        print("%2d | %5d -> %5d | %d %3d %5d | %8.1f | %8.0f | %5.0f | %.3f" % (ch, rates[0], rates[1], kernel, info.slots, info.tile_frames, us, n_out * ch / us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
        del sets
