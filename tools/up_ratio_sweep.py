#!/usr/bin/env python3
"""k_up2 against k_wave2 (the kernel a k_up plan falls back to) by upsampling RATIO, long launches (~40 M output frames): which
ratios k_up should keep.  usage: up_ratio_sweep.py [radius [in:out ...]]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clownresampler_amd as cr
from bench import device_noise

radius = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ch = 2
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
api = cr.load(radius); pre = api.precomputed()
print("ratio | rates | output frames | variant: us (kernel) ...")
RATES = [(24000, 48000), (16000, 48000), (11025, 44100), (8000, 44100), (8000, 48000), (8000, 64000), (8000, 80000), (8000, 96000), (8000, 127999)]
if len(sys.argv) > 2:
    RATES = [tuple(int(x) for x in a.split(":")) for a in sys.argv[2:]]
for rates in [(a, b, a) for a, b in RATES]:
    st0 = api.LowLevel_State(); assert api.LowLevel_Init(st0, ch, *rates)
    R = st0.lowest_level.integer_stretched_kernel_radius
    frames = 40000000 * rates[0] // rates[1]
    n_out = api.CountOutputFrames(st0, frames)
    sets = [(device_noise((frames + 2 * R) * ch, -R * ch + k * 977, dev), torch.empty(n_out * ch, dtype=torch.int32, device=dev)) for k in range(3)]
    row = []
    for variant in (27, 30, 13, 28, 0xFFFF):
        api.DebugSetVariant(variant)
        plan = api.PlanCreate(st0, pre)
        took = api.PlanGetInfo(plan)

        def launch(k):
            st = cr.LowLevel_State.from_buffer_copy(st0)
            pcm, out = sets[k % 3]
            api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
        for k in range(20):
            launch(k)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(40):
                launch(k)
            e1.record(stream); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
        row.append("%d: %.1f (%d)" % (variant, best, took.kernel))
    api.DebugSetVariant(0xFFFF)
    print("%.2f | %s | %d | %s" % (rates[1] / rates[0], rates, n_out, "  ".join(row)), flush=True)
    del sets
