#!/usr/bin/env python3
"""k_seg against the kernel the plan otherwise takes (k_up2 / k_wave2), by upsampling RATIO, stereo 8 lobes, ~40 M output frames per launch:
which ratios k_seg should keep (cr_context.c CR_SEG_MIN_INCREMENT / the instance's ring).  usage: seg_ratio_sweep.py [in:out ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clownresampler_amd as cr
from bench import device_noise

radius, ch = 8, 2
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
api = cr.load(radius); pre = api.precomputed()
RATES = [(8000, 36000), (8000, 40000), (8000, 44100), (8000, 48000), (8000, 56000), (8000, 64000), (8000, 72000), (8000, 80000), (8000, 88000), (8000, 96000), (8000, 104000),
         (8000, 112000), (8000, 127999)]
if len(sys.argv) > 1:
    RATES = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
print("ratio | rates | increment | output frames | without k_seg: us (kernel) | k_seg forced: us | by the rule: us (k_seg launches)")
for rates in [(a, b, a) for a, b in RATES]:
    st0 = api.LowLevel_State(); assert api.LowLevel_Init(st0, ch, *rates)
    R = st0.lowest_level.integer_stretched_kernel_radius
    frames = 40000000 * rates[0] // rates[1]
    n_out = api.CountOutputFrames(st0, frames)
    sets = [(device_noise((frames + 2 * R) * ch, -R * ch + k * 977, dev), torch.empty(n_out * ch, dtype=torch.int32, device=dev)) for k in range(3)]
    plan = api.PlanCreate(st0, pre)
    took = api.PlanGetInfo(plan)
    row = []
    for mode in (2, 1, 0):
        api.DebugSegKernel(mode)

        def launch(k):
            st = cr.LowLevel_State.from_buffer_copy(st0)
            pcm, out = sets[k % 3]
            api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
        before = api.LaunchCount(8)
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
        seg = api.LaunchCount(8) - before
        row.append("%.1f (%s)" % (best, {2: "kernel %d" % took.kernel, 1: "%d k_seg launches of 140" % seg, 0: "%d k_seg" % seg}[mode]))
    api.DebugSegKernel(0)
    print("%.2f | %s | %d | %d | %s" % (rates[1] / rates[0], rates, st0.increment, n_out, " | ".join(row)), flush=True)
    del sets
