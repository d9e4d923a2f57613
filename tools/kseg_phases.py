"""Where k_seg's waves spend their shader cycles (the stamped diagnostic instance, CLOWNRESAMPLER_AMD_SEG_FORM=4): python tools/kseg_phases.py [workload]"""
import os, sys
os.environ["CLOWNRESAMPLER_AMD_SEG_FORM"] = "4"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import numpy as np
import torch
import bench
import clownresampler_amd as cr
import _checkers as ck

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
radius, ch, rates, frames = bench.WORKLOADS[name]
api = cr.load(radius)
pre = api.precomputed()
st0 = api.LowLevel_State()
assert api.LowLevel_Init(st0, ch, *rates)
R = st0.lowest_level.integer_stretched_kernel_radius
plan = api.PlanCreate(st0, pre)
pcm = torch.from_numpy(ck.pad_frames(ck.noise_pcm(frames * ch, 5), ch, R)).cuda()
total = api.CountOutputFrames(st0, frames)
out = torch.empty((total + 64) * ch, dtype=torch.int32, device="cuda")
stamp = torch.zeros(8 * 256, dtype=torch.int64, device="cuda")
api.lib.ClownResamplerAMD_DebugSetStampBuffer.argtypes = [C.c_void_p]
api.lib.ClownResamplerAMD_DebugSetStampBuffer(stamp.data_ptr())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for rep in range(3):
    st = cr.LowLevel_State.from_buffer_copy(st0)
    stamp.zero_()
    torch.cuda.synchronize()
    ev[0].record()
    n, left, ran_out = api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), total + 1, torch.cuda.current_stream().cuda_stream)
    ev[1].record()
    torch.cuda.synchronize()
us = ev[0].elapsed_time(ev[1]) * 1e3
p = stamp.cpu().numpy().reshape(-1, 8).astype(float)
p = p[p[:, 7] > 0]
names = ["tile prologue (requests, wait, conversions)", "wait at the head of a frame (scalar row, LDS)", "frame arithmetic", "copy-out", "counted wait + ring requests", "position advance"]
tot = p[:, :6].sum(axis=1).mean()
print("%s: k_seg launches %d, %.1f us with stamps; wave 0 of %d workgroups: %.0f frames, %.1f tiles, %.0f stamped cycles each" % (name, api.LaunchCount(8), us, len(p), p[:, 6].mean(), p[:, 7].mean(), tot))
for k, nm in enumerate(names):
    print("  %-50s %9.0f cycles  %5.1f %%   %7.1f per frame" % (nm, p[:, k].mean(), 100 * p[:, k].mean() / tot, p[:, k].mean() / p[:, 6].mean()))
