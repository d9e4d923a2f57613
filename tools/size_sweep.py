#!/usr/bin/env python3
"""Kernel time against stream length (stereo 44.1 -> 48 kHz, device-resident): where the persistent grid and its tile size
stop being efficient.  usage: size_sweep.py [channels]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clownresampler_amd as cr
from bench import device_noise

ch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rates = (44100, 48000, 44100)
api = cr.load(3); dev = torch.device("cuda", 0); pre = api.precomputed()
stream = torch.cuda.current_stream(dev)
print("seconds of audio | input frames | tile | blocks | us/launch | GB/s | frac of 8 TB/s")
for seconds in (0.01, 0.1, 0.5, 1, 2, 5, 10, 30, 60, 120, 300, 600):
    frames = int(44100 * seconds)
    st0 = api.LowLevel_State(); assert api.LowLevel_Init(st0, ch, *rates)
    R = st0.lowest_level.integer_stretched_kernel_radius
    n_out = api.CountOutputFrames(st0, frames)
    nsets = max(3, min(64, int(400e6 // max(1, frames * ch * 6))))       # rotate through > 256 MB where the stream is short
    sets = [(device_noise((frames + 2 * R) * ch, -R * ch + k * 977, dev), torch.empty(n_out * ch, dtype=torch.int32, device=dev)) for k in range(nsets)]
    plan = api.PlanCreate(st0, pre)
    info = api.PlanGetInfo(plan)

    def launch(k):
        st = cr.LowLevel_State.from_buffer_copy(st0)
        pcm, out = sets[k % nsets]
        api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 0.2:
        for _ in range(20):
            launch(k); k += 1
        torch.cuda.synchronize()
    reps = 200 if seconds < 30 else 60
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for k in range(reps):
        launch(k)
    e1.record(stream); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    nbytes = frames * ch * 2 + n_out * ch * 4
    tiles = (n_out + info.tile_frames - 1) // info.tile_frames
    print("%8.2f | %10d | %5d | %5d | %8.2f | %6.0f | %.3f" % (seconds, frames, info.tile_frames, min(tiles, info.max_blocks), us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000))
    del sets
