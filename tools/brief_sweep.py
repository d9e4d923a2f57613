#!/usr/bin/env python3
"""k_up against the kernel its plan sends BRIEF launches to, by launch length in wave-tiles per wave of k_up's grid: where the
two cross is CR_BRIEF_HALF_TILES* (cr_context.c).  One process, one box: the same plan built three ways (k_up forced, the other kernel
forced, the default that chooses by length).  usage: brief_sweep.py"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import clownresampler_amd as cr
from bench import device_noise

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
CASES = [(8, 2, (8000, 96000, 8000)), (8, 2, (8000, 64000, 8000)), (8, 2, (8000, 80000, 8000)), (3, 2, (8000, 64000, 8000)), (3, 2, (6000, 96000, 6000))]
print("radius ch rates | tiles per wave | output frames | k_up us | other us | default us (kernel it took)")
for radius, ch, rates in CASES:
    api = cr.load(radius); pre = api.precomputed()
    st0 = api.LowLevel_State(); assert api.LowLevel_Init(st0, ch, *rates)
    R = st0.lowest_level.integer_stretched_kernel_radius
    api.DebugSetVariant(0xFFFF)
    info = api.PlanGetInfo(api.PlanCreate(st0, pre))
    if info.kernel != 3 or not info.brief_below:
        print(radius, ch, rates, "no k_up plan / no brief shape:", info.asdict()); continue
    per_tile = (info.tile_frames // 4) * info.max_blocks * (info.threads // 64)    # output frames of one wave-tile per wave
    for tiles in (0.25, 0.5, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24):
        n_want = int(per_tile * tiles)
        frames = max(1, n_want * rates[0] // rates[1])
        n_out = api.CountOutputFrames(st0, frames)
        nsets = max(3, min(64, int(400e6 // max(1, frames * ch * 2 + n_out * ch * 4))))
        sets = [(device_noise((frames + 2 * R) * ch, -R * ch + k * 977, dev), torch.empty(n_out * ch, dtype=torch.int32, device=dev)) for k in range(nsets)]
        row = []
        for variant in (info.variant if info.variant != 0xFFFF else 27, info.brief_variant, 0xFFFF):
            api.DebugSetVariant(variant)
            plan = api.PlanCreate(st0, pre)
            took = api.PlanGetInfo(plan)

            def launch(k):
                st = cr.LowLevel_State.from_buffer_copy(st0)
                pcm, out = sets[k % nsets]
                api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
            t0 = time.perf_counter(); k = 0
            while time.perf_counter() - t0 < 0.15:
                for _ in range(20):
                    launch(k); k += 1
                torch.cuda.synchronize()
            reps = 200 if n_out < 20e6 else 60
            best = 1e9
            for rnd in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(reps):
                    launch(k)
                e1.record(stream); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / reps * 1e3)
            row.append((best, took.kernel if variant != 0xFFFF or n_out >= took.brief_below else took.brief_kernel))
        api.DebugSetVariant(0xFFFF)
        print("%d %d %s | %5.2f | %9d | %7.1f | %7.1f | %7.1f (%d)" % (radius, ch, rates, tiles, n_out, row[0][0], row[1][0], row[2][0], row[2][1]), flush=True)
        del sets
