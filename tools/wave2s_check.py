#!/usr/bin/env python3
"""k_wave2s (a lane per channel pair) against the oracle through the C ABI, forced with variant 32 for every channel count it
accepts (3-16) and a spread of long-window shapes; asserts through the launch counters that it ran.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _checkers as ck
import _product

bad = 0
for radius, rates_list in ((3, [(44100, 8000, 8000), (48000, 11025, 11025), (96000, 44100, 44100)]), (8, [(44100, 48000, 44100), (48000, 44100, 44100), (48000, 19200, 19200)])):
    p = _product.Product(radius)
    o = ck.oracle(radius)
    p.api.DebugSetVariant(32)
    for rates in rates_list:
        for ch in range(3, 17):
            for frames in (1, 700, 30011):
                ok, st = p.low_init(ch, *rates)
                ok, ost = o.low_init(ch, *rates)
                R = int(ost.cfg.radius_frames)
                padded = ck.pad_frames(ck.noise_pcm(frames * ch, seed=frames + ch), ch, R)
                expect = p.api.PlanGetInfo(p.api.PlanCreate(st.raw, p.pre)).kernel == 6   # (specialised shapes keep their instance)
                before = p.api.LaunchCount(6)
                got, left, ran = p.low_resample_i32(st, padded, frames)
                want, oleft, oran = o.low_resample_i32(ost, padded, frames)
                took = p.api.LaunchCount(6) - before if expect else 1
                same = np.array_equal(got, want) and st.astuple() == ost.astuple()
                if not same or took < 1:
                    bad += 1
                    d = np.flatnonzero(got != want) if got.size == want.size else []
                    print("MISMATCH" if not same else "NOT k_wave2s", radius, rates, ch, frames, got.size, want.size, "first diffs", list(d[:6]), "launches", took)
            ok, st = p.low_init(ch, *rates)
            ok, ost = o.low_init(ch, *rates)
            frames = 9000
            padded = ck.pad_frames(ck.noise_pcm(frames * ch, seed=5), ch, int(ost.cfg.radius_frames))
            got16, _, _ = p.api.LowLevel_ResampleBulkS16(st.raw, p.pre, padded, frames)
            want, _, _ = o.low_resample_i32(ost, padded, frames)
            if not np.array_equal(got16, np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)):
                bad += 1
                print("S16 MISMATCH", radius, rates, ch)
        print(radius, rates, "done; k_wave2s launches", p.api.LaunchCount(6))
    p.api.DebugSetVariant(-1)
print("wave2s_check:", "FAIL %d" % bad if bad else "ok")
sys.exit(1 if bad else 0)
