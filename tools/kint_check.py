#!/usr/bin/env python3
"""k_int (whole-number downsampling ratios) against the oracle through the C ABI: every instance, tile tails, capacity stops,
chunked resume, int16 output; asserts through the launch counters that k_int is what ran.  GPU box only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _checkers as ck
import _product

p = _product.Product(3)
o = ck.oracle(3)
bad = 0
for ch in (2, 1, 4, 6, 8):
    for rates in ((48000, 8000, 8000), (48000, 12000, 12000), (96000, 32000, 32000), (96000, 48000, 48000)):
        ok, probe = p.low_init(ch, *rates)
        if p.api.PlanKernelAt(p.api.PlanCreate(probe.raw, p.pre), 0) != 5:
            continue   # (no k_int instance for this channel count at this ratio)
        for frames in (1, 7, 383, 2304, 2305, 10000, 300001):
            ok, st = p.low_init(ch, *rates)
            ok, ost = o.low_init(ch, *rates)
            R = int(ost.cfg.radius_frames)
            padded = ck.pad_frames(ck.noise_pcm(frames * ch, seed=frames), ch, R)
            before = p.api.LaunchCount(5)
            got, left, ran = p.low_resample_i32(st, padded, frames)
            want, oleft, oran = o.low_resample_i32(ost, padded, frames)
            same = np.array_equal(got, want) and st.astuple() == ost.astuple()
            took = p.api.LaunchCount(5) - before
            if not same or took < 1:
                bad += 1
                d = np.flatnonzero(got != want) if got.size == want.size else []
                print("MISMATCH" if not same else "NOT k_int", ch, rates, frames, got.size, want.size, "first diffs", list(d[:5]), "k_int launches", took)
        # chunked: feed the stream in pieces (the state carries the overshoot; the fraction stays 0)
        frames = 50000
        pcm = ck.noise_pcm(frames * ch, seed=11)
        ok, st = p.low_init(ch, *rates)
        ok, ost = o.low_init(ch, *rates)
        R = int(ost.cfg.radius_frames)
        padded = ck.pad_frames(pcm, ch, R)
        a, b = [], []
        at = 0
        for piece in (1234, 7, 20000, 28759):
            view = padded[at * ch:(at + piece + 2 * R) * ch]
            g, _, _ = p.low_resample_i32(st, view, piece)
            w, _, _ = o.low_resample_i32(ost, view, piece)
            a.append(g)
            b.append(w)
            at += piece
        if not np.array_equal(np.concatenate(a), np.concatenate(b)) or st.astuple() != ost.astuple():
            bad += 1
            print("CHUNKED MISMATCH", ch, rates)
    print("ch", ch, "done; k_int launches so far", p.api.LaunchCount(5), "others", [p.api.LaunchCount(k) for k in range(5)])
print("kint_check:", "FAIL %d" % bad if bad else "ok")
sys.exit(1 if bad else 0)
