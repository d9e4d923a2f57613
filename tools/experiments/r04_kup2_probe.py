#!/usr/bin/env python3
"""probe: k_up2 (variant 27) against the oracle over random ratios / tiny launches; prints every mismatch with its shape"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _checkers as ck, _product
products = {3: _product.Product(3), 8: _product.Product(8)}
rng = random.Random(4711 + 27)
bad = 0
for draw in range(80):
    radius = rng.choice([3, 8])
    p, o = products[radius], ck.oracle(radius)
    i = rng.randrange(4000, 48000)
    out = int(i * rng.uniform(2.0, 16.0))
    frames = rng.choice([rng.randrange(1, 200), rng.randrange(200, 5000), rng.randrange(5000, 40000)])
    ok, a = p.low_init(2, i, out, i)
    ok2, b = o.low_init(2, i, out, i)
    p.api.DebugSetVariant(27)
    info = p.api.PlanGetInfo(p.api.PlanCreate(a.raw, p.pre))
    padded = ck.pad_frames(ck.noise_pcm(frames * 2, 1000 + draw), 2, int(b.cfg.radius_frames))
    total = ck.count_output_frames(b, frames)
    cut = rng.randrange(1, total) if total > 1 else None
    for part in (0, 1):
        if part == 0 and cut is None:
            continue
        cap = cut if part == 0 else None
        st = (a.pos_int, a.pos_frac)
        before = [p.api.LaunchCount(k) for k in range(8)]
        xa, la, ra = p.low_resample_i32(a, padded, frames, capacity=cap)
        ran = [p.api.LaunchCount(k) - before[k] for k in range(8)]
        xb, lb, rb = o.low_resample_i32(b, padded, frames, capacity=cap)
        if not np.array_equal(xa, xb):
            bad += 1
            d = np.flatnonzero(xa != xb)
            print("MISMATCH draw %d part %d radius %d rates %d->%d inc %d frames %d cut %s state %s n_out %d kernel %d tile %d brief_below %d launches %s: %d differ, first at %d, got %s want %s" %
                  (draw, part, radius, i, out, a.increment, frames, cut, st, xb.size // 2, info.kernel, info.tile_frames, info.brief_below, ran, d.size, d[0], xa[d[0]:d[0]+4], xb[d[0]:d[0]+4]))
        if part == 0:
            padded = padded[(frames - la) * 2:]
            frames = la
    p.api.DebugSetVariant(0xFFFF)
print("mismatches:", bad)
