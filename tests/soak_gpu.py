#!/usr/bin/env python3
"""Randomised soak of the product against the oracle on a GPU box (test infrastructure; `tests/test_gpu_parity.py::test_soak_short`
runs a fixed-seed minute of it, `python tests/soak_gpu.py --seconds 600` as long as one likes).

Every trial draws a configuration the fixed case list of tests/_cases.py cannot cover by enumeration - any radius built, 1 ... 16
channels, common and arbitrary rate triples (refused ones included: both sides must refuse), lengths from 0 frames to a few million
(the host's kernel choice depends on the LENGTH of a launch: k_seg, dual mono, the brief-launch kernels and the small-call path all
have thresholds), every input kind - and plays it through one of the entry points: one bulk call, chunks with a carried state, a
consumer that stops, the high-level streaming API with an arbitrary pull size, the clamped int16 output.  Bit-exact or it is a
failure, printed as a JSON case that `run_case` replays."""
import argparse
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (HERE, os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

import _checkers  # noqa: E402
from _cases import run_case  # noqa: E402
from _product import Product  # noqa: E402

COMMON = [8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000]
KERNELS = ["k_generic", "k_poly", "k_wave", "k_up2", "k_wave2", "k_int", "k_wave2s", "(of which ticketed)", "k_seg"]


FOCUS = [(8000, 96000), (8000, 64000), (8000, 48000), (11025, 96000), (8000, 44100), (44100, 48000), (48000, 44100), (8000, 88200),
         (16000, 96000), (8000, 36000), (6000, 96000), (8000, 128000), (44100, 8000), (48000, 8000), (48000, 24000), (48000, 32000)]


def draw_focus_case(rng, radii=(3, 5, 8)):
    """Long mono / stereo launches at the ratios whose kernels only LONG launches reach: k_seg (its last block of 64 segments may waste
    6 % at most: lengths just under a whole number of blocks), k_up2, dual mono, the periodic k_int shapes."""
    radius = rng.choice([r for r in (8, 8, 8, 3) if r in radii] or list(radii))
    ch = rng.choice([1, 2, 2])
    i, o = rng.choice(FOCUS)
    increment = (i << 16) // o
    out_frames = int(10 ** rng.uniform(5.8, 7.1))
    if rng.random() < 0.6 and 4096 <= increment < 65536:
        import math
        seg = max(65536 // math.gcd(increment, 65536), 1024)
        blocks = rng.choice([1, 1, 2, 3]) if seg >= 16384 else rng.randint(2, 200)
        out_frames = int(blocks * 64 * seg * rng.uniform(0.945, 1.0))
    frames = min(max(1000, out_frames * increment >> 16), 6000000)
    est = frames * max(o / i, 1e-3) * ch * (2 * radius * max(i / o, 1.0) + 1)
    while est > 2.5e9:
        frames //= 2
        est /= 2
    mode = rng.choice(["device", "device", "device", "device2", "low", "chunked", "s16"])
    case = {"channels": ch, "rates": [i, o, min(i, o)], "radius": radius, "frames": frames, "input": rng.choice(["noise", "noise", "square", "min"]),
            "seed": rng.randint(1, 1 << 30), "mode": "low", "focus": True}
    if mode in ("device", "device2"):
        # ONE ClownResamplerAMD_ResampleDevice call over the whole device-resident stream (the host-pointer entry points cut a call into
        # 4 Mi-frame batches, which never reach k_seg / k_up2 / dual mono); "device2": stopped by the output capacity, then continued
        case["device"] = 2 if mode == "device2" else 1
        case["split"] = rng.uniform(0.05, 0.95)
    elif mode == "chunked":
        case["mode"] = "chunked"
        case["chunks"] = [max(frames // rng.randint(2, 6), 1), max(frames // rng.randint(3, 40), 1)]
    elif mode == "s16":
        case["s16"] = True
    return case


def draw_case(rng, big_budget, radii=(3, 5, 8)):
    if big_budget and rng.random() < 0.05:
        return draw_focus_case(rng, radii)
    if rng.random() < 0.03:
        # variable rate on the device: a list of constant-rate segments over ONE device-resident timeline
        # (ClownResamplerAMD_ResampleSegmentsDevice = Adjust + a low-level call per chunk in the reference, clownresampler.h:1052-1056)
        radius = rng.choice(list(radii))
        segs = []
        for _ in range(rng.randint(1, 40)):
            n = rng.choice([0, 1, 2, rng.randint(1, 200), int(10 ** rng.uniform(2, 4.6))])
            if rng.random() < 0.6:
                i, o = rng.choice(COMMON), rng.choice(COMMON)
            else:
                i, o = rng.randint(4000, 200000), rng.randint(4000, 200000)
            segs.append([n, i, o, min(i, o) if rng.random() < 0.7 else max(1, int(min(i, o) * rng.uniform(0.4, 1.0)))])
        return {"mode": "segments", "radius": radius, "channels": rng.choice([1, 2, 2, 3, 5, 8, 12]), "segments": segs, "s16": rng.random() < 0.25,
                "segments_mode": rng.choice([0, 0, 1, 2]), "rates": segs[0][1:], "frames": sum(x[0] for x in segs), "input": "noise", "seed": rng.randint(1, 1 << 30)}
    if rng.random() < 0.03:
        # ONE stream over several shards (devices) in one C call: ClownResamplerAMD_ResampleShardedDevice
        i, o = (rng.choice(COMMON), rng.choice(COMMON)) if rng.random() < 0.7 else (rng.randint(4000, 200000), rng.randint(4000, 200000))
        return {"mode": "sharded", "radius": rng.choice(list(radii)), "channels": rng.choice([1, 2, 2, 3, 6, 8, 11]), "rates": [i, o, min(i, o)],
                "frames": rng.choice([0, 1, rng.randint(2, 300), int(10 ** rng.uniform(2.5, 5.5))]), "shards": rng.randint(1, 12), "gather": rng.choice([0, 1, 1]),
                "s16": rng.random() < 0.25, "input": "noise", "seed": rng.randint(1, 1 << 30)}
    if rng.random() < 0.04:
        # a scripted high-level session with Adjusts between the calls (tests/_scripts.py), at one of the streaming windows
        return {"mode": "script", "radius": rng.choice(list(radii)), "seed": rng.randint(1, 1 << 30), "window": rng.choice([0, 0, 5000, 1 << 18]),
                "channels": 0, "rates": [0, 0, 0], "frames": 0, "input": "noise"}
    radius = rng.choice([r for r in (3, 3, 5, 8, 8) if r in radii])
    ch = rng.choice([1, 1, 2, 2, 2, 2, 3, 4, 5, 6, 7, 8, rng.randint(9, 16)])
    kind = rng.random()
    if kind < 0.55:
        i, o = rng.choice(COMMON), rng.choice(COMMON)
    elif kind < 0.8:
        i, o = rng.randint(4000, 200000), rng.randint(4000, 200000)
    else:   # small integers: exact periodic ratios (k_int), extreme ones, refused ones
        i, o = rng.randint(1, 64), rng.randint(1, 64)
    lp = min(i, o) if rng.random() < 0.7 else max(1, int(min(i, o) * rng.uniform(0.3, 1.0)))
    if rng.random() < 0.03:
        lp = rng.choice([0, 1, max(i, o) * 2])
    size = rng.random()
    if size < 0.15:
        frames = rng.randint(0, 300)
    elif size < 0.75:
        frames = int(10 ** rng.uniform(2.5, 5.3))
    elif size < 0.93 or not big_budget:
        frames = int(10 ** rng.uniform(5.3, 6.2))
    else:
        frames = int(10 ** rng.uniform(6.2, 6.9))
    # keep the oracle's share of a trial bounded: output samples x taps
    est = frames * max(o / max(i, 1), 1e-3) * ch * (2 * radius * max(i / max(o, 1), 1.0) + 1)
    while est > 6e8 and frames > 1000:
        frames //= 2
        est /= 2
    mode = rng.choice(["low", "low", "low", "chunked", "earlystop", "high", "s16", "callback"])
    case = {"channels": ch, "rates": [i, o, lp], "radius": radius, "frames": frames,
            "input": rng.choice(["noise", "noise", "noise", "noise", "square", "min", "max", "ramp", "impulse"]), "seed": rng.randint(1, 1 << 30)}
    if mode == "chunked":
        case["mode"] = "chunked"
        case["chunks"] = [max(1, int(10 ** rng.uniform(0, 5))) for _ in range(rng.randint(1, 5))]
        if frames / min(case["chunks"]) > 3000:
            case["chunks"] = [max(c, frames // 2000 + 1) for c in case["chunks"]]
    elif mode == "earlystop":
        case["mode"] = "earlystop"
        case["stop_every"] = max(1, int(10 ** rng.uniform(0, 5)))
        out_frames = frames * o / max(i, 1)
        if out_frames / case["stop_every"] > 2000:
            case["stop_every"] = int(out_frames // 1500 + 1)
    elif mode == "high":
        case["mode"] = "high"
        case["pull_chunk"] = rng.choice([0, 0, 1, 7, 256, 4096, 100000])
        if frames > 400000:
            case["frames"] = frames = rng.randint(1000, 400000)   # (the high-level API delivers frame by frame through a callback)
    elif mode == "s16":
        case["mode"] = "low"
        case["s16"] = True
    elif mode == "callback":
        # ClownResampler_LowLevel_Resample, the reference's callback signature, the consumer stopping after `budgets` frames in turn and
        # the caller resuming behind the frames that were consumed (clownresampler.h:1058-1092; examples/low-level.c:87-102)
        case["mode"] = "callback"
        out_frames = frames * o / max(i, 1)
        if out_frames > 60000:
            case["frames"] = frames = max(1, int(frames * 60000 / out_frames))
        case["budgets"] = [max(1, int(10 ** rng.uniform(0, 4.5))) for _ in range(rng.randint(1, 6))]
    else:
        case["mode"] = "low"
    return case


def run_trial(products, oracles, case):
    """None when product and oracle agree (or both refuse), else a description of the difference."""
    radius, ch, rates = case["radius"], case["channels"], case["rates"]
    prod, orc = products[radius], oracles[radius]
    if case["mode"] == "sharded":
        from _cases import make_input
        from _checkers import pad_frames
        api = prod.api
        ok_o, so = orc.low_init(ch, *rates)
        ok_p, sp = prod.low_init(ch, *rates)
        if bool(ok_o) != bool(ok_p):
            return "Init: oracle %s, product %s" % (ok_o, ok_p)
        if not ok_o:
            return None
        pcm = make_input(case)
        frames = len(pcm) // ch
        R = int(so.cfg.radius_frames)
        padded = pad_frames(pcm, ch, R)
        want, _, _ = orc.low_resample_i32(so, padded, frames)
        total = want.size // ch
        s16 = bool(case.get("s16"))
        unit = 2 if s16 else 4
        if s16:
            want = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
        ndev = max(1, api.DeviceCount())
        nshards = case["shards"]
        allocated, shards, places = [], [], []
        try:
            for r in range(nshards):
                sh = api.PlanShard(sp.raw, frames, r, nshards)
                dev = (r * 7 + case["seed"]) % ndev
                lo, hi = sh.first_input_frame, sh.first_input_frame + sh.input_frames + 2 * R
                piece = np.ascontiguousarray(padded[lo * ch: hi * ch]) if sh.output_frames else np.zeros(8, dtype=np.int16)
                d_in = api.DeviceAllocOn(dev, piece.nbytes + 64)
                d_out = api.DeviceAllocOn(dev, max(1, sh.output_frames) * ch * unit + 64)
                allocated += [d_in, d_out]
                api.CopyToDevice(d_in, piece)
                shards.append((dev, d_in, d_out, None))
                places.append((sh.first_output_frame, sh.output_frames, d_out))
            root_shard = case["seed"] % nshards
            root = api.DeviceAllocOn(shards[root_shard][0], max(1, total) * ch * unit + 64)
            allocated.append(root)
            n = api.ResampleShardedDevice(sp.raw, prod.pre, frames, shards, s16=s16, gather_mode=case["gather"], root_shard=root_shard, root_output=root if case["gather"] else None)
            if api.ShardedSynchronize(shards) != 0:
                return "sharded: ShardedSynchronize failed"
            if n != total:
                return "sharded: %d frames, expected %d" % (n, total)
            got = np.empty_like(want)
            if case["gather"]:
                if want.size:
                    api.CopyFromDevice(got, root)
            else:
                for first, count, d_out in places:
                    if count:
                        part = np.empty(count * ch, dtype=want.dtype)
                        api.CopyFromDevice(part, d_out)
                        got[first * ch:(first + count) * ch] = part
        finally:
            for a in allocated:
                api.DeviceFree(a)
        if not np.array_equal(got, want):
            d = np.flatnonzero(got != want)
            return "sharded (%d shards): %d samples differ, first at %d" % (nshards, d.size, d[0])
        if tuple(int(v) for v in sp.astuple()) != tuple(int(v) for v in so.astuple()):
            return "sharded: final state differs"
        return None
    if case["mode"] == "segments":
        from _checkers import noise_pcm, pad_frames
        api = prod.api
        segments = [tuple(x) for x in case["segments"]]
        # every segment's triple must be one the reference accepts; the halo is the widest radius among them
        halo = 0
        for n, *r3 in segments:
            ok, st = orc.low_init(ch, *r3)
            if not ok or int(st.cfg.table_step) == 0:
                return None
            halo = max(halo, int(st.cfg.radius_frames))
        frames = case["frames"]
        pcm = noise_pcm(max(frames, 1) * ch, case["seed"])[: frames * ch]
        padded = pad_frames(pcm, ch, halo)
        _, so = orc.low_init(ch, *segments[0][1:])
        pos, outs, want_counts = 0, [], []
        for n, *r3 in segments:
            if not orc.low_adjust(so, *r3):
                return None
            R = int(so.cfg.radius_frames)
            x, left, ran_out = orc.low_resample_i32(so, padded[(pos + halo - R) * ch:], n)
            outs.append(np.array(x, dtype=np.int32))
            want_counts.append(len(x) // ch)
            pos += n
        want = np.concatenate(outs) if outs else np.zeros(0, dtype=np.int32)
        if case.get("s16"):
            want = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
        sp = api.LowLevel_State()
        api.LowLevel_Init(sp, ch, *segments[0][1:])
        d_in = api.DeviceAlloc(padded.nbytes + 64)
        d_out = api.DeviceAlloc(want.nbytes + 64)
        api.DebugSegmentsMode(case.get("segments_mode", 0))
        try:
            api.CopyToDevice(d_in, padded)
            n, counts = api.ResampleSegmentsDevice(sp, prod.pre, d_in + halo * ch * 2, halo, segments, d_out, max(want.size // ch, 1), s16=bool(case.get("s16")))
            api.StreamSynchronize()
            got = np.empty_like(want)
            if want.size:
                api.CopyFromDevice(got, d_out)
        finally:
            api.DebugSegmentsMode(0)
            api.DeviceFree(d_in)
            api.DeviceFree(d_out)
        if n != want.size // ch or list(counts) != want_counts:
            return "segments: %d frames %s, expected %d %s" % (n, list(counts)[:8], want.size // ch, want_counts[:8])
        if not np.array_equal(got, want):
            d = np.flatnonzero(got != want)
            return "segments: %d samples differ, first at %d" % (d.size, d[0])
        if (sp.lowest_level.stretched_kernel_radius, sp.position_integer, sp.position_fractional, sp.increment) != (so.cfg.stretched_radius, so.pos_int, so.pos_frac, so.increment):
            return "segments: final state differs"
        return None
    if case["mode"] == "script":
        import _scripts
        script = _scripts.make_script(case["seed"], radius)
        if not _scripts.usable(script, orc):
            return None
        prod.api.SetStreamingWindow(case["window"])
        try:
            # (a flush while the source still has frames only with the reference's own window: _scripts.play)
            a = _scripts.play(prod, script, early_end=case["window"] == 0)
        finally:
            prod.api.SetStreamingWindow(1 << 18)
        b = _scripts.play(orc, script, early_end=case["window"] == 0)
        d = _scripts.first_difference(a, b)
        return None if d is None else "scripted session differs: %s" % (d,)
    ok_o, _ = orc.low_init(ch, *rates)
    ok_p, _ = prod.low_init(ch, *rates)
    if bool(ok_o) != bool(ok_p):
        return "Init: oracle %s, product %s" % (ok_o, ok_p)
    if not ok_o:
        return None
    if case["mode"] == "high":
        # a window wider than the reference's 0x1000-sample staging buffer: the reference's Init accepts it and then overruns the buffer
        # (undefined: clownresampler.h:1154 with the subtraction negative); the product's Init refuses, by the rule of :1202
        _, so = orc.low_init(ch, *rates)
        if int(so.cfg.radius_frames) * 2 >= 0x1000 // ch:
            ok_h, _ = prod.high_init(ch, *rates)
            return None if not ok_h else "HighLevel_Init accepted a window wider than the staging buffer"
    if case.get("device"):
        from _cases import make_input
        from _checkers import count_output_frames, pad_frames
        api = prod.api
        pcm = make_input(case)
        frames = len(pcm) // ch
        _, so = orc.low_init(ch, *rates)
        _, sp = prod.low_init(ch, *rates)
        padded = pad_frames(pcm, ch, int(so.cfg.radius_frames))
        want = orc.low_resample_i32_mt(so, padded, frames, threads=min(32, os.cpu_count() or 1))
        total = want.size // ch
        d_in = api.DeviceAlloc(padded.nbytes + 64)
        d_out = api.DeviceAlloc(want.nbytes + 64)
        try:
            api.CopyToDevice(d_in, padded)
            plan = api.PlanCreate(sp.raw, prod.pre)
            if case["device"] == 1 or total < 2:
                n, left, ran_out = api.ResampleDevice(plan, sp.raw, d_in, frames, d_out, total + 1)
                api.StreamSynchronize()
                if (n, left, ran_out) != (total, 0, 1):
                    return "ResampleDevice returned %s, expected (%d, 0, 1)" % ((n, left, ran_out), total)
            else:
                # stopped by the capacity after `first` frames (clownresampler.h:1085-1088: the frames not yet consumed come back), then the rest
                first = max(1, min(total - 1, int(total * case["split"])))
                n1, left1, ran1 = api.ResampleDevice(plan, sp.raw, d_in, frames, d_out, first)
                api.StreamSynchronize()
                if n1 != first or ran1 != 0:
                    return "first ResampleDevice call returned %s" % ((n1, left1, ran1),)
                consumed = frames - left1
                n2, left2, ran2 = api.ResampleDevice(plan, sp.raw, d_in + consumed * ch * 2, left1, d_out + first * ch * 4, total - first + 1)
                api.StreamSynchronize()
                if (n1 + n2, left2, ran2) != (total, 0, 1):
                    return "two ResampleDevice calls returned %s + %s, expected %d frames" % ((n1, left1, ran1), (n2, left2, ran2), total)
            got = np.empty_like(want)
            api.CopyFromDevice(got, d_out)
        finally:
            api.DeviceFree(d_in)
            api.DeviceFree(d_out)
        if not np.array_equal(got, want):
            d = np.flatnonzero(got != want)
            return "%d samples differ, first at %d (frame %d): oracle %d, product %d" % (d.size, d[0], d[0] // ch, want[d[0]], got[d[0]])
        end = total * int(so.increment)
        if (int(sp.pos_int), int(sp.pos_frac)) != ((end >> 16) - frames, end & 0xFFFF):
            return "final state %s" % ((int(sp.pos_int), int(sp.pos_frac)),)
        return None
    if case["mode"] == "callback":
        from _cases import make_input
        from _checkers import pad_frames
        pcm = make_input(case)
        frames = len(pcm) // ch
        results = []
        for engine in (orc, prod):
            _, st = engine.low_init(ch, *rates)
            padded = pad_frames(pcm, ch, int(st.cfg.radius_frames))
            out, calls = [], []
            pos, left, k = 0, frames, 0
            while True:
                budget = [case["budgets"][k % len(case["budgets"])]]
                k += 1

                def emit(frame):
                    out.extend(frame)
                    budget[0] -= 1
                    return budget[0] > 0

                before = left
                r, left = engine.low_resample_cb(st, padded[pos * ch:], left, emit)
                pos += before - left
                calls.append((int(bool(r)), int(left)))
                if r or k > 200000:
                    break
            results.append((out, calls, tuple(int(v) for v in st.astuple())))
        if results[0][0] != results[1][0]:
            return "callback API: frames differ (%d against %d samples)" % (len(results[0][0]), len(results[1][0]))
        if results[0][1:] != results[1][1:]:
            return "callback API: return values or state differ: oracle %s, product %s" % (results[0][1:][-1], results[1][1:][-1])
        return None
    if case.get("s16"):
        # the clamped int16 output (examples/low-level.c:69-80 fused in): the oracle's int32 stream clamped the same way
        from _cases import make_input
        from _checkers import pad_frames
        pcm = make_input(case)
        frames = len(pcm) // ch
        _, so = orc.low_init(ch, *rates)
        _, sp = prod.low_init(ch, *rates)
        padded = pad_frames(pcm, ch, int(so.cfg.radius_frames))
        want, left_o, ran_o = orc.low_resample_i32(so, padded, frames)
        got, left_p, ran_p = prod.api.LowLevel_ResampleBulkS16(sp.raw, prod.pre, padded, frames)
        want16 = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
        if got.size != want16.size or not np.array_equal(got, want16):
            return "int16 output differs (%d against %d samples)" % (got.size, want16.size)
        if (left_o, ran_o) != (left_p, ran_p) or so.astuple() != sp.astuple():
            return "int16 call: state or return values differ"
        return None
    a = run_case(orc, case, keep_output=True)
    b = run_case(prod, case, keep_output=True)
    if a["frames"] != b["frames"]:
        return "frames: oracle %d, product %d" % (a["frames"], b["frames"])
    if not np.array_equal(a["_out"], b["_out"]):
        d = np.flatnonzero(a["_out"] != b["_out"])
        return "%d samples differ, first at %d (frame %d): oracle %d, product %d" % (d.size, d[0], d[0] // ch, a["_out"][d[0]], b["_out"][d[0]])
    if a["state"] != b["state"]:
        return "final state: oracle %s, product %s" % (a["state"], b["state"])
    return None


def soak(seconds, seed, big_budget=True, log=print, max_trials=None, trace=None, radii=(3, 5, 8), abi="c89"):
    rng = random.Random(seed)
    products = {r: Product(r, abi) for r in radii}
    oracles = {r: _checkers.oracle(r) for r in radii}
    api = products[radii[0]].api
    start_counts = [api.LaunchCount(k) for k in range(len(KERNELS))]
    t0 = time.time()
    trials = failures = 0
    by_mode = {}
    while time.time() - t0 < seconds and (max_trials is None or trials < max_trials):
        case = draw_case(rng, big_budget, radii)
        trials += 1
        if trace is not None:   # (written BEFORE the trial: a crash of the process leaves its case as the last line)
            trace.write("%d %s\n" % (trials, json.dumps(case)))
            trace.flush()
        key = "device" if case.get("device") else case["mode"] if case["mode"] in ("segments", "script", "sharded") else "s16" if case.get("s16") else case["mode"]
        by_mode[key] = by_mode.get(key, 0) + 1
        try:
            diff = run_trial(products, oracles, case)
        except Exception as e:   # an error report from the library is a finding too
            diff = "exception: %r" % (e,)
        if diff is not None:
            failures += 1
            log("FAIL trial %d: %s\n  case %s" % (trials, diff, json.dumps(case)))
    counts = [api.LaunchCount(k) - s for k, s in enumerate(start_counts)]
    log("soak: seed %d, %d trials in %.0f s, %d failure(s); by entry point %s; launches by kernel %s" % (
        seed, trials, time.time() - t0, failures, by_mode, {KERNELS[k]: c for k, c in enumerate(counts) if c}))
    return trials, failures


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-big", action="store_true", help="no multi-million-frame trials")
    ap.add_argument("--abi", default="c89", help="c99: libclownresampler_amd_c99.so, the CC_USE_C99_INTEGERS build")
    ap.add_argument("--radii", default="3,5,8", help="the radii the library under test was built for")
    ap.add_argument("--trace", default="", help="file that receives every case before it runs")
    ap.add_argument("--replay", default="", help="a JSON case (as printed) to run once instead of the soak")
    args = ap.parse_args()
    _checkers.build_checkers()
    if args.replay:
        case = json.loads(args.replay)
        diff = run_trial({case["radius"]: Product(case["radius"])}, {case["radius"]: _checkers.oracle(case["radius"])}, case)
        print("replay:", diff or "agrees")
        sys.exit(1 if diff else 0)
    trials, failures = soak(args.seconds, args.seed, not args.no_big, trace=open(args.trace, "w") if args.trace else None,
                            radii=tuple(int(r) for r in args.radii.split(",")), abi=args.abi)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
