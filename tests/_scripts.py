"""Scripted sessions of the HIGH-LEVEL streaming API with ClownResampler_HighLevel_Adjust between calls (reference
clownresampler.h:1120-1176 Resample, :1183-1209 Adjust, :1242-1250 ResampleEnd), shared by the CPU test that pins the oracle to
the compiled reference and the GPU test that holds the product to the oracle.

A script is plain data: the first triple, the input, and a list of steps
    ("run", budget)            one ClownResampler_HighLevel_Resample call whose output callback accepts `budget` frames, then stops
    ("end", budget)            the same for ClownResampler_HighLevel_ResampleEnd
    ("adjust", (i, o, lp))     ClownResampler_HighLevel_Adjust - accepted or not
`play(engine, script)` returns everything observable: per step the return value, the frames emitted, the state's scalars."""
import random

import numpy as np

import _checkers as ck

RATES = [8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000]


def _state_of(st):
    return tuple(int(v) for v in st.low.astuple()) + (int(st.max_radius_frames), int(st.lead_needed), int(st.trail_left))


def make_script(seed, radius, max_out_frames=30000):
    """Random session: a first triple whose radius leaves room for SMALLER ones later (:1165), Adjusts to triples that are accepted
    (same, smaller radius), rejected for a larger radius (:1195), and rejected by ClownResampler_LowLevel_Adjust itself (a zero
    rate, a ratio beyond the 16.16 increment: :919, :939, :974); runs stopped by the consumer after 1 ... 4,000 frames."""
    rng = random.Random(seed)
    ch = rng.choice([1, 2, 2, 3, 5, 8, 11, 16])
    i = rng.choice(RATES)
    o = rng.choice(RATES)
    # low-pass below the input rate at Init: a wide first kernel, so that later triples fit under it
    lp = rng.choice([min(i, o), max(1, min(i, o) // 2), max(1, min(i, o) // 3), i])
    first = (i, o, lp)
    frames = rng.choice([rng.randrange(1, 300), rng.randrange(300, 6000), rng.randrange(6000, 30000)])
    pull_chunk = rng.choice([0, 1, 13, 333, 2042, 100000])
    steps = []
    budget_left = max_out_frames
    phase = "run"
    for _ in range(rng.randrange(4, 40)):
        kind = rng.random()
        if kind < 0.45:
            b = rng.choice([1, 2, rng.randrange(1, 50), rng.randrange(50, 1500), rng.randrange(1500, 4000)])
            steps.append((phase, b))
            budget_left -= b
        elif kind < 0.9:
            pick = rng.random()
            if pick < 0.5:      # another ratio / low-pass at random: accepted or too wide, as it comes
                t = (rng.choice(RATES), rng.choice(RATES), rng.choice(RATES + [i // 2 or 1, lp, 4000]))
            elif pick < 0.7:    # narrower kernel than at Init: low-pass at (or above) the input rate
                ni = rng.choice(RATES)
                t = (ni, rng.choice(RATES), rng.choice([ni, 2 * ni]))
            elif pick < 0.8:    # much wider: rejected (:1195)
                ni = rng.choice(RATES)
                t = (ni, rng.choice(RATES), max(1, ni // rng.choice([7, 20, 64])))
            elif pick < 0.9:    # ClownResampler_LowLevel_Adjust itself fails
                t = rng.choice([(0, 48000, 48000), (48000, 0, 48000), (48000, 48000, 0), (1 << 31, 1, 1 << 31), (70000, 1, 70000), (4097, 1, 1)])
            else:               # back to the first triple
                t = first
            steps.append(("adjust", t))
        else:
            phase = "end" if phase == "run" and rng.random() < 0.3 else phase
        if budget_left <= 0:
            break
    # drain: whatever is left comes out in big bites, with one more Adjust on the way
    steps += [("run", 1 << 30), ("adjust", rng.choice([first, (i, o, i)])), ("run", 1 << 30), ("end", 1), ("adjust", (rng.choice(RATES), rng.choice(RATES), rng.choice(RATES))),
              ("end", 3), ("adjust", rng.choice([first, (i, o, i), (o, i, o)])), ("end", 1 << 30), ("end", 1 << 30)]
    return dict(seed=seed, radius=radius, channels=ch, first=first, frames=frames, pull_chunk=pull_chunk, steps=steps)


def usable(script, oracle):
    """the configurations the reference itself handles: Init accepted, a table step, and halos that fit its 0x1000-sample staging
    buffer (beyond that the REFERENCE overruns its own buffer, :1112, :1154)"""
    # (the LOW-level Init answers both questions: the reference's high-level Init of a window wider than its staging buffer already
    # writes beyond it - :1112 zeroes radius * channels samples - and the oracle restates it faithfully)
    ok, low = oracle.low_init(script["channels"], *script["first"])
    if not ok or low.cfg.table_step == 0:
        return False
    if 2 * int(low.cfg.radius_frames) * script["channels"] >= 0x1000 - script["channels"]:
        return False
    # bound the work: the per-frame Python callback is the slow part
    worst_ratio = max(RATES) / min(RATES)
    return script["frames"] * worst_ratio <= 400000


def play(engine, script, early_end=True):
    """early_end=False: an "end" step BEFORE the source has run dry (no ClownResampler_HighLevel_Resample call has returned true yet) is
    played as a "run" step instead.  Flushing while the source still has frames is where a read-ahead window shows: the reference then
    pads behind the frames of ITS 0x1000-sample buffer, the product behind everything it has already pulled
    from the caller - up to its streaming window - so the two agree on such a session only with the reference's own window (0)."""
    ch = script["channels"]
    pcm = ck.noise_pcm(script["frames"] * ch, 7000 + script["seed"])
    ok, st = engine.high_init(ch, *script["first"])
    assert ok
    pos = [0]
    trace = [("init", _state_of(st))]

    def pull(n):
        k = min(n, script["frames"] - pos[0])
        if script["pull_chunk"]:
            k = min(k, script["pull_chunk"])
        a = pcm[pos[0] * ch:(pos[0] + k) * ch]
        pos[0] += k
        return a

    dry = False
    for kind, arg in script["steps"]:
        if kind == "end" and not dry and not early_end:
            kind = "run"
        if kind == "adjust":
            r = engine.high_adjust(st, *arg)
            trace.append(("adjust", arg, int(bool(r)), _state_of(st)))
            continue
        out = []
        budget = [arg]

        def emit(frame):
            out.append(tuple(frame))
            budget[0] -= 1
            return budget[0] > 0

        r = engine.high_resample_cb(st, pull, emit) if kind == "run" else engine.high_end_cb(st, emit)
        dry = dry or (kind == "run" and bool(r))
        # (which input frame the source stands at is NOT compared: the product reads ahead)
        trace.append((kind, arg, int(bool(r)), len(out), hash(tuple(out)), out[:2], out[-2:], _state_of(st)))
    return trace


def first_difference(a, b):
    for k, (x, y) in enumerate(zip(a, b)):
        if x != y:
            return k, x, y
    return (len(a), None, None) if len(a) != len(b) else None
