"""Parity cases shared by the golden generator, the oracle tests (CPU) and the HIP parity tests (GPU).

A case is a plain dict (JSON-able).  `run_case(engine, case)` drives an *engine* - anything with the
`Checker` interface of tests/_checkers.py: the oracle, the compiled reference, or the product adapter of
tests/_product.py - and returns a result dict (frame count, sha256 / stream hash of the int32-LE stream
in the reference harness's on-disk layout tests/test-low-level.c:43-49, head samples, range, final state).
"""
import os

import numpy as np

from _checkers import (NOISE_SEED, count_output_frames, noise_pcm, pad_frames, sha256_i32, stream_hash)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FLAC_PCM = os.path.join(GOLDEN_DIR, "test_flac_s16le_2ch_192000.pcm")


def make_input(case):
    kind = case["input"]
    ch = case["channels"]
    if kind == "flac":
        pcm = np.fromfile(FLAC_PCM, dtype="<i2")
        assert ch == 2
        frames = case.get("frames", len(pcm) // 2)
        return pcm[: frames * 2].astype(np.int16)
    frames = case["frames"]
    n = frames * ch
    if kind == "noise":
        return noise_pcm(n, case.get("seed", NOISE_SEED))
    if kind == "square":  # full-scale, per-channel phase offset; period in frames
        period = case.get("period", 7)
        f = (np.arange(frames)[:, None] + np.arange(ch)[None, :]) // period
        return np.where(f % 2 == 0, 32767, -32768).astype(np.int16).reshape(-1)
    if kind == "min":  # -32768 everywhere: the only way to a per-tap product of INT32_MIN
        return np.full(n, -32768, dtype=np.int16)
    if kind == "max":
        return np.full(n, 32767, dtype=np.int16)
    if kind == "impulse":
        a = np.zeros(n, dtype=np.int16)
        a[(frames // 2) * ch: (frames // 2) * ch + ch] = 32767
        return a
    if kind == "ramp":
        return ((np.arange(n) * 2654435761) >> 7).astype(np.int16)
    raise ValueError(kind)


def summarize(out, ch, extra=None):
    out = np.ascontiguousarray(out, dtype=np.int32)
    res = {
        "frames": int(out.size // ch),
        "sha256": sha256_i32(out),
        "fnv": "%016x" % stream_hash(out),
        "head": [int(v) for v in out[:16]],
        "min": int(out.min()) if out.size else 0,
        "max": int(out.max()) if out.size else 0,
    }
    if extra:
        res.update(extra)
    return res


def run_case(engine, case, keep_output=False):
    """Runs one case; returns summary dict (plus '_out' array when keep_output)."""
    ch = case["channels"]
    rates = case["rates"]
    mode = case.get("mode", "low")
    pcm = make_input(case)
    frames = len(pcm) // ch

    if mode == "high":
        ok, st = engine.high_init(ch, *rates)
        assert ok
        out = engine.high_run_i32(st, pcm, case.get("pull_chunk", 0))
        res = summarize(out, ch, {"state": [int(st.low.pos_int), int(st.low.pos_frac)]})
    else:
        ok, st = engine.low_init(ch, *rates)
        assert ok, case
        R = int(st.cfg.radius_frames)
        padded = pad_frames(pcm, ch, R)
        total = int(count_output_frames(st, frames))
        out = np.empty(max(total, 1) * ch, dtype=np.int32)
        if mode == "low":
            o, left, ran_out = engine.low_resample_i32(st, padded, frames)
            assert ran_out == 1 and left == 0
            out = o
        elif mode == "chunked":
            # input handed over in chunks with carried state; halo = real neighbours (clownresampler.h:725-733)
            pos, w = 0, 0
            chunks = case["chunks"]
            i = 0
            while pos < frames:
                n = min(chunks[i % len(chunks)], frames - pos)
                i += 1
                o, left, ran_out = engine.low_resample_i32(st, padded[pos * ch:], n)
                assert ran_out == 1 and left == 0
                out[w: w + o.size] = o
                w += o.size
                pos += n
            out = out[:w]
        elif mode == "earlystop":
            # consumer stops every `stop_every` frames; caller advances by the consumed frames
            # (examples/low-level.c:87-102 usage pattern)
            every = case["stop_every"]
            pos, w, left = 0, 0, frames
            guard = 0
            while True:
                before = left
                o, left, ran_out = engine.low_resample_i32(st, padded[pos * ch:], left, capacity=every)
                out[w: w + o.size] = o
                w += o.size
                pos += before - left
                guard += 1
                assert guard < 10_000_000
                if ran_out:
                    break
            out = out[:w]
        else:
            raise ValueError(mode)
        res = summarize(out, ch, {"state": [int(st.pos_int), int(st.pos_frac)]})
    if keep_output:
        res["_out"] = np.array(out, dtype=np.int32)
    return res


# ---------------------------------------------------------------------------------------------
# The case list.  Small enough that the oracle runs all of it in well under a minute.
# ---------------------------------------------------------------------------------------------
def _lp(i, o):
    return [i, o, min(i, o)]


CASES = []


def _add(name, **kw):
    kw["name"] = name
    kw.setdefault("radius", 3)
    CASES.append(kw)


# the reference's own fixture through the reference's own ctest triples (tests/CMakeLists.txt:25-46) + cfg 1
_add("flac_cfg1_low", channels=2, rates=[44100, 48000, 44100], input="flac")
_add("flac_cfg1_high", channels=2, rates=[44100, 48000, 44100], input="flac", mode="high")
_add("flac_cfg1_lp48000", channels=2, rates=[44100, 48000, 48000], input="flac")
_add("flac_ctest1_low", channels=2, rates=[8000, 44100, 44100], input="flac")
_add("flac_ctest1_high", channels=2, rates=[8000, 44100, 44100], input="flac", mode="high")
_add("flac_ctest2_low", channels=2, rates=[8000, 44100, 8000], input="flac")
_add("flac_ctest3_low", channels=2, rates=[44100, 8000, 44100], input="flac")
_add("flac_ctest3_high", channels=2, rates=[44100, 8000, 44100], input="flac", mode="high")
_add("flac_ctest4_low", channels=2, rates=[44100, 8000, 8000], input="flac")
_add("flac_ctest4_high", channels=2, rates=[44100, 8000, 8000], input="flac", mode="high")

# BASELINE.json configs at 1-minute length (SURVEY.md 8(d) known answers)
_add("cfg2_1min", channels=2, rates=_lp(44100, 48000), input="noise", frames=2646000)
_add("cfg3_1min", channels=2, rates=_lp(8000, 96000), input="noise", frames=480000, radius=8)
_add("cfg4_1min", channels=8, rates=_lp(48000, 44100), input="noise", frames=2880000)

# resume semantics: chunked input with carried state, early stop every n frames
_add("cfg2_chunked", channels=2, rates=_lp(44100, 48000), input="noise", frames=300000, mode="chunked", chunks=[1, 7, 1000, 3, 123457])
_add("cfg2_earlystop", channels=2, rates=_lp(44100, 48000), input="noise", frames=200000, stop_every=1000, mode="earlystop")
_add("down_chunked", channels=2, rates=_lp(44100, 8000), input="noise", frames=120000, mode="chunked", chunks=[50, 4099, 17])
_add("down_earlystop", channels=1, rates=_lp(48000, 44100), input="noise", frames=60000, stop_every=777, mode="earlystop")
_add("high_small_pulls", channels=2, rates=_lp(44100, 48000), input="noise", frames=50000, mode="high", pull_chunk=97)
_add("high_down", channels=3, rates=_lp(48000, 11025), input="noise", frames=50000, mode="high", pull_chunk=1000)

# channel counts 1..16 (CLOWNRESAMPLER_MAXIMUM_CHANNELS, clownresampler.h:458-460)
for _ch in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
    _add("ch%d_up" % _ch, channels=_ch, rates=_lp(44100, 48000), input="noise", frames=20000, seed=NOISE_SEED + _ch)
    _add("ch%d_down" % _ch, channels=_ch, rates=_lp(48000, 44100), input="noise", frames=20000, seed=NOISE_SEED + 100 + _ch)

# ratio sweep: identity, integer, near-unity, extreme (stretch up to ~46 -> 276 taps), low-pass below both rates
for _i, _o, _l in [(1, 1, 1), (1, 2, 1), (2, 1, 1), (48000, 48000, 24000), (44100, 44101, 44100), (44101, 44100, 44100),
                   (96000, 44100, 44100), (44100, 96000, 44100), (48000, 8000, 8000), (192000, 8000, 8000),
                   (8000, 192000, 8000), (44100, 48000, 20000), (48000, 44100, 10000), (3, 7, 2), (65537, 65536, 65536),
                   (1000, 999, 999), (44100, 1000, 1000), (22050, 48000, 22050), (44100, 16000, 16000), (88200, 48000, 48000), (192000, 44100, 44100), (176400, 48000, 48000)]:
    _add("ratio_%d_%d_%d" % (_i, _o, _l), channels=2, rates=[_i, _o, _l], input="noise", frames=12000, seed=NOISE_SEED ^ (_i * 31 + _o))
# the speech front end's conversion, mono (16-slot windows)
_add("mono_44100_16000", channels=1, rates=[44100, 16000, 16000], input="noise", frames=12000, seed=NOISE_SEED ^ 16000)

# 8-lobe build (CLOWNRESAMPLER_KERNEL_RADIUS=8, clownresampler.h:445-447)
for _i, _o in [(8000, 96000), (44100, 48000), (48000, 44100), (48000, 8000)]:
    _add("r8_%d_%d" % (_i, _o), channels=2, rates=_lp(_i, _o), input="noise", frames=15000, radius=8)
_add("r8_8ch", channels=8, rates=_lp(48000, 44100), input="noise", frames=8000, radius=8)

# any other radius (CLOWNRESAMPLER_KERNEL_RADIUS is a free compile-time knob, clownresampler.h:445-447): 5 lobes, built by default
for _i, _o in [(44100, 48000), (48000, 44100), (8000, 96000), (44100, 8000), (96000, 48000)]:
    _add("r5_%d_%d" % (_i, _o), channels=2, rates=_lp(_i, _o), input="noise", frames=15000, radius=5)
_add("r5_5ch_high", channels=5, rates=_lp(48000, 44100), input="noise", frames=9000, radius=5, mode="high", pull_chunk=700)

# adversarial amplitudes: full-scale square, constant -32768 / +32767, impulse
for _kind in ("square", "min", "max", "impulse", "ramp"):
    _add("amp_%s_up" % _kind, channels=2, rates=_lp(44100, 48000), input=_kind, frames=6000)
    _add("amp_%s_down" % _kind, channels=2, rates=_lp(44100, 8000), input=_kind, frames=6000)
    _add("amp_%s_r8" % _kind, channels=2, rates=_lp(8000, 96000), input=_kind, frames=3000, radius=8)

# ragged / tiny inputs
for _n in (0, 1, 2, 3, 5, 64, 65, 255, 257):
    _add("tiny_%d" % _n, channels=2, rates=_lp(44100, 48000), input="noise", frames=_n)
    _add("tiny_down_%d" % _n, channels=2, rates=_lp(44100, 8000), input="noise", frames=_n)

CASE_BY_NAME = {c["name"]: c for c in CASES}
RADII = (3, 5, 8)   # what csrc/Makefile builds by default and oracle/Makefile compiles the reference for

# configuration scalars: (in, out, lowpass) triples incl. the failure cases of clownresampler.h:919,939,974
CONFIG_TRIPLES = [
    (44100, 48000, 44100), (44100, 48000, 48000), (8000, 96000, 8000), (48000, 44100, 44100), (44100, 8000, 8000),
    (44100, 8000, 44100), (8000, 44100, 44100), (8000, 44100, 8000), (1, 1, 1), (1, 2, 1), (2, 1, 1), (0, 48000, 48000),
    (48000, 0, 48000), (48000, 48000, 0), (0, 0, 0), (4096, 1, 1), (4095, 1, 1), (4097, 1, 1), (1, 4096, 1), (1, 65536, 1),
    (1, 65537, 1), (65536, 1, 65536), (65536, 1, 1), (1 << 31, 1, 1 << 31), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF),
    (0xFFFFFFFF, 1, 0xFFFFFFFF), (1, 0xFFFFFFFF, 1), (48000, 44100, 1), (48000, 44100, 12), (192000, 44100, 96000),
    (3, 7, 2), (7, 3, 5), (1000, 999, 999), (999, 1000, 1000), (123456789, 987654321, 100000000), (44100, 48000, 1 << 20),
    (1023, 1, 1), (1024, 1, 1), (1025, 1, 1), (2048, 1, 1),
]
