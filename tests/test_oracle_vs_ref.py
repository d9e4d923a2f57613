"""CPU, build container only: the oracle against the REAL reference compiled in place
(oracle/_ref/libclownref_r*.so built by oracle/Makefile from /root/reference), over randomized sweeps
wider than the committed fixtures.  Skipped where no compiled reference is available."""
import random

import numpy as np
import pytest

import _checkers as ck

pytestmark = pytest.mark.skipif(ck.reference(3) is None, reason="compiled reference (oracle/_ref) not available")


def test_struct_layout_is_the_references():
    r = ck.reference(3)
    L = r.lib
    import ctypes as C
    for f in ("ref_sizeof_config", "ref_sizeof_lowlevel", "ref_sizeof_highlevel", "ref_sizeof_precomputed"):
        getattr(L, f).restype = C.c_size_t
    L.ref_offsetof_lowlevel.restype = C.c_size_t
    L.ref_offsetof_highlevel.restype = C.c_size_t
    assert L.ref_sizeof_config() == C.sizeof(ck.Config) == 32
    assert L.ref_sizeof_lowlevel() == C.sizeof(ck.LowLevel) == 64
    assert L.ref_sizeof_highlevel() == C.sizeof(ck.HighLevel) == 8296
    assert L.ref_sizeof_precomputed() == 6144 * 8
    assert [L.ref_offsetof_lowlevel(i) for i in range(5)] == [getattr(ck.LowLevel, n).offset for n in ("cfg", "channels", "pos_int", "pos_frac", "increment")]
    assert [L.ref_offsetof_highlevel(i) for i in range(7)] == [getattr(ck.HighLevel, n).offset for n in
                                                             ("low", "staging", "win_begin", "win_end", "max_radius_frames", "lead_needed", "trail_left")]


def test_ratio_sweep():
    o, r = ck.oracle(3), ck.reference(3)
    rng = random.Random(1)
    vals = [0, 1, 2, 3, 65535, 65536, 65537, 1 << 20, (1 << 31) - 1, 1 << 31, 0xFFFFFFFF, 44100, 48000, 8000, 96000]
    for a in vals:
        for b in vals:
            assert o.ratio(a, b) == r.ratio(a, b), (a, b)
    for _ in range(20000):
        a = rng.choice([rng.randrange(1, 1 << 32), rng.randrange(1, 200000), rng.randrange(1, 70000)])
        b = rng.choice([rng.randrange(1, 1 << 32), rng.randrange(1, 200000), rng.randrange(1, 70000)])
        assert o.ratio(a, b) == r.ratio(a, b), (a, b)


@pytest.mark.parametrize("radius", [3, 5, 8])
def test_configure_sweep(radius):
    o, r = ck.oracle(radius), ck.reference(radius)
    rng = random.Random(2)
    for _ in range(5000):
        rates = [rng.choice([0, 1, rng.randrange(1, 400000), rng.randrange(1, 1 << 32)]) for _ in range(3)]
        a, b = ck.LowLevel(), ck.LowLevel()
        ok_a, a = o.low_init(3, *rates, a)
        ok_b, b = r.low_init(3, *rates, b)
        assert ok_a == ok_b and a.astuple() == b.astuple(), rates


def _random_case(rng):
    radius = rng.choice([3, 3, 8, 5])
    ch = rng.choice([1, 2, 2, 3, 4, 6, 8, 16])
    i, o = rng.randrange(1, 200000), rng.randrange(1, 200000)
    if rng.random() < 0.3:
        o = max(1, int(i * rng.choice([0.03, 0.5, 0.9, 0.999, 1.0, 1.001, 1.1, 2, 12, 40])))
    lp = rng.choice([min(i, o), i, o, max(1, min(i, o) // rng.randrange(1, 5)), rng.randrange(1, 200000)])
    frames = rng.randrange(0, 3000)
    return radius, ch, (i, o, lp), frames


def test_random_streams_bit_exact():
    rng = random.Random(3)
    done = 0
    while done < 150:
        radius, ch, rates, frames = _random_case(rng)
        o, r = ck.oracle(radius), ck.reference(radius)
        ok_a, a = o.low_init(ch, *rates)
        ok_b, b = r.low_init(ch, *rates)
        assert ok_a == ok_b
        if not ok_a or a.cfg.table_step == 0:   # step 0: reference divides by zero
            continue
        if a.cfg.radius_frames > 400:
            continue
        pcm = ck.noise_pcm(frames * ch, 99 + done)
        padded = ck.pad_frames(pcm, ch, int(a.cfg.radius_frames))
        xa, la, ea = o.low_resample_i32(a, padded, frames)
        xb, lb, eb = r.low_resample_i32(b, padded, frames)
        assert np.array_equal(xa, xb) and (la, ea) == (lb, eb) and a.astuple() == b.astuple(), (radius, ch, rates, frames)
        done += 1


def test_random_early_stop_and_chunks_bit_exact():
    rng = random.Random(4)
    for it in range(60):
        radius, ch, rates, frames = _random_case(rng)
        frames += 200
        o, r = ck.oracle(radius), ck.reference(radius)
        ok, a = o.low_init(ch, *rates)
        ok2, b = r.low_init(ch, *rates)
        if not ok or a.cfg.table_step == 0 or a.cfg.radius_frames > 400:
            continue
        R = int(a.cfg.radius_frames)
        padded = ck.pad_frames(ck.noise_pcm(frames * ch, 1000 + it), ch, R)
        pos = 0
        left_a = left_b = frames
        for _ in range(10000):
            cap = rng.randrange(0, 50)
            xa, left_a2, ea = o.low_resample_i32(a, padded[pos * ch:], left_a, capacity=cap)
            xb, left_b2, eb = r.low_resample_i32(b, padded[pos * ch:], left_b, capacity=cap)
            assert np.array_equal(xa, xb) and left_a2 == left_b2 and ea == eb and a.astuple() == b.astuple()
            pos += left_a - left_a2
            left_a = left_b = left_a2
            if ea:
                break
            if rng.random() < 0.1:   # mid-stream re-configuration (clownresampler.h:1052-1056)
                nr = _random_case(rng)[2]
                ra, rb = o.low_adjust(a, *nr), r.low_adjust(b, *nr)
                assert ra == rb and a.astuple() == b.astuple()
                if not ra or a.cfg.table_step == 0 or a.cfg.radius_frames > R:
                    break
        else:
            raise AssertionError("did not terminate")


def test_callback_api_and_highlevel_bit_exact():
    rng = random.Random(5)
    for it in range(25):
        radius, ch, rates, frames = _random_case(rng)
        o, r = ck.oracle(radius), ck.reference(radius)
        ok, a = o.high_init(ch, *rates)
        ok2, b = r.high_init(ch, *rates)
        assert ok == ok2
        if not ok or a.low.cfg.table_step == 0 or a.low.cfg.radius_frames * 2 * ch >= 0x1000 // 2:
            continue
        pcm = ck.noise_pcm(frames * ch, 2000 + it)
        chunk = rng.choice([0, 1, 13, 500])
        xa = o.high_run_i32(a, pcm, chunk)
        xb = r.high_run_i32(b, pcm, chunk)
        assert np.array_equal(xa, xb), (radius, ch, rates, frames, chunk)
        assert a.low.astuple() == b.low.astuple()
        assert (a.lead_needed, a.trail_left, a.max_radius_frames) == (b.lead_needed, b.trail_left, b.max_radius_frames)
        # high-level == low-level one-shot over zero padding (tests/CMakeLists.txt shares the goldens)
        ok, l = o.low_init(ch, *rates)
        xl, _, _ = o.low_resample_i32(l, ck.pad_frames(pcm, ch, int(l.cfg.radius_frames)), frames)
        assert np.array_equal(xa, xl)


def test_highlevel_adjust_matches():
    o, r = ck.oracle(3), ck.reference(3)
    for first, then in [((48000, 8000, 8000), (48000, 44100, 44100)), ((44100, 48000, 44100), (48000, 8000, 8000)),
                        ((48000, 24000, 24000), (0, 1, 1)), ((48000, 24000, 24000), (48000, 12000, 12000))]:
        ok, a = o.high_init(2, *first)
        ok2, b = r.high_init(2, *first)
        ra, rb = o.high_adjust(a, *then), r.high_adjust(b, *then)
        assert ra == rb and a.low.astuple() == b.low.astuple(), (first, then)


def test_python_callback_path_early_stop_state():
    # output callback returning 0 (clownresampler.h:1081-1089) through the real callback ABI
    o, r = ck.oracle(3), ck.reference(3)
    ok, a = o.low_init(2, 44100, 48000, 44100)
    ok, b = r.low_init(2, 44100, 48000, 44100)
    padded = ck.pad_frames(ck.noise_pcm(2000), 2, 3)
    got_a, got_b = [], []
    ra = o.low_resample_cb(a, padded, 1000, lambda f: (got_a.append(f), len(got_a) < 100)[1])
    rb = r.low_resample_cb(b, padded, 1000, lambda f: (got_b.append(f), len(got_b) < 100)[1])
    assert ra == rb and got_a == got_b and len(got_a) == 100 and a.astuple() == b.astuple()


@pytest.mark.parametrize("radius", [3, 8])
def test_highlevel_adjust_mid_stream_scripts(radius):
    """ClownResampler_HighLevel_Adjust BETWEEN ClownResampler_HighLevel_Resample / ResampleEnd calls (clownresampler.h:1183-1209 on a
    stream in flight: radius kept, shrunk :1165, or the change rejected :1195): the oracle's restatement against the compiled
    reference over random scripted sessions - return values, every emitted frame, the state after every step.  The same scripts
    (tests/_scripts.py) hold the GPU path to the oracle in tests/test_gpu_parity.py."""
    import _scripts
    o, r = ck.oracle(radius), ck.reference(radius)
    done = accepted = rejected = shrunk = 0
    seed = 0
    while done < 40:
        seed += 1
        script = _scripts.make_script(1000 * radius + seed, radius)
        if not _scripts.usable(script, o):
            continue
        a, b = _scripts.play(o, script), _scripts.play(r, script)
        assert _scripts.first_difference(a, b) is None, (script["seed"], _scripts.first_difference(a, b))
        done += 1
        max_radius = a[0][1][8]
        for step in a:
            if step[0] == "adjust":
                accepted += step[2]
                rejected += 1 - step[2]
                shrunk += step[2] and step[3][1] < max_radius
    assert accepted > 40 and rejected > 20 and shrunk > 10, (accepted, rejected, shrunk)
