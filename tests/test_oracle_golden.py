"""CPU: the oracle (oracle/cr_oracle.c) against the committed known answers of the REAL reference
(tests/golden/golden.json, made by tests/golden/make_golden.py from /root/reference compiled in place).
Runs anywhere - also on the GPU box, where /root/reference does not exist."""
import hashlib

import numpy as np
import pytest

import _cases
import _checkers as ck


def test_tables_match_reference(golden):
    # clownresampler.h:955-961; hashes also quoted in SURVEY.md 8(a) a-4
    for r in _cases.RADII:
        t = ck.oracle(r).table()
        g = golden["table"][str(r)]
        assert len(t) == g["len"]
        assert hashlib.sha256(t.astype("<i4").tobytes()).hexdigest() == g["sha256_i32le"]
        assert int(t.sum()) == g["sum"] and int(np.abs(t).sum()) == g["abs_sum"] and int(t[512]) == g["t512"]
        assert t[0] == 0 and t[len(t) // 2] == 65536


def test_survey_table_hashes(golden):
    assert golden["table"]["3"]["sha256_i32le"] == "cdeefa3536912c3caa5bb94bb345ab0ea65a953aec18081552fc48a4a0309240"
    assert golden["table"]["8"]["sha256_i32le"] == "40aa850733dcdcabb6b13f37d51c6d0970b35414826d7a3e15450a5259d80ea8"


@pytest.mark.parametrize("radius", list(_cases.RADII))
def test_config_scalars(golden, radius):
    # clownresampler.h:913-953, :963-984, :1044-1056 incl. failure cases; state is pre-poisoned so that
    # "what is left untouched on failure" is part of the comparison
    o = ck.oracle(radius)
    for row in golden["config"][str(radius)]:
        st = ck.LowLevel()
        for f in ("pos_int", "pos_frac", "increment"):
            setattr(st, f, 0x5A5A5A5A)
        st.cfg.stretched_radius = st.cfg.radius_frames = st.cfg.radius_delta = st.cfg.table_step = 0x5A5A5A5A
        ok, st = o.low_init(2, *row["rates"], st)
        assert int(ok) == row["ok"], row
        assert [int(v) for v in st.astuple()] == row["state"], row
        assert int(o.ratio(row["rates"][0], row["rates"][1])) == row["ratio_in_out"]


def test_known_config_values():
    # SURVEY.md 8(a) a-3 "verified values"
    o = ck.oracle(3)
    assert o.low_init(2, 44100, 48000, 44100)[1].astuple() == (196608, 3, 0, 1024, 2, 0, 0, 60211)
    assert o.low_init(8, 48000, 44100, 44100)[1].astuple() == (213993, 4, 48151, 940, 8, 0, 0, 71331)
    assert o.low_init(2, 44100, 8000, 8000)[1].astuple() == (1083801, 17, 30311, 185, 2, 0, 0, 361267)
    assert ck.oracle(8).low_init(2, 8000, 96000, 8000)[1].astuple() == (524288, 8, 0, 1024, 2, 0, 0, 5461)


@pytest.mark.parametrize("case", _cases.CASES, ids=[c["name"] for c in _cases.CASES])
def test_case_matches_reference(golden, case):
    res = _cases.run_case(ck.oracle(case["radius"]), case, keep_output=True)
    out = res.pop("_out")
    assert res == golden["cases"][case["name"]]
    if case["name"] in golden["vectors"]:
        assert [int(v) for v in out] == golden["vectors"][case["name"]]


def test_harness_outputs(golden):
    # the reference's own harness binaries (tests/test-low-level.c, tests/test-high-level.c) on tests/test.flac:
    # high == low for every triple (tests/CMakeLists.txt uses one golden for both), and the oracle reproduces them
    h = golden["harness"]
    for name in ("cfg1", "ctest1", "ctest2", "ctest3", "ctest4"):
        assert h[name + "_low"] == h[name + "_high"]
        assert golden["cases"]["flac_%s_low" % name]["sha256"] == h[name + "_low"]["sha256"]
    assert h["cfg1_low"]["sha256"] == "a94f16df9e50fb1d8ab3f12e02a26f7b8f29aa47d61cd23e375c03bfbc3f782a"
    assert h["cfg1_low"]["bytes"] == 1671848


def test_reference_golden_test3_pins_pre_normalisation():
    """tests/test3 (== tests/test4) of the reference is stale w.r.t. the shipped normalisation
    (finding 1) but is reproduced exactly when only the last step is replaced by the
    constant gain ratio(lowpass, in): it pins tap bounds, table indices, per-tap truncation and stepping."""
    gold = np.fromfile(_cases.GOLDEN_DIR + "/ref_test3.bin", dtype="<i4")
    o = ck.oracle(3)
    pcm = np.fromfile(_cases.FLAC_PCM, dtype="<i2")
    for rates in ((44100, 8000, 44100), (44100, 8000, 8000)):
        ok, st = o.low_init(2, *rates)
        padded = ck.pad_frames(pcm, 2, int(st.cfg.radius_frames))
        gain = o.ratio(min(rates), rates[0])
        out, left, ran_out = o.low_resample_i32(st, padded, len(pcm) // 2, norm_mode=ck.NORM_LEGACY_GAIN, legacy_gain=gain)
        assert ran_out == 1 and out.size == gold.size
        assert np.array_equal(out, gold)
        # and the shipped normalisation differs from it, by little (max |delta| 11 per the survey)
        ok, st = o.low_init(2, *rates)
        cur, _, _ = o.low_resample_i32(st, padded, len(pcm) // 2)
        d = np.abs(cur.astype(np.int64) - gold)
        assert 0 < d.max() <= 11


def test_single_frames(golden):
    # clownresampler.h:986-1035 incl. the "+=" into a non-zero accumulator
    for f in golden["single_frames"]:
        o = ck.oracle(f["radius"])
        ok, cfg = o.configure(*f["rates"])
        pcm = ck.noise_pcm(f["pcm_frames"] * f["channels"], f["seed"])
        out = o.frame(cfg, f["channels"], pcm, f["pos_int"], f["pos_frac"], f["acc_in"])
        assert [int(v) for v in out] == f["acc_out"], f


def test_noise_generator_head():
    # SURVEY.md 8(d)
    assert list(ck.noise_pcm(12)) == [-9189, 25840, 31495, 12383, 11499, -26864, -25902, -9814, -8790, -28786, 2292, 11949]


def test_closed_form_count():
    o = ck.oracle(3)
    ok, st = o.low_init(2, 44100, 48000, 44100)
    assert ck.count_output_frames(st, 26460000) == 28800096     # cfg 2, SURVEY.md 8(a) a-2
    assert ck.count_output_frames(st, 158760000) == 172800574   # cfg 5
    assert ck.count_output_frames(st, 192000) == 208981         # cfg 1
    assert o._count(st, 26460000) == 28800096


def test_multithreaded_baseline_equals_single_thread():
    o = ck.oracle(3)
    ok, st = o.low_init(2, 44100, 48000, 44100)
    frames = 50000
    padded = ck.pad_frames(ck.noise_pcm(frames * 2), 2, 3)
    one, _, _ = o.low_resample_i32(st, padded, frames)
    for threads in (1, 2, 3, 8):
        ok, fresh = o.low_init(2, 44100, 48000, 44100)
        assert np.array_equal(o.low_resample_i32_mt(fresh, padded, frames, threads), one)


def test_multithreaded_oracle_from_a_carried_state_equals_single_thread():
    """bench.py checks every rank's WHOLE shard with the all-core driver; a shard's state is not fresh (position carried over from
    the closed form, clownresampler.h:1076-1078), so the driver must start its timeline there."""
    o = ck.oracle(3)
    for ch, rates, pos in ((2, (44100, 48000, 44100), (0, 12345)), (1, (48000, 44100, 44100), (2, 65535)), (3, (44100, 8000, 8000), (7, 1)), (2, (8000, 44100, 8000), (0, 0))):
        frames = 30011
        ok, st = o.low_init(ch, *rates)
        R = int(st.cfg.radius_frames)
        padded = ck.pad_frames(ck.noise_pcm(frames * ch, 5), ch, R)
        st.pos_int, st.pos_frac = pos
        ok, carried = o.low_init(ch, *rates)
        carried.pos_int, carried.pos_frac = pos
        one, _, _ = o.low_resample_i32(st, padded, frames)
        assert one.size == ck.count_output_frames(carried, frames) * ch
        for threads in (1, 2, 5, 8):
            assert np.array_equal(o.low_resample_i32_mt(carried, padded, frames, threads), one), (ch, rates, pos, threads)
