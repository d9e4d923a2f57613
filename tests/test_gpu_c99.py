"""The CC_USE_C99_INTEGERS ABI on the GPU (reference clownresampler.h:483-501: int_least32_t table entries = 4 bytes,
uint_fast8_t channel counts and callback sample counts = 1 byte on glibc x86-64).  It is a second build of the same host sources
(libclownresampler_amd_c99.so, tools/bin/cr_resample_c99) over the same HIP objects; what differs on the way to the device is
the table upload (int32 entries instead of int64) and every struct / callback that carries a cc_* type.  These tests load THAT
library - through a second ctypes mirror (`cr.load(radius, "c99")`) and through its C client - and hold it to the same oracle
arrays and golden vectors as the default library."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

import _cases
import _checkers as ck
import _product
import clownresampler_amd as cr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def products99():
    assert cr.load(3, "c99").DeviceCount() > 0, "these tests need the GPU: the library has no other path"
    return {r: _product.Product(r, "c99") for r in (3, 5, 8)}


def test_c99_library_is_the_one_loaded(products99):
    p = products99[3]
    assert p.api.lib._name.endswith("libclownresampler_amd_c99.so")
    assert p.api.BuildId() == cr.load(3).BuildId()                       # one set of sources, two ABIs
    import ctypes as C
    assert C.sizeof(p.api.Precomputed) == 6144 * 4 and C.sizeof(cr.load(3).Precomputed) == 6144 * 8
    assert C.sizeof(p.api.T.cc_u8f) == 1
    # the 4-byte table holds the reference's values (SURVEY a-4)
    assert np.array_equal(p.table(), ck.oracle(3).table())
    assert np.array_equal(products99[8].table(), ck.oracle(8).table())


def test_c99_harness_reproduces_reference_harness_outputs(golden, tmp_path):
    """tools/cr_resample.c compiled with -DCC_USE_C99_INTEGERS against libclownresampler_amd_c99.so: the on-disk int32-LE stream
    (tests/test-low-level.c:43-49) does not depend on the integer ABI, so the sha256s are the real reference harnesses'."""
    exe = os.path.join(ROOT, "tools", "bin", "cr_resample_c99")
    assert os.path.exists(exe), "build() makes tools/bin/cr_resample_c99"
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libclownresampler_amd_c99.so" in ldd and "libclownresampler_amd.so" not in ldd
    for name, rates in [("cfg1", (44100, 48000, 44100)), ("ctest1", (8000, 44100, 44100)), ("ctest3", (44100, 8000, 44100)), ("ctest4", (44100, 8000, 8000))]:
        for mode in ("low", "high", "bulk"):
            outp = str(tmp_path / "o.bin")
            subprocess.run([exe, mode, _cases.FLAC_PCM, outp, "2"] + [str(r) for r in rates], check=True, timeout=300)
            got = hashlib.sha256(open(outp, "rb").read()).hexdigest()
            assert got == golden["harness"][name + "_low"]["sha256"], (name, mode)


C99_CASES = ["cfg2_1min", "cfg3_1min", "cfg4_1min", "ch16_down", "ch1_up", "flac_cfg1_high", "high_down", "r5_5ch_high", "cfg2_earlystop", "down_chunked",
             "r8_48000_44100", "r5_44100_8000", "ratio_192000_8000_8000", "amp_min_down", "tiny_1", "tiny_0"]


@pytest.mark.parametrize("name", C99_CASES)
def test_c99_case_bit_exact(golden, products99, name):
    """Cases of tests/_cases.py through the C99-integer library: every kernel family (k_poly cfg 2 / cfg 4 / 16 channels, k_up2 /
    k_wave2 cfg 3 and the 8-lobe shapes, k_int 192 -> 8 kHz), the callback-free and high-level entry points, resume semantics:
    arrays equal the oracle's, summaries equal the real reference's (tests/golden/golden.json)."""
    case = _cases.CASE_BY_NAME[name]
    p = products99[case["radius"]]
    res = _cases.run_case(p, case, keep_output=True)
    out = res.pop("_out")
    want = _cases.run_case(ck.oracle(case["radius"]), case, keep_output=True)["_out"]
    assert np.array_equal(out, want)
    assert res == golden["cases"][name]


@pytest.mark.parametrize("name", ["cfg2_1min", "cfg3_1min", "ch16_down", "ratio_44100_1000_1000", "tiny_65"])
def test_c99_generic_kernel_bit_exact(golden, products99, name):
    """k_generic reads the caller's TABLE on the device (the other kernels only see rows derived from it on the host): the path on
    which the 4-byte entries of this ABI are what gets uploaded."""
    case = _cases.CASE_BY_NAME[name]
    p = products99[case["radius"]]
    p.api.DebugForceGenericKernel(True)
    try:
        before = p.api.LaunchCount(0)
        res = _cases.run_case(p, case)
        assert p.api.LaunchCount(0) > before or res["frames"] == 0
    finally:
        p.api.DebugForceGenericKernel(False)
    assert res == golden["cases"][name]


def test_c99_callback_api_and_single_frames(golden, products99):
    """The reference's own signatures with this ABI's widths: ClownResampler_LowLevel_Resample's per-frame callback (cc_u8f is one
    byte here) with an early stop, and ClownResampler_LowestLevel_Resample accumulating into the caller's frame."""
    p, o = products99[3], ck.oracle(3)
    ch, rates, frames = 3, (48000, 44100, 44100), 9000
    ok, st = p.low_init(ch, *rates)
    ok, ost = o.low_init(ch, *rates)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch, 99), ch, int(ost.cfg.radius_frames))
    a, b = [], []
    ra = p.low_resample_cb(st, padded, frames, lambda f: (a.append(f), len(a) < 777)[1])
    rb = o.low_resample_cb(ost, padded, frames, lambda f: (b.append(f), len(b) < 777)[1])
    assert ra == rb and a == b and st.astuple() == ost.astuple()
    ok, cfg = p.configure(*rates)
    ok, ocfg = o.configure(*rates)
    for pos_int, pos_frac, acc in [(0, 0, None), (5, 12345, [7, -9, 100000]), (100, 65535, [-(1 << 20), 1 << 20, 0])]:
        got = p.frame(cfg, ch, padded, pos_int, pos_frac, acc)
        want = o.frame(ocfg, ch, padded, pos_int, pos_frac, acc)
        assert np.array_equal(got, want)


def test_c99_highlevel_adjust_mid_stream(products99):
    """The scripted high-level sessions of tests/_scripts.py (Resample stopped by the consumer / Adjust accepted, shrunk, rejected /
    ResampleEnd in pieces) through this ABI's structs and callbacks."""
    import _scripts
    done, seed = 0, 0
    while done < 10:
        seed += 1
        radius = (3, 8)[seed % 2]
        p, o = products99[radius], ck.oracle(radius)
        script = _scripts.make_script(90000 + seed, radius)
        if not _scripts.usable(script, o):
            continue
        a, b = _scripts.play(p, script, early_end=False), _scripts.play(o, script, early_end=False)
        assert _scripts.first_difference(a, b) is None, (script["seed"], _scripts.first_difference(a, b))
        done += 1
