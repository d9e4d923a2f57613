"""TEST INFRASTRUCTURE: the product's HOST logic with the product's REAL instance tables, without a GPU, under AddressSanitizer.

Runs in a child process of tests/test_hostshim.py with CLOWNRESAMPLER_AMD_LIBRARY = tests/hostshim/build/libcr_hostshim_inst_asan.so (the host C sources
+ tests/hostshim/crhip_fake.c -DFAKE_REAL_INSTANCES + the HIP objects of the product build, their seam functions weakened) and gcc's libasan preloaded.
Every plan takes the kernel, variant and geometry it takes on the GPU; every launch is checked by the product's own launch function, validated range
by range inside exact-size mallocs ("device memory" here: AddressSanitizer's red zones are the guard pages) and computed by a scalar model from its
arguments alone; every result is held to the oracle.  The plan cache holds TWO plans: every other call evicts and frees rows images.

What it drives - the call sequences of the GPU tests around the place GPUTEST_r05 died, and the launch shapes the plain fake seam cannot reach:
  cases        tests/_cases.py's low-level cases, device-resident, on buffers that hold what clownresampler.h:725-733 asks for and not a byte more
  bulk         the same cases through the host-pointer entry points (staging sets sized by cr_run_host), int32 and clamped int16
  direct       ... with every host buffer "page-locked" (CRA_FAKE_PAGE_LOCKED=1): cr_run_host's one-launch path on the caller's own, exact buffers
  segments     ClownResamplerAMD_ResampleSegmentsDevice, the dying test's segment lists, modes 1 / 2 / 0, radius 3 / 8 / 5, the exact halo
  long         dual mono (k_poly and k_wave2 partners), k_seg (forced and by the rule), k_up2 and its brief shape, k_int with ticket groups and a
               mid-period start, ticketed k_poly, padded tiles, the run-time-slot k_wave2, wide frames, int16 stores
  long_up2     the k_up2 cases again in a process where k_up2 takes launches of any length (CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES=0)
  callback     the callback and high-level forms over specialised plans
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import _cases  # noqa: E402
import _checkers as ck  # noqa: E402
import _product  # noqa: E402
import clownresampler_amd as cr  # noqa: E402

assert "hostshim" in cr.LIB_PATH, "this driver is for the fake-seam library, not the product: " + cr.LIB_PATH
products = {3: _product.Product(3), 5: _product.Product(5), 8: _product.Product(8)}
api = products[3].api
api.SetPlanCacheLimit(2)
kernels_seen = set()
t_start = time.time()


def note(plan):
    info = api.PlanGetInfo(plan)
    kernels_seen.add((info.kernel, info.specialised, info.variant))


class ExactProduct(_product.Product):
    """the low-level call device-resident (ClownResamplerAMD_ResampleDevice) on exact-size "device" buffers (the fake's mallocs)"""

    def __init__(self, base, s16=False, offsets=(0, 0)):
        self.__dict__.update(base.__dict__)
        self.s16, self.offsets = s16, offsets

    def low_resample_i32(self, st, padded, frames, capacity=None, **_kw):
        a = self.api
        ch = st.channels
        R = int(st.cfg.radius_frames)
        src = np.ascontiguousarray(padded, dtype=np.int16)[:(frames + 2 * R) * ch]
        assert src.size == (frames + 2 * R) * ch
        total = int(ck.count_output_frames(st, frames))
        cap = total if capacity is None else min(capacity, total)
        item = 2 if self.s16 else 4
        d_in = a.DeviceAlloc(src.nbytes + self.offsets[0])
        d_out = a.DeviceAlloc(cap * ch * item + self.offsets[1])
        try:
            a.CopyToDevice(d_in + self.offsets[0], src)
            plan = a.PlanCreate(st.raw, self.pre)
            note(plan)
            n, left, ran_out = a.ResampleDevice(plan, st.raw, d_in + self.offsets[0], frames, d_out + self.offsets[1], total + 1 if capacity is None else capacity, s16=self.s16)
            a.StreamSynchronize()
            assert n == cap
            out = np.empty(n * ch, dtype=np.int16 if self.s16 else np.int32)
            a.CopyFromDevice(out, d_out + self.offsets[1])
        finally:
            a.DeviceFree(d_in)
            a.DeviceFree(d_out)
        return out, left, ran_out


def compare(tag, got, want):
    if got.size != want.size or not np.array_equal(got, want):
        bad = np.nonzero(got[:min(got.size, want.size)] != want[:min(got.size, want.size)])[0]
        raise SystemExit("FAIL %s: %d / %d samples, first difference at %s" % (tag, got.size, want.size, bad[:1]))


def run_cases(which):
    for case in _cases.CASES:
        mode = case.get("mode", "low")
        if which == "cases" and mode == "high":
            continue
        want = _cases.run_case(ck.oracle(case["radius"]), case, keep_output=True)["_out"]
        base = products[case["radius"]]
        engine = ExactProduct(base) if which == "cases" else base
        compare("%s %s" % (which, case["name"]), _cases.run_case(engine, case, keep_output=True)["_out"], want)
    if which == "cases":
        # every alignment phase of the device pointers, every channel count
        for ch in range(1, 17):
            for rates, frames in (((44100, 48000, 44100), 1203), ((48000, 44100, 44100), 1203), ((44100, 8000, 8000), 901)):
                o = ck.oracle(3)
                ok, ost = o.low_init(ch, *rates)
                padded = ck.pad_frames(ck.noise_pcm(frames * ch, 77 + ch), ch, int(ost.cfg.radius_frames))
                want, _, _ = o.low_resample_i32(ost, padded, frames)
                for offsets in ((2, 4), (6, 12), (14, 8)):
                    e = ExactProduct(products[3], offsets=offsets)
                    ok, st = e.low_init(ch, *rates)
                    got, left, ran_out = e.low_resample_i32(st, padded, frames)
                    compare("alignment %d ch %r %r" % (ch, rates, offsets), got, want)
    if which == "bulk":
        for name in ("cfg2_1min", "cfg4_1min", "cfg3_1min", "ch1_up", "ch3_down", "ch5_up", "ch12_down", "amp_square_up", "r8_48000_8000", "tiny_65", "ratio_44100_1000_1000", "ratio_2_1_1"):
            case = _cases.CASE_BY_NAME[name]
            p, o = products[case["radius"]], ck.oracle(case["radius"])
            pcm = _cases.make_input(case)
            ch = case["channels"]
            ok, st = p.low_init(ch, *case["rates"])
            ok, ost = o.low_init(ch, *case["rates"])
            padded = ck.pad_frames(pcm, ch, int(ost.cfg.radius_frames))
            want, _, _ = o.low_resample_i32(ost, padded, len(pcm) // ch)
            got, left, ran_out = p.api.LowLevel_ResampleBulkS16(st.raw, p.pre, padded, len(pcm) // ch)
            compare("bulk s16 " + name, got, np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16))


def oracle_segments(o, ch, pcm, halo, segments, first):
    ok, st = o.low_init(ch, *first)
    padded = ck.pad_frames(pcm, ch, halo)
    pos, out, counts = 0, [], []
    for n, *rates in segments:
        assert o.low_adjust(st, *rates)
        R = int(st.cfg.radius_frames)
        x, left, ran_out = o.low_resample_i32(st, padded[(pos + halo - R) * ch:], n)
        assert left == 0 and ran_out == 1
        out.append(x)
        counts.append(len(x) // ch)
        pos += n
    return np.concatenate(out), counts, st


def run_segments():
    first = (44100, 48000, 44100)
    for radius, ch, s16, seed in ((3, 2, False, 82), (3, 1, False, 81), (3, 5, True, 85), (8, 2, False, 87), (8, 1, False, 186), (5, 3, False, 185)):
        p, o = products[radius], ck.oracle(radius)
        rng = np.random.default_rng(seed)
        segments = [(20000, 44100, 48000, 44100), (1, 48000, 44100, 44100), (0, 44100, 44100, 44100), (15000, 48000, 44100, 44100),
                    (9000, 44100, 44100, 22050), (3, 44100, 8000, 8000), (16000, 44100, 88200, 44100), (2, 8000, 44100, 8000)]
        segments += [(int(rng.integers(1, 4000)), int(rng.integers(8000, 96000)), int(rng.integers(8000, 96000)), int(rng.integers(8000, 96000))) for _ in range(24)]
        frames = sum(s[0] for s in segments)
        probe = o.low_init(ch, *first)[1]
        halo = 0
        for n, *rates in segments:
            o.low_adjust(probe, *rates)
            halo = max(halo, int(probe.cfg.radius_frames))
        pcm = ck.noise_pcm(frames * ch, 5)
        want, want_counts, ost = oracle_segments(o, ch, pcm, halo, segments, first)
        if s16:
            want = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
        timeline = ck.pad_frames(pcm, ch, halo)
        for mode in (1, 2, 0):
            st = p.api.LowLevel_State()
            p.api.LowLevel_Init(st, ch, *first)
            d_in = p.api.DeviceAlloc(timeline.nbytes)
            d_out = p.api.DeviceAlloc(want.nbytes)
            p.api.DebugSegmentsMode(mode)
            try:
                p.api.CopyToDevice(d_in, timeline)
                n, counts = p.api.ResampleSegmentsDevice(st, p.pre, d_in + halo * ch * 2, halo, segments, d_out, len(want) // ch, s16=s16)
                p.api.StreamSynchronize()
                got = np.empty_like(want)
                p.api.CopyFromDevice(got, d_out)
            finally:
                p.api.DebugSegmentsMode(0)
                p.api.DeviceFree(d_in)
                p.api.DeviceFree(d_out)
            assert n == len(want) // ch and counts == want_counts, (radius, ch, mode)
            compare("segments r%d %d ch mode %d" % (radius, ch, mode), got, want)
            assert (st.position_integer, st.position_fractional, st.increment) == (ost.pos_int, ost.pos_frac, ost.increment)


LONG = [
    ("dual_mono_k_poly", 3, 1, (44100, 48000, 44100), 1400000, False, None, 0),
    ("dual_mono_k_wave2", 8, 1, (44100, 48000, 44100), 1400000, False, None, 0),
    ("dual_mono_mid_stream", 3, 1, (44100, 48000, 44100), 1400000, False, None, 12345),
    ("k_seg_12x_forced", 8, 2, (8000, 96000, 8000), 40000, False, "seg", 0),
    ("k_seg_8x_forced", 8, 2, (8000, 64000, 8000), 40000, False, "seg", 7),
    ("cfg3_one_minute", 8, 2, (8000, 96000, 8000), 480000, False, None, 0),
    ("k_up2_12x", 8, 2, (8000, 96000, 8000), 60000, False, "noseg", 0),
    ("k_up2_brief", 8, 2, (8000, 96000, 8000), 700, False, None, 0),
    ("k_up2_mono_10x", 8, 1, (8000, 80000, 8000), 50000, False, None, 0),
    ("k_int_2to1", 3, 2, (96000, 48000, 48000), 900000, False, None, 0),
    ("k_int_3to2_s16", 3, 2, (72000, 48000, 48000), 300000, True, None, 0),
    ("k_int_3to2_mid_period", 3, 2, (72000, 48000, 48000), 300000, False, None, 1),
    ("k_int_4to1_8ch", 3, 8, (192000, 48000, 48000), 200000, False, None, 0),
    ("k_int_6to1", 3, 2, (48000, 8000, 8000), 300000, False, None, 0),
    ("ticketed_stereo", 3, 2, (44100, 48000, 44100), 4000000, False, None, 0),
    ("wide_12ch_down", 3, 12, (48000, 44100, 44100), 100000, False, None, 0),
    ("padded_tiles_9ch", 3, 9, (44100, 48000, 44100), 100000, False, None, 0),
    ("rt_wave2_5ch_r8", 8, 5, (48000, 44100, 44100), 100000, False, None, 0),
    ("hq48_s16", 8, 2, (48000, 44100, 44100), 300000, True, None, 0),
    ("hq44", 8, 2, (44100, 48000, 44100), 300000, False, None, 0),
    ("dn8", 8, 2, (44100, 8000, 8000), 300000, False, None, 0),
    ("up8_3lobes", 3, 8, (44100, 48000, 44100), 300000, False, None, 0),
    ("mono_down", 3, 1, (48000, 44100, 44100), 300000, False, None, 0),
]


def run_long(only=None):
    for name, radius, ch, rates, frames, s16, hook, skip in LONG:
        if only is not None and not name.startswith(only):
            continue
        base, o = products[radius], ck.oracle(radius)
        e = ExactProduct(base, s16=s16)
        ok, st = e.low_init(ch, *rates)
        ok, ost = o.low_init(ch, *rates)
        R = int(ost.cfg.radius_frames)
        padded = ck.pad_frames(ck.noise_pcm(frames * ch, 4242 + frames), ch, R)
        pos = 0
        if skip:
            # a stream resumed mid-way: a first call of `skip` input frames carries position and fraction into the launch under test
            got0, left, _ = e.low_resample_i32(st, padded, skip)
            want0, _, _ = o.low_resample_i32(ost, padded, skip)
            compare(name + " (lead-in)", got0, want0 if not s16 else np.clip(want0, -0x7FFF, 0x7FFF).astype(np.int16))
            pos = skip
        want, _, _ = o.low_resample_i32(ost, padded[pos * ch:], frames - pos)
        if s16:
            want = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
        if hook == "seg":
            base.api.DebugSegKernel(1)
        elif hook == "noseg":
            base.api.DebugSegKernel(2)
        try:
            before = [base.api.LaunchCount(k) for k in range(9)]
            got, left, ran_out = e.low_resample_i32(st, padded[pos * ch:], frames - pos)
            launched = [base.api.LaunchCount(k) - before[k] for k in range(9)]
        finally:
            base.api.DebugSegKernel(0)
        compare(name, got, want)
        assert st.astuple() == tuple(int(v) for v in ost.astuple()), name
        if hook == "seg":
            assert launched[8] == 1, (name, launched)
        if name.startswith("k_int"):
            assert launched[5] >= 1, (name, launched)
        if only == "k_up2" and ch == 2:
            assert launched[3] == 1, (name, launched)   # (mono has no k_up2: the chain kernels / k_wave2 take it)
        print("  %-24s launches by kernel %r" % (name, launched), flush=True)


def run_callback():
    for name in ("flac_cfg1_high", "high_small_pulls", "high_down", "r5_5ch_high", "cfg2_earlystop", "down_earlystop"):
        case = _cases.CASE_BY_NAME[name]
        want = _cases.run_case(ck.oracle(case["radius"]), case, keep_output=True)["_out"]
        compare("callback " + name, _cases.run_case(products[case["radius"]], case, keep_output=True)["_out"], want)
    # the callback form over a long specialised plan: growing batches, then the compute-ahead thread
    p, o = products[3], ck.oracle(3)
    frames, ch = 2500000, 2
    ok, st = p.low_init(ch, 44100, 48000, 44100)
    ok, ost = o.low_init(ch, 44100, 48000, 44100)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch, 9), ch, 3)
    want, _, _ = o.low_resample_i32(ost, padded, frames)
    out = []
    r, left = p.low_resample_cb(st, padded, frames, lambda frame: (out.extend(frame), True)[1])
    compare("callback long", np.array(out, dtype=np.int64).astype(np.int32), want)


steps = sys.argv[1:] or ["cases", "bulk", "direct", "segments", "long", "callback"]
for step in steps:
    t0 = time.time()
    if step in ("cases", "bulk"):
        run_cases(step)
    elif step == "direct":
        os.environ["CRA_FAKE_PAGE_LOCKED"] = "1"
        try:
            run_cases("bulk")
        finally:
            os.environ["CRA_FAKE_PAGE_LOCKED"] = "0"
    elif step == "segments":
        run_segments()
    elif step == "long":
        run_long()
    elif step == "long_up2":
        # (to be run with CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES=0 in the environment: k_up2 then takes launches of any length - on the GPU it
        # takes them from a few million frames on, more than the scalar models here should be asked for)
        assert os.environ.get("CLOWNRESAMPLER_AMD_BRIEF_HALF_TILES") == "0"
        run_long("k_up2")
    elif step == "callback":
        run_callback()
    bad = cr.self_check()
    assert not bad, (step, bad)
    print("ok %s (%.0f s)" % (step, time.time() - t0), flush=True)
print("kernels (kernel, specialised, variant) seen:", sorted(kernels_seen))
api.Shutdown()
print("instances driver: all passed")
