"""ctypes bindings for the CHECKERS: the CPU restatement (oracle/libcr_oracle.so) and, when it
has been built, the real reference compiled in place (oracle/_ref/libclownref_r<R>.so).

Test infrastructure only - the product package never imports this.  Both libraries export the
same function set (prefix ``oracle_`` / ``ref_``) over LP64-layout structs identical to the
reference's (clownresampler.h:632-659), so `Checker` wraps either.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REFERENCE_DIR = "/root/reference"

NORM_CURRENT = 0
NORM_LEGACY_GAIN = 1


class Config(C.Structure):  # clownresampler.h:632-638
    _fields_ = [("stretched_radius", C.c_uint64), ("radius_frames", C.c_uint64),
                ("radius_delta", C.c_uint64), ("table_step", C.c_uint64)]

    def astuple(self):
        return (self.stretched_radius, self.radius_frames, self.radius_delta, self.table_step)


class LowLevel(C.Structure):  # clownresampler.h:640-648
    _fields_ = [("cfg", Config), ("channels", C.c_uint32), ("pos_int", C.c_uint64),
                ("pos_frac", C.c_uint64), ("increment", C.c_uint64)]

    def astuple(self):
        return self.cfg.astuple() + (self.channels, self.pos_int, self.pos_frac, self.increment)


class HighLevel(C.Structure):  # clownresampler.h:650-659
    _fields_ = [("low", LowLevel), ("staging", C.c_int16 * 0x1000), ("win_begin", C.c_void_p),
                ("win_end", C.c_void_p), ("max_radius_frames", C.c_uint64), ("lead_needed", C.c_uint64),
                ("trail_left", C.c_uint64)]


OUTPUT_CB = C.CFUNCTYPE(C.c_uint8, C.c_void_p, C.POINTER(C.c_int64), C.c_uint32)
INPUT_CB = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(C.c_int16), C.c_size_t)


def build_checkers():
    """(Re)build the oracle and, when /root/reference exists, the compiled reference."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class Checker:
    """One checker library: prefix 'oracle' (radius is a run-time argument) or 'ref' (radius baked in)."""

    def __init__(self, path, prefix, radius):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        self.radius = radius
        L = self.lib
        u64, u32, u8, sz, vp = C.c_uint64, C.c_uint32, C.c_uint8, C.c_size_t, C.c_void_p
        pi64, pi16, pi32 = C.POINTER(C.c_int64), C.POINTER(C.c_int16), C.POINTER(C.c_int32)

        def fn(name, res, args):
            f = getattr(L, prefix + "_" + name)
            f.restype, f.argtypes = res, args
            return f

        self._table_len = fn("table_len", sz, [C.c_uint])
        self._precompute = fn("precompute", None, [pi64, C.c_uint])
        self._ratio = fn("ratio", u64, [u64, u64])
        self._configure = fn("configure", u8, [C.POINTER(Config), C.c_uint, u64, u64, u64])
        self._frame = fn("frame", None, [C.POINTER(Config), pi64, sz, pi64, u32, pi16, u64, u64])
        self._low_init = fn("low_init", u8, [C.POINTER(LowLevel), C.c_uint, u32, u64, u64, u64])
        self._low_adjust = fn("low_adjust", u8, [C.POINTER(LowLevel), C.c_uint, u64, u64, u64])
        self._low_resample = fn("low_resample", u8, [C.POINTER(LowLevel), pi64, sz, pi16, C.POINTER(sz), OUTPUT_CB, vp])
        self._high_init = fn("high_init", u8, [C.POINTER(HighLevel), C.c_uint, u32, u64, u64, u64])
        self._high_resample = fn("high_resample", u8, [C.POINTER(HighLevel), pi64, sz, INPUT_CB, OUTPUT_CB, vp])
        self._high_adjust = fn("high_adjust", u8, [C.POINTER(HighLevel), C.c_uint, u64, u64, u64])
        self._high_end = fn("high_end", u8, [C.POINTER(HighLevel), pi64, sz, OUTPUT_CB, vp])
        self._low_resample_i32 = fn("low_resample_i32", sz, [C.POINTER(LowLevel), pi64, sz, pi16, C.POINTER(sz), pi32, sz,
                                                            C.c_int, u64, C.POINTER(u8)])
        self._high_run_i32 = fn("high_run_i32", sz, [C.POINTER(HighLevel), pi64, sz, pi16, sz, sz, pi32, sz])
        if prefix == "oracle":
            self._count = fn("count_output_frames", u64, [C.POINTER(LowLevel), u64])
            self._hash = fn("stream_hash", u64, [pi32, sz, u64])
            self._noise = fn("fill_noise", u64, [pi16, sz, u64])
            self._mt = fn("low_resample_i32_mt", sz, [C.POINTER(LowLevel), pi64, sz, pi16, sz, pi32, C.c_uint])
        self._table = None

    # -- table / scalars ---------------------------------------------------
    def table(self):
        if self._table is None:
            n = self._table_len(self.radius)
            t = np.zeros(n, dtype=np.int64)
            self._precompute(_ptr(t, C.c_int64), self.radius)
            self._table = t
        return self._table

    def ratio(self, a, b):
        return self._ratio(a, b)

    def configure(self, in_rate, out_rate, lowpass, cfg=None):
        cfg = cfg if cfg is not None else Config()
        ok = self._configure(C.byref(cfg), self.radius, in_rate, out_rate, lowpass)
        return ok, cfg

    def low_init(self, channels, in_rate, out_rate, lowpass, st=None):
        st = st if st is not None else LowLevel()
        ok = self._low_init(C.byref(st), self.radius, channels, in_rate, out_rate, lowpass)
        return ok, st

    def low_adjust(self, st, in_rate, out_rate, lowpass):
        return self._low_adjust(C.byref(st), self.radius, in_rate, out_rate, lowpass)

    def high_init(self, channels, in_rate, out_rate, lowpass, st=None):
        st = st if st is not None else HighLevel()
        ok = self._high_init(C.byref(st), self.radius, channels, in_rate, out_rate, lowpass)
        return ok, st

    def high_adjust(self, st, in_rate, out_rate, lowpass):
        return self._high_adjust(C.byref(st), self.radius, in_rate, out_rate, lowpass)

    # -- frame / streams ---------------------------------------------------
    def frame(self, cfg, channels, padded, pos_int, pos_frac, accum=None):
        t = self.table()
        acc = np.zeros(channels, dtype=np.int64) if accum is None else np.array(accum, dtype=np.int64)
        padded = np.ascontiguousarray(padded, dtype=np.int16)
        self._frame(C.byref(cfg), _ptr(t, C.c_int64), len(t), _ptr(acc, C.c_int64), channels, _ptr(padded, C.c_int16), pos_int, pos_frac)
        return acc

    def low_resample_i32(self, st, padded, frames, capacity=None, norm_mode=NORM_CURRENT, legacy_gain=0, out=None):
        """Returns (out[:written*ch] int32, frames_left, ran_out_of_input)."""
        t = self.table()
        padded = np.ascontiguousarray(padded, dtype=np.int16)
        ch = st.channels
        if capacity is None:
            capacity = int(count_output_frames(st, frames)) + 1  # never reached: the input runs out first
        if out is None:
            out = np.empty(max(capacity, 1) * ch, dtype=np.int32)
        left = C.c_size_t(frames)
        ran_out = C.c_uint8(0)
        n = self._low_resample_i32(C.byref(st), _ptr(t, C.c_int64), len(t), _ptr(padded, C.c_int16), C.byref(left),
                                   _ptr(out, C.c_int32), capacity, norm_mode, legacy_gain, C.byref(ran_out))
        return out[: n * ch], left.value, ran_out.value

    def low_resample_cb(self, st, padded, frames, emit):
        """Callback form; emit(list_of_samples) -> truthy to continue.  Returns (ran_out_of_input, frames_left)."""
        t = self.table()
        padded = np.ascontiguousarray(padded, dtype=np.int16)

        def tramp(_user, frame, n):
            return 1 if emit([frame[i] for i in range(n)]) else 0

        cb = OUTPUT_CB(tramp)
        left = C.c_size_t(frames)
        r = self._low_resample(C.byref(st), _ptr(t, C.c_int64), len(t), _ptr(padded, C.c_int16), C.byref(left), cb, None)
        return r, left.value

    def high_run_i32(self, st, pcm, pull_chunk=0, capacity=None):
        t = self.table()
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        ch = st.low.channels
        frames = len(pcm) // ch
        if capacity is None:
            capacity = int(frames * 65536 // max(st.low.increment, 1)) + 64
        out = np.empty(max(capacity, 1) * ch, dtype=np.int32)
        n = self._high_run_i32(C.byref(st), _ptr(t, C.c_int64), len(t), _ptr(pcm, C.c_int16), frames, pull_chunk,
                               _ptr(out, C.c_int32), capacity)
        return out[: n * ch]

    def high_resample_cb(self, st, pull, emit):
        t = self.table()

        def tramp_in(_user, buf, n):
            data = pull(n)
            k = len(data) // st.low.channels
            for i, v in enumerate(data):
                buf[i] = v
            return k

        def tramp_out(_user, frame, n):
            return 1 if emit([frame[i] for i in range(n)]) else 0

        return self._high_resample(C.byref(st), _ptr(t, C.c_int64), len(t), INPUT_CB(tramp_in), OUTPUT_CB(tramp_out), None)

    def high_end_cb(self, st, emit):
        t = self.table()

        def tramp_out(_user, frame, n):
            return 1 if emit([frame[i] for i in range(n)]) else 0

        return self._high_end(C.byref(st), _ptr(t, C.c_int64), len(t), OUTPUT_CB(tramp_out), None)

    # -- oracle-only helpers ----------------------------------------------
    def low_resample_i32_mt(self, fresh, padded, frames, threads, out=None):
        t = self.table()
        padded = np.ascontiguousarray(padded, dtype=np.int16)
        cap = int(count_output_frames(fresh, frames))
        if out is None:
            out = np.empty(max(cap, 1) * fresh.channels, dtype=np.int32)
        n = self._mt(C.byref(fresh), _ptr(t, C.c_int64), len(t), _ptr(padded, C.c_int16), frames, _ptr(out, C.c_int32), threads)
        return out[: n * fresh.channels]


def count_output_frames(st, frames):
    """Closed form of SURVEY.md 8(a) a-2 (python ints: exact)."""
    start = st.pos_int * 65536 + st.pos_frac
    limit = frames * 65536
    if start >= limit:
        return 0
    return (limit - start + st.increment - 1) // st.increment


_ORACLE_LIB = os.path.join(ORACLE_DIR, "libcr_oracle.so")


def oracle(radius=3):
    if not os.path.exists(_ORACLE_LIB):
        build_checkers()
    return Checker(_ORACLE_LIB, "oracle", radius)


def reference(radius=3):
    """The real reference, compiled in place; None when neither /root/reference nor a prebuilt _ref exists."""
    path = os.path.join(ORACLE_DIR, "_ref", "libclownref_r%d.so" % radius)
    if not os.path.exists(path) and os.path.isdir(REFERENCE_DIR):
        build_checkers()
    if not os.path.exists(path):
        return None
    return Checker(path, "ref", radius)


# ---------------------------------------------------------------------------
# data helpers shared by tests, bench and the golden generator
# ---------------------------------------------------------------------------
NOISE_SEED = 0x9E3779B97F4A7C15


def noise_pcm(samples, seed=NOISE_SEED):
    """xorshift64 white noise of SURVEY.md 8(d) (C loop in the oracle library: fast)."""
    lib = C.CDLL(_ORACLE_LIB) if os.path.exists(_ORACLE_LIB) else oracle().lib
    f = lib.oracle_fill_noise
    f.restype, f.argtypes = C.c_uint64, [C.POINTER(C.c_int16), C.c_size_t, C.c_uint64]
    a = np.empty(samples, dtype=np.int16)
    f(_ptr(a, C.c_int16), samples, seed)
    return a


def pad_frames(pcm, channels, radius_frames):
    """Zero halo of `radius_frames` frames each side (tests/test-low-level.c:133-152)."""
    pcm = np.asarray(pcm, dtype=np.int16)
    z = np.zeros(radius_frames * channels, dtype=np.int16)
    return np.concatenate([z, pcm, z])


def stream_hash(samples, seed=0):
    lib = C.CDLL(_ORACLE_LIB)
    f = lib.oracle_stream_hash
    f.restype, f.argtypes = C.c_uint64, [C.POINTER(C.c_int32), C.c_size_t, C.c_uint64]
    a = np.ascontiguousarray(samples, dtype=np.int32)
    return f(_ptr(a, C.c_int32), a.size, seed)


def sha256_i32(samples):
    return hashlib.sha256(np.ascontiguousarray(samples, dtype="<i4").tobytes()).hexdigest()
