"""Adapter that gives the product library (through clownresampler_amd's ctypes mirror of the C ABI) the `Checker`
interface of tests/_checkers.py, so the same case runner (tests/_cases.py) drives oracle, reference and product."""
import numpy as np

import clownresampler_amd as cr


class _Cfg:
    def __init__(self, ll):
        self._ll = ll

    stretched_radius = property(lambda s: s._ll.stretched_kernel_radius)
    radius_frames = property(lambda s: s._ll.integer_stretched_kernel_radius)
    radius_delta = property(lambda s: s._ll.stretched_kernel_radius_delta)
    table_step = property(lambda s: s._ll.kernel_step_size)

    def astuple(self):
        return (self.stretched_radius, self.radius_frames, self.radius_delta, self.table_step)


class LowView:
    """oracle-style field names over a ClownResampler_LowLevel_State"""

    def __init__(self, raw):
        self.raw = raw
        self.cfg = _Cfg(raw.lowest_level)

    channels = property(lambda s: s.raw.channels)
    pos_int = property(lambda s: s.raw.position_integer)
    pos_frac = property(lambda s: s.raw.position_fractional)
    increment = property(lambda s: s.raw.increment)

    def astuple(self):
        return self.raw.astuple()


class HighView:
    def __init__(self, raw):
        self.raw = raw
        self.low = LowView(raw.low_level)

    lead_needed = property(lambda s: s.raw.leading_padding_frames_needed)
    trail_left = property(lambda s: s.raw.trailing_padding_frames_remaining)
    max_radius_frames = property(lambda s: s.raw.maximum_integer_stretched_kernel_radius)


class Product:
    def __init__(self, radius=3, abi="c89"):
        # abi "c99": libclownresampler_amd_c99.so, the CC_USE_C99_INTEGERS build (reference clownresampler.h:483-501)
        self.api = cr.load(radius, abi)
        self.radius = radius
        self.abi = abi
        self.pre = self.api.precomputed()

    def table(self):
        return np.array(self.pre.lanczos_kernel_table[:], dtype=np.int64)

    def ratio(self, a, b):
        st = self.api.LowLevel_State()
        self.api.LowLevel_Adjust(st, a, b, a)
        return st.increment

    def configure(self, i, o, l, cfg=None):
        cfg = cfg if cfg is not None else self.api.LowestLevel_Configuration()
        ok = self.api.LowestLevel_Configure(cfg, i, o, l)
        return ok, cfg

    def low_init(self, ch, i, o, l, st=None):
        raw = st.raw if st is not None else self.api.LowLevel_State()
        ok = self.api.LowLevel_Init(raw, ch, i, o, l)
        return ok, LowView(raw)

    def low_adjust(self, st, i, o, l):
        return self.api.LowLevel_Adjust(st.raw, i, o, l)

    def high_init(self, ch, i, o, l, st=None):
        raw = st.raw if st is not None else self.api.HighLevel_State()
        ok = self.api.HighLevel_Init(raw, ch, i, o, l)
        return ok, HighView(raw)

    def high_adjust(self, st, i, o, l):
        return self.api.HighLevel_Adjust(st.raw, i, o, l)

    def frame(self, cfg, channels, padded, pos_int, pos_frac, accum=None):
        acc = [0] * channels if accum is None else list(accum)
        return np.array(self.api.LowestLevel_Resample(cfg, self.pre, acc, channels, padded, pos_int, pos_frac), dtype=np.int64)

    def low_resample_i32(self, st, padded, frames, capacity=None, **_kw):
        return self.api.LowLevel_ResampleBulk(st.raw, self.pre, padded, frames, capacity)

    def low_resample_cb(self, st, padded, frames, emit):
        r, left = self.api.LowLevel_Resample(st.raw, self.pre, padded, frames, emit)
        return int(r), left

    def high_resample_cb(self, st, pull, emit):
        return self.api.HighLevel_Resample(st.raw, self.pre, pull, emit)

    def high_end_cb(self, st, emit):
        return self.api.HighLevel_ResampleEnd(st.raw, self.pre, emit)

    def high_run_i32(self, st, pcm, pull_chunk=0, capacity=None):
        """tests/test-high-level.c:126-127 through the callback ABI: Resample until the input dries up, then ResampleEnd."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        ch = st.low.channels
        pos = [0]
        out = []

        def pull(n):
            k = min(n, len(pcm) // ch - pos[0])
            if pull_chunk:
                k = min(k, pull_chunk)
            a = pcm[pos[0] * ch:(pos[0] + k) * ch]
            pos[0] += k
            return a

        def emit(frame):
            out.extend(frame)
            return True

        if self.api.HighLevel_Resample(st.raw, self.pre, pull, emit):
            self.api.HighLevel_ResampleEnd(st.raw, self.pre, emit)
        return np.array(out, dtype=np.int64).astype(np.int32)
