"""clownresampler_amd - Python host-side mirror of the clownresampler C API over libclownresampler_amd.so.

The product is the C-ABI shared library (include/clownresampler.h + include/clownresampler_amd.h); this module
only binds it with ctypes, one-to-one, under the reference's own names (ClownResampler_LowLevel_Init ->
`Api.LowLevel_Init`, ...), so that the parity tests read like the reference's harnesses
(tests/test-low-level.c, tests/test-high-level.c).  There is no Python or CPU implementation of the resampling
arithmetic here: if the library (and therefore the HIP kernels) cannot be loaded, importing `load()` raises.

    import clownresampler_amd as cr
    api = cr.load(radius=3)
    pre = api.Precomputed(); api.Precompute(pre)
    st = api.LowLevel_State(); api.LowLevel_Init(st, 2, 44100, 48000, 44100)
    out, left, ran_out = api.LowLevel_ResampleBulk(st, pre, padded_int16, frames)
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CLOWNRESAMPLER_AMD_LIBRARY: development hook - A/B timing of two builds on one box, tools/ab.sh)
LIB_PATH = os.environ.get("CLOWNRESAMPLER_AMD_LIBRARY") or os.path.join(_HERE, "libclownresampler_amd.so")
SUPPORTED_RADII = (3, 5, 8)   # csrc/Makefile RADII (ClownResamplerAMD_BuiltRadii() says what the loaded library really has)

KERNEL_RESOLUTION = 0x400      # CLOWNRESAMPLER_KERNEL_RESOLUTION, reference clownresampler.h:452-454
MAXIMUM_CHANNELS = 16          # CLOWNRESAMPLER_MAXIMUM_CHANNELS, reference clownresampler.h:458-460

# error codes of include/clownresampler_amd.h
OK, ERROR_NO_DEVICE, ERROR_HIP, ERROR_ARGUMENT, ERROR_PLAN_MISMATCH = 0, 1, 2, 3, 4


class ClownResamplerError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("clownresampler_amd error %d: %s" % (code, message))
        self.code = code
        self.message = message


def build_native(verbose=False):
    """Compiles the HIP kernels (hipcc --offload-arch=gfx950) and the host C code into LIB_PATH, in-tree."""
    jobs = max(1, min(8, os.cpu_count() or 1))   # the kernel instances are spread over several units that compile side by side
    cmd = ["make", "-j%d" % jobs, "-C", os.path.join(_HERE, "csrc")] + ([] if verbose else ["-s"])
    subprocess.run(cmd, check=True)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("build did not produce " + LIB_PATH)


# ---- the reference's POD structs, per integer ABI on LP64 (reference clownresampler.h:483-501 / :546-560, :627-659) ----
# "c89": the reference's default types (8-byte `long` table entries, `unsigned int` channel counts) - libclownresampler_amd.so
# "c99": CC_USE_C99_INTEGERS (int_least32_t table entries = 4 bytes, uint_fast8_t channel counts = 1 byte on glibc x86-64;
#        int_fast32_t / uint_fast32_t stay 8 bytes) - libclownresampler_amd_c99.so
ABIS = ("c89", "c99")


class _Types:
    pass


def _make_types(abi):
    T = _Types()
    T.abi = abi
    T.cc_s16l, T.cc_s32f, T.cc_u32f, T.cc_bool = C.c_short, C.c_long, C.c_ulong, C.c_ubyte
    if abi == "c89":
        T.cc_s32l, T.cc_u8f = C.c_long, C.c_uint
    elif abi == "c99":
        T.cc_s32l, T.cc_u8f = C.c_int32, C.c_uint8
    else:
        raise ValueError("abi is one of %s" % (ABIS,))

    class LowestLevel_Configuration(C.Structure):
        _fields_ = [("stretched_kernel_radius", C.c_size_t), ("integer_stretched_kernel_radius", C.c_size_t),
                    ("stretched_kernel_radius_delta", C.c_size_t), ("kernel_step_size", C.c_size_t)]

    class LowLevel_State(C.Structure):
        _fields_ = [("lowest_level", LowestLevel_Configuration), ("channels", T.cc_u8f), ("position_integer", C.c_size_t),
                    ("position_fractional", T.cc_u32f), ("increment", T.cc_u32f)]

        def astuple(self):
            ll = self.lowest_level
            return (ll.stretched_kernel_radius, ll.integer_stretched_kernel_radius, ll.stretched_kernel_radius_delta, ll.kernel_step_size,
                    self.channels, self.position_integer, self.position_fractional, self.increment)

    class HighLevel_State(C.Structure):
        _fields_ = [("low_level", LowLevel_State), ("input_buffer", T.cc_s16l * 0x1000), ("input_buffer_start", C.c_void_p),
                    ("input_buffer_end", C.c_void_p), ("maximum_integer_stretched_kernel_radius", C.c_size_t),
                    ("leading_padding_frames_needed", C.c_size_t), ("trailing_padding_frames_remaining", C.c_size_t)]

    class Shard(C.Structure):  # ClownResamplerAMD_Shard
        _fields_ = [("first_output_frame", C.c_size_t), ("output_frames", C.c_size_t), ("first_input_frame", C.c_size_t),
                    ("input_frames", C.c_size_t), ("halo_frames", C.c_size_t), ("state", LowLevel_State)]

    class Segment(C.Structure):  # ClownResamplerAMD_Segment
        _fields_ = [("input_frames", C.c_size_t), ("input_sample_rate", T.cc_u32f), ("output_sample_rate", T.cc_u32f), ("low_pass_filter_sample_rate", T.cc_u32f)]

    T.LowestLevel_Configuration, T.LowLevel_State, T.HighLevel_State, T.Shard, T.Segment = LowestLevel_Configuration, LowLevel_State, HighLevel_State, Shard, Segment
    T.InputCallback = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.POINTER(T.cc_s16l), C.c_size_t)             # reference clownresampler.h:661
    T.OutputCallback = C.CFUNCTYPE(T.cc_bool, C.c_void_p, C.POINTER(T.cc_s32f), T.cc_u8f)                # reference clownresampler.h:662
    return T


_TYPES = {abi: _make_types(abi) for abi in ABIS}
# the default ABI's types under their plain names (what every client of the default library uses)
_T89 = _TYPES["c89"]
cc_s16l, cc_s32l, cc_s32f, cc_u32f, cc_u8f, cc_bool = _T89.cc_s16l, _T89.cc_s32l, _T89.cc_s32f, _T89.cc_u32f, _T89.cc_u8f, _T89.cc_bool
LowestLevel_Configuration, LowLevel_State, HighLevel_State = _T89.LowestLevel_Configuration, _T89.LowLevel_State, _T89.HighLevel_State
Shard, Segment, InputCallback, OutputCallback = _T89.Shard, _T89.Segment, _T89.InputCallback, _T89.OutputCallback


class DeviceShard(C.Structure):  # ClownResamplerAMD_DeviceShard
    _fields_ = [("device", C.c_int), ("device_input", C.c_void_p), ("device_output", C.c_void_p), ("hip_stream", C.c_void_p)]


GATHER_NONE, GATHER_PEER_COPY, GATHER_RCCL = 0, 1, 2


class PlanInfo(C.Structure):  # ClownResamplerAMD_PlanInfo
    _fields_ = [(n, C.c_uint32) for n in ("kernel", "channels", "slots", "first_slot", "rows", "row_stride", "row_mode", "threads",
                                          "tile_frames", "lds_bytes", "max_blocks", "specialised", "variant", "norm_mode",
                                          "brief_kernel", "brief_variant")] + [("brief_below", C.c_uint64)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


ErrorHandler = C.CFUNCTYPE(None, C.c_int, C.c_char_p, C.c_void_p)

LIB_PATH_C99 = os.environ.get("CLOWNRESAMPLER_AMD_LIBRARY_C99") or os.path.join(_HERE, "libclownresampler_amd_c99.so")
_libs = {}
_handler_keepalive = {}


def _load_library(abi="c89"):
    if abi in _libs:
        return _libs[abi]
    path = LIB_PATH if abi == "c89" else LIB_PATH_C99
    if not os.path.exists(path):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                          "There is no fallback implementation." % path)
    lib = C.CDLL(path)

    # Errors surface as Python exceptions instead of abort(): the handler records, the wrappers raise.
    def _on_error(code, message, _user):
        pass  # the code/message are kept per thread by the library; wrappers poll them

    _handler_keepalive[abi] = ErrorHandler(_on_error)
    lib.ClownResamplerAMD_SetErrorHandler.argtypes = [ErrorHandler, C.c_void_p]
    lib.ClownResamplerAMD_SetErrorHandler(_handler_keepalive[abi], None)
    lib.ClownResamplerAMD_LastErrorCode.restype = C.c_int
    lib.ClownResamplerAMD_LastErrorMessage.restype = C.c_char_p
    lib.ClownResamplerAMD_DebugDumpFlightRecorder.argtypes = [C.c_int]
    lib.ClownResamplerAMD_DebugDumpFlightRecorder.restype = None
    lib.ClownResamplerAMD_DebugInstallAbortDump.restype = C.c_int
    lib.ClownResamplerAMD_DebugSelfCheck.argtypes = [C.c_char_p, C.c_size_t]
    lib.ClownResamplerAMD_DebugSelfCheck.restype = C.c_int
    _libs[abi] = lib
    if _abort_dump["on"]:
        lib.ClownResamplerAMD_DebugInstallAbortDump()
    return lib


_abort_dump = {"on": os.environ.get("CLOWNRESAMPLER_AMD_ABORT_DUMP", "") not in ("", "0")}


def install_abort_dump():
    """From now on a SIGABRT of this process (the HIP runtime's answer to a GPU memory fault) first writes the library's flight
    recorder - the last 64 launches / allocations with their address ranges - to stderr, then goes on to the handler that was
    installed before (ClownResamplerAMD_DebugInstallAbortDump); for the libraries loaded so far and any loaded later."""
    _abort_dump["on"] = True
    for lib in _libs.values():
        lib.ClownResamplerAMD_DebugInstallAbortDump()


def self_check():
    """ClownResamplerAMD_DebugSelfCheck on every library this process has loaded: [] when all are at rest, else [(abi, findings, message)]."""
    bad = []
    for abi, lib in _libs.items():
        buf = C.create_string_buffer(2048)
        n = lib.ClownResamplerAMD_DebugSelfCheck(buf, len(buf))
        if n != 0:
            bad.append((abi, n, buf.value.decode("utf-8", "replace")))
    return bad


def dump_flight_recorder(fd=2):
    for lib in _libs.values():
        lib.ClownResamplerAMD_DebugDumpFlightRecorder(fd)


def _raise_if_failed(lib):
    code = lib.ClownResamplerAMD_LastErrorCode()
    if code != 0:
        message = lib.ClownResamplerAMD_LastErrorMessage().decode("utf-8", "replace")
        lib.ClownResamplerAMD_ClearError()
        raise ClownResamplerError(code, message)


def _as_i16(a):
    a = np.ascontiguousarray(a, dtype=np.int16)
    return a, a.ctypes.data_as(C.POINTER(cc_s16l))


class Api:
    """The API instance for one CLOWNRESAMPLER_KERNEL_RADIUS (symbols ..._R<radius> for radius != 3) of one integer ABI
    ("c89": libclownresampler_amd.so, "c99": the CC_USE_C99_INTEGERS build, libclownresampler_amd_c99.so)."""

    def __init__(self, radius=3, abi="c89"):
        if radius not in SUPPORTED_RADII:
            raise ValueError("library built for radii %s" % (SUPPORTED_RADII,))
        self.radius = radius
        self.abi = abi
        self.T = T = _TYPES[abi]
        self.lib = lib = _load_library(abi)
        self.table_len = radius * 2 * KERNEL_RESOLUTION
        cc_s16l, cc_s32l, cc_s32f, cc_u32f, cc_u8f, cc_bool = T.cc_s16l, T.cc_s32l, T.cc_s32f, T.cc_u32f, T.cc_u8f, T.cc_bool
        LowestLevel_Configuration, LowLevel_State, HighLevel_State = T.LowestLevel_Configuration, T.LowLevel_State, T.HighLevel_State
        Shard, Segment, InputCallback, OutputCallback = T.Shard, T.Segment, T.InputCallback, T.OutputCallback

        class Precomputed(C.Structure):  # reference clownresampler.h:627-630
            _fields_ = [("lanczos_kernel_table", cc_s32l * self.table_len)]

        self.Precomputed = Precomputed
        self.LowLevel_State = LowLevel_State
        self.HighLevel_State = HighLevel_State
        self.LowestLevel_Configuration = LowestLevel_Configuration

        sfx = "" if radius == 3 else "_R%d" % radius
        P = C.POINTER

        def fn(name, res, args, suffixed=True):
            f = getattr(lib, name + (sfx if suffixed else ""))
            f.restype, f.argtypes = res, args
            return f

        self._Precompute = fn("ClownResampler_Precompute", None, [P(Precomputed)])
        self._Configure = fn("ClownResampler_LowestLevel_Configure", cc_bool, [P(LowestLevel_Configuration), cc_u32f, cc_u32f, cc_u32f])
        self._LowestResample = fn("ClownResampler_LowestLevel_Resample", None, [P(LowestLevel_Configuration), P(Precomputed), P(cc_s32f), cc_u8f, P(cc_s16l), C.c_size_t, cc_u32f])
        self._LowInit = fn("ClownResampler_LowLevel_Init", cc_bool, [P(LowLevel_State), cc_u8f, cc_u32f, cc_u32f, cc_u32f])
        self._LowAdjust = fn("ClownResampler_LowLevel_Adjust", cc_bool, [P(LowLevel_State), cc_u32f, cc_u32f, cc_u32f])
        self._LowResample = fn("ClownResampler_LowLevel_Resample", cc_bool, [P(LowLevel_State), P(Precomputed), P(cc_s16l), P(C.c_size_t), OutputCallback, C.c_void_p])
        self._HighInit = fn("ClownResampler_HighLevel_Init", cc_bool, [P(HighLevel_State), cc_u8f, cc_u32f, cc_u32f, cc_u32f])
        self._HighResample = fn("ClownResampler_HighLevel_Resample", cc_bool, [P(HighLevel_State), P(Precomputed), InputCallback, OutputCallback, C.c_void_p])
        self._HighAdjust = fn("ClownResampler_HighLevel_Adjust", cc_bool, [P(HighLevel_State), cc_u32f, cc_u32f, cc_u32f])
        self._HighEnd = fn("ClownResampler_HighLevel_ResampleEnd", cc_bool, [P(HighLevel_State), P(Precomputed), OutputCallback, C.c_void_p])
        self._Bulk = fn("ClownResampler_LowLevel_ResampleBulk", C.c_size_t, [P(LowLevel_State), P(Precomputed), P(cc_s16l), P(C.c_size_t), P(C.c_int32), C.c_size_t, P(cc_bool)])
        self._BulkS16 = fn("ClownResampler_LowLevel_ResampleBulkS16", C.c_size_t, [P(LowLevel_State), P(Precomputed), P(cc_s16l), P(C.c_size_t), P(C.c_int16), C.c_size_t, P(cc_bool)])
        self._PlanCreate = fn("ClownResamplerAMD_PlanCreate", C.c_void_p, [P(LowLevel_State), P(Precomputed)])
        self._BuildRows = fn("ClownResamplerAMD_BuildRows", C.c_int, [P(LowestLevel_Configuration), P(Precomputed), P(PlanInfo), P(P(C.c_int32)), P(C.c_int), P(C.c_char_p), P(C.c_uint32)])
        # radius-independent
        self._PlanGetInfo = fn("ClownResamplerAMD_PlanGetInfo", None, [C.c_void_p, P(PlanInfo)], False)
        self._PlanRows = fn("ClownResamplerAMD_PlanRows", P(C.c_int32), [C.c_void_p], False)
        self._PeriodicShape = fn("ClownResamplerAMD_PeriodicShape", C.c_int, [P(LowestLevel_Configuration), P(Precomputed), C.c_uint64, P(C.c_uint32), P(C.c_uint32),
                                                                                P(C.c_uint32), C.c_uint32 * 4, P(C.c_uint64), P(C.c_uint64), P(C.c_uint64)])
        self._PlanRowOf = fn("ClownResamplerAMD_PlanRowOf", C.c_uint32, [C.c_void_p, C.c_uint32], False)
        self._ResampleDevice = fn("ClownResamplerAMD_ResampleDevice", C.c_size_t, [C.c_void_p, P(LowLevel_State), C.c_void_p, P(C.c_size_t), C.c_void_p, C.c_size_t, C.c_void_p, P(cc_bool)], False)
        self._ResampleDeviceS16 = fn("ClownResamplerAMD_ResampleDeviceS16", C.c_size_t, [C.c_void_p, P(LowLevel_State), C.c_void_p, P(C.c_size_t), C.c_void_p, C.c_size_t, C.c_void_p, P(cc_bool)], False)
        self._SetPlanCacheLimit = fn("ClownResamplerAMD_SetPlanCacheLimit", None, [C.c_size_t], False)
        self._PlanCacheCount = fn("ClownResamplerAMD_PlanCacheCount", C.c_size_t, [], False)
        self._ResampleSegmentsDevice = fn("ClownResamplerAMD_ResampleSegmentsDevice", C.c_size_t, [P(LowLevel_State), P(self.Precomputed), C.c_void_p, C.c_size_t, P(Segment), C.c_size_t,
                                                                                                      C.c_void_p, C.c_size_t, C.c_int, P(C.c_size_t), C.c_void_p])
        self._ResampleSharded = fn("ClownResamplerAMD_ResampleShardedDevice", C.c_size_t, [P(LowLevel_State), P(self.Precomputed), C.c_size_t, P(DeviceShard), C.c_uint, C.c_int,
                                                                                              C.c_int, C.c_uint, C.c_void_p])
        self._ShardedSync = fn("ClownResamplerAMD_ShardedSynchronize", C.c_int, [P(DeviceShard), C.c_uint], False)
        self._SetThreadDevice = fn("ClownResamplerAMD_SetThreadDevice", C.c_int, [C.c_int], False)
        self._GetDevice = fn("ClownResamplerAMD_GetDevice", C.c_int, [], False)
        self._ReserveCapture = fn("ClownResamplerAMD_ReserveCaptureLaunches", C.c_int, [C.c_size_t], False)
        self._ReleaseCaptured = fn("ClownResamplerAMD_ReleaseCapturedLaunches", C.c_int, [], False)
        self._PlanKernelAt = fn("ClownResamplerAMD_PlanKernelAt", C.c_uint32, [C.c_void_p, C.c_uint32], False)
        self._LaunchCount = fn("ClownResamplerAMD_DebugLaunchCount", C.c_ulonglong, [C.c_uint], False)
        self._BuildId = fn("ClownResamplerAMD_BuildId", C.c_char_p, [], False)
        self._DisableInt = fn("ClownResamplerAMD_DebugDisableIntKernel", None, [C.c_int], False)
        self._DisableDual = fn("ClownResamplerAMD_DebugDisableDualMono", None, [C.c_int], False)
        self._SegKernel = fn("ClownResamplerAMD_DebugSegKernel", None, [C.c_int], False)
        self._PlanDualMono = fn("ClownResamplerAMD_PlanDualMonoKernel", C.c_uint32, [C.c_void_p], False)
        self._PlanPadded = fn("ClownResamplerAMD_PlanPaddedTiles", C.c_uint32, [C.c_void_p], False)
        self._PlanSeg = fn("ClownResamplerAMD_PlanSegKernel", C.c_uint32, [C.c_void_p], False)
        self._SegmentsMode = fn("ClownResamplerAMD_DebugSegmentsMode", None, [C.c_int], False)
        self._HighRelease = fn("ClownResamplerAMD_HighLevel_Release", None, [P(HighLevel_State)], False)
        self._WindowCount = fn("ClownResamplerAMD_StreamingWindowCount", C.c_size_t, [], False)
        self._DeviceAllocOn = fn("ClownResamplerAMD_DeviceAllocOn", C.c_void_p, [C.c_int, C.c_size_t], False)
        self._Count = fn("ClownResamplerAMD_CountOutputFrames", C.c_size_t, [P(LowLevel_State), C.c_size_t], False)
        self._Advance = fn("ClownResamplerAMD_AdvanceState", None, [P(LowLevel_State), C.c_size_t], False)
        self._PlanShard = fn("ClownResamplerAMD_PlanShard", C.c_int, [P(LowLevel_State), C.c_size_t, C.c_uint, C.c_uint, P(Shard)], False)
        self._DeviceCount = fn("ClownResamplerAMD_DeviceCount", C.c_int, [], False)
        self._IsUsable = fn("ClownResamplerAMD_IsUsable", C.c_int, [], False)
        self._SetDevice = fn("ClownResamplerAMD_SetDevice", C.c_int, [C.c_int], False)
        self._Shutdown = fn("ClownResamplerAMD_Shutdown", None, [], False)
        self._DeviceAlloc = fn("ClownResamplerAMD_DeviceAlloc", C.c_void_p, [C.c_size_t], False)
        self._DeviceFree = fn("ClownResamplerAMD_DeviceFree", None, [C.c_void_p], False)
        self._ToDevice = fn("ClownResamplerAMD_CopyToDevice", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t], False)
        self._FromDevice = fn("ClownResamplerAMD_CopyFromDevice", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t], False)
        self._StreamSync = fn("ClownResamplerAMD_StreamSynchronize", C.c_int, [C.c_void_p], False)
        self._ForceGeneric = fn("ClownResamplerAMD_DebugForceGenericKernel", None, [C.c_int], False)
        self._SetVariant = fn("ClownResamplerAMD_DebugSetVariant", None, [C.c_int], False)
        self._SetStreamingWindow = fn("ClownResamplerAMD_SetStreamingWindow", None, [C.c_size_t], False)
        self._HostVisible = fn("ClownResamplerAMD_DebugHostIsDeviceVisible", C.c_int, [C.c_void_p, C.c_size_t], False)
        self._libc_free = C.CDLL(None).free
        self._libc_free.argtypes = [C.c_void_p]

    # ---- reference API, same names and argument meaning ----
    def Precompute(self, precomputed):
        self._Precompute(C.byref(precomputed))

    def precomputed(self):
        p = self.Precomputed()
        self.Precompute(p)
        return p

    def LowestLevel_Configure(self, configuration, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate):
        return bool(self._Configure(C.byref(configuration), input_sample_rate, output_sample_rate, low_pass_filter_sample_rate))

    def LowestLevel_Resample(self, configuration, precomputed, output_frame, channels, input_buffer, position_integer, position_fractional):
        """output_frame: sequence of `channels` ints, accumulated into (reference clownresampler.h:1020,1033). Returns the new frame."""
        frame = (self.T.cc_s32f * channels)(*[int(v) for v in output_frame])
        keep, ptr = _as_i16(input_buffer)
        self._LowestResample(C.byref(configuration), C.byref(precomputed), frame, channels, ptr, position_integer, position_fractional)
        _raise_if_failed(self.lib)
        return [int(v) for v in frame]

    def LowLevel_Init(self, resampler, channels, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate):
        return bool(self._LowInit(C.byref(resampler), channels, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate))

    def LowLevel_Adjust(self, resampler, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate):
        return bool(self._LowAdjust(C.byref(resampler), input_sample_rate, output_sample_rate, low_pass_filter_sample_rate))

    def LowLevel_Resample(self, resampler, precomputed, input_buffer, total_input_frames, output_callback):
        """output_callback(list_of_samples) -> truthy to continue.  Returns (ran_out_of_input, frames_not_processed)."""
        keep, ptr = _as_i16(input_buffer)

        def tramp(_user, frame, n):
            return 1 if output_callback([frame[i] for i in range(n)]) else 0

        left = C.c_size_t(total_input_frames)
        r = self._LowResample(C.byref(resampler), C.byref(precomputed), ptr, C.byref(left), self.T.OutputCallback(tramp), None)
        _raise_if_failed(self.lib)
        return bool(r), left.value

    def HighLevel_Init(self, resampler, channels, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate):
        return bool(self._HighInit(C.byref(resampler), channels, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate))

    def HighLevel_Adjust(self, resampler, input_sample_rate, output_sample_rate, low_pass_filter_sample_rate):
        return bool(self._HighAdjust(C.byref(resampler), input_sample_rate, output_sample_rate, low_pass_filter_sample_rate))

    def HighLevel_Resample(self, resampler, precomputed, input_callback, output_callback):
        """input_callback(max_frames) -> int16 array of whole frames (empty = end); output_callback(list) -> truthy to continue."""
        ch = resampler.low_level.channels

        def tramp_in(_user, buf, n):
            data = np.asarray(input_callback(n), dtype=np.int16).reshape(-1)
            C.memmove(buf, data.ctypes.data, data.size * 2)
            return data.size // ch

        def tramp_out(_user, frame, n):
            return 1 if output_callback([frame[i] for i in range(n)]) else 0

        r = self._HighResample(C.byref(resampler), C.byref(precomputed), self.T.InputCallback(tramp_in), self.T.OutputCallback(tramp_out), None)
        _raise_if_failed(self.lib)
        return bool(r)

    def HighLevel_ResampleEnd(self, resampler, precomputed, output_callback):
        def tramp_out(_user, frame, n):
            return 1 if output_callback([frame[i] for i in range(n)]) else 0

        r = self._HighEnd(C.byref(resampler), C.byref(precomputed), self.T.OutputCallback(tramp_out), None)
        _raise_if_failed(self.lib)
        return bool(r)

    # ---- extension (include/clownresampler_amd.h) ----
    def LowLevel_ResampleBulk(self, resampler, precomputed, input_buffer, total_input_frames, output_capacity_frames=None, output=None):
        """Returns (int32 array of written samples, frames_not_processed, ran_out_of_input)."""
        keep, ptr = _as_i16(input_buffer)
        ch = resampler.channels
        if output_capacity_frames is None:
            output_capacity_frames = self.CountOutputFrames(resampler, total_input_frames) + 1
        if output is None:
            output = np.empty(max(output_capacity_frames, 1) * ch, dtype=np.int32)
        left = C.c_size_t(total_input_frames)
        ran_out = cc_bool(0)
        n = self._Bulk(C.byref(resampler), C.byref(precomputed), ptr, C.byref(left), output.ctypes.data_as(C.POINTER(C.c_int32)),
                       output_capacity_frames, C.byref(ran_out))
        _raise_if_failed(self.lib)
        return output[: n * ch], left.value, int(ran_out.value)

    def LowLevel_ResampleBulkS16(self, resampler, precomputed, input_buffer, total_input_frames, output_capacity_frames=None, output=None):
        """Clamped int16 output (examples/low-level.c:69-80 fused in).  Returns (int16 array, frames_not_processed, ran_out_of_input)."""
        keep, ptr = _as_i16(input_buffer)
        ch = resampler.channels
        if output_capacity_frames is None:
            output_capacity_frames = self.CountOutputFrames(resampler, total_input_frames) + 1
        if output is None:
            output = np.empty(max(output_capacity_frames, 1) * ch, dtype=np.int16)
        left = C.c_size_t(total_input_frames)
        ran_out = cc_bool(0)
        n = self._BulkS16(C.byref(resampler), C.byref(precomputed), ptr, C.byref(left), output.ctypes.data_as(C.POINTER(C.c_int16)),
                          output_capacity_frames, C.byref(ran_out))
        _raise_if_failed(self.lib)
        return output[: n * ch], left.value, int(ran_out.value)

    def CountOutputFrames(self, state, total_input_frames):
        return self._Count(C.byref(state), total_input_frames)

    def AdvanceState(self, state, frames):
        self._Advance(C.byref(state), frames)

    def PlanShard(self, state, total_input_frames, shard, shard_count):
        s = self.T.Shard()
        if self._PlanShard(C.byref(state), total_input_frames, shard, shard_count, C.byref(s)) != 0:
            raise ValueError("bad shard index")
        return s

    def BuildRows(self, configuration, precomputed, row_map=None):
        """Host-only polyphase rows (no GPU needed): returns (PlanInfo, rows ndarray [rows, row_stride], eligible, reason).
        row_map: optional uint32[65536] array that receives the row index of every fractional position."""
        info, rows, eligible, reason = PlanInfo(), C.POINTER(C.c_int32)(), C.c_int(0), C.c_char_p()
        rm = row_map.ctypes.data_as(C.POINTER(C.c_uint32)) if row_map is not None else None
        r = self._BuildRows(C.byref(configuration), C.byref(precomputed), C.byref(info), C.byref(rows), C.byref(eligible), C.byref(reason), rm)
        msg = (reason.value or b"").decode()
        if r != 0 or not rows:
            if rows:
                self._libc_free(rows)
            return info, None, False, msg
        arr = np.ctypeslib.as_array(rows, shape=(info.rows, info.row_stride)).copy()
        self._libc_free(rows)
        return info, arr, bool(eligible.value), msg

    def PeriodicShape(self, configuration, precomputed, increment):
        """Host-only: what a k_int instance for this configuration / increment is compiled for (tools/int_shapes.py), or None:
        dict(period, ratio, slots, starts, negmask, safemask, zeromask) at fractional position 0."""
        period, ratio, slots = C.c_uint32(), C.c_uint32(), C.c_uint32()
        starts = (C.c_uint32 * 4)()
        neg, safe, zero = C.c_uint64(), C.c_uint64(), C.c_uint64()
        if self._PeriodicShape(C.byref(configuration), C.byref(precomputed), increment, C.byref(period), C.byref(ratio), C.byref(slots), starts,
                               C.byref(neg), C.byref(safe), C.byref(zero)) != 0:
            return None
        return dict(period=period.value, ratio=ratio.value, slots=slots.value, starts=list(starts)[:period.value], negmask=neg.value,
                    safemask=safe.value, zeromask=zero.value)

    def PlanCreate(self, state, precomputed):
        plan = self._PlanCreate(C.byref(state), C.byref(precomputed))
        _raise_if_failed(self.lib)
        if not plan:
            raise ClownResamplerError(-1, "PlanCreate returned NULL")
        return plan

    def PlanGetInfo(self, plan):
        info = PlanInfo()
        self._PlanGetInfo(plan, C.byref(info))
        return info

    def PlanRows(self, plan):
        info = self.PlanGetInfo(plan)
        return np.ctypeslib.as_array(self._PlanRows(plan), shape=(info.rows, info.row_stride)).copy()

    def PlanRowOf(self, plan, position_fractional):
        return self._PlanRowOf(plan, position_fractional)

    def ResampleDevice(self, plan, resampler, device_input, total_input_frames, device_output, output_capacity_frames, hip_stream=None, s16=False):
        """device_input / device_output: integer device addresses (e.g. torch_tensor.data_ptr()).  Enqueues on hip_stream and
        returns (frames, frames_not_processed, ran_out_of_input) without synchronising."""
        left = C.c_size_t(total_input_frames)
        ran_out = cc_bool(0)
        n = (self._ResampleDeviceS16 if s16 else self._ResampleDevice)(plan, C.byref(resampler), C.c_void_p(device_input), C.byref(left), C.c_void_p(device_output),
                                 output_capacity_frames, C.c_void_p(hip_stream or 0), C.byref(ran_out))
        _raise_if_failed(self.lib)
        return n, left.value, int(ran_out.value)

    def ResampleSegmentsDevice(self, resampler, precomputed, device_timeline, halo_frames, segments, device_output, output_capacity_frames, hip_stream=None, s16=False):
        """Variable rate on the device (ClownResamplerAMD_ResampleSegmentsDevice).  segments: [(input_frames, in_rate, out_rate, low_pass), ...];
        device_timeline: device address of input frame 0.  Returns (total_frames, [frames per segment]); not synchronised."""
        array = (self.T.Segment * max(1, len(segments)))(*[self.T.Segment(*seg) for seg in segments])
        counts = (C.c_size_t * max(1, len(segments)))()
        n = self._ResampleSegmentsDevice(C.byref(resampler), C.byref(precomputed), C.c_void_p(device_timeline), halo_frames, array, len(segments),
                                         C.c_void_p(device_output), output_capacity_frames, 1 if s16 else 0, counts, C.c_void_p(hip_stream or 0))
        _raise_if_failed(self.lib)
        return n, list(counts)[:len(segments)]

    def ResampleShardedDevice(self, resampler, precomputed, total_input_frames, shards, s16=False, gather_mode=GATHER_NONE, root_shard=0, root_output=None):
        """One stream over several devices (ClownResamplerAMD_ResampleShardedDevice).  shards: [(device, device_input, device_output, hip_stream), ...].
        Returns the total number of output frames; nothing is synchronised (ShardedSynchronize)."""
        array = (DeviceShard * len(shards))(*[DeviceShard(d, C.c_void_p(i), C.c_void_p(o), C.c_void_p(s or 0)) for d, i, o, s in shards])
        n = self._ResampleSharded(C.byref(resampler), C.byref(precomputed), total_input_frames, array, len(shards), 1 if s16 else 0,
                                  gather_mode, root_shard, C.c_void_p(root_output or 0))
        _raise_if_failed(self.lib)
        return n

    def ShardedSynchronize(self, shards):
        array = (DeviceShard * len(shards))(*[DeviceShard(d, C.c_void_p(i), C.c_void_p(o), C.c_void_p(s or 0)) for d, i, o, s in shards])
        r = self._ShardedSync(array, len(shards))
        _raise_if_failed(self.lib)
        return r

    def SetThreadDevice(self, ordinal):
        r = self._SetThreadDevice(ordinal)
        _raise_if_failed(self.lib)
        return r

    def GetDevice(self):
        return self._GetDevice()

    def ReserveCaptureLaunches(self, launches):
        r = self._ReserveCapture(launches)
        _raise_if_failed(self.lib)
        return r

    def PlanKernelAt(self, plan, position_fractional=0):
        """kernel id a launch from this fractional position takes (PlanInfo.kernel numbering, 5 = k_int)"""
        return int(self._PlanKernelAt(plan, position_fractional))

    def HostIsDeviceVisible(self, address, nbytes):
        """1 when the host-pointer entry points would use [address, address + nbytes) in place (page-locked, device-addressable), 0 when they stage it"""
        return int(self._HostVisible(C.c_void_p(address), nbytes))

    def DebugSegmentsMode(self, mode):
        """0: the rule picks, 1: one launch per segment, 2: one launch for all segments (segment table)"""
        self._SegmentsMode(mode)

    def PlanSegKernel(self, plan):
        """8 when long launches of this plan may take k_seg, else 0"""
        return int(self._PlanSeg(plan))

    def PlanDualMonoKernel(self, plan):
        """0, or the kernel id of the stereo instance long launches of this mono plan run on (dual mono)"""
        return int(self._PlanDualMono(plan))

    def PlanPaddedTiles(self, plan):
        """1 when the plan's k_poly instance repacks its tiles to 32-byte frames (9-11, 13-15 channels, no specialised instance, up to 2:1)"""
        return int(self._PlanPadded(plan))

    def DebugSegKernel(self, mode):
        """0: the rule, 1: k_seg for every launch it can take, 2: never (kernel 8 of LaunchCount counts its launches)"""
        self._SegKernel(mode)

    def DebugDisableDualMono(self, on):
        self._DisableDual(1 if on else 0)

    def DebugDisableIntKernel(self, on):
        self._DisableInt(1 if on else 0)

    def BuildId(self):
        return self._BuildId().decode()

    def LaunchCount(self, kernel):
        return int(self._LaunchCount(kernel))

    def ReleaseCapturedLaunches(self):
        r = self._ReleaseCaptured()
        _raise_if_failed(self.lib)
        return r

    def HighLevel_Release(self, resampler):
        self._HighRelease(C.byref(resampler))

    def StreamingWindowCount(self):
        return self._WindowCount()

    def DeviceAllocOn(self, device, nbytes):
        p = self._DeviceAllocOn(device, nbytes)
        _raise_if_failed(self.lib)
        return p

    def SetPlanCacheLimit(self, plans):
        self._SetPlanCacheLimit(plans)

    def PlanCacheCount(self):
        return self._PlanCacheCount()

    def SetStreamingWindow(self, frames):
        self._SetStreamingWindow(frames)

    def DebugSetVariant(self, variant):
        self._SetVariant(variant)

    def DebugForceGenericKernel(self, on):
        self._ForceGeneric(1 if on else 0)

    def IsUsable(self):
        """1 when a gfx950 device is there for this library; never raises, never aborts (ClownResamplerAMD_IsUsable)"""
        return int(self._IsUsable())

    def DeviceCount(self):
        return self._DeviceCount()

    def SetDevice(self, ordinal):
        r = self._SetDevice(ordinal)
        _raise_if_failed(self.lib)
        return r

    def Shutdown(self):
        self._Shutdown()

    def DeviceAlloc(self, nbytes):
        p = self._DeviceAlloc(nbytes)
        _raise_if_failed(self.lib)
        return p

    def DeviceFree(self, p):
        self._DeviceFree(p)

    def CopyToDevice(self, dst, host_array):
        a = np.ascontiguousarray(host_array)
        self._ToDevice(dst, a.ctypes.data, a.nbytes)
        _raise_if_failed(self.lib)

    def CopyFromDevice(self, host_array, src):
        assert host_array.flags["C_CONTIGUOUS"]
        self._FromDevice(host_array.ctypes.data, src, host_array.nbytes)
        _raise_if_failed(self.lib)

    def StreamSynchronize(self, stream=None):
        self._StreamSync(C.c_void_p(stream or 0))
        _raise_if_failed(self.lib)


_apis = {}


def load(radius=3, abi="c89"):
    if (radius, abi) not in _apis:
        _apis[(radius, abi)] = Api(radius, abi)
    return _apis[(radius, abi)]
