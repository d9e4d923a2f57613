"""Device buffers between UNMAPPED pages, for the GPU parity tests (test infrastructure; nothing in the product uses it).

There is no GPU AddressSanitizer on this pool, and every buffer a test gets from torch's caching allocator or from hipMalloc sits
inside a 2 MiB (or larger) mapping: a kernel that reads a few hundred bytes past what it was given lands in mapped memory on one box
and on a hole on another - an abort "one run in eight".  Here a buffer is placed with HIP's virtual-memory API (hipMemAddressReserve /
hipMemCreate / hipMemMap / hipMemSetAccess) so that it ENDS on the last mapped byte ("end") or STARTS on the first ("start"), the
neighbouring pages reserved and left unmapped: the first stray access is a GPU memory fault, every run, and the library's flight
recorder (ClownResamplerAMD_DebugInstallAbortDump, tests/conftest.py) names the launch.  What a caller's buffer must hold - and
nothing beyond - is clownresampler.h:725-733.
"""
import ctypes as C
import os

import numpy as np

_hip = None
KEEP_ADDRESSES = os.environ.get("CRA_GUARDED_KEEP", "1") != "0"


class _Location(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class _AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class _AllocationProp(C.Structure):  # hipMemAllocationProp
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", _Location), ("win32HandleMetaData", C.c_void_p), ("allocFlags", _AllocFlags)]


class _AccessDesc(C.Structure):  # hipMemAccessDesc
    _fields_ = [("location", _Location), ("flags", C.c_int)]


def hip():
    """the HIP runtime this process has loaded (torch's copy when torch came first: tests/conftest.py)"""
    global _hip
    if _hip is None:
        names = []
        try:
            import torch
            names.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        except Exception:
            pass
        names += ["libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"]
        for name in names:
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("no libamdhip64.so")
        _hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        _hip.hipMemAddressFree.argtypes = [C.c_void_p, C.c_size_t]
        _hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(_AllocationProp), C.c_ulonglong]
        _hip.hipMemRelease.argtypes = [C.c_void_p]
        _hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
        _hip.hipMemUnmap.argtypes = [C.c_void_p, C.c_size_t]
        _hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_AccessDesc), C.c_size_t]
        _hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(_AllocationProp), C.c_int]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipGetErrorString.restype = C.c_char_p
        _hip.hipGetErrorString.argtypes = [C.c_int]
    return _hip


def _check(code, what):
    if code != 0:
        raise RuntimeError("%s failed: %s (hipError %d)" % (what, hip().hipGetErrorString(code).decode(), code))


def _prop(device):
    p = _AllocationProp()
    p.type = 1              # hipMemAllocationTypePinned
    p.requestedHandleType = 0
    p.location.type = 1     # hipMemLocationTypeDevice
    p.location.id = device
    return p


_granularity = {}


def granularity(device=0):
    if device not in _granularity:
        g = C.c_size_t(0)
        prop = _prop(device)
        _check(hip().hipMemGetAllocationGranularity(C.byref(g), C.byref(prop), 0), "hipMemGetAllocationGranularity")
        _granularity[device] = max(int(g.value), 4096)
    return _granularity[device]


class Guarded:
    """`nbytes` of device memory at .ptr with an unmapped page behind its last byte (place "end") or in front of its first ("start");
    `offset` moves an "end" buffer that many bytes DOWN / a "start" buffer that many bytes UP (alignment phases - the gap then is
    `offset` bytes of mapped memory, which the kernels' aligned fetches are allowed: same 16-byte block, same page)."""

    def __init__(self, nbytes, place="end", offset=0, device=0, fill=None):
        h = hip()
        self.nbytes = int(nbytes)
        g = granularity(device)
        self.mapped = (max(self.nbytes, 1) + offset + g - 1) // g * g
        self.reserved = self.mapped + 2 * g
        self.base = C.c_void_p()
        self.handle = C.c_void_p()
        _check(h.hipMemAddressReserve(C.byref(self.base), self.reserved, g, None, 0), "hipMemAddressReserve")
        prop = _prop(device)
        code = h.hipMemCreate(C.byref(self.handle), self.mapped, C.byref(prop), 0)
        if code != 0:
            h.hipMemAddressFree(self.base, self.reserved)
            _check(code, "hipMemCreate")
        self.first = self.base.value + g          # first mapped byte
        _check(h.hipMemMap(C.c_void_p(self.first), self.mapped, 0, self.handle, 0), "hipMemMap")
        desc = _AccessDesc()
        desc.location.type = 1
        desc.location.id = device
        desc.flags = 3                            # hipMemAccessFlagsProtReadWrite
        _check(h.hipMemSetAccess(C.c_void_p(self.first), self.mapped, C.byref(desc), 1), "hipMemSetAccess")
        self.ptr = self.first + self.mapped - self.nbytes - offset if place == "end" else self.first + offset
        self.place = place
        self.live = True
        if fill is not None:
            # (a host-side pattern copied in, and the device waited for: hipMemset on such a mapping was seen to land LATE - after a
            # device-to-host copy issued behind it - on this runtime; tools/experiments/r06/vmm_probe.py)
            pattern = np.full(self.mapped, fill, dtype=np.uint8)
            _check(h.hipMemcpy(C.c_void_p(self.first), pattern.ctypes.data_as(C.c_void_p), self.mapped, 1), "hipMemcpy(H2D, fill)")
            _check(h.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def write(self, array, at=0):
        a = np.ascontiguousarray(array)
        assert at + a.nbytes <= self.nbytes, (at, a.nbytes, self.nbytes)
        if a.nbytes:
            _check(hip().hipMemcpy(C.c_void_p(self.ptr + at), a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy(H2D)")
            _check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")

    def read(self, dtype, count=None, at=0):
        dtype = np.dtype(dtype)
        count = (self.nbytes - at) // dtype.itemsize if count is None else count
        out = np.empty(count, dtype=dtype)
        assert at + out.nbytes <= self.nbytes
        if out.nbytes:
            _check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
            _check(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + at), out.nbytes, 2), "hipMemcpy(D2H)")
        return out

    def read_mapped(self):
        """every mapped byte (the buffer and the slack the placement left inside the mapping)"""
        out = np.empty(self.mapped, dtype=np.uint8)
        _check(hip().hipDeviceSynchronize(), "hipDeviceSynchronize")
        _check(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.first), out.nbytes, 2), "hipMemcpy(D2H)")
        return out

    def close(self):
        """Unmaps and gives the physical pages back; the ADDRESS RANGE stays reserved for the life of the process (KEEP_ADDRESSES).  On this
        runtime a range that was unmapped, freed and handed out again by the next hipMemAddressReserve showed kernels, hipMemset and the
        copy engines DIFFERENT contents - some of them still translating through the old mapping (tools/experiments/r06/vmm_probe.py,
        profiles/r06_vmm_probe.log: 16 of 48 buffers bad with reuse, 0 of 48 without).  A test allocator must not add a hazard of its own."""
        if self.live:
            h = hip()
            h.hipDeviceSynchronize()
            h.hipMemUnmap(C.c_void_p(self.first), self.mapped)
            h.hipMemRelease(self.handle)
            if not KEEP_ADDRESSES:
                h.hipMemAddressFree(self.base, self.reserved)
            self.live = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
