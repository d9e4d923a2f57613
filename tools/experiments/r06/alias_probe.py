"""Does the library mistake memory the RUNTIME page-locked for its own copy (a > 1 MiB copy into pageable memory) for memory the CLIENT page-locked
(hipHostRegister / hipHostMalloc)?  The host-pointer entry points use page-locked caller buffers in place (cr_run_host, crhip_host_alias)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path[:0] = [os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tests")]
torch.cuda.init()
import clownresampler_amd as cr

api = cr.load(3)
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
n = 5 << 20
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
a = np.zeros(n, dtype=np.uint8)
print("fresh pageable array: device-visible?", api.HostIsDeviceVisible(a.ctypes.data, n))
hip.hipMemcpy(C.c_void_p(a.ctypes.data), C.c_void_p(src.data_ptr()), n, 2)
print("after a synchronous 5 MB copy into it (the runtime pins it in place):", api.HostIsDeviceVisible(a.ctypes.data, n))
hip.hipMemcpyAsync(C.c_void_p(a.ctypes.data), C.c_void_p(src.data_ptr()), n, 2, None)
print("right after an ASYNC copy into it:", api.HostIsDeviceVisible(a.ctypes.data, n))
torch.cuda.synchronize()
print("after the synchronise:", api.HostIsDeviceVisible(a.ctypes.data, n))
b = src.cpu().numpy()
print("a tensor.cpu() result:", api.HostIsDeviceVisible(b.ctypes.data, n))
t = torch.empty(n, dtype=torch.uint8).pin_memory().numpy()
print("torch pinned memory (must be 1):", api.HostIsDeviceVisible(t.ctypes.data, n))
