"""Does hipFree wait for work in flight (the library's plan eviction relies on it: cr_context.c store_release)?  A spinning kernel is
queued, then a buffer from hipMalloc is freed: the call's duration says.  Through the HIP runtime the process has loaded (torch's copy)."""
import ctypes as C
import os
import sys
import time

import torch

lib = None
for name in (os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"), "libamdhip64.so"):
    try:
        lib = C.CDLL(name)
        break
    except OSError:
        pass
lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
lib.hipFree.argtypes = [C.c_void_p]
lib.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
lib.hipHostFree.argtypes = [C.c_void_p]

torch.zeros(1, device="cuda")
torch.cuda.synchronize()
v = C.c_int()
lib.hipRuntimeGetVersion(C.byref(v))
print("hip runtime version", v.value, "torch", torch.__version__)


def spin_ms(stream=None):
    # ~ cycles of the shader clock
    if stream is None:
        torch.cuda._sleep(400_000_000)
    else:
        with torch.cuda.stream(stream):
            torch.cuda._sleep(400_000_000)


for label, stream in (("null stream", None), ("side stream", torch.cuda.Stream())):
    for what in ("hipFree", "hipHostFree"):
        p = C.c_void_p()
        if what == "hipFree":
            assert lib.hipMalloc(C.byref(p), 1 << 16) == 0
        else:
            assert lib.hipHostMalloc(C.byref(p), 1 << 16, 0) == 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        spin_ms(stream)
        t1 = time.perf_counter()
        r = (lib.hipFree if what == "hipFree" else lib.hipHostFree)(p)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%-12s %-12s: launch %.2f ms, %s %.2f ms (rc %d), rest of the kernel %.2f ms" % (label, what, (t1 - t0) * 1e3, what, (t2 - t1) * 1e3, r, (t3 - t2) * 1e3))
sys.stdout.flush()
