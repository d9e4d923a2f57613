"""Does the ordinary hipMalloc / hipFree path show what the virtual-memory API showed (vmm_probe.py: a range freed and handed out again
seen differently by kernels and by the copy engines)?  Allocate, fill by a kernel, read by the copy engine, fill by the copy engine, read
by a kernel, free; sizes shuffled so that ranges are reused with different physical pages.  usage: va_reuse_probe.py [seconds]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
lib.hipFree.argtypes = [C.c_void_p]
lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
torch.zeros(1, device="cuda")
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(7)
sizes = [4096, 65536, 100000, 1 << 20, (1 << 20) + 4096, 3 << 20, 20 << 20]
src = torch.randint(0, 256, (max(sizes),), dtype=torch.uint8, device="cuda")
src_host = src.cpu().numpy()
host = np.empty(max(sizes), dtype=np.uint8)
seen, live, bad, iters = {}, [], 0, 0
t0 = time.time()
while time.time() - t0 < seconds:
    iters += 1
    n = int(sizes[rng.integers(len(sizes))])
    p = C.c_void_p()
    assert lib.hipMalloc(C.byref(p), n) == 0
    seen[p.value] = seen.get(p.value, 0) + 1
    k = int(rng.integers(0, 256))
    # a kernel writes (memset), the copy engine reads
    lib.hipMemset(p, k, n)
    lib.hipMemcpy(host.ctypes.data_as(C.c_void_p), p, n, 2)
    ok1 = bool(np.all(host[:n] == k))
    # the copy engine writes, a kernel reads (device-to-device copy into a torch tensor)
    off = int(rng.integers(0, max(sizes) - n + 1))
    lib.hipMemcpy(p, src_host[off:].ctypes.data_as(C.c_void_p), n, 1)
    t = torch.empty(n, dtype=torch.uint8, device="cuda")
    lib.hipMemcpy(C.c_void_p(t.data_ptr()), p, n, 3)
    torch.cuda.synchronize()
    ok2 = bool(torch.equal(t, src[off:off + n]))
    if not (ok1 and ok2):
        bad += 1
        print("iteration %d: %d bytes at 0x%x (handed out %d times): kernel write / engine read %s, engine write / kernel read %s" % (iters, n, p.value, seen[p.value], ok1, ok2), flush=True)
    live.append(p)
    while len(live) > int(rng.integers(0, 4)):
        lib.hipFree(live.pop(int(rng.integers(len(live)))))
print("%d iterations, %d distinct addresses, %d reused, bad: %d" % (iters, len(seen), sum(1 for v in seen.values() if v > 1), bad))
