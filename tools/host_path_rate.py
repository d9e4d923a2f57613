#!/usr/bin/env python3
"""End-to-end (PCIe-inclusive) rates of the host-buffer entry points on cfg 2: the boundary hands over HOST buffers, so
these include H2D of the int16 input, the kernel, and D2H of the int32 output (and, for the callback form, one indirect
call per frame).  Never the bench `value`; quoted in DESIGN.md."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C  # noqa: E402
import torch  # noqa: E402
torch.cuda.init()   # before the library's own first HIP call: torch refuses to initialise afterwards on this image
import numpy as np  # noqa: E402
import _product  # noqa: E402


def pad_frames(pcm, channels, radius_frames):
    z = np.zeros(radius_frames * channels, dtype=np.int16)
    return np.concatenate([z, np.asarray(pcm, dtype=np.int16), z])


def noise_pcm(samples):
    return np.random.default_rng(20261002).integers(-32768, 32768, samples, dtype=np.int16)
import clownresampler_amd as cr  # noqa: E402

p = _product.Product(3)
ch, rates, frames = 2, (44100, 48000, 44100), 26460000
padded = pad_frames(noise_pcm(frames * ch), ch, 3)
for rep in range(3):
    ok, st = p.low_init(ch, *rates)
    out = np.zeros((28800096 + 1) * ch, dtype=np.int32)      # fresh, untouched pages: the first write faults them in
    t0 = time.perf_counter()
    got, left, ran_out = p.api.LowLevel_ResampleBulk(st.raw, p.pre, padded, frames, None, out)
    dt = time.perf_counter() - t0
    print("ResampleBulk (host buffers, pageable, output pages never touched before): %.1f ms  %.0f Msamples/s" % (dt * 1e3, got.size / dt / 1e6))
out = np.zeros((28800096 + 1) * ch, dtype=np.int32)
for rep in range(4):
    ok, st = p.low_init(ch, *rates)
    t0 = time.perf_counter()
    got, left, ran_out = p.api.LowLevel_ResampleBulk(st.raw, p.pre, padded, frames, None, out)
    dt = time.perf_counter() - t0
    print("ResampleBulk (host buffers, pageable, same buffers as the call before): %.1f ms  %.0f Msamples/s" % (dt * 1e3, got.size / dt / 1e6))

# callback form with a C callback (storing int32), as a C client would use it
src = r'''
#include <stddef.h>
typedef struct { int *out; size_t n; } sink;
unsigned char store(void *u, const long *frame, unsigned int samples) { sink *s = (sink *)u; unsigned int c; for (c = 0; c < samples; ++c) s->out[s->n++] = (int)frame[c]; return 1; }
'''
import subprocess, tempfile
d = tempfile.mkdtemp()
open(d + "/cb.c", "w").write(src)
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", d + "/cb.c", "-o", d + "/cb.so"], check=True)
cb = C.CDLL(d + "/cb.so")


class Sink(C.Structure):
    _fields_ = [("out", C.c_void_p), ("n", C.c_size_t)]


for rep in range(2):
    ok, st = p.low_init(ch, *rates)
    out = np.zeros((28800096 + 1) * ch, dtype=np.int32)
    sink = Sink(out.ctypes.data, 0)
    left = C.c_size_t(frames)
    fn = p.api.lib.ClownResampler_LowLevel_Resample
    fn.restype = C.c_ubyte
    fn.argtypes = [C.c_void_p] * 6
    t0 = time.perf_counter()
    fn(C.addressof(st.raw), C.addressof(p.pre), padded.ctypes.data, C.addressof(left), C.cast(cb.store, C.c_void_p), C.addressof(sink))
    dt = time.perf_counter() - t0
    print("LowLevel_Resample (reference signature, C callback per frame): %.1f ms  %.0f Msamples/s" % (dt * 1e3, sink.n / dt / 1e6))

# high-level streaming API with C callbacks (pull from memory, push storing int32)
src2 = r'''
#include <stddef.h>
#include <string.h>
typedef struct { const short *in; size_t left; unsigned ch; int *out; size_t n; size_t pulls; } io;
size_t pull(void *u, short *buf, size_t frames) { io *s = (io *)u; size_t k = frames < s->left ? frames : s->left; memcpy(buf, s->in, k * s->ch * 2); s->in += k * s->ch; s->left -= k; s->pulls++; return k; }
unsigned char push(void *u, const long *frame, unsigned int samples) { io *s = (io *)u; unsigned int c; for (c = 0; c < samples; ++c) s->out[s->n++] = (int)frame[c]; return 1; }
'''
open(d + "/hl.c", "w").write(src2)
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", d + "/hl.c", "-o", d + "/hl.so"], check=True)
hl = C.CDLL(d + "/hl.so")


class IO(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("left", C.c_size_t), ("ch", C.c_uint), ("out", C.c_void_p), ("n", C.c_size_t), ("pulls", C.c_size_t)]


pcm = padded[3 * ch: 3 * ch + 2646000 * ch].copy()   # 1 minute
for rep in range(2):
    hs = p.api.HighLevel_State()
    p.api.HighLevel_Init(hs, ch, *rates)
    out = np.zeros((2880010 + 8) * ch, dtype=np.int32)
    io = IO(pcm.ctypes.data, 2646000, ch, out.ctypes.data, 0, 0)
    f1 = p.api.lib.ClownResampler_HighLevel_Resample
    f2 = p.api.lib.ClownResampler_HighLevel_ResampleEnd
    f1.restype = f2.restype = C.c_ubyte
    f1.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    t0 = time.perf_counter()
    f1(C.addressof(hs), C.addressof(p.pre), C.cast(hl.pull, C.c_void_p), C.cast(hl.push, C.c_void_p), C.addressof(io))
    f2(C.addressof(hs), C.addressof(p.pre), C.cast(hl.push, C.c_void_p), C.addressof(io))
    dt = time.perf_counter() - t0
    print("HighLevel_Resample + ResampleEnd (1 min stereo, C callbacks, %d input pulls): %.1f ms  %.0f Msamples/s" % (io.pulls, dt * 1e3, io.n / dt / 1e6))

# variable rate on the device (ClownResamplerAMD_ResampleSegmentsDevice): the cfg 2 timeline cut into constant-rate segments
# whose output rate wanders around 48 kHz (clock drift, pitch bend), device-resident buffers, one launch per segment
dev = torch.device("cuda", 0)
halo = 4
d_in = torch.from_numpy(pad_frames(padded[3 * ch: (3 + frames) * ch], ch, halo)).to(dev)
for seg_frames in (4410000, 441000, 44100, 4410):
    nseg = frames // seg_frames
    segs = [(seg_frames, 44100, 48000 + (k % 997) - 498, 44100) for k in range(nseg)]    # up to 997 distinct ratios, one configuration
    d_out = torch.empty(int(frames * 48003 / 44100 + 16) * ch, dtype=torch.int32, device=dev)
    for mode, what in ((1, "one launch per segment"), (2, "ONE launch, segment table"), (0, "the rule's choice")):
        p.api.DebugSegmentsMode(mode)
        for rep in range(3):
            st = p.api.LowLevel_State()
            p.api.LowLevel_Init(st, ch, *rates)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n, counts = p.api.ResampleSegmentsDevice(st, p.pre, d_in.data_ptr() + halo * ch * 2, halo, segs, d_out.data_ptr(), d_out.numel() // ch)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print("ResampleSegmentsDevice: %5d segments of %7d frames, up to 997 distinct ratios, %-27s: host %.2f ms, done %.2f ms  %.0f Msamples/s"
              % (nseg, seg_frames, what, (t1 - t0) * 1e3, dt * 1e3, n * ch / dt / 1e6))
    p.api.DebugSegmentsMode(0)
