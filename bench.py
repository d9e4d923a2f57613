#!/usr/bin/env python3
"""bench.py - throughput of the windowed-sinc hot path on MI355X, BASELINE.json's metric.

One "step" = one pass of the hot path (ClownResampler_LowLevel_Resample's per-frame loop, reference
clownresampler.h:1058-1092 / :986-1035, as the HIP kernel k_poly) over one batch of synthetic PCM, inputs and outputs
resident in HBM.

N = 1 (default): the batch is BASELINE configs[1]: stereo int16, 44.1 -> 48 kHz, 3-lobe Lanczos, 10 minutes
(26,460,000 -> 28,800,096 frames).
N > 1 (default): BASELINE configs[4]: ONE 1-hour stereo 44.1 -> 48 kHz stream (158,760,000 -> 172,800,574 frames) whose
output timeline is split into N contiguous blocks by ClownResamplerAMD_PlanShard, one rank (= one process, one GPU) per
block, input halo replicated, no collective in the data path: STRONG scaling.  `--scaling weak` gives every rank its own
10-minute shard of an N x 10-minute stream instead.  The final concatenate (north_star) is timed apart from the kernel:
gather-to-root and all-gather over RCCL, reported as `gather` beside the kernel-only `value`.

Every line carries its curve's N = 1 point: the default N = 1 line has `strong_curve_n1` (the hour of configs[4] as one launch on this
GPU), every N > 1 line `efficiency_vs_n1` (each rank also runs the WHOLE stream alone on its own GPU, in the same process, after the
timed region).  `parity_full_stream`: every sample of the last timed launch of every rank is compared with the all-core oracle.
Ranks carry a wall-clock limit (--rank-timeout) and leave with status 124 when it passes.

`python bench.py --gpus N` starts its own N workers (a child `python -m torch.distributed.run`, started before this
process has touched the GPU); under an existing torch.distributed.run (RANK/WORLD_SIZE in the environment) it is a worker.

Prints ONE JSON line (rank 0).  `value` = output Msamples/s of the whole job, kernel time only: one HIP event pair on
the launch stream around exactly K launches, barrier + synchronize on both sides, max over ranks (`ms_per_step`); a second
region of 10 blocks of launches gives the distribution (`launch_us`: median / min / p95) and `value_at_median`.  `roofline` prices the same kernel against the
8 TB/s HBM peak with ALGORITHMIC bytes (each input sample read once + each int32 output written once);
for workloads that PMC counters show to be bound by the integer VALU, a second object `roofline_valu` prices the measured
VALU wave-instructions against the SIMDs' measured issue rate.  `cpu_baseline` is the reference C path (oracle/_ref, the
real header compiled in place, when that prebuilt checker is present; else the oracle restatement) timed on this box's
host cores on the same workload - a baseline, not a target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# The runtime copies more than 1 MiB between the device and pageable host memory by page-locking the host range in place; the only aborts this
# project has seen on the GPU box - GPU page faults at heap addresses during such copies by torch (tests/conftest.py, HISTORY.md round 6) - came
# from there.  This process's torch copies (the parity check reads every sample of the last timed launch back) stay on the staging path;
# nothing in the timed region is a host copy.  Read when libamdhip64 initialises: set before torch is imported; child rank processes inherit it.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
SIMDS = 1024           # 256 CUs x 4 SIMDs

WORKLOADS = {
    # name: (radius, channels, (in, out, lowpass), input frames)
    "cfg2": (3, 2, (44100, 48000, 44100), 26460000),   # BASELINE configs[1] - the headline
    "cfg3": (8, 2, (8000, 96000, 8000), 4800000),      # configs[2] (10 min assumed)
    "cfg4": (3, 8, (48000, 44100, 44100), 28800000),   # configs[3] (10 min assumed)
    "cfg5": (3, 2, (44100, 48000, 44100), 158760000),  # configs[4]: the 1-hour stream
    # not a BASELINE configuration: the headline conversion with the reference's high-quality 8-lobe build (tuning only)
    "hq48": (8, 2, (44100, 48000, 44100), 26460000),
    "up55": (3, 2, (8000, 44100, 8000), 4800000),      # one of the reference's ctest triples (tests/CMakeLists.txt), 10 min, 3 lobes
    "ch6": (3, 6, (44100, 48000, 44100), 8820000),     # 5.1 surround through the run-time-slot instance
    "ch3": (3, 3, (48000, 44100, 44100), 14400000),
    "ch16": (3, 16, (44100, 48000, 44100), 3307500),   # the reference's maximum channel count
    "dn8": (3, 2, (44100, 8000, 8000), 26460000),      # ctest triple 44100 -> 8000: 33-tap windows
    "mono": (3, 1, (44100, 48000, 44100), 52920000),
    "dn2": (3, 2, (48000, 44100, 44100), 28800000),    # stereo 48 -> 44.1 kHz, 10 min
    "dn1": (3, 1, (48000, 44100, 44100), 57600000),    # mono 48 -> 44.1 kHz, 20 min
    "hq44": (8, 2, (48000, 44100, 44100), 28800000),   # 8 lobes, 48 -> 44.1 kHz (17-tap windows)
    "hq48m": (8, 1, (44100, 48000, 44100), 52920000),  # 8 lobes, mono
    "hq44m": (8, 1, (48000, 44100, 44100), 57600000),
    "hq48c4": (8, 4, (44100, 48000, 44100), 13230000), # 8 lobes, 4 / 6 / 8 channels
    "hq48c6": (8, 6, (44100, 48000, 44100), 8820000),
    "hq48c8": (8, 8, (44100, 48000, 44100), 6615000),
    "hq44c6": (8, 6, (48000, 44100, 44100), 9600000),
    "hq48c3": (8, 3, (44100, 48000, 44100), 17640000),
    "hq48c5": (8, 5, (44100, 48000, 44100), 10584000),
    "dn21": (3, 2, (96000, 48000, 48000), 57600000),   # stereo 2:1, 12-slot windows
    "dn32": (3, 2, (48000, 32000, 32000), 28800000),   # stereo 3:2, 9-slot windows
    "dn31": (3, 2, (96000, 32000, 32000), 57600000),   # stereo 3:1, 18-slot windows
    "dn96": (3, 2, (96000, 44100, 44100), 57600000),   # stereo 96 -> 44.1 kHz, 13-slot windows
    "dn4432": (3, 2, (44100, 32000, 32000), 26460000), # stereo 44.1 -> 32 kHz, 8-slot windows
    "cfg3s": (8, 2, (8000, 96000, 8000), 80000),       # cfg 3's conversion, 10 s and 1 min of it (launch floor of k_up2)
    "cfg3m": (8, 2, (8000, 96000, 8000), 480000),
    "up2x": (3, 2, (48000, 96000, 48000), 14400000),   # stereo 3 lobes 2x / 4x / 3x upsampling, 5 minutes in
    "up4x": (3, 2, (48000, 192000, 48000), 7200000),
    "up3x": (3, 2, (16000, 48000, 16000), 9600000),
    "up8xm": (3, 1, (12000, 96000, 12000), 7200000),   # exactly 8x / 16x, 3 lobes: the rows of neighbouring lanes share an LDS bank slot unless rotated
    "up8x4": (3, 4, (12000, 96000, 12000), 3600000),
    "up16xm": (3, 1, (12000, 192000, 12000), 3600000),
    "up16x6": (3, 6, (12000, 192000, 12000), 1200000),
    "up8x3": (3, 3, (12000, 96000, 12000), 3600000),
    "up16x8": (3, 8, (12000, 192000, 12000), 900000),
    "up4": (3, 4, (44100, 48000, 44100), 26460000),    # 10 minutes of 4 / 6 / 8 channels, 3 lobes, both directions (tuning only)
    "up6": (3, 6, (44100, 48000, 44100), 26460000),
    "up8": (3, 8, (44100, 48000, 44100), 26460000),
    "dn4": (3, 4, (48000, 44100, 44100), 28800000),
    "dn6": (3, 6, (48000, 44100, 44100), 28800000),
    "up12": (3, 12, (44100, 48000, 44100), 26460000),
    "hq48c16": (8, 16, (44100, 48000, 44100), 3307500),  # 8 lobes, the widest frame: k_wave2s (a lane per channel pair)
    "hq48c12": (8, 12, (44100, 48000, 44100), 4410000),
    "dn8c12": (3, 12, (44100, 8000, 8000), 4410000),
    "dn21c4": (3, 4, (96000, 48000, 48000), 28800000),   # 2:1 with 4 / 6 / 8 channels, 3:1 with 4: k_int, a frame's channel pairs one after the other
    "dn21c6": (3, 6, (96000, 48000, 48000), 19200000),
    "dn21c8": (3, 8, (96000, 48000, 48000), 14400000),
    "dn31c4": (3, 4, (96000, 32000, 32000), 28800000),
    "dn6x": (3, 2, (48000, 8000, 8000), 57600000),     # stereo / mono 6:1 (increment exact in 16.16: every frame uses ONE row), 36-slot windows
    "dn6xm": (3, 1, (48000, 8000, 8000), 115200000),
    "dn8m": (3, 1, (44100, 8000, 8000), 52920000),
    "dn4x": (3, 2, (48000, 12000, 12000), 57600000),   # stereo 4:1, 24-slot windows
    "dn21m": (3, 1, (96000, 48000, 48000), 115200000),
}
CONFIG_NAMES = {"cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2]", "cfg4": "BASELINE configs[3]", "cfg5": "BASELINE configs[4]"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_workers(n, timeout_s):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children of THIS process, which has
    not initialised the GPU (no torch import, no HIP call - never an exec of a process that has), wait, pass on their status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # the children are FRESH processes in their own group; a launcher that outlives its limit takes the whole group down and reports it
    child = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return child.wait(timeout=timeout_s + 60.0)
    except subprocess.TimeoutExpired:
        print("bench: the %d ranks did not finish within %.0f s - killing them" % (n, timeout_s + 60.0), file=sys.stderr)
        try:
            os.killpg(child.pid, 9)
        except OSError:
            pass
        child.wait()
        return 124


def arm_watchdog(seconds, rank):
    """Per-rank wall-clock limit: a rank stuck in a collective (a peer died, a link hung) must end the job with a non-zero status, not
    sit in the driver's slot.  A daemon timer that leaves with os._exit - no exec, no signal to anything but this process."""
    import threading

    def fire():
        print("bench: rank %d exceeded its wall-clock limit of %.0f s - exiting 124" % (rank, seconds), file=sys.stderr, flush=True)
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def device_noise(n_samples, first_index, device):
    """Full-scale white int16 PCM as a pure function of the ABSOLUTE sample index (splitmix-style hash), generated on the
    device: every rank can materialise any range of one and the same stream, halo included."""
    import torch
    out = torch.empty(n_samples, dtype=torch.int16, device=device)
    step = 1 << 24
    for s in range(0, n_samples, step):
        n = min(step, n_samples - s)
        x = torch.arange(first_index + s, first_index + s + n, dtype=torch.int64, device=device)
        x = x * -7046029254386353131            # 0x9E3779B97F4A7C15 as int64
        x = (x ^ (x >> 31)) * -4658895280553007687  # 0xBF58476D1CE4E5B9
        x = x ^ (x >> 29)
        out[s:s + n] = (x >> 40).to(torch.int16)
    return out


def stream_slice(first_logical, frames, ch, total_frames, salt, device):
    """Padded-buffer view of the whole stream: logical frames [first_logical, first_logical + frames), zeros outside
    [0, total_frames) (the reference's zero padding at the two ends of the stream, clownresampler.h:725-733)."""
    pcm = device_noise(frames * ch, first_logical * ch + salt, device)
    lo_pad = min(frames, max(0, -first_logical))
    hi_pad = min(frames, max(0, first_logical + frames - total_frames))
    if lo_pad:
        pcm[: lo_pad * ch] = 0
    if hi_pad:
        pcm[(frames - hi_pad) * ch:] = 0
    return pcm


def cpu_baseline(radius, ch, rates, frames, max_seconds=30.0):
    """Times the reference C path on the host: real reference (.so prebuilt from /root/reference) if present, else oracle.
    Bounded sample: at most 10 minutes of the workload's stream (the 1-thread leg takes ~0.5 s per minute of stereo)."""
    import numpy as np
    import _checkers as ck
    frames = min(frames, 28800000)
    ref = ck.reference(radius)
    eng, kind = (ref, "reference") if ref is not None else (ck.oracle(radius), "port")
    ok, st = eng.low_init(ch, *rates)
    R = int(st.cfg.radius_frames)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch), ch, R)
    n_out = int(ck.count_output_frames(st, frames))
    out = np.zeros((n_out + 1) * ch, dtype=np.int32)   # pre-faulted
    t0 = time.perf_counter()
    got, left, ran_out = eng.low_resample_i32(st, padded, frames, out=out)
    dt = time.perf_counter() - t0
    res = {"value": got.size / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": kind,
           "sample": "%d -> %d frames x %d ch of the workload, callback API storing int32, gcc -O2, 1 thread, %.2f s" % (frames, n_out, ch, dt)}
    # all host cores, independent states over contiguous input ranges (oracle driver)
    # (this leg read 765 Msamples/s on one box and 2,678 on another.  It is 256 threads for a tenth of a second: what it
    # measures first is how many of the machine's hardware threads THIS process may use and how fast they start.  So: one thread per CPU the
    # process is allowed - its affinity mask, the cgroup's quota - not per CPU the machine has; the best of three; and the line says all of it.)
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    cores = max(1, min(allowed, int(quota + 0.5)) if quota else allowed)
    if cores > 1:
        o = ck.oracle(radius)
        ok, fresh = o.low_init(ch, *rates)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            got = o.low_resample_i32_mt(fresh, padded, frames, cores, out=out)
            times.append(time.perf_counter() - t0)
        dt = min(times)
        res["all_cores"] = {"value": got.size / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "seconds": dt,
                            "seconds_each_of_three": times, "machine_cpus": os.cpu_count(), "cpus_in_affinity_mask": allowed, "cgroup_cpu_quota": quota}
    return res


def pmc_summary(workload, build_id):
    """Per-dispatch PMC means committed under profiles/ for this workload, latest round first: (dict, file name, note).  A summary
    is quoted only when it is STAMPED with the source id of the library that is loaded now (tools/pmc_passes.sh writes
    ClownResamplerAMD_BuildId() into it): counters of another build say nothing about this one's kernels."""
    import glob
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s_pmc_summary.txt" % workload)), reverse=True):
        if os.path.exists(path):
            vals, stamp = {}, None
            for ln in open(path):
                f = ln.split()
                if len(f) > 3 and f[1] == "per-dispatch":
                    vals[f[0]] = float(f[3])
                elif len(f) == 2 and f[0] == "library_source_id":
                    stamp = f[1]
            if stamp == build_id:
                return vals, os.path.basename(path), None
            stale = stale or "profiles/%s is of library build %s, the loaded library is %s: its counters are not quoted" % (os.path.basename(path), stamp or "(unstamped)", build_id)
    return None, None, stale or "no PMC summary for this workload under profiles/"


def host_pcm(frames, ch, R, seed=12345):
    """Full-scale white int16 PCM in HOST memory, R zero frames of padding each side (clownresampler.h:725-733)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    pcm = np.zeros((frames + 2 * R) * ch, dtype=np.int16)
    pcm[R * ch:(R + frames) * ch] = rng.integers(-32768, 32768, size=frames * ch, dtype=np.int16)
    return pcm


def end_to_end(api, pre, radius, ch, rates, frames, check=True):
    """"separately report end-to-end (H2D + kernel + D2H)".  ONE ClownResampler_LowLevel_ResampleBulk over the whole
    workload from HOST memory - upload, kernel, download, synchronise: what a client of the host-pointer entry point gets.
    Pageable numpy buffers (first call: the runtime pins the ranges; then warm), and registered (pinned) ones.  Never `value`."""
    import numpy as np
    import torch
    import clownresampler_amd as cr
    fresh = api.LowLevel_State()
    assert api.LowLevel_Init(fresh, ch, *rates)
    R = fresh.lowest_level.integer_stretched_kernel_radius
    n_out = int(api.CountOutputFrames(fresh, frames))
    pcm = host_pcm(frames, ch, R)
    out = np.zeros((n_out + 1) * ch, dtype=np.int32)   # pre-faulted

    def once(src, dst):
        st = cr.LowLevel_State.from_buffer_copy(fresh)
        t0 = time.perf_counter()
        got, left, ran_out = api.LowLevel_ResampleBulk(st, pre, src, frames, n_out + 1, output=dst)
        dt = (time.perf_counter() - t0) * 1e3
        assert got.size == n_out * ch and left == 0 and ran_out == 1
        return dt

    pageable = [once(pcm, out) for _ in range(4)]
    pin_in = torch.empty(pcm.size, dtype=torch.int16).pin_memory()
    pin_out = torch.empty(out.size, dtype=torch.int32).pin_memory()
    pin_in.numpy()[:] = pcm
    pinned = [once(pin_in.numpy(), pin_out.numpy()) for _ in range(4)]
    same = bool(np.array_equal(pin_out.numpy()[: n_out * ch], out[: n_out * ch]))
    res = {"call": "one ClownResampler_LowLevel_ResampleBulk, %d -> %d frames x %d ch, host int16 in / host int32 out (upload + kernel + download + synchronise)" % (frames, n_out, ch),
           "pageable_first_ms": pageable[0], "pageable_ms": min(pageable[1:]), "pinned_ms": min(pinned[1:]),
           "Msamples/s": {"pageable": n_out * ch / (min(pageable[1:]) * 1e-3) / 1e6, "pinned": n_out * ch / (min(pinned[1:]) * 1e-3) / 1e6},
           "host_bytes": {"in": int(pcm.nbytes), "out": int(n_out * ch * 4)}, "pinned_equals_pageable": same}
    if check:
        import _checkers as ck
        o = ck.oracle(radius)
        ok, ost = o.low_init(ch, *rates)
        n_chk = min(50000, n_out)
        need = int(n_chk * fresh.increment / 65536) + 2 * R + 8
        want, _, _ = o.low_resample_i32(ost, pcm[: (need + 2 * R) * ch], need, capacity=n_chk)
        res["head_matches_oracle"] = bool(np.array_equal(out[: n_chk * ch], want[: n_chk * ch]))
    return res, pcm, out, n_out


def callback_api(api, pre, ch, rates, frames, pcm, expected, n_out):
    """The reference's own signature: ClownResampler_LowLevel_Resample(state, table, host PCM, &frames, CALLBACK, user) with a C
    callback that stores every frame as int32 (tools/cb_store.c - the callback the CPU baseline's reference leg uses)."""
    import ctypes as C
    import numpy as np
    import clownresampler_amd as cr
    path = os.path.join(ROOT, "tools", "bin", "libcr_cbstore.so")
    if not os.path.exists(path):
        return {"error": "tools/bin/libcr_cbstore.so not built (make -C clownresampler_amd/csrc tools)"}
    lib = C.CDLL(path)

    class Store(C.Structure):
        _fields_ = [("out", C.POINTER(C.c_int)), ("at", C.c_size_t), ("capacity", C.c_size_t)]

    out = np.zeros((n_out + 1) * ch, dtype=np.int32)
    cb = C.cast(lib.cr_store_frame, cr.OutputCallback)
    times = []
    for _ in range(3):
        st = api.LowLevel_State()
        assert api.LowLevel_Init(st, ch, *rates)
        store = Store(out.ctypes.data_as(C.POINTER(C.c_int)), 0, out.size)
        left = C.c_size_t(frames)
        t0 = time.perf_counter()
        r = api._LowResample(C.byref(st), C.byref(pre), pcm.ctypes.data_as(C.POINTER(cr.cc_s16l)), C.byref(left), cb, C.cast(C.pointer(store), C.c_void_p))
        times.append((time.perf_counter() - t0) * 1e3)
        assert r and left.value == 0 and store.at == n_out * ch
    return {"call": "ClownResampler_LowLevel_Resample with a C output callback storing int32, %d -> %d frames x %d ch from host memory" % (frames, n_out, ch),
            "ms": min(times), "Msamples/s": n_out * ch / (min(times) * 1e-3) / 1e6,
            "equals_bulk_output": bool(np.array_equal(out[: n_out * ch], expected[: n_out * ch]))}


def full_stream_check(radius, ch, rates, shard, d_pcm, d_out, s16):
    """Every output sample of one launch against the oracle: the launch's input is copied back from the device, the all-core
    oracle driver (oracle/cr_oracle.c, one state per host thread from the closed form) runs the shard from the shard's state,
    and the whole output is compared.  Returns (ok, detail)."""
    import numpy as np
    import _checkers as ck
    t0 = time.perf_counter()
    o = ck.oracle(radius)
    ok, ost = o.low_init(ch, *rates)
    ost.pos_int, ost.pos_frac = int(shard.state.position_integer), int(shard.state.position_fractional)
    host_in = d_pcm.cpu().numpy()
    cores = os.cpu_count() or 1
    want = o.low_resample_i32_mt(ost, host_in, int(shard.input_frames), cores)
    n = int(shard.output_frames) * ch
    if want.size < n:
        return False, {"error": "the oracle produced %d samples, the launch %d" % (want.size, n)}
    want = want[:n]    # (a shard is ended by its output capacity: clownresampler_amd.h, ClownResamplerAMD_Shard)
    if s16:
        want = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
    got = d_out[:n].cpu().numpy()
    same = bool(np.array_equal(got, want))
    detail = {"frames": int(shard.output_frames), "samples": n, "oracle_threads": cores, "seconds": time.perf_counter() - t0,
              "stream_hash": "%016x" % ck.stream_hash(got.astype(np.int32))}
    if not same:
        bad = np.flatnonzero(got != want)
        detail["first_difference_at_sample"] = int(bad[0])
        detail["differing_samples"] = int(bad.size)
    return same, detail


def single_gpu_reference(api, pre, cr, name, device, stream, steps, warmup, check, prewarm_ms=250.0):
    """The N = 1 point of a multi-GPU curve, measured in THIS process on THIS rank's GPU: the whole stream of workload `name`
    device-resident, K launches between one event pair after W warm-up launches (the timed region's own method), and - check -
    every sample of the last launch against the all-core oracle.  Buffers are rotated only when one set is smaller than a GiB
    (a larger one cannot live in the 256 MiB Infinity Cache)."""
    import torch
    radius, ch, rates, frames = WORKLOADS[name]
    whole = api.LowLevel_State()
    assert api.LowLevel_Init(whole, ch, *rates)
    R = whole.lowest_level.integer_stretched_kernel_radius
    shard = api.PlanShard(whole, frames, 0, 1)
    plan = api.PlanCreate(whole, pre)
    set_bytes = (shard.input_frames + 2 * R) * ch * 2 + shard.output_frames * ch * 4
    sets = []
    for k in range(1 if set_bytes >= (1 << 30) else 3):
        sets.append((stream_slice(-R, shard.input_frames + 2 * R, ch, frames, k * 7919, device),
                     torch.empty(shard.output_frames * ch, dtype=torch.int32, device=device)))

    def step(i):
        pcm, out = sets[i % len(sets)]
        st = cr.LowLevel_State.from_buffer_copy(shard.state)
        n = api.ResampleDevice(plan, st, pcm.data_ptr(), shard.input_frames, out.data_ptr(), shard.output_frames, stream.cuda_stream)[0]
        assert n == shard.output_frames

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(stream):
        # clock ramp, as for the timed region proper: the GPU has been idle while the host checked the main workload's output
        t_pre, i_pre = time.perf_counter(), 0
        while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
            for _ in range(4):
                step(i_pre)
                i_pre += 1
            torch.cuda.synchronize(device)
        for i in range(max(1, warmup)):
            step(i)
        torch.cuda.synchronize(device)
        ev0.record(stream)
        for i in range(steps):
            step(i)
        ev1.record(stream)
    torch.cuda.synchronize(device)
    ms = ev0.elapsed_time(ev1) / steps
    nbytes = shard.input_frames * ch * 2 + shard.output_frames * ch * 4
    res = {"workload": name, "what": "the WHOLE stream (%d -> %d frames x %d ch) as one launch on this GPU, %d launches between one event pair after %.0f ms of untimed launches and %d warm-up launches, %d buffer set(s)"
                                     % (shard.input_frames, shard.output_frames, ch, steps, prewarm_ms, max(1, warmup), len(sets)),
           "ms_per_step": ms, "value": shard.output_frames * ch / (ms * 1e-3) / 1e6, "unit": "Msamples/s",
           "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}
    if check:
        last = (steps - 1) % len(sets)
        ok, detail = full_stream_check(radius, ch, rates, shard, sets[last][0], sets[last][1], False)
        res["parity_full_stream"] = ok
        res["parity_full_stream_detail"] = detail
        if not ok:
            raise SystemExit("bench: the single-GPU run of %s differs from the oracle (%s) - numbers void" % (name, detail))
    del sets
    torch.cuda.empty_cache()
    return res


def percentile(sorted_values, q):
    if not sorted_values:
        return None
    i = min(len(sorted_values) - 1, max(0, int(round(q * (len(sorted_values) - 1)))))
    return sorted_values[i]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--lead-in-ms", type=float, default=6.0, help="untimed launches enqueued right in front of the K timed ones (same stream, no synchronisation in between), at least this much device time: the timed window then starts with the queue full and the clocks settled")
    ap.add_argument("--prewarm-ms", type=float, default=250.0, help="untimed launches before the W warmup steps, until this much wall time has passed: the chip needs ~100 ms of load to leave its idle clocks")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: cfg2 at N=1, cfg5 (the 1-hour stream, sharded) at N>1")
    ap.add_argument("--scaling", default=None, choices=("strong", "weak"), help="N>1: strong = ONE stream of the workload's length split over the ranks (default); weak = every rank one stream-length shard of an N times longer stream")
    ap.add_argument("--sets", type=int, default=3, help="rotating buffer sets (defeats the 256 MiB Infinity Cache)")
    ap.add_argument("--rank-timeout", type=float, default=900.0, help="wall-clock limit per rank in seconds: a rank still running then exits with status 124 (N > 1: so does the launcher, 60 s later, after killing its children)")
    ap.add_argument("--no-n1-reference", action="store_true", help="N > 1: skip the same-process single-GPU run of the whole stream behind efficiency_vs_n1; N = 1: skip strong_curve_n1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-host-paths", action="store_true", help="skip the end_to_end / callback_api measurements (N = 1; outside the timed region either way)")
    ap.add_argument("--s16", action="store_true", help="opt-in extension: clamped int16 output (NOT the BASELINE metric; writes 2 B per sample instead of 4)")
    ap.add_argument("--graph", action="store_true", help="time one hipGraph replay of the K steps instead of eager launches (measured SLOWER on ROCm 7.2 for this kernel)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_workers(args.gpus, args.rank_timeout))

    import numpy as np
    import torch
    import clownresampler_amd as cr

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    watchdog = arm_watchdog(args.rank_timeout, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the GPU: the product has no CPU path")
    # Fewer GPUs than ranks (a 1-GPU box asked for --gpus 8), or CRA_BENCH_BACKEND=gloo: VALIDATION mode - the ranks share the
    # GPUs there are and synchronise over gloo; everything but RCCL itself is exercised, and the line says so (its timings
    # are those of ranks contending for one GPU).  The driver's multi-GPU runs have one GPU per rank and use nccl = RCCL.
    n_dev = torch.cuda.device_count()
    backend = os.environ.get("CRA_BENCH_BACKEND", "nccl" if n_dev >= world else "gloo")
    shared_gpus = backend != "nccl"
    if shared_gpus:
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    workload = args.workload or ("cfg2" if world == 1 else "cfg5")
    scaling = args.scaling or "strong"
    radius, ch, rates, frames = WORKLOADS[workload]
    api = cr.load(radius)
    api.SetDevice(local_rank)
    pre = api.precomputed()
    whole = api.LowLevel_State()
    assert api.LowLevel_Init(whole, ch, *rates)
    R = whole.lowest_level.integer_stretched_kernel_radius
    total_frames = frames * world if (scaling == "weak" and world > 1) else frames
    shard = api.PlanShard(whole, total_frames, rank, world)
    plan = api.PlanCreate(whole, pre)
    info = api.PlanGetInfo(plan)
    out_frames_all = int(api.CountOutputFrames(whole, total_frames))
    out_samples_all = out_frames_all * ch

    # this rank's slice of the stream + halo; logical frame f of the stream is padded frame f + R; the zero padding of the
    # stream's two ends is materialised only where a shard touches it
    in_frames = shard.input_frames + 2 * R
    first_logical = shard.first_input_frame - R
    sets = []
    for s in range(max(1, args.sets)):
        pcm = stream_slice(first_logical, in_frames, ch, total_frames, s * 7919, device)
        out = torch.empty(max(1, shard.output_frames) * ch, dtype=torch.int16 if args.s16 else torch.int32, device=device)
        sets.append((pcm, out))
    stream = torch.cuda.current_stream(device)

    def step(i):
        pcm, out = sets[i % len(sets)]
        st = cr.LowLevel_State.from_buffer_copy(shard.state)
        n = api.ResampleDevice(plan, st, pcm.data_ptr(), shard.input_frames, out.data_ptr(), shard.output_frames, stream.cuda_stream, s16=args.s16)[0]
        assert n == shard.output_frames
        return n

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    def host_tensor_max(values):
        if world == 1:
            return [float(v) for v in values]
        t = torch.tensor(values, dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]

    # Clock ramp: untimed launches until --prewarm-ms have passed (then the W warmup steps, then the K timed ones).
    t_pre = time.perf_counter()
    i_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(20):
            step(i_pre)
            i_pre += 1
        torch.cuda.synchronize(device)
    est_launch_ms = (time.perf_counter() - t_pre) * 1e3 / max(i_pre, 1)   # (an upper estimate: it includes the synchronisations)

    # Optional: the K timed steps captured once into a hipGraph and replayed.  Eager launches keep the queue full here
    # (the host needs ~15 us per launch, the kernel 65+), and the graph replay measured slower, so eager is the default.
    graph = None
    if args.graph:
        try:
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                stream = side
                for i in range(min(2, args.warmup)):
                    step(i)
                torch.cuda.synchronize(device)
                graph = torch.cuda.CUDAGraph()
                capture_counts_before = [api.LaunchCount(k) for k in range(9)]
                with torch.cuda.graph(graph, stream=side):
                    stream = torch.cuda.current_stream(device)
                    for i in range(args.steps):
                        step(i)
                # what the CAPTURE enqueued, by kernel: a replay runs exactly these (not a made-up count)
                captured_by_kernel = [api.LaunchCount(k) - capture_counts_before[k] for k in range(9)]
            stream = side
        except Exception as e:  # capture not possible on this stack: fall back to eager launches
            if rank == 0:
                print("bench: hipGraph capture failed (%s); timing eager launches" % e, file=sys.stderr)
            graph = None
            stream = torch.cuda.current_stream(device)

    # (1) The contract's timed region: EXACTLY K steps between one event pair on the launch stream, barrier + synchronize on
    #     both sides -> ms_per_step (mean of the K launches) and `value`.
    # (2) A second region for the DISTRIBUTION: 10 blocks of 20 more launches (fixed, whatever K is), an event between blocks
    #     -> median / min / p95 of the per-block mean launch time.  Not an event around every launch: an event record between
    #     two launches costs ~3 us of marker packet on this stack (measured: 67.6 us per launch with, 64.7 us without,
    #     profiles/r02_event_overhead.log), i.e. it would be measuring itself.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # The K timed launches follow LEAD-IN launches that are already in flight when ev0 is recorded (>= --lead-in-ms of them, untimed, inside
    # the same barrier bracket): straight after a barrier the chip has idled, and a 1.2 ms window (the driver's --steps 20 on cfg 2) caught
    # it 2-3 % above the sustained figure (mean of the first window 60.7 us against 62.6 us sustained on the same box).
    # With the queue kept full across ev0 the timed K launches run at the sustained clocks; `value` is that, and the rocprofv3 average
    # over the timed dispatches (tools/r05_final_profiles.sh) reproduces it.
    lead_in_launches = [0]

    def run_steps():
        if graph is None and args.lead_in_ms > 0:
            n_lead = max(20, int(args.lead_in_ms / max(est_launch_ms, 1e-3)) + 1)
            for i in range(n_lead):
                step(i)
            lead_in_launches[0] = n_lead
        ev0.record(stream)
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)
        ev1.record(stream)

    if graph is not None:
        with torch.cuda.stream(stream):
            for _ in range(max(1, args.warmup // max(1, args.steps))):
                graph.replay()
    else:
        for i in range(args.warmup):
            step(i)
    barrier()
    counts_before = [api.LaunchCount(k) for k in range(9)]
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        run_steps()
    barrier()
    wall = time.perf_counter() - t0
    launches_by_kernel = [api.LaunchCount(k) - counts_before[k] for k in range(9)]   # what the bracketed region ran: lead-in + the K timed launches (graph replays launch nothing new)
    if graph is not None:
        launches_by_kernel = captured_by_kernel   # (one replay of the captured K steps: the launches the capture recorded)
    dev_ms = max(ev0.elapsed_time(ev1), 0.0)
    mean_ms = dev_ms / args.steps
    last_set = (args.steps - 1) % len(sets)
    out_snapshot = None if args.no_check else sets[last_set][1].clone()   # what the LAST TIMED launch wrote (checked in full below)

    blocks, block_len = 10, 20   # fixed, whatever --steps is: with the driver's --steps 20 a K/10 rule gave 2-launch blocks (+-10 %)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
    with torch.cuda.stream(stream):
        # (the same lead-in as the timed region: after the barrier above - and the copy of the last timed output - the memory clocks have
        # dropped, and a block that starts cold reads 2 % above the sustained figure on the memory-bound workloads)
        for i in range(lead_in_launches[0]):
            step(i)
        marks[0].record(stream)
        for b in range(blocks):
            for i in range(block_len):
                step(b * block_len + i)
            marks[b + 1].record(stream)
    barrier()
    each = sorted(marks[b].elapsed_time(marks[b + 1]) / block_len for b in range(blocks))
    median_ms, min_ms, p95_ms, max_ms = percentile(each, 0.5), each[0], percentile(each, 0.95), each[-1]
    # max over ranks of each rank's statistic
    mean_all, median_all, wall_ms = host_tensor_max([mean_ms, median_ms, wall * 1e3])
    ms_per_step = mean_all
    value = out_samples_all / (ms_per_step * 1e-3) / 1e6

    # the dominant (only) kernel: algorithmic bytes of THIS rank's launch / its launch duration
    launch_bytes = shard.input_frames * ch * 2 + shard.output_frames * ch * (2 if args.s16 else 4)
    achieved = launch_bytes / (mean_ms * 1e-3) / 1e9
    # (a k_up plan sends launches of fewer than brief_below output frames to its other kernel)
    shard_state = cr.LowLevel_State.from_buffer_copy(shard.state)
    ran = api.PlanKernelAt(plan, shard_state.position_fractional)   # (5 = k_int: whole-number ratios, decided per launch by its fraction)
    if ran != 5:
        ran = info.brief_kernel if (info.kernel == 3 and shard.output_frames < info.brief_below) else info.kernel
    kernel_name = {1: "k_poly<%d,%d>", 2: "k_wave<%d,%d>", 3: "k_up2<%d,%d>" if ch == 2 else "k_up<%d,%d>",   # (stereo: k_up2 - variant 27, the default, with the FP32 round-toward-zero chain, 26 with the integer chain; other channel counts k_up)
                   4: "k_wave2<%d,%d>", 5: "k_int<%d,%d>", 6: "k_wave2s<%d,%d>"}[ran] % (ch, info.slots) if ran else "k_generic"
    # a long mono launch runs on the STEREO instance of its configuration (dual mono: frames j and j + H as the two channels)
    dual = api.PlanDualMonoKernel(plan) if (ch == 1 and not args.s16 and ran != 5 and os.environ.get("CLOWNRESAMPLER_AMD_NO_DUAL_MONO") is None) else 0
    if dual and launches_by_kernel[dual] >= args.steps:
        kernel_name = "%s<2,%d> as dual mono (mono plan: %s)" % ({1: "k_poly", 4: "k_wave2"}[dual], info.slots, kernel_name)
        ran = dual
    # long launches of k_up2's shape run on k_seg (the lanes of a wave on frames of equal fraction, the row in scalar registers): the launch counters say
    if ran in (3, 4) and launches_by_kernel[8] >= args.steps:
        kernel_name = "k_seg<%d,%d>" % (ch, info.slots)
        ran = 8
    if launches_by_kernel[ran] < args.steps:
        raise SystemExit("bench: expected the timed launches on kernel %d (%s); launch counters say %s" % (ran, kernel_name, launches_by_kernel))
    pmc, pmc_file, traffic_note = pmc_summary(workload, api.BuildId()) if (world == 1 and not args.s16) else (None, None, "N > 1 / int16 output: no PMC summary applies")
    traffic = None
    if pmc and "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # KiB units; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads: x2 (MI355X_MICROARCH.md, HBM)
        traffic = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
        traffic_note = ("NOT measured in this run: bytes per launch from an earlier rocprofv3 --pmc run of this command (FETCH_SIZE x2 gfx950 "
                        "correction + WRITE_SIZE, separate passes), committed as profiles/" + pmc_file)
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": traffic_note, "kernel": kernel_name,
                "algorithmic_bytes_per_launch": launch_bytes, "launch_ms": mean_ms, "launch_ms_is": "mean of the %d timed launches (one HIP event pair on the launch stream)" % args.steps,
                "read_only_GBs": shard.input_frames * ch * 2 / (mean_ms * 1e-3) / 1e9}
    roofline_valu = None
    if pmc and "SQ_ACTIVE_INST_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        # Which ceiling binds?  VALU busy = the cycles the SIMDs spent issuing VALU instructions over the cycles the launch
        # lasted, both from PMC counters of a profiled run of this same command (rocprofv3's VALUBusy: SQ_ACTIVE_INST_VALU counts
        # in units of 4 cycles and is summed over the 1,024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs).
        busy = pmc["SQ_ACTIVE_INST_VALU"] * 4.0 / SIMDS
        available = pmc["GRBM_GUI_ACTIVE"] / 8.0
        insts = pmc.get("SQ_INSTS_VALU", 0.0)
        roofline_valu = {"bound": "valu", "achieved": busy, "peak": available, "unit": "cycles per SIMD per launch (VALU issuing / elapsed)", "frac": busy / available,
                         "valu_wave_instructions_per_launch": insts,
                         "valu_per_tap_channel": insts * 64.0 / (shard.output_frames * ch * max(1, info.slots)) if insts else None,
                         "lds_bank_conflict_share": (pmc["SQ_LDS_BANK_CONFLICT"] / pmc["SQ_LDS_IDX_ACTIVE"]) if pmc.get("SQ_LDS_IDX_ACTIVE") else None,
                         "note": "NOT measured in this run: SQ_ACTIVE_INST_VALU x 4 / %d SIMDs against GRBM_GUI_ACTIVE / 8 XCDs, per-dispatch means of profiles/%s" % (SIMDS, pmc_file)}
        if roofline_valu["frac"] > roofline["frac"]:
            roofline["binding"] = "valu (see roofline_valu): the SIMDs issue VALU instructions for %.0f %% of the launch; the HBM figure above is what that leaves" % (100 * roofline_valu["frac"])

    def oracle_window(first_out, n_chk, salt):
        """Frames [first_out, first_out + n_chk) of the WHOLE stream's output from the oracle: closed-form state at first_out
        (clownresampler.h:1076-1078 applied first_out times), input slice regenerated from the stream function."""
        import _checkers as ck
        o = ck.oracle(radius)
        st = cr.LowLevel_State.from_buffer_copy(whole)
        api.AdvanceState(st, first_out)
        pi, pf = st.position_integer, st.position_fractional
        need = int(n_chk * whole.increment / 65536) + 4 * R + 8
        host_in = stream_slice(pi - R, need, ch, total_frames, salt, device).cpu().numpy()
        ok, ost = o.low_init(ch, *rates)
        ost.pos_int, ost.pos_frac = 0, pf
        want, _, _ = o.low_resample_i32(ost, host_in, need - 2 * R, capacity=n_chk)
        return want

    # correctness of what was just timed: EVERY rank checks its WHOLE shard - every sample the last timed launch wrote (snapshot taken
    # right behind the timed region, before the distribution region's launches) - against the all-core oracle run over the very input
    # the launch read, copied back from the device.  Outside the timed region.
    check, check_detail = None, None
    if not args.no_check:
        ok_here, check_detail = full_stream_check(radius, ch, rates, shard, sets[last_set][0], out_snapshot, args.s16)
        del out_snapshot
        if world > 1:
            t = torch.tensor([1 if ok_here else 0])
            tt = t.to(device) if backend == "nccl" else t
            dist.all_reduce(tt, op=dist.ReduceOp.MIN)
            check = bool(int(tt.cpu()[0]))
        else:
            check = ok_here
        if not check:
            raise SystemExit("bench: device output differs from the oracle on rank %d (ok here: %s, %s) - numbers void" % (rank, ok_here, check_detail))

    # The N = 1 point of the strong-scaling curve, self-contained in every line.  N = 1 (default workload): the hour of configs[4] as
    # one launch beside the 10-minute headline.  N > 1: every rank runs the WHOLE stream of this job's workload alone on its own GPU
    # (same process, after and outside the timed region), and efficiency_vs_n1 compares the job with the slowest of those.
    n1_ref, strong_n1 = None, None
    if not args.no_n1_reference and not args.s16:
        if world == 1 and workload == "cfg2":
            strong_n1 = single_gpu_reference(api, pre, cr, "cfg5", device, stream, 20, 5, not args.no_check, args.prewarm_ms)
        elif world > 1:
            del sets[1:]   # (the gather below reads set 0 only)
            torch.cuda.empty_cache()
            barrier()
            n1_ref = single_gpu_reference(api, pre, cr, workload, device, stream, max(5, min(args.steps, 20)), 5, False, args.prewarm_ms)
            barrier()

    # N > 1: the line proves what ran.  Every rank reports its GPU's identity and its own timing; the backend is asked to SUM a
    # one per rank (a collective that only comes out at N if N ranks took part in it).
    per_rank, backend_sum = None, None
    if world > 1:
        props = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "local_rank": local_rank,
                "device": {"name": props.name, "uuid": str(getattr(props, "uuid", "")),
                           "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))},
                "ms_per_step": mean_ms, "median_ms": median_ms, "output_frames": int(shard.output_frames), "input_frames": int(shard.input_frames),
                "n1_ms_per_step": n1_ref["ms_per_step"] if n1_ref else None,
                "first_output_frame": int(shard.first_output_frame), "kernel_launches_timed": [int(v) for v in launches_by_kernel]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        one = torch.ones(1, dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        backend_sum = int(one.cpu()[0])

    gather = None
    if world > 1 and not args.s16 and not args.no_gather:
        # The final concatenate (north_star): the ranks' int32 shards, padded to the common per-rank size
        # PlanShard uses, (a) gathered to rank 0 - ncclGather semantics, RCCL send/recv: each peer -> root transfer rides its
        # own xGMI link - and (b) all-gathered (every rank gets the stream).  Timed apart from the kernel; a failure here must
        # not cost the run its bench line.
        try:
            per = (out_frames_all + world - 1) // world * ch
            on_device = backend == "nccl"
            src = sets[0][1][: shard.output_frames * ch]
            send = torch.zeros(per, dtype=torch.int32, device=device if on_device else "cpu")
            send[: shard.output_frames * ch] = src if on_device else src.cpu()
            recv = torch.empty(per * world, dtype=torch.int32, device=send.device)  # (root's gather target; everyone's all-gather target)
            pieces = list(recv.split(per)) if rank == 0 else None
            reps = 5 if on_device else 1

            def timed(fn):
                fn()
                barrier()
                ts = []
                for _ in range(reps):
                    g0 = time.perf_counter()
                    fn()
                    barrier()
                    ts.append((time.perf_counter() - g0) * 1e3)
                return host_tensor_max([sorted(ts)[len(ts) // 2]])[0]

            root_ms = timed(lambda: dist.gather(send, pieces, dst=0))
            # what arrived is the one-shot stream (PlanShard's blocks are `per` frames each, so the pieces are contiguous):
            # the root checks a window across every shard boundary against the oracle
            seams_ok = None
            if rank == 0 and not args.no_check:
                seams_ok = True
                half = 2000
                for r in range(1, world):
                    b = api.PlanShard(whole, total_frames, r, world).first_output_frame
                    if b < half or b + half > out_frames_all:
                        continue
                    want = oracle_window(b - half, 2 * half, 0)
                    got = recv[(b - half) * ch: (b + half) * ch].cpu().numpy()
                    seams_ok = seams_ok and bool(np.array_equal(got, want))
            all_ms = timed(lambda: dist.all_gather_into_tensor(recv, send))
            gather = {"to_root": {"collective": "gather (RCCL send/recv, ncclGather semantics)" if on_device else "gather (gloo, host copies: validation only)", "ms": root_ms},
                      "all_gather": {"collective": "all_gather_into_tensor (RCCL)" if on_device else "all_gather_into_tensor (gloo, host copies: validation only)", "ms": all_ms},
                      "ms": root_ms, "bytes_per_rank": per * 4, "seams_match_oracle": seams_ok,
                      # what one peer -> root transfer moved per second while all of them ran (each over its own xGMI link with one GPU per rank)
                      "link_GBs": per * 4 / (root_ms * 1e-3) / 1e9, "all_gather_link_GBs": per * 4 * (world - 1) / (all_ms * 1e-3) / 1e9,
                      "value_with_gather": out_samples_all / ((ms_per_step + root_ms) * 1e-3) / 1e6,
                      "value_with_all_gather": out_samples_all / ((ms_per_step + all_ms) * 1e-3) / 1e6}
            if seams_ok is False:
                raise SystemExit("bench: gathered stream differs from the oracle at a shard boundary - numbers void")
        except SystemExit:
            raise
        except Exception as e:
            gather = {"error": str(e)[:300]}

    if rank == 0:
        named = CONFIG_NAMES.get(workload)
        line = {
            "metric": ("output Msamples/s at 44.1->48 kHz stereo" if workload in ("cfg2", "cfg5") else "output Msamples/s (%s)" % workload) + (" [int16-clamped output extension]" if args.s16 else ""),
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "s16 in / int32 16.16 fixed-point arithmetic / int32 out", "data": "synthetic",
            "config": {"workload": "%s%s: %d-ch int16 %d->%d Hz, %d-lobe Lanczos, ONE stream of %d -> %d frames%s, device-resident, %d rotating buffer sets"
                                   % (workload, " (%s)" % named if named else "", ch, rates[0], rates[1], radius, total_frames, out_frames_all,
                                      " split over %d ranks (%s scaling: %d input frames per rank)" % (world, scaling, shard.input_frames) if world > 1 else "", len(sets)),
                       "sharding": "output timeline split in %d contiguous blocks (ClownResamplerAMD_PlanShard), input halo of %d frames replicated, no data-path collective" % (world, R),
                       "plan": info.asdict()},
            "ms_per_step_is": "max over ranks of (one HIP event pair around the K launches on the launch stream) / K; %d untimed lead-in launches are in flight on that stream when the first event is recorded (--lead-in-ms %g)" % (lead_in_launches[0], args.lead_in_ms),
            "launch_us": {"median": median_ms * 1e3, "min": min_ms * 1e3, "p95": p95_ms * 1e3, "max": max_ms * 1e3, "mean_of_timed_region": mean_ms * 1e3,
                          "of": "rank 0; median / min / p95 / max over %d blocks of %d further launches each (per-block mean; an event per launch would add ~3 us to each)" % (blocks, block_len)},
            "value_at_median": out_samples_all / (median_all * 1e-3) / 1e6,
            "roofline": roofline,
            "launch_mode": "hipGraph replay of the K steps" if graph is not None else "eager",
            # which of this process's launches of the kernel the timed region was (in launch order, 0-based): tools/trace_timed_mean.py averages
            # exactly those rows of a rocprofv3 --kernel-trace of the same command, so that profiles/ reproduces ms_per_step
            "timed_dispatches": {"first": i_pre + args.warmup + lead_in_launches[0], "count": args.steps},
            "wall_ms_per_launch": wall_ms / (args.steps + lead_in_launches[0]),
            "wall_ms_per_step": wall_ms / (args.steps + lead_in_launches[0]),   # (the key's name until round 4; the same figure)
            "parity_full_stream": check,
            "parity_full_stream_is": "every rank: EVERY sample its last timed launch wrote == the all-core oracle over the input that launch read (copied back from the device); all ranks agree",
            "parity_full_stream_detail": check_detail,
        }
        if roofline_valu:
            line["roofline_valu"] = roofline_valu
        if world > 1:
            line["backend"] = "nccl (RCCL), one GPU per rank" if backend == "nccl" else \
                "%s; VALIDATION ONLY: %d ranks share %d GPU(s), timings are those of ranks contending for a GPU" % (backend, world, n_dev)
            line["per_rank"] = per_rank
            line["distinct_devices"] = len({(r["device"]["uuid"], r["device"]["pci"]) for r in per_rank})
            line["world_size_seen_by_backend"] = {"get_world_size": dist.get_world_size(), "all_reduce_sum_of_ones": backend_sum,
                                                  "backend": dist.get_backend()}
            line["workload_note"] = ("N > 1 default = BASELINE configs[4] (the 1-hour stream) split over the ranks, STRONG scaling: each rank's launch is 1/N of the hour "
                                     "(%d output frames here); the N = 1 point of THIS curve is `--gpus 1 --workload cfg5` (the default N = 1 line is configs[1], 10 minutes); "
                                     "with the concatenate the curve is bound by the gather (SURVEY 8(e): bytes_per_rank over one xGMI link against a kernel of tens of microseconds)"
                                     % shard.output_frames) if workload == "cfg5" else None
            if n1_ref:
                # strong: N ranks share ONE stream, ideal time = t1 / N; weak: every rank has a stream of the N = 1 size, ideal = t1
                t1 = max(r["n1_ms_per_step"] for r in per_rank)
                ideal = t1 / world if scaling == "strong" else t1
                line["efficiency_vs_n1"] = {"value": ideal / ms_per_step, "n1_ms_per_step": t1, "n1_value": n1_ref["value"] if scaling == "strong" else n1_ref["value"] * world,
                                            "is": "(%s) / ms_per_step of this job; t1 = the slowest rank's own single-GPU run of %s, measured in this process after the timed region (kernel only, like `value`)"
                                                  % ("t1 / N" if scaling == "strong" else "t1", n1_ref["what"]),
                                            "n1_of_rank_0": n1_ref}
        if strong_n1:
            line["strong_curve_n1"] = strong_n1
        if gather:
            line["gather"] = gather
        if world == 1 and not args.no_host_paths and not args.s16:
            # what a client of the reference's host-pointer signatures gets, measured after and outside the timed region
            try:
                e2e, pcm_h, out_h, n_out_h = end_to_end(api, pre, radius, ch, rates, min(frames, 28800000), check=not args.no_check)
                line["end_to_end"] = e2e
                line["callback_api"] = callback_api(api, pre, ch, rates, min(frames, 28800000), pcm_h, out_h, n_out_h)
                del pcm_h, out_h
            except Exception as e:   # these are reports beside the metric: a failure here must not cost the run its line
                line["end_to_end"] = {"error": str(e)[:300]}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(radius, ch, rates, frames)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    watchdog.cancel()


if __name__ == "__main__":
    main()
