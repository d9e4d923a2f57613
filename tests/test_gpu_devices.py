"""GPU box: the library as a C client that owns several devices would drive it - per-device contexts, the one-call sharded
entry point (ClownResamplerAMD_ResampleShardedDevice) with its two concatenate forms, per-thread devices, launches captured
into hipGraphs, and caller-supplied tables that the 32-bit kernels must refuse.  The pool's boxes have ONE GPU: device lists
name ordinal 0 several times where more than one shard is wanted (every shard still has its own stream and buffers)."""
import copy
import ctypes as C
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import _checkers as ck
import _product
import clownresampler_amd as cr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def products():
    assert cr.load(3).DeviceCount() > 0, "these tests need the GPU: the library has no other path"
    return {3: _product.Product(3), 8: _product.Product(8)}


@pytest.mark.parametrize("radius,ch,rates,frames,nshards,s16", [
    (3, 2, (44100, 48000, 44100), 2646000, 3, False),     # BASELINE configs[4] at 1/60 size
    (3, 2, (44100, 48000, 44100), 100003, 8, True),
    (8, 2, (8000, 96000, 8000), 480000, 2, False),        # configs[2], one minute
    (3, 8, (48000, 44100, 44100), 288000, 4, False),      # configs[3], six seconds
    (3, 1, (44100, 8000, 8000), 70001, 5, False),
    (3, 2, (44100, 48000, 44100), 7, 4, False),           # fewer output frames than some shards would need: empty shards
])
def test_sharded_call_equals_one_shot(products, radius, ch, rates, frames, nshards, s16):
    """One call, `nshards` shards, each with its own input slice, output buffer and stream; concatenated on the root by peer
    copies: the oracle's one-shot stream, and the state ONE low-level call over the whole input leaves."""
    import torch
    p, o = products[radius], ck.oracle(radius)
    api = p.api
    dev = torch.device("cuda", 0)
    ok, st = p.low_init(ch, *rates)
    ok, ost = o.low_init(ch, *rates)
    R = int(ost.cfg.radius_frames)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch, 77), ch, R)
    want, left, ran_out = o.low_resample_i32(ost, padded, frames)
    total = want.size // ch
    unit = 2 if s16 else 4
    streams = [torch.cuda.Stream(dev) for _ in range(nshards)]
    keep, shards = [], []
    for r in range(nshards):
        sh = api.PlanShard(st.raw, frames, r, nshards)
        lo, hi = sh.first_input_frame, sh.first_input_frame + sh.input_frames + 2 * R
        d_in = torch.from_numpy(np.ascontiguousarray(padded[lo * ch: hi * ch])).to(dev) if sh.output_frames else torch.zeros(8, dtype=torch.int16, device=dev)
        d_out = torch.zeros(max(1, sh.output_frames) * ch, dtype=torch.int16 if s16 else torch.int32, device=dev)
        keep += [d_in, d_out]
        shards.append((0, d_in.data_ptr(), d_out.data_ptr(), streams[r].cuda_stream))
    root = torch.zeros(max(1, total) * ch, dtype=torch.int16 if s16 else torch.int32, device=dev)
    torch.cuda.synchronize()
    n = api.ResampleShardedDevice(st.raw, p.pre, frames, shards, s16=s16, gather_mode=cr.GATHER_PEER_COPY, root_shard=nshards - 1, root_output=root.data_ptr())
    assert api.ShardedSynchronize(shards) == 0
    assert n == total
    got = root.cpu().numpy()[: total * ch]
    expect = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16) if s16 else want
    assert np.array_equal(got, expect)
    assert st.astuple() == ost.astuple()
    assert unit in (2, 4)


def test_sharded_call_without_gather_leaves_blocks_in_place(products):
    import torch
    p, o = products[3], ck.oracle(3)
    api = p.api
    dev = torch.device("cuda", 0)
    ch, rates, frames = 2, (48000, 44100, 44100), 90001
    ok, st = p.low_init(ch, *rates)
    ok, ost = o.low_init(ch, *rates)
    R = int(ost.cfg.radius_frames)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch, 5), ch, R)
    want = o.low_resample_i32(ost, padded, frames)[0]
    whole_in = torch.from_numpy(padded).to(dev)
    whole_out = torch.zeros(want.size, dtype=torch.int32, device=dev)
    shards = []
    for r in range(4):
        sh = api.PlanShard(st.raw, frames, r, 4)
        # (shards may also simply point into one buffer each side when they live on one device)
        shards.append((0, whole_in.data_ptr() + sh.first_input_frame * ch * 2, whole_out.data_ptr() + sh.first_output_frame * ch * 4, None))
    torch.cuda.synchronize()
    n = api.ResampleShardedDevice(st.raw, p.pre, frames, shards)
    api.ShardedSynchronize(shards)
    assert n * ch == want.size and np.array_equal(whole_out.cpu().numpy(), want)


_RCCL_SNIPPET = r"""
import sys, numpy as np
sys.path[:0] = [%(root)r, %(tests)r]
import torch, _checkers as ck, _product, clownresampler_amd as cr
p, o = _product.Product(3), ck.oracle(3)
api = p.api
dev = torch.device("cuda", 0)
ch, rates, frames = 2, (44100, 48000, 44100), 200001
ok, st = p.low_init(ch, *rates); ok, ost = o.low_init(ch, *rates)
padded = ck.pad_frames(ck.noise_pcm(frames * ch, 3), ch, 3)
want = o.low_resample_i32(ost, padded, frames)[0]
d_in = torch.from_numpy(padded).to(dev)
d_out = torch.zeros(want.size, dtype=torch.int32, device=dev)
root = torch.zeros(want.size, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev)
torch.cuda.synchronize()
shards = [(0, d_in.data_ptr(), d_out.data_ptr(), s.cuda_stream)]
try:
    n = api.ResampleShardedDevice(st.raw, p.pre, frames, shards, gather_mode=cr.GATHER_RCCL, root_shard=0, root_output=root.data_ptr())
except cr.ClownResamplerError as e:
    print("RCCL-UNAVAILABLE", e); sys.exit(3)
api.ShardedSynchronize(shards)
assert n * ch == want.size
assert np.array_equal(root.cpu().numpy(), want), "gathered stream differs"
api.Shutdown()
print("RCCL-OK")
"""


def test_sharded_call_gathers_with_rccl():
    """ncclGather through librccl (loaded on first use) with the one rank this box has: the call path, the communicator set
    and the stream ordering behind the kernel.  In a child process with a time limit: a communicator that cannot be set up
    on this box must not take the test session with it."""
    code = _RCCL_SNIPPET % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        pytest.skip("RCCL communicator setup did not finish on this box")
    if r.returncode == 3:
        pytest.skip("librccl unusable on this box: " + r.stdout[-300:])
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_devices_keep_their_contexts(products):
    """SetDevice / SetThreadDevice select; they tear nothing down.  Plans survive, and a plan launches on ITS device."""
    p = products[3]
    api = p.api
    ok, st = p.low_init(2, 44100, 48000, 44100)
    plan = api.PlanCreate(st.raw, p.pre)
    count = api.PlanCacheCount()
    assert api.SetDevice(0) == 0 and api.SetThreadDevice(0) == 0 and api.GetDevice() == 0
    assert api.PlanCacheCount() == count
    assert api.PlanCreate(st.raw, p.pre) == plan
    with pytest.raises(cr.ClownResamplerError) as e:
        api.SetThreadDevice(api.DeviceCount())
    assert e.value.code == cr.ERROR_NO_DEVICE
    assert api.SetThreadDevice(-1) == 0
    # threads with their own device choice, working at the same time
    o = ck.oracle(3)
    ok, ost = o.low_init(2, 44100, 48000, 44100)
    padded = ck.pad_frames(ck.noise_pcm(2 * 300000, 21), 2, 3)
    want = o.low_resample_i32(ost, padded, 300000)[0]
    bad = []

    def work():
        api.SetThreadDevice(0)
        for _ in range(3):
            ok, s = p.low_init(2, 44100, 48000, 44100)
            got = p.low_resample_i32(s, padded, 300000)[0]
            if not np.array_equal(got, want):
                bad.append(1)

    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad


def test_launches_captured_into_a_graph(products):
    """More launches than a stream's ticket ring holds (64), captured into one hipGraph, replayed twice while another stream
    keeps launching: captured launches own never-recycled ticket blocks, so nothing collides."""
    import torch
    p, o = products[3], ck.oracle(3)
    api = p.api
    dev = torch.device("cuda", 0)
    ch, rates, frames = 2, (44100, 48000, 44100), 200000
    ok, ost = o.low_init(ch, *rates)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch, 31), ch, 3)
    want = o.low_resample_i32(ost, padded, frames)[0]
    d_in = torch.from_numpy(padded).to(dev)
    launches = 100
    outs = [torch.zeros(want.size, dtype=torch.int32, device=dev) for _ in range(launches)]
    side_out = [torch.zeros(want.size, dtype=torch.int32, device=dev) for _ in range(8)]
    ok, st0 = p.low_init(ch, *rates)
    plan = api.PlanCreate(st0.raw, p.pre)
    assert api.PlanGetInfo(plan).kernel in (1, 2)
    api.ReserveCaptureLaunches(launches)
    n_out = want.size // ch
    cap_stream, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cap_stream):
        with torch.cuda.graph(graph, stream=cap_stream):
            s = torch.cuda.current_stream(dev)
            for k in range(launches):
                st = cr.LowLevel_State.from_buffer_copy(st0.raw)
                api.ResampleDevice(plan, st, d_in.data_ptr(), frames, outs[k].data_ptr(), n_out + 8, s.cuda_stream)
    for rep in range(2):
        for t in outs:
            t.zero_()
        torch.cuda.synchronize()
        graph.replay()
        for k in range(8):
            st = cr.LowLevel_State.from_buffer_copy(st0.raw)
            api.ResampleDevice(plan, st, d_in.data_ptr(), frames, side_out[k].data_ptr(), n_out + 8, side.cuda_stream)
        torch.cuda.synchronize()
        for k in range(launches):
            assert np.array_equal(outs[k].cpu().numpy(), want), (rep, k)
        for k in range(8):
            assert np.array_equal(side_out[k].cpu().numpy(), want), (rep, "side", k)


def test_segment_kernel_on_streams_and_in_a_graph(products):
    """k_seg (round 5) through the paths its launches share with the other kernels: two streams launching at once (each launch its own ticket
    block), and launches captured into a hipGraph and replayed (never-recycled blocks from the capture pool) - forced, on a stream short
    enough for the test (the rule takes it for ten-minute launches)."""
    import torch
    p, o = products[8], ck.oracle(8)
    api = p.api
    dev = torch.device("cuda", 0)
    ch, rates, frames = 2, (8000, 64000, 8000), 60000
    ok, ost = o.low_init(ch, *rates)
    padded = ck.pad_frames(ck.noise_pcm(frames * ch, 77), ch, int(ost.cfg.radius_frames))
    want = o.low_resample_i32(ost, padded, frames)[0]
    n_out = want.size // ch
    d_in = torch.from_numpy(padded).to(dev)
    ok, st0 = p.low_init(ch, *rates)
    plan = api.PlanCreate(st0.raw, p.pre)
    api.DebugSegKernel(1)
    try:
        streams = [torch.cuda.Stream(dev) for _ in range(3)]
        outs = [torch.zeros(want.size + 64, dtype=torch.int32, device=dev) for _ in range(12)]
        before = api.LaunchCount(8)
        torch.cuda.synchronize()
        for k, t in enumerate(outs):
            st = cr.LowLevel_State.from_buffer_copy(st0.raw)
            api.ResampleDevice(plan, st, d_in.data_ptr(), frames, t.data_ptr(), n_out + 8, streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        assert api.LaunchCount(8) == before + len(outs)
        for k, t in enumerate(outs):
            assert np.array_equal(t.cpu().numpy()[:want.size], want), ("streams", k)
        launches = 10
        api.ReserveCaptureLaunches(launches)
        gouts = [torch.zeros(want.size + 64, dtype=torch.int32, device=dev) for _ in range(launches)]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(streams[0]):
            with torch.cuda.graph(graph, stream=streams[0]):
                s = torch.cuda.current_stream(dev)
                for k in range(launches):
                    st = cr.LowLevel_State.from_buffer_copy(st0.raw)
                    api.ResampleDevice(plan, st, d_in.data_ptr(), frames, gouts[k].data_ptr(), n_out + 8, s.cuda_stream)
        for rep in range(2):
            for t in gouts:
                t.zero_()
            torch.cuda.synchronize()
            graph.replay()
            st = cr.LowLevel_State.from_buffer_copy(st0.raw)
            api.ResampleDevice(plan, st, d_in.data_ptr(), frames, outs[0].data_ptr(), n_out + 8, streams[1].cuda_stream)
            torch.cuda.synchronize()
            for k in range(launches):
                assert np.array_equal(gouts[k].cpu().numpy()[:want.size], want), ("graph", rep, k)
            assert np.array_equal(outs[0].cpu().numpy()[:want.size], want)
    finally:
        api.DebugSegKernel(0)


@pytest.mark.parametrize("scale,expect_kernel", [(1, "fast"), (2, "generic"), (-1, "generic")])
def test_caller_supplied_tables(products, scale, expect_kernel):
    """Plans are keyed by table CONTENTS, so a caller's own table is legitimate input.  The 32-bit kernels multiply with the
    24-bit multiplier and keep 32 bits of the product: only for -65536 < weight <= 65536 is that the product.  A table scaled
    by 2 (peak 131072) or negated (-65536 at the centre) must take the 64-bit generic kernel - and match the oracle run on
    that same table either way."""
    p = products[3]
    api = p.api
    o = copy.copy(ck.oracle(3))
    o._table = ck.oracle(3).table() * scale
    pre = api.Precomputed()
    for i, v in enumerate(o._table):
        pre.lanczos_kernel_table[i] = int(v)
    for ch, rates, frames in ((2, (44100, 48000, 44100), 50000), (2, (48000, 44100, 44100), 30000), (1, (44100, 8000, 8000), 40000)):
        ok, st = p.low_init(ch, *rates)
        ok, ost = o.low_init(ch, *rates)
        R = int(ost.cfg.radius_frames)
        padded = ck.pad_frames(ck.noise_pcm(frames * ch, 13), ch, R)
        # full-scale negative samples meet the largest weights
        padded[R * ch: R * ch + 4000] = -32768
        want = o.low_resample_i32(ost, padded, frames)[0]
        got = api.LowLevel_ResampleBulk(st.raw, pre, padded, frames)[0]
        assert np.array_equal(got, want), (scale, ch, rates)
        info = api.PlanGetInfo(api.PlanCreate(p.low_init(ch, *rates)[1].raw, pre))
        assert (info.kernel != 0) == (expect_kernel == "fast"), (scale, info.asdict())


@pytest.mark.parametrize("args", [[], ["3", "500001", "peer"], ["1", "300000", "rccl"], ["4", "100003", "none"]])
def test_c_client_drives_every_device(args):
    """tools/cr_multi.c: a C89 program with no HIP headers - DeviceCount, DeviceAllocOn, SetThreadDevice + CopyToDevice,
    PlanShard, ONE ResampleShardedDevice call with the concatenate on device 0, compared against a single-device bulk call."""
    exe = os.path.join(ROOT, "tools", "bin", "cr_multi")
    assert os.path.exists(exe), "build() makes it"
    try:
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        if "rccl" in args:
            pytest.skip("RCCL communicator setup did not finish on this box")
        raise
    if "rccl" in args and r.returncode != 0 and "librccl" in r.stderr:
        pytest.skip("librccl unusable on this box: " + r.stderr[-300:])
    assert r.returncode == 0 and "cr_multi: OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_rccl_gather_between_shards_is_opt_in(products):
    """CLOWNRESAMPLER_AMD_GATHER_RCCL has never issued an RCCL operation between two devices (one GPU per box here).
    With more than one shard the call is REFUSED before anything is launched - error ARGUMENT, 0 frames, the state untouched, the
    message naming the way out - unless CLOWNRESAMPLER_AMD_EXPERIMENTAL_RCCL=1; the peer-copy gather, which has run with eight shards, is
    what a client gets to use."""
    import torch
    assert not os.environ.get("CLOWNRESAMPLER_AMD_EXPERIMENTAL_RCCL")
    p = products[3]
    api = p.api
    dev = torch.device("cuda", 0)
    ch, frames = 2, 50000
    ok, st = p.low_init(ch, 44100, 48000, 44100)
    before = st.astuple()
    d_in = torch.zeros((frames + 6) * ch, dtype=torch.int16, device=dev)
    d_out = torch.zeros(60000 * ch, dtype=torch.int32, device=dev)
    shards = [(0, d_in.data_ptr(), d_out.data_ptr(), None), (0, d_in.data_ptr(), d_out.data_ptr(), None)]
    launches = [api.LaunchCount(k) for k in range(9)]
    with pytest.raises(cr.ClownResamplerError) as e:
        api.ResampleShardedDevice(st.raw, p.pre, frames, shards, gather_mode=cr.GATHER_RCCL, root_shard=0, root_output=d_out.data_ptr())
    assert "CLOWNRESAMPLER_AMD_EXPERIMENTAL_RCCL" in str(e.value) and "PEER_COPY" in str(e.value)
    assert st.astuple() == before and [api.LaunchCount(k) for k in range(9)] == launches


def test_preflight_dry_run(tmp_path):
    """tools/multi_gpu_preflight.sh - the one command for the day a multi-GPU node is at hand - in its quick dry mode on THIS box: the
    one-call sharded entry point with 8 shards and peer copies through the C client, bench.py at 1 rank and at 2 ranks (sharing the GPU
    over gloo), the table: the script itself stays runnable, step by step in fresh processes."""
    env = dict(os.environ, CRA_PREFLIGHT_DRY="1", CRA_PREFLIGHT_QUICK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRA_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "multi_gpu_preflight.sh"), str(tmp_path)], capture_output=True, text=True, timeout=1500, env=env)
    log = open(os.path.join(str(tmp_path), "preflight.log")).read() if os.path.exists(os.path.join(str(tmp_path), "preflight.log")) else ""
    assert r.returncode == 0 and "ALL STEPS PASSED" in r.stdout, (r.stdout[-3000:], log[-3000:])


_COPY_PATH_SNIPPET = r"""
import os, sys
sys.path[:0] = [%(root)r, %(tests)r]
import numpy as np
import torch
torch.cuda.init()
import _checkers as ck, _product
p, o = _product.Product(3), ck.oracle(3)
ch, rates, frames = 2, (44100, 48000, 44100), 1500000          # 6 MB in, 13 MB out: far beyond the runtime's 1 MiB threshold
ok, st = p.low_init(ch, *rates); ok, ost = o.low_init(ch, *rates)
padded = ck.pad_frames(ck.noise_pcm(frames * ch, 11), ch, 3)
want = o.low_resample_i32(ost, padded, frames)[0]
print("=====BEGIN", file=sys.stderr, flush=True)
got, left, ran_out = p.api.LowLevel_ResampleBulk(st.raw, p.pre, padded, frames)
d = p.api.DeviceAlloc(padded.nbytes)
p.api.CopyToDevice(d, padded)
back = np.empty_like(padded)
p.api.CopyFromDevice(back, d)
p.api.DeviceFree(d)
print("=====END", file=sys.stderr, flush=True)
assert left == 0 and np.array_equal(got, want) and np.array_equal(back, padded)
print("COPIES-OK")
"""


def test_pageable_copies_stay_off_the_runtimes_pinned_path():
    """Round 6's regression test: every abort of a GPU test run that could be placed was a GPU page fault at a HEAP address
    during a copy of more than 1 MiB between the device and pageable memory, which the runtime serves by page-locking the host range in place
    ("HSA Copy Using Pinned resource" in its own log).  The library's copies of a client's pageable memory - the staged host-pointer entry
    point, CopyToDevice / CopyFromDevice - now leave in 1 MiB pieces, which the runtime stages ("... Using Staging resource"): with the
    runtime's defaults in force, not ONE pinned copy between the markers; with CLOWNRESAMPLER_AMD_PAGEABLE_PIECE=0 (whole copies, the library of
    rounds 1-5) the same calls make them - so the check can see what it checks."""
    code = _COPY_PATH_SNIPPET % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    counts = {}
    for piece in ("1048576", "0"):
        env = dict(os.environ, AMD_LOG_LEVEL="4", CLOWNRESAMPLER_AMD_PAGEABLE_PIECE=piece, GPU_PINNED_MIN_XFER_SIZE="1")   # (1 MiB: the runtime's default)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "COPIES-OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
        log = r.stderr.split("=====BEGIN", 1)[1].split("=====END", 1)[0]
        counts[piece] = (log.count("Using Pinned resource"), log.count("Using Staging resource"))
    if counts["0"][0] == 0:
        pytest.skip("this runtime's log does not name its pinned copies (%r): nothing to hold the library against" % (counts,))
    assert counts["1048576"][0] == 0 and counts["1048576"][1] > 0, counts
