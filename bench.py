#!/usr/bin/env python3
"""bench.py - throughput of the windowed-sinc hot path on MI355X, BASELINE.json's metric.

One "step" = one pass of the hot path (ClownResampler_LowLevel_Resample's per-frame loop, reference
clownresampler.h:1058-1092 / :986-1035, as the HIP kernel k_poly) over one batch of synthetic PCM, inputs and outputs
resident in HBM.  At N=1 the batch is BASELINE configs[1]: stereo int16, 44.1 -> 48 kHz, 3-lobe Lanczos, 10 minutes
(26,460,000 -> 28,800,096 frames).  With N ranks every rank owns one such 10-minute shard of an N x 10-minute stream
(output-timeline sharding, input halo replicated, no collective in the data path): weak scaling.

Prints ONE JSON line (rank 0).  `value` = output Msamples/s of the whole job, kernel time only (HIP events on the
launch stream, barrier + synchronize on both sides, max over ranks).  `roofline` prices the same kernel against the
8 TB/s HBM peak with ALGORITHMIC bytes (each input sample read once + each int32 output written once).
`cpu_baseline` is the reference C path (oracle/_ref, the real header compiled in place, when that prebuilt checker is
present; else the oracle restatement) timed on this box's host cores on the same workload - a baseline, not a target.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"

WORKLOADS = {
    # name: (radius, channels, (in, out, lowpass), input frames per GPU)
    "cfg2": (3, 2, (44100, 48000, 44100), 26460000),   # BASELINE configs[1] - the headline
    "cfg3": (8, 2, (8000, 96000, 8000), 4800000),      # configs[2] (10 min assumed)
    "cfg4": (3, 8, (48000, 44100, 44100), 28800000),   # configs[3] (10 min assumed)
    "cfg5": (3, 2, (44100, 48000, 44100), 158760000),  # configs[4] on ONE GPU (1 hour)
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
}


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


def cpu_baseline(radius, ch, rates, frames, max_seconds=30.0):
    """Times the reference C path on the host: real reference (.so prebuilt from /root/reference) if present, else oracle."""
    import numpy as np
    import _checkers as ck
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
           "sample": "full workload, %d -> %d frames x %d ch, callback API storing int32, gcc -O2, 1 thread, %.2f s" % (frames, n_out, ch, dt)}
    # all host cores, independent states over contiguous input ranges (oracle driver)
    cores = os.cpu_count() or 1
    if cores > 1:
        o = ck.oracle(radius)
        ok, fresh = o.low_init(ch, *rates)
        t0 = time.perf_counter()
        got = o.low_resample_i32_mt(fresh, padded, frames, cores, out=out)
        dt = time.perf_counter() - t0
        res["all_cores"] = {"value": got.size / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port", "seconds": dt}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm-ms", type=float, default=250.0, help="untimed launches before the W warmup steps, until this much wall time has passed: the chip needs ~100 ms of load to leave its idle clocks")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--sets", type=int, default=3, help="rotating buffer sets (defeats the 256 MiB Infinity Cache)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--s16", action="store_true", help="opt-in extension: clamped int16 output (NOT the BASELINE metric; writes 2 B per sample instead of 4)")
    ap.add_argument("--graph", action="store_true", help="time one hipGraph replay of the K steps instead of eager launches (measured SLOWER on ROCm 7.2 for this kernel)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import clownresampler_amd as cr
    from clownresampler_amd import distributed as crd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d: launch with torch.distributed.run --nproc-per-node N" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the GPU: the product has no CPU path")
    # CRA_BENCH_BACKEND=gloo is a validation hook for boxes with fewer GPUs than ranks: the ranks share the GPUs there are
    # and synchronise over gloo (everything but RCCL itself is exercised); the driver's runs use nccl = RCCL, one GPU per rank
    backend = os.environ.get("CRA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    radius, ch, rates, frames_per_gpu = WORKLOADS[args.workload]
    api = cr.load(radius)
    api.SetDevice(local_rank)
    pre = api.precomputed()
    whole = api.LowLevel_State()
    assert api.LowLevel_Init(whole, ch, *rates)
    R = whole.lowest_level.integer_stretched_kernel_radius
    total_frames = frames_per_gpu * world
    shard = crd.shard_of(api, whole, total_frames, rank, world)
    plan = api.PlanCreate(whole, pre)
    info = api.PlanGetInfo(plan)

    # this rank's slice of the stream + halo; logical frame f of the stream is padded frame f + R; the zero padding of the
    # stream's two ends is materialised only where a shard touches it
    in_frames = shard.input_frames + 2 * R
    first_logical = shard.first_input_frame - R
    sets = []
    for s in range(max(1, args.sets)):
        pcm = device_noise(in_frames * ch, (first_logical * ch) + s * 7919, device)
        lo_pad = max(0, -first_logical)
        hi_pad = max(0, first_logical + in_frames - total_frames)
        if lo_pad:
            pcm[: lo_pad * ch] = 0
        if hi_pad:
            pcm[(in_frames - hi_pad) * ch:] = 0
        out = torch.empty(shard.output_frames * ch, dtype=torch.int16 if args.s16 else torch.int32, device=device)
        sets.append((pcm, out))
    stream = torch.cuda.current_stream(device)

    def step(i):
        pcm, out = sets[i % len(sets)]
        if args.s16:
            st = cr.LowLevel_State.from_buffer_copy(shard.state)
            return api.ResampleDevice(plan, st, pcm.data_ptr(), shard.input_frames, out.data_ptr(), shard.output_frames, stream.cuda_stream, s16=True)[0]
        return crd.resample_shard_device(api, plan, shard, pcm.data_ptr(), out.data_ptr(), stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    # Clock ramp: untimed launches until --prewarm-ms have passed (then the W warmup steps, then the K timed ones).
    t_pre = time.perf_counter()
    i_pre = 0
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:
        for _ in range(20):
            step(i_pre)
            i_pre += 1
        torch.cuda.synchronize(device)

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
                with torch.cuda.graph(graph, stream=side):
                    stream = torch.cuda.current_stream(device)
                    for i in range(args.steps):
                        step(i)
            stream = side
        except Exception as e:  # capture not possible on this stack: fall back to eager launches
            if rank == 0:
                print("bench: hipGraph capture failed (%s); timing eager launches" % e, file=sys.stderr)
            graph = None
            stream = torch.cuda.current_stream(device)

    def run_steps():
        if graph is not None:
            graph.replay()
        else:
            for i in range(args.steps):
                step(i)

    if graph is not None:
        with torch.cuda.stream(stream):
            for _ in range(max(1, args.warmup // max(1, args.steps))):
                run_steps()
    else:
        for i in range(args.warmup):
            step(i)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        ev0.record(stream)
        run_steps()
        ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    ms_rank = max(dev_ms, 0.0)
    if world > 1:
        t = torch.tensor([ms_rank, wall * 1e3], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms_rank, wall_ms = float(t[0]), float(t[1])
    else:
        wall_ms = wall * 1e3

    out_frames_all = int(api.CountOutputFrames(whole, total_frames))
    out_samples_all = out_frames_all * ch
    ms_per_step = ms_rank / args.steps
    value = out_samples_all / (ms_per_step * 1e-3) / 1e6

    # the dominant (only) kernel: algorithmic bytes of THIS rank's launch / its average duration
    launch_bytes = shard.input_frames * ch * 2 + shard.output_frames * ch * (2 if args.s16 else 4)
    achieved = launch_bytes / (dev_ms / args.steps * 1e-3) / 1e9
    traffic, traffic_note = None, "no PMC summary for this workload under profiles/"
    pmc_file = os.path.join(ROOT, "profiles", "r01_%s_pmc_summary.txt" % args.workload)
    if world == 1 and os.path.exists(pmc_file):
        vals = {}
        for ln in open(pmc_file):
            f = ln.split()
            if len(f) > 3 and f[1] == "per-dispatch":
                vals[f[0]] = float(f[3])
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            # KiB units; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced reads: x2 (MI355X_MICROARCH.md, HBM)
            traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
            traffic_note = "bytes per launch from rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes: profiles/" + os.path.basename(pmc_file)
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_note": traffic_note, "kernel": {1: "k_poly<%d,%d>", 2: "k_wave<%d,%d>", 3: "k_up<%d,%d>"}[info.kernel] % (ch, info.slots) if info.kernel else "k_generic",
                "algorithmic_bytes_per_launch": launch_bytes, "avg_launch_ms": dev_ms / args.steps,
                "read_only_GBs": shard.input_frames * ch * 2 / (dev_ms / args.steps * 1e-3) / 1e9}

    # correctness of what was just timed: spot-check a window of this rank's output against the oracle
    check = None
    if not args.no_check and rank == 0:
        import _checkers as ck
        o = ck.oracle(radius)
        n_chk = min(200000, shard.output_frames)
        pcm, out = sets[(args.steps - 1) % len(sets)]
        ok, ost = o.low_init(ch, *rates)
        ost.pos_int, ost.pos_frac = shard.state.position_integer, shard.state.position_fractional
        need = min(in_frames, int(n_chk * whole.increment / 65536) + 4 * R + 8)
        host_in = pcm[: need * ch].cpu().numpy()
        want, _, _ = o.low_resample_i32(ost, host_in, need - 2 * R, capacity=n_chk)
        got = out[: want.size].cpu().numpy()
        if args.s16:
            want = np.clip(want, -0x7FFF, 0x7FFF).astype(np.int16)
        check = bool(np.array_equal(got, want))
        if not check:
            raise SystemExit("bench: device output differs from the oracle - numbers void")

    gather = None
    if world > 1 and not args.s16 and backend == "nccl":
        # the final concatenate (north_star): all ranks' int32 shards gathered over xGMI by RCCL; timed apart from the kernel
        # (reported beside the headline; a failure here must not cost the run its bench line)
        try:
            per = (out_frames_all + world - 1) // world * ch
            send = torch.zeros(per, dtype=torch.int32, device=device)
            send[: shard.output_frames * ch] = sets[0][1]
            recv = torch.empty(per * world, dtype=torch.int32, device=device)
            dist.all_gather_into_tensor(recv, send)
            barrier()
            g0 = time.perf_counter()
            reps = 5
            for _ in range(reps):
                dist.all_gather_into_tensor(recv, send)
            barrier()
            g_ms = (time.perf_counter() - g0) * 1e3 / reps
            gather = {"collective": "all_gather_into_tensor (RCCL)", "ms": g_ms, "bytes_per_rank": per * 4,
                      "value_with_gather": out_samples_all / ((ms_per_step + g_ms) * 1e-3) / 1e6}
        except Exception as e:
            gather = {"collective": "all_gather_into_tensor (RCCL)", "error": str(e)[:200]}

    if rank == 0:
        line = {
            "metric": ("output Msamples/s at 44.1->48 kHz stereo" if args.workload in ("cfg2", "cfg5") else "output Msamples/s (%s)" % args.workload) + (" [int16-clamped output extension]" if args.s16 else ""),
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "s16 in / int32 16.16 fixed-point arithmetic / int32 out", "data": "synthetic",
            "config": {"workload": "%s: %d-ch int16 %d->%d Hz, %d-lobe Lanczos, %d input frames per GPU (%d -> %d frames in all), device-resident, %d rotating buffer sets"
                                   % (args.workload, ch, rates[0], rates[1], radius, frames_per_gpu, total_frames, out_frames_all, len(sets)),
                       "sharding": "output timeline split in %d contiguous blocks, input halo of %d frames replicated, no data-path collective" % (world, R),
                       "plan": info.asdict()},
            "roofline": roofline,
            "launch_mode": "hipGraph replay of the K steps" if graph is not None else "eager",
            "wall_ms_per_step": wall_ms / args.steps,
            "parity_spot_check": check,
        }
        if gather:
            line["gather"] = gather
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(radius, ch, rates, frames_per_gpu)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
