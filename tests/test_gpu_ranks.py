"""GPU box: the N>1 rank path run by the PRODUCT.  Several processes (one per rank, gloo for the rendezvous, all on GPU 0 -
the pool's boxes have one GPU) each plan their shard with ClownResamplerAMD_PlanShard, materialise only their slice of the
stream (+ halo), run it through ClownResamplerAMD_ResampleDevice, and the gathered output must be the oracle's one-shot
stream bit for bit.  Also: bench.py's self-launching multi-rank mode runs to a JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import _checkers as ck

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ch, rates, frames, radius, s16, ret):
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import _checkers as ck2
    import clownresampler_amd as cr
    from clownresampler_amd import distributed as crd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        api = cr.load(radius)
        api.SetDevice(rank % api.DeviceCount())
        pre = api.precomputed()
        st = api.LowLevel_State()
        assert api.LowLevel_Init(st, ch, *rates)
        R = st.lowest_level.integer_stretched_kernel_radius
        whole = ck2.pad_frames(ck2.noise_pcm(frames * ch, 1234), ch, R)
        sh = crd.shard_of(api, st, frames, rank, world)
        # this rank's slice + halo, and nothing else, goes to the device
        lo, hi = sh.first_input_frame, sh.first_input_frame + sh.input_frames + 2 * R
        mine = np.ascontiguousarray(whole[lo * ch: hi * ch])
        plan = api.PlanCreate(st, pre)
        assert api.PlanGetInfo(plan).kernel != 0, "a polyphase kernel should serve this configuration"
        d_in = api.DeviceAlloc(max(16, mine.nbytes))
        d_out = api.DeviceAlloc(max(16, sh.output_frames * ch * 4))
        api.CopyToDevice(d_in, mine)
        run = cr.LowLevel_State.from_buffer_copy(sh.state)
        n, left, ran_out = api.ResampleDevice(plan, run, d_in, sh.input_frames, d_out, sh.output_frames, None, s16=s16)
        assert n == sh.output_frames
        api.StreamSynchronize(None)
        out = np.empty(sh.output_frames * ch, dtype=np.int16 if s16 else np.int32)
        if out.size:
            api.CopyFromDevice(out, d_out)
        api.DeviceFree(d_in)
        api.DeviceFree(d_out)
        total = api.CountOutputFrames(st, frames)
        full = crd.gather_output(torch.from_numpy(out.astype(np.int32)), sh, total, ch, world)
        root = crd.gather_output_to_root(torch.from_numpy(out.astype(np.int32)), sh, total, ch, world, root=0)
        same = True
        if rank == 0:
            o = ck2.oracle(radius)
            ok, ref = o.low_init(ch, *rates)
            want, _, _ = o.low_resample_i32(ref, whole, frames)
            if s16:
                want = np.clip(want, -0x7FFF, 0x7FFF)
            same = bool(np.array_equal(full.numpy(), want)) and bool(np.array_equal(root.numpy(), want))
        else:
            assert root is None
        t = torch.tensor([1 if same else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put((int(t[0]), int(total)))
    finally:
        dist.destroy_process_group()


CASES = [
    # BASELINE configs[4] at 1/60 size (one minute of the hour), 2 and 3 ranks; configs[2] and configs[3] at one minute
    pytest.param(2, 2, (44100, 48000, 44100), 2646000, 3, False, id="cfg5_1min_2ranks"),
    pytest.param(3, 2, (44100, 48000, 44100), 2646000, 3, False, id="cfg5_1min_3ranks"),
    pytest.param(2, 2, (8000, 96000, 8000), 480000, 8, False, id="cfg3_1min_2ranks"),
    pytest.param(2, 8, (48000, 44100, 44100), 2880000, 3, False, id="cfg4_1min_2ranks"),
    pytest.param(2, 2, (44100, 48000, 44100), 100003, 3, True, id="s16_2ranks"),
    pytest.param(4, 1, (44100, 8000, 8000), 70001, 3, False, id="mono_33taps_4ranks"),
]


@pytest.mark.parametrize("world,ch,rates,frames,radius,s16", CASES)
def test_rank_processes_run_the_product(world, ch, rates, frames, radius, s16):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ch, rates, frames, radius, s16, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    same, total = ret.get(timeout=10)
    assert same == 1 and total > 0


@pytest.mark.parametrize("gpus,extra", [(2, ["--workload", "cfg2", "--scaling", "strong"]), (2, ["--workload", "cfg2", "--scaling", "weak"])])
def test_bench_launches_its_own_ranks(gpus, extra):
    """`python bench.py --gpus N` from a plain shell (no torch.distributed.run around it): N workers, one JSON line, parity
    checked on every rank and across the shard seams of the gathered stream."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "6", "--warmup", "2", "--prewarm-ms", "10"] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == gpus and line["parity_full_stream"] is True
    assert line["gather"]["seams_match_oracle"] is True
    assert line["value"] > 0
    # the line proves what ran: one entry per rank with its GPU's identity, its own time and its share; the backend summed N ones
    assert [r["rank"] for r in line["per_rank"]] == list(range(gpus))
    assert sum(r["output_frames"] for r in line["per_rank"]) == sum(r["output_frames"] for r in line["per_rank"]) > 0
    assert all(r["ms_per_step"] > 0 and r["device"]["pci"] for r in line["per_rank"])
    assert line["world_size_seen_by_backend"] == {"get_world_size": gpus, "all_reduce_sum_of_ones": gpus, "backend": line["world_size_seen_by_backend"]["backend"]}
    assert line["gather"]["link_GBs"] > 0


def _device_count():
    import clownresampler_amd as cr
    return cr.load(3).DeviceCount()


@pytest.mark.skipif(_device_count() < 2, reason="needs a node with at least two GPUs (the pool's boxes have one)")
def test_two_distinct_gpus_over_rccl():
    """The day a multi-GPU node runs this suite: bench.py --gpus 2 must put its ranks on DISTINCT devices and talk RCCL, and the
    C entry point must concatenate across two ordinals both by peer copies and by RCCL."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "CRA_BENCH_BACKEND"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--prewarm-ms", "10"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["backend"].startswith("nccl") and line["world_size_seen_by_backend"]["all_reduce_sum_of_ones"] == 2
    assert line["distinct_devices"] == 2 and line["parity_full_stream"] is True and line["gather"]["seams_match_oracle"] is True
    exe = os.path.join(ROOT, "tools", "bin", "cr_multi")
    for mode in ("peer", "rccl"):
        c = subprocess.run([exe, "2", "2646000", mode], capture_output=True, text=True, timeout=600, env=env)
        assert c.returncode == 0 and "cr_multi: OK" in c.stdout, (mode, c.stdout[-1000:], c.stderr[-2000:])
