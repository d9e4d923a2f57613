"""CPU, world_size 2, gloo: the N>1 path's plumbing - shard planning per rank, per-rank compute, the final all_gather -
reassembles the one-shot stream bit for bit.  The per-shard compute here is the ORACLE (test infrastructure; there is no
GPU in this container); on the GPU box tests/test_gpu_parity.py runs the same shards through the HIP kernel."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _checkers as ck


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ch, rates, frames, radius, ret):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.dirname(here)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import _checkers as ck2
    import clownresampler_amd as cr
    from clownresampler_amd import distributed as crd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        api = cr.load(radius)
        o = ck2.oracle(radius)
        st = api.LowLevel_State()
        assert api.LowLevel_Init(st, ch, *rates)
        R = st.lowest_level.integer_stretched_kernel_radius
        # every rank materialises only ITS slice (+ halo) of the same stream
        whole = ck2.pad_frames(ck2.noise_pcm(frames * ch, 77), ch, R)
        sh = crd.shard_of(api, st, frames, rank, world)
        lo, hi = sh.first_input_frame, sh.first_input_frame + sh.input_frames + 2 * R
        mine = whole[lo * ch: hi * ch].copy()
        ok, ost = o.low_init(ch, *rates)
        ost.pos_int, ost.pos_frac = sh.state.position_integer, sh.state.position_fractional
        out, left, ran_out = o.low_resample_i32(ost, mine, sh.input_frames, capacity=sh.output_frames)
        assert out.size == sh.output_frames * ch
        total = api.CountOutputFrames(st, frames)
        full = crd.gather_output(torch.from_numpy(np.ascontiguousarray(out)), sh, total, ch, world)
        ok, ref = o.low_init(ch, *rates)
        want, _, _ = o.low_resample_i32(ref, whole, frames)
        same = bool(np.array_equal(full.numpy(), want))
        at_root = crd.gather_output_to_root(torch.from_numpy(np.ascontiguousarray(out)), sh, total, ch, world, root=0)
        same = same and ((at_root is None) if rank != 0 else bool(np.array_equal(at_root.numpy(), want)))
        t = torch.tensor([1 if same else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret.put((int(t[0]), int(total)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("ch,rates,frames,radius", [(2, (44100, 48000, 44100), 50001, 3), (8, (48000, 44100, 44100), 9000, 3), (2, (8000, 96000, 8000), 1500, 8)])
def test_two_ranks_gloo(ch, rates, frames, radius):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ch, rates, frames, radius, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    same, total = ret.get(timeout=10)
    assert same == 1 and total > 0
