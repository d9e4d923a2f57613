#!/usr/bin/env python3
"""Times every tuning variant of the specialised k_poly instance of a workload in ONE process, interleaved rounds
(cdna_hip_programming.md rule 24), checking each variant's output against variant 0 bit for bit.
Usage: python tools/sweep_variants.py [--workload cfg2] [--rounds 5] [--steps 10] [--variants 0,1,2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import clownresampler_amd as cr  # noqa: E402
from bench import WORKLOADS, device_noise  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--variants", default="")
    args = ap.parse_args()
    radius, ch, rates, frames = WORKLOADS[args.workload]
    api = cr.load(radius)
    dev = torch.device("cuda", 0)
    pre = api.precomputed()
    st0 = api.LowLevel_State()
    api.LowLevel_Init(st0, ch, *rates)
    R = st0.lowest_level.integer_stretched_kernel_radius
    n_out = api.CountOutputFrames(st0, frames)
    sets = []
    for s in range(3):
        pcm = device_noise((frames + 2 * R) * ch, -R * ch + s * 7919, dev)
        pcm[: R * ch] = 0
        pcm[(frames + R) * ch:] = 0
        sets.append((pcm, torch.empty(n_out * ch, dtype=torch.int32, device=dev)))
    stream = torch.cuda.current_stream(dev)
    nvar = api.lib.crhip_poly_variants() if hasattr(api.lib, "crhip_poly_variants") else 48
    variants = [int(v) for v in args.variants.split(",")] if args.variants else list(range(26))
    plans, infos = {}, {}
    for v in variants:
        api.DebugSetVariant(v)
        plans[v] = api.PlanCreate(st0, pre)
        infos[v] = api.PlanGetInfo(plans[v])

    stamp = torch.zeros(4 * 4096 + 4 * 64 + 32 * 4096, dtype=torch.int64, device=dev)
    api.lib.ClownResamplerAMD_DebugSetStampBuffer.argtypes = [__import__('ctypes').c_void_p]
    api.lib.ClownResamplerAMD_DebugSetStampBuffer(stamp.data_ptr())

    def run(v, i):
        pcm, out = sets[i % len(sets)]
        st = cr.LowLevel_State.from_buffer_copy(st0)
        api.ResampleDevice(plans[v], st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)

    # correctness: every variant equals variant[0] on buffer set 0
    run(variants[0], 0)
    torch.cuda.synchronize()
    ref = sets[0][1].clone()
    bad = []
    for v in [x for x in variants[1:] if x < 1000]:
        sets[0][1].zero_()
        run(v, 0)
        torch.cuda.synchronize()
        if not torch.equal(sets[0][1], ref):
            bad.append(v)
    times = {v: [] for v in variants}
    for r in range(args.rounds):
        for v in variants:
            run(v, 0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for i in range(args.steps):
                run(v, i)
            e1.record(stream)
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / args.steps * 1e3)
    nbytes = frames * ch * 2 + n_out * ch * 4
    print("variant geo(thr,vec) asm U nt  tile lds blocks | us median  min | GB/s(median)  frac of 8 TB/s")
    geos = [(256, 2), (512, 1), (512, 2), (1024, 1), (1024, 2)]
    for v in sorted(variants, key=lambda v: sorted(times[v])[len(times[v]) // 2]):
        t = sorted(times[v])
        med, mn = t[len(t) // 2], t[0]
        i = infos[v]
        if v >= 1000:
            extra = ""
            if v in (1006, 1007, 1008):
                import numpy as _np
                raw = stamp.cpu().numpy()
                ph = raw[4 * 4096:4 * 4096 + 4 * 64].reshape(-1, 4).astype(float)
                st = raw[:4 * 4096].reshape(-1, 4)
                st = st[st[:, 2] != 0]
                t0 = st[:, 1].min()
                life = (st[:, 2] - st[:, 1]) / 100.0
                endt = (st[:, 2] - t0) / 100.0
                clk = st[:, 0] / _np.maximum(st[:, 2] - st[:, 1], 1) / 10.0
                extra = "\n   %d workgroups: clock %.2f-%.2f GHz; start spread %.1f us; lifetime us min/median/max %.1f/%.1f/%.1f; end time us min/median/max %.1f/%.1f/%.1f" % (
                    len(st), clk.min(), clk.max(), (st[:, 1].max() - t0) / 100.0, life.min(), _np.median(life), life.max(), endt.min(), _np.median(endt), endt.max())
                if v == 1006 and ph.sum() > 0:
                    tot = ph.sum(axis=1).mean()
                    extra += "\n   wave 0 of workgroups 0-63, cycles per workgroup: issue next DMA+ticket %.0f (%.0f%%), arithmetic+stores %.0f (%.0f%%), wait DMA %.0f (%.0f%%), barrier %.0f (%.0f%%)" % tuple(
                        x for k in range(4) for x in (ph[:, k].mean(), 100 * ph[:, k].mean() / tot))
                if v == 1008 and ph.sum() > 0:
                    tot = ph.sum(axis=1).mean()
                    extra += "\n   wave 0 of workgroups 0-63, cycles per wave: tile setup + window unpack %.0f (%.0f%%), frames %.0f (%.0f%%), staged results to global %.0f (%.0f%%), waiting for the next window %.0f (%.0f%%)" % tuple(
                        x for k in range(4) for x in (ph[:, k].mean(), 100 * ph[:, k].mean() / tot))
                for x in range(8):
                    m = st[:, 3] == x
                    if m.any():
                        extra += "\n   XCC %d: %3d workgroups, end time median %.1f max %.1f us" % (x, m.sum(), _np.median(endt[m]), endt[m].max())
            print("ablation %d (timing only%s): %7.1f us median %7.1f min%s" % (v - 1000, "" if v in (1006, 1007, 1008) else ", results wrong by design" + {1009: "; k_up2 without global stores", 1010: "; k_up2 without the frames", 1011: "; k_up2 with one row read per wave-tile", 1012: "; k_up2 with conflict-free staging writes", 1013: "; k_up2 with both"}.get(v, ""), med, mn, extra))
            continue
        if v >= 22 and v < 26:
            print("%3d  2 lanes/frame geo %s nt=%d %5d %6d %4d | %7.1f %7.1f | %7.0f  %.3f %s" % (v, [(1024, 2), (512, 2)][(v - 22) % 2], 1 - (v - 22) // 2, i.tile_frames, i.lds_bytes, i.max_blocks, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000, "MISMATCH" if v in bad else ""))
            continue
        if v in (26, 27, 28, 29, 30):
            kind = {26: "k_up (round 1)", 27: "k_up2", 28: "k_poly 64-bit chain", 29: "k_wave/k_poly 64-bit chain", 30: "k_wave2"}[v]
            print("%3d  %-26s kernel %d %5d %6d %4d | %7.1f %7.1f | %7.0f  %.3f %s" % (v, kind, i.kernel, i.tile_frames, i.lds_bytes, i.max_blocks, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000, "MISMATCH" if v in bad else ""))
            continue
        if v in (20, 21):
            print("%3d  k_wave     nt=%d          %5d %6d %4d | %7.1f %7.1f | %7.0f  %.3f %s" % (v, 21 - v, i.tile_frames, i.lds_bytes, i.max_blocks, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000, "MISMATCH" if v in bad else ""))
            continue
        print("%3d  %-10s %d  %d  %d  %5d %6d %4d | %7.1f %7.1f | %7.0f  %.3f %s" % (v, geos[v % 5], 1, 1 << ((v // 5) % 2), (v // 10) % 2,
              i.tile_frames, i.lds_bytes, i.max_blocks, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000, "MISMATCH" if v in bad else ""))
    print("mismatching variants:", bad)


if __name__ == "__main__":
    main()
