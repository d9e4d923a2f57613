import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import ctypes, numpy as np, torch
import clownresampler_amd as cr
from bench import WORKLOADS, device_noise
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
radius, ch, rates, frames = WORKLOADS[wl]
api = cr.load(radius); dev = torch.device("cuda", 0); pre = api.precomputed()
st0 = api.LowLevel_State(); api.LowLevel_Init(st0, ch, *rates)
R = st0.lowest_level.integer_stretched_kernel_radius
n_out = api.CountOutputFrames(st0, frames)
pcm = device_noise((frames + 2 * R) * ch, -R * ch, dev)
out = torch.empty(n_out * ch, dtype=torch.int32, device=dev)
stamp = torch.zeros(4 * 4096 + 4 * 64 + 32 * 4096, dtype=torch.int64, device=dev)
api.lib.ClownResamplerAMD_DebugSetStampBuffer.argtypes = [ctypes.c_void_p]
api.lib.ClownResamplerAMD_DebugSetStampBuffer(stamp.data_ptr())
api.DebugSetVariant(int(sys.argv[2]) if len(sys.argv) > 2 else 1006)
plan = api.PlanCreate(st0, pre)
stream = torch.cuda.current_stream(dev)
for rep in range(30):
    st = cr.LowLevel_State.from_buffer_copy(st0)
    api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
if os.environ.get("STAMP_COLD"):
    # one launch from idle
    torch.cuda.synchronize()
    stamp.zero_()
    st = cr.LowLevel_State.from_buffer_copy(st0)
    api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
else:
    # the last of a sustained series (every launch overwrites the stamps of the one before)
    for rep in range(300):
        st = cr.LowLevel_State.from_buffer_copy(st0)
        api.ResampleDevice(plan, st, pcm.data_ptr(), frames, out.data_ptr(), n_out, stream.cuda_stream)
torch.cuda.synchronize()
raw = stamp.cpu().numpy()[:4 * 4096].reshape(-1, 4)
raw = raw[raw[:, 2] != 0]
t0 = raw[:, 1].min()
start = (raw[:, 1] - t0) / 100.0; end = (raw[:, 2] - t0) / 100.0
print(len(raw), "workgroups; start time histogram (us):", np.histogram(start, bins=[0, 1, 2, 5, 10, 20, 30, 40, 50, 60, 70, 80])[0].tolist())
print("lifetime histogram (us):", np.histogram(end - start, bins=[0, 1, 2, 5, 10, 20, 30, 40, 50, 60, 70, 80])[0].tolist())
order = np.argsort(start)
print("block ids of the 10 latest starters:", order[-10:].tolist(), "their starts", np.round(start[order[-10:]], 1).tolist())
print("block ids of the 10 earliest:", order[:10].tolist())
late = start > 5
print("late starters per XCC:", [int((late & (raw[:, 3] == x)).sum()) for x in range(8)], " early per XCC:", [int((~late & (raw[:, 3] == x)).sum()) for x in range(8)])
# per-tile stamps (k_poly diagnostic instance): tile index << 48 | tick
pt = stamp.cpu().numpy()[4 * 4096 + 4 * 64:].reshape(4096, 32)[:len(raw)]
idx = (pt >> 48).astype(np.int64); tick = (pt & 0xFFFFFFFFFFFF).astype(np.int64)
tick = np.where(tick >= (t0 & 0xFFFFFFFFFFFF), tick, 0)   # older launches' entries
ntile = (tick != 0).sum(axis=1)
print("tiles per workgroup histogram:", np.bincount(ntile).tolist())
tt = np.where(tick != 0, (tick - t0) / 100.0, np.nan)
dur = np.diff(np.concatenate([start[:, None], tt], axis=1), axis=1)
print("tile duration us: p5/p50/p95/max", np.nanpercentile(dur, [5, 50, 95, 100]).round(2).tolist())
for it in range(0, 20):
    col = dur[:, it]
    if np.isfinite(col).any():
        print("  tile #%2d of a workgroup: n=%3d duration p5/p50/p95/max" % (it, np.isfinite(col).sum()), np.nanpercentile(col, [5, 50, 95, 100]).round(2).tolist(), "end p50", np.nanpercentile(tt[:, it], 50).round(1))
early = np.argsort(end)[:5]; latew = np.argsort(end)[-5:]
for w in list(early) + list(latew):
    k = ntile[w]
    print("wg %3d lane %d xcc %d end %.1f tiles" % (w, w % 8, raw[w, 3] & 0xF, end[w]), idx[w, :k].tolist(), "ends", np.round(tt[w, :k], 1).tolist())
# in which order were the tiles of sequence 0 finished?
seq = [(tt[w, i], idx[w, i], w) for w in range(len(raw)) for i in range(ntile[w]) if w % 8 == 0]
seq.sort()
print("sequence 0: last 12 tiles finished:", [(round(float(a), 1), int(b), int(c)) for a, b, c in seq[-12:]])
