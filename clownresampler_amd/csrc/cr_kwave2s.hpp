// cr_kwave2s.hpp - k_wave2s: k_wave2's expanded window and 2-instruction tap for WIDE frames - one lane per channel PAIR of a frame.
#ifndef CR_KWAVE2S_HPP
#define CR_KWAVE2S_HPP

#include "cr_device.hpp"
#include "cr_kup.hpp"   // wait_vmcnt_at_most

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_wave2s - long windows with many channels.  BUILT, BIT-EXACT, MEASURED - AND NOT A DEFAULT: see the verdict at the end of this comment
// ---------------------------------------------------------------------------------------------------------
// k_wave2 gives a lane a whole output frame: with 9-16 channels that is 9-16 accumulator pairs and window reads per slot, and
// such frames ran on the run-time-slot k_poly instead (two lanes per frame, SDWA taps: ~5.6 VALU per tap and channel;
// 0.16-0.31 of the roofline for 15-33 slots, profiles/r02_channel_table.log).  Here a frame is spread over
// LPF = ceil(channels / 2) lanes, each taking TWO neighbouring channels (the last lane of an odd frame one channel and a
// phantom), and a wave-instruction covers FW = 64 / LPF frames:
//   * the row is read once per lane as before - but the LPF lanes of a frame read the SAME address, which the LDS broadcasts;
//   * a window slot is ONE ds_read_b64 per lane (even channel counts; two ds_read_b32 for odd ones, whose frames start on
//     4-byte boundaries every other time), and the LPF lanes of a frame read consecutive dwords: no conflicts inside a frame;
//   * the taps are k_wave2's any-sign form on two pinned accumulator pairs (one SDWA xor to arm, one v_mad_i64_i32), two
//     independent chains per lane, whatever the channel count;
//   * a frame's samples leave as LPF neighbouring 8-byte stores: coalesced.
// Everything else - the expanded window (one dword per sample, X = sample << 16), a private double-buffered LDS-DMA window per
// wave, chunks dealt round-robin with a ticketed tail, rows rotated within blocks of 16 while they are staged - is k_wave2's
// run-time-slot form (cr_kwave2.hpp); the wave-tile is `a.wave_tile` frames (a multiple of FW chosen by the host so that the
// window of a tile fits the wave's slice of LDS), not 64.
// Verdict (profiles/r03_wave2s.log, r03_hq48c16_*_pmc_summary.txt): 16 channels x 15 slots (8 lobes, 44.1 -> 48 kHz) 177 us against the
// run-time-slot k_poly's 132-135; 44.1 -> 8 kHz (33 slots) 0.17-0.20 of the roofline against 0.16-0.27.  The counters say why: 163
// VALU instructions per (frame, channel pair) for its 30 tap-channels - 5.4 per tap-channel, MORE than k_poly's 5.0 - because
// everything that is per frame in k_wave2 (row index, window and row addresses, the normalise's setup, the store's address, four
// address increments per plane since the frame stride is not a compile-time constant) is paid by every one of a frame's 8 lanes
// here and amortised over two channels instead of eight.  The 2-instruction tap only pays where a lane has several channels'
// worth of them per frame.  The kernel stays selectable (variant 32, CLOWNRESAMPLER_AMD_WAVE2S_MIN_CHANNELS) and in the parity tests.
//   EVENCH  1: even channel count (8-byte window reads and stores), 0: odd (dwords; the phantom channel is computed, not stored)
// ---------------------------------------------------------------------------------------------------------
template <int EVENCH, int OUT16, int NT>
__global__ __launch_bounds__(1024) void k_wave2s(const crhip_poly_launch a)
{
	const unsigned chf = a.channels;                   // channels per frame
	const unsigned lpf = (chf + 1u) >> 1;              // lanes per frame
	const unsigned FW = 64u / lpf;                     // frames per wave-instruction
	const unsigned FB = chf * 2u;
	const unsigned n_waves = blockDim.x >> 6;
	const unsigned NTHREADS = n_waves * 64u;
	const unsigned slots = a.slots;
	const unsigned WT = a.wave_tile;                   // frames per wave-tile
	const unsigned CHUNK = a.tile_frames;              // frames per chunk: a power-of-two multiple of WT
	const unsigned CW = CHUNK / WT;
	const unsigned nvw = a.vecs - 150u;
	const unsigned BUF = nvw * 1024u;                  // bytes per packed window
	const unsigned XBUF = 2u * BUF;                    // bytes of the expanded window (4 per sample)
	const unsigned PER_WAVE = 2u * BUF + XBUF;
	const unsigned planes_total = a.row_stride / 4u;
	const unsigned weight_planes = planes_total - 1u;

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned lane = tid & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;
	unsigned char *my_buf = smem + rows_bytes + wave * PER_WAVE;
	unsigned char *my_x = my_buf + 2u * BUF;
	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + rows_bytes + n_waves * PER_WAVE);
	if (tid == 0)
		*waves_done = 0;

	const uint64_t n_chunks = (a.n_out + CHUNK - 1) / CHUNK;
	const uint64_t global_wave = (uint64_t)wave * gridDim.x + blockIdx.x;
	const uint64_t global_waves = (uint64_t)gridDim.x * n_waves;

	// who takes which chunk: as k_wave2 (static rounds, the last two drawn as tickets over 32 counter lanes)
	const uint64_t rounds = n_chunks / global_waves;
	const uint64_t static_limit = rounds > 3u ? (rounds - 2u) * global_waves : n_chunks;
	const uint64_t region_chunks = n_chunks - static_limit;
	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	const uint64_t lane_chunks = region_chunks > lane_id ? (region_chunks - lane_id + LANES - 1u) / LANES : 0;
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	auto draw_issue = [&]() -> unsigned { return draw_ticket(lane_counter); };
	auto draw_resolve = [&](unsigned got) -> uint64_t {
		const uint64_t k = (uint64_t)lane_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)got);
		return k < lane_chunks ? static_limit + lane_id + (uint64_t)LANES * k : ~0ull;
	};
	auto retire = [&]() {
		if (lane == 0 && __hip_atomic_fetch_add(waves_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == n_waves - 1u)
		{
			unsigned *finished = a.d_tickets + 32u * 32u;
			if (__hip_atomic_fetch_add(finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u)
			{
				for (unsigned c = 0; c < 32u; ++c)
					__hip_atomic_store(a.d_tickets + c * 32u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	};

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;

	// LDS-DMA of the packed window of the wave-tile of `n` frames starting at output frame `first` (see k_wave2)
	auto fetch = [&](uint64_t first, unsigned n, unsigned char *buf) -> unsigned {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		const uint64_t first_byte = in_base + ((pos >> 16) + a.first_slot) * FB;
		const uint64_t aligned = first_byte & ~(uint64_t)15;
		const unsigned shift = (unsigned)(first_byte - aligned);
		const unsigned last_rel = (unsigned)(((pos & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16);
		const unsigned frames = last_rel + slots + a.window_extra;
		uint64_t want = (uint64_t)shift + (uint64_t)frames * FB;
		uint64_t avail = in_end > aligned ? in_end - aligned : 0;
		if (want > avail)
			want = avail;
		want = (want + 3u) & ~(uint64_t)3u;   // whole dwords: see k_poly
		const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)aligned);
		const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(aligned >> 32));
		const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)want);
		const __amdgpu_buffer_rsrc_t rsrc =
		    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
		for (unsigned v = 0; v < nvw; ++v)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + v * 1024u), 16,
			                                         (int)(v * 1024u + lane * 16u), 0, 0, 0);
		return (unsigned)__builtin_amdgcn_readfirstlane((int)(shift | (frames << 16)));
	};

	unsigned first_info = 0;
	if (global_wave < n_chunks)
	{
		const uint64_t first = global_wave * CHUNK;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		first_info = fetch(first, n, my_buf);
	}

	// stage the rows once per workgroup, rotated within their blocks of 16 (run-time-slot image: weight planes, then the reciprocals)
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		auto place = [&](unsigned r) { return (r & ~15u) | ((__umul24(r >> 4, a.swizzle) + r) & 15u); };
		for (unsigned q = 0; q < planes_total; ++q)
			for (unsigned r = tid; r < a.plane_rows; r += NTHREADS)
				dst[q * a.plane_rows + place(r)] = src[q * a.plane_rows + r];
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	// packed window -> X = sample << 16, one dword per sample, in window order; returns 1 when the window's first sample is the high
	// half of its dword (odd channel counts): sample j is then X[1 + j]
	auto expand = [&](const unsigned char *buf, unsigned shift_frames) -> unsigned {
		const unsigned shift = shift_frames & 0xFFFFu, frames = shift_frames >> 16;
		const unsigned odd = (shift >> 1) & 1u;
		const unsigned dwords = (odd + frames * chf + 1u) / 2u;
		const int *from = reinterpret_cast<const int *>(buf + (shift & ~3u)) + lane;
		i32x2 *to = reinterpret_cast<i32x2 *>(my_x) + lane;
		for (unsigned k = 0; k * 64u < dwords; ++k)
		{
			const int f = from[k * 64u];
			i32x2 x;
			x.x = (int)((unsigned)f << 16);
			x.y = (int)((unsigned)f & 0xFFFF0000u);
			to[k * 64u] = x;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		return odd;
	};

	// this lane's place in a frame, fixed for the kernel
	const unsigned frame_lane = lane / lpf;            // which of the FW frames of a wave-instruction
	const unsigned pair = lane - frame_lane * lpf;     // which channel pair of it
	const bool lane_used = frame_lane < FW;
	const bool full_pair = 2u * pair + 1u < chf;       // false: the last lane of an odd frame (its second channel is a phantom)
	const unsigned stride = chf * 4u;                  // bytes between consecutive frames of the expanded window

	// the two channels of one frame from the expanded window
	auto one_frame = [&](unsigned rel, unsigned x_odd, int &out0, int &out1) {
		unsigned shift, row;
		if (a.row_mode == CRHIP_ROWMODE_UPSAMPLE)   // (wave-uniform)
			row = row_of<CRHIP_ROWMODE_UPSAMPLE>(a, rel & 0xFFFFu, shift);
		else
			row = row_of<CRHIP_ROWMODE_AFFINE>(a, rel & 0xFFFFu, shift);
		const unsigned phys = (row & ~15u) | ((__umul24(row >> 4, a.swizzle) + row) & 15u);
		unsigned win0 = (unsigned)(uintptr_t)my_x + (x_odd + ((rel >> 16) + shift) * chf + 2u * pair) * 4u;
		unsigned win1 = win0 + stride, win2 = win1 + stride, win3 = win2 + stride;
		unsigned row_q = (unsigned)(uintptr_t)smem + phys * 16u;
		const unsigned plane_bytes = a.plane_rows * 16u;
		int lo0, hi0 = 0, lo1, hi1 = 0;
#define CRHIP_W2S_TAP(LO, HI, VLO, VHI, X, W)                                                                                      \
	asm("v_xor_b32_sdwa v" #LO ", sext(%2), sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"   \
	    "v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]"                                                      \
	    : "=&{v" #LO "}"(VLO), "+{v" #HI "}"(VHI) : "v"(X), "v"(W) : "vcc")
		// one plane per trip: a ds_read_b128 of four weights + the window samples of its four slots, then their taps.  (Requesting
		// plane q + 1 before the taps of plane q - two register sets - measured SLOWER: 16 channels x 15 slots 177 -> 206 us,
		// profiles/r03_wave2s.log; the kernel is short of instruction slots, not waiting for the LDS.)
		for (unsigned q = 0; q < weight_planes; ++q)
		{
			i32x4 wq;
			i32x2 xv[4];
			int xa[4], xb[4];
			asm volatile("ds_read_b128 %0, %1" : "=v"(wq) : "v"(row_q));
			if constexpr (EVENCH)
			{
				asm volatile("ds_read_b64 %0, %1" : "=v"(xv[0]) : "v"(win0));
				asm volatile("ds_read_b64 %0, %1" : "=v"(xv[1]) : "v"(win1));
				asm volatile("ds_read_b64 %0, %1" : "=v"(xv[2]) : "v"(win2));
				asm volatile("ds_read_b64 %0, %1" : "=v"(xv[3]) : "v"(win3));
			}
			else
			{
				asm volatile("ds_read_b32 %0, %1" : "=v"(xa[0]) : "v"(win0));
				asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(xb[0]) : "v"(win0));
				asm volatile("ds_read_b32 %0, %1" : "=v"(xa[1]) : "v"(win1));
				asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(xb[1]) : "v"(win1));
				asm volatile("ds_read_b32 %0, %1" : "=v"(xa[2]) : "v"(win2));
				asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(xb[2]) : "v"(win2));
				asm volatile("ds_read_b32 %0, %1" : "=v"(xa[3]) : "v"(win3));
				asm volatile("ds_read_b32 %0, %1 offset:4" : "=v"(xb[3]) : "v"(win3));
			}
			constexpr int RPS = EVENCH ? 1 : 2;   // reads per slot
			asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * RPS) : "memory");
			asm volatile("" : "+v"(wq));
			const int wv[4] = {wq.x, wq.y, wq.z, wq.w};
#pragma unroll
			for (int s = 0; s < 4; ++s)
			{
				if (s > 0)
					asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"((3 - s) * RPS) : "memory");
				if constexpr (EVENCH)
				{
					asm volatile("" : "+v"(xv[s]));
					xa[s] = xv[s].x;
					xb[s] = xv[s].y;
				}
				else
					asm volatile("" : "+v"(xa[s]), "+v"(xb[s]));
				// (a padded slot has weight 0 and adds exactly 0 whatever its window read returns)
				CRHIP_W2S_TAP(120, 121, lo0, hi0, xa[s], wv[s]);
				CRHIP_W2S_TAP(122, 123, lo1, hi1, xb[s], wv[s]);
			}
			row_q += plane_bytes;
			win0 += 4u * stride;
			win1 += 4u * stride;
			win2 += 4u * stride;
			win3 += 4u * stride;
		}
#undef CRHIP_W2S_TAP
		(void)lo0;
		(void)lo1;
		int reciprocal;
		asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(reciprocal) : "v"(row_q) : "memory");
		// (acc * reciprocal) / 32768 toward zero in 64 bits: right for either range class of the host's
		const long long v0 = (long long)hi0 * (long long)reciprocal + (long long)((unsigned)(hi0 >> 31) >> 17);
		const long long v1 = (long long)hi1 * (long long)reciprocal + (long long)((unsigned)(hi1 >> 31) >> 17);
		out0 = (int)(v0 >> 15);
		out1 = (int)(v1 >> 15);
	};

	auto store_pair = [&](uint64_t frame, int out0, int out1) {
		if constexpr (OUT16)
		{
			short *dst = reinterpret_cast<short *>(a.d_out) + frame * chf + 2u * pair;
			if (EVENCH || full_pair)
			{
				// (a frame of an even channel count starts on a 4-byte boundary of the output; an odd one every other time)
				if constexpr (EVENCH)
					*reinterpret_cast<int *>(dst) = (clamp_s16(out0) & 0xFFFF) | (clamp_s16(out1) << 16);
				else
				{
					dst[0] = (short)clamp_s16(out0);
					dst[1] = (short)clamp_s16(out1);
				}
			}
			else
				dst[0] = (short)clamp_s16(out0);
		}
		else
		{
			int *dst = reinterpret_cast<int *>(a.d_out) + frame * chf + 2u * pair;
			if constexpr (EVENCH)
			{
				const int v[2] = {out0, out1};
				store_ints<2, NT>(dst, v);
			}
			else
			{
				dst[0] = out0;
				if (full_pair)
					dst[1] = out1;
			}
		}
	};

	if (global_wave >= n_chunks)
	{
		retire();
		return;
	}

	uint64_t chunk = global_wave;
	unsigned cur = 0, packed_info = first_info;

	auto run_chunk = [&](uint64_t this_chunk, auto next_of) -> uint64_t {
		uint64_t next_chunk = ~0ull;
		const uint64_t chunk_first = this_chunk * CHUNK;

		for (unsigned j = 0; j < CW; ++j)
		{
			const uint64_t first = chunk_first + (uint64_t)j * WT;
			const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
			const bool last_of_stream = first + n >= a.n_out;
			unsigned next_info = 0;
			bool have_next = false;

			const unsigned x_odd = expand(my_buf + cur * BUF, packed_info);

			if (!last_of_stream && j + 1 < CW)
			{
				const uint64_t nf = first + WT;
				const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
				next_info = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
				have_next = true;
			}
			else
			{
				next_chunk = next_of();
				if (next_chunk != ~0ull && !last_of_stream)
				{
					const uint64_t nf = next_chunk * CHUNK;
					const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
					next_info = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
					have_next = true;
				}
			}

			const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
			const unsigned frac0 = (unsigned)(pos & 0xFFFFu);
			unsigned trips = 0;   // wave-uniform: the stores this wave-tile issues at the least
			for (unsigned base = 0; base < n; base += FW)
			{
				const unsigned fi = base + frame_lane;
				if (lane_used && fi < n)
				{
					int out0, out1;
					one_frame(__umul24(fi, a.increment) + frac0, x_odd, out0, out1);
					store_pair(first + fi, out0, out1);
				}
				++trips;
			}

			if (last_of_stream || !have_next)
				return ~0ull;
			// own DMA landed once only this wave-tile's stores are outstanding (vmcnt is in order; `trips` is a lower bound of them)
			wait_vmcnt_at_most(trips);
			cur ^= 1u;
			packed_info = next_info;
		}
		return next_chunk;
	};

	while (chunk != ~0ull && chunk < static_limit)
	{
		const uint64_t this_chunk = chunk;
		chunk = run_chunk(this_chunk, [&]() -> uint64_t {
			const uint64_t next = this_chunk + global_waves;
			return next < n_chunks ? next : ~0ull;
		});
	}
	while (chunk != ~0ull)
	{
		const unsigned ticket = draw_issue();
		chunk = run_chunk(chunk, [&]() -> uint64_t { return draw_resolve(ticket); });
	}

	retire();
}

} // namespace

#endif // CR_KWAVE2S_HPP
