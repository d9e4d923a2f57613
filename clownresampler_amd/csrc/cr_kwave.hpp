// cr_kwave.hpp - k_wave: the k_poly arithmetic with wave-autonomous streaming.
#ifndef CR_KWAVE_HPP
#define CR_KWAVE_HPP

#include "cr_device.hpp"

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_wave - the same arithmetic with WAVE-AUTONOMOUS streaming: no workgroup barrier after the rows are staged
// ---------------------------------------------------------------------------------------------------------
// k_poly pays about a microsecond per tile in its barrier (every wave waits for the slowest, then all start their LDS
// reads at once): ~12 us of a 64 us launch.  Here every wave owns a private, double-buffered 1 KiB (x NVW) slice of LDS,
// fills it with its own LDS-DMA and only ever waits for itself (s_waitcnt vmcnt): the rows are the one thing the waves
// of a workgroup share, read-only.  Work is handed out per WAVE in chunks of 4 wave-tiles (4 x 64 x ITER output
// frames): the first chunk by global wave number, the rest by atomic tickets over 32 counter lanes (see k_poly), drawn
// one chunk ahead.
//   WAVES  waves per workgroup          NVW  1 KiB DMA pieces per wave-tile          ITER  frames per lane per wave-tile
// ASM: arithmetic form of full wave-tiles, as in k_poly (1 = SDWA, 2 | NEGMASK << 8 = 64-bit multiply-add chain)
template <int CH, int TT, int MODE, int NORM, int WAVES, int NVW, int ITER, int OUT16, int NT, int ABL = 0, int ASM = 1>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_num_sgpr(CRHIP_SGPR_BUDGET))) void k_wave(const crhip_poly_launch a)
{
	static_assert(TT > 0, "k_wave exists for specialised slot counts only");
	constexpr unsigned FB = CH * 2;
	constexpr unsigned NTHREADS = WAVES * 64;
	constexpr unsigned WT = 64u * ITER;            // frames per wave-tile
	constexpr unsigned CW = 4;                     // wave-tiles per chunk (ticket)
	constexpr unsigned CHUNK = WT * CW;
	constexpr unsigned BUF = NVW * 1024u;          // bytes per wave-tile buffer
	constexpr int STORES_PER_FRAME = min_stores_of_bytes(CH * (OUT16 ? 2 : 4));   // a lower bound: see cr_device.hpp

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned lane = tid & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	unsigned long long stamp_cycles = 0, stamp_ticks = 0;
	if constexpr (ABL == 6)
	{
		stamp_cycles = __builtin_amdgcn_s_memtime();
		stamp_ticks = __builtin_amdgcn_s_memrealtime();
	}

	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;
	const int *rows = reinterpret_cast<const int *>(smem);
	unsigned char *my_buf = smem + rows_bytes + wave * (2u * BUF);

	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + rows_bytes + WAVES * (2u * BUF));
	if (tid == 0)
		*waves_done = 0;

	// stage the polyphase rows once per workgroup: the only barrier of the kernel
	{
		const unsigned nvec = rows_bytes / 16u;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		for (unsigned i = tid; i < nvec; i += NTHREADS)
			dst[i] = src[i];
	}
	__syncthreads();

	const uint64_t n_chunks = (a.n_out + CHUNK - 1) / CHUNK;
	const uint64_t global_wave = (uint64_t)wave * gridDim.x + blockIdx.x;   // (a short launch spreads over the CUs, not over a CU's waves)
	const uint64_t global_waves = (uint64_t)gridDim.x * WAVES;

	// tickets: as in k_poly, per wave, 32 counter lanes
	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	const uint64_t lane_chunks = n_chunks > lane_id ? (n_chunks - lane_id + LANES - 1u) / LANES : 0;
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	// the draw is split: the atomic is issued at the start of a chunk, its result is first looked at when the last
	// wave-tile of the chunk needs it - by then the per-wave-tile vmcnt waits have long covered it
	auto draw_issue = [&]() -> unsigned { return draw_ticket(lane_counter); };   // (scalar, the whole wave: cr_device.hpp)
	auto draw_resolve = [&](unsigned got) -> uint64_t {
		const uint64_t k = (uint64_t)lane_waves + __builtin_amdgcn_readfirstlane(got);
		return k < lane_chunks ? lane_id + (uint64_t)LANES * k : ~0ull;
	};
	// a wave that has run out of tickets retires; the waves of a workgroup count down in LDS and only the last of them
	// touches the global finished counter (8,192 waves on one word would serialise for ~100 us: one word takes ~88
	// atomics per microsecond), and the last workgroup zeroes the ticket block for the next launch
	auto retire = [&]() {
		if (lane == 0 && __hip_atomic_fetch_add(waves_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == WAVES - 1u)
		{
			unsigned *finished = a.d_tickets + 32u * 32u;
			if (__hip_atomic_fetch_add(finished, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u)
			{
				for (unsigned c = 0; c < 32u; ++c)
					__hip_atomic_store(a.d_tickets + c * 32u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				__hip_atomic_store(finished, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		if constexpr (ABL == 6)
		{
			if (lane == 0 && wave == 0 && a.debug_stamps != nullptr)
			{
				a.debug_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - stamp_cycles;
				a.debug_stamps[4 * blockIdx.x + 1] = stamp_ticks;
				a.debug_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
				a.debug_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(63508) & 0xF;
			}
		}
	};

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;

	// LDS-DMA of the input window of the wave-tile of `n` frames starting at output frame `first` into `buf`; returns the
	// byte offset of the window's first frame inside the buffer.  Not waited for.
	auto fetch = [&](uint64_t first, unsigned n, unsigned char *buf) -> unsigned {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		const uint64_t first_byte = in_base + ((pos >> 16) + a.first_slot) * FB;
		const uint64_t aligned = first_byte & ~(uint64_t)15;
		const unsigned shift = (unsigned)(first_byte - aligned);
		const unsigned last_rel = (unsigned)(((pos & 0xFFFFu) + (uint64_t)(n - 1) * a.increment) >> 16);
		uint64_t want = (uint64_t)shift + (uint64_t)(last_rel + TT + a.window_extra) * FB;
		uint64_t avail = in_end > aligned ? in_end - aligned : 0;
		if (want > avail)
			want = avail;
		want = (want + 3u) & ~(uint64_t)3u;   // whole dwords: see k_poly
		const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)aligned);
		const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(aligned >> 32));
		const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)want);
		const __amdgpu_buffer_rsrc_t rsrc =
		    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
#pragma unroll
		for (int v = 0; v < NVW; ++v)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(buf + v * 1024u), 16,
			                                         (int)(v * 1024u + lane * 16u), 0, 0, 0);
		return shift;
	};

	// one full wave-tile (WT frames, ITER per lane) from `buf`, software-pipelined; leaves its stores in flight
	auto wave_tile = [&](uint64_t first, const unsigned char *base) {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		const unsigned lane_rel = __umul24(lane, a.increment) + (unsigned)(pos & 0xFFFFu);
		int *out32 = reinterpret_cast<int *>(a.d_out) + first * CH;
		short *out16 = reinterpret_cast<short *>(a.d_out) + first * CH;

		FrameData<CH, TT> d[2];
		fetch_frame<CH, TT, MODE, 0>(a, rows, base, lane_rel, d[0]);
#pragma unroll
		for (int i = 0; i < ITER; ++i)
		{
			int outv[CH];
			if (i + 1 < ITER)
				fetch_frame<CH, TT, MODE, 0>(a, rows, base, lane_rel + (unsigned)(i + 1) * 64u * a.increment, d[(i + 1) & 1]);
			__builtin_amdgcn_sched_barrier(0);
			compute_frame<CH, TT, NORM, ASM>(d[i & 1], outv);
			if constexpr (OUT16)
				store_shorts<CH, NT>(out16 + (size_t)(i * 64u) * CH + lane * CH, outv);
			else
				store_ints<CH, NT>(out32 + (size_t)(i * 64u) * CH + lane * CH, outv);
			__builtin_amdgcn_sched_barrier(0);
		}
	};

	if (global_wave >= n_chunks)
	{
		retire();
		return;
	}

	uint64_t chunk = global_wave;
	unsigned parity = 0;
	// first wave-tile of the first chunk
	{
		const uint64_t first = chunk * CHUNK;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		const unsigned shift0 = fetch(first, n, my_buf);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		// `shift` of the buffered wave-tile travels in a scalar
		parity = shift0 << 1;   // bit 0: buffer index, bits 1..: shift
	}

	for (;;)
	{
		const unsigned ticket = draw_issue();          // one chunk ahead; resolved at the end of this chunk
		uint64_t next_chunk = ~0ull;
		const uint64_t chunk_first = chunk * CHUNK;
		const bool full = chunk_first + CHUNK <= a.n_out;

		if (full)
		{
#pragma unroll
			for (unsigned j = 0; j < CW; ++j)
			{
				const uint64_t first = chunk_first + j * WT;
				const unsigned cur = parity & 1u;
				const unsigned shift = parity >> 1;
				unsigned shift_next = 0;
				bool have_next = true;

				// start the DMA of the wave-tile after this one (the other buffer was consumed one step ago)
				if (j + 1 == CW)
					next_chunk = draw_resolve(ticket);

				if (j + 1 < CW)
					shift_next = fetch(first + WT, WT, my_buf + (cur ^ 1u) * BUF);
				else if (next_chunk != ~0ull)
				{
					const uint64_t nf = next_chunk * CHUNK;
					const unsigned n = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
					shift_next = fetch(nf, n, my_buf + (cur ^ 1u) * BUF);
				}
				else
					have_next = false;

				wave_tile(first, my_buf + cur * BUF + shift);

				// own DMA landed once only this wave-tile's stores are outstanding (vmcnt is in order); no barrier:
				// nobody else reads this wave's buffers
				if constexpr (ITER * STORES_PER_FRAME <= 63)
					asm volatile("s_waitcnt vmcnt(%0)" ::"i"(ITER * STORES_PER_FRAME) : "memory");
				else
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				(void)have_next;
				parity = (shift_next << 1) | (cur ^ 1u);
			}
		}
		else
		{
			// the stream's ragged last chunk: frame by frame with bounds checks, wave-tile by wave-tile
			for (uint64_t first = chunk_first; first < a.n_out; first += WT)
			{
				const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
				const unsigned cur = parity & 1u;
				const unsigned shift = parity >> 1;
				const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
				const unsigned frac0 = (unsigned)(pos & 0xFFFFu);
				const unsigned char *base = my_buf + cur * BUF + shift;

				for (unsigned jl = lane; jl < n; jl += 64u)
				{
					int outv[CH];
					one_frame<CH, TT, MODE, NORM, 1, 0>(a, rows, base, __umul24(jl, a.increment) + frac0, outv);
					if constexpr (OUT16)
						store_shorts<CH, NT>(reinterpret_cast<short *>(a.d_out) + (first + jl) * CH, outv);
					else
						store_ints<CH, NT>(reinterpret_cast<int *>(a.d_out) + (first + jl) * CH, outv);
				}

				if (first + WT < a.n_out)
				{
					const uint64_t nf = first + WT;
					const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
					const unsigned shift_next = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
					parity = (shift_next << 1) | (cur ^ 1u);
				}
			}
		}

		if (!full)
		{
			(void)draw_resolve(ticket);   // the last chunk of the stream is the last of its sequence: nothing follows
			break;
		}
		if (next_chunk == ~0ull)
			break;
		chunk = next_chunk;
	}

	retire();
}

} // namespace

#endif // CR_KWAVE_HPP
