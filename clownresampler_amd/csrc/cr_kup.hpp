// cr_kup.hpp - k_up: input-stationary form for strong pure upsampling.
#ifndef CR_KUP_HPP
#define CR_KUP_HPP

#include "cr_device.hpp"

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_up - input-stationary form for strong pure upsampling (increment <= 32768: two or more output frames per input
// position).  k_poly / k_wave give every output frame its own lane, which then unpacks its whole tap window from LDS
// and fixes up the truncation of every product from the product's sign (4 VALU per tap and channel, ~15 cycles).
// When several output frames share one integer position they share the WINDOW, and in pure upsampling the weight a
// window frame meets always comes from the same lobe of the kernel, so its sign is known per slot at compile time
// (NEGMASK; the host checks the plan's rows against it).  So here a lane owns one INPUT position: it unpacks the
// window once into sign-extended samples S and truncation biases B (0xFFFF where sample * weight will be negative,
// decided by the sample's sign alone), and then every frame of that position costs per tap and channel
//     x = v_mad_i32_i24(S, w, B);   acc += x >> 16          (2-3 VALU, ~8.5 cycles)
// which is the reference's (sample * weight) / 65536 with C truncation (clownresampler.h:1020 via :625), exactly.
// A lane's frames are consecutive in the output, so results are staged through LDS and leave as coalesced stores.
// Wave-autonomous like k_wave: no barrier after the rows are staged; wave-tiles are dealt round-robin.
// ---------------------------------------------------------------------------------------------------------
// k_up: one tap of two channels as one statement: x = sample * weight + bias (24-bit multiply-add, exact), acc += x >> 16
// taken as the sign-extended high word of x.
__device__ __forceinline__ void up_tap_pair(int &acc0, int &acc1, int sample0, int sample1, int weight, int bias0, int bias1)
{
	int x0, x1;
	asm("v_mad_i32_i24 %2, %4, %6, %7\n\t"
	    "v_mad_i32_i24 %3, %5, %6, %8\n\t"
	    "v_add_u32_sdwa %0, %0, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
	    "v_add_u32_sdwa %1, %1, sext(%3) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
	    : "+v"(acc0), "+v"(acc1), "=&v"(x0), "=&v"(x1)
	    : "v"(sample0), "v"(sample1), "v"(weight), "v"(bias0), "v"(bias1));
}

// rows of a plane of the device image in pure-upsampling row mode: 1,025 rows ((65536 - fraction) >> 6), rounded up to 16
constexpr unsigned UP_PLANE_ROWS = 1040;

__device__ __forceinline__ void wait_vmcnt_at_most(unsigned n)
{
	// s_waitcnt takes a literal: one arm per count (n is wave-uniform)
	switch (n)
	{
#define CRHIP_WAIT_ARM(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
		CRHIP_WAIT_ARM(1) CRHIP_WAIT_ARM(2) CRHIP_WAIT_ARM(3) CRHIP_WAIT_ARM(4) CRHIP_WAIT_ARM(5) CRHIP_WAIT_ARM(6) CRHIP_WAIT_ARM(7) CRHIP_WAIT_ARM(8)
		CRHIP_WAIT_ARM(9) CRHIP_WAIT_ARM(10) CRHIP_WAIT_ARM(11) CRHIP_WAIT_ARM(12) CRHIP_WAIT_ARM(13) CRHIP_WAIT_ARM(14) CRHIP_WAIT_ARM(15) CRHIP_WAIT_ARM(16)
		CRHIP_WAIT_ARM(17) CRHIP_WAIT_ARM(18) CRHIP_WAIT_ARM(19) CRHIP_WAIT_ARM(20) CRHIP_WAIT_ARM(21) CRHIP_WAIT_ARM(22) CRHIP_WAIT_ARM(23) CRHIP_WAIT_ARM(24)
		CRHIP_WAIT_ARM(25) CRHIP_WAIT_ARM(26) CRHIP_WAIT_ARM(27) CRHIP_WAIT_ARM(28) CRHIP_WAIT_ARM(29) CRHIP_WAIT_ARM(30) CRHIP_WAIT_ARM(31) CRHIP_WAIT_ARM(32)
#undef CRHIP_WAIT_ARM
		default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
	}
}

// CHAIN: 1 = the tap as one 64-bit multiply-add on (sample << 16) whose addend pair is {bias << 16, running sum}: the carry out
//        of the low dword IS the truncation, the high dword the accumulator; the low dword is re-armed with a plain move per tap
template <int CH, int TT, int NORM, unsigned NEGMASK, int WAVES, int OUT16, int NT, int ABL = 0, int CHAIN = 0>
__global__ __launch_bounds__(WAVES * 64) void k_up(const crhip_poly_launch a)
{
	constexpr unsigned NTHREADS = WAVES * 64u;
	constexpr unsigned FB = CH * 2;                // bytes per input frame
	constexpr unsigned BUF = 1024u;                // bytes per window buffer: one 16-byte DMA per lane
	constexpr unsigned UNIT = OUT16 ? CH * 2 : CH * 4;   // bytes per output frame
	constexpr int RS = (TT + 1 + 3) & ~3;
	static_assert((63 + TT) * FB + 16 <= BUF, "the window of 64 input positions must fit one DMA piece");
	static_assert(UNIT % 4 == 0, "output frames are moved as dwords");
	constexpr unsigned VEC = UNIT % 8 == 0 ? 8 : 4;   // bytes per lane per store

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned lane = tid & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	// diagnostic instance (ABL == 6, variant 1008) only: where wave 0's cycles go, summed over its wave-tiles - [0] DMA issue,
	// ticket, the lane's frame range and the window unpack, [1] the frames, [2] staged results to global memory, [3] waiting
	// for the next window (vmcnt)
	unsigned long long stamp_cycles = 0, stamp_ticks = 0, phase[4] = {0, 0, 0, 0}, t_mark = 0;
	if constexpr (ABL == 6)
	{
		stamp_cycles = __builtin_amdgcn_s_memtime();
		stamp_ticks = __builtin_amdgcn_s_memrealtime();
	}
	auto mark = [&](int which) {
		if constexpr (ABL == 6)
		{
			__builtin_amdgcn_sched_barrier(0);
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			__builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): s_memtime returns through the scalar data path
			if (which >= 0)
				phase[which] += now - t_mark;
			t_mark = now;
			__builtin_amdgcn_sched_barrier(0);
		}
	};

	const unsigned WT = a.tile_frames / 4u;        // output frames per wave-tile: at most 64 input positions (the host passes 4 wave-tiles)
	const unsigned stage_bytes = (WT * UNIT + 15u) & ~15u;

	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;
	const int *rows = reinterpret_cast<const int *>(smem);
	unsigned char *my_buf = smem + rows_bytes + wave * (2u * BUF + stage_bytes);
	unsigned char *my_stage = my_buf + 2u * BUF;

	if (tid == 0)
		*reinterpret_cast<unsigned *>(smem + rows_bytes + WAVES * (2u * BUF + stage_bytes)) = 0;

	// stage the polyphase rows once per workgroup: the only barrier of the kernel
	{
		const unsigned nvec = rows_bytes / 16u;
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		for (unsigned i = tid; i < nvec; i += NTHREADS)
			dst[i] = src[i];
	}
	__syncthreads();

	// Wave-tiles are dealt round-robin to the WORKGROUPS of the persistent grid, and inside a workgroup its waves draw them
	// from a counter in LDS.  The kernel is VALU-bound, so workgroups progress alike; waves of one SIMD do not (the oldest
	// wave is issued first), and with a fixed share per wave the favoured waves leave early and the rest run on an
	// under-occupied SIMD (measured: 199 us against 182).  Global tickets as in k_wave would balance that too, but reading a
	// ticket's result costs a full vmcnt(0) drain - the atomic returns through the same counter as the wave-tile's stores -
	// which measured at ~30 % of the kernel; an LDS atomic returns through lgkmcnt.
	const uint64_t n_tiles = (a.n_out + WT - 1) / WT;
	unsigned *next_draw = reinterpret_cast<unsigned *>(smem + rows_bytes + WAVES * (2u * BUF + stage_bytes));
	auto draw = [&]() -> uint64_t {   // this workgroup's next wave-tile, or >= n_tiles
		unsigned d = 0;
		if (lane == 0)
			d = __hip_atomic_fetch_add(next_draw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		return (uint64_t)blockIdx.x + (uint64_t)gridDim.x * __builtin_amdgcn_readfirstlane(d);
	};
	auto finish = [&]() {
		if constexpr (ABL == 6)
		{
			if (lane == 0 && wave == 0 && a.debug_stamps != nullptr)
			{
				a.debug_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - stamp_cycles;
				a.debug_stamps[4 * blockIdx.x + 1] = stamp_ticks;
				a.debug_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
				a.debug_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(63508) & 0xF;
				if (blockIdx.x < 64u)
					for (int q = 0; q < 4; ++q)
						a.debug_stamps[4 * 4096 + 4 * blockIdx.x + q] = phase[q];
			}
		}
	};

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;
	const float inv_increment = __builtin_amdgcn_rcpf((float)a.increment);

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
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)buf, 16, (int)(lane * 16u), 0, 0, 0);
		return shift;
	};

	// smallest k with frac0 + k * increment >= l * 65536: the first frame (relative to the wave-tile) of input position l
	auto first_frame_of = [&](unsigned l, unsigned frac0) -> unsigned {
		if (l == 0)
			return 0;
		const unsigned x = (l << 16) - frac0;                     // 1 .. 2^22
		unsigned k = (unsigned)((float)x * inv_increment);        // within one of the quotient; made exact below
		k += (__umul24(k, a.increment) < x) ? 1u : 0u;
		k += (__umul24(k, a.increment) < x) ? 1u : 0u;
		k -= (k != 0 && __umul24(k - 1u, a.increment) >= x) ? 1u : 0u;
		k -= (k != 0 && __umul24(k - 1u, a.increment) >= x) ? 1u : 0u;
		return k;
	};

	// one wave-tile: n output frames from `first`, window at `base`; returns the number of store instructions issued
	auto wave_tile = [&](uint64_t first, unsigned n, const unsigned char *base) -> unsigned {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		const unsigned frac0 = (unsigned)(pos & 0xFFFFu);
		unsigned k = first_frame_of(lane, frac0);
		unsigned k_end = first_frame_of(lane + 1u, frac0);
		k = k < n ? k : n;
		k_end = k_end < n ? k_end : n;

		// the window of this lane's input position, unpacked once
		int S[TT][CH], B[TT][CH];
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			Frame<CH> f;
			f.load(base + (lane + (unsigned)s) * FB);
#pragma unroll
			for (int c = 0; c < CH; ++c)
			{
				int sample;
				if constexpr (Frame<CH>::PACKED)
					sample = (c & 1) ? (f.v[c / 2] >> 16) : (int)(short)f.v[c / 2];
				else
					sample = f.v[c];
				// the product with this slot's weight is negative iff the sample's sign differs from the slot's
				const unsigned bias = (unsigned)(((NEGMASK >> s) & 1u) ? -sample : sample) >> 16;
				S[s][c] = CHAIN ? (int)((unsigned)sample << 16) : sample;
				B[s][c] = CHAIN ? (int)(bias << 16) : (int)bias;
				asm volatile("" : "+v"(B[s][c]));   // keep it in a register: hipcc otherwise recomputes the shift in every frame
				if constexpr (CHAIN)
					asm volatile("" : "+v"(S[s][c]));
			}
		}

		unsigned frac = frac0 + __umul24(k, a.increment) - (lane << 16);   // fraction of frame k: its integer position is lane's

		auto read_row = [&](unsigned fraction, int (&w)[RS]) {
			const unsigned row = (65536u - (fraction & 0xFFFFu)) >> 6;   // (the masked case is a prefetch past the lane's last frame)
			const i32x4 *plane0 = reinterpret_cast<const i32x4 *>(rows) + row;
#pragma unroll
			for (int q = 0; q < RS / 4; ++q)
			{
				const i32x4 v = plane0[q * UP_PLANE_ROWS];   // compile-time stride: the three further planes are immediate offsets
				w[4 * q] = v.x;
				w[4 * q + 1] = v.y;
				w[4 * q + 2] = v.z;
				w[4 * q + 3] = v.w;
			}
		};
		auto one = [&](const int (&w)[RS], unsigned at) {
			// The loop is VALU-bound, so instruction cycles are what matters.  Per tap and channel: one 24-bit multiply-add and one
			// SDWA add that takes the high word of the product (the shift by 16) directly - 8.4 cycles per wave.  (A single
			// full-rate v_mad_i64_i32 on (sample << 16) with the bias in the low dword of the addend, plus a plain add, is 6.6
			// cycles on paper and bit-exact too, but measured slower: 168 VGPRs, spills, and a lower clock.)
			int acc[CH];
			if constexpr (CHAIN)
			{
				// Two chains per channel (even / odd slots): neighbouring multiply-adds are independent.  The accumulator pairs are
				// pinned to physical registers: the re-arming of the low dword is then ONE plain v_mov_b32 (given a 64-bit asm operand
				// hipcc copies the whole pair twice per tap instead).
				static_assert(CH == 2, "the chain form of k_up is written for stereo");
				int lo[4], hi[4] = {0, 0, 0, 0};
#define CRHIP_CHAIN_STEP(K, LO, HI, SAMPLE, WEIGHT, BIAS)                                                                  \
	lo[K] = (BIAS);                                                                                                    \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %2, %3, v[" #LO ":" #HI "]" : "+{v" #LO "}"(lo[K]), "+{v" #HI "}"(hi[K]) : "v"(SAMPLE), "v"(WEIGHT) : "vcc")
#pragma unroll
				for (int s = 0; s < TT; ++s)
				{
					if (s & 1)
					{
						CRHIP_CHAIN_STEP(1, 122, 123, S[s][0], w[s], B[s][0]);
						CRHIP_CHAIN_STEP(3, 126, 127, S[s][1], w[s], B[s][1]);
					}
					else
					{
						CRHIP_CHAIN_STEP(0, 120, 121, S[s][0], w[s], B[s][0]);
						CRHIP_CHAIN_STEP(2, 124, 125, S[s][1], w[s], B[s][1]);
					}
				}
#undef CRHIP_CHAIN_STEP
				acc[0] = hi[0] + hi[1];
				acc[1] = hi[2] + hi[3];
			}
			else
			{
#pragma unroll
			for (int c = 0; c < CH; ++c)
				acc[c] = (__mul24(S[0][c], w[0]) + B[0][c]) >> 16;
#pragma unroll
			for (int s = 1; s < TT; ++s)
			{
				if constexpr (CH % 2 == 0)
				{
#pragma unroll
					for (int c = 0; c < CH; c += 2)
						up_tap_pair(acc[c], acc[c + 1], S[s][c], S[s][c + 1], w[s], B[s][c], B[s][c + 1]);
				}
				else
				{
#pragma unroll
					for (int c = 0; c < CH; ++c)
						acc[c] = sdwa_add_word1_signed(acc[c], __mul24(S[s][c], w[s]) + B[s][c]);
				}
			}
			}

			int outv[CH];
#pragma unroll
			for (int c = 0; c < CH; ++c)
				outv[c] = normalise<NORM>(acc[c], w[TT]);

			if constexpr (OUT16)
			{
				int *dst = reinterpret_cast<int *>(my_stage) + at * (CH / 2);
#pragma unroll
				for (int c = 0; c < CH; c += 2)
					dst[c / 2] = (clamp_s16(outv[c]) & 0xFFFF) | (clamp_s16(outv[c + 1]) << 16);
			}
			else
			{
				int *dst = reinterpret_cast<int *>(my_stage) + at * CH;
#pragma unroll
				for (int c = 0; c < CH; ++c)
					dst[c] = outv[c];
			}
		};

		mark(0);
		// the row of frame k + 1 is read before the arithmetic of frame k (two register sets, loop unrolled by two)
		int wa[RS], wb[RS];
		if (k < k_end)
			read_row(frac, wa);
		while (k < k_end)
		{
			read_row(frac + a.increment, wb);
			__builtin_amdgcn_sched_barrier(0);
			one(wa, k);
			++k;
			frac += a.increment;
			if (k >= k_end)
				break;
			read_row(frac + a.increment, wa);
			__builtin_amdgcn_sched_barrier(0);
			one(wb, k);
			++k;
			frac += a.increment;
		}

		mark(1);
		// the staged frames of the other lanes: same wave, LDS operations of a wave complete in order
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		const unsigned vectors = n * UNIT / VEC;
		unsigned char *out = reinterpret_cast<unsigned char *>(a.d_out) + first * UNIT;
		unsigned stores = 0;
		typedef typename std::conditional<VEC == 8, i32x2, int>::type vec_t;
		const vec_t *staged = reinterpret_cast<const vec_t *>(my_stage);
		vec_t *dst = reinterpret_cast<vec_t *>(out);
		auto put = [&](unsigned i, vec_t v) {
			if constexpr (NT)
				__builtin_nontemporal_store(v, dst + i);
			else
				dst[i] = v;
		};
		unsigned done = 0;   // wave-uniform
		// four LDS reads in flight per trip: a read-then-store pair at a time would pay the LDS latency per store
		for (; done + 256u <= vectors; done += 256u)
		{
			const unsigned i = done + lane;
			const vec_t v0 = staged[i], v1 = staged[i + 64u], v2 = staged[i + 128u], v3 = staged[i + 192u];
			put(i, v0);
			put(i + 64u, v1);
			put(i + 128u, v2);
			put(i + 192u, v3);
		}
		for (unsigned i = done + lane; i < vectors; i += 64u)
			put(i, staged[i]);
		stores = (vectors + 63u) / 64u;
		__builtin_amdgcn_wave_barrier();
		mark(2);
		return __builtin_amdgcn_readfirstlane(stores);
	};

	uint64_t tile = draw();
	if (tile >= n_tiles)
	{
		finish();
		return;
	}

	unsigned cur = 0, shift = 0;
	{
		const uint64_t first = tile * WT;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		shift = fetch(first, n, my_buf);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	}
	mark(-1);

	for (;;)
	{
		const uint64_t first = tile * WT;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		const uint64_t next = draw();
		const bool have_next = next < n_tiles;
		unsigned shift_next = 0;

		// start the DMA of this wave's next wave-tile (its buffer was consumed one step ago)
		if (have_next)
		{
			const uint64_t nf = next * WT;
			const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
			shift_next = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
		}

		const unsigned stores = wave_tile(first, n, my_buf + cur * BUF + shift);

		if (!have_next)
			break;
		// the DMA was issued before this wave-tile's stores and vmcnt retires in order
		wait_vmcnt_at_most(stores);
		mark(3);
		cur ^= 1u;
		shift = shift_next;
		tile = next;
	}

	finish();
}

// ---------------------------------------------------------------------------------------------------------
// k_up2 - the input-stationary kernel, second form.  Same ownership as k_up (a lane owns one INPUT position of a wave-tile
// and produces all of its output frames; results staged through LDS, coalesced stores; wave-autonomous), different arithmetic
// and bookkeeping:
//
//   * The tap is  P = v_mad_i64_i32(S, W, P)  on a 64-bit accumulator pair P = {lo, hi} with
//         S  = sample * 32768, negated for the slots whose weights are <= 0      (once per window, one multiply)
//         W  = 2 * |weight|                                                      (once per workgroup, while the rows are staged)
//         lo = S >> 31  (all ones where S < 0)                                   (once per window; one plain move per tap re-arms it)
//     S * W = sample * weight * 65536 exactly (the two sign flips cancel), so the product's integer part lands in `hi` and its
//     16 fraction bits in the top of `lo`.  With every W >= 0 the product is negative exactly where S is, and there the all-ones
//     `lo` makes the carry into `hi` round the quotient up: hi += trunc(sample * weight / 65536), C's division
//     (clownresampler.h:1020 via :625), for negative and positive products alike ((p << 16) + 0xFFFFFFFF carries iff the
//     fraction bits of p are not all zero).  No per-sample sign work in the frame loop: a move and a multiply-add per tap and
//     channel, against 4 VALU of the per-output-frame kernels.
//   * The final (acc * reciprocal) / 32768 (clownresampler.h:1033) is one more 64-bit multiply-add and a funnel shift.
//   * The frames of a lane run under a WAVE-UNIFORM trip count (floor(65536 / increment), plus one only for waves in which
//     some position has the extra frame), with one predicate for the staging store - no per-lane loop control - and the
//     lane's first frame is found once per wave-tile (its end is the next lane's start).
//   * The copy-out of the staged frames runs on a wave-uniform base with one lane offset: no address arithmetic per vector.
// ---------------------------------------------------------------------------------------------------------
template <int CH, int TT, int NORM, unsigned NEGMASK, int WAVES, int OUT16, int NT, int ABL = 0>
__global__ __launch_bounds__(WAVES * 64) void k_up2(const crhip_poly_launch a)
{
	static_assert(CH == 2, "k_up2 is written for stereo (one packed dword per input frame, two accumulator pairs)");
	constexpr unsigned NTHREADS = WAVES * 64u;
	constexpr unsigned FB = CH * 2;                // bytes per input frame
	constexpr unsigned BUF = 1024u;                // bytes per window buffer: one 16-byte DMA per lane
	constexpr unsigned UNIT = OUT16 ? CH * 2 : CH * 4;   // bytes per output frame
	constexpr int RS = (TT + 1 + 3) & ~3;
	static_assert((63 + TT) * FB + 16 <= BUF, "the window of 64 input positions must fit one DMA piece");
	static_assert(UNIT % 4 == 0, "output frames are moved as dwords");
	constexpr unsigned VEC = UNIT % 8 == 0 ? 8 : 4;   // bytes per lane per store

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned lane = tid & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);

	// diagnostic instances only (ABL: 6 = clock stamps and where wave 0's cycles go - [0] tile setup + window unpack, [1] the
	// frames, [2] staged results to global memory, [3] waiting for the next window; timing-only, WRONG results: 1 = no global
	// stores, 2 = no frames)
	unsigned long long stamp_cycles = 0, stamp_ticks = 0, phase[4] = {0, 0, 0, 0}, t_mark = 0;
	if constexpr (ABL == 6)
	{
		stamp_cycles = __builtin_amdgcn_s_memtime();
		stamp_ticks = __builtin_amdgcn_s_memrealtime();
	}
	auto mark = [&](int which) {
		if constexpr (ABL == 6)
		{
			__builtin_amdgcn_sched_barrier(0);
			const unsigned long long now = __builtin_amdgcn_s_memtime();
			__builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): s_memtime returns through the scalar data path
			if (which >= 0)
				phase[which] += now - t_mark;
			t_mark = now;
			__builtin_amdgcn_sched_barrier(0);
		}
	};
	auto finish = [&]() {
		if constexpr (ABL == 6)
		{
			if (lane == 0 && wave == 0 && a.debug_stamps != nullptr)
			{
				a.debug_stamps[4 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - stamp_cycles;
				a.debug_stamps[4 * blockIdx.x + 1] = stamp_ticks;
				a.debug_stamps[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
				a.debug_stamps[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(63508) & 0xF;
				if (blockIdx.x < 64u)
					for (int q = 0; q < 4; ++q)
						a.debug_stamps[4 * 4096 + 4 * blockIdx.x + q] = phase[q];
			}
		}
	};

	const unsigned WT = a.tile_frames / 4u;        // output frames per wave-tile: at most 64 input positions
	const unsigned stage_bytes = (WT * UNIT + 15u) & ~15u;

	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;
	unsigned char *my_buf = smem + rows_bytes + wave * (2u * BUF + stage_bytes);
	unsigned char *my_stage = my_buf + 2u * BUF;
	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + rows_bytes + WAVES * (2u * BUF + stage_bytes));

	if (tid == 0)
		*waves_done = 0;

	// Stage the polyphase rows once per workgroup - as W = 2 * |weight|: plane q of the image holds int32 [4q, 4q + 4) of every
	// row, i.e. slots 4q .. 4q + 3 (the reciprocal sits in slot TT and stays as it is).  The only barrier of the kernel.
	{
		const u32x4 *src = reinterpret_cast<const u32x4 *>(a.d_rows);
		u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
		// (every load first, then the stores: two rows per thread and plane are eight loads, and one round trip instead of eight
		// is 5 us of every launch - k_up2's floor was 27 us)
		constexpr int PER = (UP_PLANE_ROWS + NTHREADS - 1) / NTHREADS;
		u32x4 v[RS / 4][PER];
#pragma unroll
		for (int q = 0; q < RS / 4; ++q)
#pragma unroll
			for (int k = 0; k < PER; ++k)
			{
				const unsigned r = tid + (unsigned)k * NTHREADS;
				if (r < UP_PLANE_ROWS)
					v[q][k] = src[q * UP_PLANE_ROWS + r];
			}
#pragma unroll
		for (int q = 0; q < RS / 4; ++q)
#pragma unroll
			for (int k = 0; k < PER; ++k)
			{
				const unsigned r = tid + (unsigned)k * NTHREADS;
				if (r < UP_PLANE_ROWS)
				{
					int e[4] = {(int)v[q][k].x, (int)v[q][k].y, (int)v[q][k].z, (int)v[q][k].w};
#pragma unroll
					for (int j = 0; j < 4; ++j)
					{
						const int slot = 4 * q + j;
						if (slot < TT)
							e[j] = ((NEGMASK >> slot) & 1u) ? -2 * e[j] : 2 * e[j];
						else if (slot == TT && NORM == CRHIP_NORM_U32)
							e[j] = 2 * e[j];   // the reciprocal, doubled: see the normalisation in `one`
					}
					u32x4 w;
					w.x = (unsigned)e[0];
					w.y = (unsigned)e[1];
					w.z = (unsigned)e[2];
					w.w = (unsigned)e[3];
					dst[q * UP_PLANE_ROWS + r] = w;
				}
			}
	}
	__syncthreads();

	// Wave-tiles are TICKETS, as in k_wave: the first by global wave number, every further one from 32 global counter lanes.
	// The XCDs of one chip do not run this kernel at one speed - under sustained load their clocks sit between 1.9 and 2.1 GHz,
	// and with equal static shares the launch lasted as long as the slowest XCD (end times 143 .. 159 us,
	// profiles/r02_kup2_sweep.log) - so whoever is free takes the next tile.  A ticket is a scalar atomic (draw_ticket, cr_device.hpp:
	// through lgkmcnt, not through the vmcnt the stores and the DMA share), drawn a tile ahead of its use, its round trip under the
	// wait for the next window.
	const uint64_t n_tiles = (a.n_out + WT - 1) / WT;
	const uint64_t global_wave = (uint64_t)wave * gridDim.x + blockIdx.x;   // (a short launch spreads over the CUs, not over a CU's waves)
	const uint64_t global_waves = (uint64_t)gridDim.x * WAVES;
	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	const uint64_t lane_tiles = n_tiles > lane_id ? (n_tiles - lane_id + LANES - 1u) / LANES : 0;   // tiles of this counter's sequence
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);          // its waves = its pre-assigned tiles
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	auto draw_issue = [&]() -> unsigned { return draw_ticket(lane_counter); };   // (scalar, the whole wave: cr_device.hpp)
	auto draw_resolve = [&](unsigned got) -> uint64_t {
		const uint64_t k = (uint64_t)lane_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)got);
		return k < lane_tiles ? lane_id + (uint64_t)LANES * k : ~0ull;
	};
	// a wave that has run out of tickets retires; the last wave of a workgroup reports to the global finished counter and the
	// last workgroup zeroes the ticket block for the next launch (see k_wave)
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
	};

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;
	const float inv_increment = __builtin_amdgcn_rcpf((float)a.increment);
	// frames every input position is sure to have: floor(65536 / increment) (wave-uniform; 2 .. 16)
	unsigned n_min = (unsigned)(65536.0f * inv_increment);
	n_min += (n_min + 1u) * a.increment <= 65536u ? 1u : 0u;
	n_min -= n_min * a.increment > 65536u ? 1u : 0u;
	n_min = __builtin_amdgcn_readfirstlane(n_min);

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
		__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)buf, 16, (int)(lane * 16u), 0, 0, 0);
		return shift;
	};

	// smallest k with frac0 + k * increment >= l * 65536: the first frame (relative to the wave-tile) of input position l >= 1
	auto first_frame_of = [&](unsigned l, unsigned frac0) -> unsigned {
		const unsigned x = (l << 16) - frac0;                     // 1 .. 2^22
		unsigned k = (unsigned)((float)x * inv_increment);        // within one of the quotient; made exact below
		k += (__umul24(k, a.increment) < x) ? 1u : 0u;
		k += (__umul24(k, a.increment) < x) ? 1u : 0u;
		k -= (k != 0 && __umul24(k - 1u, a.increment) >= x) ? 1u : 0u;
		k -= (k != 0 && __umul24(k - 1u, a.increment) >= x) ? 1u : 0u;
		return k;
	};

	// one wave-tile: n output frames from `first`, window at `base`; returns the number of store instructions issued
	auto wave_tile = [&](uint64_t first, unsigned n, const unsigned char *base) -> unsigned {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		const unsigned frac0 = (unsigned)(pos & 0xFFFFu);

		// this lane's frames: [k0, k1) = [first frame of position `lane`, first frame of position `lane + 1`), clipped to the tile
		unsigned k1 = first_frame_of(lane + 1u, frac0);
		k1 = k1 < n ? k1 : n;
		unsigned k0 = (unsigned)__shfl_up((int)k1, 1);
		k0 = lane == 0 ? 0u : k0;
		const unsigned count = k1 - k0;
		// wave-uniform trip count: n_min, or one more where some lane has its position's extra frame
		const unsigned trips = ABL == 2 ? 0u : __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(count > n_min) != 0 ? n_min + 1u : n_min);

		// the window of this lane's input position: S = sample * +-32768, once
		int f[TT];
#pragma unroll
		for (int s = 0; s < TT; ++s)
			f[s] = *reinterpret_cast<const int *>(base + (lane + (unsigned)s) * FB);
		int S[TT][CH], B[TT][CH];   // B = all ones where S < 0: the low dword a tap's accumulator pair starts from
#pragma unroll
		for (int s = 0; s < TT; ++s)
		{
			const int k = ((NEGMASK >> s) & 1u) ? -32768 : 32768;
			S[s][0] = __mul24((int)(short)f[s], k);
			S[s][1] = __mul24(f[s] >> 16, k);
			B[s][0] = S[s][0] >> 31;
			B[s][1] = S[s][1] >> 31;
			asm volatile("" : "+v"(B[s][0]), "+v"(B[s][1]));   // keep them in registers: hipcc otherwise re-forms the shift in every frame
		}

		// g = 65536 - fraction of the lane's current frame: the row is g >> 6 (pure-upsampling row index), 16 bytes per row and plane
		unsigned g = 65536u - ((frac0 + __umul24(k0, a.increment)) & 0xFFFFu);
		unsigned stage_at = k0 * UNIT;   // byte offset of the lane's current frame in the staging buffer

		typedef __attribute__((address_space(3))) i32x4 lds_i32x4;
		const unsigned smem_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
		int thirty_one = 31;
		asm volatile("" : "+v"(thirty_one));   // (a register: SDWA takes no literal shift amount)

		auto read_row = [&](unsigned gg, int (&w)[RS]) {
			// (a prefetch past the lane's last frame stays inside the image; bit-field extract + shift-add: one instruction fewer than
			// shift, mask and the add of the segment's base)
			unsigned row;
			asm("v_bfe_u32 %0, %1, 6, 11" : "=v"(row) : "v"(gg));   // (asm: hipcc turns the builtin back into shift + mask)
			const unsigned at = (row << 4) + smem_base;
			const lds_i32x4 *plane0 = (const lds_i32x4 *)(uintptr_t)at;
#pragma unroll
			for (int q = 0; q < RS / 4; ++q)
			{
				const i32x4 v = plane0[q * UP_PLANE_ROWS];   // compile-time stride: the further planes are immediate offsets
				w[4 * q] = v.x;
				w[4 * q + 1] = v.y;
				w[4 * q + 2] = v.z;
				w[4 * q + 3] = v.w;
			}
		};

		auto one = [&](const int (&w)[RS], bool mine) {
			// accumulator pairs pinned to physical registers; the low dword is re-armed by a plain move of the slot's bias register
			// (a v_ashrrev from S instead would save the bias registers but measured 30 % slower per frame at this occupancy:
			// tools/microbench/chainbench.hip, profiles/r02_chainbench.log)
			int lo0, hi0 = 0, lo1, hi1 = 0;
			// How the statement is declared matters as much as what is in it: hipcc treats every asm statement as a possible
			// forwarding hazard and pads with an s_nop whenever a register one statement DEFINES is touched - read or written - by
			// the next instruction with nothing but other asm statements in between.  Declared with the low dword as an output and
			// vcc as the carry-out of both chains, the frame loop carried 15 s_nop per frame beside its 73 VALU instructions
			// (profiles/r03_kup2_nops.log).  So: (1) the two chains take their carry-out in different registers, vcc and
			// s[94:95], and may sit back to back; (2) the low dword is declared as an INPUT only - the multiply-add leaves the
			// product's fraction bits there, which nothing ever reads: the next tap's arming move overwrites it, and that move is
			// hipcc's own instruction, free to follow the statement directly.  (What the compiler believes about v120 / v124 after
			// a statement - "still the bias I put there" - is never used: every tap arms with a different register.)
#define CRHIP_UP2_TAP(LO, HI, VLO, VHI, SAMPLE, WEIGHT, BIAS)                                                                       \
	VLO = (BIAS);                                                                                                                  \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %1, %2, v[" #LO ":" #HI "]" : "+{v" #HI "}"(VHI) : "v"(SAMPLE), "v"(WEIGHT), "{v" #LO "}"(VLO) : "vcc")
#define CRHIP_UP2_TAP_B(LO, HI, VLO, VHI, SAMPLE, WEIGHT, BIAS)                                                                     \
	VLO = (BIAS);                                                                                                                  \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], s[94:95], %1, %2, v[" #LO ":" #HI "]" : "+{v" #HI "}"(VHI) : "v"(SAMPLE), "v"(WEIGHT), "{v" #LO "}"(VLO) : "s94", "s95")
#pragma unroll
			for (int s = 0; s < TT; ++s)
			{
				CRHIP_UP2_TAP(120, 121, lo0, hi0, S[s][0], w[s], B[s][0]);
				CRHIP_UP2_TAP_B(124, 125, lo1, hi1, S[s][1], w[s], B[s][1]);
			}
#undef CRHIP_UP2_TAP_B
#undef CRHIP_UP2_TAP
			(void)lo0;
			(void)lo1;

			int out0, out1;
			if constexpr (NORM == CRHIP_NORM_U32)
			{
				// (acc * reciprocal) / 32768 with C truncation (clownresampler.h:1033).  With p = acc * reciprocal that is
				// floor((2p + (p < 0 ? 65535 : 0)) / 65536): for p = -32768 q - r (0 <= r < 32768) the numerator is -65536 q + (65535 - 2r)
				// with 0 <= 65535 - 2r < 65536.  The rows carry 2 * reciprocal (staged so above), the 65535 is ONE instruction - the
				// sign shift written to the low word of a zero-padded register - and the 64-bit multiply-add and a funnel shift finish:
				// 3 VALU per channel (it was 4 with the 0x7FFF / shift-by-15 form).
				unsigned b0, b1;
				asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(b0) : "v"(thirty_one), "v"(hi0));
				asm("v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD" : "=v"(b1) : "v"(thirty_one), "v"(hi1));
				const long long v0 = (long long)hi0 * (long long)w[TT] + (long long)b0;
				const long long v1 = (long long)hi1 * (long long)w[TT] + (long long)b1;
				out0 = (int)(v0 >> 16);
				out1 = (int)(v1 >> 16);
			}
			else
			{
				out0 = normalise<NORM>(hi0, w[TT]);
				out1 = normalise<NORM>(hi1, w[TT]);
			}

			if (mine)
			{
				if constexpr (OUT16)
					*reinterpret_cast<int *>(my_stage + stage_at) = (clamp_s16(out0) & 0xFFFF) | (clamp_s16(out1) << 16);
				else
				{
					i32x2 q;
					q.x = out0;
					q.y = out1;
					*reinterpret_cast<i32x2 *>(my_stage + stage_at) = q;
				}
			}
		};

		mark(0);
		// the row of frame j + 1 is read before the arithmetic of frame j (two register sets, loop unrolled by two)
		int wa[RS], wb[RS];
		read_row(g, wa);
		for (unsigned j = 0; j < trips; j += 2u)
		{
			read_row(g - a.increment, wb);
			__builtin_amdgcn_sched_barrier(0);
			one(wa, j < count);
			stage_at += UNIT;
			if (j + 1u >= trips)
				break;
			read_row(g - 2u * a.increment, wa);
			__builtin_amdgcn_sched_barrier(0);
			one(wb, j + 1u < count);
			stage_at += UNIT;
			g -= 2u * a.increment;
		}

		mark(1);
		// the staged frames of the other lanes: same wave, LDS operations of a wave complete in order
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// copy-out: wave-uniform base, one lane offset; four LDS reads in flight per trip
		typedef typename std::conditional<VEC == 8, i32x2, int>::type vec_t;
		const unsigned vectors = n * UNIT / VEC;
		const vec_t *staged = reinterpret_cast<const vec_t *>(my_stage) + lane;
		const uint64_t out_first = reinterpret_cast<uint64_t>(a.d_out) + first * UNIT;
		const unsigned out_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)out_first);
		const unsigned out_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(out_first >> 32));
		// (a GLOBAL pointer by type: through a generic one these were flat stores, which go down the LDS path as well)
		typedef __attribute__((address_space(1))) vec_t global_vec;
		global_vec *dst = (global_vec *)(((uint64_t)out_hi << 32) | out_lo);
		auto put = [&](global_vec *to, vec_t v) {
			if constexpr (ABL == 1)
			{
				asm volatile("" ::"v"(v));
				return;
			}
			if constexpr (NT)
				__builtin_nontemporal_store(v, to);
			else
				*to = v;
		};
		unsigned done = 0;   // wave-uniform
		for (; done + 256u <= vectors; done += 256u)
		{
			const vec_t v0 = staged[done], v1 = staged[done + 64u], v2 = staged[done + 128u], v3 = staged[done + 192u];
			put(dst + done + lane, v0);
			put(dst + done + 64u + lane, v1);
			put(dst + done + 128u + lane, v2);
			put(dst + done + 192u + lane, v3);
		}
		for (; done < vectors; done += 64u)
			if (done + lane < vectors)
				put(dst + done + lane, staged[done]);
		__builtin_amdgcn_wave_barrier();
		mark(2);
		return ABL == 1 ? 0u : __builtin_amdgcn_readfirstlane((vectors + 63u) / 64u);
	};

	uint64_t tile = global_wave;
	if (tile >= n_tiles)
	{
		retire();
		finish();
		return;
	}

	unsigned cur = 0, shift = 0;
	uint64_t next;
	{
		const unsigned ticket = draw_issue();
		const uint64_t first = tile * WT;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		shift = fetch(first, n, my_buf);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		next = draw_resolve(ticket);
	}
	mark(-1);

	for (;;)
	{
		const uint64_t first = tile * WT;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);
		const bool have_next = next != ~0ull;
		unsigned shift_next = 0;

		// the DMA of the next tile (its buffer was consumed one step ago): ahead of this tile's stores in vmcnt's order
		if (have_next)
		{
			const uint64_t nf = next * WT;
			const unsigned nn = (unsigned)((a.n_out - nf < WT) ? (a.n_out - nf) : WT);
			shift_next = fetch(nf, nn, my_buf + (cur ^ 1u) * BUF);
		}

		const unsigned stores = wave_tile(first, n, my_buf + cur * BUF + shift);

		if (!have_next)
			break;
		// the ticket for the tile after the next one: its round trip runs under the wait for the next tile's window
		unsigned ticket = draw_ticket_begin(lane_counter);
		wait_vmcnt_at_most(stores);
		ticket = draw_ticket_end(ticket);
		mark(3);
		cur ^= 1u;
		shift = shift_next;
		tile = next;
		next = draw_resolve(ticket);
	}

	retire();
	finish();
}

} // namespace

#endif // CR_KUP_HPP
