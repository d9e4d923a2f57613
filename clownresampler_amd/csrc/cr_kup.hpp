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
//   * The tap is  P = v_mad_i64_i32(X, W, P)  on a 64-bit accumulator pair P = {lo, hi}, in k_wave2's mov-armed form:
//         X  = 2 * sample, negated for the slots whose weights are <= 0          (once per window: one SDWA multiply)
//         W  = |weight| << 15                                                    (once per workgroup, while the rows are staged)
//         lo = X                                                                 (ONE plain move per tap re-arms it)
//     X * W = sample * weight * 65536 exactly (the two sign flips cancel), so the product's integer part lands in `hi` and its
//     16 fraction bits in the top of `lo`.  With every W >= 0 the product is negative exactly where X is, and X ITSELF is a valid
//     arm: the product is a multiple of 65536, so a `lo` in [2^32 - 65536, 2^32) carries into `hi` exactly when the fraction is not
//     zero and a `lo` in [0, 65536] never does: hi += trunc(sample * weight / 65536), C's division (clownresampler.h:1020 via :625).
//     |weight| << 15 needs |weight| < 65536: the two slots around the kernel's centre (mad_safemask<TT>(), weights up to 65536, never
//     negative) keep the plain weight and take S = sample << 16 with the sign-shift arm B = S >> 31 - two registers per sample for
//     those two slots only (round 2's form had S and B for all fifteen: 60 VALU per window and 60 registers; now 38 and 34).
//   * A frame's FIRST tap adds to a pinned pair {X of slot 0, 0}: no zeroing of the accumulator, no arming move for that tap.
//   * The final (acc * reciprocal) / 32768 (clownresampler.h:1033) is one more 64-bit multiply-add on a pinned pair {bias, 0}
//     and a funnel shift.
//   * Every lane runs the SAME number of frames: n_min = floor(65536 / increment) without any predicate - a lane clipped by the
//     tile's end writes its surplus into slack behind the staged frames, lane 0 (whose position may have begun in the previous
//     tile) runs end-aligned and writes its surplus into slack in front - plus ONE predicated frame in the waves where some
//     position has n_min + 1.
//   * The copy-out of the staged frames goes through a buffer descriptor of exactly the tile's bytes: wave-uniform base, one
//     lane offset, no bounds arithmetic (stores beyond the tile are dropped by the range check).
// ---------------------------------------------------------------------------------------------------------
// bytes of slack on either side of a wave's staging buffer (16 surplus frames of 8 bytes, rounded up to 16 bytes)
constexpr unsigned UP2_SLACK = 144u;
// the least a wave's staging buffer holds: 80 converted input frames of 16 bytes (64 positions + a window of up to 16 slots)
constexpr unsigned UP2_ENTRIES_BYTES = 1280u;
//
// FCHAIN = 1: the tap as ONE instruction, on the FLOAT pipe (round 4).  An FP32 accumulator that starts at 2^23 has an ulp of exactly 1,
// and under round-toward-zero fma(|v|, |w| / 65536, acc) = acc + floor(|v| |w| / 65536): the product is exact inside the fused
// operation (15 x 17 bits), the sum is rounded once, toward zero = downward for a positive sum - C's truncating division
// (clownresampler.h:1020 via :625) of a non-negative product, accumulated, no arming, no 64-bit pair.  A negative product
// truncates toward zero the same way on an accumulator that starts at -2^23.  Which of the two a tap feeds is the sign of
// sample x weight - the weight's sign is the slot's (NEGMASK), the sample's is known once per window - so the window is
// unpacked into pairs {max(v, 0), min(v, 0)} of v = +-sample as floats, the rows are staged as |weight| / 65536 (exact: at most
// 17 bits), and v_pk_fma_f32 advances BOTH chains of a channel in one instruction (one of the two adds zero).  The frame's sum is
// (bits of the positive chain) - (bits of the negative chain) + 2^31.  No slot is special (a weight of 65536 is 1.0f), nothing is
// pinned.  |sum| < 2^23 is what the host proves for every 32-bit kernel; the chains' partial sums are bounded by the sum of
// |weight| x 2^15 / 65536 < 2^19 for any table the 32-bit kernels accept.  tools/microbench/rtzchain.hip checks 2.7e8 random
// 15-tap frames against the integer definition (0 differ) and tools/microbench/valurate.hip prices the instructions:
// v_pk_fma_f32 2.0 ns per wave-instruction and SIMD at this occupancy against 2.95 for v_mov_b32 + v_mad_i64_i32.
template <int CH, int TT, int NORM, unsigned NEGMASK, int WAVES, int OUT16, int NT, int ABL = 0, int FCHAIN = 0>
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
	// (at least UP2_ENTRIES_BYTES: the float chain parks the wave-tile's converted input frames there before the first frame is staged)
	const unsigned stage_bytes = ((WT * UNIT + 15u) & ~15u) > UP2_ENTRIES_BYTES ? ((WT * UNIT + 15u) & ~15u) : UP2_ENTRIES_BYTES;
	static_assert(17u * UNIT <= UP2_SLACK, "slack for a lane's surplus frames on either side of the staged tile");
	static_assert((64u + TT - 1u) * 16u <= UP2_ENTRIES_BYTES, "one 16-byte entry per input frame of a wave-tile's window");

	// per wave: two window buffers, slack, the staged frames of a wave-tile, slack (the host sizes the workgroup's LDS for this:
	// cr_context.c, plan_geometry)
	const unsigned rows_bytes = a.plane_rows * a.row_stride * 4u;
	unsigned char *my_buf = smem + rows_bytes + wave * (2u * BUF + stage_bytes + 2u * UP2_SLACK);
	unsigned char *my_stage = my_buf + 2u * BUF + UP2_SLACK;
	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + rows_bytes + WAVES * (2u * BUF + stage_bytes + 2u * UP2_SLACK));

	if (tid == 0)
		*waves_done = 0;

	// Stage the polyphase rows once per workgroup - as W = |weight| << 15 (mad_staged_weight): plane q of the image holds int32
	// [4q, 4q + 4) of every row, i.e. slots 4q .. 4q + 3 (the reciprocal sits in slot TT).  The only barrier of the kernel.
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
						if (slot < TT && FCHAIN)
							e[j] = (int)__float_as_uint((float)(e[j] < 0 ? -e[j] : e[j]) * (1.0f / 65536.0f));   // |weight| / 65536 as a float: exact
						else if (slot < TT)
							e[j] = mad_staged_weight<TT, NEGMASK>(e[j], slot);   // |weight| << 15; the centre slots: the weight as it is
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
		// x < 2^22 is exact as a float, the reciprocal and the product are good to 2^-22 of a quotient below 2^11: the truncated
		// estimate is the answer or one off it, either way
		unsigned k = (unsigned)((float)x * inv_increment);
		k += (__umul24(k, a.increment) < x) ? 1u : 0u;
		k -= (k != 0 && __umul24(k - 1u, a.increment) >= x) ? 1u : 0u;
		return k;
	};

	// one wave-tile: n output frames from `first`, window at `base`; returns the number of store instructions issued
	constexpr unsigned SAFEMASK = mad_safemask<TT>();
	static_assert(TT > 1 && SAFEMASK != 0 && (SAFEMASK & NEGMASK) == 0 && !(SAFEMASK & 1u) && !(NEGMASK & 0u), "slot classes of the instance: slot 0 ordinary, the centre slots never negative");
	auto wave_tile = [&](uint64_t first, unsigned n, const unsigned char *base) -> unsigned {
		const uint64_t pos = a.pos0 + first * (uint64_t)a.increment;
		const unsigned frac0 = (unsigned)(pos & 0xFFFFu);

		// this lane's frames: [k0, k1) = [first frame of position `lane`, first frame of position `lane + 1`), clipped to the tile
		const unsigned k1_whole = first_frame_of(lane + 1u, frac0);   // (where the position ends in the stream, tile or no tile)
		const unsigned k1 = k1_whole < n ? k1_whole : n;
		unsigned k0 = (unsigned)__shfl_up((int)k1, 1);
		k0 = lane == 0 ? 0u : k0;
		const unsigned count = k1 - k0;
		// Every lane runs n_min frames from `start` WITHOUT a predicate, all of them frames of ITS position (the running row index
		// g only ever moves within one position: it does not survive the wrap of the fraction).  A position has n_min or n_min + 1
		// frames in the stream.  A lane clipped by the tile's end keeps its start and writes its surplus at frame numbers >= n:
		// slack behind the staged tile (nobody's frames).  Lane 0's position may have begun before the tile: it runs END-aligned
		// on the position's last n_min frames, [k1_whole - n_min, k1_whole) - a start <= 0, the surplus at negative frame numbers:
		// slack in front (and, where the tile ends before the position does, behind).  The one frame that would land on a
		// neighbour's - the (n_min + 1)-th of a position that has only n_min - is the predicated extra trip.
		const bool extra = count > n_min;
		const int end_aligned = (int)k1_whole - (int)n_min;
		const int start = (lane == 0 && !extra) ? (end_aligned < 0 ? end_aligned : 0) : (int)k0;
		const bool any_extra = ABL != 2 && __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(extra) != 0);
		const unsigned trips = ABL == 2 ? 0u : n_min;

		// the window of this lane's input position, once: X = +-2 * sample (operand AND arm of the ordinary slots); the centre
		// slots: S = sample << 16 and its sign B
		typedef float f32x2 __attribute__((ext_vector_type(2)));
		typedef float f32x4 __attribute__((ext_vector_type(4)));
		f32x2 P[FCHAIN ? TT : 1][CH];   // FCHAIN: {max(v, 0), min(v, 0)} of v = the sample of window frame s, channel c
		int f[FCHAIN ? 1 : TT];
		if constexpr (FCHAIN)
		{
			// The windows of neighbouring lanes overlap in all but one frame, so every input frame of the wave-tile is converted ONCE -
			// lane l converts frames l and 64 + l (the window of the last lane ends at frame 63 + TT - 1) - into {max(v, 0), min(v, 0)}
			// per channel, 16 bytes, parked in the staging buffer (free until the first frame is staged: the LDS operations of a wave
			// complete in order), and every lane then reads its TT entries back: 12 VALU per wave-tile where a lane converting its own
			// window took 90.  A slot whose weights are negative takes the pair swapped and negated, which the multiply-add does for
			// free (op_sel / neg_lo / neg_hi): {max(-v, 0), min(-v, 0)} = {-min(v, 0), -max(v, 0)}.
			static_assert(64 + 63 < (int)(BUF / FB) && TT <= 64 && (2u * BUF + UP2_SLACK) % 16u == 0, "the second frame of every lane lies inside the window buffer; 16-byte entries");
			auto convert = [&](int packed) {
				float v0, v1;
				f32x4 e;
				asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(v0) : "v"(packed));
				asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(v1) : "v"(packed));
				asm("v_max_f32_e64 %0, %1, 0" : "=v"(e.x) : "v"(v0));
				asm("v_min_f32_e64 %0, %1, 0" : "=v"(e.y) : "v"(v0));
				asm("v_max_f32_e64 %0, %1, 0" : "=v"(e.z) : "v"(v1));
				asm("v_min_f32_e64 %0, %1, 0" : "=v"(e.w) : "v"(v1));
				return e;
			};
			f32x4 *entries = reinterpret_cast<f32x4 *>(my_stage);
			const int own = *reinterpret_cast<const int *>(base + lane * FB);
			const int beyond = *reinterpret_cast<const int *>(base + (64u + lane) * FB);   // (lanes from TT - 1 on: past the window, never read back)
			entries[lane] = convert(own);
			if (lane < (unsigned)TT - 1u)
				entries[64u + lane] = convert(beyond);
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
			for (int s = 0; s < TT; ++s)
			{
				const f32x4 e = entries[lane + (unsigned)s];
				P[s][0].x = e.x;
				P[s][0].y = e.y;
				P[s][1].x = e.z;
				P[s][1].y = e.w;
			}
			// (the frames' staging writes below come after these reads: same wave, in order)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
		else
		{
#pragma unroll
			for (int s = 0; s < TT; ++s)
				f[s] = *reinterpret_cast<const int *>(base + (lane + (unsigned)s) * FB);
		}
		int X[TT][CH], B[TT][CH];
		int plus_two = 2, minus_two = -2;
#pragma unroll
		for (int s = 0; s < (FCHAIN ? 0 : TT); ++s)
		{
			if ((SAFEMASK >> s) & 1u)
			{
				X[s][0] = (int)((unsigned)f[s] << 16);
				X[s][1] = (int)((unsigned)f[s] & 0xFFFF0000u);
				asm volatile("" : "+v"(X[s][0]), "+v"(X[s][1]));   // (registers: hipcc otherwise re-forms them in every frame)
				B[s][0] = X[s][0] >> 31;
				B[s][1] = X[s][1] >> 31;
				asm volatile("" : "+v"(B[s][0]), "+v"(B[s][1]));
			}
			else
			{
				// one SDWA multiply per sample, straight from the packed frame (written out: for the factor + 2 hipcc prefers a shift
				// pair per sample); the result is operand AND arm - B is the same register
				const int k = ((NEGMASK >> s) & 1u) ? minus_two : plus_two;
				asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(X[s][0]) : "v"(f[s]), "v"(k));
				asm("v_mul_i32_i24_sdwa %0, sext(%1), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(X[s][1]) : "v"(f[s]), "v"(k));
				B[s][0] = X[s][0];
				B[s][1] = X[s][1];
			}
		}

		// g = 65536 - fraction of the lane's current frame: the row is g >> 6 (pure-upsampling row index), 16 bytes per row and plane
		unsigned g = 65536u - ((unsigned)((int)frac0 + __mul24(start, (int)a.increment)) & 0xFFFFu);
		unsigned stage_at = (unsigned)(start * (int)UNIT);   // byte offset of the lane's current frame from my_stage (lane 0: may be negative)
		// (timing-only ablations 4 / 5: every step's 64 frames side by side - the staging writes free of bank conflicts, results WRONG)
		constexpr unsigned STAGE_STEP = (ABL == 4 || ABL == 5) ? 56u * UNIT : UNIT;
		if constexpr (ABL == 4 || ABL == 5)
			stage_at = lane * UNIT;

		typedef __attribute__((address_space(3))) i32x4 lds_i32x4;
		const unsigned smem_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)smem);
		int thirty_one = 31;
		asm volatile("" : "+v"(thirty_one));   // (a register: SDWA takes no literal shift amount)
		// the high halves of the four pinned addend pairs: zeros the compiler must not know to be zeros (it would re-materialise a
		// constant next to every use - four moves per frame - instead of keeping v113 / v115 / v117 / v119 set across the loop)
		int zero_a = 0, zero_b = 0, zero_c = 0, zero_d = 0;
		asm volatile("" : "+{v113}"(zero_a), "+{v115}"(zero_b), "+{v117}"(zero_c), "+{v119}"(zero_d));

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

		f32x2 chain_base;   // FCHAIN: where a frame's two chains start
		chain_base.x = 8388608.0f;
		chain_base.y = -8388608.0f;
		asm volatile("" : "+v"(chain_base));
		// PRED: the frame is stored only by the lanes with `extra`
		auto one = [&](const int (&w)[RS], auto pred_tag) {
			constexpr bool PRED = decltype(pred_tag)::value;
			// Accumulator pairs pinned to physical registers: v[120:121] / v[124:125]; the first tap's addend pairs {X of slot 0, 0}:
			// v[112:113] / v[114:115]; the normalisation's addend pairs {bias, 0}: v[116:117] / v[118:119], its products v[122:123] /
			// v[126:127].  How a statement is DECLARED matters as much as what is in it: hipcc treats every asm statement as a
			// possible forwarding hazard and pads with an s_nop whenever a register one statement DEFINES is touched - read or
			// written - by the next instruction with nothing but other asm statements in between (15 s_nop per frame beside 73 VALU
			// in round 3's first form, profiles/r03_kup2_nops.log).  So (1) the two chains take their carry-out in different
			// registers, vcc and s[94:95], and may sit back to back; (2) a tap declares its low dword as an INPUT and a CLOBBER-free
			// one: the multiply-add leaves the product's fraction bits there, which nothing ever reads - the next tap's arming move
			// (hipcc's own instruction, free to follow the statement directly) overwrites it, and the last tap's is dead.  What
			// the compiler may wrongly believe - "v120 still holds the arm I moved there" - it never uses: every tap arms with a
			// different register (static_assert(TT > 1) above; tests/test_gpu_parity.py::test_input_stationary_* run every instance
			// against the oracle).
			int lo0, hi0, lo1, hi1;
			if constexpr (FCHAIN)
			{
				// one v_pk_fma_f32 per tap and channel: {positive chain, negative chain} += {max(v, 0), min(v, 0)} * (|weight| / 65536),
				// round toward zero (set around the frame loop); the weight is the low or the high half of a register pair of the row
				// (Written as one or two asm statements, not one per tap: between asm statements that touch a common register hipcc pads
				// with s_nop - 10 per frame here - while dependent VALU instructions inside a statement need none: the hardware
				// interlocks them.  An asm statement takes at most 30 operands: 15 slots are two statements.)
				f32x2 a0, a1;
				auto wpair = [&](int k) {
					f32x2 wp;
					wp.x = __uint_as_float((unsigned)w[2 * k]);
					wp.y = __uint_as_float((unsigned)w[2 * k + 1]);
					return wp;
				};
// E / O: the weight is the low / high half of its register pair; P / N: the slot's weights are >= 0 / <= 0 (N: the sample pair swapped
// and negated); F: the first tap of a frame, which adds to the chains' base
#define CR_SEL_EP " op_sel_hi:[1,0,1]\n\t"
#define CR_SEL_OP " op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define CR_SEL_EN " op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define CR_SEL_ON " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
#define CR_T(acc, p, wq, SEL) "v_pk_fma_f32 %" #acc ", %" #p ", %" #wq ", %" #acc SEL
#define CR_F(acc, p, wq, base, SEL) "v_pk_fma_f32 %" #acc ", %" #p ", %" #wq ", %" #base SEL
				if constexpr (TT == 15)
				{
					static_assert(NEGMASK == 0x2A55u, "slots 0, 2, 4, 6, 9, 11, 13 negative: the strings below are written for this mask");
					asm volatile(CR_F(0, 2, 18, 22, CR_SEL_EN) CR_F(1, 10, 18, 22, CR_SEL_EN) CR_T(0, 3, 18, CR_SEL_OP) CR_T(1, 11, 18, CR_SEL_OP)
					             CR_T(0, 4, 19, CR_SEL_EN) CR_T(1, 12, 19, CR_SEL_EN) CR_T(0, 5, 19, CR_SEL_OP) CR_T(1, 13, 19, CR_SEL_OP)
					             CR_T(0, 6, 20, CR_SEL_EN) CR_T(1, 14, 20, CR_SEL_EN) CR_T(0, 7, 20, CR_SEL_OP) CR_T(1, 15, 20, CR_SEL_OP)
					             CR_T(0, 8, 21, CR_SEL_EN) CR_T(1, 16, 21, CR_SEL_EN) CR_T(0, 9, 21, CR_SEL_OP) CR_T(1, 17, 21, CR_SEL_OP)
					             : "=&v"(a0), "=&v"(a1)
					             : "v"(P[0][0]), "v"(P[1][0]), "v"(P[2][0]), "v"(P[3][0]), "v"(P[4][0]), "v"(P[5][0]), "v"(P[6][0]), "v"(P[7][0]),
					               "v"(P[0][1]), "v"(P[1][1]), "v"(P[2][1]), "v"(P[3][1]), "v"(P[4][1]), "v"(P[5][1]), "v"(P[6][1]), "v"(P[7][1]),
					               "v"(wpair(0)), "v"(wpair(1)), "v"(wpair(2)), "v"(wpair(3)), "v"(chain_base));
					// slots 8 (+), 9 (-), 10 (+), 11 (-), 12 (+), 13 (-), 14 (+)
					asm volatile(CR_T(0, 2, 16, CR_SEL_EP) CR_T(1, 9, 16, CR_SEL_EP) CR_T(0, 3, 16, CR_SEL_ON) CR_T(1, 10, 16, CR_SEL_ON)
					             CR_T(0, 4, 17, CR_SEL_EP) CR_T(1, 11, 17, CR_SEL_EP) CR_T(0, 5, 17, CR_SEL_ON) CR_T(1, 12, 17, CR_SEL_ON)
					             CR_T(0, 6, 18, CR_SEL_EP) CR_T(1, 13, 18, CR_SEL_EP) CR_T(0, 7, 18, CR_SEL_ON) CR_T(1, 14, 18, CR_SEL_ON)
					             CR_T(0, 8, 19, CR_SEL_EP) CR_T(1, 15, 19, CR_SEL_EP)
					             : "+v"(a0), "+v"(a1)
					             : "v"(P[8][0]), "v"(P[9][0]), "v"(P[10][0]), "v"(P[11][0]), "v"(P[12][0]), "v"(P[13][0]), "v"(P[14][0]),
					               "v"(P[8][1]), "v"(P[9][1]), "v"(P[10][1]), "v"(P[11][1]), "v"(P[12][1]), "v"(P[13][1]), "v"(P[14][1]),
					               "v"(wpair(4)), "v"(wpair(5)), "v"(wpair(6)), "v"(wpair(7)));
				}
				else
				{
					static_assert(TT == 5 && NEGMASK == 0x12u, "k_up2's float chain is written out for 15 slots (mask 0x2A55) and 5 slots (mask 0x12: slots 1 and 4 negative)");
					asm volatile(CR_F(0, 2, 12, 15, CR_SEL_EP) CR_F(1, 7, 12, 15, CR_SEL_EP) CR_T(0, 3, 12, CR_SEL_ON) CR_T(1, 8, 12, CR_SEL_ON)
					             CR_T(0, 4, 13, CR_SEL_EP) CR_T(1, 9, 13, CR_SEL_EP) CR_T(0, 5, 13, CR_SEL_OP) CR_T(1, 10, 13, CR_SEL_OP)
					             CR_T(0, 6, 14, CR_SEL_EN) CR_T(1, 11, 14, CR_SEL_EN)
					             : "=&v"(a0), "=&v"(a1)
					             : "v"(P[0][0]), "v"(P[1][0]), "v"(P[2][0]), "v"(P[3][0]), "v"(P[4][0]),
					               "v"(P[0][1]), "v"(P[1][1]), "v"(P[2][1]), "v"(P[3][1]), "v"(P[4][1]),
					               "v"(wpair(0)), "v"(wpair(1)), "v"(wpair(2)), "v"(chain_base));
				}
#undef CR_F
#undef CR_T
#undef CR_SEL_ON
#undef CR_SEL_EN
#undef CR_SEL_OP
#undef CR_SEL_EP
				// positive chain = 2^23 + p (bits 0x4B000000 + p), negative chain = -(2^23 + q) (bits 0xCB000000 + q): p - q
				hi0 = (int)(__float_as_uint(a0.x) - __float_as_uint(a0.y) + 0x80000000u);
				hi1 = (int)(__float_as_uint(a1.x) - __float_as_uint(a1.y) + 0x80000000u);
			}
			else
			{
			asm("v_mad_i64_i32 v[120:121], vcc, %1, %2, v[112:113]" : "={v121}"(hi0) : "v"(X[0][0]), "v"(w[0]), "{v112}"(X[0][0]), "{v113}"(zero_a) : "vcc", "v120");
			asm("v_mad_i64_i32 v[124:125], s[94:95], %1, %2, v[114:115]" : "={v125}"(hi1) : "v"(X[0][1]), "v"(w[0]), "{v114}"(X[0][1]), "{v115}"(zero_b) : "s94", "s95", "v124");
#define CRHIP_UP2_TAP(LO, HI, VLO, VHI, SAMPLE, WEIGHT, ARM)                                                                        \
	VLO = (ARM);                                                                                                                   \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], vcc, %1, %2, v[" #LO ":" #HI "]" : "+{v" #HI "}"(VHI) : "v"(SAMPLE), "v"(WEIGHT), "{v" #LO "}"(VLO) : "vcc")
#define CRHIP_UP2_TAP_B(LO, HI, VLO, VHI, SAMPLE, WEIGHT, ARM)                                                                      \
	VLO = (ARM);                                                                                                                   \
	asm("v_mad_i64_i32 v[" #LO ":" #HI "], s[94:95], %1, %2, v[" #LO ":" #HI "]" : "+{v" #HI "}"(VHI) : "v"(SAMPLE), "v"(WEIGHT), "{v" #LO "}"(VLO) : "s94", "s95")
#pragma unroll
			for (int s = 1; s < TT; ++s)
			{
				CRHIP_UP2_TAP(120, 121, lo0, hi0, X[s][0], w[s], B[s][0]);
				CRHIP_UP2_TAP_B(124, 125, lo1, hi1, X[s][1], w[s], B[s][1]);
			}
#undef CRHIP_UP2_TAP_B
#undef CRHIP_UP2_TAP
			}
			(void)lo0;
			(void)lo1;

			int out0, out1;
			if constexpr (NORM == CRHIP_NORM_U32)
			{
				// (acc * reciprocal) / 32768 with C truncation (clownresampler.h:1033).  With p = acc * reciprocal that is
				// floor((2p + (p < 0 ? 65535 : 0)) / 65536): for p = -32768 q - r (0 <= r < 32768) the numerator is -65536 q + (65535 - 2r)
				// with 0 <= 65535 - 2r < 65536.  The rows carry 2 * reciprocal (staged so above), the 65535 is ONE instruction - the
				// sign shift written to the low word of a zero-padded register, which is the low half of a pinned pair {bias, 0} -
				// and the 64-bit multiply-add and a funnel shift finish: 3 VALU per channel.  One statement for both channels: the
				// two SDWA writes are each an instruction away from the multiply-add that reads them (dst-forwarding needs one).
				asm("v_ashrrev_i32_sdwa v116, %4, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
				    "v_ashrrev_i32_sdwa v118, %4, %3 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
				    "v_mad_i64_i32 v[122:123], vcc, %2, %5, v[116:117]\n\t"
				    "v_mad_i64_i32 v[126:127], vcc, %3, %5, v[118:119]\n\t"
				    "v_alignbit_b32 %0, v123, v122, 16\n\t"
				    "v_alignbit_b32 %1, v127, v126, 16"
				    : "=&v"(out0), "=&v"(out1)
				    : "v"(hi0), "v"(hi1), "v"(thirty_one), "v"(w[TT]), "{v117}"(zero_c), "{v119}"(zero_d)
				    : "vcc", "v116", "v118", "v122", "v123", "v126", "v127");
			}
			else
			{
				out0 = normalise<NORM>(hi0, w[TT]);
				out1 = normalise<NORM>(hi1, w[TT]);
			}

			if (!PRED || extra)
			{
				if constexpr (OUT16)
					*reinterpret_cast<int *>(my_stage + (int)stage_at) = (clamp_s16(out0) & 0xFFFF) | (clamp_s16(out1) << 16);
				else
				{
					i32x2 q;
					q.x = out0;
					q.y = out1;
					*reinterpret_cast<i32x2 *>(my_stage + (int)stage_at) = q;
				}
			}
		};

		mark(0);
		unsigned fp_mode = 0;   // (the wave's FP32 rounding mode is put back as it was found: ADVICE r4)
		if constexpr (FCHAIN)
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE, 0, 2)\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3" : "=s"(fp_mode));   // FP32 rounding: toward zero, for the chains of the frames
		// the row of frame j + 1 is read before the arithmetic of frame j (two register sets, loop unrolled by two)
		int wa[RS], wb[RS];
		read_row(g, wa);
		// (timing-only ablations 3 / 5: the row of the lane's first frame serves all of its frames - no row reads in the loop, results WRONG)
		constexpr bool ROWS_ONCE = ABL == 3 || ABL == 5;
		if constexpr (ROWS_ONCE)
		{
#pragma unroll
			for (int i = 0; i < RS; ++i)
				wb[i] = wa[i];
		}
		for (unsigned j = 0; j < trips; j += 2u)
		{
			if constexpr (!ROWS_ONCE)
				read_row(g - a.increment, wb);
			__builtin_amdgcn_sched_barrier(0);
			one(wa, std::false_type());
			stage_at += STAGE_STEP;
			g -= a.increment;
			if (j + 1u >= trips)
			{
#pragma unroll
				for (int i = 0; i < RS; ++i)
					wa[i] = wb[i];
				break;
			}
			if constexpr (!ROWS_ONCE)
				read_row(g - a.increment, wa);
			__builtin_amdgcn_sched_barrier(0);
			one(wb, std::false_type());
			stage_at += STAGE_STEP;
			g -= a.increment;
		}
		if (any_extra)
			one(wa, std::true_type());   // (its row was read as the prefetch of the last trip)
		if constexpr (FCHAIN)
			asm volatile("s_setreg_b32 hwreg(HW_REG_MODE, 0, 2), %0" ::"s"(fp_mode));   // back to the mode the wave came with

		mark(1);
		// the staged frames of the other lanes: same wave, LDS operations of a wave complete in order
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// copy-out: a buffer descriptor of exactly the tile's bytes (wave-uniform), one lane offset; four LDS reads in flight per
		// trip; what a trip reads or would write beyond the tile is slack / dropped by the descriptor's range check
		typedef typename std::conditional<VEC == 8, i32x2, int>::type vec_t;
		const unsigned vectors = n * UNIT / VEC;
		const vec_t *staged = reinterpret_cast<const vec_t *>(my_stage) + lane;
		const uint64_t out_first = reinterpret_cast<uint64_t>(a.d_out) + first * UNIT;
		const unsigned out_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)out_first);
		const unsigned out_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(out_first >> 32));
		const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)out_hi << 32) | out_lo), 0,
		                                                                          (int)__builtin_amdgcn_readfirstlane((int)(n * UNIT)), 0x00020000);
		auto put = [&](unsigned at_vector, vec_t v) {   // at_vector: wave-uniform
			if constexpr (ABL == 1)
			{
				asm volatile("" ::"v"(v));
				return;
			}
			if constexpr (VEC == 8)
				__builtin_amdgcn_raw_buffer_store_b64(v, out_rsrc, (int)(lane * VEC), (int)(at_vector * VEC), NT ? 2 : 0);
			else
				__builtin_amdgcn_raw_buffer_store_b32(v, out_rsrc, (int)(lane * VEC), (int)(at_vector * VEC), NT ? 2 : 0);
		};
		unsigned stores = 0;   // wave-uniform
		for (unsigned done = 0; done < vectors; done += 256u)
		{
			const vec_t v0 = staged[done], v1 = staged[done + 64u], v2 = staged[done + 128u], v3 = staged[done + 192u];
			put(done, v0);
			put(done + 64u, v1);
			put(done + 128u, v2);
			put(done + 192u, v3);
			stores += 4u;
		}
		__builtin_amdgcn_wave_barrier();
		mark(2);
		return ABL == 1 ? 0u : __builtin_amdgcn_readfirstlane(stores);
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
