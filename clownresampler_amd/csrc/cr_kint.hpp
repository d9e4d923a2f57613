// cr_kint.hpp - k_int: the input-stationary kernel for WHOLE-NUMBER downsampling ratios (2:1, 3:1, 4:1, 6:1 ...).
#ifndef CR_KINT_HPP
#define CR_KINT_HPP

#include "cr_device.hpp"
#include "cr_kup.hpp"   // wait_vmcnt_at_most

namespace
{

// ---------------------------------------------------------------------------------------------------------
// k_int - increment = R << 16 exactly: every output frame of a launch has the SAME fractional position, hence the same row of
// weights (clownresampler.h:993-1008 depend on position_fractional only), and consecutive frames' windows are exactly R
// input frames apart.  Two things follow, and neither is available to the lane-per-frame kernels (k_poly / k_wave2), which
// read a row AND a private tap window from LDS per frame (12 bytes per stereo tap-slot: LDS 56 % busy beside a VALU 47 % busy
// on the 33-slot windows, profiles/r03_dn8_pmc_summary.txt):
//   * the weights are wave-uniform: they arrive in the KERNEL ARGUMENTS (the host picks the launch's row, checks it against
//     the instance's slot classes and stages it), i.e. in SGPRs - no row image in LDS, no row reads, no row index;
//   * a lane that owns K CONSECUTIVE output frames reads its R (K - 1) + TT input frames ONCE, packed, straight into
//     registers (ds_read_b128 at a lane stride chosen to be conflict-free), unpacks every sample once
//     (X = 2 * sample, one SDWA shift) and uses it in up to TT / R taps of different frames.
// The tap is k_wave2's mov-armed 64-bit multiply-add (cr_kwave2.hpp): P = {X, acc}; P = v_mad_i64_i32(X, W, P) with
// W = |weight| << 15 in an SGPR: `acc += trunc(sample * |weight| / 65536)`, C's division (clownresampler.h:1020 via :625);
// slots whose weights are negative accumulate into a second pair that is subtracted at the end (truncation toward zero is
// odd-symmetric).  Which slots those are is a compile-time property of the instance (NEGMASK; the lobes of the stretched
// kernel are R slots wide) that the host verifies against the row of EVERY launch; a row that does not match (a resumed
// stream whose fraction is not the one the instance was derived for, a caller's own table) takes the plan's ordinary kernel.
// Slots in SAFEMASK may hold weights up to 65536 (the kernel's centre): plain |weight| and X << 15 there.
// Streaming is k_up2's: wave-autonomous, LDS-DMA of the wave-tile's packed window (ONE buffer per wave: the lanes hold
// the whole tile in registers before the arithmetic starts, so the next tile's DMA is issued into the same buffer right
// after the reads), results staged through LDS so that the K frames of a lane - consecutive in the output - leave as
// coalesced stores, counted vmcnt.  No barrier at all.
// The final (acc * reciprocal) / 32768 is the 64-bit form (either range class).
//   CH channels (1 to 8: a frame's channel pairs one after the other)   R input frames per output frame   TT slots   K frames per lane   WAVES per workgroup
// ---------------------------------------------------------------------------------------------------------
// Accumulator pairs pinned to physical registers v[INT_ACC_BASE + 2 P : + 1] (the constraint of the asm statement: the tap's
// multiply-add).  Left to its own allocation hipcc assembled the pair {x, acc} with v_mov_b64 and took it apart again - 1.4 moves
// per tap instead of 1 (profiles/r03_kint_first.log: 214 VALU per frame where the taps are 144).  The arming move stays in C:
// hipcc assumes a forwarding hazard between two ADJACENT asm statements that define a common register (here: vcc) and pads them
// with an s_nop - one per tap when the move was part of the statement; its own v_mov_b32 between two statements is a wait
// state it can count.
constexpr int INT_ACC_BASE = 64;
template <int P>
__device__ __forceinline__ void int_tap(int &lo, int &hi, int multiplicand, int weight);
#define CRK_INT_TAP(P, LO, HI)                                                                                                      \
	template <>                                                                                                                    \
	__device__ __forceinline__ void int_tap<P>(int &lo, int &hi, int multiplicand, int weight)                                      \
	{                                                                                                                              \
		asm("v_mad_i64_i32 [%0,%1], vcc, %2, %3, [%0,%1]" : "+{v" #LO "}"(lo), "+{v" #HI "}"(hi) : "v"(multiplicand), "s"(weight) : "vcc"); \
	}
CRK_INT_TAP(0, 64, 65) CRK_INT_TAP(1, 66, 67) CRK_INT_TAP(2, 68, 69) CRK_INT_TAP(3, 70, 71) CRK_INT_TAP(4, 72, 73) CRK_INT_TAP(5, 74, 75)
CRK_INT_TAP(6, 76, 77) CRK_INT_TAP(7, 78, 79) CRK_INT_TAP(8, 80, 81) CRK_INT_TAP(9, 82, 83) CRK_INT_TAP(10, 84, 85) CRK_INT_TAP(11, 86, 87)
CRK_INT_TAP(12, 88, 89) CRK_INT_TAP(13, 90, 91) CRK_INT_TAP(14, 92, 93) CRK_INT_TAP(15, 94, 95) CRK_INT_TAP(16, 96, 97) CRK_INT_TAP(17, 98, 99)
CRK_INT_TAP(18, 100, 101) CRK_INT_TAP(19, 102, 103) CRK_INT_TAP(20, 104, 105) CRK_INT_TAP(21, 106, 107) CRK_INT_TAP(22, 108, 109) CRK_INT_TAP(23, 110, 111)
#undef CRK_INT_TAP

// Periodic ratios (P > 1: the fractional position repeats every P output frames, which consume R input frames together): frame k of a
// lane is phase k % P of period k / P and its window starts int_start(k) input frames after the lane's first (OFFS: the phases'
// starts within a period, one byte each, phase 0's is 0).  P == 1: the whole-number ratios, int_start(k) = R k.
constexpr int int_start(int r, int p, unsigned offs, int k) { return r * (k / p) + (int)((offs >> (8 * (k % p))) & 0xFFu); }
constexpr int int_input_frames(int r, int p, unsigned offs, int tt, int k)   // input frames a lane reads for its k output frames
{
	int most = 0;
	for (int i = 0; i < k; ++i)
		most = int_start(r, p, offs, i) + tt > most ? int_start(r, p, offs, i) + tt : most;
	return most;
}
constexpr unsigned int_lane_bytes(int ch, int r, int k, int p = 1) { return (unsigned)(r * (k / p) * ch * 2); }
constexpr unsigned int_lane_vecs(int ch, int r, int tt, int k, int p = 1, unsigned offs = 0) { return ((unsigned)(int_input_frames(r, p, offs, tt, k) * ch * 2) + 15u) / 16u; }
// bytes of LDS a wave-tile's window occupies (whole 1 KiB DMA pieces)
constexpr unsigned int_window_bytes(int ch, int r, int tt, int k, int p = 1, unsigned offs = 0)
{
	return (63u * int_lane_bytes(ch, r, k, p) + 16u * int_lane_vecs(ch, r, tt, k, p, offs) + 1023u) & ~1023u;
}
constexpr unsigned int_stage_bytes(int ch, int k, int out16) { return 64u * (unsigned)k * (unsigned)ch * (out16 ? 2u : 4u); }
constexpr unsigned int_wave_bytes(int ch, int r, int tt, int k, int out16, int p = 1, unsigned offs = 0)
{
	return int_window_bytes(ch, r, tt, k, p, offs) + ((int_stage_bytes(ch, k, out16) + 15u) & ~15u);
}

template <int CH, int R, int TT, int K, unsigned long long NEGMASK, unsigned long long SAFEMASK, int WAVES, int OUT16, int NT, int P = 1, unsigned OFFS = 0,
          unsigned long long ZEROMASK = 0, int OS = (P > 1)>
__global__ __launch_bounds__(WAVES * 64) void k_int(const crhip_int_launch a)
{
	static_assert(CH >= 1 && CH <= 16, "one to sixteen channels: two at a time over the same accumulator registers");
	static_assert(TT * P <= CRHIP_INT_MAX_SLOTS, "the weights travel in the kernel arguments");
	static_assert((P == 1 || P == 2 || P == 4) && K % P == 0 && (OFFS & 0xFFu) == 0, "whole periods per lane; phase 0 starts the period");
	// ZEROMASK (periodic ratios): phase-slots whose weight is 0 in the rows the instance is for - pure upsampling's phase 0 is the input
	// sample itself, one slot of five - are left out of the arithmetic; the host checks them like the signs.
	static_assert((ZEROMASK & (NEGMASK | SAFEMASK)) == 0 && (P > 1 || ZEROMASK == 0), "a skipped slot has no class");
	// OS: the output-stationary loop order (below) - always for periodic ratios, and for whole-number ratios whose windows are too long
	// for the input-stationary order's accumulators (the 5- and 8-lobe tables: 10 / 16 frames in progress per lane)
	static_assert(OS || P == 1, "periodic ratios take the output-stationary order");
	constexpr unsigned FB = CH * 2;                        // bytes per input frame
	constexpr unsigned UNIT = OUT16 ? CH * 2 : CH * 4;     // bytes per output frame
	constexpr unsigned WT = 64u * K;                       // output frames per wave-tile
	constexpr int NX = int_input_frames(R, P, OFFS, TT, K);   // input frames a lane reads (P == 1: R (K - 1) + TT)
	constexpr unsigned LANE_BYTES = int_lane_bytes(CH, R, K, P);
	constexpr int XV = (int)int_lane_vecs(CH, R, TT, K, P, OFFS);   // ds_read_b128 per lane
	constexpr unsigned WIN = int_window_bytes(CH, R, TT, K, P, OFFS);
	constexpr int NVW = (int)(WIN / 1024u);
	constexpr unsigned STAGE = (int_stage_bytes(CH, K, OUT16) + 15u) & ~15u;
	// a lane's window starts LANE_BYTES after its neighbour's: 16-byte reads where that is a multiple of 16, 8- or 4-byte reads where not
	constexpr unsigned VECB = LANE_BYTES % 16u == 0 ? 16u : (LANE_BYTES % 8u == 0 ? 8u : 4u);
	static_assert(LANE_BYTES % 4u == 0, "a lane's window starts on a dword of the tile's");
	static_assert((K * UNIT) % 4u == 0, "a lane's frames are staged as dwords");

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	const unsigned tid = threadIdx.x;
	const unsigned lane = tid & 63u;
	const unsigned wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	unsigned char *my_buf = smem + wave * (WIN + STAGE);
	unsigned char *my_stage = my_buf + WIN;

	const uint64_t n_tiles = (a.n_out + WT - 1) / WT;
	const uint64_t global_wave = (uint64_t)wave * gridDim.x + blockIdx.x;   // (a short launch spreads over the CUs, not over a CU's waves)
	const uint64_t global_waves = (uint64_t)gridDim.x * WAVES;

	// Wave-tiles are TICKETS where the launch brings counters (long launches): the first by global wave number, every further one
	// from 32 global counter lanes, as in k_up2 - the XCDs of one chip hold different clocks under this kind of load (1.9-2.1 GHz),
	// and with equal static shares a launch lasts as long as its slowest XCD.  A ticket is a scalar atomic (draw_ticket: through
	// lgkmcnt, not through the vmcnt the stores and the DMA share).  The last wave of the last workgroup zeroes the block again.
	// A ticket is worth G consecutive tiles: 8,192 waves drawing 75,000 single-tile tickets from 32 counters were bound by the atomics
	// (mono 6:1: 66.5 us static, 71.9 us ticketed; profiles/r03_kint_tickets.log).  The unit of scheduling is the GROUP below.
	const bool ticketed = a.d_tickets != nullptr;
	const unsigned G = ticketed ? a.ticket_tiles : 1u;
	const uint64_t n_groups = (n_tiles + G - 1u) / G;
	const unsigned LANES = global_waves < 32u ? (unsigned)global_waves : 32u;
	const unsigned lane_id = (unsigned)(global_wave % LANES);
	const uint64_t lane_tiles = n_groups > lane_id ? (n_groups - lane_id + LANES - 1u) / LANES : 0;   // groups of this counter's sequence
	const unsigned lane_waves = (unsigned)((global_waves - lane_id + LANES - 1u) / LANES);
	unsigned *lane_counter = a.d_tickets + lane_id * 32u;
	auto resolve = [&](unsigned got) -> uint64_t {
		const uint64_t k = (uint64_t)lane_waves + (unsigned)__builtin_amdgcn_readfirstlane((int)got);
		return k < lane_tiles ? lane_id + (uint64_t)LANES * k : ~0ull;
	};
	unsigned *waves_done = reinterpret_cast<unsigned *>(smem + WAVES * (WIN + STAGE));
	auto retire = [&]() {
		if (!ticketed)
			return;
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
	if (ticketed)
	{
		if (tid == 0)
			*waves_done = 0;
		__syncthreads();   // (the kernel's only barrier: waves_done is zero before any wave can retire)
	}
	if (global_wave >= n_groups)
	{
		retire();
		return;
	}

	const uint64_t in_base = reinterpret_cast<uint64_t>(a.d_in);
	const uint64_t in_end = in_base + a.in_valid_bytes;

	// LDS-DMA of the packed window of wave-tile `tile`, from ITS first byte: LDS-DMA takes a source of any alignment, 2-byte
	// included, at full rate (tools/microbench/dmaalign.hip, profiles/r03_dmaalign.log), so the window always lands at the
	// start of the buffer and every lane's share on a 16-byte boundary.  Bytes beyond the caller's buffer arrive as zeros.
	// The descriptor's range check works on whole dwords COUNTED FROM `from`: where the readable bytes end half a dword into one
	// (a 2-byte-aligned window over a buffer that ends 2 bytes past a dword of it), that last sample is not delivered - and
	// rounding the range up instead, as k_poly does from ITS 16-byte-aligned base, would read past the caller's buffer here.
	// fetch() returns the LDS byte offset of such a sample (else ~0) and patch() stores it by hand once the DMA has landed.
	auto fetch = [&](uint64_t tile) -> unsigned {
		const uint64_t from = in_base + (a.first_frame + tile * (uint64_t)(WT / P * R)) * FB;
		uint64_t want = WIN;
		const uint64_t avail = in_end > from ? in_end - from : 0;
		unsigned lost = ~0u;
		if (want > avail)
		{
			want = avail & ~(uint64_t)3;
			if (avail & 2u)
				lost = (unsigned)want;
		}
		const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)from);
		const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(from >> 32));
		const unsigned rec = __builtin_amdgcn_readfirstlane((unsigned)want);
		const __amdgpu_buffer_rsrc_t rsrc =
		    __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((uint64_t)hi << 32) | lo), 0, (int)rec, 0x00020000);
#pragma unroll
		for (int v = 0; v < NVW; ++v)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(my_buf + v * 1024u), 16,
			                                         (int)(v * 1024u + lane * 16u), 0, 0, 0);
		return (unsigned)__builtin_amdgcn_readfirstlane((int)lost);
	};
	auto patch = [&](unsigned lost) {
		if (lost != ~0u)
		{
			if (lane == 0)
				*reinterpret_cast<short *>(my_buf + lost) = *(const __attribute__((address_space(1))) short *)(in_end - 2u);   // (a GLOBAL pointer by type: not a flat load)
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		}
	};

	// `next` is known a tile ahead: its ticket is drawn at the END of the previous tile, the atomic's round trip running under the
	// wait for that tile's DMA (a draw that is waited for on the spot cost the short mono tiles 7 %: profiles/r03_kint_tickets.log)
	// (`next_group`: the group after the one in progress; within a group the tiles follow each other)
	uint64_t tile = global_wave * G;
	unsigned in_group = 1;
	unsigned lost = fetch(tile);
	uint64_t next_group = ticketed ? resolve(draw_ticket(lane_counter)) : (global_wave + global_waves < n_groups ? global_wave + global_waves : ~0ull);
	uint64_t next = (G > 1u && tile + 1u < n_tiles) ? tile + 1u : (next_group != ~0ull ? next_group * G : ~0ull);
	if (G > 1u && tile + 1u >= n_tiles)
		in_group = G;   // (the stream's last group is a short one)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

	for (;;)
	{
		const uint64_t first = tile * WT;
		const unsigned n = (unsigned)((a.n_out - first < WT) ? (a.n_out - first) : WT);

		patch(lost);
		// the lane's window, packed, into registers: XV x 16 bytes at a stride of LANE_BYTES, in aligned reads of VECB bytes
		int d[XV * 4];
		{
			if constexpr (VECB == 16u)
			{
				const i32x4 *src = reinterpret_cast<const i32x4 *>(my_buf + lane * LANE_BYTES);
				i32x4 raw[XV];
#pragma unroll
				for (int v = 0; v < XV; ++v)
					raw[v] = src[v];
#pragma unroll
				for (int v = 0; v < XV; ++v)
				{
					d[4 * v] = raw[v].x;
					d[4 * v + 1] = raw[v].y;
					d[4 * v + 2] = raw[v].z;
					d[4 * v + 3] = raw[v].w;
				}
			}
			else if constexpr (VECB == 8u)
			{
				const i32x2 *src = reinterpret_cast<const i32x2 *>(my_buf + lane * LANE_BYTES);
				i32x2 raw[XV * 2];
#pragma unroll
				for (int v = 0; v < XV * 2; ++v)
					raw[v] = src[v];
#pragma unroll
				for (int v = 0; v < XV * 2; ++v)
				{
					d[2 * v] = raw[v].x;
					d[2 * v + 1] = raw[v].y;
				}
			}
			else
			{
				const int *src = reinterpret_cast<const int *>(my_buf + lane * LANE_BYTES);
#pragma unroll
				for (int v = 0; v < XV * 4; ++v)
					d[v] = src[v];
			}
		}
		// the reads must have landed before the next DMA may overwrite the buffer (the registers are the compiler's, hence its
		// builtin: it keeps its own loads above it)
		__builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)
		asm volatile("" ::: "memory");

		// the next tile's window into the same buffer, under this tile's arithmetic
		const bool have_next = next != ~0ull;
		if (have_next)
			lost = fetch(next);

		// accumulator high dwords (the sums): [live frame][channel][0: slots with weights >= 0, 1: slots with weights <= 0].  At any
		// time TT / R frames of a lane are in progress; frame k uses set k % LIVE, pinned to physical registers (int_tap).
		// (OS - periodic ratios, long windows: the other loop order below - one frame after the other, two accumulator sets in turn)
		constexpr int LIVE = OS ? 2 : TT / R;
		static_assert(OS || (TT % R == 0 && LIVE * 2 * 2 <= 24), "pinned accumulator pairs");
		int acc[LIVE][2][2], arm[LIVE][2][2];
		const unsigned stage_at = lane * (K * UNIT);

		// X = 2 * sample, sign-extended: one SDWA shift per sample, issued ONE input frame ahead of its taps (hipcc pads an asm
		// statement whose result the very next instruction reads with an s_nop: 132 per tile)
		int xs[(NX + 1) * CH];
		auto unpack = [&](auto w_tag) {
			constexpr int word = decltype(w_tag)::value;
			if constexpr (word < NX * CH)
			{
				// (named first: asm operands alone do not make a generic lambda capture for clang)
				int &to = xs[word];
				const int from = d[word >> 1];
				if constexpr (word & 1)
					asm("v_lshlrev_b32_sdwa %0, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(to) : "v"(from), "v"(1));
				else
					asm("v_lshlrev_b32_sdwa %0, %2, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(to) : "v"(from), "v"(1));
			}
		};

		// More than two channels: the frame's channel PAIRS one after the other over the same window registers and the same 24
		// accumulator pairs - a 6-channel 2:1 instance is the stereo 6:1 body's size (K x TT x CH taps either way).
		constexpr int PASSES = (CH + 1) / 2;
		static_for<PASSES>([&](auto p_tag) {
			constexpr int first_ch = 2 * decltype(p_tag)::value;
			constexpr int PC = first_ch + 1 < CH ? 2 : 1;        // channels of this pass
			// a finished frame k (compile-time) of this pass: normalise (clownresampler.h:1025-1033) and stage
			auto finish = [&](auto k_tag) {
				constexpr int k = decltype(k_tag)::value;
				int out[2] = {0, 0};
				static_for<PC>([&](auto c_tag) {
					constexpr int c = decltype(c_tag)::value;
					constexpr bool any_neg = (NEGMASK & ~ZEROMASK & (((1ull << TT) - 1ull) << ((k % P) * TT))) != 0;   // (of this frame's phase)
					int sum = acc[k % LIVE][c][0];
					if constexpr (any_neg)
						sum -= acc[k % LIVE][c][1];
					// (acc * reciprocal) / 32768 toward zero in 64 bits: right for either range class of the host's (CRHIP_NORM_*)
					const long long v = (long long)sum * (long long)a.reciprocal[k % P] + (long long)((unsigned)(sum >> 31) >> 17);
					out[c] = (int)(v >> 15);
				});
				if ((unsigned)k + lane * K < n)
				{
					unsigned char *at = my_stage + stage_at + k * UNIT + first_ch * (OUT16 ? 2 : 4);
					if constexpr (OUT16)
					{
						if constexpr (PC == 2 && CH % 2 == 0)
							*reinterpret_cast<int *>(at) = (clamp_s16(out[0]) & 0xFFFF) | (clamp_s16(out[1]) << 16);
						else
						{
							// (the frames of an odd channel count start on 2-byte boundaries every other time)
							reinterpret_cast<short *>(at)[0] = (short)clamp_s16(out[0]);
							if constexpr (PC == 2)
								reinterpret_cast<short *>(at)[1] = (short)clamp_s16(out[1]);
						}
					}
					else if constexpr (PC == 2 && CH % 2 == 0)
					{
						i32x2 q;
						q.x = out[0];
						q.y = out[1];
						*reinterpret_cast<i32x2 *>(at) = q;
					}
					else
					{
						reinterpret_cast<int *>(at)[0] = out[0];
						if constexpr (PC == 2)
							reinterpret_cast<int *>(at)[1] = out[1];
					}
				}
			};

			if constexpr (OS)
			{
				// PERIODIC ratios (and long windows): output-stationary.  The lane's whole window is unpacked first (it is short: a few periods + one
				// row's slots), then the lane's frames one after the other - frame k is phase k % P, its window starts int_start(k)
				// frames into the lane's, its weights are that phase's row at a.w[(k % P) TT ...] - on two accumulator sets in turn.
				static_for<NX>([&](auto i_tag) {
					static_for<PC>([&](auto c_tag) { unpack(std::integral_constant<int, decltype(i_tag)::value * CH + first_ch + decltype(c_tag)::value>()); });
				});
				static_for<K>([&](auto k_tag) {
					constexpr int k = decltype(k_tag)::value;
					constexpr int ph = k % P, at = int_start(R, P, OFFS, k);
					static_for<TT>([&](auto s_tag) {
						constexpr int sl = decltype(s_tag)::value;
						static_for<PC>([&](auto c_tag) {
							constexpr int c = decltype(c_tag)::value;
							constexpr int bit = ph * TT + sl;
							constexpr int cls = (int)((NEGMASK >> bit) & 1ull);
							constexpr unsigned long long phase_bits = ((1ull << TT) - 1ull) << (ph * TT), before = ((1ull << bit) - 1ull) & phase_bits;
							constexpr bool first_of_class = cls ? ((NEGMASK & ~ZEROMASK & before) == 0) : ((~NEGMASK & ~ZEROMASK & before) == 0);
							constexpr int PAIR = ((k % LIVE) * 2 + c) * 2 + cls;
							if constexpr (!((ZEROMASK >> bit) & 1ull))
							{
								const int x = xs[(at + sl) * CH + first_ch + c];
								int &hi = acc[k % LIVE][c][cls], &lo = arm[k % LIVE][c][cls];
								if constexpr (first_of_class)
									hi = 0;
								lo = x;
								if constexpr ((SAFEMASK >> bit) & 1ull)
									int_tap<PAIR>(lo, hi, (int)((unsigned)x << 15), a.w[bit]);
								else
									int_tap<PAIR>(lo, hi, x, a.w[bit]);
							}
						});
					});
					finish(k_tag);
				});
			}
			else
			{
			static_for<PC>([&](auto c_tag) { unpack(std::integral_constant<int, first_ch + decltype(c_tag)::value>()); });

			static_for<NX>([&](auto i_tag) {
				constexpr int i = decltype(i_tag)::value;
				static_for<PC>([&](auto c_tag) { unpack(std::integral_constant<int, (i + 1) * CH + first_ch + decltype(c_tag)::value>()); });
				static_for<PC>([&](auto c_tag) {
					constexpr int c = decltype(c_tag)::value;
					const int x = xs[i * CH + first_ch + c];
					static_for<K>([&](auto k_tag) {
						constexpr int k = decltype(k_tag)::value;
						constexpr int s = i - R * k;
						if constexpr (s >= 0 && s < TT)
						{
							constexpr int cls = (int)((NEGMASK >> s) & 1ull);
							// is this the first slot of its class?
							constexpr unsigned long long before = (1ull << s) - 1ull;
							constexpr bool first_of_class = cls ? ((NEGMASK & before) == 0) : ((~NEGMASK & before) == 0);
							constexpr int PAIR = ((k % LIVE) * 2 + c) * 2 + cls;
							int &hi = acc[k % LIVE][c][cls], &lo = arm[k % LIVE][c][cls];
							if constexpr (first_of_class)
								hi = 0;
							lo = x;
							if constexpr ((SAFEMASK >> s) & 1ull)
								int_tap<PAIR>(lo, hi, (int)((unsigned)x << 15), a.w[s]);
							else
								int_tap<PAIR>(lo, hi, x, a.w[s]);
						}
					});
				});
				static_for<K>([&](auto k_tag) {
					constexpr int k = decltype(k_tag)::value;
					if constexpr (i - R * k == TT - 1)
						finish(k_tag);
				});
			});
			}
		});

		// the staged frames of the other lanes: same wave, LDS operations of a wave complete in order
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// copy-out: the tile's n frames are contiguous in the output; wave-uniform base, one lane offset: 16 bytes per lane (dword
		// alignment is all gfx950 asks of global_store_dwordx4), then the tail by dwords and - int16 mono - one last sample
		unsigned stores = 0;
		{
			const unsigned bytes = n * UNIT;
			const unsigned vectors = bytes / 16u, dwords = bytes / 4u;
			const uint64_t out_first = reinterpret_cast<uint64_t>(a.d_out) + first * UNIT;
			const unsigned out_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)out_first);
			const unsigned out_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(out_first >> 32));
			// (GLOBAL pointers by type: through generic ones these would be flat stores, which go down the LDS path as well)
			typedef __attribute__((address_space(1))) i32x4 global_vec;
			typedef __attribute__((address_space(1))) int global_int;
			typedef __attribute__((address_space(1))) short global_short;
			global_vec *dst = (global_vec *)(((uint64_t)out_hi << 32) | out_lo);
			const i32x4 *staged = reinterpret_cast<const i32x4 *>(my_stage);
			auto put = [&](global_vec *to, i32x4 v) {
				if constexpr (NT)
					__builtin_nontemporal_store(v, to);
				else
					*to = v;
			};
			unsigned done = 0;   // wave-uniform
			for (; done + 128u <= vectors; done += 128u)
			{
				const i32x4 v0 = staged[done + lane], v1 = staged[done + 64u + lane];
				put(dst + done + lane, v0);
				put(dst + done + 64u + lane, v1);
				stores += 2u;
			}
			for (; done < vectors; done += 64u)
			{
				if (done + lane < vectors)
					put(dst + done + lane, staged[done + lane]);
				stores += 1u;
			}
			if (dwords > 4u * vectors)
			{
				if (4u * vectors + lane < dwords)
					((global_int *)dst)[4u * vectors + lane] = reinterpret_cast<const int *>(my_stage)[4u * vectors + lane];
				stores += 1u;
			}
			if (bytes > 4u * dwords)
			{
				if (lane == 0)
					((global_short *)dst)[2u * dwords] = reinterpret_cast<const short *>(my_stage)[2u * dwords];
				stores += 1u;
			}
			__builtin_amdgcn_wave_barrier();
		}

		if (!have_next)
			break;
		// the DMA was issued before this tile's stores and vmcnt retires in order: the stores stay in flight.  The ticket for the
		// tile after the next one travels meanwhile (nothing but our own waits between the two halves of the draw).
		// (`next` was the tile being fetched; what comes after it: the next tile of its group, or the first of the following group -
		// whose own successor is drawn now, when the wave enters a new group)
		const bool entering = in_group == G;       // `next` is the first tile of next_group
		if (entering && ticketed)
		{
			unsigned ticket = draw_ticket_begin(lane_counter);
			wait_vmcnt_at_most(stores);
			ticket = draw_ticket_end(ticket);
			next_group = resolve(ticket);
		}
		else
		{
			wait_vmcnt_at_most(stores);
			if (entering)
				next_group = (next / G) + global_waves < n_groups ? (next / G) + global_waves : ~0ull;
		}
		tile = next;
		in_group = entering ? 1u : in_group + 1u;
		next = (in_group < G && tile + 1u < n_tiles) ? tile + 1u : (next_group != ~0ull ? next_group * G : ~0ull);
		if (in_group < G && tile + 1u >= n_tiles)
			in_group = G;   // (the stream's last group is a short one)
	}
	retire();
}

} // namespace

#endif // CR_KINT_HPP
